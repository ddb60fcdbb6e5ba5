"""The HIP library builds from clean: every source is compiled for gfx950 into a temporary path (hipcc
cross-compiles without a GPU) and the result exports every symbol the headers declare.  Guards against a
shipped binary that no longer corresponds to the tree."""
import ctypes as C
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_module():
    spec = importlib.util.spec_from_file_location("_pagan_build", os.path.join(ROOT, "pagan2-msa_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def declared_symbols():
    names = set()
    for h in ("pagan_dp.h", "pagan_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(pagan_[a-z0-9_]+)\s*\(", text))
    return names - {"pagan_batch_fn"}


def test_library_builds_from_clean_and_exports_the_abi(tmp_path):
    mod = _build_module()
    out = mod.build(force=True, out=str(tmp_path / "libpagan_dp_fresh.so"))
    assert os.path.getsize(out) > 100000
    lib = C.CDLL(out)
    missing = [s for s in sorted(declared_symbols()) if not hasattr(lib, s)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    assert not mod.stale(out)


def test_in_tree_library_matches_the_sources():
    mod = _build_module()
    assert not mod.stale(), "pagan2-msa_amd/libpagan_dp.so was not built from the current sources: run __graft_entry__.build()"
    lib = C.CDLL(mod.LIB)
    missing = [s for s in sorted(declared_symbols()) if not hasattr(lib, s)]
    assert not missing, missing


def test_generated_hot_loop_takes_part_in_the_digest(tmp_path, monkeypatch):
    """dp_pipe.hip includes dp_pipe_hot.inc: regenerating it must make the shipped library stale (round-2 advisor finding)."""
    mod = _build_module()
    names = [os.path.basename(p) for p in mod.inputs()]
    assert "dp_pipe_hot.inc" in names and "dp_pipe.hip" in names and "pagan_dp.h" in names
    assert len(names) == len(set(names))
    # the same check on a copy of csrc/ with one byte of the .inc changed
    import shutil
    csrc = tmp_path / "csrc"
    shutil.copytree(mod.CSRC, csrc)
    before = mod.digest()
    monkeypatch.setattr(mod, "CSRC", str(csrc))
    assert mod.digest() == before
    with open(csrc / "dp_pipe_hot.inc", "a") as f:
        f.write("\n")
    assert mod.digest() != before
