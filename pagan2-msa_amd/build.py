"""Builds libpagan_dp.so (HIP, gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

The library is rebuilt whenever the digest of its inputs (compiler flags + every source and header it is made
from) differs from the one recorded beside it at the last build, so a stale or differently-flagged binary
(tools/build_stamps.sh writes its diagnostic build to another file for the same reason) is never taken for
the product."""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpagan_dp.so")
SOURCES = ["dp_abi.hip", "dp_kernels.hip", "dp_pipe.hip", "dp_tiles.hip", "dp_fb.hip", "dp_anchors.hip", "dp_parent.hip", "host_model.cpp", "host_graph.cpp",
           "host_anchors.cpp", "host_tree.cpp", "host_pileup.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-result", "-Wno-unused-value", "-pthread"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpagan_dp.so cannot be built")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def inputs():
    inc = os.path.join(HERE, "..", "include")
    # every text file of csrc/ takes part (headers, the generated dp_pipe_hot.inc that dp_pipe.hip includes, sources
    # not in SOURCES yet): a file the compiler may read must never change without the digest changing
    listed = set(sources())
    extra = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                   if f.endswith((".h", ".inc", ".hip", ".cpp", ".hpp")) and os.path.join(CSRC, f) not in listed)
    return sources() + extra + sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h"))


def digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in inputs():
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stale(lib=LIB):
    if not os.path.exists(lib) or not os.path.exists(lib + ".digest"):
        return True
    with open(lib + ".digest") as f:
        return f.read().strip() != digest()


def build(force=False, verbose=False, out=LIB):
    """Compiles every source into `out`.  Returns the path."""
    if not force and not stale(out):
        return out
    cmd = [_hipcc()] + FLAGS + ["-o", out] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(out + ".digest", "w") as f:
        f.write(digest() + "\n")
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
