"""The hand-scheduled loop of the banded fill kernel (pagan2-msa_amd/csrc/dp_pipe_hot.inc): the committed text is what
tools/gen_hot_asm.py generates, and it passes the static hazard / clobber checks of tools/check_hot_asm.py (inline asm
gets no wait states from the assembler).  No GPU: text only; the arithmetic is covered by the -m gpu parity tests."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_loop_is_the_generators_output(tmp_path):
    out = tmp_path / "hot.inc"
    env = dict(os.environ, PG_HOT_OUT=str(out))
    env.pop("PG_HOT_EXP", None)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_hot_asm.py")], check=True, env=env, capture_output=True)
    with open(os.path.join(ROOT, "pagan2-msa_amd", "csrc", "dp_pipe_hot.inc")) as f:
        assert f.read() == out.read_text()


def test_loop_passes_the_static_checks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_hot_asm.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_the_checker_sees_a_mask_read_too_early(tmp_path):
    """a compare followed directly by the select that reads its mask must be reported"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_hot_asm as chk
    inc = tmp_path / "bad.inc"
    inc.write_text('"v_cmp_gt_f64_e64 s[52:53], v[0:1], v[2:3]\\n\\t"\n"v_cndmask_b32_e64 v4, 1, 2, s[52:53]\\n\\t"\n')
    old = chk.INC
    chk.INC = str(inc)
    try:
        assert chk.main() == 1
    finally:
        chk.INC = old
