"""No scratch memory on the per-diagonal paths of the banded fill kernel (pg_fill_pipe, dp_pipe.hip).

The fill is a chain of ~2e5 dependent anti-diagonals whose speed is the latency of one step: a register the compiler
spills inside a per-diagonal loop costs a scratch round trip (and an s_waitcnt vmcnt(0) behind the wave's stores) on that
chain -- round 4 found one by accident in front of the assist waves' staging stores.  This test compiles
dp_pipe.hip to gfx950 assembly (no GPU needed: tools/check_scratch.py) and pins down where scratch is touched:

  * the functions that carry the per-diagonal loops -- hot_run (its loop is the generated assembly), wide_run, the assist
    waves' functions, the strip feeder, the follower workgroups -- have NO scratch access inside any loop: what they have
    is prologue / epilogue saves of callee-saved registers and the WaveCtx hand-over, once per call;
  * the large-table kernels (their per-diagonal step is compiled C++ in the kernel body) spill no VGPR at all;
  * the small-table kernels' bodies (general steps, set-up, the hand-over to hot_run / wide_run) stay within a recorded
    bound, so that a change that makes the compiler spill more is seen.

The kernels' .private_segment_fixed_size cannot be zero: the run functions are functions of their own (so that their
register allocation does not depend on the rest of the kernel) and take the wave's state through an object in scratch."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# mangled-name fragments of the functions with per-diagonal loops
LOOP_FUNCS = ["7hot_runILb0", "7hot_runILb1", "8wide_run", "16pipe_assist_leanILb0", "16pipe_assist_leanILb1", "11pipe_assistILb0ELb0",
              "11pipe_assistILb0ELb1", "19assist_general_diag", "20assist_general_cells", "12strip_feeder", "12follow_chunk", "9wide_run7", "15assist_wide_run",
              "13pipe_follower", "11widest_step"]


@pytest.fixture(scope="module")
def report():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scratch.py"), "--json"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_no_scratch_inside_the_loops_of_the_run_functions(report):
    funcs = report["functions"]
    for frag in LOOP_FUNCS:
        hits = [k for k in funcs if frag in k]
        for k in hits:                               # (a function without any scratch access is not listed at all)
            assert funcs[k]["in_loop"] == 0, "%s: scratch inside a loop: %s" % (k, funcs[k]["in_loop_lines"][:5])


def test_kernel_spill_counts(report):
    ker = report["kernels"]
    names = {k: v for k, v in ker.items() if "pg_fill_pipe" in k}
    assert len(names) == 4, sorted(ker)
    for k, v in names.items():
        big_table = "ILb0E" in k                     # pg_fill_pipe<false, .>: the compiled C++ step is the per-diagonal path
        if big_table:
            assert v["vgpr_spill_count"] == 0, (k, v)
        else:
            # bodies of the small-table kernels: general steps and set-up only (round 5: 11 / 43 VGPRs -- the strips' body with its
            # written-through score stores)
            assert v["vgpr_spill_count"] <= 48, (k, v)
        assert v["private_segment_fixed_size"] <= 1536, (k, v)
