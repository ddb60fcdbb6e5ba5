"""The LDS-ring forward/backward sweeps (dp_fb.hip: pg_fb_forward_ring / pg_fb_backward_ring) keep memory loads off a step's path.

A step of these sweeps is a latency chain (2e5 of them for a tunnel of 2 x 100 kb), and vector memory operations complete in order:
a load on the step's path waits behind the wave's own stores.  The first version of the kernels had two such loads that only the
disassembly showed ("tunnels"): a uniform `J.imin[d]` compiled as a flat VECTOR load, and
`tab_lds ? lds_table[k] : memory_table[k]` compiled as ONE flat load of a selected pointer.  This test compiles dp_fb.hip to gfx950
assembly (no GPU needed) and pins down, for the instantiations whose score table is in LDS (`ALL_LDS`):

  * no scratch (.private_segment_fixed_size 0) in any ring kernel;
  * no flat load of a selected LDS-or-memory pointer (`src_shared_base`) anywhere in them;
  * inside the sweep's loop over the diagonals, memory LOADS appear only in the staging section (before the loop's first
    `s_barrier`, every FB_RG_REFILL diagonals), none between that barrier and the step's own."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pagan2-msa_amd", "csrc")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("fbasm") / "dp_fb.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-device-only", "-S", "-o", out, "dp_fb.hip"],
                   check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    return open(out).read()


def functions(asm):
    """{mangled name: body lines} of the ring kernels"""
    out = {}
    for m in re.finditer(r"^(_ZN[^\n:]*pg_fb_(?:forward|backward)_ringILb([01])ELi([123])E[^\n:]*):", asm, re.M):
        end = asm.index("s_endpgm", m.end())
        out[m.group(1)] = (m.group(2) == "1", int(m.group(3)), asm[m.end():end].splitlines())
    return out


def test_ring_kernels_have_no_scratch(asm):
    sizes = re.findall(r"\.name:\s+(\S*pg_fb_\w+_ring\S*)\s+\.private_segment_fixed_size:\s+(\d+)", asm)
    assert len(sizes) == 16, sizes                                   # forward / backward x ALL_LDS x (NSPLIT 1, 2, 3 at 512 rows; 1 at 1,024)
    assert all(int(sz) == 0 for _name, sz in sizes), sizes


def test_no_memory_load_on_a_step_of_the_all_lds_sweeps(asm):
    fs = functions(asm)
    assert len(fs) == 16
    checked = 0
    for name, (all_lds, nsplit, lines) in fs.items():
        if not all_lds:
            continue
        assert not any("src_shared_base" in ln for ln in lines), name
        # the sweep's loop: from the first depth-1 loop header that contains an s_barrier to the last s_barrier before the epilogue
        barriers = [k for k, ln in enumerate(lines) if ln.strip() == "s_barrier"]
        assert len(barriers) >= 4, (name, len(barriers))            # set-up, (rotated) step end, staging, step end, epilogue
        # the staging section's barrier is the one preceded by global / flat loads since the barrier before it; the step section
        # runs from it to the next barrier and holds no load from memory
        load = re.compile(r"\b(global_load|flat_load|buffer_load|scratch_load)")
        staging = [b for a, b in zip(barriers, barriers[1:]) if any(load.search(ln) for ln in lines[a:b])]
        assert staging, name
        stage_end = staging[0]
        step_end = next(b for b in barriers if b > stage_end)
        step = lines[stage_end:step_end]
        assert len(step) > 200, (name, len(step))                    # (the arithmetic of a cell is in there)
        assert not any(load.search(ln) for ln in step), (name, [ln for ln in step if load.search(ln)][:3])
        assert not any("vmcnt" in ln for ln in step), (name, [ln for ln in step if "vmcnt" in ln][:3])
        checked += 1
    assert checked == 8
