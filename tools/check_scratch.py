#!/usr/bin/env python3
"""Where does the banded fill kernel touch scratch memory?

Compiles dp_pipe.hip to gfx950 assembly (device only, the product's flags) and lists, per function, every scratch_load /
scratch_store together with whether it sits INSIDE A LOOP of that function (between a label and a later branch back to it) --
i.e. whether it can be on a per-diagonal path -- or in straight-line code that runs once per call (prologue / epilogue saves of
callee-saved registers, the WaveCtx hand-over of hot_run / wide_run, the kernel's set-up).  A register spilled inside the
per-diagonal loops costs a scratch round trip per anti-diagonal on the dependency chain ("a spill on the
chain"); scratch outside them costs nothing measurable.

    python tools/check_scratch.py            # table per function
    python tools/check_scratch.py --json     # the same as JSON (tests/test_scratch_cpu.py reads it)

The kernels' .private_segment_fixed_size is not zero and cannot be: hot_run / wide_run / the assist functions are functions of
their own (their register allocation must not depend on the rest of the kernel), the compute waves' state travels to them
through a WaveCtx object in scratch once per RUN, and the AMDGPU calling convention saves callee-saved VGPRs in the prologue.
What this tool (and the test) pins down is that none of it is inside a loop."""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "pagan2-msa_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-device-only", "-S"]

FUNC = re.compile(r"^([A-Za-z_][A-Za-z0-9_$.]*):\s*(;.*)?$")
LABEL = re.compile(r"^(\.L[A-Za-z0-9_$.]*):")
BRANCH = re.compile(r"^\s*s_(?:c?branch[a-z0-9_]*)\s+(\.L[A-Za-z0-9_$.]*)")
SCRATCH = re.compile(r"^\s*(scratch_(?:load|store)[a-z0-9_]*)")


def assemble(source="dp_pipe.hip", extra=()):
    out = os.path.join(tempfile.mkdtemp(prefix="pg_isa_"), "out.s")
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    subprocess.run([hipcc] + FLAGS + list(extra) + ["-o", out, source], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    try:
        p = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True, check=True)
        return dict(zip(names, p.stdout.strip().split("\n")))
    except Exception:
        return {n: n for n in names}


def analyse(path):
    """{function: {"scratch": n, "in_loop": n, "in_loop_lines": [...]}} and the kernels' metadata"""
    funcs, order, cur = {}, [], None
    meta = {}
    with open(path) as f:
        lines = f.read().split("\n")
    for no, line in enumerate(lines):
        m = FUNC.match(line)
        if m and not line.startswith(".") and not line.startswith("\t"):
            cur = m.group(1)
            if cur not in funcs:
                funcs[cur] = {"lines": []}
                order.append(cur)
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None if line.startswith("\t.end_amdhsa_kernel") else cur
        if cur is not None:
            funcs[cur]["lines"].append((no + 1, line))
    # metadata block
    name = None
    for line in lines:
        s = line.strip()
        if s.startswith(".name:"):
            name = s.split(":", 1)[1].strip()
            meta.setdefault(name, {})
        elif name and s.startswith((".private_segment_fixed_size:", ".sgpr_spill_count:", ".vgpr_spill_count:", ".vgpr_count:", ".sgpr_count:")):
            k, v = s.split(":")
            meta[name][k.strip(". ")] = int(v)
    res = {}
    for fn in order:
        body = funcs[fn]["lines"]
        label_at = {}
        for idx, (no, line) in enumerate(body):
            m = LABEL.match(line)
            if m:
                label_at[m.group(1)] = idx
        loops = []
        for idx, (no, line) in enumerate(body):
            m = BRANCH.match(line)
            if m and m.group(1) in label_at and label_at[m.group(1)] <= idx:
                loops.append((label_at[m.group(1)], idx))
        n = 0
        inside = []
        for idx, (no, line) in enumerate(body):
            m = SCRATCH.match(line)
            if not m:
                continue
            n += 1
            if any(a <= idx <= b for a, b in loops):
                inside.append("%d: %s" % (no, line.strip()))
        if n:
            res[fn] = {"scratch": n, "in_loop": len(inside), "in_loop_lines": inside}
    return res, meta


def main():
    path = assemble()
    res, meta = analyse(path)
    names = demangle(list(res))
    out = {"functions": {names[k]: v for k, v in res.items()}, "kernels": {demangle([k])[k]: v for k, v in meta.items() if v}}
    if "--json" in sys.argv:
        print(json.dumps(out))
        return
    print("%-90s %8s %8s" % ("function", "scratch", "in loops"))
    for k, v in sorted(out["functions"].items(), key=lambda kv: -kv[1]["scratch"]):
        print("%-90s %8d %8d" % (k[:90], v["scratch"], v["in_loop"]))
        for l in v["in_loop_lines"][:40]:
            print("      " + l)
    for k, v in out["kernels"].items():
        print(k[:70], v)


if __name__ == "__main__":
    main()
