#!/usr/bin/env python3
"""Static checks of pagan2-msa_amd/csrc/dp_pipe_hot.inc (the assembler inserts no wait states in inline asm):

  1. a VALU instruction that reads a lane mask (SGPR pair or vcc) written by a VALU compare comes at least TWO
     instructions after that compare (gfx940+: VALU-writes-SGPR -> VALU-reads-SGPR needs two wait states);
  2. a DPP move does not read a VGPR written by one of the two instructions before it;
  3. every register the text names is in the asm statement's clobber list (dp_pipe.hip) or an operand;
  4. a DPP move comes at least FIVE instructions after a scalar write of exec (SALU-writes-EXEC -> VALU DPP: five wait
     states; the class 1 blocks run under reduced exec masks since round 5).

Straight-line scan: a label resets the window (the code behind a branch target is checked from there on, and the
generator places nothing mask-dependent directly behind a label).  Exit code 1 on a finding."""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
INC = os.path.join(HERE, "..", "pagan2-msa_amd", "csrc", "dp_pipe_hot.inc")
HIP = os.path.join(HERE, "..", "pagan2-msa_amd", "csrc", "dp_pipe.hip")


def instructions():
    out = []
    for line in open(INC):
        m = re.match(r'^"(.*)\\n\\t"$', line.strip())
        if not m:
            continue
        t = m.group(1).strip()
        if not t or t.startswith(";"):
            continue
        out.append(t)
    return out


def sregs(tok):
    """SGPR numbers named by a token like s[52:53], s68, vcc"""
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return {"s%d" % k for k in range(int(m.group(1)), int(m.group(2)) + 1)}
    if re.fullmatch(r"s\d+", tok):
        return {tok}
    if tok == "vcc":
        return {"vcc"}
    return set()


def vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return {"v%d" % k for k in range(int(m.group(1)), int(m.group(2)) + 1)}
    if re.fullmatch(r"v\d+", tok):
        return {tok}
    return set()


def operands(t):
    body = t.split(None, 1)[1] if " " in t else ""
    body = re.sub(r"\b(offset\d*|row_mask|bank_mask|wave_shr|lgkmcnt|vmcnt)\S*", "", body)
    return [x.strip() for x in body.split(",") if x.strip()]


def main():
    ins = instructions()
    bad = 0
    since_exec = 99       # instructions since the last scalar write of exec
    pending = []          # (age, set of mask registers written by a VALU compare)
    recent_vwrites = []   # VGPRs written by the last two instructions
    named = set()
    for t in ins:
        if t.endswith(":"):
            pending = []
            recent_vwrites = []
            continue                                  # (since_exec carries over a label: fall-through paths count)
        op = t.split()[0]
        ops = operands(t)
        for o in ops:
            named |= sregs(o) | vregs(o)
        is_valu = op.startswith("v_")
        # --- rule 1 ---
        if is_valu and not op.startswith("v_cmp") and not op.startswith("v_readfirstlane"):
            reads = set()
            for o in ops[1:]:
                reads |= sregs(o)
            if op.endswith("_e32") and op.startswith("v_cndmask"):
                reads.add("vcc")
            for age, regs in pending:
                if age < 2 and reads & regs:
                    print("mask read %d instruction(s) after the compare that writes it: %s" % (age, t))
                    bad += 1
        # --- rule 4 ---
        if "_dpp" in op and since_exec < 5:
            print("DPP %d instruction(s) after a scalar write of exec: %s" % (since_exec, t))
            bad += 1
        since_exec = 0 if (op.startswith("s_") and ("saveexec" in op or (ops and ops[0] == "exec"))) else since_exec + 1
        # --- rule 2 ---
        if "_dpp" in op:
            src = vregs(ops[1]) if len(ops) > 1 else set()
            for w in recent_vwrites:
                if src & w:
                    print("DPP source written within two instructions: %s" % t)
                    bad += 1
        # advance the windows
        pending = [(age + 1, regs) for age, regs in pending if age + 1 < 3]
        if op.startswith("v_cmp"):
            dst = sregs(ops[0]) if ops and not op.endswith("_e32") else {"vcc"}
            pending.append((0, dst))
        w = vregs(ops[0]) if is_valu and ops and not op.startswith("v_cmp") and not op.startswith("v_readfirstlane") else set()
        recent_vwrites = ([w] + recent_vwrites)[:2]
    # --- rule 3 ---
    hip = open(HIP).read()
    a = hip.index('#include "dp_pipe_hot.inc"')
    b = hip.index(");", a)
    clob = set(re.findall(r'"([vs]\d+|vcc|scc)"', hip[a:b]))
    if "PG_HOT_CLOBBERS" in hip[a:b]:
        # the macro in front of hot_run: every VGPR from v136 up, the SGPRs s36 .. s87 (its expansion is checked here too)
        m = re.search(r"#define PG_HOT_CLOBBERS(.*?)\n\n", hip, re.S)
        text = m.group(1)
        for pre, lo in re.findall(r"PG_([VS])8\((\d+)\)", text):
            clob |= {"%s%d%d" % (pre.lower(), int(lo), k) for k in range(10)}
        clob |= set(re.findall(r'"([vs]\d+)"', text))
    missing = sorted(x for x in named if x not in clob and x != "vcc")
    if missing:
        print("registers named by the loop but not clobbered:", " ".join(missing))
        bad += 1
    print("%d instructions checked, %d finding(s)" % (len([t for t in ins if not t.endswith(':')]), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
