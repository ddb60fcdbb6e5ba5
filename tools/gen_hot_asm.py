#!/usr/bin/env python3
"""Generates pagan2-msa_amd/csrc/dp_pipe_hot.inc: the text of the inline-asm loop that runs consecutive class 0
anti-diagonals of dp_pipe.hip's compute waves (hot_run).  Written as a generator because the loop is unrolled by two
with the register sets' roles swapped (no copies in the steady state), and because symbolic names keep ~220 lines of
gfx950 assembly checkable.  Run it after changing the schedule:  python tools/gen_hot_asm.py

Register plan (all fixed registers are in the asm statement's clobber list; operands are %[name]):
  v[224:229]  P: this lane's cell of the previous diagonal, X Y M (what the step writes back, -inf outside the band)
  Q0 = v[180:185], Q1 = v[230:235]: the shifted cell (row-1, .) -- C of this step in one, A of this step in the other
  v[186:213]  temporaries of the candidates, v214..216 the three back-pointer words, v217..v223 addresses
  s[36:43] / s[44:51]  descriptor of this diagonal / of the next one (roles swap)
  s[52:63] compare masks, s[64:67] store bases, s68/s69 scratch, s[70:71] descriptor pointer
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pagan2-msa_amd", "csrc", "dp_pipe_hot.inc")

P = (224, 226, 228)                      # X, Y, M (register pairs start)
Q = ((180, 182, 184), (230, 232, 234))


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


def step(k):
    """One diagonal.  k = 0 / 1: which half of the unrolled pair (selects descriptor and Q roles)."""
    cur = 36 if k == 0 else 44
    nxt = 44 if k == 0 else 36
    A = Q[k]          # receives (row-1, j) on d-1
    C = Q[1 - k]      # holds (row-1, j-1) on d-2
    lo, hi, zlo, zhi, s4, s7 = cur, cur + 1, cur + 2, cur + 3, cur + 4, cur + 7
    L = []
    a = L.append
    a("; ---- diagonal, half %d ----" % k)
    # LDS batch: upstream flag, lane 0's operand, the column record two steps ahead, next step's model score, the row record
    a("ds_read_b32 v223, %[fup]")
    a("s_sub_i32 s68, %[sb], 0x1800")
    a("s_cmp_lt_i32 s68, 0")
    a("s_cselect_b32 s68, 0x15000, s68")                       # ring row of d-1
    a("v_add_u32_e32 v217, s68, %[bpos24]")
    a("ds_read2_b64 v[%d:%d], v217 offset1:1" % (A[0], A[0] + 3))
    a("ds_read_b64 %s, v217 offset:16" % pair(A[2]))
    a("s_add_i32 s69, %[d], 2")
    a("v_sub_u32_e32 v218, s69, %[row]")
    a("v_and_b32_e32 v218, 0x1ff, v218")
    a("v_lshl_add_u32 v218, v218, 4, %[bR]")
    a("v_and_b32_e32 v219, 0xffff, %[colx]")
    a("v_and_b32_e32 v220, 0xffff, %[rowx]")
    a("v_cvt_f64_f32_e32 v[186:187], %[sm]")                   # this step's model score, before its register is reloaded
    a("v_mad_u32_u24 v219, v219, %[S], v220")
    a("v_and_b32_e32 v219, 0xff, v219")
    a("v_lshl_add_u32 v219, v219, 2, %[bT]")
    a("ds_read_b32 %[colx], v218")
    a("ds_read_b32 %[sm], v219")
    a("v_and_b32_e32 v221, 0x1ff, %[row]")
    a("v_lshl_add_u32 v221, v221, 4, %[bL]")
    a("ds_read_b32 %[rowx], v221")
    # Y from P, M from C (no shift needed): in the reference's order, first wins
    a("v_add_f64 v[192:193], %s, %%[ge]" % pair(P[1]))        # y1 = PY + ge         -> Y | ADJR = 9
    a("v_add_f64 v[194:195], %s, %%[go]" % pair(P[0]))        # y2 = PX + go         -> X | ADJR = 8
    a("v_add_f64 v[196:197], %s, %%[ng]" % pair(P[2]))        # y3 = (PM + ng) + go  -> M | ADJR = 10
    a("v_add_f64 v[188:189], %[tng2], v[186:187]")             # tM
    a("v_add_f64 v[190:191], %[tng1], v[186:187]")             # tX
    a("v_add_f64 v[196:197], v[196:197], %[go]")
    a("v_add_f64 v[200:201], %s, v[188:189]" % pair(C[2]))     # m1 = CM + tM -> 14
    a("v_add_f64 v[202:203], %s, v[190:191]" % pair(C[0]))     # m2 = CX + tX -> 12
    a("v_add_f64 v[204:205], %s, v[190:191]" % pair(C[1]))     # m3 = CY + tX -> 13
    a("v_max_f64 v[198:199], v[194:195], v[196:197]")          # y23
    a("v_cmp_gt_f64_e64 s[52:53], v[196:197], v[194:195]")     # y3 > y2
    a("v_max_f64 v[206:207], v[202:203], v[204:205]")          # m23
    a("v_cmp_gt_f64_e64 s[54:55], v[204:205], v[202:203]")     # m3 > m2
    a("v_max_f64 v[210:211], v[192:193], v[198:199]")          # by
    a("v_cmp_gt_f64_e64 s[56:57], v[198:199], v[192:193]")     # y23 > y1
    a("v_max_f64 v[212:213], v[200:201], v[206:207]")          # bm
    a("v_cmp_gt_f64_e64 s[58:59], v[206:207], v[200:201]")     # m23 > m1
    a("v_cndmask_b32_e64 v215, 8, 10, s[52:53]")
    a("v_cndmask_b32_e64 v216, 12, 13, s[54:55]")
    a("v_cndmask_b32_e64 v215, 9, v215, s[56:57]")
    a("v_cndmask_b32_e64 v216, 14, v216, s[58:59]")
    # the LDS batch and the descriptor of this diagonal (requested a step ago) are here
    a("s_waitcnt lgkmcnt(0)")
    a("s_and_b32 s68, s%d, 15" % s4)
    a("s_cmp_lg_u32 s68, 0")
    a("s_cbranch_scc1 .Lpg_exit%d_%%=" % k)                    # not class 0
    a("s_cmp_ge_i32 %[d], %[sleep]")
    a("s_cbranch_scc1 .Lpg_exit%d_%%=" % k)
    a("s_cmp_gt_i32 %[d], %[okuntil]")
    a("s_cbranch_scc1 .Lpg_exit%d_%%=" % k)                    # the loader's flags have to be looked at
    a("s_cmp_gt_i32 s%d, %%[pdn]" % s7)
    a("s_cbranch_scc1 .Lpg_exit%d_%%=" % k)                    # the downstream wave's flag has to be looked at
    a("v_readfirstlane_b32 s68, v223")
    a("s_max_i32 %[pup], %[pup], s68")
    a("s_sub_i32 s69, %[d], 1")
    a("s_cmp_lt_i32 %[pup], s69")
    a("s_cbranch_scc0 .Lpg_upok%d_%%=" % k)
    # the upstream wave has not completed d-1 yet: it is, as a rule, a fraction of a step away.  Look again a few times
    # (flag first, then lane 0's operand: LDS executes in order) before handing the wait to the caller's poll.
    a("s_mov_b32 s72, 48")
    a(".Lpg_upretry%d_%%=:" % k)
    a("s_sleep 1")
    a("ds_read_b32 v223, %[fup]")
    a("ds_read2_b64 v[%d:%d], v217 offset1:1" % (A[0], A[0] + 3))
    a("ds_read_b64 %s, v217 offset:16" % pair(A[2]))
    a("s_waitcnt lgkmcnt(0)")
    a("v_readfirstlane_b32 s68, v223")
    a("s_max_i32 %[pup], %[pup], s68")
    a("s_cmp_lt_i32 %[pup], s69")
    a("s_cbranch_scc0 .Lpg_upok%d_%%=" % k)
    a("s_sub_i32 s72, s72, 1")
    a("s_cmp_lg_u32 s72, 0")
    a("s_cbranch_scc1 .Lpg_upretry%d_%%=" % k)
    a("s_branch .Lpg_exit%d_%%=" % k)
    a(".Lpg_upok%d_%%=:" % k)
    # the next descriptor: a whole step to land
    a("s_load_dwordx8 s[%d:%d], s[70:71], 0x20" % (nxt, nxt + 7))
    # row hand-over
    a("v_cmp_gt_i32_e32 vcc, s%d, %%[row]" % lo)
    a("v_add_u32_e32 v222, 0x100, %[row]")
    a("s_add_u32 s64, %%[sclo], s%d" % zlo)                    # score / back-pointer rows of this diagonal
    a("v_cndmask_b32_e32 %[row], %[row], v222, vcc")
    # shift: lane n takes lane n-1's cell, lane 0 keeps what it read from the ring
    for c in range(3):
        a("v_mov_b32_dpp v%d, v%d wave_shr:1 row_mask:0xf bank_mask:0xf" % (A[c], P[c]))
        a("v_mov_b32_dpp v%d, v%d wave_shr:1 row_mask:0xf bank_mask:0xf" % (A[c] + 1, P[c] + 1))
    a("s_addc_u32 s65, %%[schi], s%d" % zhi)
    a("s_lshr_b64 s[68:69], s[%d:%d], 1" % (zlo, zhi))
    # X from A
    a("v_add_f64 v[192:193], %s, %%[ge]" % pair(A[0]))        # x1 = AX + ge        -> X | ADJL = 4
    a("v_add_f64 v[194:195], %s, %%[go]" % pair(A[1]))        # x2 = AY + go        -> Y | ADJL = 5
    a("v_add_f64 v[196:197], %s, %%[ng]" % pair(A[2]))        # x3 = (AM + ng) + go -> M | ADJL = 6
    a("s_add_u32 s66, %[bplo], s68")
    a("s_addc_u32 s67, %[bphi], s69")
    a("v_cmp_ge_i32_e64 s[60:61], s%d, %%[row]" % hi)          # active: row <= hi
    a("v_add_f64 v[196:197], v[196:197], %[go]")
    a("v_subrev_u32_e32 v217, s%d, %%[row]" % lo)             # row - lo
    a("v_mul_u32_u24_e32 v218, 12, v217")
    a("v_max_f64 v[198:199], v[194:195], v[196:197]")          # x23
    a("v_cmp_gt_f64_e64 s[52:53], v[196:197], v[194:195]")
    a("v_lshlrev_b32_e32 v217, 1, v218")
    a("v_add_u32_e32 v219, %[sb], %[tid24]")
    a("v_max_f64 v[208:209], v[192:193], v[198:199]")          # bx
    a("v_cmp_gt_f64_e64 s[54:55], v[198:199], v[192:193]")
    a("v_cndmask_b32_e64 v214, 5, 6, s[52:53]")
    # results: -inf outside the band; a state that stayed -inf has no back-pointer
    a("v_cndmask_b32_e64 v%d, 0, v212, s[60:61]" % P[2])
    a("v_cndmask_b32_e64 v%d, %%[nihi], v213, s[60:61]" % (P[2] + 1))
    a("v_cndmask_b32_e64 v%d, 0, v210, s[60:61]" % P[1])
    a("v_cndmask_b32_e64 v%d, %%[nihi], v211, s[60:61]" % (P[1] + 1))
    a("v_cndmask_b32_e64 v214, 4, v214, s[54:55]")
    a("v_cndmask_b32_e64 v%d, 0, v208, s[60:61]" % P[0])
    a("v_cndmask_b32_e64 v%d, %%[nihi], v209, s[60:61]" % (P[0] + 1))
    a("v_cmp_lg_f64_e64 s[52:53], %s, %%[ni]" % pair(P[2]))
    a("v_cmp_lg_f64_e64 s[54:55], %s, %%[ni]" % pair(P[1]))
    a("v_cmp_lg_f64_e64 s[56:57], %s, %%[ni]" % pair(P[0]))
    a("ds_write2_b64 v219, %s, %s offset1:1" % (pair(P[0]), pair(P[1])))
    a("ds_write_b64 v219, %s offset:16" % pair(P[2]))
    a("v_cndmask_b32_e64 v216, 3, v216, s[52:53]")
    a("v_cndmask_b32_e64 v215, 3, v215, s[54:55]")
    a("v_cndmask_b32_e64 v214, 3, v214, s[56:57]")
    a("v_mov_b32_e32 v220, %[d]")
    a("s_and_saveexec_b64 s[62:63], s[60:61]")
    a("global_store_dwordx4 v217, v[%d:%d], s[64:65]" % (P[0], P[0] + 3))
    a("global_store_dwordx2 v217, %s, s[64:65] offset:16" % pair(P[2]))
    a("global_store_dwordx3 v218, v[214:216], s[66:67]")
    a("s_mov_b64 exec, s[62:63]")
    a("s_waitcnt vmcnt(24)")                                   # all but the last 8 steps' stores have retired (far reads rely on it)
    a("ds_write_b32 %[fme], v220")                             # progress: after the ring writes (a wave's LDS operations execute in order)
    # next diagonal
    a("s_add_i32 %[d], %[d], 1")
    a("s_add_i32 %[sb], %[sb], 0x1800")
    a("s_cmp_eq_u32 %[sb], 0x16800")
    a("s_cselect_b32 %[sb], 0, %[sb]")
    a("s_add_u32 s70, s70, 0x20")
    a("s_addc_u32 s71, s71, 0")
    return L


def main():
    L = []
    a = L.append
    a("; ==== class 0 run of a compute wave: generated by tools/gen_hot_asm.py ====")
    for c in range(3):
        a("v_mov_b64_e32 %s, %%[p%d]" % (pair(P[c]), c))
        a("v_mov_b64_e32 %s, %%[c%d]" % (pair(Q[1][c]), c))     # half 0 reads C from Q1
    a("s_mov_b64 s[70:71], %[dptr]")
    a("s_load_dwordx8 s[36:43], s[70:71], 0x0")
    a(".Lpg_loop_%=:")
    L += step(0)
    L += step(1)
    a("s_branch .Lpg_loop_%=")
    # exits: the step that could not run has changed nothing but the operand pipeline (reloaded by the caller) and,
    # possibly, the row (idempotent).  C is in Q1 when half 0 gives up, in Q0 when half 1 does.
    a(".Lpg_exit0_%=:")
    for c in range(3):
        a("v_mov_b64_e32 %%[c%d], %s" % (c, pair(Q[1][c])))
    a("s_branch .Lpg_done_%=")
    a(".Lpg_exit1_%=:")
    for c in range(3):
        a("v_mov_b64_e32 %%[c%d], %s" % (c, pair(Q[0][c])))
    a(".Lpg_done_%=:")
    for c in range(3):
        a("v_mov_b64_e32 %%[p%d], %s" % (c, pair(P[c])))
    a("s_mov_b64 %[dptr], s[70:71]")
    a("s_waitcnt lgkmcnt(0)")
    with open(OUT, "w") as f:
        f.write("// generated by tools/gen_hot_asm.py -- do not edit\n")
        for l in L:
            f.write('"%s\\n\\t"\n' % l)
    print("wrote", OUT, len([l for l in L if not l.startswith(';') and not l.endswith(':')]), "instructions")


if __name__ == "__main__":
    main()
