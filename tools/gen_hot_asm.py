#!/usr/bin/env python3
"""Generates pagan2-msa_amd/csrc/dp_pipe_hot.inc: the text of the inline-asm loop that runs consecutive class 0 and
class 1 anti-diagonals of dp_pipe.hip's compute waves (hot_run).  Written as a generator because the loop is unrolled
by two with the register sets' roles swapped (no copies in the steady state), because the multi-edge blocks of a class 1
step are the same code for the left and the right site, and because symbolic names keep ~700 lines of gfx950 assembly
checkable.  Run it after changing the schedule:  python tools/gen_hot_asm.py

What a step computes is stated in dp_pipe.hip (hot_run: the C++ rendering of the same step, which stays the reference for
every diagonal this loop hands back).  Register plan (fixed registers are in the asm statement's clobber list; operands
are %[name]):
  v[224:229]  P: this lane's cell of the previous diagonal, X Y M (what the step writes back, -inf outside the band)
  Q0 = v[180:185], Q1 = v[230:235]: the shifted cell (row-1, .) -- C of this step in one, A of this step in the other
  v[186:213]  candidates of the base step; v214..216 the three back-pointer words; v217..v223 addresses
  class 1: v[236:239] / v[240:243] the row's / column's site record, v244/v245 kL/kR, v[246:253] the four edge weights,
           v254 the left other-edge's ring column, v[160:179] temporaries, U = v[192:197], V = v[200:205] operand cells
  s[36:43] / s[44:51]  descriptor of this diagonal / of the next one (roles swap)
  s[52:59] compare masks, s[60:61] in-band lanes, s[62:63] saved exec, s[64:67] store bases, s68/s69/s72/s73 scratch,
  s[70:71] descriptor pointer, s[74:75] l2, s[76:77] r2, s[78:79] lS, s[80:81] rS, s[82:83] l2 & r2, s[84:87] scratch masks
Hazards observed by construction (the assembler inserts nothing): two wait states between a VALU compare and the VALU
that reads its mask; no VALU write of a DPP source within two; LDS results only behind s_waitcnt.
"""
import os

OUT = os.environ.get("PG_HOT_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pagan2-msa_amd", "csrc", "dp_pipe_hot.inc")

P = (224, 226, 228)                      # X, Y, M (register pairs start)
Q = ((180, 182, 184), (230, 232, 234))
TM, TX = 188, 190
BX, BY, BM = 208, 210, 212
PXW, PYW, PMW = 214, 215, 216
U = (192, 194, 196)                      # operand cell: x, y, m
V = (200, 202, 204)
W1 = (140, 142, 144)                     # more operand cells of a class 1 step
W2 = (146, 148, 150)
W3 = (152, 154, 156)
RL, CR = 236, 240                        # site records: x, y, z, w (64-bit aligned tuples)
KL, KR = 244, 245
LWA, LWS, RWA, RWS = 246, 248, 250, 252
POSL = 254
T = [160 + 2 * i for i in range(10)]     # temporary pairs v[160:179]


def pr(r):
    return "v[%d:%d]" % (r, r + 1)


class Emit:
    def __init__(self):
        self.L = []
        self.ool = []          # out-of-line blocks, emitted behind the loop
        self.cur = self.L

    def a(self, s):
        self.cur.append(s)

    def fmax3(self, dst, code, c1, c2, c3, f1, f2, f3, m23, w3="s[52:53]", w23="s[54:55]"):
        """first-wins maximum of three candidates (pairs c1..c3) into pair dst; `code` = f1/f2/f3 of the first one equal to it"""
        a = self.a
        a("v_max_f64 %s, %s, %s" % (pr(m23), pr(c2), pr(c3)))
        a("v_cmp_gt_f64_e64 %s, %s, %s" % (w3, pr(c3), pr(c2)))
        a("v_max_f64 %s, %s, %s" % (pr(dst), pr(c1), pr(m23)))
        a("v_cmp_gt_f64_e64 %s, %s, %s" % (w23, pr(m23), pr(c1)))
        a("v_cndmask_b32_e64 v%d, %d, %d, %s" % (code, f2, f3, w3))
        a("s_nop 0")
        a("v_cndmask_b32_e64 v%d, %d, v%d, %s" % (code, f1, code, w23))

    def take_better(self, va, pa, vb, pb, first):
        """(vb, pb) replaces (va, pa) if strictly greater, or equal and listed first (mask `first`)"""
        a = self.a
        a("v_cmp_gt_f64_e64 s[84:85], %s, %s" % (pr(vb), pr(va)))
        a("v_cmp_eq_f64_e64 s[86:87], %s, %s" % (pr(vb), pr(va)))
        a("s_and_b64 s[86:87], s[86:87], %s" % first)
        a("s_or_b64 s[84:85], s[84:85], s[86:87]")
        a("v_cndmask_b32_e64 v%d, v%d, v%d, s[84:85]" % (va, va, vb))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, s[84:85]" % (va + 1, va + 1, vb + 1))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, s[84:85]" % (pa, pa, pb))

    def ring_addr(self, dst, age_bytes, col, present):
        """dst = LDS address of the ring cell `age_bytes` (VGPR: age * 0x1800) back from this diagonal's row in ring column
        `col` (VGPR: absolute LDS address of the column in ring row 0), or the all -inf null cell where `present` is off"""
        a = self.a
        a("v_sub_u32_e32 v%d, %%[sb], v%d" % (dst, age_bytes))
        a("v_add_u32_e32 v223, 0x16800, v%d" % dst)
        a("v_min_u32_e32 v%d, v%d, v223" % (dst, dst))
        a("v_add_u32_e32 v%d, v%d, %s" % (dst, dst, col))
        a("v_cndmask_b32_e64 v%d, %%[nulla], v%d, %s" % (dst, dst, present))

    def read_cell(self, cell, addr):
        self.a("ds_read2_b64 v[%d:%d], v%d offset1:1" % (cell[0], cell[0] + 3, addr))
        self.a("ds_read_b64 %s, v%d offset:16" % (pr(cell[2]), addr))

    def gap_cands(self, cell, own, other, c1, c2, c3):
        """the three gap candidates of an operand cell: own state + ext, other gap state + open, (M + non-gap) + open"""
        a = self.a
        a("v_add_f64 %s, %s, %%[ge]" % (pr(c1), pr(cell[own])))
        a("v_add_f64 %s, %s, %%[go]" % (pr(c2), pr(cell[other])))
        a("v_add_f64 %s, %s, %%[ng]" % (pr(c3), pr(cell[2])))
        a("v_add_f64 %s, %s, %%[go]" % (pr(c3), pr(c3)))

    def pair_cands(self, cell, lw, rw, c1, c2, c3):
        """the three match candidates of one (left edge, right edge) pair: ((S + t) + lw) + rw for S = M, X, Y"""
        a = self.a
        a("v_add_f64 %s, %s, %s" % (pr(c1), pr(cell[2]), pr(TM)))
        a("v_add_f64 %s, %s, %s" % (pr(c2), pr(cell[0]), pr(TX)))
        a("v_add_f64 %s, %s, %s" % (pr(c3), pr(cell[1]), pr(TX)))
        for c in (c1, c2, c3):
            a("v_add_f64 %s, %s, %s" % (pr(c), pr(c), pr(lw)))
        for c in (c1, c2, c3):
            a("v_add_f64 %s, %s, %s" % (pr(c), pr(c), pr(rw)))


EXP = os.environ.get("PG_HOT_EXP", "")


def class1(E, k):
    """The multi-edge part of a class 1 step (dp_pipe.hip, hot_run): entered with the base candidates done (bx, by from
    the previous-site edges; bm is recomputed with edge weights), leaves bx/by/bm and the back-pointer words final."""
    a = E.a
    C = Q[1 - k]
    sfx = "%d_%%=" % k
    a("; ---- class 1: this wave's own multi-edge cells ----")
    a("s_waitcnt lgkmcnt(0)")                                   # the two site records
    a("v_and_b32_e32 v217, v%d, v%d" % (RL, CR))
    a("v_and_b32_e32 v217, 0x10000, v217")
    a("v_cmp_eq_u32_e32 vcc, 0, v217")                          # not (both simple)
    a("s_and_b64 vcc, vcc, s[60:61]")
    a("s_cbranch_vccz .Lpg_c1done%s" % sfx)                     # none of the diagonal's multi-edge cells is this wave's
    if EXP == "c":
        a("s_branch .Lpg_c1done%s" % sfx)
    # decode: two edges? the other edge listed first? its distance, the two weights (previous-site edge / other edge)
    for (rec, l2, lS, kk, wA, wS, t0, t1) in ((RL, "s[74:75]", "s[78:79]", KL, LWA, LWS, 217, 218), (CR, "s[76:77]", "s[80:81]", KR, RWA, RWS, 219, 220)):
        a("v_bfe_u32 v%d, v%d, 17, 7" % (t0, rec))
        a("v_and_b32_e32 v%d, 0xffff, v%d" % (t1, rec + 1))
        a("v_cmp_eq_u32_e64 %s, 2, v%d" % (l2, t0))
        a("v_cmp_ne_u32_e64 %s, 1, v%d" % (lS, t1))
        a("v_lshrrev_b32_e32 v%d, 16, v%d" % (t0, rec + 1))
        a("s_and_b64 %s, %s, s[60:61]" % (l2, l2))
        a("s_and_b64 %s, %s, %s" % (lS, lS, l2))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (kk, t0, t1, lS))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (t0, rec + 2, rec + 3, lS))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (t1, rec + 3, rec + 2, lS))
        a("v_cvt_f64_f32_e32 %s, v%d" % (pr(wA), t0))
        a("v_cvt_f64_f32_e32 %s, v%d" % (pr(wS), t1))
    a("s_and_b64 s[82:83], s[74:75], s[76:77]")
    # list slots of the previous-site edges in the back-pointers: v221 = lbA (slot << 4), v222 = rbA (slot << 18)
    a("v_cndmask_b32_e64 v221, 0, 16, s[78:79]")
    a("v_cndmask_b32_e64 v222, 0, %[c18], s[80:81]")
    c1, c2, c3, m23 = T[0], T[1], T[2], T[3]
    # ---- every operand cell of the blocks below requested in ONE batch (each block only if a lane of the wave needs it):
    #   right site's other edge: (row, j-kR) -> U, (row-1, j-kR) -> V; left site's: (row-kL, j) -> W1, (row-kL, j-1) -> W2;
    #   both: (row-kL, j-kR) -> W3.  The weighted pair of the two previous-site edges runs while they are in flight.
    a("s_cmp_eq_u64 s[76:77], 0")
    a("s_cbranch_scc1 .Lpg_rdL%s" % sfx)
    a("v_mul_u32_u24_e32 v217, 0x1800, v%d" % KR)
    E.ring_addr(218, 217, "%[tid24]", "s[76:77]")
    a("v_add_u32_e32 v217, 0x1800, v217")
    E.ring_addr(219, 217, "%[bpos24]", "s[76:77]")
    E.read_cell(U, 218)
    E.read_cell(V, 219)
    a(".Lpg_rdL%s:" % sfx)
    a("s_cmp_eq_u64 s[74:75], 0")
    a("s_cbranch_scc1 .Lpg_rdX%s" % sfx)
    a("v_sub_u32_e32 v%d, %%[tid], v%d" % (POSL, KL))
    a("v_and_b32_e32 v%d, 0xff, v%d" % (POSL, POSL))
    a("v_mad_u32_u24 v%d, v%d, 24, %%[ringb]" % (POSL, POSL))
    a("v_mul_u32_u24_e32 v217, 0x1800, v%d" % KL)
    E.ring_addr(218, 217, "v%d" % POSL, "s[74:75]")
    a("v_add_u32_e32 v219, 0x1800, v217")
    E.ring_addr(220, 219, "v%d" % POSL, "s[74:75]")
    E.read_cell(W1, 218)
    E.read_cell(W2, 220)
    a("s_cmp_eq_u64 s[82:83], 0")
    a("s_cbranch_scc1 .Lpg_rdX%s" % sfx)
    a("v_add_u32_e32 v217, v%d, v%d" % (KL, KR))
    a("v_mul_u32_u24_e32 v217, 0x1800, v217")
    E.ring_addr(220, 217, "v%d" % POSL, "s[82:83]")
    E.read_cell(W3, 220)
    a(".Lpg_rdX%s:" % sfx)
    a("v_or_b32_e32 v%d, v%d, v221" % (PXW, PXW))
    a("v_or_b32_e32 v%d, v%d, v222" % (PYW, PYW))
    # the pair of the two previous-site edges, with their weights (the base step's bm left them out)
    E.pair_cands(C, LWA, RWA, c1, c2, c3)
    E.fmax3(BM, PMW, c1, c2, c3, 14, 12, 13, m23)
    a("v_or3_b32 v%d, v%d, v221, v222" % (PMW, PMW))
    a("v_cndmask_b32_e64 v220, %[c18], 0, s[80:81]")           # rbS
    a("v_cndmask_b32_e64 v218, 16, 0, s[78:79]")                # lbS
    a("s_waitcnt lgkmcnt(0)")
    # ---- the right site's other edge: Y from (row, j-kR), the pair (previous-site left edge, it) from (row-1, j-kR) ----
    a("s_cmp_eq_u64 s[76:77], 0")
    a("s_cbranch_scc1 .Lpg_noR%s" % sfx)
    E.gap_cands(U, 1, 0, c1, c2, c3)
    E.fmax3(T[4], 217, c1, c2, c3, 1, 0, 2, m23)
    a("v_or_b32_e32 v217, v217, v220")
    E.take_better(BY, PYW, T[4], 217, "s[80:81]")
    E.pair_cands(V, LWA, RWS, c1, c2, c3)
    E.fmax3(T[4], 217, c1, c2, c3, 6, 4, 5, m23)
    a("v_or3_b32 v217, v217, v221, v220")
    E.take_better(BM, PMW, T[4], 217, "s[80:81]")
    a(".Lpg_noR%s:" % sfx)
    # ---- the left site's other edge: X from (row-kL, j), the pair (it, previous-site right edge) from (row-kL, j-1) ----
    a("s_cmp_eq_u64 s[74:75], 0")
    a("s_cbranch_scc1 .Lpg_noL%s" % sfx)
    E.gap_cands(W1, 0, 1, c1, c2, c3)
    E.fmax3(T[4], 219, c1, c2, c3, 0, 1, 2, m23)
    a("v_or_b32_e32 v219, v219, v218")
    E.take_better(BX, PXW, T[4], 219, "s[78:79]")
    E.pair_cands(W2, LWS, RWA, c1, c2, c3)
    E.fmax3(T[5], 219, c1, c2, c3, 10, 8, 9, m23)               # m2 (T5), p2 (v219)
    a("v_or3_b32 v219, v219, v218, v222")
    # both sites have another edge: the pair of the two, from (row-kL, j-kR)
    a("s_cmp_eq_u64 s[82:83], 0")
    a("s_cbranch_scc1 .Lpg_noLR%s" % sfx)
    E.pair_cands(W3, LWS, RWS, c1, c2, c3)
    E.fmax3(T[4], 217, c1, c2, c3, 2, 0, 1, m23)
    a("v_or3_b32 v217, v217, v218, v220")
    E.take_better(T[5], 219, T[4], 217, "s[80:81]")
    a(".Lpg_noLR%s:" % sfx)
    E.take_better(BM, PMW, T[5], 219, "s[78:79]")
    a(".Lpg_noL%s:" % sfx)
    a(".Lpg_c1done%s:" % sfx)


def class2(E, k):
    """The merge of a class 2 step: what the diagonal's assist wave staged -- for every multi-edge cell, the best X / Y
    candidate over the edges that do not come from the previous site and the best M over all edge pairs (dp_pipe.hip,
    pipe_assist) -- into the base step's candidates.  X / Y by value, a tie by list position (PS_FIRST: the staged winner
    precedes the previous-site edge; PS_ONLY: there is no previous-site edge); M as staged."""
    a = E.a
    sfx = "%d_%%=" % k
    ex, ey, em = T[0], T[1], T[2]
    a("; ---- class 2: merge what the assist wave of this diagonal staged ----")
    a("s_mul_hi_u32 s68, %[d], 0xaaaaaaab")
    a("s_lshr_b32 s68, s68, 1")
    a("s_mul_i32 s68, s68, 3")
    a("s_sub_i32 s72, %[d], s68")                              # staging slot (and assist wave): d % 3
    a("s_lshl_b32 s68, s72, 2")
    a("s_add_u32 s68, s68, %[asd]")
    a("v_mov_b32_e32 v217, s68")                               # address of the assist wave's progress flag
    a("s_lshl_b32 s69, s72, 11")
    a("v_add_u32_e32 v218, s69, %[stx]")                       # sx[slot][lane]; sy, sM follow at 6144-byte strides
    a("s_lshl_b32 s69, s72, 10")
    a("v_add_u32_e32 v219, s69, %[spxa]")                      # spx[slot][lane]; spy, spm at 3072-byte strides
    a("s_mov_b32 s72, 64")
    a(".Lpg_c2read%s:" % sfx)
    a("ds_read_b32 v223, v217")                                # flag first: LDS executes in order
    a("ds_read_b64 %s, v218" % pr(ex))
    a("ds_read_b64 %s, v218 offset:6144" % pr(ey))
    a("ds_read_b64 %s, v218 offset:12288" % pr(em))
    a("ds_read_b32 v220, v219")
    a("ds_read_b32 v221, v219 offset:3072")
    a("ds_read_b32 v222, v219 offset:6144")
    a("s_waitcnt lgkmcnt(0)")                                   # (and the two site records)
    a("v_readfirstlane_b32 s68, v223")
    a("s_cmp_lt_i32 s68, %[d]")
    a("s_cbranch_scc0 .Lpg_c2ok%s" % sfx)
    a("s_sleep 1")                                              # the assist wave is, as a rule, nearly there
    a("s_sub_i32 s72, s72, 1")
    a("s_cmp_lg_u32 s72, 0")
    a("s_cbranch_scc1 .Lpg_c2read%s" % sfx)
    a("s_branch .Lpg_exit%s" % sfx)                             # nothing of this step is committed: the caller's poll takes over
    a(".Lpg_c2ok%s:" % sfx)
    a("v_and_b32_e32 v217, 0x10000, v%d" % RL)
    a("v_and_b32_e32 v218, 0x10000, v%d" % CR)
    a("v_cmp_eq_u32_e64 s[74:75], 0, v217")                     # msL: the left site is a multi-edge one
    a("v_cmp_eq_u32_e64 s[76:77], 0, v218")                     # msR
    for (sv, val, bst, pw, ms, first_shift, slot_shift, keep) in ((220, ex, BX, PXW, "s[74:75]", 18, 4, 0x3ffff), (221, ey, BY, PYW, "s[76:77]", 4, 18, 0x01fc000f)):
        a("v_cmp_gt_f64_e64 s[84:85], %s, %s" % (pr(val), pr(bst)))
        a("v_cmp_eq_f64_e64 s[86:87], %s, %s" % (pr(val), pr(bst)))
        a("v_and_b32_e32 v217, 0x40000000, v%d" % sv)
        a("v_cmp_gt_i32_e64 s[78:79], 0, v%d" % sv)             # PS_ONLY (bit 31)
        a("v_cmp_ne_u32_e64 s[80:81], 0, v217")                 # PS_FIRST
        a("v_bfe_u32 v218, v%d, %d, 7" % (sv, first_shift))     # the previous-site edge's list slot
        a("s_and_b64 s[86:87], s[86:87], s[80:81]")
        a("s_or_b64 s[84:85], s[84:85], s[86:87]")
        a("v_lshlrev_b32_e32 v218, %d, v218" % slot_shift)
        a("s_or_b64 s[84:85], s[84:85], s[78:79]")
        a("s_and_b64 s[84:85], s[84:85], %s" % ms)              # take the staged candidate
        a("v_cndmask_b32_e64 v218, 0, v218, %s" % ms)
        a("v_and_b32_e32 v217, 0x%x, v%d" % (keep, sv))
        a("v_or_b32_e32 v%d, v%d, v218" % (pw, pw))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, s[84:85]" % (bst, bst, val))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, s[84:85]" % (bst + 1, bst + 1, val + 1))
        a("v_cndmask_b32_e64 v%d, v%d, v217, s[84:85]" % (pw, pw))
    a("s_or_b64 s[84:85], s[74:75], s[76:77]")
    a("s_nop 0")
    a("v_cndmask_b32_e64 v%d, v%d, v%d, s[84:85]" % (BM, BM, em))
    a("v_cndmask_b32_e64 v%d, v%d, v%d, s[84:85]" % (BM + 1, BM + 1, em + 1))
    a("v_cndmask_b32_e64 v%d, v%d, v222, s[84:85]" % (PMW, PMW))


def step(E, k):
    """One diagonal.  k = 0 / 1: which half of the unrolled pair (selects descriptor and Q roles)."""
    a = E.a
    cur = 36 if k == 0 else 44
    nxt = 44 if k == 0 else 36
    A = Q[k]          # receives (row-1, j) on d-1
    C = Q[1 - k]      # holds (row-1, j-1) on d-2
    lo, hi, zlo, zhi, s4, s7 = cur, cur + 1, cur + 2, cur + 3, cur + 4, cur + 7
    sfx = "%d_%%=" % k
    a("; ---- diagonal, half %d ----" % k)
    # LDS batch: upstream flag, lane 0's operand, the column record two steps ahead, next step's model score, the row record
    a("ds_read_b32 v223, %[fup]")
    a("s_sub_i32 s68, %[sb], 0x1800")
    a("s_cmp_lt_i32 s68, 0")
    a("s_cselect_b32 s68, 0x15000, s68")                       # ring row of d-1
    a("v_add_u32_e32 v217, s68, %[bpos24]")
    a("ds_read2_b64 v[%d:%d], v217 offset1:1" % (A[0], A[0] + 3))
    a("ds_read_b64 %s, v217 offset:16" % pr(A[2]))
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
    a("v_add_f64 v[192:193], %s, %%[ge]" % pr(P[1]))          # y1 = PY + ge         -> Y | ADJR = 9
    a("v_add_f64 v[194:195], %s, %%[go]" % pr(P[0]))          # y2 = PX + go         -> X | ADJR = 8
    a("v_add_f64 v[196:197], %s, %%[ng]" % pr(P[2]))          # y3 = (PM + ng) + go  -> M | ADJR = 10
    a("v_add_f64 %s, %%[tng2], v[186:187]" % pr(TM))
    a("v_add_f64 %s, %%[tng1], v[186:187]" % pr(TX))
    a("v_add_f64 v[196:197], v[196:197], %[go]")
    a("v_add_f64 v[200:201], %s, %s" % (pr(C[2]), pr(TM)))    # m1 = CM + tM -> 14
    a("v_add_f64 v[202:203], %s, %s" % (pr(C[0]), pr(TX)))    # m2 = CX + tX -> 12
    a("v_add_f64 v[204:205], %s, %s" % (pr(C[1]), pr(TX)))    # m3 = CY + tX -> 13
    a("v_max_f64 v[198:199], v[194:195], v[196:197]")          # y23
    a("v_cmp_gt_f64_e64 s[52:53], v[196:197], v[194:195]")     # y3 > y2
    a("v_max_f64 v[206:207], v[202:203], v[204:205]")          # m23
    a("v_cmp_gt_f64_e64 s[54:55], v[204:205], v[202:203]")     # m3 > m2
    a("v_max_f64 %s, v[192:193], v[198:199]" % pr(BY))
    a("v_cmp_gt_f64_e64 s[56:57], v[198:199], v[192:193]")     # y23 > y1
    a("v_max_f64 %s, v[200:201], v[206:207]" % pr(BM))
    a("v_cmp_gt_f64_e64 s[58:59], v[206:207], v[200:201]")     # m23 > m1
    a("v_cndmask_b32_e64 v215, 8, 10, s[52:53]")
    a("v_cndmask_b32_e64 v216, 12, 13, s[54:55]")
    a("v_cndmask_b32_e64 v215, 9, v215, s[56:57]")
    a("v_cndmask_b32_e64 v216, 14, v216, s[58:59]")
    # the LDS batch and the descriptor of this diagonal (requested a step ago) are here
    a("s_waitcnt lgkmcnt(0)")
    a("s_and_b32 s73, s%d, 15" % s4)
    a("s_cmp_gt_u32 s73, 2")
    a("s_cbranch_scc1 .Lpg_exit%s" % sfx)                      # not class 0 .. 2
    if EXP == "b":
        a("s_mov_b32 s73, 0")
    if EXP == "d":
        a("s_cmp_eq_u32 s73, 2")
        a("s_cselect_b32 s73, 0, s73")
    if EXP == "e":
        a("s_mov_b32 s73, 0")
    a("s_cmp_ge_i32 %[d], %[sleep]")
    a("s_cbranch_scc1 .Lpg_exit%s" % sfx)
    a("s_cmp_gt_i32 %[d], %[okuntil]")
    a("s_cbranch_scc1 .Lpg_exit%s" % sfx)                      # the loader's flags have to be looked at
    a("s_cmp_gt_i32 s%d, %%[pdn]" % s7)
    a("s_cbranch_scc1 .Lpg_dnwait%s" % sfx)                    # the downstream wave's flag has to be looked at
    a(".Lpg_dnok%s:" % sfx)
    E.cur = E.ool
    # ring row reuse: the downstream wave must have completed the last diagonal that reads the row this step overwrites.
    # Where long edges are about it may lag two diagonals at most: look at its flag here, a few times, before giving up
    a(".Lpg_dnwait%s:" % sfx)
    a("s_mov_b32 s72, 48")
    a(".Lpg_dnretry%s:" % sfx)
    a("ds_read_b32 v222, %[fdn]")
    a("s_waitcnt lgkmcnt(0)")
    a("v_readfirstlane_b32 s68, v222")
    a("s_max_i32 %[pdn], %[pdn], s68")
    a("s_cmp_gt_i32 s%d, %%[pdn]" % s7)
    a("s_cbranch_scc0 .Lpg_dnok%s" % sfx)
    a("s_sleep 1")
    a("s_sub_i32 s72, s72, 1")
    a("s_cmp_lg_u32 s72, 0")
    a("s_cbranch_scc1 .Lpg_dnretry%s" % sfx)
    a("s_branch .Lpg_exit%s" % sfx)
    E.cur = E.L
    a("v_readfirstlane_b32 s68, v223")
    a("s_max_i32 %[pup], %[pup], s68")
    a("s_sub_i32 s69, %[d], 1")
    a("s_cmp_lt_i32 %[pup], s69")
    a("s_cbranch_scc1 .Lpg_upwait%s" % sfx)
    a(".Lpg_upok%s:" % sfx)
    E.cur = E.ool
    # the upstream wave has not completed d-1 yet: it is, as a rule, a fraction of a step away.  Look again a few times
    # (flag first, then lane 0's operand: LDS executes in order) before handing the wait to the caller's poll.
    a(".Lpg_upwait%s:" % sfx)
    a("s_mov_b32 s72, 48")
    a(".Lpg_upretry%s:" % sfx)
    a("s_sleep 1")
    a("ds_read_b32 v223, %[fup]")
    a("ds_read2_b64 v[%d:%d], v217 offset1:1" % (A[0], A[0] + 3))
    a("ds_read_b64 %s, v217 offset:16" % pr(A[2]))
    a("s_waitcnt lgkmcnt(0)")
    a("v_readfirstlane_b32 s68, v223")
    a("s_max_i32 %[pup], %[pup], s68")
    a("s_cmp_lt_i32 %[pup], s69")
    a("s_cbranch_scc0 .Lpg_upok%s" % sfx)
    a("s_sub_i32 s72, s72, 1")
    a("s_cmp_lg_u32 s72, 0")
    a("s_cbranch_scc1 .Lpg_upretry%s" % sfx)
    a("s_branch .Lpg_exit%s" % sfx)
    E.cur = E.L
    # row hand-over
    a("v_cmp_gt_i32_e32 vcc, s%d, %%[row]" % lo)
    a("v_add_u32_e32 v222, 0x100, %[row]")
    a("s_add_u32 s64, %%[sclo], s%d" % zlo)                    # score / back-pointer rows of this diagonal
    a("v_cndmask_b32_e32 %[row], %[row], v222, vcc")
    a("s_addc_u32 s65, %%[schi], s%d" % zhi)
    # class 0: the next descriptor now (a whole step to land).  Class 1: the two site records of this step's cell now, the
    # descriptor after the multi-edge part (its waits would wait for the descriptor as well)
    a("s_cmp_lg_u32 s73, 0")
    a("s_cbranch_scc1 .Lpg_recs%s" % sfx)
    a("s_load_dwordx8 s[%d:%d], s[70:71], 0x20" % (nxt, nxt + 7))
    a(".Lpg_shift%s:" % sfx)
    E.cur = E.ool
    a(".Lpg_recs%s:" % sfx)
    a("v_and_b32_e32 v218, 0x1ff, %[row]")
    a("v_sub_u32_e32 v219, %[d], %[row]")
    a("v_lshl_add_u32 v218, v218, 4, %[bL]")
    a("v_and_b32_e32 v219, 0x1ff, v219")
    a("ds_read_b128 v[%d:%d], v218" % (RL, RL + 3))
    a("v_lshl_add_u32 v219, v219, 4, %[bR]")
    a("ds_read_b128 v[%d:%d], v219" % (CR, CR + 3))
    a("s_branch .Lpg_shift%s" % sfx)
    E.cur = E.L
    # shift: lane n takes lane n-1's cell, lane 0 keeps what it read from the ring
    for c in range(3):
        a("v_mov_b32_dpp v%d, v%d wave_shr:1 row_mask:0xf bank_mask:0xf" % (A[c], P[c]))
        a("v_mov_b32_dpp v%d, v%d wave_shr:1 row_mask:0xf bank_mask:0xf" % (A[c] + 1, P[c] + 1))
    a("s_lshr_b64 s[68:69], s[%d:%d], 1" % (zlo, zhi))
    # X from A
    a("v_add_f64 v[192:193], %s, %%[ge]" % pr(A[0]))          # x1 = AX + ge        -> X | ADJL = 4
    a("v_add_f64 v[194:195], %s, %%[go]" % pr(A[1]))          # x2 = AY + go        -> Y | ADJL = 5
    a("v_add_f64 v[196:197], %s, %%[ng]" % pr(A[2]))          # x3 = (AM + ng) + go -> M | ADJL = 6
    a("s_add_u32 s66, %[bplo], s68")
    a("s_addc_u32 s67, %[bphi], s69")
    a("v_cmp_ge_i32_e64 s[60:61], s%d, %%[row]" % hi)          # active: row <= hi
    a("v_add_f64 v[196:197], v[196:197], %[go]")
    a("v_max_f64 v[198:199], v[194:195], v[196:197]")          # x23
    a("v_cmp_gt_f64_e64 s[52:53], v[196:197], v[194:195]")
    a("v_max_f64 %s, v[192:193], v[198:199]" % pr(BX))
    a("v_cmp_gt_f64_e64 s[54:55], v[198:199], v[192:193]")
    a("v_cndmask_b32_e64 v214, 5, 6, s[52:53]")
    a("s_nop 0")
    a("v_cndmask_b32_e64 v214, 4, v214, s[54:55]")
    a("s_cmp_lg_u32 s73, 0")
    a("s_cbranch_scc1 .Lpg_c1%s" % sfx)
    a(".Lpg_commit%s:" % sfx)
    E.cur = E.ool
    a(".Lpg_c1%s:" % sfx)
    a("s_cmp_eq_u32 s73, 2")
    a("s_cbranch_scc1 .Lpg_c2%s" % sfx)
    class1(E, k)
    a("s_load_dwordx8 s[%d:%d], s[70:71], 0x20" % (nxt, nxt + 7))
    a("s_branch .Lpg_commit%s" % sfx)
    a(".Lpg_c2%s:" % sfx)
    class2(E, k)
    a("s_load_dwordx8 s[%d:%d], s[70:71], 0x20" % (nxt, nxt + 7))
    a("s_branch .Lpg_commit%s" % sfx)
    E.cur = E.L
    # results: -inf outside the band; a state that stayed -inf has no back-pointer
    a("v_subrev_u32_e32 v217, s%d, %%[row]" % lo)              # row - lo
    a("v_cndmask_b32_e64 v%d, 0, v%d, s[60:61]" % (P[2], BM))
    a("v_cndmask_b32_e64 v%d, %%[nihi], v%d, s[60:61]" % (P[2] + 1, BM + 1))
    a("v_cndmask_b32_e64 v%d, 0, v%d, s[60:61]" % (P[1], BY))
    a("v_cndmask_b32_e64 v%d, %%[nihi], v%d, s[60:61]" % (P[1] + 1, BY + 1))
    a("v_mul_u32_u24_e32 v218, 12, v217")
    a("v_cndmask_b32_e64 v%d, 0, v%d, s[60:61]" % (P[0], BX))
    a("v_cndmask_b32_e64 v%d, %%[nihi], v%d, s[60:61]" % (P[0] + 1, BX + 1))
    a("v_add_u32_e32 v219, %[sb], %[tid24]")
    a("v_cmp_lg_f64_e64 s[52:53], %s, %%[ni]" % pr(P[2]))
    a("v_cmp_lg_f64_e64 s[54:55], %s, %%[ni]" % pr(P[1]))
    a("v_cmp_lg_f64_e64 s[56:57], %s, %%[ni]" % pr(P[0]))
    a("ds_write2_b64 v219, %s, %s offset1:1" % (pr(P[0]), pr(P[1])))
    a("ds_write_b64 v219, %s offset:16" % pr(P[2]))
    a("v_cndmask_b32_e64 v216, 3, v216, s[52:53]")
    a("v_cndmask_b32_e64 v215, 3, v215, s[54:55]")
    a("v_cndmask_b32_e64 v214, 3, v214, s[56:57]")
    a("v_lshlrev_b32_e32 v217, 1, v218")
    a("v_mov_b32_e32 v220, %[d]")
    a("s_and_saveexec_b64 s[62:63], s[60:61]")
    a("global_store_dwordx4 v217, v[%d:%d], s[64:65]" % (P[0], P[0] + 3))
    a("global_store_dwordx2 v217, %s, s[64:65] offset:16" % pr(P[2]))
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


def main():
    E = Emit()
    a = E.a
    a("; ==== class 0 / class 1 run of a compute wave: generated by tools/gen_hot_asm.py ====")
    for c in range(3):
        a("v_mov_b64_e32 %s, %%[p%d]" % (pr(P[c]), c))
        a("v_mov_b64_e32 %s, %%[c%d]" % (pr(Q[1][c]), c))       # half 0 reads C from Q1
    a("s_mov_b64 s[70:71], %[dptr]")
    a("s_load_dwordx8 s[36:43], s[70:71], 0x0")
    # Device functions are 4-byte aligned: without this the loop's place in the instruction cache lines -- and with it the
    # step time, by a few percent -- moves whenever any code in front of it changes size.  (s_nop padding, run once.)
    a(".p2alignl 6, 3212836864")
    for _ in range(int(os.environ.get("PG_HOT_PAD", "1"))):       # measured: 4 bytes past a 32-byte boundary is the best place
        a("s_nop 0")
    a(".Lpg_loop_%=:")
    step(E, 0)
    step(E, 1)
    a("s_branch .Lpg_loop_%=")
    # exits: the step that could not run has changed nothing but the operand pipeline (reloaded by the caller) and,
    # possibly, the row (idempotent).  C is in Q1 when half 0 gives up, in Q0 when half 1 does.
    a(".Lpg_exit0_%=:")
    for c in range(3):
        a("v_mov_b64_e32 %%[c%d], %s" % (c, pr(Q[1][c])))
    a("s_branch .Lpg_done_%=")
    a(".Lpg_exit1_%=:")
    for c in range(3):
        a("v_mov_b64_e32 %%[c%d], %s" % (c, pr(Q[0][c])))
    a(".Lpg_done_%=:")
    for c in range(3):
        a("v_mov_b64_e32 %%[p%d], %s" % (c, pr(P[c])))
    a("s_mov_b64 %[dptr], s[70:71]")
    a("s_waitcnt lgkmcnt(0)")
    a("s_branch .Lpg_end_%=")
    E.L += E.ool
    a(".Lpg_end_%=:")
    text = []
    for l in E.L:
        text.append(l)
    with open(OUT, "w") as f:
        f.write("// generated by tools/gen_hot_asm.py -- do not edit\n")
        for l in text:
            f.write('"%s\\n\\t"\n' % l)
    print("wrote", OUT, len([l for l in text if not l.startswith(';') and not l.endswith(':')]), "instructions")


if __name__ == "__main__":
    main()
