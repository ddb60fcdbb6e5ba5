#!/usr/bin/env python3
"""Generates pagan2-msa_amd/csrc/dp_pipe_hot.inc: the text of the inline-asm loop that runs consecutive class 0, 1 and 2
anti-diagonals of dp_pipe.hip's compute waves (hot_run).  Written as a generator because the loop is unrolled by two with
the register sets' roles swapped (no copies in the steady state), because the multi-edge blocks of a class 1 step are the
same code for the left and the right site, and because symbolic names keep the gfx950 assembly checkable.  Run it after
changing the schedule:  python tools/gen_hot_asm.py

SCORES ONLY (round 3).  The wave that carries the dependency chain computes and stores the three scores of a cell and
nothing else: a back-pointer is a function of scores that are final by then, and pg_backptr (dp_kernels.hip) re-derives
all of them after the fill, one thread per cell.  Without the winner to track, a state's value is a plain maximum, and
(fp64 addition is monotone: a <= b implies a + c <= b + c in round-to-nearest, so max(a + c, b + c) == max(a, b) + c bit
for bit; no score is NaN, none is -0.0 -- dp_abi.hip keeps jobs with a negative zero among their parameters off this
kernel) the candidates of the reference's lists (VA:1328-1349, 1396-1433, 2029-2219) fold:
    X[i][j] = max(X[p][j] + ge, max(Y[p][j], M[p][j] + ng) + go)                     per bwd edge p -> i     (3 add, 2 max)
    Y[i][j] = max(Y[i][q] + ge, max(X[i][q], M[i][q] + ng) + go)                     per bwd edge q -> j
    M[i][j] = (max(M[p][q] + tM, max(X[p][q], Y[p][q]) + tX) + lw) + rw              per edge pair           (2 add, 2 max)
with tM = D(2*ng) + D(s), tX = D(0+ng) + D(s) read ready from LDS (PM.tab2).  Every addition is one the reference performs
on the same operands; only maxima are regrouped, which cannot change their value.

Register plan (fixed registers are in the asm statement's clobber list; operands are %[name]):
  v[224:229]  P: this lane's cell of the previous diagonal, X Y M (what the step writes back, -inf outside the band)
  Q0 = v[180:185], Q1 = v[230:235]: the shifted cell (row-1, .) -- C of this step in one, A of this step in the other
  v[188:191] / v[192:195]  tM, tX of this step / of the next one (roles swap)
  v[196:207] temporaries of the base step; v[208:213] BX BY BM; v[214:215] the band limit (+inf inside, -inf outside)
  v217..v223 addresses and scratch
  class 1: v[236:239] / v[240:243] (half 0) and v[136:139] / v[244:247] (half 1) the row's / column's site record, v158/v159
           kL/kR, v[186:187] v[248:253] the four edge weights, v216 the left other-edge's ring column, v[140:157] W1 W2 W3,
           v[160:171] U V operand cells, v[172:179] temporaries; s67 the diagonal whose records a class 1 step has read ahead
  s[36:43] / s[44:51]  descriptor of this diagonal / of the next one (roles swap)
  s[52:59] scratch masks, s[60:61] in-band lanes, s[62:63] saved exec, s[64:65] store base, s66 / %[sb] the previous and this diagonal's ring rows (they swap from half to half),
  s68/s69/s72/s73 scratch, s[70:71] descriptor pointer, s[74:75] l2, s[76:77] r2, s[78:79] lS, s[80:81] rS, s[82:83] l2 & r2,
  s[84:87] scratch masks
Hazards observed by construction (the assembler inserts nothing): two wait states between a VALU compare and the VALU
that reads its mask; no VALU write of a DPP source within two; LDS results only behind s_waitcnt; no VALU write of a wide
store's data registers within two.
"""
import os

# STRIP: the variant for row strips of wide jobs (dp_pipe_hot_strip.inc, hot_run<true>).  A strip's lanes keep their rows and
# sweep every column, so a cell of the first or the last column sits on half of a strip's diagonals: the x-gap state's
# extension rate is picked per lane and step -- v[254:255] = (d == row or d - (Ly-1) == row) ? gE : ge, five vector
# instructions -- where the banded kernel sends the few diagonals with such a cell to the general step.
def _ring_depth():
    """PG_PIPE_RING of dp_device.h: the ring's depth in diagonals (rows of 256 cells x 24 bytes = 0x1800)"""
    import re
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pagan2-msa_amd", "csrc", "dp_device.h")) as f:
        return int(re.search(r"#define\s+PG_PIPE_RING\s+(\d+)", f.read()).group(1))


RING_ROWS = _ring_depth()
RING_BYTES = "0x%x" % (RING_ROWS * 0x1800)               # the ring
RING_LAST = "0x%x" % ((RING_ROWS - 1) * 0x1800)          # its last row
STRIP = bool(os.environ.get("PG_HOT_STRIP"))
OUT = os.environ.get("PG_HOT_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pagan2-msa_amd", "csrc",
                                                   "dp_pipe_hot_strip.inc" if STRIP else "dp_pipe_hot.inc")
GEX = 254

P = (224, 226, 228)                      # X, Y, M (register pairs start)
Q = ((180, 182, 184), (230, 232, 234))
TMX = (188, 192)                         # tM at +0, tX at +2
BX, BY, BM = 208, 210, 212
LIM = 214
W1 = (140, 142, 144)                     # operand cells of a class 1 step
W2 = (146, 148, 150)
W3 = (152, 154, 156)
U = (160, 162, 164)
V = (166, 168, 170)
RLK, CRK = (236, 136), (240, 244)        # site records (x, y, z, w) of this step's cell: one set per half of the unrolled pair --
                                         # a class 1 step reads the NEXT step's records into the other half's set
KL, KR = 158, 159
LWA, LWS, RWA, RWS = 186, 248, 250, 252
POSL = 216
T = [172, 174, 176, 178, 196, 198, 200, 202, 204, 206]     # temporary pairs of the class 1 / 2 parts


def pr(r):
    return "v[%d:%d]" % (r, r + 1)


class Emit:
    def __init__(self):
        self.L = []
        self.ool = []          # out-of-line blocks, emitted behind the loop
        self.cur = self.L
        self.cursb = "%[sb]"   # the SGPR that holds this diagonal's ring row (byte offset); "%[sb]" and s66 swap roles from half to half

    def a(self, s):
        self.cur.append(s)

    def ring_row(self, dst, age_bytes):
        """dst = byte offset (inside the ring) of the ring row `age_bytes` (VGPR: age * 0x1800, at most the ring) back from this
        diagonal's: sb - age, plus the ring's size where that wraps (unsigned minimum of the two)"""
        a = self.a
        a("v_sub_u32_e32 v%d, %s, v%d" % (dst, self.cursb, age_bytes))
        a("v_add_u32_e32 v223, %s, v%d" % (RING_BYTES, dst))
        a("v_min_u32_e32 v%d, v%d, v223" % (dst, dst))

    def ring_row_before(self, dst, src):
        """dst = the ring row one diagonal before the row at `src` (both byte offsets inside the ring)"""
        a = self.a
        a("v_subrev_u32_e32 v%d, 0x1800, v%d" % (dst, src))
        a("v_add_u32_e32 v223, %s, v%d" % (RING_BYTES, dst))
        a("v_min_u32_e32 v%d, v%d, v223" % (dst, dst))

    def read_cell(self, cell, addr):
        self.a("ds_read2_b64 v[%d:%d], v%d offset1:1" % (cell[0], cell[0] + 3, addr))
        self.a("ds_read_b64 %s, v%d offset:16" % (pr(cell[2]), addr))

    def gap_value(self, cell, own, other, dst, t1, t2):
        """dst = max(own + ge, max(other, M + ng) + go): the value an edge from `cell` offers the gap state `own` (the y-gap
        state extends at the LANE's rate, %[gey]: the first and the last row of the matrix have their own, and a lane of a row
        strip keeps its row -- dp_pipe.hip, hot_run; the x-gap state's differs by COLUMN, and a diagonal that holds a cell of
        the first or the last column does not run here)"""
        a = self.a
        a("v_add_f64 %s, %s, %%[ng]" % (pr(t1), pr(cell[2])))
        if own == 0 and STRIP:
            a("v_add_f64 %s, %s, %s" % (pr(t2), pr(cell[own]), pr(GEX)))
        else:
            a("v_add_f64 %s, %s, %%[%s]" % (pr(t2), pr(cell[own]), "gey" if own == 1 else "ge"))
        a("v_max_f64 %s, %s, %s" % (pr(t1), pr(cell[other]), pr(t1)))
        a("v_add_f64 %s, %s, %%[go]" % (pr(t1), pr(t1)))
        a("v_max_f64 %s, %s, %s" % (pr(dst), pr(t2), pr(t1)))

    def pair_value(self, cell, tmx, lw, rw, dst, t1):
        """dst = (max(M + tM, max(X, Y) + tX) + lw) + rw: the value the pair of edges from `cell` offers M"""
        a = self.a
        a("v_max_f64 %s, %s, %s" % (pr(t1), pr(cell[0]), pr(cell[1])))
        a("v_add_f64 %s, %s, %s" % (pr(dst), pr(cell[2]), pr(tmx)))
        a("v_add_f64 %s, %s, %s" % (pr(t1), pr(t1), pr(tmx + 2)))
        a("v_max_f64 %s, %s, %s" % (pr(dst), pr(dst), pr(t1)))
        a("v_add_f64 %s, %s, %s" % (pr(dst), pr(dst), pr(lw)))
        a("v_add_f64 %s, %s, %s" % (pr(dst), pr(dst), pr(rw)))


EXP = os.environ.get("PG_HOT_EXP", "")


def c1_test(E, k, wait):
    """Does this wave hold a multi-edge cell of the diagonal?  Branches to .Lpg_c1done (nothing to do) if not."""
    a = E.a
    RL, CR = RLK[k], CRK[k]
    sfx = "%d_%%=" % k
    a("; ---- class 1: this wave's own multi-edge cells ----")
    if wait:
        a("s_waitcnt lgkmcnt(0)")                               # the two site records
    a("v_and_b32_e32 v217, v%d, v%d" % (RL, CR))
    a("v_and_b32_e32 v217, 0x10000, v217")
    a("v_cmp_eq_u32_e32 vcc, 0, v217")                          # not (both simple)
    a("s_and_b64 vcc, vcc, s[60:61]")


def c1_ahead(E, k):
    """The NEXT step's two site records into the other half's registers (they come back with whatever this step waits for
    next, or with the next step's own batch); s67 = the diagonal they belong to."""
    a = E.a
    a("s_add_i32 s67, %[d], 1")
    a("v_and_b32_e32 v217, 0x1ff, %[row]")
    a("v_sub_u32_e32 v218, s67, %[row]")
    a("v_lshl_add_u32 v217, v217, 4, %[bL]")
    a("v_and_b32_e32 v218, 0x1ff, v218")
    a("ds_read_b128 v[%d:%d], v217" % (RLK[1 - k], RLK[1 - k] + 3))
    a("v_lshl_add_u32 v218, v218, 4, %[bR]")
    a("ds_read_b128 v[%d:%d], v218" % (CRK[1 - k], CRK[1 - k] + 3))


def c1_issue(E, k, tag):
    """Decode of the two records, every operand cell of the blocks requested in one batch, the next step's records behind it.

    The loader NORMALISES the records of this kernel's jobs (dp_pipe.hip, load_rec_chunk<., true>): a site with two bwd edges of
    which exactly one starts at the previous site carries that edge in slot 0, the other one in slot 1, and the PR_TWO flag
    (bit 30; bit 31 is PR_FAR, bits 24-28 name the far history lines: dp_pipe.hip) -- the fill computes values only, so the order of a site's list does not matter to it.  What is
    left of the decode: flag, the other edge's distance, two conversions per side.  Each block runs with exec = the lanes whose
    site has the other edge: no lane without one reads or merges anything (before: an all -inf cell selected per operand)."""
    a = E.a
    RL, CR = RLK[k], CRK[k]
    sfx = "%s%d_%%=" % (tag, k)
    a("v_cmp_lt_i32_e64 s[74:75], %%[c24], v%d" % RL)             # l2: the left site has another edge IN THE RING (signed x > 0x3fffffff: PR_TWO, and not PR_FAR)
    a("v_cmp_lt_i32_e64 s[76:77], %%[c24], v%d" % CR)             # r2
    a("v_lshrrev_b32_e32 v%d, 16, v%d" % (KL, RL + 1))
    a("v_lshrrev_b32_e32 v%d, 16, v%d" % (KR, CR + 1))
    a("s_and_b64 s[74:75], s[74:75], s[60:61]")
    a("s_and_b64 s[76:77], s[76:77], s[60:61]")
    a("v_cvt_f64_f32_e32 %s, v%d" % (pr(LWA), RL + 2))            # weights of the previous-site edges (the other edges': in their blocks)
    a("v_cvt_f64_f32_e32 %s, v%d" % (pr(RWA), CR + 2))
    a("s_and_b64 s[82:83], s[74:75], s[76:77]")
    # ---- every operand cell of the blocks below requested in ONE batch (each block only if a lane of the wave needs it):
    #   right site's other edge: (row, j-kR) -> U, (row-1, j-kR) -> V; left site's: (row-kL, j) -> W1, (row-kL, j-1) -> W2;
    #   both: (row-kL, j-kR) -> W3.
    a("s_cmp_eq_u64 s[76:77], 0")
    a("s_cbranch_scc1 .Lpg_rdL%s" % sfx)
    a("s_and_saveexec_b64 s[62:63], s[76:77]")
    a("v_mul_u32_u24_e32 v217, 0x1800, v%d" % KR)
    a("v_cvt_f64_f32_e32 %s, v%d" % (pr(RWS), CR + 3))
    E.ring_row(218, 217)
    E.ring_row_before(219, 218)
    a("v_add_u32_e32 v218, v218, %[tid24]")
    a("v_add_u32_e32 v219, v219, %[bpos24]")
    E.read_cell(U, 218)
    E.read_cell(V, 219)
    a("s_mov_b64 exec, s[62:63]")
    a(".Lpg_rdL%s:" % sfx)
    a("s_cmp_eq_u64 s[74:75], 0")
    a("s_cbranch_scc1 .Lpg_rdX%s" % sfx)
    a("s_and_saveexec_b64 s[62:63], s[74:75]")
    a("v_sub_u32_e32 v%d, %%[tid], v%d" % (POSL, KL))
    a("v_mul_u32_u24_e32 v217, 0x1800, v%d" % KL)
    a("v_cvt_f64_f32_e32 %s, v%d" % (pr(LWS), RL + 3))
    a("v_and_b32_e32 v%d, 0xff, v%d" % (POSL, POSL))
    E.ring_row(218, 217)
    a("v_mad_u32_u24 v%d, v%d, 24, %%[ringb]" % (POSL, POSL))
    E.ring_row_before(220, 218)
    a("v_add_u32_e32 v218, v218, v%d" % POSL)
    a("v_add_u32_e32 v220, v220, v%d" % POSL)
    E.read_cell(W1, 218)
    E.read_cell(W2, 220)
    a("s_cmp_eq_u64 s[82:83], 0")
    a("s_cbranch_scc1 .Lpg_rdY%s" % sfx)
    a("s_mov_b64 exec, s[82:83]")                                # (a subset of the block's lanes)
    a("v_add_u32_e32 v217, v%d, v%d" % (KL, KR))
    a("v_mul_u32_u24_e32 v217, 0x1800, v217")
    E.ring_row(220, 217)
    a("v_add_u32_e32 v220, v220, v%d" % POSL)
    E.read_cell(W3, 220)
    a(".Lpg_rdY%s:" % sfx)
    a("s_mov_b64 exec, s[62:63]")
    a(".Lpg_rdX%s:" % sfx)
    c1_ahead(E, k)


def c1_math(E, k, tag):
    """The blocks' arithmetic, behind the wait for their operands: leaves BX / BY / BM final."""
    a = E.a
    tmx = TMX[k]
    sfx = "%s%d_%%=" % (tag, k)
    # the pair of the two previous-site edges: its weights (the base step left them out; + 0.0 for a simple site)
    a("v_add_f64 %s, %s, %s" % (pr(BM), pr(BM), pr(LWA)))
    a("v_add_f64 %s, %s, %s" % (pr(BM), pr(BM), pr(RWA)))
    a("s_waitcnt lgkmcnt(0)")
    # ---- the right site's other edge: Y from (row, j-kR), the pair (previous-site left edge, it) from (row-1, j-kR) ----
    a("s_cmp_eq_u64 s[76:77], 0")
    a("s_cbranch_scc1 .Lpg_noR%s" % sfx)
    a("s_and_saveexec_b64 s[62:63], s[76:77]")
    E.gap_value(U, 1, 0, T[0], T[1], T[2])
    E.pair_value(V, tmx, LWA, RWS, T[3], T[4])
    a("v_max_f64 %s, %s, %s" % (pr(BY), pr(BY), pr(T[0])))
    a("v_max_f64 %s, %s, %s" % (pr(BM), pr(BM), pr(T[3])))
    a("s_mov_b64 exec, s[62:63]")
    a(".Lpg_noR%s:" % sfx)
    # ---- the left site's other edge: X from (row-kL, j), the pair (it, previous-site right edge) from (row-kL, j-1) ----
    a("s_cmp_eq_u64 s[74:75], 0")
    a("s_cbranch_scc1 .Lpg_noL%s" % sfx)
    a("s_and_saveexec_b64 s[62:63], s[74:75]")
    E.gap_value(W1, 0, 1, T[0], T[1], T[2])
    E.pair_value(W2, tmx, LWS, RWA, T[3], T[4])
    a("v_max_f64 %s, %s, %s" % (pr(BX), pr(BX), pr(T[0])))
    a("v_max_f64 %s, %s, %s" % (pr(BM), pr(BM), pr(T[3])))
    # both sites have another edge: the pair of the two, from (row-kL, j-kR)
    a("s_cmp_eq_u64 s[82:83], 0")
    a("s_cbranch_scc1 .Lpg_noB%s" % sfx)
    a("s_mov_b64 exec, s[82:83]")
    E.pair_value(W3, tmx, LWS, RWS, T[5], T[6])
    a("v_max_f64 %s, %s, %s" % (pr(BM), pr(BM), pr(T[5])))
    a(".Lpg_noB%s:" % sfx)
    a("s_mov_b64 exec, s[62:63]")
    a(".Lpg_noL%s:" % sfx)


def third_pass(E, k):
    """Sites with THREE bwd edges (dp_pipe.hip, PR_THREE: one from the previous site, two others inside the ring's reach; the
    loader keeps them as (previous-site edge, other, other) in the record and in the edge window).  The class 1 blocks above took
    the first other edge (the site looks like a two-edge site to them); on the diagonals the host flags (descriptor word 4, bit
    19) this pass takes the second: its distance and weight from entry 2 of the site's list in the edge window (two dependent LDS
    reads), then the blocks' operands and arithmetic once more -- X (Y) from the edge's cell, the pair with the other site's
    previous-site edge, and the pair with the other site's own other edge where it has one.  A cell where two such sites meet is
    on a class 2 diagonal (the host sees to it), so a lane is in at most one of the two halves."""
    a = E.a
    RL, CR = RLK[k], CRK[k]
    tmx = TMX[k]
    sfx = "%d_%%=" % k
    a("v_and_b32_e32 v217, 0x20000000, v%d" % RL)
    a("v_and_b32_e32 v218, 0x20000000, v%d" % CR)
    a("v_cmp_ne_u32_e64 s[52:53], 0, v217")                    # t3L
    a("v_cmp_ne_u32_e64 s[54:55], 0, v218")                    # t3R
    a("v_sub_u32_e32 v222, %[d], %[row]")                      # the cell's column j
    a("s_and_b64 s[52:53], s[52:53], s[60:61]")
    a("s_and_b64 s[54:55], s[54:55], s[60:61]")
    a("s_or_b64 s[56:57], s[52:53], s[54:55]")
    a("s_cmp_eq_u64 s[56:57], 0")
    a("s_cbranch_scc1 .Lpg_t3end%s" % sfx)
    a("s_and_b64 s[84:85], s[52:53], s[76:77]")                # the left site's second other edge x the right site's other edge
    a("s_and_b64 s[86:87], s[54:55], s[74:75]")                # ... and the other way round
    # ---- the sites' first edge (edge numbering), then entry 2 of their lists: start site and weight ----
    a("v_and_b32_e32 v217, 0x1ff, %[row]")
    a("v_and_b32_e32 v218, 0x1ff, v222")
    a("v_lshl_add_u32 v217, v217, 2, %[ebl]")
    a("v_lshl_add_u32 v218, v218, 2, %[ebr]")
    a("ds_read_b32 v174, v217")
    a("ds_read_b32 v175, v218")
    a("s_waitcnt lgkmcnt(0)")
    a("v_add_u32_e32 v174, 2, v174")
    a("v_add_u32_e32 v175, 2, v175")
    a("v_and_b32_e32 v174, 0x3ff, v174")
    a("v_and_b32_e32 v175, 0x3ff, v175")
    a("v_lshl_add_u32 v217, v174, 2, %[esl]")
    a("v_lshl_add_u32 v218, v175, 2, %[esr]")
    a("v_lshl_add_u32 v219, v174, 2, %[ewl]")
    a("v_lshl_add_u32 v220, v175, 2, %[ewr]")
    a("ds_read_b32 v174, v217")                                # start site of the left site's third edge
    a("ds_read_b32 v175, v218")
    a("ds_read_b32 v172, v219")                                # its weight
    a("ds_read_b32 v173, v220")
    a("s_waitcnt lgkmcnt(0)")
    # ---- the operands ----
    a("s_cmp_eq_u64 s[54:55], 0")
    a("s_cbranch_scc1 .Lpg_t3rdL%s" % sfx)
    a("s_and_saveexec_b64 s[62:63], s[54:55]")
    a("v_sub_u32_e32 v%d, v222, v175" % KR)                    # kR of the second other edge
    a("v_cvt_f64_f32_e32 %s, v173" % pr(RWS))
    a("v_mul_u32_u24_e32 v217, 0x1800, v%d" % KR)
    E.ring_row(218, 217)
    E.ring_row_before(219, 218)
    a("v_add_u32_e32 v218, v218, %[tid24]")
    a("v_add_u32_e32 v219, v219, %[bpos24]")
    E.read_cell(U, 218)
    E.read_cell(V, 219)
    a("s_cmp_eq_u64 s[86:87], 0")
    a("s_cbranch_scc1 .Lpg_t3rdRb%s" % sfx)
    a("s_mov_b64 exec, s[86:87]")
    a("v_add_u32_e32 v217, v%d, v%d" % (KL, KR))
    a("v_mul_u32_u24_e32 v217, 0x1800, v217")
    E.ring_row(220, 217)
    a("v_add_u32_e32 v220, v220, v%d" % POSL)                  # (the left site's other edge: its ring column is the class 1 block's)
    E.read_cell(W3, 220)
    a(".Lpg_t3rdRb%s:" % sfx)
    a("s_mov_b64 exec, s[62:63]")
    a(".Lpg_t3rdL%s:" % sfx)
    a("s_cmp_eq_u64 s[52:53], 0")
    a("s_cbranch_scc1 .Lpg_t3rdX%s" % sfx)
    a("s_and_saveexec_b64 s[62:63], s[52:53]")
    a("v_sub_u32_e32 v%d, %%[row], v174" % KL)                 # kL of the second other edge
    a("v_cvt_f64_f32_e32 %s, v172" % pr(LWS))
    a("v_sub_u32_e32 v%d, %%[tid], v%d" % (POSL, KL))
    a("v_mul_u32_u24_e32 v217, 0x1800, v%d" % KL)
    a("v_and_b32_e32 v%d, 0xff, v%d" % (POSL, POSL))
    E.ring_row(218, 217)
    a("v_mad_u32_u24 v%d, v%d, 24, %%[ringb]" % (POSL, POSL))
    E.ring_row_before(220, 218)
    a("v_add_u32_e32 v218, v218, v%d" % POSL)
    a("v_add_u32_e32 v220, v220, v%d" % POSL)
    E.read_cell(W1, 218)
    E.read_cell(W2, 220)
    a("s_cmp_eq_u64 s[84:85], 0")
    a("s_cbranch_scc1 .Lpg_t3rdLb%s" % sfx)
    a("s_mov_b64 exec, s[84:85]")
    a("v_add_u32_e32 v217, v%d, v%d" % (KL, KR))
    a("v_mul_u32_u24_e32 v217, 0x1800, v217")
    E.ring_row(220, 217)
    a("v_add_u32_e32 v220, v220, v%d" % POSL)
    E.read_cell(W3, 220)
    a(".Lpg_t3rdLb%s:" % sfx)
    a("s_mov_b64 exec, s[62:63]")
    a(".Lpg_t3rdX%s:" % sfx)
    a("s_waitcnt lgkmcnt(0)")
    # ---- the arithmetic ----
    a("s_cmp_eq_u64 s[54:55], 0")
    a("s_cbranch_scc1 .Lpg_t3noR%s" % sfx)
    a("s_and_saveexec_b64 s[62:63], s[54:55]")
    E.gap_value(U, 1, 0, T[0], T[1], T[2])
    E.pair_value(V, tmx, LWA, RWS, T[3], T[4])
    a("v_max_f64 %s, %s, %s" % (pr(BY), pr(BY), pr(T[0])))
    a("v_max_f64 %s, %s, %s" % (pr(BM), pr(BM), pr(T[3])))
    a("s_mov_b64 exec, s[62:63]")
    a(".Lpg_t3noR%s:" % sfx)
    a("s_cmp_eq_u64 s[52:53], 0")
    a("s_cbranch_scc1 .Lpg_t3noL%s" % sfx)
    a("s_and_saveexec_b64 s[62:63], s[52:53]")
    E.gap_value(W1, 0, 1, T[0], T[1], T[2])
    E.pair_value(W2, tmx, LWS, RWA, T[3], T[4])
    a("v_max_f64 %s, %s, %s" % (pr(BX), pr(BX), pr(T[0])))
    a("v_max_f64 %s, %s, %s" % (pr(BM), pr(BM), pr(T[3])))
    a("s_mov_b64 exec, s[62:63]")
    a(".Lpg_t3noL%s:" % sfx)
    # the pairs of a second other edge with the other site's own other edge (either way round: a lane is in one of the two masks)
    a("s_or_b64 s[84:85], s[84:85], s[86:87]")
    a("s_cmp_eq_u64 s[84:85], 0")
    a("s_cbranch_scc1 .Lpg_t3end%s" % sfx)
    a("s_and_saveexec_b64 s[62:63], s[84:85]")
    E.pair_value(W3, tmx, LWS, RWS, T[5], T[6])
    a("v_max_f64 %s, %s, %s" % (pr(BM), pr(BM), pr(T[5])))
    a("s_mov_b64 exec, s[62:63]")
    a(".Lpg_t3end%s:" % sfx)


def hist_tail(E, k, cls2):
    """Far histories (dp_abi.hip, plan_far_hist; PipeSmem::hist): the part of a step on a diagonal whose descriptor says that a
    history line's reader or writer has a cell on it (word 4, bit 5).  Behind the class 1 / class 2 part, BX / BY / BM final but
    for this:
      * READERS (class 1 diagonals; on a class 2 diagonal the assist wave staged the far site's candidates with the rest): a
        site whose other edge reaches k >= reach sites back (PR_FAR; flag byte bit 7, line in bits 0-1) takes that edge's two
        operands from the line -- left site: (row-k, j) and (row-k, j-1) at entries j % 64 and (j-1) % 64 of a line indexed by
        column; right site: (row, j-k) and (row-1, j-k) at entries row % 64 and (row-1) % 64 of a line indexed by row -- the
        gap candidate from the first, the pair (it, the other site's previous-site edge) from the second.  An operand OUTSIDE
        THE BAND was never written (nobody computes it): its diagonal's rows come from the loader's descriptor window and the
        candidate of an operand outside them is left out.
      * WRITERS (flag byte bit 6, line in bits 4-5; rows and columns alike): the lane that holds the cell of a line's row
        (column) appends it -- the final values, in-band lanes only.
    A cell where a far site meets a site with an other edge of its own is on a class 2 diagonal (the host sees to it): the pair of
    the two other edges is nobody's here."""
    a = E.a
    RL, CR = RLK[k], CRK[k]
    tmx = TMX[k]
    sfx = "%d_%%=" % k
    tag = "h2" if cls2 else "h1"
    a("v_sub_u32_e32 v222, %[d], %[row]")                      # the cell's column j
    if not cls2:
        a("v_cmp_gt_i32_e32 vcc, 0, v%d" % RL)                  # farL: PR_FAR (the record's sign)
        a("s_and_b64 s[52:53], vcc, s[60:61]")
        a("v_cmp_gt_i32_e32 vcc, 0, v%d" % CR)                  # farR
        a("s_and_b64 s[54:55], vcc, s[60:61]")
        a("s_or_b64 s[56:57], s[52:53], s[54:55]")
        a("s_cmp_eq_u64 s[56:57], 0")
        a("s_cbranch_scc1 .Lpg_%swr%s" % (tag, sfx))
        for (left, mask, rec, kk, wS, wA_other) in ((True, "s[52:53]", RL, KL, LWS, RWA), (False, "s[54:55]", CR, KR, RWS, LWA)):
            side = "L" if left else "R"
            a("s_cmp_eq_u64 %s, 0" % mask)
            a("s_cbranch_scc1 .Lpg_%sno%s%s" % (tag, side, sfx))
            a("s_and_saveexec_b64 s[62:63], %s" % mask)
            a("v_bfe_u32 v217, v%d, 27, 2" % rec)                                   # the line
            a("v_sub_u32_e32 v218, %%[d], v%d" % kk)                                 # e: the diagonal of the first operand
            a("v_mul_u32_u24_e32 v217, 0x600, v217")
            a("v_cvt_f64_f32_e32 %s, v%d" % (pr(wS), rec + 3))                       # the far edge's weight
            a("v_add_u32_e32 v217, %[histb], v217")
            # entries: left site -- column j and j - 1; right site -- row and row - 1
            if left:
                a("v_and_b32_e32 v221, 63, v222")
                a("v_add_u32_e32 v223, -1, v222")
            else:
                a("v_and_b32_e32 v221, 63, %[row]")
                a("v_add_u32_e32 v223, -1, %[row]")
            a("v_mad_u32_u24 v221, v221, 24, v217")
            a("v_and_b32_e32 v223, 63, v223")
            E.read_cell(W1, 221)
            a("v_mad_u32_u24 v223, v223, 24, v217")
            E.read_cell(W2, 223)
            # the rows of the operands' diagonals e and e - 1 (the loader's descriptor window: lo, hi)
            a("v_and_b32_e32 v221, 0x7f, v218")
            a("v_add_u32_e32 v223, -1, v218")
            a("v_lshl_add_u32 v221, v221, 4, %[drb]")
            a("v_and_b32_e32 v223, 0x7f, v223")
            a("ds_read_b64 v[160:161], v221")
            a("v_lshl_add_u32 v223, v223, 4, %[drb]")
            a("ds_read_b64 v[162:163], v223")
            # the operands' rows: left site row - k (both); right site row and row - 1
            if left:
                a("v_sub_u32_e32 v217, %%[row], v%d" % kk)
                a("v_mov_b32_e32 v218, v217")
            else:
                a("v_mov_b32_e32 v217, %[row]")
                a("v_add_u32_e32 v218, -1, %[row]")
            a("s_waitcnt lgkmcnt(0)")
            a("v_cmp_le_i32_e64 s[84:85], v160, v217")                              # lo(e) <= row of the first operand
            a("v_cmp_ge_i32_e64 s[86:87], v161, v217")                              # ... <= hi(e)
            a("v_cmp_le_i32_e64 s[56:57], v162, v218")
            a("v_cmp_ge_i32_e64 s[58:59], v163, v218")
            if left:
                E.gap_value(W1, 0, 1, T[0], T[1], T[2])
                E.pair_value(W2, tmx, wS, wA_other, T[3], T[4])
            else:
                E.gap_value(W1, 1, 0, T[0], T[1], T[2])
                E.pair_value(W2, tmx, wA_other, wS, T[3], T[4])
            a("s_and_b64 s[84:85], s[84:85], s[86:87]")
            a("s_and_b64 s[56:57], s[56:57], s[58:59]")
            a("s_and_b64 exec, exec, s[84:85]")                                     # (exec = the far lanes: those whose first operand lies in the band)
            a("v_max_f64 %s, %s, %s" % (pr(BX if left else BY), pr(BX if left else BY), pr(T[0])))
            a("s_and_b64 exec, %s, s[56:57]" % mask)
            a("v_max_f64 %s, %s, %s" % (pr(BM), pr(BM), pr(T[3])))
            a("s_mov_b64 exec, s[62:63]")
            a(".Lpg_%sno%s%s:" % (tag, side, sfx))
    # ---- writers ----
    a(".Lpg_%swr%s:" % (tag, sfx))
    a("v_and_b32_e32 v217, 0x1000000, v%d" % RL)               # PR_SRC
    a("v_and_b32_e32 v218, 0x1000000, v%d" % CR)
    a("v_cmp_ne_u32_e64 s[52:53], 0, v217")
    a("v_cmp_ne_u32_e64 s[54:55], 0, v218")
    a("s_and_b64 s[52:53], s[52:53], s[60:61]")
    a("s_and_b64 s[54:55], s[54:55], s[60:61]")
    for (left, mask, rec) in ((True, "s[52:53]", RL), (False, "s[54:55]", CR)):
        side = "L" if left else "R"
        a("s_cmp_eq_u64 %s, 0" % mask)
        a("s_cbranch_scc1 .Lpg_%snw%s%s" % (tag, side, sfx))
        a("s_and_saveexec_b64 s[62:63], %s" % mask)
        a("v_bfe_u32 v217, v%d, 25, 2" % rec)                                       # the line
        if left:
            a("v_and_b32_e32 v221, 63, v222")                                       # a row's line: by column
        else:
            a("v_and_b32_e32 v221, 63, %[row]")                                     # a column's line: by row
        a("v_mul_u32_u24_e32 v217, 0x600, v217")
        a("v_add_u32_e32 v217, %[histb], v217")
        a("v_mad_u32_u24 v221, v221, 24, v217")
        a("ds_write2_b64 v221, %s, %s offset1:1" % (pr(BX), pr(BY)))
        a("ds_write_b64 v221, %s offset:16" % pr(BM))
        a("s_mov_b64 exec, s[62:63]")
        a(".Lpg_%snw%s%s:" % (tag, side, sfx))


def class2(E, k):
    """The merge of a class 2 step: what the diagonal's assist wave staged -- for every multi-edge cell, the best X / Y
    candidate over the edges that do not come from the previous site and the best M over all edge pairs (dp_pipe.hip,
    pipe_assist) -- into the base step's values.  X / Y: the larger of the two, or the staged one alone where the site has
    no edge from the previous site (PS_ONLY, the staged word's sign bit); M as staged."""
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
    a("v_add_u32_e32 v219, s69, %[spxa]")                      # spx[slot][lane]; spy at a 3072-byte stride
    a("s_mov_b32 s72, 64")
    a(".Lpg_c2read%s:" % sfx)
    a("ds_read_b32 v223, v217")                                # flag first: LDS executes in order
    a("ds_read_b64 %s, v218" % pr(ex))
    a("ds_read_b64 %s, v218 offset:6144" % pr(ey))
    a("ds_read_b64 %s, v218 offset:12288" % pr(em))
    a("ds_read_b32 v220, v219")
    a("ds_read_b32 v221, v219 offset:3072")
    a("s_waitcnt lgkmcnt(0)")                                   # (and the two site records)
    a("v_readfirstlane_b32 s68, v223")
    a("s_cmp_lt_i32 s68, %[d]")
    a("s_cbranch_scc0 .Lpg_c2ok%s" % sfx)
    # not staged yet: watch the flag alone (one broadcast read per look: the six reads above, repeated by every waiting
    # wave, take LDS cycles from the assist waves everybody is waiting for), then read the batch again
    a(".Lpg_c2spin%s:" % sfx)
    a("s_sleep 1")
    a("s_sub_i32 s72, s72, 1")
    a("s_cmp_eq_u32 s72, 0")
    a("s_cbranch_scc1 .Lpg_exit%s" % sfx)                      # nothing of this step is committed: the caller's poll takes over
    a("ds_read_b32 v223, v217")
    a("s_waitcnt lgkmcnt(0)")
    a("v_readfirstlane_b32 s68, v223")
    a("s_cmp_lt_i32 s68, %[d]")
    a("s_cbranch_scc1 .Lpg_c2spin%s" % sfx)
    a("s_branch .Lpg_c2read%s" % sfx)
    a(".Lpg_c2ok%s:" % sfx)
    a("v_and_b32_e32 v217, 0x10000, v%d" % RLK[k])
    a("v_and_b32_e32 v218, 0x10000, v%d" % CRK[k])
    a("v_cmp_eq_u32_e64 s[74:75], 0, v217")                     # msL: the left site is a multi-edge one
    a("v_cmp_eq_u32_e64 s[76:77], 0, v218")                     # msR
    a("v_cmp_gt_i32_e64 s[78:79], 0, v220")                     # PS_ONLY of the staged X (bit 31)
    a("v_cmp_gt_i32_e64 s[80:81], 0, v221")                     # ... of the staged Y
    a("v_max_f64 %s, %s, %s" % (pr(T[3]), pr(BX), pr(ex)))
    a("v_max_f64 %s, %s, %s" % (pr(T[4]), pr(BY), pr(ey)))
    a("s_or_b64 s[84:85], s[74:75], s[76:77]")
    for (val, mx, bst, ms, only) in ((ex, T[3], BX, "s[74:75]", "s[78:79]"), (ey, T[4], BY, "s[76:77]", "s[80:81]")):
        a("v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (mx, mx, val, only))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (mx + 1, mx + 1, val + 1, only))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (bst, bst, mx, ms))
        a("v_cndmask_b32_e64 v%d, v%d, v%d, %s" % (bst + 1, bst + 1, mx + 1, ms))
    a("v_cndmask_b32_e64 v%d, v%d, v%d, s[84:85]" % (BM, BM, em))
    a("v_cndmask_b32_e64 v%d, v%d, v%d, s[84:85]" % (BM + 1, BM + 1, em + 1))


def step(E, k):
    """One diagonal.  k = 0 / 1: which half of the unrolled pair (selects descriptor, Q and tM/tX roles)."""
    a = E.a
    cur = 36 if k == 0 else 44
    nxt = 44 if k == 0 else 36
    A = Q[k]          # receives (row-1, j) on d-1
    C = Q[1 - k]      # holds (row-1, j-1) on d-2
    tmx, tmn = TMX[k], TMX[1 - k]
    lo, hi, zlo, zhi, s4, s7 = cur, cur + 1, cur + 2, cur + 3, cur + 4, cur + 7
    sfx = "%d_%%=" % k
    cursb, prevsb = ("%[sb]", "s66") if k == 0 else ("s66", "%[sb]")
    E.cursb = cursb
    a("; ---- diagonal, half %d ----" % k)
    # LDS batch: upstream flag, lane 0's operand, the column record two steps ahead, next step's match terms, the row record
    a("v_add_u32_e32 v217, %s, %%[bpos24]" % prevsb)           # ring row of d-1
    if "A" in EXP:
        a("s_mov_b64 exec, 1")                                   # only lane 0 keeps what it reads here (the shift overwrites the others)
    a("ds_read_b32 v223, %[fup]")
    a("ds_read2_b64 v[%d:%d], v217 offset1:1" % (A[0], A[0] + 3))
    a("ds_read_b64 %s, v217 offset:16" % pr(A[2]))
    if "A" in EXP:
        a("s_mov_b64 exec, -1")
    if "m" in EXP:       # timing experiment: no model pipeline
        a("v_mov_b64_e32 %s, %s" % (pr(tmn), pr(tmx)))
        a("v_mov_b64_e32 %s, %s" % (pr(tmn + 2), pr(tmx + 2)))
    else:
        a("s_add_i32 s69, %[d], 2")
        a("v_sub_u32_e32 v218, s69, %[row]")
        a("v_mad_u32_u24 v219, %[colx], %[S], %[rowx]")            # state pair of (row, d+1-row)
        a("v_and_b32_e32 v218, 0x1ff, v218")
        a("v_and_b32_e32 v219, 0xff, v219")
        a("v_lshl_add_u32 v218, v218, 4, %[bR]")
        a("v_lshl_add_u32 v219, v219, 4, %[bT]")
        a("ds_read_u16 %[colx], v218")
        a("ds_read_b128 v[%d:%d], v219" % (tmn, tmn + 3))
        a("v_and_b32_e32 v221, 0x1ff, %[row]")
        a("v_lshl_add_u32 v221, v221, 4, %[bL]")
        a("ds_read_u16 %[rowx], v221")
    if "B" in EXP:           # timing experiment: four branches that are never taken (what does one cost?)
        a("s_cmp_eq_u32 s67, -2")
        for q in range(4):
            a("s_cbranch_scc1 .Lpg_exit%s" % sfx)
    if "N" in EXP:           # timing experiment: four scalar instructions
        for q in range(4):
            a("s_cmp_eq_u32 s67, -2")
    # Y from P, M from C (no shift needed)
    a("v_add_f64 v[196:197], %s, %%[ng]" % pr(P[2]))          # PM + ng
    a("v_add_f64 v[200:201], %s, %%[gey]" % pr(P[1]))         # PY + ge (the lane's own rate for the y-gap state)
    a("v_max_f64 v[202:203], %s, %s" % (pr(C[0]), pr(C[1])))  # max(CX, CY)
    a("v_add_f64 v[204:205], %s, %s" % (pr(C[2]), pr(tmx)))   # CM + tM
    a("v_max_f64 v[196:197], %s, v[196:197]" % pr(P[0]))      # max(PX, PM + ng)
    a("v_add_f64 v[202:203], v[202:203], %s" % pr(tmx + 2))   # max(CX, CY) + tX
    a("v_add_f64 v[196:197], v[196:197], %[go]")
    a("v_max_f64 %s, v[204:205], v[202:203]" % pr(BM))
    a("v_max_f64 %s, v[200:201], v[196:197]" % pr(BY))
    # the LDS batch and the descriptor of this diagonal (requested a step ago) are here
    a("s_waitcnt lgkmcnt(0)")
    # Four things can keep the step from running -- the diagonal is not class 0..2; the wave's interval ends here or the
    # loader's flags have to be looked at; the downstream wave has not yet read the ring row this step overwrites; the upstream
    # wave has not completed d-1 -- and as a rule none does: ONE branch for the four (a branch that is not taken costs as much as
    # seven scalar instructions here: tools/probe_one_wave.py with the timing variants B and N), on the sign of the OR of four
    # differences; which of them it was is sorted out off the path.
    a("s_and_b32 s73, s%d, 15" % s4)
    upchk = "f" not in EXP and "u" not in EXP
    dnchk = "f" not in EXP and "n" not in EXP
    if upchk:
        a("v_readfirstlane_b32 s68, v223")
    a("s_sub_i32 s72, 2, s73")                                 # < 0: not class 0 .. 2
    a("s_sub_i32 s69, %[stopm1], %[d]")                        # < 0: d >= stop
    a("s_or_b32 s72, s72, s69")
    if dnchk:
        a("s_sub_i32 s69, %%[pdn], s%d" % s7)                   # < 0: the downstream wave's flag has to be looked at
        a("s_or_b32 s72, s72, s69")
    a("s_sub_i32 s69, %[d], 1")
    if upchk:
        a("s_max_i32 %[pup], %[pup], s68")
        a("s_sub_i32 s68, %[pup], s69")                        # < 0: the upstream wave has not completed d-1
        a("s_or_b32 s72, s72, s68")
    if "b" in EXP:
        a("s_mov_b32 s73, 0")
    if "d" in EXP:
        a("s_cmp_eq_u32 s73, 2")
        a("s_cselect_b32 s73, 0, s73")
    a("s_cmp_lt_i32 s72, 0")
    a("s_cbranch_scc1 .Lpg_attn%s" % sfx)
    a(".Lpg_upok%s:" % sfx)
    E.cur = E.ool
    a(".Lpg_attn%s:" % sfx)
    a("s_and_b32 s72, s%d, 15" % s4)
    a("s_cmp_gt_u32 s72, 2")
    a("s_cbranch_scc1 .Lpg_exit%s" % sfx)                      # not class 0 .. 2
    a("s_cmp_gt_i32 %[d], %[stopm1]")
    a("s_cbranch_scc1 .Lpg_exit%s" % sfx)                      # the wave's interval ends, or the loader's flags have to be looked at
    if dnchk:
        a("s_cmp_gt_i32 s%d, %%[pdn]" % s7)
        a("s_cbranch_scc1 .Lpg_dnwait%s" % sfx)                # the downstream wave's flag has to be looked at
    a(".Lpg_dnok%s:" % sfx)
    if upchk:
        a("s_cmp_lt_i32 %[pup], s69")
        a("s_cbranch_scc1 .Lpg_upwait%s" % sfx)
    a("s_branch .Lpg_upok%s" % sfx)
    # ring row reuse: the downstream wave must have completed the last diagonal that reads the row this step overwrites.
    # Where long edges are about it may lag two diagonals at most: look at its flag here, a few times, before giving up
    a(".Lpg_dnwait%s:" % sfx)
    if "k" in EXP:
        a("s_add_u32 %[k2], %[k2], 1")
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
    if "k" in EXP:
        a("s_add_u32 %[k2], %[k2], 0x10000")
    a("s_branch .Lpg_exit%s" % sfx)
    # the upstream wave has not completed d-1 yet: it is, as a rule, a fraction of a step away.  Look again a few times
    # (the flag alone, then lane 0's operand: LDS executes in order) before handing the wait to the caller's poll.
    a(".Lpg_upwait%s:" % sfx)
    if "k" in EXP:
        a("s_add_u32 %[k0], %[k0], 1")
    if "i" in EXP:
        # only a wave with a row about to use (row-1, .) has to wait for the upstream wave (as the kernel's step() does)
        a("s_add_i32 s72, s%d, 1" % hi)
        a("v_cmp_ge_i32_e32 vcc, s72, %[row]")
        a("s_cbranch_vccz .Lpg_upok%s" % sfx)
    a("s_mov_b32 s72, 48")
    a(".Lpg_upretry%s:" % sfx)
    if "k" in EXP:
        a("s_add_u32 %[k1], %[k1], 1")
    a("s_sleep 8" if "z" in EXP else "s_sleep 1")
    a("ds_read_b32 v223, %[fup]")                              # the flag alone (a broadcast read), the operand once it is there
    a("s_waitcnt lgkmcnt(0)")
    a("v_readfirstlane_b32 s68, v223")
    a("s_max_i32 %[pup], %[pup], s68")
    a("s_cmp_lt_i32 %[pup], s69")
    a("s_cbranch_scc0 .Lpg_upgot%s" % sfx)
    a("s_sub_i32 s72, s72, 1")
    a("s_cmp_lg_u32 s72, 0")
    a("s_cbranch_scc1 .Lpg_upretry%s" % sfx)
    if "k" in EXP:
        a("s_add_u32 %[k0], %[k0], 0x10000")
    a("s_branch .Lpg_exit%s" % sfx)
    a(".Lpg_upgot%s:" % sfx)
    a("ds_read2_b64 v[%d:%d], v217 offset1:1" % (A[0], A[0] + 3))
    a("ds_read_b64 %s, v217 offset:16" % pr(A[2]))
    a("s_waitcnt lgkmcnt(0)")
    a("s_branch .Lpg_upok%s" % sfx)
    E.cur = E.L
    # row hand-over
    a("v_cmp_gt_i32_e32 vcc, s%d, %%[row]" % lo)
    a("v_add_u32_e32 v222, 0x100, %[row]")
    a("s_add_u32 s64, %%[sclo], s%d" % zlo)                    # score row of this diagonal
    a("v_cndmask_b32_e32 %[row], %[row], v222, vcc")
    a("s_addc_u32 s65, %%[schi], s%d" % zhi)
    # class 0: the next descriptor now (a whole step to land).  Class 1 / 2: the two site records of this step's cell now,
    # the descriptor after the multi-edge part (its waits would wait for the descriptor as well)
    a("s_cmp_lg_u32 s73, 0")
    a("s_cbranch_scc1 .Lpg_recs%s" % sfx)
    def next_desc():
        if "X" in EXP:       # timing experiment: no scalar load, the descriptor is copied (wrong rows, same schedule)
            for q in range(0, 8, 2):
                a("s_mov_b64 s[%d:%d], s[%d:%d]" % (nxt + q, nxt + q + 1, cur + q, cur + q + 1))
        else:
            a("s_load_dwordx8 s[%d:%d], s[70:71], %s" % (nxt, nxt + 7, "0x20" if k == 0 else "0x40"))

    next_desc()
    a(".Lpg_shift%s:" % sfx)

    def shift_and_x(have_active=False):
        # shift: lane n takes lane n-1's cell, lane 0 keeps what it read from the ring
        for c in range(3):
            if "p" in EXP:   # timing experiment: plain moves instead of the DPP shift
                a("v_mov_b32_e32 v%d, v%d" % (A[c], P[c]))
                a("v_mov_b32_e32 v%d, v%d" % (A[c] + 1, P[c] + 1))
                continue
            a("v_mov_b32_dpp v%d, v%d wave_shr:1 row_mask:0xf bank_mask:0xf" % (A[c], P[c]))
            a("v_mov_b32_dpp v%d, v%d wave_shr:1 row_mask:0xf bank_mask:0xf" % (A[c] + 1, P[c] + 1))
        # X from A; the band limit of this lane
        a("v_add_f64 v[196:197], %s, %%[ng]" % pr(A[2]))          # AM + ng
        if STRIP:
            # the x-gap state's rate of this lane's cell: the terminal one in column 0 (d == row) and column Ly-1
            a("s_sub_i32 s72, %[d], %[lym1]")
            a("v_cmp_eq_u32_e32 vcc, %[d], %[row]")
            a("v_cmp_eq_u32_e64 s[84:85], s72, %[row]")
            if not have_active:
                a("v_cmp_ge_i32_e64 s[60:61], s%d, %%[row]" % hi)      # active: row <= hi
            a("v_max_f64 v[196:197], %s, v[196:197]" % pr(A[1]))  # max(AY, AM + ng)
            a("s_or_b64 vcc, vcc, s[84:85]")
            a("v_add_f64 v[196:197], v[196:197], %[go]")
            a("v_cndmask_b32_e64 v%d, %%[nihi], %%[pihi], s[60:61]" % (LIM + 1))
            a("v_cndmask_b32_e32 v%d, %%[gel], %%[gEl], vcc" % GEX)
            a("v_cndmask_b32_e32 v%d, %%[geh], %%[gEh], vcc" % (GEX + 1))
            a("v_add_f64 v[200:201], %s, %s" % (pr(A[0]), pr(GEX)))    # AX + the lane's rate
        else:
            a("v_add_f64 v[200:201], %s, %%[ge]" % pr(A[0]))          # AX + ge
            if not have_active:
                a("v_cmp_ge_i32_e64 s[60:61], s%d, %%[row]" % hi)          # active: row <= hi
            a("v_max_f64 v[196:197], %s, v[196:197]" % pr(A[1]))      # max(AY, AM + ng)
            a("v_add_f64 v[196:197], v[196:197], %[go]")
            a("v_cndmask_b32_e64 v%d, %%[nihi], %%[pihi], s[60:61]" % (LIM + 1))
        a("v_max_f64 %s, v[200:201], v[196:197]" % pr(BX))

    E.cur = E.ool
    # Class 1 / 2: the two site records of this step's cell.  A class 1 step reads the next step's ahead (c1_ahead): if the
    # step before did (s67 == d), they are here and the step takes the short way -- no read, no wait for one; decode, every
    # operand cell and the records after next requested at once; the shift and the X candidates while those are in flight; one
    # wait.  Otherwise: the records now, shift and X candidates while they come, then the same parts behind their waits.
    a(".Lpg_recs%s:" % sfx)
    a("s_cmp_lg_u32 s67, %[d]")
    a("s_cbranch_scc1 .Lpg_read%s" % sfx)                      # (the step before did not read them ahead: the first step of a run)
    a("s_cmp_eq_u32 s73, 2")
    a("s_cbranch_scc1 .Lpg_slow%s" % sfx)                      # (class 2: the usual way, minus the read)
    if "y" in EXP:           # timing experiment / debugging: never the short way
        a("s_branch .Lpg_slow%s" % sfx)
    # the steady class 1 step falls through to here
    a("v_cmp_ge_i32_e64 s[60:61], s%d, %%[row]" % hi)          # active: row <= hi
    if "E" in EXP:           # (A/B: the next descriptor requested first -- the step's one wait then covers it)
        next_desc()
    c1_issue(E, k, "f")                                        # (a wave without a multi-edge cell finds its masks empty: the decode and the records after next is all it does)
    shift_and_x(have_active=True)
    c1_math(E, k, "f")
    a("s_and_b32 s72, s%d, 0x80020" % s4)                      # a three-edge site (bit 19) or a far history's reader / writer (bit 5) has a cell on this diagonal
    a("s_cbranch_scc1 .Lpg_extra%s" % sfx)
    if "E" not in EXP:
        next_desc()
    a("s_branch .Lpg_commit%s" % sfx)
    a(".Lpg_read%s:" % sfx)
    a("v_and_b32_e32 v218, 0x1ff, %[row]")
    a("v_sub_u32_e32 v219, %[d], %[row]")
    a("v_lshl_add_u32 v218, v218, 4, %[bL]")
    a("v_and_b32_e32 v219, 0x1ff, v219")
    a("ds_read_b128 v[%d:%d], v218" % (RLK[k], RLK[k] + 3))
    a("v_lshl_add_u32 v219, v219, 4, %[bR]")
    a("ds_read_b128 v[%d:%d], v219" % (CRK[k], CRK[k] + 3))
    a(".Lpg_slow%s:" % sfx)
    shift_and_x()                                              # (a copy: the class 0 path runs into its own without a second test of the class)
    a("s_branch .Lpg_c1%s" % sfx)
    E.cur = E.L
    shift_and_x()
    a(".Lpg_commit%s:" % sfx)
    E.cur = E.ool
    a(".Lpg_c1%s:" % sfx)
    a("s_cmp_eq_u32 s73, 2")
    a("s_cbranch_scc1 .Lpg_c2%s" % sfx)
    c1_test(E, k, wait=True)
    a("s_cbranch_vccnz .Lpg_c1go%s" % sfx)
    c1_ahead(E, k)                                             # none of the diagonal's multi-edge cells is this wave's: the next step's records all the same
    a("s_branch .Lpg_c1done%s" % sfx)
    a(".Lpg_c1go%s:" % sfx)
    if "c" in EXP:
        a("s_branch .Lpg_c1done%s" % sfx)
    c1_issue(E, k, "s")
    c1_math(E, k, "s")
    a(".Lpg_c1done%s:" % sfx)
    a("s_and_b32 s72, s%d, 0x80020" % s4)
    a("s_cbranch_scc1 .Lpg_extra%s" % sfx)
    next_desc()
    a("s_branch .Lpg_commit%s" % sfx)
    a(".Lpg_extra%s:" % sfx)
    a("s_bitcmp1_b32 s%d, 19" % s4)
    a("s_cbranch_scc0 .Lpg_hist%s" % sfx)
    if "t" not in EXP:           # (timing experiment 't', WRONG RESULTS: no third pass)
        third_pass(E, k)
    a("s_bitcmp1_b32 s%d, 5" % s4)
    a("s_cbranch_scc0 .Lpg_nohist%s" % sfx)
    a(".Lpg_hist%s:" % sfx)
    if "h" not in EXP:           # (timing experiment 'h', WRONG RESULTS: no history readers / writers on class 1 diagonals)
        hist_tail(E, k, cls2=False)
    a(".Lpg_nohist%s:" % sfx)
    next_desc()
    a("s_branch .Lpg_commit%s" % sfx)
    a(".Lpg_c2%s:" % sfx)
    class2(E, k)
    a("s_bitcmp1_b32 s%d, 5" % s4)
    a("s_cbranch_scc1 .Lpg_hist2%s" % sfx)
    next_desc()
    a("s_branch .Lpg_commit%s" % sfx)
    a(".Lpg_hist2%s:" % sfx)
    hist_tail(E, k, cls2=True)
    next_desc()
    a("s_branch .Lpg_commit%s" % sfx)
    E.cur = E.L
    # results: -inf outside the band
    a("v_subrev_u32_e32 v217, s%d, %%[row]" % lo)              # row - lo
    a("v_min_f64 %s, %s, %s" % (pr(P[0]), pr(BX), pr(LIM)))
    a("v_min_f64 %s, %s, %s" % (pr(P[1]), pr(BY), pr(LIM)))
    a("v_min_f64 %s, %s, %s" % (pr(P[2]), pr(BM), pr(LIM)))
    a("v_add_u32_e32 v219, %s, %%[tid24]" % cursb)
    a("v_mul_u32_u24_e32 v218, 24, v217")
    a("v_mov_b32_e32 v220, %[d]")
    if "R" in EXP or "Q" in EXP:   # timing experiments: ring writes by the in-band lanes only (R) / by lane 63 only (Q)
        if "R" in EXP:
            a("s_and_saveexec_b64 s[62:63], s[60:61]")
        else:
            a("s_mov_b64 s[62:63], exec")
            a("s_mov_b64 exec, 1")
            a("s_lshl_b64 exec, exec, 63")
        a("ds_write2_b64 v219, %s, %s offset1:1" % (pr(P[0]), pr(P[1])))
        a("ds_write_b64 v219, %s offset:16" % pr(P[2]))
        a("s_mov_b64 exec, s[62:63]")
    elif "r" not in EXP:   # (timing experiment r: no ring writes)
        a("ds_write2_b64 v219, %s, %s offset1:1" % (pr(P[0]), pr(P[1])))
        a("ds_write_b64 v219, %s offset:16" % pr(P[2]))
    if "s" not in EXP:   # (timing experiment s: no HBM stores)
        a("s_and_saveexec_b64 s[62:63], s[60:61]")
        # (row strips: written through to memory -- the strip below may run on another XCD, whose L2 is not this one's)
        wt = " sc1" if STRIP else ""
        a("global_store_dwordx4 v218, v[%d:%d], s[64:65]%s" % (P[0], P[0] + 3, wt))
        a("global_store_dwordx2 v218, %s, s[64:65] offset:16%s" % (pr(P[2]), wt))
        a("s_mov_b64 exec, s[62:63]")
        a("s_waitcnt vmcnt(16)")                               # all but the last 8 steps' stores have retired (far reads rely on it)
    a("ds_write_b32 %[fme], v220")                             # progress: after the ring writes (a wave's LDS operations execute in order)
    # next diagonal
    a("s_add_i32 %[d], %[d], 1")
    a("s_add_i32 %s, %s, 0x1800" % (prevsb, cursb))            # the next diagonal's ring row, into the register that held the previous one's
    a("s_cmp_eq_u32 %s, %s" % (prevsb, RING_BYTES))
    a("s_cselect_b32 %s, 0, %s" % (prevsb, prevsb))
    if k == 1:                                                  # the descriptor pointer moves once per pair (half 0 reads at +0x20, half 1 at +0x40)
        a("s_add_u32 s70, s70, 0x40")
        a("s_addc_u32 s71, s71, 0")


def main():
    E = Emit()
    a = E.a
    a("; ==== class 0 / 1 / 2 run of a compute wave: generated by tools/gen_hot_asm.py ====")
    for c in range(3):
        a("v_mov_b64_e32 %s, %%[p%d]" % (pr(P[c]), c))
        a("v_mov_b64_e32 %s, %%[c%d]" % (pr(Q[1][c]), c))       # half 0 reads C from Q1
    a("v_mov_b64_e32 %s, %%[tm]" % pr(TMX[0]))
    a("v_mov_b64_e32 %s, %%[tx]" % pr(TMX[0] + 2))
    a("v_mov_b32_e32 v%d, 0" % LIM)
    a("s_mov_b64 s[70:71], %[dptr]")
    a("s_mov_b32 s67, -1")                                      # no site records read ahead
    a("s_load_dwordx8 s[36:43], s[70:71], 0x0")
    a("s_sub_i32 s66, %[sb], 0x1800")                           # ring row of the diagonal before
    a("s_cmp_lt_i32 s66, 0")
    a("s_cselect_b32 s66, %s, s66" % RING_LAST)
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
    a("s_mov_b32 %[sb], s66")                                  # (half 1's ring row)
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
