"""TEST INFRASTRUCTURE ONLY: ctypes front end of the CPU oracle (oracle_dp.cpp, oracle_host.cpp).

PARITY UNPINNED -- see the header of oracle_dp.cpp.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; the product (pagan2-msa_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from pagan2_msa_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libpagan_oracle.so")
_lib = None

DNA_ALPHABET = "ACGTRYMKWSBDHVN"      # Model_factory::dna_full_char_alphabet, model_factory.cpp:103


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("oracle_dp.cpp", "oracle_host.cpp", "oracle_model.cpp", "oracle_fb.cpp", "wag_data.h", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "pagan_dp.h"))
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        gp, mp, bp, op, rp = (C.POINTER(abi.CGraph), C.POINTER(abi.CModel), C.POINTER(abi.CBand),
                              C.POINTER(abi.COpts), C.POINTER(abi.CResult))
        i32p, f32p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.oracle_dp_align.argtypes = [gp, gp, mp, bp, op, rp]
        L.oracle_dp_align.restype = C.c_int
        L.oracle_result_free.argtypes = [rp]
        L.oracle_result_free.restype = None
        L.oracle_graph_leaf.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.oracle_graph_leaf.restype = C.c_void_p
        L.oracle_graph_set_state.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_graph_set_state.restype = None
        L.oracle_graph_free.argtypes = [C.c_void_p]
        L.oracle_graph_free.restype = None
        for f in ("oracle_graph_n_sites", "oracle_graph_n_edges", "oracle_graph_n_bwd"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.oracle_graph_flatten.argtypes = [C.c_void_p, i32p, i32p, i32p, f32p, i32p]
        L.oracle_graph_flatten.restype = None
        L.oracle_graph_attrs.argtypes = [C.c_void_p, i32p, f32p, i32p, f32p]
        L.oracle_graph_attrs.restype = None
        L.oracle_graph_fwd.argtypes = [C.c_void_p, i32p, i32p]
        L.oracle_graph_fwd.restype = None
        L.oracle_graph_mark_used.argtypes = [C.c_void_p, C.c_int, i32p]
        L.oracle_graph_mark_used.restype = None
        L.oracle_graph_parent.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.CCol), C.c_int, C.c_float,
                                          C.c_float, i32p, C.c_int, C.c_int, C.c_int]
        L.oracle_graph_parent.restype = C.c_void_p
        L.oracle_graph_string.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p]
        L.oracle_graph_string.restype = C.c_int
        L.oracle_define_tunnel.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                           C.c_int, i32p, i32p]
        L.oracle_define_tunnel.restype = C.c_int
        L.oracle_eliminate_bad_hits.argtypes = [i32p, C.c_int, C.c_int, C.c_int]
        L.oracle_eliminate_bad_hits.restype = C.c_int
        L.oracle_tunnel_overlapping.argtypes = [i32p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, i32p, i32p, i32p, C.c_int]
        L.oracle_tunnel_overlapping.restype = C.c_int
        L.oracle_force_gap.argtypes = [i32p, i32p, C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_force_gap.restype = C.c_int
        L.oracle_dna_parsimony.argtypes = [i32p]
        L.oracle_dna_parsimony.restype = None
        L.oracle_dna_model.argtypes = [f32p, C.c_double, C.c_int, f32p, f32p]
        L.oracle_dna_model.restype = C.c_int
        L.oracle_protein_model.argtypes = [C.c_double, f32p, f32p, i32p]
        L.oracle_protein_model.restype = C.c_int
        L.oracle_codon_model.argtypes = [C.c_double, f32p, f32p, i32p]
        L.oracle_codon_model.restype = C.c_int
        L.oracle_codon_alphabet.argtypes = [C.c_char_p, i32p]
        L.oracle_codon_alphabet.restype = C.c_int
        L.oracle_codon_translate.argtypes = [C.c_char_p, C.c_char_p]
        L.oracle_codon_translate.restype = C.c_int
        L.oracle_codon_states.argtypes = [C.c_char_p, i32p]
        L.oracle_codon_states.restype = C.c_int
        L.oracle_eigen_qrev.argtypes = [f64p, f64p, C.c_int, f64p, f64p, f64p]
        L.oracle_eigen_qrev.restype = C.c_int
        L.oracle_model_prob.argtypes = [C.c_int, f32p, C.c_double, f32p, f32p]
        L.oracle_model_prob.restype = C.c_int
        L.oracle_fb.argtypes = [gp, gp, C.c_int32, f32p, C.c_float, C.c_float, C.c_float, bp, C.c_int, f64p, f64p, f64p, f64p]
        L.oracle_fb.restype = C.c_int
        L.oracle_sample_path.argtypes = [gp, gp, C.c_int32, f32p, C.c_float, C.c_float, C.c_float, f64p, f64p, C.c_int,
                                         i32p, i32p]
        L.oracle_sample_path.restype = C.c_int
        _lib = L
    return _lib


def dp_align(left, right, model, band=None, flags=0):
    """CPU restatement of Viterbi_alignment::align; same inputs/outputs as pagan2_msa_amd.align."""
    L = lib()
    opts = abi.COpts(flags, -1)
    res = abi.CResult()
    rc = L.oracle_dp_align(C.byref(left.c), C.byref(right.c), C.byref(model.c),
                           C.byref(band.c) if band is not None else None, C.byref(opts), C.byref(res))
    try:
        if rc != 0:
            raise RuntimeError("oracle_dp_align failed with code %d" % rc)
        return abi.Result(res)
    finally:
        L.oracle_result_free(C.byref(res))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OGraph:
    """Handle on an oracle-side Sequence graph (linked edge lists as in the reference)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def leaf(cls, seq, alphabet=DNA_ALPHABET, flags=0):
        return cls(lib().oracle_graph_leaf(seq.encode(), alphabet.encode(), flags))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_graph_free(self.h)
            self.h = None

    def flatten(self):
        L = lib()
        ns, ne, nb = L.oracle_graph_n_sites(self.h), L.oracle_graph_n_edges(self.h), L.oracle_graph_n_bwd(self.h)
        state = np.zeros(ns, np.int32)
        off = np.zeros(ns + 1, np.int32)
        src = np.zeros(max(nb, 1), np.int32)
        lw = np.zeros(max(nb, 1), np.float32)
        eid = np.zeros(max(nb, 1), np.int32)
        L.oracle_graph_flatten(self.h, _ip(state), _ip(off), _ip(src), _fp(lw), _ip(eid))
        return abi.Graph(state, off, src[:nb], lw[:nb], eid[:nb], n_edges=ne)

    def attrs(self):
        L = lib()
        ns, ne = L.oracle_graph_n_sites(self.h), L.oracle_graph_n_edges(self.h)
        sa = np.zeros((ns, 8), np.int32)
        sd = np.zeros(ns, np.float32)
        ea = np.zeros((max(ne, 1), 6), np.int32)
        ef = np.zeros((max(ne, 1), 3), np.float32)
        L.oracle_graph_attrs(self.h, _ip(sa), _fp(sd), _ip(ea), _fp(ef))
        return sa, sd, ea[:ne], ef[:ne]

    def fwd(self):
        L = lib()
        ns, ne = L.oracle_graph_n_sites(self.h), L.oracle_graph_n_edges(self.h)
        off = np.zeros(ns + 1, np.int32)
        eid = np.zeros(max(ne, 1), np.int32)
        L.oracle_graph_fwd(self.h, _ip(off), _ip(eid))
        return off, eid[:off[-1]]

    def set_state(self, pos, state):
        lib().oracle_graph_set_state(self.h, int(pos), int(state))

    def mark_used(self, eids):
        e = np.ascontiguousarray(eids, np.int32)
        lib().oracle_graph_mark_used(self.h, int(e.shape[0]), _ip(e))

    def string(self, with_gaps, alphabet=DNA_ALPHABET):
        L = lib()
        buf = C.create_string_buffer(L.oracle_graph_n_sites(self.h) + 1)
        n = L.oracle_graph_string(self.h, 1 if with_gaps else 0, alphabet.encode(), buf)
        return buf.raw[:n].decode()

    @classmethod
    def parent(cls, left, right, result, lbl, rbl, parsimony, char_as, flags=0):
        """left/right: OGraph children; result: abi.Result of their alignment."""
        left.mark_used(result.left_used)
        right.mark_used(result.right_used)
        cols = np.ascontiguousarray(result.cols, np.int32)
        pars = np.ascontiguousarray(parsimony, np.int32)
        S = int(round(pars.size ** 0.5))
        h = lib().oracle_graph_parent(left.h, right.h, C.cast(_ip(cols), C.POINTER(abi.CCol)), int(cols.shape[0]),
                                      lbl, rbl, _ip(pars), S, char_as, flags)
        return cls(h)


def dna_parsimony():
    t = np.zeros(225, np.int32)
    lib().oracle_dna_parsimony(_ip(t))
    return t


PROTEIN_ALPHABET = "ARNDCQEGHILKMFPSTWYV"      # Model_factory::protein_char_alphabet, model_factory.cpp:104


def protein_leaf_alphabet():
    """Model_factory::get_protein_full_char_alphabet (model_factory.h:144-155): 20 residues, X, then the 190
    pair codes written as the lower-case first residue."""
    a = PROTEIN_ALPHABET
    return a + "X" + "".join(a[i].lower() for i in range(19) for _ in range(i + 1, 20))


def dna_model(base_freq, dist, pileup=False):
    """(abi.Model, params) of Model_factory::dna_model + alignment_model, restated in oracle_model.cpp."""
    bf = np.ascontiguousarray(base_freq, np.float32)
    table = np.zeros(225, np.float32)
    params = np.zeros(4, np.float32)
    lib().oracle_dna_model(_fp(bf), float(dist), 1 if pileup else 0, _fp(table), _fp(params))
    return abi.Model(table.reshape(15, 15).T, *params)


def protein_model(dist):
    """(abi.Model, parsimony[211*211]) of Model_factory::protein_model (WAG) + alignment_model."""
    table = np.zeros(211 * 211, np.float32)
    params = np.zeros(4, np.float32)
    pars = np.zeros(211 * 211, np.int32)
    lib().oracle_protein_model(float(dist), _fp(table), _fp(params), _ip(pars))
    return abi.Model(table.reshape(211, 211).T, *params), pars


CODON_STATES = 61 + 1 + 1830                    # sense codons, NNN, one code per unordered codon pair


def codon_model(dist):
    """(abi.Model, parsimony[1892*1892]) of Model_factory::codon_model (Kosiol & Goldman) + alignment_model."""
    S = CODON_STATES
    table = np.zeros(S * S, np.float32)
    params = np.zeros(4, np.float32)
    pars = np.zeros(S * S, np.int32)
    lib().oracle_codon_model(float(dist), _fp(table), _fp(params), _ip(pars))
    return abi.Model(table.reshape(S, S).T, *params), pars


def codon_alphabet():
    """(the 1892 three-letter names an ancestral state prints as, mostcommon[61*61])"""
    buf = C.create_string_buffer(3 * CODON_STATES + 1)
    mc = np.zeros(61 * 61, np.int32)
    n = lib().oracle_codon_alphabet(buf, _ip(mc))
    txt = buf.value.decode()
    return [txt[3 * k:3 * k + 3] for k in range(n)], mc


def codon_states(nucleotides):
    """Sequence::create_codon_sequence: the state of every triplet (61 = NNN for what is not a sense codon)."""
    out = np.zeros(len(nucleotides) // 3 + 2, np.int32)
    n = lib().oracle_codon_states(nucleotides.encode(), _ip(out))
    return out[:n].copy()


def codon_translate(codon_string):
    """Codon_translation::gapped_DNA_to_protein: one letter per triplet, X for what the table does not hold."""
    buf = C.create_string_buffer(len(codon_string) // 3 + 2)
    n = lib().oracle_codon_translate(codon_string.encode(), buf)
    return buf.raw[:n].decode()


def model_prob(data_type, dist, base_freq=None):
    """abi.ModelProb (Evol_model::score / gap_open / gap_ext / non_gap) restated in oracle_model.cpp."""
    S = CODON_STATES if data_type == 3 else 211 if data_type == 2 else 15
    score = np.zeros(S * S, np.float32)
    params = np.zeros(3, np.float32)
    bf = np.ascontiguousarray(base_freq if base_freq is not None else [0.25] * 4, np.float32)
    lib().oracle_model_prob(int(data_type), _fp(bf), float(dist), _fp(score), _fp(params))
    return abi.ModelProb(score.reshape(S, S).T, *params)


def fb(left, right, mp, band=None, log_space=True, matrices=True):
    """Forward/backward restated (oracle_fb.cpp).  Returns (log_fwd, log_bwd, posterior[Lx,Ly,3], log_f[Lx,Ly,3])."""
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    lf, lb = C.c_double(), C.c_double()
    post = np.zeros((Lx, Ly, 3)) if matrices else None
    logf = np.zeros((Lx, Ly, 3)) if matrices else None
    dp = C.POINTER(C.c_double)
    rc = lib().oracle_fb(C.byref(left.c), C.byref(right.c), mp.n_states, _fp(mp.table), mp.gap_open, mp.gap_ext, mp.non_gap,
                         C.byref(band.c) if band is not None else None, 1 if log_space else 0, C.byref(lf), C.byref(lb),
                         post.ctypes.data_as(dp) if matrices else None, logf.ctypes.data_as(dp) if matrices else None)
    if rc != 0:
        raise RuntimeError("oracle_fb failed: %d" % rc)
    return lf.value, lb.value, post, logf


def sample_path(left, right, mp, log_f, u):
    """sample_new_path restated: visited cells end -> start as rows (i, j, state), and the end pick (state, i, j)."""
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    cells = np.zeros((Lx + Ly + 2, 3), np.int32)
    end = np.zeros(3, np.int32)
    lf = np.ascontiguousarray(log_f, np.float64)
    uu = np.ascontiguousarray(u, np.float64)
    dp = C.POINTER(C.c_double)
    n = lib().oracle_sample_path(C.byref(left.c), C.byref(right.c), mp.n_states, _fp(mp.table), mp.gap_open, mp.gap_ext,
                                 mp.non_gap, lf.ctypes.data_as(dp), uu.ctypes.data_as(dp), int(uu.shape[0]), _ip(cells), _ip(end))
    if n < 0:
        raise RuntimeError("oracle_sample_path failed")
    return cells[:n], end


def eigen_qrev(Q, pi):
    """Eigen::eigenQREV restated: (root, U, V) with Q = U diag(root) V."""
    Q = np.ascontiguousarray(Q, np.float64)
    pi = np.ascontiguousarray(pi, np.float64)
    n = pi.shape[0]
    root, U, V = np.zeros(n), np.zeros((n, n)), np.zeros((n, n))
    dp = C.POINTER(C.c_double)
    lib().oracle_eigen_qrev(Q.ctypes.data_as(dp), pi.ctypes.data_as(dp), n, root.ctypes.data_as(dp),
                            U.ctypes.data_as(dp), V.ctypes.data_as(dp))
    return root, U, V


def eliminate_bad_hits(hits, thr_total=50, thr_partly=400):
    h = np.ascontiguousarray(hits, np.int32).reshape(-1, 4).copy()
    n = lib().oracle_eliminate_bad_hits(_ip(h), int(h.shape[0]), thr_total, thr_partly)
    return h[:n].copy()


def tunnel_from_hits(hits, g1, g2, width=15):
    """Find_anchors::define_tunnel alone on a given hit list (positions in the ungapped strings)."""
    h = np.ascontiguousarray(hits, np.int32).reshape(-1, 4)
    up = np.zeros(len(g1) + 1, np.int32)
    lo = np.zeros(len(g1) + 1, np.int32)
    L = lib()
    L.oracle_tunnel_from_hits.restype = C.c_int
    L.oracle_tunnel_from_hits(_ip(h), int(h.shape[0]), g1.encode(), g2.encode(), int(width), _ip(up), _ip(lo))
    return abi.Band(up, lo)


def order_conflicts(hits, len1, len2, trim=5):
    """Find_anchors::check_hits_order_conflict alone: the surviving hits."""
    h = np.ascontiguousarray(hits, np.int32).reshape(-1, 4).copy()
    L = lib()
    L.oracle_order_conflicts.restype = C.c_int
    n = L.oracle_order_conflicts(_ip(h), int(h.shape[0]), int(len1), int(len2), int(trim))
    return h[:n].copy()


def tunnel_overlapping(hits, g1, g2, width=15):
    h = np.ascontiguousarray(hits, np.int32).reshape(-1, 4)
    up = np.zeros(len(g1) + 1, np.int32)
    lo = np.zeros(len(g1) + 1, np.int32)
    cap = len(g1) + 2
    blocks = np.zeros((cap, 4), np.int32)
    n = lib().oracle_tunnel_overlapping(_ip(h), int(h.shape[0]), g1.encode(), g2.encode(), width, _ip(up), _ip(lo), _ip(blocks), cap)
    return abi.Band(up, lo), blocks[:n].copy()


def force_gap(band, blocks, threshold=40000, width=15, wide=False):
    up, lo = band.upper.copy(), band.lower.copy()
    b = np.ascontiguousarray(blocks, np.int32).reshape(-1, 4)
    done = lib().oracle_force_gap(_ip(up), _ip(lo), int(up.shape[0]), _ip(b), int(b.shape[0]), threshold, width, 1 if wide else 0)
    return bool(done), abi.Band(up, lo), (b[:-1].copy() if done else b.copy())


def define_tunnel(left, right, min_length=30, trim=5, width=15, alphabet=DNA_ALPHABET):
    """--use-prefix-anchors band for two OGraphs (Viterbi_alignment::define_tunnel)."""
    s1, s2 = left.string(False, alphabet), right.string(False, alphabet)
    g1, g2 = left.string(True, alphabet), right.string(True, alphabet)
    up = np.zeros(len(g1) + 1, np.int32)
    lo = np.zeros(len(g1) + 1, np.int32)
    nh = lib().oracle_define_tunnel(s1.encode(), s2.encode(), g1.encode(), g2.encode(), min_length, trim, width,
                                    _ip(up), _ip(lo))
    return abi.Band(up, lo), nh
