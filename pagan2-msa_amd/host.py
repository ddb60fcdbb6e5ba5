"""ctypes front end of include/pagan_host.h: the host-side guide-tree walk (Node mirror), the
host graph builder (Sequence / build_ancestral_sequence mirror), anchors and the DNA model.
All of it lives in libpagan_dp.so; the DP itself always runs on the GPU."""
import ctypes as C

import numpy as np

from . import abi

PAGAN_E_TREE = -20
PAGAN_E_MEMCAP = -21
DNA_FULL = "ACGTRYMKWSBDHVN"

_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


class CMsaOpts(C.Structure):
    _fields_ = [("use_anchors", C.c_int32), ("anchors_offset", C.c_int32), ("prefix_hit_length", C.c_int32),
                ("hit_trim", C.c_int32), ("dp_flags", C.c_uint32), ("leaf_flags", C.c_int32),
                ("keep_all_edges", C.c_int32), ("n_devices", C.c_int32), ("first_device", C.c_int32),
                ("host_threads", C.c_int32), ("truncate_branches", C.c_float), ("device_mem_budget", C.c_int64),
                ("data_type", C.c_int32), ("pileup_rates", C.c_int32), ("anchor_mode", C.c_int32),
                ("overlap_total", C.c_int32), ("overlap_partly", C.c_int32), ("force_gap", C.c_int32),
                ("force_gap_threshold", C.c_int32), ("force_gap_wide", C.c_int32), ("mostcommon", C.c_int32)]


class CNodeInfo(C.Structure):
    _fields_ = [("node", C.c_int32), ("left", C.c_int32), ("right", C.c_int32), ("level", C.c_int32),
                ("left_sites", C.c_int32), ("right_sites", C.c_int32), ("sites", C.c_int32), ("n_hits", C.c_int32),
                ("cells", C.c_int64), ("dist", C.c_double), ("score", C.c_double), ("status", C.c_int32),
                ("n_forced_gaps", C.c_int32)]


class CTiming(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("total_s", "model_s", "anchors_s", "dp_wall_s", "dp_fill_dev_s",
                                           "dp_trace_dev_s", "build_s")]


# pagan_batch_fn (include/pagan_host.h): the test seam's callback type
BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_int32, C.POINTER(abi.CJob), C.POINTER(abi.COpts), C.POINTER(abi.CResult), C.c_void_p)

class CPileupOpts(C.Structure):
    _fields_ = [("leaf_flags", C.c_int32), ("dp_flags", C.c_uint32), ("query_distance", C.c_float), ("min_overlap", C.c_float),
                ("min_identity", C.c_float), ("use_anchors", C.c_int32), ("anchors_offset", C.c_int32),
                ("prefix_hit_length", C.c_int32), ("hit_trim", C.c_int32), ("device", C.c_int32)]


class CPileupStep(C.Structure):
    _fields_ = [("read", C.c_int32), ("accepted", C.c_int32), ("overlap", C.c_float), ("identity", C.c_float),
                ("aligned", C.c_int32), ("matched", C.c_int32), ("read_length", C.c_int32), ("left_sites", C.c_int32),
                ("right_sites", C.c_int32), ("status", C.c_int32), ("n_cols", C.c_int32), ("score", C.c_double),
                ("cells", C.c_int64)]


_declared = False


def _lib():
    from . import lib
    L = lib()
    global _declared
    if not _declared:
        vp = C.c_void_p
        L.pagan_msa_default_opts.argtypes = [C.POINTER(CMsaOpts)]
        L.pagan_msa_default_opts.restype = None
        L.pagan_msa_create.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_char_p,
                                       C.POINTER(CMsaOpts), C.POINTER(vp)]
        L.pagan_msa_create.restype = C.c_int
        L.pagan_msa_align.argtypes = [vp]
        L.pagan_msa_align.restype = C.c_int
        L.pagan_msa_n_internal.argtypes = [vp]
        L.pagan_msa_n_internal.restype = C.c_int
        L.pagan_msa_node_info.argtypes = [vp, C.c_int32, C.POINTER(CNodeInfo)]
        L.pagan_msa_node_info.restype = C.c_int
        L.pagan_msa_node_job.argtypes = [vp, C.c_int32, C.POINTER(abi.CJob)]
        L.pagan_msa_node_job.restype = C.c_int
        L.pagan_msa_node_result.argtypes = [vp, C.c_int32, C.POINTER(abi.CResult)]
        L.pagan_msa_node_result.restype = C.c_int
        L.pagan_msa_timing_get.argtypes = [vp, C.POINTER(CTiming)]
        L.pagan_msa_timing_get.restype = C.c_int
        L.pagan_msa_alignment_length.argtypes = [vp]
        L.pagan_msa_alignment_length.restype = C.c_int
        L.pagan_msa_alignment_row.argtypes = [vp, C.c_int32, C.c_char_p]
        L.pagan_msa_alignment_row.restype = C.c_int
        L.pagan_msa_write_fasta.argtypes = [vp, C.c_char_p, C.c_int32]
        L.pagan_msa_write_fasta.restype = C.c_int
        L.pagan_msa_write_fasta_nodes.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32]
        L.pagan_msa_write_fasta_nodes.restype = C.c_int
        L.pagan_msa_node_graph.argtypes = [vp, C.c_int32]
        L.pagan_msa_node_graph.restype = vp
        L.pagan_msa_destroy.argtypes = [vp]
        L.pagan_msa_destroy.restype = None
        L.pagan_hgraph_leaf.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
        L.pagan_hgraph_leaf.restype = vp
        L.pagan_hgraph_parent.argtypes = [vp, vp, C.POINTER(abi.CResult), C.c_float, C.c_float, _i32p, C.c_int32,
                                          C.c_int32, C.c_int32]
        L.pagan_hgraph_parent.restype = vp
        L.pagan_hgraph_parent_device.argtypes = [vp, vp, C.POINTER(abi.CResult), C.c_float, C.c_float, _i32p, C.c_int32,
                                                 C.c_int32, C.c_int32, _i32p]
        L.pagan_hgraph_parent_device.restype = vp
        L.pagan_parents_device_calls.argtypes = []
        L.pagan_parents_device_calls.restype = C.c_longlong
        L.pagan_hgraph_view.argtypes = [vp, C.POINTER(abi.CGraph)]
        L.pagan_hgraph_view.restype = None
        L.pagan_hgraph_attrs.argtypes = [vp, _i32p, _f32p, _i32p, _f32p]
        L.pagan_hgraph_attrs.restype = None
        L.pagan_hgraph_fwd.argtypes = [vp, _i32p, _i32p]
        L.pagan_hgraph_fwd.restype = None
        L.pagan_hgraph_string.argtypes = [vp, C.c_int32, C.c_char_p, C.c_char_p]
        L.pagan_hgraph_string.restype = C.c_int
        L.pagan_hgraph_free.argtypes = [vp]
        L.pagan_hgraph_free.restype = None
        L.pagan_define_tunnel.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32,
                                          C.c_int32, _i32p, _i32p]
        L.pagan_define_tunnel.restype = C.c_int
        L.pagan_assign_units.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.c_int32, _i32p]
        L.pagan_assign_units.restype = None
        L.pagan_prefix_hits.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, _i32p, C.c_int32]
        L.pagan_prefix_hits.restype = C.c_int
        L.pagan_anchors_device_calls.argtypes = []
        L.pagan_anchors_device_calls.restype = C.c_longlong
        L.pagan_drop_bad_hits.argtypes = [_i32p, C.c_int32, C.c_int32, C.c_int32]
        L.pagan_drop_bad_hits.restype = C.c_int
        L.pagan_define_tunnel_overlapping.argtypes = [_i32p, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32, _i32p, _i32p,
                                                      _i32p, C.c_int32]
        L.pagan_define_tunnel_overlapping.restype = C.c_int
        L.pagan_force_gap.argtypes = [_i32p, _i32p, C.c_int32, _i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.pagan_force_gap.restype = C.c_int
        L.pagan_dna_model.argtypes = [_f32p, C.c_double, _f32p, _f32p, _i32p]
        L.pagan_dna_model.restype = C.c_int
        L.pagan_protein_model.argtypes = [C.c_double, _f32p, _f32p, _i32p]
        L.pagan_protein_model.restype = C.c_int
        L.pagan_codon_model.argtypes = [C.c_double, _f32p, _f32p, _i32p]
        L.pagan_codon_model.restype = C.c_int
        L.pagan_codon_alphabet.argtypes = [C.c_char_p, _i32p]
        L.pagan_codon_alphabet.restype = C.c_int
        L.pagan_codon_translate.argtypes = [C.c_char_p, C.c_char_p]
        L.pagan_codon_translate.restype = C.c_int
        L.pagan_codon_states.argtypes = [C.c_char_p, _i32p]
        L.pagan_codon_states.restype = C.c_int
        L.pagan_hgraph_leaf_codon.argtypes = [C.c_char_p]
        L.pagan_hgraph_leaf_codon.restype = vp
        L.pagan_model_prob_table.argtypes = [C.c_int32, _f32p, C.c_double, _f32p, _f32p]
        L.pagan_model_prob_table.restype = C.c_int
        L.pagan_model_alphabets.argtypes = [C.c_int32, C.c_char_p, C.c_char_p]
        L.pagan_model_alphabets.restype = C.c_int
        f64p = C.POINTER(C.c_double)
        L.pagan_eigen_qrev.argtypes = [f64p, f64p, C.c_int32, f64p, f64p, f64p]
        L.pagan_eigen_qrev.restype = C.c_int
        L.pagan_msa_ready.argtypes = [vp, _i32p, C.c_int32]
        L.pagan_msa_ready.restype = C.c_int
        L.pagan_msa_remaining.argtypes = [vp]
        L.pagan_msa_remaining.restype = C.c_int
        L.pagan_msa_node_cost.argtypes = [vp, C.c_int32]
        L.pagan_msa_node_cost.restype = C.c_int64
        L.pagan_msa_align_nodes.argtypes = [vp, C.c_int32, _i32p]
        L.pagan_msa_align_nodes.restype = C.c_int
        L.pagan_msa_export_result.argtypes = [vp, C.c_int32, C.c_void_p, C.c_int64]
        L.pagan_msa_export_result.restype = C.c_int64
        L.pagan_msa_import_result.argtypes = [vp, C.c_void_p, C.c_int64]
        L.pagan_msa_import_result.restype = C.c_int
        L.pagan_msa_finish.argtypes = [vp]
        L.pagan_msa_finish.restype = C.c_int
        L.pagan_msa_finish_lazy.argtypes = [vp]
        L.pagan_msa_finish_lazy.restype = C.c_int
        L.pagan_msa_parents_built.argtypes = [vp]
        L.pagan_msa_parents_built.restype = C.c_int
        L.pagan_msa_data_type.argtypes = [vp]
        L.pagan_msa_data_type.restype = C.c_int
        L.pagan_msa_node_device.argtypes = [vp, C.c_int32]
        L.pagan_msa_node_device.restype = C.c_int
        L.pagan_msa_set_batch_backend.argtypes = [vp, BATCH_FN, C.c_void_p]
        L.pagan_msa_set_batch_backend.restype = C.c_int
        L.pagan_pileup_default_opts.argtypes = [C.POINTER(CPileupOpts)]
        L.pagan_pileup_default_opts.restype = None
        L.pagan_pileup_create.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(CPileupOpts), C.POINTER(vp)]
        L.pagan_pileup_create.restype = C.c_int
        L.pagan_pileup_align.argtypes = [vp]
        L.pagan_pileup_align.restype = C.c_int
        L.pagan_pileup_n_steps.argtypes = [vp]
        L.pagan_pileup_n_steps.restype = C.c_int
        L.pagan_pileup_step_info.argtypes = [vp, C.c_int32, C.POINTER(CPileupStep)]
        L.pagan_pileup_step_info.restype = C.c_int
        L.pagan_pileup_step_job.argtypes = [vp, C.c_int32, C.POINTER(abi.CJob)]
        L.pagan_pileup_step_job.restype = C.c_int
        L.pagan_pileup_step_result.argtypes = [vp, C.c_int32, C.POINTER(abi.CResult)]
        L.pagan_pileup_step_result.restype = C.c_int
        L.pagan_pileup_alignment_length.argtypes = [vp]
        L.pagan_pileup_alignment_length.restype = C.c_int
        L.pagan_pileup_alignment_row.argtypes = [vp, C.c_int32, C.c_char_p]
        L.pagan_pileup_alignment_row.restype = C.c_int
        L.pagan_pileup_set_batch_backend.argtypes = [vp, BATCH_FN, C.c_void_p]
        L.pagan_pileup_set_batch_backend.restype = C.c_int
        L.pagan_pileup_destroy.argtypes = [vp]
        L.pagan_pileup_destroy.restype = None
        _declared = True
    return L


HOST_EXPORTED = ["pagan_assign_units", "pagan_msa_default_opts", "pagan_msa_create", "pagan_msa_align", "pagan_msa_n_internal",
                 "pagan_msa_node_info", "pagan_msa_node_job", "pagan_msa_node_result", "pagan_msa_timing_get",
                 "pagan_msa_alignment_length", "pagan_msa_alignment_row", "pagan_msa_write_fasta", "pagan_msa_write_fasta_nodes",
                 "pagan_msa_node_graph", "pagan_msa_finish_lazy", "pagan_msa_parents_built",
                 "pagan_msa_destroy", "pagan_hgraph_leaf", "pagan_hgraph_parent", "pagan_hgraph_parent_device", "pagan_parents_device_calls", "pagan_hgraph_view",
                 "pagan_hgraph_attrs", "pagan_hgraph_fwd", "pagan_hgraph_string", "pagan_hgraph_free",
                 "pagan_define_tunnel", "pagan_prefix_hits", "pagan_anchors_device_calls", "pagan_drop_bad_hits", "pagan_define_tunnel_overlapping",
                 "pagan_force_gap", "pagan_dna_model", "pagan_protein_model", "pagan_model_prob_table", "pagan_model_alphabets",
                 "pagan_codon_model", "pagan_codon_alphabet", "pagan_codon_states", "pagan_codon_translate", "pagan_hgraph_leaf_codon",
                 "pagan_eigen_qrev", "pagan_msa_ready", "pagan_msa_remaining", "pagan_msa_node_cost",
                 "pagan_msa_align_nodes", "pagan_msa_export_result", "pagan_msa_import_result", "pagan_msa_finish",
                 "pagan_msa_data_type", "pagan_msa_node_device", "pagan_msa_set_batch_backend",
                 "pagan_pileup_default_opts", "pagan_pileup_create", "pagan_pileup_align", "pagan_pileup_n_steps",
                 "pagan_pileup_step_info", "pagan_pileup_step_job", "pagan_pileup_step_result",
                 "pagan_pileup_alignment_length", "pagan_pileup_alignment_row", "pagan_pileup_set_batch_backend",
                 "pagan_pileup_destroy"]


def _ip(a):
    return a.ctypes.data_as(_i32p)


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _graph_from_view(v):
    """Copies a borrowed pagan_graph view into an abi.Graph."""
    ns = v.n_sites
    off = np.ctypeslib.as_array(v.bwd_off, shape=(ns + 1,)).copy()
    nb = int(off[-1])
    if nb:
        src = np.ctypeslib.as_array(v.bwd_src, shape=(nb,)).copy()
        lw = np.ctypeslib.as_array(v.bwd_logw, shape=(nb,)).copy()
        eid = np.ctypeslib.as_array(v.bwd_eid, shape=(nb,)).copy()
    else:
        src, lw, eid = np.zeros(0, np.int32), np.zeros(0, np.float32), np.zeros(0, np.int32)
    return abi.Graph(np.ctypeslib.as_array(v.state, shape=(ns,)).copy(), off, src, lw, eid, n_edges=v.n_edges)


class HGraph:
    """Host sequence graph (leaf or parent).  `owned=False` for graphs borrowed from an Msa."""

    def __init__(self, handle, owned=True, keep=None):
        self.h, self.owned, self._keep = handle, owned, keep

    @classmethod
    def leaf(cls, seq, alphabet=DNA_FULL, flags=0):
        return cls(_lib().pagan_hgraph_leaf(seq.encode(), alphabet.encode(), flags))

    @classmethod
    def codon_leaf(cls, nucleotides):
        """One site per triplet (Sequence::create_codon_sequence)."""
        return cls(_lib().pagan_hgraph_leaf_codon(nucleotides.encode()))

    @classmethod
    def parent(cls, left, right, result, lbl, rbl, parsimony, char_as, flags=0):
        """result: abi.Result (columns + used edges) of aligning left and right."""
        r = abi.CResult()
        cols = np.ascontiguousarray(result.cols, np.int32)
        lu = np.ascontiguousarray(result.left_used, np.int32)
        ru = np.ascontiguousarray(result.right_used, np.int32)
        r.n_cols = int(cols.shape[0])
        r.cols = C.cast(_ip(cols), C.POINTER(abi.CCol))
        r.n_left_used, r.left_used = int(lu.shape[0]), _ip(lu)
        r.n_right_used, r.right_used = int(ru.shape[0]), _ip(ru)
        pars = np.ascontiguousarray(parsimony, np.int32)
        S = int(round(pars.size ** 0.5))
        return cls(_lib().pagan_hgraph_parent(left.h, right.h, C.byref(r), lbl, rbl, _ip(pars), S, char_as, flags))

    @classmethod
    def parent_device(cls, left, right, result, lbl, rbl, parsimony, char_as, flags=0):
        """The same parent built on the current HIP device (csrc/dp_parent.hip).  Raises without a device: there is no
        host graph in its place.  The returned graph carries `.build_info` = (runs of skipped sites, rounds of the boundary
        pass, deleted sites, weights outside the log table, log weights corrected by the host)."""
        r = abi.CResult()
        cols = np.ascontiguousarray(result.cols, np.int32)
        lu = np.ascontiguousarray(result.left_used, np.int32)
        ru = np.ascontiguousarray(result.right_used, np.int32)
        r.n_cols = int(cols.shape[0])
        r.cols = C.cast(_ip(cols), C.POINTER(abi.CCol))
        r.n_left_used, r.left_used = int(lu.shape[0]), _ip(lu)
        r.n_right_used, r.right_used = int(ru.shape[0]), _ip(ru)
        pars = np.ascontiguousarray(parsimony, np.int32)
        S = int(round(pars.size ** 0.5))
        info = np.zeros(8, np.int32)
        h = _lib().pagan_hgraph_parent_device(left.h, right.h, C.byref(r), lbl, rbl, _ip(pars), S, char_as, flags, _ip(info))
        if not h:
            raise RuntimeError("pagan_hgraph_parent_device failed: no HIP device, or a HIP error")
        g = cls(h)
        g.build_info = tuple(int(x) for x in info[:5])
        return g

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            _lib().pagan_hgraph_free(self.h)
            self.h = None

    def flatten(self):
        v = abi.CGraph()
        _lib().pagan_hgraph_view(self.h, C.byref(v))
        return _graph_from_view(v)

    def attrs(self):
        v = abi.CGraph()
        L = _lib()
        L.pagan_hgraph_view(self.h, C.byref(v))
        ns, ne = v.n_sites, v.n_edges
        sa = np.zeros((ns, 8), np.int32)
        sd = np.zeros(ns, np.float32)
        ea = np.zeros((max(ne, 1), 6), np.int32)
        ef = np.zeros((max(ne, 1), 3), np.float32)
        L.pagan_hgraph_attrs(self.h, _ip(sa), _fp(sd), _ip(ea), _fp(ef))
        return sa, sd, ea[:ne], ef[:ne]

    def fwd(self):
        v = abi.CGraph()
        L = _lib()
        L.pagan_hgraph_view(self.h, C.byref(v))
        off = np.zeros(v.n_sites + 1, np.int32)
        eid = np.zeros(max(v.n_edges, 1), np.int32)
        L.pagan_hgraph_fwd(self.h, _ip(off), _ip(eid))
        return off, eid[:off[-1]]

    def string(self, with_gaps, alphabet=DNA_FULL):
        v = abi.CGraph()
        L = _lib()
        L.pagan_hgraph_view(self.h, C.byref(v))
        buf = C.create_string_buffer(3 * v.n_sites + 1)              # codon graphs write three characters per site
        n = L.pagan_hgraph_string(self.h, 1 if with_gaps else 0, alphabet.encode(), buf)
        return buf.raw[:n].decode()


def assign_units(costs, n_workers):
    """owner[k] for every unit: largest first, least-loaded worker (pagan_assign_units)."""
    c = np.ascontiguousarray(costs, np.int64)
    owner = np.zeros(c.shape[0], np.int32)
    _lib().pagan_assign_units(int(c.shape[0]), c.ctypes.data_as(C.POINTER(C.c_int64)), int(n_workers), _ip(owner))
    return owner


def define_tunnel(s1, s2, g1, g2, prefix_hit_length=30, hit_trim=5, offset=15):
    up = np.zeros(len(g1) + 1, np.int32)
    lo = np.zeros(len(g1) + 1, np.int32)
    n = _lib().pagan_define_tunnel(s1.encode(), s2.encode(), g1.encode(), g2.encode(), prefix_hit_length, hit_trim,
                                   offset, _ip(up), _ip(lo))
    return abi.Band(up, lo), n


def prefix_hits(s1, s2, min_length=30):
    """Find_anchors::find_long_substrings: [n, 4] int32 rows (start 1, start 2, length, score)."""
    cap = max(len(s1), 1)
    out = np.zeros((cap, 4), np.int32)
    n = _lib().pagan_prefix_hits(s1.encode(), s2.encode(), min_length, _ip(out), cap)
    return out[:n].copy()


def anchors_device_calls():
    """how often the prefix-anchor finder has run on the device in this process"""
    return int(_lib().pagan_anchors_device_calls())


def drop_bad_hits(hits, thr_total=50, thr_partly=400):
    h = np.ascontiguousarray(hits, np.int32).reshape(-1, 4).copy()
    n = _lib().pagan_drop_bad_hits(_ip(h), int(h.shape[0]), thr_total, thr_partly)
    return h[:n].copy()


def define_tunnel_overlapping(hits, g1, g2, width=15):
    """(Band, blocks[n, 4]) from possibly overlapping hits (define_tunnel_with_overlapping_hits)."""
    h = np.ascontiguousarray(hits, np.int32).reshape(-1, 4)
    up = np.zeros(len(g1) + 1, np.int32)
    lo = np.zeros(len(g1) + 1, np.int32)
    cap = len(g1) + 2
    blocks = np.zeros((cap, 4), np.int32)
    n = _lib().pagan_define_tunnel_overlapping(_ip(h), int(h.shape[0]), g1.encode(), g2.encode(), width, _ip(up), _ip(lo),
                                               _ip(blocks), cap)
    if n < 0:
        raise RuntimeError("pagan_define_tunnel_overlapping failed: %d" % n)
    return abi.Band(up, lo), blocks[:n].copy()


def force_gap(band, blocks, threshold=40000, width=15, wide=False):
    """One round of --force-gap: (replaced?, new Band, remaining blocks)."""
    up, lo = band.upper.copy(), band.lower.copy()
    b = np.ascontiguousarray(blocks, np.int32).reshape(-1, 4)
    done = _lib().pagan_force_gap(_ip(up), _ip(lo), int(up.shape[0]), _ip(b), int(b.shape[0]), threshold, width, 1 if wide else 0)
    return bool(done), abi.Band(up, lo), (b[:-1].copy() if done else b.copy())


def dna_model(base_freq, dist):
    """(abi.Model, parsimony[225]) for a DNA alignment at distance `dist`."""
    bf = np.ascontiguousarray(base_freq, np.float32)
    table = np.zeros(225, np.float32)
    params = np.zeros(4, np.float32)
    pars = np.zeros(225, np.int32)
    rc = _lib().pagan_dna_model(_fp(bf), float(dist), _fp(table), _fp(params), _ip(pars))
    if rc != 0:
        raise RuntimeError("pagan_dna_model failed: %d" % rc)
    return abi.Model(table.reshape(15, 15).T, *params), pars


def protein_model(dist):
    """(abi.Model, parsimony[211*211]) for a protein (WAG) alignment at distance `dist`."""
    table = np.zeros(211 * 211, np.float32)
    params = np.zeros(4, np.float32)
    pars = np.zeros(211 * 211, np.int32)
    rc = _lib().pagan_protein_model(float(dist), _fp(table), _fp(params), _ip(pars))
    if rc != 0:
        raise RuntimeError("pagan_protein_model failed: %d" % rc)
    return abi.Model(table.reshape(211, 211).T, *params), pars


CODON_STATES = 61 + 1 + 1830


def codon_model(dist):
    """(abi.Model, parsimony[1892*1892]) for a codon alignment (Kosiol & Goldman's empirical model) at distance `dist`."""
    S = CODON_STATES
    table = np.zeros(S * S, np.float32)
    params = np.zeros(4, np.float32)
    pars = np.zeros(S * S, np.int32)
    rc = _lib().pagan_codon_model(float(dist), _fp(table), _fp(params), _ip(pars))
    if rc != 0:
        raise RuntimeError("pagan_codon_model failed: %d" % rc)
    return abi.Model(table.reshape(S, S).T, *params), pars


def codon_alphabet():
    """(flat string of three-letter state names -- the first 62 are the leaf codons and NNN --, mostcommon[61*61])"""
    buf = C.create_string_buffer(3 * CODON_STATES + 1)
    mc = np.zeros(61 * 61, np.int32)
    _lib().pagan_codon_alphabet(buf, _ip(mc))
    return buf.value.decode(), mc


def codon_translate(codon_string):
    """One amino-acid letter per triplet (Codon_translation::gapped_DNA_to_protein)."""
    buf = C.create_string_buffer(len(codon_string) // 3 + 2)
    n = _lib().pagan_codon_translate(codon_string.encode(), buf)
    if n < 0:
        raise RuntimeError("pagan_codon_translate failed: %d" % n)
    return buf.raw[:n].decode()


def codon_states(nucleotides):
    out = np.zeros(len(nucleotides) // 3 + 2, np.int32)
    n = _lib().pagan_codon_states(nucleotides.encode(), _ip(out))
    if n < 0:
        raise RuntimeError("pagan_codon_states failed: %d" % n)
    return out[:n].copy()


def model_prob(data_type, dist, base_freq=None):
    """abi.ModelProb for a distance: the probability-space view (Evol_model::score, gap_open, gap_ext, non_gap)."""
    S = CODON_STATES if data_type == 3 else 211 if data_type == 2 else 15
    score = np.zeros(S * S, np.float32)
    params = np.zeros(3, np.float32)
    bf = np.ascontiguousarray(base_freq if base_freq is not None else [0.25] * 4, np.float32)
    rc = _lib().pagan_model_prob_table(int(data_type), _fp(bf), float(dist), _fp(score), _fp(params))
    if rc != 0:
        raise RuntimeError("pagan_model_prob_table failed: %d" % rc)
    return abi.ModelProb(score.reshape(S, S).T, *params)


def alphabets(data_type):
    """(leaf alphabet, ancestral alphabet) of a data type: 1 DNA, 2 protein."""
    a, b = C.create_string_buffer(256), C.create_string_buffer(256)
    _lib().pagan_model_alphabets(int(data_type), a, b)
    return a.value.decode(), b.value.decode()


def eigen_qrev(Q, pi):
    """Eigen::eigenQREV as the library restates it: (root, U, V) with Q = U diag(root) V."""
    Q = np.ascontiguousarray(Q, np.float64)
    pi = np.ascontiguousarray(pi, np.float64)
    n = pi.shape[0]
    root, U, V = np.zeros(n), np.zeros((n, n)), np.zeros((n, n))
    dp = C.POINTER(C.c_double)
    rc = _lib().pagan_eigen_qrev(Q.ctypes.data_as(dp), pi.ctypes.data_as(dp), n, root.ctypes.data_as(dp),
                                 U.ctypes.data_as(dp), V.ctypes.data_as(dp))
    if rc != 0:
        raise RuntimeError("pagan_eigen_qrev failed: %d" % rc)
    return root, U, V


def _job_copy(j):
    """(Graph, Graph, Model, Band|None) copies of a borrowed CJob."""
    left, right = _graph_from_view(j.left.contents), _graph_from_view(j.right.contents)
    m = j.model.contents
    S = m.n_states
    table = np.ctypeslib.as_array(m.log_score, shape=(S * S,)).copy()
    model = abi.Model(table.reshape(S, S).T, m.log_gap_open, m.log_gap_ext, m.log_gap_end_ext, m.log_non_gap)
    band = None
    if j.band:
        b = j.band.contents
        band = abi.Band(np.ctypeslib.as_array(b.upper, shape=(b.n,)).copy(), np.ctypeslib.as_array(b.lower, shape=(b.n,)).copy())
    return left, right, model, band


class Pileup:
    """Reads_aligner::pileup_alignment mirror: read 0 is the reference, every further read is aligned (GPU) against the
    growing root and joins it when it overlaps well enough."""

    def __init__(self, names, seqs, **opts):
        L = _lib()
        o = CPileupOpts()
        L.pagan_pileup_default_opts(C.byref(o))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError("unknown option %s" % k)
            setattr(o, k, v)
        self.n = len(names)
        na = (C.c_char_p * self.n)(*[s.encode() for s in names])
        sa = (C.c_char_p * self.n)(*[s.encode() for s in seqs])
        self._h = C.c_void_p()
        rc = L.pagan_pileup_create(self.n, na, sa, C.byref(o), C.byref(self._h))
        if rc != 0:
            from . import PaganError
            raise PaganError(rc, "pagan_pileup_create")
        self._L = L

    def set_batch_backend(self, fn):
        """TEST SEAM: `fn` stands in for pagan_dp_align_batch (see Msa.set_batch_backend)."""
        self._backend = BATCH_FN(fn) if fn is not None else C.cast(None, BATCH_FN)
        self._L.pagan_pileup_set_batch_backend(self._h, self._backend, None)

    def align(self):
        rc = self._L.pagan_pileup_align(self._h)
        if rc != 0:
            from . import PaganError
            raise PaganError(rc, "pagan_pileup_align")
        return self

    @property
    def n_steps(self):
        return self._L.pagan_pileup_n_steps(self._h)

    def step(self, k):
        s = CPileupStep()
        self._L.pagan_pileup_step_info(self._h, k, C.byref(s))
        return s

    def step_job(self, k):
        j = abi.CJob()
        rc = self._L.pagan_pileup_step_job(self._h, k, C.byref(j))
        if rc != 0:
            raise RuntimeError("pagan_pileup_step_job failed: %d" % rc)
        return _job_copy(j)

    def step_result(self, k):
        r = abi.CResult()
        rc = self._L.pagan_pileup_step_result(self._h, k, C.byref(r))
        if rc != 0:
            raise RuntimeError("pagan_pileup_step_result failed: %d" % rc)
        return abi.Result(r)

    def alignment(self):
        n = self._L.pagan_pileup_alignment_length(self._h)
        buf = C.create_string_buffer(n + 1)
        rows = []
        for k in range(self.n):
            ln = self._L.pagan_pileup_alignment_row(self._h, k, buf)
            rows.append(buf.raw[:ln].decode())
        return rows

    def close(self):
        if self._h:
            self._L.pagan_pileup_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Msa:
    """Progressive alignment of sequences on a rooted binary guide tree (Node mirror)."""

    def __init__(self, names, seqs, newick, **opts):
        L = _lib()
        o = CMsaOpts()
        L.pagan_msa_default_opts(C.byref(o))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError("unknown option %s" % k)
            setattr(o, k, v)
        self.n = len(names)
        na = (C.c_char_p * self.n)(*[s.encode() for s in names])
        sa = (C.c_char_p * self.n)(*[s.encode() for s in seqs])
        self._h = C.c_void_p()
        rc = L.pagan_msa_create(self.n, na, sa, newick.encode(), C.byref(o), C.byref(self._h))
        if rc != 0:
            from . import PaganError
            raise PaganError(rc, "pagan_msa_create")
        self._L = L

    def align(self):
        rc = self._L.pagan_msa_align(self._h)
        if rc != 0:
            from . import PaganError
            raise PaganError(rc, "pagan_msa_align")
        return self

    @property
    def n_internal(self):
        return self._L.pagan_msa_n_internal(self._h)

    @property
    def data_type(self):
        return self._L.pagan_msa_data_type(self._h)

    # ---- the walk one round at a time (one process per GPU; see dist.align_sharded) ----
    def ready(self):
        ids = np.zeros(max(self.n, 1), np.int32)
        cnt = self._L.pagan_msa_ready(self._h, _ip(ids), int(ids.shape[0]))
        return [int(x) for x in ids[:cnt]]

    @property
    def remaining(self):
        return self._L.pagan_msa_remaining(self._h)

    def node_cost(self, node):
        return int(self._L.pagan_msa_node_cost(self._h, node))

    def align_nodes(self, nodes):
        ids = np.ascontiguousarray(nodes, np.int32)
        rc = self._L.pagan_msa_align_nodes(self._h, int(ids.shape[0]), _ip(ids))
        if rc != 0:
            from . import PaganError
            raise PaganError(rc, "pagan_msa_align_nodes")

    def export_result(self, node):
        need = self._L.pagan_msa_export_result(self._h, node, None, 0)
        if need < 0:
            from . import PaganError
            raise PaganError(int(need), "pagan_msa_export_result")
        buf = np.zeros(need, np.uint8)
        self._L.pagan_msa_export_result(self._h, node, buf.ctypes.data_as(C.c_void_p), need)
        return buf

    def import_result(self, buf):
        b = np.ascontiguousarray(buf, np.uint8)
        rc = self._L.pagan_msa_import_result(self._h, b.ctypes.data_as(C.c_void_p), int(b.shape[0]))
        if rc != 0:
            from . import PaganError
            raise PaganError(rc, "pagan_msa_import_result")

    def node_device(self, k):
        return self._L.pagan_msa_node_device(self._h, k)

    def set_batch_backend(self, fn):
        """TEST SEAM: `fn(n, jobs, opts, out, user) -> rc` stands in for pagan_dp_align_batch (None: the HIP path)."""
        self._backend = BATCH_FN(fn) if fn is not None else C.cast(None, BATCH_FN)
        self._L.pagan_msa_set_batch_backend(self._h, self._backend, None)

    def finish(self, lazy=False):
        """After the last round.  lazy: the parent graphs of imported nodes that nobody needed, and the rows, are built by
        the first call that asks for them (alignment(), write_fasta(), node_graph()) instead of now."""
        rc = (self._L.pagan_msa_finish_lazy if lazy else self._L.pagan_msa_finish)(self._h)
        if rc != 0:
            from . import PaganError
            raise PaganError(rc, "pagan_msa_finish")
        return self

    @property
    def parents_built(self):
        """parent graphs this process has built so far (rank mode: the nodes it aligned and the imported ones it needed)"""
        return self._L.pagan_msa_parents_built(self._h)

    def node_info(self, k):
        info = CNodeInfo()
        self._L.pagan_msa_node_info(self._h, k, C.byref(info))
        return info

    def node_cjob(self, k):
        """Borrowed CJob (pointers into the Msa; valid while it lives)."""
        j = abi.CJob()
        rc = self._L.pagan_msa_node_job(self._h, k, C.byref(j))
        if rc != 0:
            raise RuntimeError("pagan_msa_node_job failed: %d" % rc)
        return j

    def node_job(self, k):
        """(Graph, Graph, Model, Band|None) copies of node k's aligner inputs."""
        j = self.node_cjob(k)
        left, right = _graph_from_view(j.left.contents), _graph_from_view(j.right.contents)
        m = j.model.contents
        S = m.n_states
        table = np.ctypeslib.as_array(m.log_score, shape=(S * S,)).copy()
        model = abi.Model(table.reshape(S, S).T, m.log_gap_open, m.log_gap_ext, m.log_gap_end_ext, m.log_non_gap)
        band = None
        if j.band:
            b = j.band.contents
            band = abi.Band(np.ctypeslib.as_array(b.upper, shape=(b.n,)).copy(),
                            np.ctypeslib.as_array(b.lower, shape=(b.n,)).copy())
        return left, right, model, band

    def node_result(self, k):
        r = abi.CResult()
        rc = self._L.pagan_msa_node_result(self._h, k, C.byref(r))
        if rc != 0:
            raise RuntimeError("pagan_msa_node_result failed: %d" % rc)
        return abi.Result(r)

    def node_graph(self, node):
        return HGraph(self._L.pagan_msa_node_graph(self._h, node), owned=False, keep=self)

    def timing(self):
        t = CTiming()
        self._L.pagan_msa_timing_get(self._h, C.byref(t))
        return {k: getattr(t, k) for k, _ in CTiming._fields_}

    def alignment(self):
        n = self._L.pagan_msa_alignment_length(self._h)
        rows = []
        buf = C.create_string_buffer(n + 1)
        for k in range(self.n):
            self._L.pagan_msa_alignment_row(self._h, k, buf)
            rows.append(buf.raw[:n].decode())
        return rows

    def alignment_all(self):
        """Rows of every node: leaves 0..n-1, then the internal nodes (ancestors) in alignment order."""
        n = self._L.pagan_msa_alignment_length(self._h)
        rows = []
        buf = C.create_string_buffer(n + 1)
        for k in range(2 * self.n - 1):
            self._L.pagan_msa_alignment_row(self._h, k, buf)
            rows.append(buf.raw[:n].decode())
        return rows

    def write_fasta(self, path, chars_by_line=60, include_internal=False):
        """The rows as FASTA in guide-tree order (Fasta_reader::write_fasta over Node::get_alignment); with
        include_internal the ancestors' rows too, in Node::get_all_nodes order."""
        rc = self._L.pagan_msa_write_fasta_nodes(self._h, str(path).encode(), chars_by_line, 1 if include_internal else 0)
        if rc != 0:
            from . import PaganError
            raise PaganError(rc, "pagan_msa_write_fasta")

    def close(self):
        if self._h:
            self._L.pagan_msa_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
