"""ctypes mirror of include/pagan_dp.h and numpy-backed holders for its structs.

Plumbing only: every compute entry point lives in libpagan_dp.so (HIP).  The classes
here keep the numpy arrays alive for as long as the C structs that point into them.
"""
import ctypes as C

import numpy as np

# ---- constants (include/pagan_dp.h) ----------------------------------------------------
PAGAN_OK = 0
PAGAN_E_ARG, PAGAN_E_GRAPH, PAGAN_E_BAND, PAGAN_E_MODEL = -1, -2, -3, -4
PAGAN_E_NODEVICE, PAGAN_E_NOMEM, PAGAN_E_INTERNAL = -5, -6, -7
PAGAN_DP_REACHED, PAGAN_DP_UNREACHABLE = 0, 1
OPT_NO_TERMINAL_EDGES = 1
OPT_NO_REDUCED_TERMINAL_PEN = 2
X_MAT, Y_MAT, M_MAT = 0, 1, 2
MATCHED, XGAPPED, YGAPPED, XSKIPPED, YSKIPPED = 2, 3, 4, 5, 6

_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


class CGraph(C.Structure):
    _fields_ = [("n_sites", C.c_int32), ("n_edges", C.c_int32), ("state", _i32p), ("bwd_off", _i32p),
                ("bwd_src", _i32p), ("bwd_logw", _f32p), ("bwd_eid", _i32p)]


class CModel(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("log_score", _f32p), ("log_gap_open", C.c_float),
                ("log_gap_ext", C.c_float), ("log_gap_end_ext", C.c_float), ("log_non_gap", C.c_float)]


class CBand(C.Structure):
    _fields_ = [("n", C.c_int32), ("upper", _i32p), ("lower", _i32p)]


class COpts(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("device", C.c_int32)]


class CCol(C.Structure):
    _fields_ = [("left", C.c_int32), ("right", C.c_int32), ("path_state", C.c_int32)]


class CResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("score", C.c_double), ("end_matrix", C.c_int32),
                ("end_x", C.c_int32), ("end_y", C.c_int32), ("end_x_edge", C.c_int32), ("end_y_edge", C.c_int32),
                ("n_cols", C.c_int32), ("cols", C.POINTER(CCol)),
                ("n_left_used", C.c_int32), ("left_used", _i32p),
                ("n_right_used", C.c_int32), ("right_used", _i32p),
                ("cells", C.c_int64), ("fill_ms", C.c_double), ("trace_ms", C.c_double)]


class CJob(C.Structure):
    _fields_ = [("left", C.POINTER(CGraph)), ("right", C.POINTER(CGraph)), ("model", C.POINTER(CModel)),
                ("band", C.POINTER(CBand))]


def _p(arr, typ):
    return arr.ctypes.data_as(typ)


class Graph:
    """One child Sequence flattened to CSR (pagan_graph)."""

    def __init__(self, state, bwd_off, bwd_src, bwd_logw, bwd_eid, n_edges=None):
        self.state = np.ascontiguousarray(state, dtype=np.int32)
        self.bwd_off = np.ascontiguousarray(bwd_off, dtype=np.int32)
        self.bwd_src = np.ascontiguousarray(bwd_src, dtype=np.int32)
        self.bwd_logw = np.ascontiguousarray(bwd_logw, dtype=np.float32)
        self.bwd_eid = np.ascontiguousarray(bwd_eid, dtype=np.int32)
        self.n_sites = int(self.state.shape[0])
        if n_edges is None:
            n_edges = int(self.bwd_eid.max()) + 1 if self.bwd_eid.size else 0
        self.n_edges = int(n_edges)
        assert self.bwd_off.shape[0] == self.n_sites + 1
        self.c = CGraph(self.n_sites, self.n_edges, _p(self.state, _i32p), _p(self.bwd_off, _i32p),
                        _p(self.bwd_src, _i32p), _p(self.bwd_logw, _f32p), _p(self.bwd_eid, _i32p))

    @classmethod
    def chain(cls, states):
        """Plain leaf: start, one site per residue, stop; edge k joins site k-1 -> k
        (edge 0 is the reference's unlinked dummy edge, sequence.cpp:164-165)."""
        states = np.asarray(states, dtype=np.int32)
        n = states.shape[0] + 2
        st = np.full(n, -1, np.int32)
        st[1:-1] = states
        off = np.concatenate([[0], np.arange(0, n, dtype=np.int32)]).astype(np.int32)
        src = np.arange(0, n - 1, dtype=np.int32)
        return cls(st, off, src, np.zeros(n - 1, np.float32), np.arange(1, n, dtype=np.int32), n_edges=n)


class Model:
    def __init__(self, log_score, log_gap_open, log_gap_ext, log_gap_end_ext, log_non_gap):
        t = np.asarray(log_score, dtype=np.float32)
        assert t.ndim == 2 and t.shape[0] == t.shape[1]
        self.n_states = int(t.shape[0])
        # log_score(a,b) = table[a + b*S]  (Db_matrix::g is column-major, db_matrix.h:76-83)
        self.table = np.ascontiguousarray(t.T.reshape(-1))
        self.log_score = t
        self.params = tuple(np.float32(x) for x in (log_gap_open, log_gap_ext, log_gap_end_ext, log_non_gap))
        self.c = CModel(self.n_states, _p(self.table, _f32p), *[float(x) for x in self.params])


class CModelProb(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("score", _f32p), ("gap_open", C.c_float), ("gap_ext", C.c_float),
                ("non_gap", C.c_float)]


class ModelProb:
    """pagan_model_prob: the model in probability space (forward/backward pass)."""

    def __init__(self, score, gap_open, gap_ext, non_gap):
        t = np.asarray(score, dtype=np.float32)
        assert t.ndim == 2 and t.shape[0] == t.shape[1]
        self.n_states = int(t.shape[0])
        self.table = np.ascontiguousarray(t.T.reshape(-1))       # score(a,b) = table[a + b*S]
        self.score = t
        self.gap_open, self.gap_ext, self.non_gap = (float(np.float32(x)) for x in (gap_open, gap_ext, non_gap))
        self.c = CModelProb(self.n_states, _p(self.table, _f32p), self.gap_open, self.gap_ext, self.non_gap)


class Band:
    def __init__(self, upper, lower):
        self.upper = np.ascontiguousarray(upper, dtype=np.int32)
        self.lower = np.ascontiguousarray(lower, dtype=np.int32)
        assert self.upper.shape == self.lower.shape
        self.c = CBand(int(self.upper.shape[0]), _p(self.upper, _i32p), _p(self.lower, _i32p))


class Result:
    """Python copy of a pagan_result (the C arrays are freed by the producer's free())."""

    def __init__(self, r):
        self.status = r.status
        self.score = r.score
        self.end = (r.end_matrix, r.end_x, r.end_y, r.end_x_edge, r.end_y_edge)
        n = r.n_cols
        if n > 0:
            buf = np.ctypeslib.as_array(C.cast(r.cols, _i32p), shape=(n, 3)).copy()
        else:
            buf = np.zeros((0, 3), np.int32)
        self.cols = buf
        self.left_used = (np.ctypeslib.as_array(r.left_used, shape=(r.n_left_used,)).copy()
                          if r.n_left_used > 0 else np.zeros(0, np.int32))
        self.right_used = (np.ctypeslib.as_array(r.right_used, shape=(r.n_right_used,)).copy()
                           if r.n_right_used > 0 else np.zeros(0, np.int32))
        self.cells = r.cells
        self.fill_ms = r.fill_ms
        self.trace_ms = r.trace_ms

    def same_alignment(self, other):
        """Bit-exact comparison of everything the parent-graph builder consumes."""
        return (self.status == other.status and
                np.float64(self.score).tobytes() == np.float64(other.score).tobytes() and
                self.end == other.end and np.array_equal(self.cols, other.cols) and
                np.array_equal(self.left_used, other.left_used) and
                np.array_equal(self.right_used, other.right_used))


def declare(lib):
    """Attach argtypes/restype for every symbol include/pagan_dp.h declares."""
    gp, mp, bp, op, rp = (C.POINTER(CGraph), C.POINTER(CModel), C.POINTER(CBand), C.POINTER(COpts),
                          C.POINTER(CResult))
    lib.pagan_dp_align.argtypes = [gp, gp, mp, bp, op, rp]
    lib.pagan_dp_align.restype = C.c_int
    lib.pagan_dp_align_batch.argtypes = [C.c_int32, C.POINTER(CJob), op, rp]
    lib.pagan_dp_align_batch.restype = C.c_int
    lib.pagan_result_free.argtypes = [rp]
    lib.pagan_result_free.restype = None
    lib.pagan_dp_predict_bytes.argtypes = [C.c_int32, C.c_int32, bp]
    lib.pagan_dp_predict_bytes.restype = C.c_int64
    lib.pagan_dp_count_cells.argtypes = [C.c_int32, C.c_int32, bp]
    lib.pagan_dp_count_cells.restype = C.c_int64
    lib.pagan_dp_device_count.argtypes = []
    lib.pagan_dp_device_count.restype = C.c_int
    lib.pagan_dp_select_device.argtypes = [C.c_int32]
    lib.pagan_dp_select_device.restype = C.c_int
    lib.pagan_batch_create.argtypes = [C.c_int32, C.POINTER(CJob), op, C.POINTER(C.c_void_p)]
    lib.pagan_batch_create.restype = C.c_int
    lib.pagan_batch_run.argtypes = [C.c_void_p]
    lib.pagan_batch_run.restype = C.c_int
    lib.pagan_batch_sync.argtypes = [C.c_void_p]
    lib.pagan_batch_sync.restype = C.c_int
    lib.pagan_batch_fetch.argtypes = [C.c_void_p, rp]
    lib.pagan_batch_fetch.restype = C.c_int
    lib.pagan_batch_last_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.pagan_batch_last_ms.restype = C.c_int
    lib.pagan_batch_last_ms_detail.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.pagan_batch_last_ms_detail.restype = C.c_int
    lib.pagan_batch_cells.argtypes = [C.c_void_p]
    lib.pagan_batch_cells.restype = C.c_int64
    lib.pagan_batch_destroy.argtypes = [C.c_void_p]
    lib.pagan_batch_destroy.restype = None
    lib.pagan_batch_debug_trace.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    lib.pagan_batch_debug_trace.restype = C.c_int
    lib.pagan_dp_debug_plan.argtypes = [gp, gp, bp, C.POINTER(C.c_uint8), C.c_int32, _i32p, C.c_int32, _i32p, _i32p]
    lib.pagan_dp_debug_plan.restype = C.c_int
    lib.pagan_dp_debug_far.argtypes = [gp, gp, bp] + [C.POINTER(C.c_uint8)] * 4
    lib.pagan_dp_debug_far.restype = C.c_int
    lib.pagan_dp_debug_strips.argtypes = [gp, gp, bp, C.c_int32, _i32p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64]
    lib.pagan_dp_debug_strips.restype = C.c_int
    lib.pagan_dp_debug_tiles.argtypes = [gp, gp, bp, _i32p, C.c_int32, _i32p]
    lib.pagan_dp_debug_tiles.restype = C.c_int
    lib.pagan_dp_debug_compact.argtypes = [gp, gp, bp, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p]
    lib.pagan_dp_debug_compact.restype = C.c_int
    lib.pagan_dp_debug_tiles_staircase.argtypes = [_i32p, C.c_int32]
    lib.pagan_dp_debug_tiles_staircase.restype = C.c_int
    lib.pagan_dp_debug_route.argtypes = [gp, gp, C.POINTER(CModel), bp, _i32p]
    lib.pagan_dp_debug_route.restype = C.c_int
    lib.pagan_batch_debug_scores.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.c_int64]
    lib.pagan_batch_debug_scores.restype = C.c_int
    lib.pagan_batch_debug_backptrs.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint32), C.c_int64]
    lib.pagan_batch_debug_backptrs.restype = C.c_int
    lib.pagan_batch_debug_followed.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.pagan_batch_debug_followed.restype = C.c_int
    lib.pagan_batch_debug_poke_bp.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32]
    lib.pagan_batch_debug_poke_bp.restype = C.c_int
    lib.pagan_batch_debug_reruns.argtypes = [C.c_void_p]
    lib.pagan_batch_debug_reruns.restype = C.c_int
    lib.pagan_batch_debug_poison.argtypes = [C.c_void_p]
    lib.pagan_batch_debug_poison.restype = C.c_int
    lib.pagan_dp_release_cache.argtypes = []
    lib.pagan_dp_release_cache.restype = None
    lib.pagan_dp_cached_device_bytes.argtypes = [C.c_int32]
    lib.pagan_dp_cached_device_bytes.restype = C.c_int64
    mpp, f64p = C.POINTER(CModelProb), C.POINTER(C.c_double)
    lib.pagan_fb_run.argtypes = [gp, gp, mpp, bp, op, C.POINTER(C.c_void_p)]
    lib.pagan_fb_run.restype = C.c_int
    lib.pagan_fb_run_batch.argtypes = [C.c_int32, C.POINTER(gp), C.POINTER(gp), C.POINTER(mpp), C.POINTER(bp), op, C.POINTER(C.c_void_p)]
    lib.pagan_fb_run_batch.restype = C.c_int
    lib.pagan_fb_totals.argtypes = [C.c_void_p, f64p, f64p, C.POINTER(C.c_int64)]
    lib.pagan_fb_totals.restype = C.c_int
    lib.pagan_fb_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.pagan_fb_kernel_ms.restype = C.c_int
    lib.pagan_fb_groups.argtypes = [C.c_void_p]
    lib.pagan_fb_groups.restype = C.c_int
    lib.pagan_fb_dump.argtypes = [C.c_void_p, C.c_int32, f64p]
    lib.pagan_fb_dump.restype = C.c_int
    lib.pagan_fb_posterior_cells.argtypes = [C.c_void_p, C.c_int32, _i32p, f64p]
    lib.pagan_fb_posterior_cells.restype = C.c_int
    lib.pagan_fb_sample_path.argtypes = [C.c_void_p, f64p, C.c_int32, rp, _i32p, _i32p]
    lib.pagan_fb_sample_path.restype = C.c_int
    lib.pagan_fb_destroy.argtypes = [C.c_void_p]
    lib.pagan_fb_destroy.restype = None
    lib.pagan_dp_version.argtypes = []
    lib.pagan_dp_version.restype = C.c_char_p
    return lib


EXPORTED = ["pagan_dp_align", "pagan_dp_align_batch", "pagan_result_free", "pagan_dp_predict_bytes",
            "pagan_dp_count_cells", "pagan_dp_device_count", "pagan_dp_select_device", "pagan_batch_create",
            "pagan_batch_run", "pagan_batch_sync", "pagan_batch_fetch", "pagan_batch_last_ms",
            "pagan_batch_cells", "pagan_batch_last_ms_detail", "pagan_batch_destroy", "pagan_batch_debug_trace", "pagan_dp_debug_plan", "pagan_dp_debug_far", "pagan_dp_debug_strips", "pagan_dp_debug_tiles", "pagan_dp_debug_compact", "pagan_dp_debug_tiles_staircase", "pagan_dp_release_cache", "pagan_dp_cached_device_bytes", "pagan_dp_debug_route", "pagan_batch_debug_scores", "pagan_batch_debug_backptrs", "pagan_batch_debug_poison", "pagan_batch_debug_followed", "pagan_batch_debug_poke_bp", "pagan_batch_debug_reruns",
            "pagan_fb_run", "pagan_fb_run_batch", "pagan_fb_totals", "pagan_fb_kernel_ms", "pagan_fb_groups", "pagan_fb_dump", "pagan_fb_posterior_cells", "pagan_fb_sample_path",
            "pagan_fb_destroy", "pagan_dp_version"]
