"""pagan2-msa_amd: MI355X-native pairwise graph-vs-graph Viterbi aligner (PAGAN2's
Viterbi_alignment hot path) behind the C ABI of include/pagan_dp.h.

Python here is plumbing: it loads libpagan_dp.so (HIP kernels + C ABI, built in-tree by
build.py) and mirrors the reference's call surface for this path:

    align(left, right, model, band=None, flags=0)      <- Viterbi_alignment::align
    align_batch(jobs, flags=0)                          (ready nodes of one tree level)
    Batch(jobs).run() / .fetch()                        (inputs resident in HBM)

There is no CPU fallback: if the library or a HIP device is missing the calls raise.
"""
import ctypes as C
import os

from . import abi
from .abi import Band, Graph, Model, ModelProb, Result  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# PAGAN_DP_LIB: a diagnostic build (tools/build_stamps.sh writes libpagan_dp_stats.so) instead of the product library
LIB_PATH = os.environ.get("PAGAN_DP_LIB") or os.path.join(_HERE, "libpagan_dp.so")
_lib = None


class PaganError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with code %d" % (what, code))
        self.code = code


def lib():
    """The loaded C-ABI library.  Raises if it has not been built (see build.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libpagan_dp.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = abi.declare(C.CDLL(LIB_PATH))
    return _lib


def device_count():
    return lib().pagan_dp_device_count()


def _check(code, what):
    if code != abi.PAGAN_OK:
        raise PaganError(code, what)


def _jobs_array(jobs):
    arr = (abi.CJob * len(jobs))()
    for k, (left, right, model, band) in enumerate(jobs):
        arr[k].left = C.pointer(left.c)
        arr[k].right = C.pointer(right.c)
        arr[k].model = C.pointer(model.c)
        arr[k].band = C.pointer(band.c) if band is not None else None
    return arr


def align_batch(jobs, flags=0, device=-1):
    """jobs: list of (Graph left, Graph right, Model, Band|None).  Returns [Result]."""
    L = lib()
    arr = _jobs_array(jobs)
    opts = abi.COpts(flags, device)
    res = (abi.CResult * len(jobs))()
    rc = L.pagan_dp_align_batch(len(jobs), arr, C.byref(opts), res)
    try:
        _check(rc, "pagan_dp_align_batch")
        return [Result(r) for r in res]
    finally:
        for r in res:
            L.pagan_result_free(C.byref(r))


def align(left, right, model, band=None, flags=0, device=-1):
    """Mirror of Viterbi_alignment::align (+ define_tunnel's band as an argument)."""
    L = lib()
    opts = abi.COpts(flags, device)
    res = abi.CResult()
    rc = L.pagan_dp_align(C.byref(left.c), C.byref(right.c), C.byref(model.c),
                          C.byref(band.c) if band is not None else None, C.byref(opts), C.byref(res))
    try:
        _check(rc, "pagan_dp_align")
        return Result(res)
    finally:
        L.pagan_result_free(C.byref(res))


class FullProbability:
    """Forward/backward matrices of one alignment on the GPU (the reference's compute_full_score pass:
    --full-probability, posteriors, --sample-path), in log space.

        fb = FullProbability(left, right, model_prob, band)
        fb.log_fwd, fb.log_bwd          log max_end.fwd_score, log match[0][0].bwd_score
        fb.posterior()                  [Lx, Ly, 3] (X, Y, M), compute_posterior_score
        fb.sample_path(u)               Result with the shape of a Viterbi result
    """

    def __init__(self, left, right, model_prob, band=None, device=-1, _handle=None):
        import numpy as np
        self._np = np
        self._L = lib()
        self.left, self.right, self.model, self.band = left, right, model_prob, band     # keep the arrays alive
        if _handle is not None:                          # (full_probability_batch: the pair ran in a batch's launches)
            self._h = _handle
        else:
            opts = abi.COpts(0, device)
            self._h = C.c_void_p()
            _check(self._L.pagan_fb_run(C.byref(left.c), C.byref(right.c), C.byref(model_prob.c),
                                        C.byref(band.c) if band is not None else None, C.byref(opts), C.byref(self._h)),
                   "pagan_fb_run")
        a, b, c = C.c_double(), C.c_double(), C.c_int64()
        _check(self._L.pagan_fb_totals(self._h, C.byref(a), C.byref(b), C.byref(c)), "pagan_fb_totals")
        self.log_fwd, self.log_bwd, self.cells = a.value, b.value, c.value
        kms = (C.c_double * 2)()
        _check(self._L.pagan_fb_kernel_ms(self._h, kms), "pagan_fb_kernel_ms")
        self.forward_ms, self.backward_ms = kms[0], kms[1]
        self.groups = self._L.pagan_fb_groups(self._h)       # 1: one-workgroup sweeps; > 1: 64 x 64 blocks, a wave each; 0: LDS-ring sweeps
        self.shape = (left.n_sites - 1, right.n_sites - 1, 3)

    def _dump(self, which):
        out = self._np.zeros(self.shape, self._np.float64)
        _check(self._L.pagan_fb_dump(self._h, which, out.ctypes.data_as(C.POINTER(C.c_double))), "pagan_fb_dump")
        return out

    def log_forward(self):
        return self._dump(0)

    def log_backward(self):
        return self._dump(1)

    def posterior(self):
        return self._dump(2)

    def posterior_cells(self, cells):
        c = self._np.ascontiguousarray(cells, self._np.int32).reshape(-1, 3)
        out = self._np.zeros(c.shape[0], self._np.float64)
        _check(self._L.pagan_fb_posterior_cells(self._h, c.shape[0], c.ctypes.data_as(C.POINTER(C.c_int32)),
                                                out.ctypes.data_as(C.POINTER(C.c_double))), "pagan_fb_posterior_cells")
        return out

    def sample_path(self, u):
        """(Result, visited cells end -> start as rows (i, j, state))."""
        uu = self._np.ascontiguousarray(u, self._np.float64)
        res = abi.CResult()
        vis = self._np.zeros((self.shape[0] + self.shape[1] + 2, 3), self._np.int32)
        n = C.c_int32()
        rc = self._L.pagan_fb_sample_path(self._h, uu.ctypes.data_as(C.POINTER(C.c_double)), int(uu.shape[0]), C.byref(res),
                                          vis.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n))
        try:
            _check(rc, "pagan_fb_sample_path")
            return Result(res), vis[:n.value].copy()
        finally:
            self._L.pagan_result_free(C.byref(res))

    def close(self):
        if self._h:
            self._L.pagan_fb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def debug_tiles(left, right, band=None):
    """Diagnostic (host only): (tile side, [(tile row, tile column), ...]) the wide-matrix fill kernel would be
    launched over for this job; an empty list when the job cannot be tiled."""
    import numpy as np
    L = lib()
    cap = ((left.n_sites + 62) // 64 + 1) * ((right.n_sites + 62) // 64 + 1)
    out = np.zeros(2 * cap, np.int32)
    side = C.c_int32()
    n = L.pagan_dp_debug_tiles(C.byref(left.c), C.byref(right.c), C.byref(band.c) if band is not None else None,
                               out.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(side))
    _check(min(n, 0), "pagan_dp_debug_tiles")
    return side.value, [tuple(int(v) for v in out[2 * k: 2 * k + 2]) for k in range(min(n, cap))]


def debug_tiles_staircase(tiles):
    """Diagnostic (host only): whether a list of (tile row, tile column) pairs is a staircase (include/pagan_dp.h)."""
    import numpy as np
    a = np.ascontiguousarray(np.array(tiles, np.int32).reshape(-1, 2))
    rc = lib().pagan_dp_debug_tiles_staircase(a.ctypes.data_as(C.POINTER(C.c_int32)), len(a))
    _check(min(rc, 0), "pagan_dp_debug_tiles_staircase")
    return bool(rc)


def debug_compact(left, right, band=None):
    """Diagnostic (host only): the compacted numbering the library aligns graphs with many dead sites in:
    dict with keep_left / keep_right (compacted site -> caller's site), slot_left / slot_right (per kept bwd edge: its position
    in the caller's list of its site), upper / lower (the band over the compacted matrices)."""
    import numpy as np
    L = lib()
    p32 = C.POINTER(C.c_int32)
    kl = np.zeros(left.n_sites, np.int32); kr = np.zeros(right.n_sites, np.int32)
    sl = np.zeros(max(int(left.bwd_off[-1]), 1), np.int32); sr = np.zeros(max(int(right.bwd_off[-1]), 1), np.int32)
    up = np.zeros(left.n_sites, np.int32); lo = np.zeros(left.n_sites, np.int32)
    n = np.zeros(4, np.int32)
    _check(L.pagan_dp_debug_compact(C.byref(left.c), C.byref(right.c), C.byref(band.c) if band is not None else None,
                                    kl.ctypes.data_as(p32), kr.ctypes.data_as(p32), sl.ctypes.data_as(p32), sr.ctypes.data_as(p32),
                                    up.ctypes.data_as(p32), lo.ctypes.data_as(p32), n.ctypes.data_as(p32)), "pagan_dp_debug_compact")
    return {"keep_left": kl[:n[0]].copy(), "keep_right": kr[:n[1]].copy(), "slot_left": sl[:n[2]].copy(), "slot_right": sr[:n[3]].copy(),
            "upper": up[:n[0] - 1].copy(), "lower": lo[:n[0] - 1].copy()}


ROUTES = ("pg_fill_pipe", "pg_fill_pipe (large table)", "pg_fill_tiles_flow", "pg_fill_wavefront", "pg_fill_pipe (row strips)")


def debug_route(left, right, model, band=None):
    """Diagnostic (host only): (fill kernel pagan_batch_create would give this job, dead sites taken out first?, cells of
    the widest anti-diagonal)."""
    import numpy as np
    n = np.zeros(2, np.int32)
    rc = lib().pagan_dp_debug_route(C.byref(left.c), C.byref(right.c), C.byref(model.c), C.byref(band.c) if band is not None else None,
                                    n.ctypes.data_as(C.POINTER(C.c_int32)))
    _check(min(rc, 0), "pagan_dp_debug_route")
    return ROUTES[rc], bool(n[0]), int(n[1])


def debug_strips(left, right, band=None, max_sites=0):
    """Diagnostic (host only): the row strips a wide job would be filled as -- [(first row, last row, first diagonal, last
    diagonal + 1, feeder wave, first column staged, desc)] with desc[d - first diagonal] = (first row on d, last row, cell index
    of the first row's score in the job's arrays, class); [] when max_sites refuses the job."""
    import numpy as np
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    cap = Lx // 192 + 2
    desc_cap = (cap + 1) * (Ly + 192 + 64) + 64
    strips = np.zeros(6 * cap, np.int32)
    off = np.zeros(cap + 1, np.int64)
    desc = np.zeros(4 * desc_cap, np.int64)
    n = lib().pagan_dp_debug_strips(C.byref(left.c), C.byref(right.c), C.byref(band.c) if band is not None else None, max_sites,
                                    strips.ctypes.data_as(C.POINTER(C.c_int32)), cap, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                    desc.ctypes.data_as(C.POINTER(C.c_int64)), desc_cap)
    _check(min(n, 0), "pagan_dp_debug_strips")
    desc = desc.reshape(-1, 4)
    return [tuple(int(v) for v in strips[6 * k: 6 * k + 6]) + (desc[off[k]: off[k + 1]].copy(),) for k in range(n)]


def full_probability_batch(pairs, device=-1):
    """pagan_fb_run_batch: [(left, right, model_prob, band or None), ...] -> [FullProbability, ...]; the wide pairs' forward sweeps
    run in one launch and their backward sweeps in another (all pairs side by side on the device)."""
    L = lib()
    n = len(pairs)
    gp, mpp, bp = C.POINTER(abi.CGraph), C.POINTER(abi.CModelProb), C.POINTER(abi.CBand)
    lefts = (gp * n)(*[C.pointer(p[0].c) for p in pairs])
    rights = (gp * n)(*[C.pointer(p[1].c) for p in pairs])
    models = (mpp * n)(*[C.pointer(p[2].c) for p in pairs])
    bands = (bp * n)(*[(C.pointer(p[3].c) if p[3] is not None else bp()) for p in pairs])
    outs = (C.c_void_p * n)()
    opts = abi.COpts(0, device)
    _check(L.pagan_fb_run_batch(n, lefts, rights, models, bands, C.byref(opts), outs), "pagan_fb_run_batch")
    return [FullProbability(p[0], p[1], p[2], p[3], device=device, _handle=C.c_void_p(outs[k])) for k, p in enumerate(pairs)]


def debug_far(left, right, band=None):
    """Diagnostic (host only): the far histories of a banded job -- (n_served, hfL[Lx], hfR[Ly], hbit[nd], classes[nd])."""
    import numpy as np
    L = lib()
    Lx, Ly = left.n_sites - 1, right.n_sites - 1
    nd = Lx + Ly - 1
    hfl, hfr, hb, cls = np.zeros(Lx, np.uint8), np.zeros(Ly, np.uint8), np.zeros(nd, np.uint8), np.zeros(nd, np.uint8)
    u8 = C.POINTER(C.c_uint8)
    n = L.pagan_dp_debug_far(C.byref(left.c), C.byref(right.c), C.byref(band.c) if band is not None else None,
                             hfl.ctypes.data_as(u8), hfr.ctypes.data_as(u8), hb.ctypes.data_as(u8), cls.ctypes.data_as(u8))
    if n < 0:
        raise PaganError(n, "pagan_dp_debug_far")
    return n, hfl, hfr, hb, cls


def debug_plan(left, right, band=None, with_lead=False):
    """Diagnostic (host only): (classes[Lx+Ly-1] uint8, [awake intervals of wave 0..3]) the banded fill kernel
    would be given for this job; with_lead adds the per-diagonal downstream-progress requirement."""
    import numpy as np
    L = lib()
    nd = left.n_sites + right.n_sites - 3
    cls = np.zeros(nd, np.uint8)
    cap = 8 * nd + 64
    sched = np.zeros(cap, np.int32)
    n = C.c_int32()
    lead = np.zeros(nd, np.int32)
    _check(L.pagan_dp_debug_plan(C.byref(left.c), C.byref(right.c), C.byref(band.c) if band is not None else None,
                                 cls.ctypes.data_as(C.POINTER(C.c_uint8)), nd, sched.ctypes.data_as(C.POINTER(C.c_int32)),
                                 cap, C.byref(n), lead.ctypes.data_as(C.POINTER(C.c_int32))), "pagan_dp_debug_plan")
    waves = []
    for w in range(4):
        k = int(sched[w])
        iv = []
        while sched[k] < nd:
            iv.append((int(sched[k]), int(sched[k + 1])))
            k += 2
        waves.append(iv)
    if with_lead:
        return cls, waves, lead
    return cls, waves


class Batch:
    """Jobs uploaded once and kept resident in HBM; run() replays the hot path on them."""

    def __init__(self, jobs, flags=0, device=-1):
        self._L = lib()
        self.jobs = list(jobs)          # keeps the numpy arrays alive
        self.n = len(self.jobs)
        arr = _jobs_array(self.jobs)
        opts = abi.COpts(flags, device)
        self._h = C.c_void_p()
        _check(self._L.pagan_batch_create(self.n, arr, C.byref(opts), C.byref(self._h)), "pagan_batch_create")

    @property
    def cells(self):
        return self._L.pagan_batch_cells(self._h)

    def run(self):
        _check(self._L.pagan_batch_run(self._h), "pagan_batch_run")

    def sync(self):
        _check(self._L.pagan_batch_sync(self._h), "pagan_batch_sync")

    def last_ms(self):
        ms = (C.c_double * 2)()
        _check(self._L.pagan_batch_last_ms(self._h, ms), "pagan_batch_last_ms")
        return ms[0], ms[1]

    def fetch(self):
        res = (abi.CResult * self.n)()
        rc = self._L.pagan_batch_fetch(self._h, res)
        try:
            _check(rc, "pagan_batch_fetch")
            return [Result(r) for r in res]
        finally:
            for r in res:
                self._L.pagan_result_free(C.byref(r))

    def debug_scores(self, k):
        """Diagnostic: job k's device score array as [cells, 3] float64 (X, Y, M), diagonal-major."""
        import numpy as np
        out = np.empty((self.cells_of(k), 3), np.float64)
        _check(self._L.pagan_batch_debug_scores(self._h, k, out.ctypes.data_as(C.POINTER(C.c_double)), out.size),
               "pagan_batch_debug_scores")
        return out

    def debug_backptrs(self, k):
        """Diagnostic: job k's device back-pointer array as [cells, 3] uint32 (X, Y, M), diagonal-major."""
        import numpy as np
        out = np.empty((self.cells_of(k), 3), np.uint32)
        _check(self._L.pagan_batch_debug_backptrs(self._h, k, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size),
               "pagan_batch_debug_backptrs")
        return out

    def debug_followed(self, k):
        """Diagnostic: (chunks of 16 diagonals of job k whose back-pointers were written behind the banded fill by its
        follower workgroups, all chunks of the job); (0, 0) for a job of another kernel."""
        c = (C.c_int32 * 2)()
        _check(self._L.pagan_batch_debug_followed(self._h, k, c), "pagan_batch_debug_followed")
        return int(c[0]), int(c[1])

    def debug_poke_bp(self, k, i, j, vit, word):
        """Test hook: after the next run's fill, state `vit` of cell (i, j) of job k gets `word` as its back-pointer (once)."""
        _check(self._L.pagan_batch_debug_poke_bp(self._h, k, i, j, vit, word), "pagan_batch_debug_poke_bp")

    def debug_reruns(self):
        """How often fetch() ran the batch again after a failed path check."""
        return int(self._L.pagan_batch_debug_reruns(self._h))

    def debug_trace(self, k, n_cells):
        """Diagnostic: the first n_cells visited cells of job k's last traceback as [n, 3] int32 rows (i, j, word)."""
        import numpy as np
        out = np.empty((n_cells, 3), np.int32)
        _check(self._L.pagan_batch_debug_trace(self._h, k, out.ctypes.data_as(C.c_void_p), out.nbytes), "pagan_batch_debug_trace")
        return out

    def cells_of(self, k):
        left, right, _, band = self.jobs[k]
        return self._L.pagan_dp_count_cells(left.n_sites, right.n_sites, C.byref(band.c) if band is not None else None)

    def close(self):
        if self._h:
            self._L.pagan_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
