"""A second, deliberately different reading of the reference's pairwise aligner, in plain Python (test infrastructure).

oracle/oracle_dp.cpp restates Viterbi_alignment::align on flat CSR arrays.  This file restates the same source again,
independently and in the reference's own shapes, so that two readings can be compared on every golden and fuzz case
(parity is unpinned -- the reference cannot be built here -- and two independent readings that agree narrow what that
leaves open):

  * Site / Edge objects with per-site LINKED bwd lists and the iteration cursor of src/main/sequence.h:395-417
    (get_first_bwd_edge / has_next_bwd_edge / get_next_bwd_edge), not index ranges;
  * Matrix_pointer cells (src/main/basic_alignment.h:33-50) in a Tunnel_matrix-like store whose at() hands out a shared
    "empty" cell outside the tunnel (src/utils/tunnel_matrix.h:85-98);
  * the reference's fill ORDER: tunnel -> i outer / j inner, no tunnel -> j outer / i inner
    (src/main/viterbi_alignment.cpp:260-282), not anti-diagonals;
  * compute_fwd_scores / iterate_bwd_edges_for_gap / _for_match / _for_end_corner / score_* and backtrack_new_path with
    insert_preexisting_gap / insert_new_path_pointer written as the functions they are in the source
    (viterbi_alignment.cpp:856-971, 1038-1189, 1328-1567, 2029-2255; viterbi_alignment.h:127-200);
  * float parameters as numpy.float32 and promoted one by one, exactly where the source promotes them.
"""
import numpy as np

NEG = float("-inf")
X_MAT, Y_MAT, M_MAT = 0, 1, 2          # enum Matrix_pt, basic_alignment.h:107
NORMAL_GAP, END_GAP = 0, 1

f32 = np.float32


class Edge:                              # sequence.h:34-59
    def __init__(self, index, start, end, log_weight):
        self.index, self.start, self.end = index, start, end
        self.log_weight = f32(log_weight)
        self.next_bwd = None
        self.used = False


class Site:                              # sequence.h:216-251 (what the DP reads)
    def __init__(self, state):
        self.state = state
        self.first_bwd = None
        self.cursor = None

    def has_bwd_edge(self):
        return self.first_bwd is not None

    def get_first_bwd_edge(self):
        self.cursor = self.first_bwd
        return self.cursor

    def has_next_bwd_edge(self):
        return self.cursor is not None and self.cursor.next_bwd is not None

    def get_next_bwd_edge(self):
        self.cursor = self.cursor.next_bwd
        return self.cursor


class Sequence:
    def __init__(self, g):
        """g: pagan2_msa_amd.abi.Graph (CSR, bwd lists in iteration order)."""
        self.sites = [Site(int(s)) for s in g.state]
        self.edges = {}
        for s in range(g.n_sites):
            tail = None
            for k in range(int(g.bwd_off[s]), int(g.bwd_off[s + 1])):
                e = Edge(int(g.bwd_eid[k]), int(g.bwd_src[k]), s, g.bwd_logw[k])
                self.edges[e.index] = e
                if tail is None:
                    self.sites[s].first_bwd = e
                else:
                    tail.next_bwd = e
                tail = e

    def sites_length(self):
        return len(self.sites)

    def fwd_edge_index(self, start, end):
        """get_fwd_edge_index_at_site(start, Edge(start, end)): the edge start -> end, or -1."""
        s = self.sites[end]
        if not s.has_bwd_edge():
            return -1
        e = s.get_first_bwd_edge()
        while True:
            if e.start == start:
                return e.index
            if not s.has_next_bwd_edge():
                return -1
            e = s.get_next_bwd_edge()


class MatrixPointer:                     # basic_alignment.h:33-50
    __slots__ = ("score", "x_ind", "y_ind", "x_edge_ind", "y_edge_ind", "matrix")

    def __init__(self):
        self.score = NEG
        self.x_ind = self.y_ind = self.x_edge_ind = self.y_edge_ind = -1
        self.matrix = -1

    def copy(self):
        c = MatrixPointer()
        for k in self.__slots__:
            setattr(c, k, getattr(self, k))
        return c


class TunnelMatrix:                      # tunnel_matrix.h:45-344, semantics only
    def __init__(self, nx, ny, upper, lower):
        self.nx, self.ny = nx, ny
        self.empty = MatrixPointer()
        self.lo = [max(0, int(upper[i])) if upper is not None else 0 for i in range(nx)]
        self.hi = [min(int(lower[i]), ny - 1) if lower is not None else ny - 1 for i in range(nx)]
        self.cells = {}

    def inside(self, i, j):
        return 0 <= i < self.nx and self.lo[i] <= j <= self.hi[i]

    def at(self, i, j):
        if not self.inside(i, j):
            return self.empty                # a read outside the tunnel sees score -inf
        c = self.cells.get((i, j))
        if c is None:
            c = self.cells[(i, j)] = MatrixPointer()
        return c


def first_is_bigger(a, b):                   # basic_alignment.h:449-462
    if a == NEG and b == NEG:
        return False
    return a > b


class ViterbiAlignment:
    def __init__(self, left, right, model, band=None, flags=0):
        """left/right: abi.Graph; model: abi.Model; band: abi.Band or None; flags: PAGAN_OPT_* bits."""
        self.left, self.right = Sequence(left), Sequence(right)
        self.S = model.n_states
        self.table = model.table                                  # log_score(a, b) = table[a + b*S]
        self.log_gap_open, self.log_gap_ext, self.log_gap_end_ext, self.log_non_gap = (f32(x) for x in model.params)
        self.no_terminal_edges = bool(flags & 1)
        self.reduced_terminal_gap_penalties = not (flags & 2)     # basic_alignment.h:627-628
        self.tunnel = band is not None
        self.Lx, self.Ly = self.left.sites_length() - 1, self.right.sites_length() - 1
        up = band.upper if band is not None else None
        lo = band.lower if band is not None else None
        self.match, self.xgap, self.ygap = (TunnelMatrix(self.Lx, self.Ly, up, lo) for _ in range(3))

    # ---- penalties, basic_alignment.h:490-542 ----
    def get_log_gap_open_penalty(self, prev_site, is_x_matrix):
        if self.reduced_terminal_gap_penalties and prev_site == 0:
            return f32(0.0)
        return self.log_gap_open

    def log_score(self, a, b):
        return f32(self.table[a + b * self.S])

    # ---- align(), viterbi_alignment.cpp:187-465 ----
    def align(self):
        m00 = self.match.at(0, 0)
        m00.score = 0.0                                            # initialise_array_corner
        if self.tunnel:
            for i in range(self.Lx):
                for j in range(self.Ly):
                    if self.match.inside(i, j):
                        self.compute_fwd_scores(i, j)
        else:
            for j in range(self.Ly):
                for i in range(self.Lx):
                    self.compute_fwd_scores(i, j)
        self.max_end = MatrixPointer()
        self.iterate_bwd_edges_for_end_corner(self.left.sites[self.Lx], self.right.sites[self.Ly], self.max_end)
        if self.max_end.score == NEG:
            return None
        return self.backtrack_new_path(self.max_end)

    # ---- compute_fwd_scores, viterbi_alignment.cpp:856-971 ----
    def compute_fwd_scores(self, i, j):
        if i == 0 and j == 0:
            return
        x_gap_type = y_gap_type = NORMAL_GAP
        if (j == 0 or j == self.Ly - 1) and not self.no_terminal_edges:
            x_gap_type = END_GAP
        if (i == 0 or i == self.Lx - 1) and not self.no_terminal_edges:
            y_gap_type = END_GAP
        if i > 0:
            max_x = self.xgap.at(i, j)
            self.iterate_bwd_edges_for_gap(self.left.sites[i], lambda p: self.xgap.at(p, j), lambda p: self.ygap.at(p, j),
                                           lambda p: self.match.at(p, j), max_x, True, x_gap_type)
            max_x.y_ind = j
        if j > 0:
            max_y = self.ygap.at(i, j)
            self.iterate_bwd_edges_for_gap(self.right.sites[j], lambda q: self.ygap.at(i, q), lambda q: self.xgap.at(i, q),
                                           lambda q: self.match.at(i, q), max_y, False, y_gap_type)
            max_y.x_ind = i
        if i > 0 and j > 0:
            self.iterate_bwd_edges_for_match(self.left.sites[i], self.right.sites[j], self.match.at(i, j))

    # ---- iterate_bwd_edges_for_gap, :1328-1349 ----
    def iterate_bwd_edges_for_gap(self, site, z_slice, w_slice, m_slice, mx, is_x, gap_type):
        if site.has_bwd_edge():
            edge = site.get_first_bwd_edge()
            self.score_gap_ext(edge, z_slice, mx, is_x, gap_type)
            self.score_gap_double(edge, w_slice, mx, is_x)
            self.score_gap_open(edge, m_slice, mx, is_x)
            while site.has_next_bwd_edge():
                edge = site.get_next_bwd_edge()
                self.score_gap_ext(edge, z_slice, mx, is_x, gap_type)
                self.score_gap_double(edge, w_slice, mx, is_x)
                self.score_gap_open(edge, m_slice, mx, is_x)

    def _take_gap(self, mx, score, prev, edge, matrix, is_x):
        mx.score = score
        mx.matrix = matrix
        if is_x:
            mx.x_ind, mx.x_edge_ind = prev, edge.index
        else:
            mx.y_ind, mx.y_edge_ind = prev, edge.index

    def score_gap_ext(self, edge, z_slice, mx, is_x, gap_type):    # :2116-2156
        prev = edge.start
        this = z_slice(prev).score + float(self.log_gap_ext)
        if gap_type == END_GAP:
            this = z_slice(prev).score + float(self.log_gap_end_ext)
        if first_is_bigger(this, mx.score):
            self._take_gap(mx, this, prev, edge, X_MAT if is_x else Y_MAT, is_x)

    def score_gap_double(self, edge, w_slice, mx, is_x):           # :2158-2188; log_gap_close() is 0.0f
        prev = edge.start
        this = w_slice(prev).score + float(f32(0.0)) + float(self.log_gap_open)
        if first_is_bigger(this, mx.score):
            self._take_gap(mx, this, prev, edge, Y_MAT if is_x else X_MAT, is_x)

    def score_gap_open(self, edge, m_slice, mx, is_x):             # :2190-2219
        prev = edge.start
        this = m_slice(prev).score + float(self.log_non_gap) + float(self.get_log_gap_open_penalty(prev, is_x))
        if first_is_bigger(this, mx.score):
            self._take_gap(mx, this, prev, edge, M_MAT, is_x)

    # ---- iterate_bwd_edges_for_match, :1353-1436 ----
    def iterate_bwd_edges_for_match(self, left_site, right_site, mx):
        if not (left_site.has_bwd_edge() and right_site.has_bwd_edge()):
            return
        left_edge = left_site.get_first_bwd_edge()
        right_edge = right_site.get_first_bwd_edge()
        log_match_score = float(self.log_score(left_site.state, right_site.state))
        m_log_match = float(f32(2) * self.log_non_gap) + log_match_score               # 2*float stays a float product
        x_log_match = float(f32(0.0) + self.log_non_gap) + log_match_score             # close penalty 0.0f + float
        y_log_match = float(f32(0.0) + self.log_non_gap) + log_match_score

        def three(le, re):
            self.score_match(le, re, m_log_match, mx, self.match, M_MAT)
            self.score_match(le, re, x_log_match, mx, self.xgap, X_MAT)
            self.score_match(le, re, y_log_match, mx, self.ygap, Y_MAT)
        three(left_edge, right_edge)
        while right_site.has_next_bwd_edge():                       # first right site extra edges
            right_edge = right_site.get_next_bwd_edge()
            left_edge = left_site.get_first_bwd_edge()
            three(left_edge, right_edge)
        while left_site.has_next_bwd_edge():                        # left site extra edges then
            left_edge = left_site.get_next_bwd_edge()
            right_edge = right_site.get_first_bwd_edge()
            three(left_edge, right_edge)
            while right_site.has_next_bwd_edge():
                right_edge = right_site.get_next_bwd_edge()
                three(left_edge, right_edge)

    def score_match(self, left_edge, right_edge, log_match, mx, source, matrix):        # :2029-2112
        p, q = left_edge.start, right_edge.start
        this = source.at(p, q).score + log_match + float(left_edge.log_weight) + float(right_edge.log_weight)
        if first_is_bigger(this, mx.score):
            mx.score = this
            mx.x_ind, mx.y_ind = p, q
            mx.x_edge_ind, mx.y_edge_ind = left_edge.index, right_edge.index
            mx.matrix = matrix

    # ---- end corner, :1440-1552, score_gap_close :2221-2255 ----
    def score_gap_close(self, edge, z_slice, mx, is_x):
        prev = edge.start
        this = z_slice(prev).score + float(f32(0.0))
        if first_is_bigger(this, mx.score):
            mx.score = this
            if is_x:
                mx.matrix, mx.x_ind, mx.x_edge_ind, mx.y_edge_ind = X_MAT, prev, edge.index, -1
            else:
                mx.matrix, mx.y_ind, mx.y_edge_ind, mx.x_edge_ind = Y_MAT, prev, edge.index, -1

    def iterate_bwd_edges_for_end_corner(self, left_site, right_site, mx):
        if not (left_site.has_bwd_edge() and right_site.has_bwd_edge()):
            return
        left_edge = left_site.get_first_bwd_edge()
        right_edge = right_site.get_first_bwd_edge()
        m_log_match = float(self.log_non_gap)
        x_slice = lambda p: self.xgap.at(p, self.Ly - 1)
        y_slice = lambda q: self.ygap.at(self.Lx - 1, q)
        best = [NEG]

        def m_step(le, re):
            self.score_match(le, re, m_log_match, mx, self.match, M_MAT)
            if first_is_bigger(mx.score, best[0]):
                best[0] = mx.score

        def x_close(le):
            self.score_gap_close(le, x_slice, mx, True)
            if first_is_bigger(mx.score, best[0]):
                best[0] = mx.score
                mx.y_ind = self.Ly - 1

        def y_close(re):
            self.score_gap_close(re, y_slice, mx, False)
            if first_is_bigger(mx.score, best[0]):
                best[0] = mx.score
                mx.x_ind = self.Lx - 1
        self.score_match(left_edge, right_edge, m_log_match, mx, self.match, M_MAT)
        best[0] = mx.score
        x_close(left_edge)
        y_close(right_edge)
        while right_site.has_next_bwd_edge():
            right_edge = right_site.get_next_bwd_edge()
            left_edge = left_site.get_first_bwd_edge()
            m_step(left_edge, right_edge)
            y_close(right_edge)
        while left_site.has_next_bwd_edge():
            left_edge = left_site.get_next_bwd_edge()
            right_edge = right_site.get_first_bwd_edge()
            m_step(left_edge, right_edge)
            x_close(left_edge)
            while right_site.has_next_bwd_edge():
                right_edge = right_site.get_next_bwd_edge()
                m_step(left_edge, right_edge)
                y_close(right_edge)

    # ---- traceback, :1038-1189 + viterbi_alignment.h:127-200 ----
    def backtrack_new_path(self, fp):
        stack = []                                                  # (matrix, real_site)
        le, re = self.left.edges, self.right.edges
        vit_mat, x_ind, y_ind = fp.matrix, fp.x_ind, fp.y_ind
        first_x_site = first_y_site = True
        if fp.x_edge_ind >= 0:
            le[fp.x_edge_ind].used = True
        if fp.y_edge_ind >= 0:
            re[fp.y_edge_ind].used = True
        pos = [self.Lx - 1, self.Ly - 1]                            # i, j
        max_i, max_j = self.Lx, self.Ly

        def insert_preexisting_gap(x_ind, y_ind):
            while x_ind < pos[0]:
                stack.append((X_MAT, False))
                pos[0] -= 1
            while y_ind < pos[1]:
                stack.append((Y_MAT, False))
                pos[1] -= 1

        def insert_new_path_pointer(matrix):
            if pos[0] > 0 or pos[1] > 0:
                stack.append((matrix, True))
        insert_preexisting_gap(x_ind, y_ind)
        insert_new_path_pointer(fp.matrix)
        while True:
            i, j = pos
            if vit_mat == M_MAT:
                if first_x_site:
                    k = self.left.fwd_edge_index(x_ind, max_i)
                    if k >= 0:
                        le[k].used = True
                    first_x_site = False
                if first_y_site:
                    k = self.right.fwd_edge_index(y_ind, max_j)
                    if k >= 0:
                        re[k].used = True
                    first_y_site = False
                c = self.match.at(i, j)
                vit_mat, x_ind, y_ind = c.matrix, c.x_ind, c.y_ind
                le[c.x_edge_ind].used = True
                re[c.y_edge_ind].used = True
                pos[0] -= 1
                pos[1] -= 1
            elif vit_mat == X_MAT:
                if first_x_site:
                    k = self.left.fwd_edge_index(x_ind, max_i)
                    if k >= 0:
                        le[k].used = True
                    first_x_site = False
                c = self.xgap.at(i, j)
                vit_mat, x_ind, y_ind = c.matrix, c.x_ind, c.y_ind
                le[c.x_edge_ind].used = True
                pos[0] -= 1
            elif vit_mat == Y_MAT:
                if first_y_site:
                    k = self.right.fwd_edge_index(y_ind, max_j)
                    if k >= 0:
                        re[k].used = True
                    first_y_site = False
                c = self.ygap.at(i, j)
                vit_mat, x_ind, y_ind = c.matrix, c.x_ind, c.y_ind
                re[c.y_edge_ind].used = True
                pos[1] -= 1
            else:
                raise RuntimeError("incorrect backward pointer %r at %r" % (vit_mat, (i, j)))
            insert_preexisting_gap(x_ind, y_ind)
            insert_new_path_pointer(c.matrix)
            if pos[0] < 1 and pos[1] < 1:
                break
        return stack[::-1]

    # ---- what the parent-graph builder consumes (create_ancestral_sequence, basic_alignment.cpp:73-171) ----
    def columns(self, path):
        cols, l_pos, r_pos = [], 1, 1
        for matrix, real in path:
            if matrix == X_MAT:
                cols.append((l_pos, -1, 3 if real else 5))
                l_pos += 1
            elif matrix == Y_MAT:
                cols.append((-1, r_pos, 4 if real else 6))
                r_pos += 1
            else:
                cols.append((l_pos, r_pos, 2))
                l_pos += 1
                r_pos += 1
        return np.array(cols, np.int32).reshape(-1, 3)


def align(left, right, model, band=None, flags=0):
    """Returns a dict shaped like abi.Result: status, score, end, cols, left_used, right_used."""
    va = ViterbiAlignment(left, right, model, band, flags)
    path = va.align()
    me = va.max_end
    out = {"status": 0 if path is not None else 1, "score": me.score,
           "end": (me.matrix, me.x_ind, me.y_ind, me.x_edge_ind, me.y_edge_ind)}
    if path is None:
        return out
    out["cols"] = va.columns(path)
    out["left_used"] = np.array(sorted(k for k, e in va.left.edges.items() if e.used), np.int32)
    out["right_used"] = np.array(sorted(k for k, e in va.right.edges.items() if e.used), np.int32)
    return out


def same(py, res):
    """py: dict from align(); res: abi.Result (oracle or GPU)."""
    if py["status"] != res.status:
        return False
    if py["status"] != 0:
        return True
    return (np.float64(py["score"]).tobytes() == np.float64(res.score).tobytes() and tuple(py["end"]) == tuple(res.end) and
            np.array_equal(py["cols"], res.cols) and np.array_equal(py["left_used"], res.left_used) and
            np.array_equal(py["right_used"], res.right_used))
