"""Host WFST objects of the MI355X engine: a thin Python face over the C++ graph library in
libwfl.so (csrc/graph.cpp), with the method names the reference's criteria use on `gtn.Graph`
 so that the graph builders read like the reference's.

Unlike gtn there is no graph-level autograd here: differentiation happens in the device engine
(`engine.py`), which treats the emissions tensor as the implicit first operand of
`intersect(emissions, A)` and never materialises the composed lattice.
"""
import ctypes

import numpy as np

from . import _native as N

epsilon = N.EPSILON


class Graph:
    """Weighted finite-state transducer; arcs are numbered in insertion order
    (what `set_weights` / `transition_params` rely on: asg.py:66, transducer.py:174-179)."""

    __slots__ = ("_h", "calc_grad", "__weakref__")

    def __init__(self, calc_grad=True, _handle=None):
        self._h = N.check_handle(_handle if _handle is not None else N.lib.wfl_graph_new())
        self.calc_grad = bool(calc_grad)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and N is not None and getattr(N, "lib", None) is not None:
            N.lib.wfl_graph_free(h)

    # ---- construction -------------------------------------------------------------------------
    def add_node(self, start=False, accept=False):
        return N.lib.wfl_graph_add_node(self._h, int(bool(start)), int(bool(accept)))

    def add_arc(self, src, dst, ilabel, olabel=None, weight=0.0):
        a = N.lib.wfl_graph_add_arc(
            self._h, int(src), int(dst), int(ilabel), int(ilabel if olabel is None else olabel), float(weight)
        )
        if a < 0:
            raise N.WflError(N.ERR_INVALID, N.last_error())
        return a

    def add_nodes(self, start, accept):
        start = np.ascontiguousarray(start, dtype=np.uint8)
        accept = np.ascontiguousarray(accept, dtype=np.uint8)
        N.check(N.lib.wfl_graph_add_nodes(self._h, len(start), start.ctypes.data, accept.ctypes.data))

    def add_arcs(self, src, dst, ilabel, olabel=None, weight=None):
        """Bulk add_arc (same ordering semantics); arrays of equal length."""
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.ascontiguousarray(dst, dtype=np.int32)
        il = np.ascontiguousarray(ilabel, dtype=np.int32)
        ol = None if olabel is None else np.ascontiguousarray(olabel, dtype=np.int32)
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float32)
        N.check(
            N.lib.wfl_graph_add_arcs(
                self._h, len(src), src.ctypes.data, dst.ctypes.data, il.ctypes.data,
                None if ol is None else ol.ctypes.data, None if w is None else w.ctypes.data,
            )
        )

    # ---- inspection ---------------------------------------------------------------------------
    def num_nodes(self):
        return N.lib.wfl_graph_num_nodes(self._h)

    def num_arcs(self):
        return int(N.lib.wfl_graph_num_arcs(self._h))

    def arrays(self):
        """dict of numpy copies: start, accept [nodes]; src, dst, ilabel, olabel, weight [arcs]."""
        n, m = self.num_nodes(), self.num_arcs()
        out = dict(
            start=np.zeros(n, np.uint8), accept=np.zeros(n, np.uint8), src=np.zeros(m, np.int32),
            dst=np.zeros(m, np.int32), ilabel=np.zeros(m, np.int32), olabel=np.zeros(m, np.int32),
            weight=np.zeros(m, np.float32),
        )
        N.check(N.lib.wfl_graph_get(self._h, *(out[k].ctypes.data for k in
                                                ("start", "accept", "src", "dst", "ilabel", "olabel", "weight"))))
        return out

    def weights_to_numpy(self):
        w = np.zeros(self.num_arcs(), np.float32)
        N.check(N.lib.wfl_graph_get(self._h, None, None, None, None, None, None, w.ctypes.data))
        return w

    def labels_to_list(self, ilabel=True):
        return self.arrays()["ilabel" if ilabel else "olabel"].tolist()

    def item(self):
        if self.num_arcs() != 1:
            raise ValueError("item() needs a graph with exactly one arc")
        return float(self.weights_to_numpy()[0])

    def set_weights(self, data):
        """num_arcs float32 in arc-id order; accepts an array/tensor or a raw host pointer (int)."""
        m = self.num_arcs()
        if isinstance(data, int):
            ptr = data
        else:
            if hasattr(data, "detach"):
                data = data.detach().cpu().numpy()
            data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
            if data.size != m:
                raise ValueError("set_weights: size mismatch")
            ptr = data.ctypes.data
        N.check(N.lib.wfl_graph_set_weights(self._h, ptr))

    def arc_sort(self, olabel=False):
        N.check(N.lib.wfl_graph_arc_sort(self._h, int(bool(olabel))))

    def mark_arc_sorted(self, olabel=False):
        pass  # label-sorted adjacency is maintained internally by the library

    def zero_grad(self):
        pass  # gradients live in the device engine, not on graphs

    def __repr__(self):
        a = self.arrays()
        lines = [" ".join(map(str, np.nonzero(a["start"])[0])), " ".join(map(str, np.nonzero(a["accept"])[0]))]
        lines += [f"{s} {d} {i} {o} {w:g}" for s, d, i, o, w in
                  zip(a["src"], a["dst"], a["ilabel"], a["olabel"], a["weight"])]
        return "\n".join(lines)


def _take(ptr, n):
    """Copy a malloc'ed int32 array returned by the library and release it."""
    if not ptr:
        return np.zeros(0, np.int32)
    arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int32)), shape=(max(n, 1),))[:n].copy()
    N.lib.wfl_free(ptr)
    return arr


def compose(first, second, provenance=False):
    """gtn.compose: first.olabel x second.ilabel, epsilons advance alone, weights add, trimmed.
    With provenance=True also returns (arc ids in first, arc ids in second), -1 where absent."""
    if provenance:
        p1, p2 = ctypes.c_void_p(), ctypes.c_void_p()
        h = N.check_handle(N.lib.wfl_graph_compose(first._h, second._h, ctypes.byref(p1), ctypes.byref(p2)))
        g = Graph(first.calc_grad or second.calc_grad, _handle=h)
        m = g.num_arcs()
        return g, _take(p1.value, m), _take(p2.value, m)
    h = N.check_handle(N.lib.wfl_graph_compose(first._h, second._h, None, None))
    return Graph(first.calc_grad or second.calc_grad, _handle=h)


intersect = compose


def remove(g, ilabel=epsilon, olabel=None, provenance=False):
    olabel = ilabel if olabel is None else olabel
    if provenance:
        p = ctypes.c_void_p()
        h = N.check_handle(N.lib.wfl_graph_remove(g._h, ilabel, olabel, ctypes.byref(p)))
        out = Graph(g.calc_grad, _handle=h)
        return out, _take(p.value, out.num_arcs())
    return Graph(g.calc_grad, _handle=N.check_handle(N.lib.wfl_graph_remove(g._h, ilabel, olabel, None)))


def project_input(g):
    return Graph(g.calc_grad, _handle=N.check_handle(N.lib.wfl_graph_project(g._h, 0)))


def project_output(g):
    return Graph(g.calc_grad, _handle=N.check_handle(N.lib.wfl_graph_project(g._h, 1)))


def viterbi_path(g):
    """Best path of a (small, acyclic) host graph; ties: fewest output labels (transducer.py:226)."""
    return Graph(False, _handle=N.check_handle(N.lib.wfl_graph_viterbi_path(g._h)))


def token_alignments(tokens, tokens_target):
    """project_input(remove(compose(tokens, tokens_target))) written down directly (wfl_graph_token_alignments), or
    None if `tokens` is not make_token_graph(N, blank="optional", allow_repeats=False)."""
    h = N.lib.wfl_graph_token_alignments(tokens._h, tokens_target._h)
    return Graph(False, _handle=h) if h else None


def transducer_decode_batch(tokens, labels, offsets, nthreads=0):
    """Transducer.viterbi's decode stage for a whole batch in one native call (wfl_transducer_decode_batch;
    transducer.py:221-232 under gtn.parallel_for): per utterance the output labels of
    viterbi_path(compose(chain(labels_b), tokens)).  labels: flat int32, offsets: int64 [B+1].
    Returns (flat int32 token labels, int64 offsets [B+1])."""
    labels = np.ascontiguousarray(labels, dtype=np.int32).reshape(-1)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64).reshape(-1)
    B = len(offsets) - 1
    out = np.empty(max(int(offsets[-1] - offsets[0]), 1), np.int32)
    out_off = np.zeros(B + 1, np.int64)
    N.check(N.lib.wfl_transducer_decode_batch(tokens._h, labels.ctypes.data, offsets.ctypes.data, B, out.ctypes.data,
                                              out.size, out_off.ctypes.data, int(nthreads)))
    return out[:out_off[B]], out_off


def equal(a, b):
    return bool(N.lib.wfl_graph_equal(a._h, b._h))


def isomorphic(a, b):
    return bool(N.lib.wfl_graph_isomorphic(a._h, b._h))


def loadtxt(path):
    return Graph(True, _handle=N.check_handle(N.lib.wfl_graph_loadtxt(str(path).encode())))


def savetxt(path, g):
    N.check(N.lib.wfl_graph_savetxt(g._h, str(path).encode()))


def load(path):
    """gtn.load (utils.py:261): sniffs gtn text vs gtn's binary layout; the binary layout is restated from gtn's
    published source and unpinned (include/wfl.h) -- anything inconsistent raises instead of being mis-read."""
    return Graph(True, _handle=N.check_handle(N.lib.wfl_graph_load(str(path).encode())))


def save(path, g):
    """gtn.save (scripts/build_transitions.py:221): binary layout, see load()."""
    N.check(N.lib.wfl_graph_save(g._h, str(path).encode()))


def linear_graph(M, N_, device=None, calc_grad=True):
    """gtn.linear_graph as an explicit host graph (M+1 nodes, arc id = t*N + c).  The device engine
    never builds this -- it exists for tests and for API parity."""
    g = Graph(calc_grad)
    g.add_nodes([1] + [0] * M, [0] * M + [1])
    t = np.repeat(np.arange(M, dtype=np.int32), N_)
    c = np.tile(np.arange(N_, dtype=np.int32), M)
    if M * N_:
        g.add_arcs(t, t + 1, c)
    return g
