"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under gtn_applications_amd/ may import this file.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.

minigtn: a small, pure-Python, float64 restatement of the subset of the external `gtn` library
(github.com/facebookresearch/gtn, C++17 + pybind11; NOT vendored under /root/reference and NOT
pinned: /root/reference/requirements.txt:1, /root/reference/README.md:11) that the reference's hot
path calls.  The call sites it has to serve are

    criterions/ctc.py:15-94, criterions/asg.py:54-185,211-237, criterions/stc.py:23-129,
    criterions/transducer.py:15-123,185-348,351-367,461-556, utils.py:261 (gtn.load/loadtxt)

and the semantics restated here are the published ones of that library:

  * Graph: nodes with start/accept flags; arcs (src, dst, ilabel, olabel, weight) numbered in
    insertion order; epsilon == -1 (pinned by tests/trans_backoff_test.txt:3 `1 0 -1 -1 0`).
  * linear_graph(M, N): M+1 nodes in a chain, N parallel arcs per step, arc id = t*N + c.
  * compose / intersect: match first.olabel with second.ilabel; an epsilon on the first's output or
    on the second's input advances that side alone; weights add; only states that are accessible
    and co-accessible survive.
  * remove(g, label): epsilon-closure based removal (weights of removed arcs are dropped -- the
    reference only ever removes zero-weight arcs).
  * forward_score: log-sum-exp over all accepting paths (Kahn topological sweep, DAG only);
    viterbi_score / viterbi_path: max-plus analogue (strict '>' relaxations, first wins).
  * negate / subtract / add on 1-arc "scalar graphs"; reverse-mode autograd (`backward`) through
    the op DAG with gradients that ACCUMULATE on leaf graphs until zero_grad().

Parity status: this is a restatement, not the upstream code.  It is pinned by running the
reference's own criterion sources and unit tests against it (oracle/pin_against_reference.py, which
only runs where /root/reference exists) -- every literal known-answer vector in
/root/reference/tests/*.py.  Beyond those vectors parity with upstream gtn is unpinned.

Written for clarity, not speed: everything is Python lists + numpy float64.
"""
import ctypes
import math
from collections import deque

import numpy as np

epsilon = -1
CPU = 0
NEG_INF = float("-inf")


class Device:  # placeholder so that `gtn.Device(gtn.CPU)` works (ctc.py:41)
    def __init__(self, kind=CPU, index=0):
        self.kind = kind


def _logadd(a, b):
    if a == NEG_INF:
        return b
    if b == NEG_INF:
        return a
    m = a if a > b else b
    return m + math.log(math.exp(a - m) + math.exp(b - m))


class Graph:
    """Weighted finite-state transducer with insertion-ordered arcs."""

    def __init__(self, calc_grad=True):
        if isinstance(calc_grad, Graph):  # tolerate `gtn.Graph(gtn.CPU)`-style misuse (transducer_test.py:256)
            calc_grad = True
        self._calc_grad = bool(calc_grad)
        self.start_flag = []
        self.accept_flag = []
        self.src = []
        self.dst = []
        self.ilab = []
        self.olab = []
        self.w = []  # python floats (float64)
        self.out_arcs = []  # per node: arc ids in iteration order
        self.in_arcs = []
        # autograd
        self._inputs = []
        self._grad_fn = None
        self._grad = None
        self._ilabel_sorted = False
        self._olabel_sorted = False

    # ---- construction -------------------------------------------------------------------------
    def add_node(self, start=False, accept=False):
        self.start_flag.append(bool(start))
        self.accept_flag.append(bool(accept))
        self.out_arcs.append([])
        self.in_arcs.append([])
        return len(self.start_flag) - 1

    def add_arc(self, src, dst, ilabel, olabel=None, weight=0.0):
        if olabel is None:
            olabel = ilabel
        a = len(self.src)
        self.src.append(int(src))
        self.dst.append(int(dst))
        self.ilab.append(int(ilabel))
        self.olab.append(int(olabel))
        self.w.append(float(weight))
        self.out_arcs[src].append(a)
        self.in_arcs[dst].append(a)
        self._ilabel_sorted = self._olabel_sorted = False
        return a

    # ---- inspection ---------------------------------------------------------------------------
    def num_nodes(self):
        return len(self.start_flag)

    def num_arcs(self):
        return len(self.src)

    def num_start(self):
        return sum(self.start_flag)

    def num_accept(self):
        return sum(self.accept_flag)

    def start_nodes(self):
        return [n for n, f in enumerate(self.start_flag) if f]

    def accept_nodes(self):
        return [n for n, f in enumerate(self.accept_flag) if f]

    def item(self):
        if self.num_arcs() != 1:
            raise ValueError("item() needs a graph with exactly one arc")
        return self.w[0]

    def weights_to_numpy(self):
        return np.asarray(self.w, dtype=np.float32)  # upstream returns float32

    def weights_to_list(self):
        return list(self.w)

    def weights64(self):
        return np.asarray(self.w, dtype=np.float64)

    def labels_to_list(self, ilabel=True):
        return list(self.ilab if ilabel else self.olab)

    def set_weights(self, data):
        n = self.num_arcs()
        if isinstance(data, int):  # raw host pointer to float32 (ctc.py:44)
            buf = (ctypes.c_float * n).from_address(data)
            vals = [float(v) for v in buf]
        else:
            vals = [float(v) for v in np.asarray(data).reshape(-1)]
            if len(vals) != n:
                raise ValueError("set_weights: size mismatch")
        self.w = vals

    @property
    def calc_grad(self):
        return self._calc_grad

    @calc_grad.setter
    def calc_grad(self, v):
        self._calc_grad = bool(v)
        if not v:
            self._grad = None

    def zero_grad(self):
        self._grad = None

    def grad(self):
        if self._grad is None:
            raise RuntimeError("no gradient computed for this graph")
        g = Graph(False)
        for n in range(self.num_nodes()):
            g.add_node(self.start_flag[n], self.accept_flag[n])
        for a in range(self.num_arcs()):
            g.add_arc(self.src[a], self.dst[a], self.ilab[a], self.olab[a], float(self._grad[a]))
        return g

    def grad64(self):
        return None if self._grad is None else self._grad.copy()

    def _add_grad(self, delta):
        if not self._calc_grad:
            return
        if self._grad is None:
            self._grad = np.zeros(self.num_arcs(), dtype=np.float64)
        self._grad += delta

    # ---- arc ordering -------------------------------------------------------------------------
    def arc_sort(self, olabel=False):
        key = (lambda a: self.olab[a]) if olabel else (lambda a: self.ilab[a])
        for n in range(self.num_nodes()):
            self.out_arcs[n].sort(key=key)  # stable
            self.in_arcs[n].sort(key=key)
        if olabel:
            self._olabel_sorted, self._ilabel_sorted = True, False
        else:
            self._ilabel_sorted, self._olabel_sorted = True, False

    def mark_arc_sorted(self, olabel=False):
        if olabel:
            self._olabel_sorted = True
        else:
            self._ilabel_sorted = True

    def __repr__(self):
        lines = [" ".join(map(str, self.start_nodes())), " ".join(map(str, self.accept_nodes()))]
        for a in range(self.num_arcs()):
            lines.append(f"{self.src[a]} {self.dst[a]} {self.ilab[a]} {self.olab[a]} {self.w[a]}")
        return "\n".join(lines)


# ------------------------------------------------------------------------------------------------
# creation helpers
# ------------------------------------------------------------------------------------------------
def linear_graph(M, N, device=None, calc_grad=True):
    if isinstance(device, bool):  # linear_graph(M, N, calc_grad)
        calc_grad = device
    g = Graph(calc_grad)
    g.add_node(True, M == 0)
    for t in range(M):
        g.add_node(False, t == M - 1)
        for c in range(N):
            g.add_arc(t, t + 1, c)
    g.mark_arc_sorted(False)
    g.mark_arc_sorted(True)
    return g


def scalar_graph(weight=0.0, calc_grad=True):
    g = Graph(calc_grad)
    g.add_node(True)
    g.add_node(False, True)
    g.add_arc(0, 1, epsilon, epsilon, weight)
    return g


def _structure_copy(g, calc_grad):
    out = Graph(calc_grad)
    for n in range(g.num_nodes()):
        out.add_node(g.start_flag[n], g.accept_flag[n])
    for a in range(g.num_arcs()):
        out.add_arc(g.src[a], g.dst[a], g.ilab[a], g.olab[a], g.w[a])
    return out


def clone(g, projection=None):
    out = _structure_copy(g, g.calc_grad)
    if projection == "input":
        out.olab = list(out.ilab)
    elif projection == "output":
        out.ilab = list(out.olab)
    out._inputs = [g]
    out._grad_fn = lambda delta: g._add_grad(delta)
    return out


def project_input(g):
    return clone(g, "input")


def project_output(g):
    return clone(g, "output")


# ------------------------------------------------------------------------------------------------
# compose / intersect
# ------------------------------------------------------------------------------------------------
def compose(g1, g2):
    """first.olabel matched with second.ilabel; epsilon on either matching side moves alone."""
    n2 = g2.num_nodes()

    def pid(a, b):
        return a * n2 + b

    # index arcs of g2 by ilabel per node, arcs of g1 by olabel per node (keeps iteration order)
    def by_label(g, labs, arcs_of):
        table = []
        for n in range(g.num_nodes()):
            d = {}
            for a in arcs_of[n]:
                d.setdefault(labs[a], []).append(a)
            table.append(d)
        return table

    out1 = by_label(g1, g1.olab, g1.out_arcs)
    out2 = by_label(g2, g2.ilab, g2.out_arcs)
    in1 = by_label(g1, g1.olab, g1.in_arcs)
    in2 = by_label(g2, g2.ilab, g2.in_arcs)

    # 1. co-accessible state pairs: backward search from (accept, accept)
    coacc = set()
    queue = deque()
    for a in g1.accept_nodes():
        for b in g2.accept_nodes():
            coacc.add(pid(a, b))
            queue.append((a, b))
    while queue:
        a, b = queue.popleft()
        for lab, arcs1 in in1[a].items():
            if lab == epsilon:
                for x in arcs1:
                    p = pid(g1.src[x], b)
                    if p not in coacc:
                        coacc.add(p)
                        queue.append((g1.src[x], b))
            else:
                arcs2 = in2[b].get(lab)
                if arcs2:
                    for x in arcs1:
                        for y in arcs2:
                            p = pid(g1.src[x], g2.src[y])
                            if p not in coacc:
                                coacc.add(p)
                                queue.append((g1.src[x], g2.src[y]))
        for y in in2[b].get(epsilon, ()):
            p = pid(a, g2.src[y])
            if p not in coacc:
                coacc.add(p)
                queue.append((a, g2.src[y]))

    # 2. forward exploration restricted to co-accessible pairs
    out = Graph(g1.calc_grad or g2.calc_grad)
    node_of = {}
    queue = deque()
    origin = []  # per output arc: (arc in g1 or -1, arc in g2 or -1)

    def get_node(a, b):
        p = pid(a, b)
        n = node_of.get(p)
        if n is None:
            n = out.add_node(
                g1.start_flag[a] and g2.start_flag[b], g1.accept_flag[a] and g2.accept_flag[b]
            )
            node_of[p] = n
            queue.append((a, b))
        return n

    for a in g1.start_nodes():
        for b in g2.start_nodes():
            if pid(a, b) in coacc:
                get_node(a, b)
    while queue:
        a, b = queue.popleft()
        cur = node_of[pid(a, b)]
        for x in g1.out_arcs[a]:
            lab = g1.olab[x]
            if lab == epsilon:
                if pid(g1.dst[x], b) in coacc:
                    d = get_node(g1.dst[x], b)
                    out.add_arc(cur, d, g1.ilab[x], epsilon, g1.w[x])
                    origin.append((x, -1))
            else:
                for y in out2[b].get(lab, ()):
                    if pid(g1.dst[x], g2.dst[y]) in coacc:
                        d = get_node(g1.dst[x], g2.dst[y])
                        out.add_arc(cur, d, g1.ilab[x], g2.olab[y], g1.w[x] + g2.w[y])
                        origin.append((x, y))
        for y in out2[b].get(epsilon, ()):
            if pid(a, g2.dst[y]) in coacc:
                d = get_node(a, g2.dst[y])
                out.add_arc(cur, d, epsilon, g2.olab[y], g2.w[y])
                origin.append((-1, y))

    def grad_fn(delta):
        if g1.calc_grad:
            d1 = np.zeros(g1.num_arcs())
            for k, (x, _) in enumerate(origin):
                if x >= 0:
                    d1[x] += delta[k]
            g1._add_grad(d1)
        if g2.calc_grad:
            d2 = np.zeros(g2.num_arcs())
            for k, (_, y) in enumerate(origin):
                if y >= 0:
                    d2[y] += delta[k]
            g2._add_grad(d2)

    out._inputs = [g1, g2]
    out._grad_fn = grad_fn
    return out


def intersect(g1, g2):
    return compose(g1, g2)


# ------------------------------------------------------------------------------------------------
# remove
# ------------------------------------------------------------------------------------------------
def remove(g, ilabel=epsilon, olabel=None):
    if olabel is None:
        olabel = ilabel

    def match(a):
        return g.ilab[a] == ilabel and g.olab[a] == olabel

    out = Graph(g.calc_grad)
    new_id = [-1] * g.num_nodes()
    for n in range(g.num_nodes()):
        keep = g.start_flag[n] or any(not match(a) for a in g.in_arcs[n])
        if keep:
            new_id[n] = out.add_node(g.start_flag[n])
    origin = []
    for n in range(g.num_nodes()):
        if new_id[n] < 0:
            continue
        seen = {n}
        queue = deque([n])
        while queue:
            r = queue.popleft()
            if g.accept_flag[r]:
                out.accept_flag[new_id[n]] = True
            for a in g.out_arcs[r]:
                if match(a):
                    if g.dst[a] not in seen:
                        seen.add(g.dst[a])
                        queue.append(g.dst[a])
                else:
                    out.add_arc(new_id[n], new_id[g.dst[a]], g.ilab[a], g.olab[a], g.w[a])
                    origin.append(a)

    def grad_fn(delta):
        d = np.zeros(g.num_arcs())
        for k, a in enumerate(origin):
            d[a] += delta[k]
        g._add_grad(d)

    out._inputs = [g]
    out._grad_fn = grad_fn
    return out


# ------------------------------------------------------------------------------------------------
# shortest distance
# ------------------------------------------------------------------------------------------------
def _topo_order(g):
    deg = [len(g.in_arcs[n]) for n in range(g.num_nodes())]
    queue = deque(n for n in range(g.num_nodes()) if deg[n] == 0)
    order = []
    while queue:
        n = queue.popleft()
        order.append(n)
        for a in g.out_arcs[n]:
            d = g.dst[a]
            deg[d] -= 1
            if deg[d] == 0:
                queue.append(d)
    if len(order) != g.num_nodes():
        raise ValueError("graph has a cycle: shortest distance needs a DAG")
    return order


def _wt(g, a):
    """Arc weight with the NaN policy: a NaN weight is an impossible arc (-inf).

    Pinned by the reference's tests/gtn_stc_test.py:25-37, which expects loss 0.0 although the
    <star>\\token column holds NaN at a frame where an accepting path crosses it.
    """
    w = g.w[a]
    return NEG_INF if w != w else w


def _lse(vals):
    m = max(vals)
    if m == NEG_INF:
        return NEG_INF
    if m == float("inf"):
        return m
    return m + math.log(sum(math.exp(v - m) for v in vals))


def _forward_scores(g, order):
    alpha = [NEG_INF] * g.num_nodes()
    for n in g.start_nodes():
        alpha[n] = 0.0
    for n in order:
        if g.in_arcs[n]:
            vals = [alpha[g.src[a]] + _wt(g, a) for a in g.in_arcs[n]]
            if g.start_flag[n]:
                vals.append(0.0)
            alpha[n] = _lse(vals)
    return alpha


def _backward_scores(g, order):
    beta = [NEG_INF] * g.num_nodes()
    for n in reversed(order):
        vals = [beta[g.dst[a]] + _wt(g, a) for a in g.out_arcs[n]]
        if g.accept_flag[n]:
            vals.append(0.0)
        if vals:
            beta[n] = _lse(vals)
    return beta


def forward_score(g):
    order = _topo_order(g)
    alpha = _forward_scores(g, order)
    acc = [alpha[n] for n in g.accept_nodes()]
    z = _lse(acc) if acc else NEG_INF
    out = scalar_graph(z, g.calc_grad)

    def grad_fn(delta):
        d = np.zeros(g.num_arcs())
        if z != NEG_INF and z == z:
            beta = _backward_scores(g, order)
            for a in range(g.num_arcs()):
                v = alpha[g.src[a]] + _wt(g, a) + beta[g.dst[a]]
                if v != NEG_INF:
                    d[a] = math.exp(v - z)
        g._add_grad(d * delta[0])

    out._inputs = [g]
    out._grad_fn = grad_fn
    return out


def _viterbi(g):
    """Kahn sweep, strict '>' relaxation (first relaxation in pop/arc order wins)."""
    deg = [len(g.in_arcs[n]) for n in range(g.num_nodes())]
    score = [NEG_INF] * g.num_nodes()
    back = [-1] * g.num_nodes()
    for n in g.start_nodes():
        score[n] = 0.0
    queue = deque(n for n in range(g.num_nodes()) if deg[n] == 0)
    visited = 0
    while queue:
        n = queue.popleft()
        visited += 1
        for a in g.out_arcs[n]:
            d = g.dst[a]
            v = score[n] + _wt(g, a)
            if v > score[d]:
                score[d] = v
                back[d] = a
            deg[d] -= 1
            if deg[d] == 0:
                queue.append(d)
    if visited != g.num_nodes():
        raise ValueError("graph has a cycle: shortest distance needs a DAG")
    best, best_n = NEG_INF, -1
    for n in g.accept_nodes():
        if score[n] > best:
            best, best_n = score[n], n
    return best, best_n, back


def viterbi_score(g):
    best, best_n, back = _viterbi(g)
    out = scalar_graph(best, g.calc_grad)

    def grad_fn(delta):
        d = np.zeros(g.num_arcs())
        n = best_n
        while n >= 0 and back[n] >= 0:
            d[back[n]] = 1.0
            n = g.src[back[n]]
        g._add_grad(d * delta[0])

    out._inputs = [g]
    out._grad_fn = grad_fn
    return out


def viterbi_path(g):
    best, best_n, back = _viterbi(g)
    arcs = []
    n = best_n
    while n >= 0 and back[n] >= 0:
        arcs.append(back[n])
        n = g.src[back[n]]
    arcs.reverse()
    out = Graph(g.calc_grad)
    if best_n >= 0:
        out.add_node(True, not arcs)
        for k, a in enumerate(arcs):
            out.add_node(False, k == len(arcs) - 1)
            out.add_arc(k, k + 1, g.ilab[a], g.olab[a], g.w[a])

    def grad_fn(delta):
        d = np.zeros(g.num_arcs())
        for k, a in enumerate(arcs):
            d[a] += delta[k]
        g._add_grad(d)

    out._inputs = [g]
    out._grad_fn = grad_fn
    return out


# ------------------------------------------------------------------------------------------------
# scalar arithmetic
# ------------------------------------------------------------------------------------------------
def negate(g):
    out = scalar_graph(-g.item(), g.calc_grad)
    out._inputs = [g]
    out._grad_fn = lambda delta: g._add_grad(-delta)
    return out


def add(g1, g2):
    out = scalar_graph(g1.item() + g2.item(), g1.calc_grad or g2.calc_grad)
    out._inputs = [g1, g2]

    def grad_fn(delta):
        g1._add_grad(delta)
        g2._add_grad(delta)

    out._grad_fn = grad_fn
    return out


def subtract(g1, g2):
    out = scalar_graph(g1.item() - g2.item(), g1.calc_grad or g2.calc_grad)
    out._inputs = [g1, g2]

    def grad_fn(delta):
        g1._add_grad(delta)
        g2._add_grad(-delta)

    out._grad_fn = grad_fn
    return out


# ------------------------------------------------------------------------------------------------
# autograd driver
# ------------------------------------------------------------------------------------------------
def backward(g, grad=None, retain_graph=False):
    """backward(g, retain_graph: bool) or backward(g, grad_graph[, retain_graph])."""
    if isinstance(grad, bool) or grad is None:
        seed = np.ones(g.num_arcs(), dtype=np.float64)
    else:
        seed = grad.weights64()
    # topological order of the op DAG (post-order DFS), then reverse
    order, seen = [], set()
    stack = [(g, False)]
    while stack:
        node, done = stack.pop()
        if done:
            order.append(node)
            continue
        if id(node) in seen:
            continue
        seen.add(id(node))
        stack.append((node, True))
        for inp in node._inputs:
            if id(inp) not in seen:
                stack.append((inp, False))
    # intermediate graphs start from a clean gradient; leaves accumulate
    for node in order:
        if node._grad_fn is not None:
            node._grad = None
    g._add_grad(seed)
    for node in reversed(order):
        if node._grad_fn is not None and node._grad is not None:
            node._grad_fn(node._grad)


def parallel_for(fn, iterable):
    for i in iterable:
        fn(i)


# ------------------------------------------------------------------------------------------------
# comparisons and text I/O
# ------------------------------------------------------------------------------------------------
def equal(g1, g2):
    if g1.num_nodes() != g2.num_nodes() or g1.num_arcs() != g2.num_arcs():
        return False
    if g1.start_flag != g2.start_flag or g1.accept_flag != g2.accept_flag:
        return False

    def key(g):
        return sorted(zip(g.src, g.dst, g.ilab, g.olab, g.w))

    return key(g1) == key(g2)


def isomorphic(g1, g2):
    """Same graph up to node renumbering (labels, weights, start/accept must agree)."""
    if g1.num_nodes() != g2.num_nodes() or g1.num_arcs() != g2.num_arcs():
        return False
    if g1.num_start() != g2.num_start() or g1.num_accept() != g2.num_accept():
        return False

    def sig(g, n):
        return (
            g.start_flag[n],
            g.accept_flag[n],
            tuple(sorted((g.ilab[a], g.olab[a], g.w[a], g.dst[a] == n) for a in g.out_arcs[n])),
            tuple(sorted((g.ilab[a], g.olab[a], g.w[a]) for a in g.in_arcs[n])),
        )

    sig1 = [sig(g1, n) for n in range(g1.num_nodes())]
    sig2 = [sig(g2, n) for n in range(g2.num_nodes())]
    if sorted(sig1) != sorted(sig2):
        return False
    mapping, used = {}, set()

    def consistent(a, b):
        # every arc a->x with x already mapped must have a twin b->mapping[x] (multiset equality)
        def arcs(g, n, m, fwd):
            out = []
            for arc in (g.out_arcs[n] if fwd else g.in_arcs[n]):
                other = g.dst[arc] if fwd else g.src[arc]
                if m is None:
                    out.append((other, g.ilab[arc], g.olab[arc], g.w[arc]))
                elif other in m:
                    out.append((m[other], g.ilab[arc], g.olab[arc], g.w[arc]))
            return sorted(out)

        trial = dict(mapping)
        trial[a] = b
        inv = set(trial.values())
        for fwd in (True, False):
            lhs = arcs(g1, a, trial, fwd)
            rhs = [t for t in arcs(g2, b, None, fwd) if t[0] in inv]
            if lhs != rhs:
                return False
        return True

    nodes = sorted(range(g1.num_nodes()), key=lambda n: sig1[n])

    def solve(k):
        if k == len(nodes):
            return True
        a = nodes[k]
        for b in range(g2.num_nodes()):
            if b in used or sig2[b] != sig1[a]:
                continue
            if consistent(a, b):
                mapping[a] = b
                used.add(b)
                if solve(k + 1):
                    return True
                del mapping[a]
                used.discard(b)
        return False

    return solve(0)


def loadtxt(path):
    """Text format pinned by tests/trans_backoff_test.txt: start ids / accept ids / arcs."""
    with open(path) as fid:
        lines = [ln.strip() for ln in fid if ln.strip()]
    starts = [int(v) for v in lines[0].split()]
    accepts = [int(v) for v in lines[1].split()]
    arcs = []
    for ln in lines[2:]:
        parts = ln.split()
        s, d, il = int(parts[0]), int(parts[1]), int(parts[2])
        ol = int(parts[3]) if len(parts) > 3 else il
        w = float(parts[4]) if len(parts) > 4 else 0.0
        arcs.append((s, d, il, ol, w))
    n_nodes = 1 + max([*starts, *accepts, *[a[0] for a in arcs], *[a[1] for a in arcs]])
    g = Graph()
    ss, acc = set(starts), set(accepts)
    for n in range(n_nodes):
        g.add_node(n in ss, n in acc)
    for a in arcs:
        g.add_arc(*a)
    return g


def savetxt(path, g):
    with open(path, "w") as fid:
        fid.write(repr(g) + "\n")


load = loadtxt
save = savetxt
