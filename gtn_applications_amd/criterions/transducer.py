"""Transducer criterion -- counterpart of /root/reference/criterions/transducer.py.

Graph factories (`make_*_graph`), the `Transducer` module and `TransducerLossFunction` keep the
reference's names, arguments and error behaviour (transducer.py:15-348).  The small per-utterance
graph algebra (target o lexicon, tokens o decompositions, transitions o alignments;
transducer.py:265-281) runs in the C++ host library; everything that touches the [B,T,C] emissions
-- `intersect(emissions, .)`, `forward_score`, `viterbi_path`, `backward`
(transducer.py:283-288,321-336,216-221) -- runs on the lattice engine kernels.
"""
import ctypes
import itertools
import os
import threading

import math

import numpy as np
import torch

from .. import _native as N
from .. import engine as E
from .. import graph as G


def make_scalar_graph(weight):
    """transducer.py:15-20."""
    scalar = G.Graph()
    scalar.add_node(True)
    scalar.add_node(False, True)
    scalar.add_arc(0, 1, 0, 0, weight)
    return scalar


def make_chain_graph(sequence):
    """transducer.py:23-29."""
    seq = [int(s) for s in sequence]
    graph = G.Graph(False)
    n = len(seq)
    graph.add_nodes([1] + [0] * n, [0] * n + [1] if n else [0])
    if n:
        idx = np.arange(n, dtype=np.int32)
        graph.add_arcs(idx, idx + 1, np.asarray(seq, dtype=np.int32))
    return graph


def make_transitions_graph(ngram, num_tokens, calc_grad=False):
    """transducer.py:32-58: dense n-gram transition model; </s> is reached by epsilon arcs."""
    transitions = G.Graph(calc_grad)
    transitions.add_node(True, ngram == 1)
    state_map = {(): 0}
    for n in range(1, ngram):  # histories that still include <s>
        for state in itertools.product(range(num_tokens), repeat=n):
            state_map[state] = transitions.add_node(False, ngram == 1)
            transitions.add_arc(state_map[state[:-1]], state_map[state], state[-1])
    src, dst, lab = [], [], []
    for state in itertools.product(range(num_tokens), repeat=ngram):
        src.append(state_map[state[:-1]]), dst.append(state_map[state[1:]]), lab.append(state[-1])
    transitions.add_arcs(src, dst, lab)
    if ngram > 1:
        end_idx = transitions.add_node(False, True)
        idx = np.arange(end_idx, dtype=np.int32)
        transitions.add_arcs(idx, np.full(end_idx, end_idx, np.int32), np.full(end_idx, G.epsilon, np.int32))
    return transitions


def make_lexicon_graph(word_pieces, graphemes_to_idx):
    """transducer.py:61-75: letters -> word pieces."""
    graph = G.Graph(False)
    graph.add_node(True, True)
    for i, wp in enumerate(word_pieces):
        prev = 0
        for letter in wp[:-1]:
            n = graph.add_node()
            graph.add_arc(prev, n, graphemes_to_idx[letter], G.epsilon)
            prev = n
        graph.add_arc(prev, 0, graphemes_to_idx[wp[-1]], i)
    graph.arc_sort()
    return graph


def make_token_graph(token_list, blank="none", allow_repeats=True):
    """transducer.py:78-123: emission labels -> tokens (one or more frames per token)."""
    if not allow_repeats and blank != "optional":
        raise ValueError("Must use blank='optional' if disallowing repeats.")
    ntoks = len(token_list)
    graph = G.Graph(False)
    graph.add_node(True, True)
    graph.add_nodes([0] * ntoks, [int(blank != "forced")] * ntoks)
    if blank != "none":
        graph.add_node()
        graph.add_arc(0, ntoks + 1, ntoks, G.epsilon)  # blank index is assumed to be last (ntoks)
        graph.add_arc(ntoks + 1, 0, G.epsilon)
    entry = (ntoks + 1) if blank == "forced" else 0
    arcs = []  # (src, dst, ilabel, olabel) in the reference's insertion order
    for i in range(ntoks):
        arcs.append((entry, i + 1, i, i))
        arcs.append((i + 1, i + 1, i, G.epsilon))
        if allow_repeats:
            if blank == "forced":  # token -> blank only
                arcs.append((i + 1, ntoks + 1, ntoks, G.epsilon))
            else:  # token -> blank and every token
                arcs.append((i + 1, 0, G.epsilon, G.epsilon))
        else:  # token -> blank and every OTHER token
            arcs.append((i + 1, ntoks + 1, ntoks, G.epsilon))
            arcs.extend((i + 1, j + 1, j, j) for j in range(ntoks) if j != i)
    src, dst, il, ol = (np.fromiter((a[k] for a in arcs), dtype=np.int32, count=len(arcs)) for k in range(4))
    graph.add_arcs(src, dst, il, ol)
    return graph


def make_kernel_graph(x, blank_idx, blank_optional, spike=False, calc_grad=False):
    """transducer.py:351-367."""
    g = G.Graph(calc_grad)
    g.add_node(True, len(x) == 0)  # start in blank
    g.add_arc(0, 0, blank_idx)
    for i, c in enumerate(x):
        last = (i + 1) == len(x)
        g.add_node(False, blank_optional and last)
        g.add_node(False, last)
        g.add_arc(2 * i, 2 * i + 1, c)
        if not spike:
            g.add_arc(2 * i + 1, 2 * i + 1, c)
        g.add_arc(2 * i + 1, 2 * i + 2, blank_idx)
        g.add_arc(2 * i + 2, 2 * i + 2, blank_idx)
        if i > 0 and blank_optional and x[i - 1] != c:
            g.add_arc(2 * i - 1, 2 * i + 1, c)
    g.arc_sort(True)
    g.arc_sort()
    return g


# alignment graphs per (tokens, lexicon, transitions, target) and packed batches: content-keyed LRUs
_ALIGN_CACHE = E.LRU(4096)
_PACK_CACHE = E.LRU(32)
# One batch is packed at a time (the cache, the staging ring and the packer's host pool are shared between the caller's
# thread and the prefetch thread of Transducer.prepare)
_PACK_LOCK = threading.RLock()
_PREP = {"pool": None, "streams": {}}


class PreparedTargets:
    """A batch of targets whose alignment acceptors are being (or have been) built, packed and uploaded ahead of the
    step that uses them: what `Transducer.prepare(targets)` returns and `Transducer.forward(inputs, prepared)` accepts
    in the place of the target list.  The per-batch host work of the criterion (the graph algebra of
    transducer.py:262-281 for every utterance: ~0.3 ms at the benchmark's size) then runs on a side thread while the
    previous step is on the GPU -- what a DataLoader's prefetch does for the inputs."""

    __slots__ = ("targets", "future", "B")

    def __init__(self, targets, future, B):
        self.targets, self.future, self.B = targets, future, B

    def result(self):
        return self.future.result()

    def __len__(self):
        return self.B


def _pack_entry(targets, tokens, lexicon, transitions, C, dev, reduction):
    """(cache key, flat labels info) of a batch and its cache entry (built if it is not there): the per-sample graph
    algebra of transducer.py:262-281 for the whole batch -- one native call, threaded over the utterances like the
    reference's gtn.parallel_for (transducer.py:296) -- and the asynchronous upload on the CURRENT stream."""
    flat, offsets, lens = E.flatten_any(targets)
    B = len(lens)
    key = ("num", flat.tobytes(), tuple(lens), id(tokens), id(lexicon), id(transitions), C, dev.index)
    full_key = key + (reduction == "mean",)
    with _PACK_LOCK:
        if full_key not in _PACK_CACHE.data:
            N.lib.wfl_host_pool_wake()  # (new targets: the packer's threads are awake by the time its job is submitted)

        def build():
            if reduction == "mean":  # transducer.py:302-305: normalise by the (grapheme) target length
                sc = np.array([1.0 / n if n > 0 else 1.0 for n in lens], dtype=np.float32)
            else:
                sc = np.ones(B, dtype=np.float32)
            # (loss scale, +scale/B, -scale/B) travel with the packed batch: one asynchronous upload, no kernels
            pack = E.PackedLattice.transducer_batch(tokens, lexicon, transitions, flat, offsets, C, dev,
                                                    extra=np.concatenate([sc, sc / B, -sc / B]))
            fac = pack.extra
            return pack, fac[:B], fac[B:2 * B], fac[2 * B:], (tokens, lexicon, transitions)

        entry = _PACK_CACHE.get(full_key, build)
    return B, entry


def _prepare(targets, tokens, lexicon, transitions, C, dev, reduction):
    """_pack_entry on the prefetch thread, its upload on a stream of its own (the step that uses the pack orders itself
    behind the pack's event)."""
    if _PREP["pool"] is None:
        import concurrent.futures

        _PREP["pool"] = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="wfl-prepare")
    stream = _PREP["streams"].get(dev.index)
    if stream is None:
        stream = _PREP["streams"][dev.index] = torch.cuda.Stream(device=dev)

    def job():
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            return _pack_entry(targets, tokens, lexicon, transitions, C, dev, reduction)

    return _PREP["pool"].submit(job)


def _zero_weight_view(graph):
    """Structure of `graph` with all arc weights 0: learnable weights are added on the device."""
    a = graph.arrays()
    g = G.Graph(False)
    g.add_nodes(a["start"], a["accept"])
    g.add_arcs(a["src"], a["dst"], a["ilabel"], a["olabel"])
    return g


def _alignment_graph(target, tokens, lexicon, transitions, direct=False):
    """transducer.py:265-281: every frame-level alignment of every decomposition of `target` into
    tokens, optionally intersected with the transition model.  Returns (graph, weight ids).
    direct=True: the alignments written down without the composition where the token graph allows it
    (G.token_alignments -- what the native batch packer does; isomorphic, not identical)."""
    key = (tuple(target), id(tokens), id(lexicon), id(transitions), bool(direct))

    def build():
        tgt = make_chain_graph(target)
        tokens_target = G.remove(G.project_output(G.compose(tgt, lexicon)))
        ali = G.token_alignments(tokens, tokens_target) if direct else None
        if ali is None:
            ali = G.project_input(G.remove(G.compose(tokens, tokens_target)))
        if transitions is None:
            return ali, None, (tokens, lexicon)
        ali, from_trans, _ = G.compose(transitions, ali, provenance=True)
        return ali, from_trans, (tokens, lexicon, transitions)  # keep operands alive: ids stay unique

    return _ALIGN_CACHE.get(key, build)[:2]


# WFL_DENSE_NGRAM=0: the bigram normaliser through the general lattice sweep (A/B tests, measurements)
_DENSE_NGRAM = os.environ.get("WFL_DENSE_NGRAM", "1") != "0"


class Transducer(torch.nn.Module):
    """A generic transducer loss (transducer.py:126-234).

    tokens: list of iterables (strings, tuples, ...) -- the model's output units.
    graphemes_to_idx: grapheme -> integer index.
    ngram: order of the learned token-level transition model (0: none).
    transitions: alternatively a ready transition graph (its arcs become `transition_params`).
    blank: 'none' | 'optional' | 'forced'.   allow_repeats: allow the same token twice in a row.
    """

    def __init__(self, tokens, graphemes_to_idx, ngram=0, transitions=None, blank="none",
                 allow_repeats=True, reduction="none"):
        super(Transducer, self).__init__()
        if blank not in ["optional", "forced", "none"]:
            raise ValueError("Invalid value specificed for blank. Must be in ['optional', 'forced', 'none']")
        self.tokens = make_token_graph(tokens, blank=blank, allow_repeats=allow_repeats)
        self.lexicon = make_lexicon_graph(tokens, graphemes_to_idx)
        self._num_emission_classes = len(tokens) + int(blank != "none")  # (width of the emissions: prepare())
        self.ngram = ngram
        if ngram > 0 and transitions is not None:
            raise ValueError("Only one of ngram and transitions may be specified")
        if ngram > 0:
            transitions = make_transitions_graph(ngram, len(tokens) + int(blank != "none"), True)
        if transitions is not None:
            # the arc weights of the graph are replaced by the parameters on every call
            # (transducer.py:255-256), so only its structure matters
            self.transitions = _zero_weight_view(transitions)
            self.transitions.arc_sort()
            self.transition_params = torch.nn.Parameter(torch.zeros(self.transitions.num_arcs()))
        else:
            self.transitions = None
            self.transition_params = None
        self.reduction = reduction

    def prepare(self, targets, device=None):
        """Start building, packing and uploading the alignment acceptors of a batch the criterion will see later -- on a
        side thread and a stream of its own; returns at once.  Hand the result to forward() in the place of `targets`:

            nxt = criterion.prepare(next_targets)        # e.g. right after the step's launches, or in the loader's collate
            loss = criterion(emissions, cur); loss.backward(); ...; cur = nxt

        (reference: transducer.py:262-281 runs this per step, inside forward.)  `device`: where the emissions will
        live (default: the current device)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.tokens.arc_sort(True)
        return PreparedTargets(targets, _prepare(targets, self.tokens, self.lexicon,
                                                 _numerator_transitions(self.transitions, self._num_emission_classes),
                                                 self._num_emission_classes, dev, self.reduction), len(targets))

    @E.on_input_device
    def forward(self, inputs, targets):
        self.tokens.arc_sort(True)
        if self.transitions is None:
            # transducer.py:186-187 applies log_softmax here; the engine fuses it into the gather
            # (forward) and the gradient rows (backward) instead of materialising [B,T,C] twice
            return E.make_eager(_FusedLogSoftmaxTransducerLoss.apply(inputs, targets, self.tokens, self.lexicon, None,
                                                                     None, self.reduction))
        return TransducerLoss(inputs, targets, self.tokens, self.lexicon, self.transition_params,
                              self.transitions, self.reduction)

    def viterbi(self, outputs):
        """transducer.py:199-234: best frame-level path (under the transition model if any), then
        the shortest token sequence that path can transduce to."""
        B, T, C = outputs.shape
        dev = E.require_gpu()
        x = E.as_device_f32(outputs.detach(), dev)
        labels = None
        if self.transitions is not None:
            params = E.as_device_f32(self.transition_params.detach(), dev)
            if _DENSE_NGRAM and params.numel() == C and _dense_unigram(self.transitions, C):
                # one node, one self-loop per token: the best path takes, frame by frame, the first maximum of x + p
                labels = E.row_argmax(x + params[:C]).cpu().numpy().reshape(-1)
                offsets = np.arange(B + 1, dtype=np.int64) * T
            elif _DENSE_NGRAM and _dense_bigram(self.transitions, C):
                # the fully connected recursion of the dense engine in the max-plus semiring (as the normaliser of the
                # loss takes its log-semiring one): its best state per frame IS the label path, no epsilon to remove
                xd, Wd = _bigram_dense_operands(x, params, C)
                labels = E.dense_viterbi(xd, Wd).cpu().numpy().reshape(-1)
                offsets = np.arange(B + 1, dtype=np.int64) * T
        if labels is not None:
            pass
        elif self.transitions is not None:
            pack = _transitions_pack(self.transitions, B, C, dev)
            arc_paths, _ = E.lattice_viterbi(x, pack, weights=params)
            olab = self.transitions.arrays()["olabel"]
            # gtn.remove of the back-off arcs (transducer.py:218), for the whole batch at once
            lens = np.fromiter((0 if p is None else len(p) for p in arc_paths), np.int64, B)
            arcs = np.concatenate([p for p in arc_paths if p is not None and len(p)] or [np.zeros(0, np.int32)])
            labs = olab[arcs]
            keep = labs != G.epsilon
            offsets = np.zeros(B + 1, np.int64)
            np.cumsum(np.bincount(np.repeat(np.arange(B), lens)[keep], minlength=B), out=offsets[1:])
            labels = np.ascontiguousarray(labs[keep], dtype=np.int32)
        else:
            # viterbi_path of the bare emissions graph: per frame, the first maximal label
            labels = E.row_argmax(x).cpu().numpy().reshape(-1)
            offsets = np.arange(B + 1, dtype=np.int64) * T
        self.tokens.arc_sort()
        # one native call for the batch (the reference's gtn.parallel_for over process(b), transducer.py:232);
        # ambiguous decodings: the shortest wins (transducer.py:226-228)
        out, out_off = G.transducer_decode_batch(self.tokens, labels, offsets)
        flat = torch.from_numpy(out)  # (int32: torch.IntTensor, as transducer.py:233)
        return list(torch.split(flat, np.diff(out_off).tolist()))  # (views of one tensor: one call instead of B slices + clones)


_BIGRAM_SEEN = {}


def _dense_bigram(transitions, C):
    """True iff `transitions` has exactly the structure of make_transitions_graph(2, C) (transducer.py:32-58): start
    node 0, one node per previous token, C start arcs, C x C bigram arcs `prev a -> cur b` at arc C + a C + b, and
    an epsilon arc from every node to the accepting end node.  Its normaliser
    forward_score(intersect(emissions, transitions)) (transducer.py:286-288) is then the fully connected recursion of
    the dense engine (csrc/dense_kernels.hip) with W[0] = start scores, W[1+b][a] = bigram score and the end
    arcs' scores added to the last frame -- instead of the general lattice sweep with (C+1) C^2 arcs."""
    hit = _BIGRAM_SEEN.get(id(transitions))
    if hit is not None and hit[0] is transitions and hit[1] == C:
        return hit[2]
    ok = False
    if transitions.num_nodes() == C + 2 and transitions.num_arcs() == C + C * C + C + 1:
        a = transitions.arrays()
        idx = np.arange(C, dtype=np.int64)
        ab = np.arange(C * C, dtype=np.int64)
        eps = np.arange(C + 1, dtype=np.int64)
        s, d, il, ol = (np.asarray(a[k]) for k in ("src", "dst", "ilabel", "olabel"))
        n0, n1 = C, C + C * C
        st, ac = np.flatnonzero(np.asarray(a["start"])), np.flatnonzero(np.asarray(a["accept"]))  # (per-node flags)
        ok = (st.tolist() == [0] and ac.tolist() == [C + 1]
              and (s[:n0] == 0).all() and (d[:n0] == 1 + idx).all() and (il[:n0] == idx).all() and (ol[:n0] == idx).all()
              and (s[n0:n1] == 1 + ab // C).all() and (d[n0:n1] == 1 + ab % C).all() and (il[n0:n1] == ab % C).all()
              and (ol[n0:n1] == ab % C).all()
              and (s[n1:] == eps).all() and (d[n1:] == C + 1).all() and (il[n1:] == G.epsilon).all() and (ol[n1:] == G.epsilon).all())
    if len(_BIGRAM_SEEN) > 64:
        _BIGRAM_SEEN.clear()
    _BIGRAM_SEEN[id(transitions)] = (transitions, C, bool(ok))  # (holds the graph: its id stays unique)
    return bool(ok)


_NUM_TRANSITIONS = {}


def _numerator_transitions(transitions, C):
    """The transition graph the NUMERATOR's alignments are intersected with when the Transducer carries the dense
    bigram model (_bigram_route).  make_transitions_graph(2, C) reaches its accepting node through one epsilon arc per
    history (transducer.py:32-58), so every alignment acceptor ends in an epsilon arc -- and an acceptor with an epsilon
    arc is swept in the log domain by the general kernels (0.37 + 0.22 ms at the n-gram benchmark's shape against
    0.05 + 0.07 for the same acceptor without it).  That arc's score depends on the LAST emitted label only (node 1 + c
    after label c), so it is the dense normaliser's trick again: the end arcs' scores ride on the last frame's emissions
    and the graph loses its epsilon arcs -- every node accepts instead.  The remaining arcs are laid out like the dense
    engine's matrix W [(C+1), C] row-major (include/wfl.h): arc c = start -> c, arc (1 + b) C + a = bigram a -> b, so that
    numerator and normaliser read their weights from ONE tensor, as ASG's do.  Anything but the dense bigram: the graph
    itself."""
    if transitions is not None and _DENSE_NGRAM and transitions.num_nodes() == 1 and _dense_unigram(transitions, C):
        return None  # (the unigram model's scores are added to the emissions: _unigram_route)
    if transitions is None or not (_DENSE_NGRAM and _dense_bigram(transitions, C)):
        return transitions
    hit = _NUM_TRANSITIONS.get(id(transitions))
    if hit is not None and hit[0] is transitions:
        return hit[1]
    a = transitions.arrays()
    idx = np.arange(C, dtype=np.int32)
    prev = np.tile(idx, C)    # a: fastest
    cur = np.repeat(idx, C)   # b
    src = np.concatenate([np.zeros(C, np.int32), 1 + prev])
    dst = np.concatenate([1 + idx, 1 + cur])
    lab = np.concatenate([idx, cur])
    g = G.Graph(False)
    g.add_nodes(np.asarray(a["start"]), np.ones(C + 2, dtype=np.asarray(a["accept"]).dtype))
    g.add_arcs(src, dst, lab, lab)
    if len(_NUM_TRANSITIONS) > 64:
        _NUM_TRANSITIONS.clear()
    _NUM_TRANSITIONS[id(transitions)] = (transitions, g)
    return g


def _numerator_entry(targets, tokens, lexicon, graph, C, dev, reduction, B):
    """The packed alignment acceptors of a batch (their cache entry): from a PreparedTargets handle if it was made for
    this criterion and device, else packed here."""
    if isinstance(targets, PreparedTargets):
        nb, entry = targets.result()
        pack = entry[0]
        if pack.desc.B != B or pack.device != dev or entry[4] != (tokens, lexicon, graph):
            # prepared for other emissions (another device, another criterion): pack again, here
            nb, entry = _pack_entry(targets.targets, tokens, lexicon, graph, C, dev, reduction)
    else:
        nb, entry = _pack_entry(targets, tokens, lexicon, graph, C, dev, reduction)
    if nb != B:
        raise ValueError(f"got {nb} targets for a batch of {B}")
    pack = entry[0]
    up = getattr(pack, "_uploaded", None)
    if up is not None and up[0] != E.stream_ptr() and not getattr(pack, "_seen_here", None) == E.stream_ptr():
        # uploaded on another stream (the prefetch thread's, or an earlier step's): its memory stays this stream's too
        pack._blob.record_stream(torch.cuda.current_stream())
        pack._seen_here = E.stream_ptr()
    return entry


def _bigram_route(inputs, targets, tokens, lexicon, transition_params=None, transitions=None, reduction="none"):
    """TransducerLoss with the dense bigram model as the ASG step it is (None: not that case).  Emissions with the end
    arcs' scores on the last frame and the matrix W of the dense engine are formed from `transition_params` =
    [start C | bigram a -> b at C + a C + b | end arcs of nodes 0 .. C] by differentiable torch ops -- autograd maps the
    two gradients back -- and ASGLoss (one native call: csrc/torch_ops.cpp::asg_forward) does the rest."""
    if transitions is None or transition_params is None or inputs.dim() != 3:
        return None
    B, T, C = inputs.shape
    if T == 0 or not (_DENSE_NGRAM and _dense_bigram(transitions, C)) or transition_params.numel() != C + C * C + C + 1:
        return None
    from . import asg as _asg

    dev = E.require_gpu()
    with torch.cuda.device(dev):
        pack, scale, cpos, cneg, _ = _numerator_entry(targets, tokens, lexicon, _numerator_transitions(transitions, C), C, dev,
                                                       reduction, B)
    return _BigramAsAsg.apply(inputs, transition_params, _asg.PackedNumerator(pack, scale, cpos, cneg, B))


class _InnerCtx:
    """What ASGLossFunction's forward / backward keep on their autograd context, for a caller that runs them inside a
    node of its own (_BigramAsAsg): plain attributes, and the two hook registrars E.watch_node_hooks looks for."""

    needs_input_grad = (True, True, False, False)

    def register_hook(self, fn):
        return None

    def register_prehook(self, fn):
        return None


class _BigramAsAsg(torch.autograd.Function):
    """_bigram_route's step as ONE autograd node: the operands of the ASG step from (emissions, transition_params) --
    three small torch ops -- ASGLossFunction's forward on them (csrc/torch_ops.cpp::asg_forward), and in backward its
    two gradients mapped back to the caller's tensors.  (Spelled with differentiable torch ops around ASGLoss the step
    carried eight more autograd nodes: ~0.1 ms of host time where the step is host-bound.)"""

    @staticmethod
    @E.on_input_device
    def forward(ctx, inputs, transition_params, packed):
        from . import asg as _asg

        B, T, C = inputs.shape
        dev = E.require_gpu()
        x = E.as_device_f32(inputs.detach(), dev)
        p = E.as_device_f32(transition_params.detach(), dev).reshape(-1)
        n1 = C + C * C
        Wd = torch.cat([p[:C].view(1, C), p[C:n1].view(C, C).t()], dim=0)
        xd = x.clone()
        xd[:, -1, :] += p[n1 + 1:]
        # (what ASGLossFunction.forward asks its operands: which gradients the step will be asked for -- the end arcs'
        # gradient is the last frame's rows of the emission gradient, so the parameters alone ask for that one too)
        xd.requires_grad_(inputs.requires_grad or transition_params.requires_grad)
        Wd.requires_grad_(transition_params.requires_grad)
        ctx.bigram = (C, inputs.device, transition_params.device, transition_params.shape)
        ctx.inner = _InnerCtx()
        loss = _asg.ASGLossFunction.forward(ctx.inner, xd, Wd, packed, "none")
        return loss if inputs.is_cuda else loss.cpu()

    @staticmethod
    @E.on_input_device
    def backward(ctx, grad_output):
        from . import asg as _asg

        C, in_dev, par_dev, par_shape = ctx.bigram
        need_x, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ctx.inner.needs_input_grad = (need_x or need_p, need_p, False, False)
        dxd, dWd, _, _ = _asg.ASGLossFunction.backward(ctx.inner, grad_output)
        dp = None
        if need_p:
            # the end arcs' scores rode on the last frame's emissions: their gradient is that frame's rows, summed
            # (the emission gradient is computed for it even when the caller's emissions ask for none)
            end = dxd[:, -1, :].sum(dim=0)
            dp = torch.cat([dWd[0], dWd[1:].t().reshape(-1), end.new_zeros(1), end]).reshape(par_shape)
            if par_dev.type != "cuda":
                dp = dp.to(par_dev)
        dx = None
        if need_x:
            dx = dxd if in_dev.type == "cuda" else dxd.to(in_dev)
        return dx, dp, None


class _UnigramNormaliser:
    """forward_score(intersect(emissions, transitions)) (transducer.py:286-288) for make_transitions_graph(1, C): one
    start + accept node with a self-loop per token, so  log Z_b = sum_t logsumexp_c (x[b,t,c] + p_c)."""

    __slots__ = ("xp", "lse", "logz")

    def __init__(self, x, params):
        # NaN policy of the lattice path this replaces: a NaN score is an impossible arc (-inf)
        self.xp = torch.nan_to_num(x + params[:x.shape[2]], nan=float("-inf"), posinf=float("inf"), neginf=float("-inf"))
        self.lse = E.row_lse(self.xp)
        self.logz = self.lse.sum(dim=1)

    def posteriors(self):
        # (a frame whose scores are all -inf has no path through it: zero posteriors, as the lattice path gives)
        lse = self.lse.unsqueeze(2)
        return torch.where(torch.isfinite(lse), torch.exp(self.xp - lse), torch.zeros_like(self.xp))


def _dense_unigram(transitions, C):
    """True iff `transitions` is make_transitions_graph(1, C) (transducer.py:32-58 with ngram = 1): one start + accept
    node with a self-loop per token, arc i labelled i."""
    if transitions.num_nodes() != 1 or transitions.num_arcs() != C:
        return False
    a = transitions.arrays()
    idx = np.arange(C)
    return bool(a["start"][0] and a["accept"][0] and (a["src"] == 0).all() and (a["dst"] == 0).all()
                and (a["ilabel"] == idx).all() and (a["olabel"] == idx).all())


def _bigram_dense_operands(x, params, C):
    """(emissions with the end arcs' scores on the last frame, W [(C+1), C] of the dense engine) for the bigram
    transition model: params = [start C | bigram a -> b at a C + b | end arcs of nodes 0 .. C]"""
    Wd = torch.empty((C + 1, C), dtype=torch.float32, device=x.device)
    Wd[0] = params[:C]
    Wd[1:] = params[C:C + C * C].view(C, C).t()
    xd = x.clone()
    xd[:, -1, :] += params[C + C * C + 1:]
    return xd, Wd


def _transitions_pack(transitions, B, C, device):
    key = ("den", id(transitions), B, C, device.index)

    def build():
        wid = np.arange(transitions.num_arcs(), dtype=np.int32)
        return E.PackedLattice.from_graphs([transitions], C, device, wids=[wid], B=B, shared=True), transitions

    return _PACK_CACHE.get(key, build)[0]


_NODE = False
_PHASES = ("lattice_gather", "lattice_chain", "lattice_grad")


def _native_node():
    """csrc/torch_ops.cpp (the step's launches in one native call), or None if the extension was not built /
    WFL_TRANSDUCER_NATIVE=0 (A/B, tests: the Python spelling of the same sequence)."""
    global _NODE
    if _NODE is False:
        _NODE = None
        if os.environ.get("WFL_TRANSDUCER_NATIVE", "1") != "0":
            try:
                from .. import _wfl_torch as mod
                _NODE = mod if hasattr(mod, "lattice_loss_forward") else None
            except ImportError:
                pass
    return _NODE


class TransducerLossFunction(torch.autograd.Function):
    @staticmethod
    @E.on_input_device
    def forward(ctx, inputs, targets, tokens, lexicon, transition_params=None, transitions=None,
                reduction="none"):
        return TransducerLossFunction._forward(ctx, False, inputs, targets, tokens, lexicon, transition_params,
                                               transitions, reduction)

    @staticmethod
    def _forward(ctx, log_softmax, inputs, targets, tokens, lexicon, transition_params, transitions, reduction):
        B, T, C = inputs.shape
        if transitions is not None and transition_params is None:
            raise ValueError("Specified transitions, but not transition params.")
        if T == 0:
            raise ValueError("TransducerLoss: empty emissions (T == 0)")
        dev = E.require_gpu()
        x = E.as_device_f32(inputs.detach(), dev)
        params = E.as_device_f32(transition_params.detach(), dev) if transitions is not None else None
        pack, scale, cpos, cneg, _ = _numerator_entry(targets, tokens, lexicon, transitions, C, dev, reduction, B)
        need_grad = inputs.requires_grad or (transition_params is not None and transition_params.requires_grad)
        den = dense = None
        node = _native_node() if transitions is None else None
        timed = node is not None and E.phase_due(_PHASES)  # (a step whose launch groups bench.py brackets with events)
        if node is not None and not timed and not torch.cuda.is_current_stream_capturing():
            # gather, sweeps (with the gradient beside them), loss reduction and join in one native call
            # (csrc/torch_ops.cpp::lattice_loss_forward): the sequence below, without the interpreter between the launches
            up = getattr(pack, "_uploaded", None)
            if up is not None and up[0] != E.stream_ptr():  # a cached pack uploaded on another stream
                torch.cuda.current_stream().wait_event(up[1])
            want_dx = inputs.requires_grad and _IN_LAUNCH_GRAD
            (loss, xg, al, be, lz, lse, dx_early), in_launch = node.lattice_loss_forward(
                x, ctypes.addressof(pack.desc), pack.ints, pack.floats, scale, cneg, bool(log_softmax), want_dx, need_grad)
            num = E.LatticeState()
            num.pack, num.T, num.C, num.weights, num.bptr = pack, T, C, None, None
            num.xg, num.alpha, num.beta, num.logz = xg, al, be, lz
            num.x, num.row_lse, num.in_launch = (x if log_softmax else None), lse, in_launch
            ctx.aux = (x, params, num, None, cpos, cneg, None)
            ctx.early = None
            ctx.devices = (inputs.device, None)
            if in_launch:
                ctx.early = _EarlyGrad(dx_early, num, cneg, inputs)
                ctx.eager_take = ctx.early.take if E.may_hand_over(inputs) else None
                if ctx.eager_take is not None:
                    E.watch_node_hooks(ctx)
            return loss if inputs.is_cuda else loss.cpu()
        if transitions is not None:  # normaliser: forward_score(emissions o transitions), transducer.py:286-288
            # independent of the numerator sweep: forked onto a second stream so that the two overlap
            with E.side_stream(dev) as fork:
                if _DENSE_NGRAM and params.numel() == C and _dense_unigram(transitions, C):
                    # one state, one self-loop per token: the frames are independent and the normaliser is a sum of
                    # row log-sum-exps of x + p -- no sweep at all
                    den = _UnigramNormaliser(x, params)
                    dense = "unigram"
                elif _DENSE_NGRAM and _dense_bigram(transitions, C):
                    xd, Wd = _bigram_dense_operands(x, params, C)
                    den = E.dense_forward(xd, Wd, need_beta=need_grad)
                    dense = (xd, Wd)
                else:
                    den = E.lattice_forward(x, _transitions_pack(transitions, B, C, dev), weights=params,
                                            need_beta=need_grad)
        # without a transition model the emission gradient is the whole backward pass: the sweeps' launch computes it
        # for grad_output = 1 as it goes (E.lattice_forward grad_into), backward scales it and patches what is left
        dx_early = None
        if transitions is None and inputs.requires_grad and _IN_LAUNCH_GRAD:
            dx_early = torch.empty_like(x)
        E._PHASE_FORCE = timed
        try:
            num = E.lattice_forward(x, pack, weights=params, need_beta=need_grad, log_softmax=log_softmax,
                                    grad_into=(cneg, dx_early) if dx_early is not None else None, defer_join=True)
        finally:
            E._PHASE_FORCE = False
        if not num.in_launch:
            dx_early = None
        if den is not None:
            if dense == "unigram":
                fork.join(den.xp, den.lse, den.logz)
            elif dense is not None:
                fork.join(dense[0], dense[1], den.alpha, den.beta, den.logz, den.ws)
            else:
                fork.join(den.xg, den.alpha, den.beta, den.logz)
            loss = E.reduce_loss(den.logz, scale, 1.0, minus=num.logz)
        else:
            loss = E.reduce_loss(num.logz, scale, -1.0)
        if num.in_launch:
            E.lattice_side_join()  # (behind the loss reduction: it ran under the tail of the gradient beside the sweeps)
        ctx.aux = (x, params, num, den, cpos, cneg, dense)
        ctx.early = None
        ctx.devices = (inputs.device, None if transition_params is None else transition_params.device)
        if dx_early is not None:
            ctx.early = _EarlyGrad(dx_early, num, cneg, inputs)  # (holds no reference to ctx: no cycle to collect)
            ctx.eager_take = ctx.early.take if E.may_hand_over(inputs) else None
            if ctx.eager_take is not None:
                E.watch_node_hooks(ctx)
        return loss if inputs.is_cuda else loss.cpu()

    @staticmethod
    @E.on_input_device
    def backward(ctx, grad_output):
        E.check_not_released(ctx)
        if ctx.early is not None and ctx.early.dx is not None:
            # the launch of the sweeps wrote the gradient for grad_output = 1: scaled in place (wfl_scale returns at once
            # when grad_output is 1), then the rows of the utterances that launch did not serve.  A second pass over a
            # retained graph finds the buffer gone and recomputes below.
            dx, ctx.early.dx = ctx.early.dx, None
            gout = E.as_device_f32(grad_output.detach().reshape(1), dx.device)
            E.scale_inplace(dx, gout)
            E.lattice_grad_rest(ctx.early.num, ctx.early.cneg, gout, dx)
            return (dx if ctx.devices[0].type == "cuda" else dx.to(ctx.devices[0])), None, None, None, None, None, None
        x, params, num, den, cpos, cneg, dense = ctx.aux
        gout = E.as_device_f32(grad_output.detach().reshape(1), x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.zeros_like(params) if (params is not None and ctx.needs_input_grad[4]) else None
        if (dx is not None or dW is not None) and dense == "unigram":
            # posterior of label c in frame t under the unigram model: softmax(x + p), whatever the other frames do
            post = den.posteriors() * (cpos * gout).view(-1, 1, 1)
            if dW is not None:
                dW += post.sum(dim=(0, 1))
            if dx is not None:
                dx.copy_(post)
            E.lattice_grad(num, cneg, coef_w=cneg, gout=gout, dx=dx, accumulate=True, dW=dW)
        elif (dx is not None or dW is not None) and dense is not None:
            # the dense normaliser first (its emission gradient alone, for a moment, in dx: the last frame's rows are
            # also the gradient of the end arcs' scores), then the numerator's lattice on top
            xd, Wd = dense
            C = x.shape[2]
            ddx = dx if dx is not None else torch.empty_like(x)
            dWd = torch.empty_like(Wd) if dW is not None else None
            E.dense_grad(xd, Wd, den, cpos, coef_w=cpos, gout=gout, dx=ddx, accumulate=False, dW=dWd)
            if dW is not None:
                dW[:C] = dWd[0]
                dW[C:C + C * C] = dWd[1:].t().reshape(-1)
            if dW is not None:
                dW[C + C * C + 1:] = ddx[:, -1, :].sum(dim=0)
            E.lattice_grad(num, cneg, coef_w=cneg, gout=gout, dx=dx, accumulate=True, dW=dW)
        elif dx is not None or dW is not None:
            E.lattice_grad(num, cneg, coef_w=cneg, gout=gout, dx=dx, accumulate=False, dW=dW)
            if den is not None:
                E.lattice_grad(den, cpos, coef_w=cpos, gout=gout, dx=dx, accumulate=True, dW=dW)
        if dx is not None and ctx.devices[0].type != "cuda":
            dx = dx.to(ctx.devices[0])
        if dW is not None and ctx.devices[1].type != "cuda":
            dW = dW.to(ctx.devices[1])
        return dx, None, None, None, dW, None, None


class _FusedLogSoftmaxTransducerLoss(TransducerLossFunction):
    """TransducerLoss(log_softmax(inputs), ...) without transitions, as one operator: the gather
    subtracts the rows' log-sum-exp, the gradient kernel differentiates through it."""

    @staticmethod
    @E.on_input_device
    def forward(ctx, inputs, targets, tokens, lexicon, transition_params=None, transitions=None,
                reduction="none"):
        return TransducerLossFunction._forward(ctx, True, inputs, targets, tokens, lexicon, transition_params,
                                               transitions, reduction)


class _EarlyGrad:
    """The emission gradient the sweeps' launch wrote for grad_output = 1, until backward claims it."""

    __slots__ = ("dx", "num", "cneg", "inputs")

    def __init__(self, dx, num, cneg, inputs):
        self.dx, self.num, self.cneg, self.inputs = dx, num, cneg, inputs

    def take(self):
        """E.EagerLoss.backward: the buffer only lacks the rows of the utterances the launch did not serve."""
        if self.dx is None or not E.takes_grad(self.inputs, self.dx):
            return None
        dx, self.dx = self.dx, None
        with torch.cuda.device(dx.device):
            E.lattice_grad_rest(self.num, self.cneg, None, dx)
        return [(self.inputs, dx)]


_IN_LAUNCH_GRAD = os.environ.get("WFL_TRANSDUCER_IN_LAUNCH_GRAD", "1") != "0"  # (0: gradient in backward -- A/B, tests)


def _unigram_route(inputs, targets, tokens, lexicon, transition_params=None, transitions=None, reduction="none"):
    """TransducerLoss with the unigram model (make_transitions_graph(1, C): one node, a self-loop per label) as the
    transition-free step on other emissions (None: not that case).  Every arc with label c carries p_c in numerator and
    normaliser alike, so with x' = x + p the normaliser is  sum_t logsumexp_c x'[t]  and the loss is the negated
    numerator score of log_softmax(x') -- the fused log_softmax criterion (one native call, the gradient beside the
    sweeps); autograd sums the emission gradient over (b, t) into `transition_params`."""
    if transitions is None or transition_params is None or inputs.dim() != 3 or not _DENSE_NGRAM:
        return None
    C = inputs.shape[2]
    if inputs.shape[1] == 0 or transition_params.numel() != C or not _dense_unigram(transitions, C):
        return None
    x = inputs if inputs.dtype == torch.float32 else inputs.float()
    p = transition_params.to(device=x.device, dtype=torch.float32)
    return E.make_eager(_FusedLogSoftmaxTransducerLoss.apply(x + p, targets, tokens, lexicon, None, None, reduction))


def TransducerLoss(*args):
    """transducer.py:346 (`TransducerLoss = TransducerLossFunction.apply`): same call, same result."""
    for route in (_bigram_route, _unigram_route):
        routed = route(*args)
        if routed is not None:
            return routed
    return E.make_eager(TransducerLossFunction.apply(*args))


# -------------------------------------------------------------------------------------------------
# ConvTransduce1D (transducer.py:370-556)
# -------------------------------------------------------------------------------------------------
class _KernelTable:
    """Device description of a lexicon for csrc/conv_kernels.hip (layout: include/wfl.h).  Arc ids
    follow the insertion order of make_kernel_graph, which is also the layout of `kernel_params`
    (transducer.py:474-483 hands consecutive slices of it to the kernels' arc weights)."""

    def __init__(self, lexicon, blank_optional, spike):
        self.lexicon = [tuple(int(c) for c in tok) for tok in lexicon]
        self.blank_optional, self.spike = bool(blank_optional), bool(spike)
        tab = np.zeros((len(self.lexicon), 36), dtype=np.int32)
        ns = 0 if spike else 1
        n = 0
        for k, tok in enumerate(self.lexicon):
            if len(tok) > 15:
                raise ValueError(f"ConvTransduce1D: lexicon entry {k} has {len(tok)} sub-tokens (limit 15)")
            tab[k, 0] = len(tok)
            tab[k, 34] = n  # arc 0 -> 0
            n += 1
            for i, c in enumerate(tok):
                tab[k, 2 + i] = c
                tab[k, 18 + i] = n
                skip = i > 0 and blank_optional and tok[i - 1] != c
                if skip:
                    tab[k, 1] |= 1 << i
                n += 3 + ns + int(skip)
        self.table, self.num_arcs = tab, n
        self.flags = (N.CONV_SPIKE if spike else 0) | (N.CONV_BLANK_OPTIONAL if blank_optional else 0)
        self._dev = {}

    def on(self, device):
        t = self._dev.get(device)
        if t is None:
            t = self._dev[device] = torch.from_numpy(self.table).to(device)
        return t


class ConvTransduce1DFunction(torch.autograd.Function):
    """transducer.py:461-552.  `kernels` is the lexicon table built by ConvTransduce1D (a list of
    kernel graphs made by make_kernel_graph is accepted too and converted).  Unlike the reference
    there is no process-global CTX_GRAPHS: everything backward needs lives on `ctx` (re-entrant)."""

    @staticmethod
    @E.on_input_device
    def forward(ctx, inputs, kernels, kernel_size, stride, kernel_params=None, viterbi=False):
        B, T, C = inputs.shape
        if T < kernel_size:  # padding should be done outside of this function (transducer.py:468-470)
            raise ValueError(f"Input ({T}) too short for kernel ({kernel_size})")
        if not isinstance(kernels, _KernelTable):
            raise TypeError("ConvTransduce1DFunction: pass the ConvTransduce1D module's kernel table")
        dev = E.require_gpu()
        x = E.as_device_f32(inputs.detach(), dev)
        params = E.as_device_f32(kernel_params.detach(), dev) if kernel_params is not None else None
        if params is not None and params.numel() != kernels.num_arcs:
            raise ValueError(f"kernel_params has {params.numel()} entries, the kernels have {kernels.num_arcs} arcs")
        tab = kernels.on(dev)
        K = tab.shape[0]
        blank = kernels.blank_idx
        Tout = (T - kernel_size) // stride + 1
        out = torch.empty((B, Tout, K), dtype=torch.float32, device=dev)
        sr = N.SEMIRING_TROPICAL if viterbi else N.SEMIRING_LOG
        N.check(N.lib.wfl_conv_forward(E.ptr(x), B, T, C, E.ptr(tab), K, kernel_size, stride, blank, kernels.flags,
                                       E.ptr(params), sr, E.ptr(out), E.stream_ptr()))
        ctx.aux = (x, params, tab, kernels, kernel_size, stride, sr)
        ctx.devices = (inputs.device, None if kernel_params is None else kernel_params.device)
        return out if inputs.is_cuda else out.to(inputs.device)

    @staticmethod
    @E.on_input_device
    def backward(ctx, grad_output):
        x, params, tab, kernels, kernel_size, stride, sr = ctx.aux
        B, T, C = x.shape
        delta = E.as_device_f32(grad_output.detach(), x.device)
        dx = torch.empty_like(x)
        dparams = torch.zeros_like(params) if (params is not None and ctx.needs_input_grad[4]) else None
        N.check(N.lib.wfl_conv_grad(E.ptr(x), B, T, C, E.ptr(tab), tab.shape[0], kernel_size, stride,
                                    kernels.blank_idx, kernels.flags, E.ptr(params), sr, E.ptr(delta), E.ptr(dx),
                                    E.ptr(dparams), E.stream_ptr()))
        if ctx.devices[0].type != "cuda":
            dx = dx.to(ctx.devices[0])
        if dparams is not None and ctx.devices[1].type != "cuda":
            dparams = dparams.to(ctx.devices[1])
        return dx, None, None, None, dparams, None


class ConvTransduce1D(torch.nn.Module):
    """A 1D convolutional transducer layer (transducer.py:370-457): every lexicon entry is a small
    alignment graph over the previous layer's tokens, slid over the input like a convolution."""

    def __init__(self, lexicon, kernel_size, stride, blank_idx, blank_optional=True, learn_params=False,
                 scale="none", normalize="none", viterbi=False, spike=False):
        super().__init__()
        self.normalize = normalize
        self.viterbi = viterbi
        if scale == "none":
            self.scale = 1.0
        elif scale == "sqrt":
            self.scale = math.sqrt(kernel_size)
        elif scale == "linear":
            self.scale = kernel_size
        else:
            raise ValueError(f"Unknown scale {scale}")
        if normalize not in ["none", "pre", "post"]:
            raise ValueError(f"Unknown normalization {normalize}")
        self.kernel_size = kernel_size
        assert self.kernel_size % 2 != 0, "Use an odd kernel size for easy padding."
        self.stride = stride

        def size_with_rep(token):
            return len(token) + sum(t1 == t2 for t1, t2 in zip(token[:-1], token[1:]))

        min_kernel_size = max(size_with_rep(l) for l in lexicon)
        if kernel_size < min_kernel_size:
            raise ValueError(f"Kernel size needed of at least {min_kernel_size}.")
        self.kernels = _KernelTable(lexicon, blank_optional, spike)
        self.kernels.blank_idx = int(blank_idx)
        self.kernel_params = None
        if learn_params:
            self.kernel_params = torch.nn.Parameter(torch.zeros(self.kernels.num_arcs))

    @E.on_input_device
    def forward(self, inputs):
        # inputs are of shape [B, T, C]
        pad = self.kernel_size // 2
        inputs = torch.nn.functional.pad(inputs, (0, 0, pad, pad))
        if self.normalize == "pre":
            inputs = torch.nn.functional.log_softmax(inputs, dim=2)
        outputs = ConvTransduce1DFunction.apply(inputs, self.kernels, self.kernel_size, self.stride,
                                                self.kernel_params, self.viterbi)
        outputs = outputs / self.scale
        if self.normalize == "post":
            outputs = torch.nn.functional.softmax(outputs, dim=2)
        if self.normalize == "pre":
            outputs = outputs.exp()
        return outputs
