"""
ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/minigtn.py header).

float64 CPU restatement of the reference's four criteria on top of oracle/minigtn.py.  It has no
torch dependency and no dependency on /root/reference, so it travels to the GPU box and serves as
the checker there.  Each function cites the reference lines it follows.  Interfaces are numpy in /
numpy out; losses are already batch-reduced the way the reference's autograd Functions do it.

Pinning: the graph builders below are compared arc-by-arc with tests/golden/builder_graphs.json
(dumped from the reference's builders) and the losses/gradients with
tests/golden/criterion_cases.json + tests/golden/reference_literals.json, in
tests/test_oracle.py.
"""
import itertools
import math

import numpy as np

from . import minigtn as G


# --------------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------------
def emissions_graph(x_tc, calc_grad=True):
    """[T, C] scores as a chain graph, arc id = t*C + c (ctc.py:40-44)."""
    T, C = x_tc.shape
    g = G.linear_graph(T, C, None, calc_grad)
    g.set_weights(np.asarray(x_tc, dtype=np.float64).reshape(-1))
    return g


def _scale(reduction, n):
    if reduction == "mean":
        return 1.0 / n if n > 0 else 1.0
    if reduction != "none":
        raise ValueError("invalid value for reduction '" + str(reduction) + "'")
    return 1.0


def log_softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    return x - (m + np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True)))


# --------------------------------------------------------------------------------------------------
# CTC  (criterions/ctc.py)
# --------------------------------------------------------------------------------------------------
def ctc_graph(target, blank):
    """ctc.py:15-29 -- 2L+1 states, blank on even states, skip arcs between different labels."""
    g = G.Graph(False)
    S = 2 * len(target) + 1
    for s in range(S):
        g.add_node(s == 0, s >= S - 2)
        lab = target[(s - 1) // 2] if s % 2 else blank
        g.add_arc(s, s, lab)
        if s > 0:
            g.add_arc(s - 1, s, lab)
        if s % 2 and s > 1 and lab != target[(s - 1) // 2 - 1]:
            g.add_arc(s - 2, s, lab)
    g.arc_sort(False)
    return g


def ctc(x, targets, blank, reduction="none"):
    """ctc.py:32-94.  x: [B,T,C] "log_probs" (whatever the caller passes).  Returns (loss, dx)."""
    x = np.asarray(x, dtype=np.float64)
    B = x.shape[0]
    losses, dx = np.zeros(B), np.zeros_like(x)
    for b in range(B):
        em = emissions_graph(x[b])
        loss = G.negate(G.forward_score(G.intersect(em, ctc_graph(targets[b], blank))))
        sc = _scale(reduction, len(targets[b]))
        losses[b] = loss.item() * sc
        G.backward(loss)
        dx[b] = em.grad64().reshape(x[b].shape) * sc / B
    return float(np.mean(losses)), dx


def ctc_greedy(outputs, blank):
    """ctc.py:126-135 -- argmax, collapse repeats, drop blank."""
    res = []
    for row in np.argmax(np.asarray(outputs), axis=2):
        keep = [int(row[0])] + [int(v) for u, v in zip(row[:-1], row[1:]) if u != v] if len(row) else []
        res.append([v for v in keep if v != blank])
    return res


# --------------------------------------------------------------------------------------------------
# ASG  (criterions/asg.py)
# --------------------------------------------------------------------------------------------------
def pack_replabels(tokens, n):
    """asg.py:13-32."""
    if tokens and all(isinstance(t, list) for t in tokens):
        return [pack_replabels(t, n) for t in tokens]
    out, run, prev = [], 0, -1
    for tok in tokens:
        if tok == prev and run < n:
            run += 1
            continue
        if run:
            out.append(run - 1)
            run = 0
        out.append(tok + n)
        prev = tok
    if run:
        out.append(run - 1)
    return out


def unpack_replabels(tokens, n):
    """asg.py:35-49."""
    if tokens and all(isinstance(t, list) for t in tokens):
        return [unpack_replabels(t, n) for t in tokens]
    out, prev = [], -1
    for tok in tokens:
        if tok >= n:
            out.append(tok - n)
            prev = tok
        elif prev != -1:
            out.extend([prev - n] * (tok + 1))
            prev = -1
    return out


def asg_transitions_graph(W, calc_grad=False):
    """asg.py:54-69 -- node 0 start, nodes 1..C accept; arc id order = row-major W[(C+1),C]."""
    W = np.asarray(W, dtype=np.float64)
    C = W.shape[1]
    assert W.shape == (C + 1, C)
    g = G.Graph(calc_grad)
    g.add_node(True)
    for i in range(C):
        g.add_node(False, True)
        g.add_arc(0, i + 1, i)
    for i in range(C):
        for j in range(C):
            g.add_arc(j + 1, i + 1, i)
    g.set_weights(W.reshape(-1))
    return g


def asg_force_align_graph(target):
    """asg.py:72-81."""
    g = G.Graph(False)
    g.add_node(True)
    for l, lab in enumerate(target, start=1):
        g.add_node(False, l == len(target))
        g.add_arc(l - 1, l, lab)
        g.add_arc(l, l, lab)
    g.arc_sort(True)
    return g


def asg(x, W, targets, reduction="none"):
    """asg.py:84-185.  Returns (loss, dx[B,T,C], dW[(C+1),C])."""
    x = np.asarray(x, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    B = x.shape[0]
    losses, dx, dW = np.zeros(B), np.zeros_like(x), np.zeros((B,) + W.shape)
    for b in range(B):
        em = emissions_graph(x[b])
        tr = asg_transitions_graph(W, True)
        fal = G.forward_score(G.intersect(G.intersect(asg_force_align_graph(targets[b]), tr), em))
        fcc = G.forward_score(G.intersect(em, tr))
        loss = G.subtract(fcc, fal)
        sc = _scale(reduction, len(targets[b]))
        losses[b] = loss.item() * sc
        G.backward(loss)
        dx[b] = em.grad64().reshape(x[b].shape) * sc / B
        dW[b] = tr.grad64().reshape(W.shape) * sc
    return float(np.mean(losses)), dx, dW.mean(axis=0)


def asg_module_targets(targets, num_classes, num_replabels, use_garbage):
    """asg.py:201-208 -- replabel packing and garbage interleaving."""
    out = []
    garbage = num_classes + num_replabels
    for t in targets:
        p = pack_replabels(list(t), num_replabels)
        if use_garbage:
            q = [garbage] * (2 * len(p) + 1)
            q[1::2] = p
            p = q
        out.append(p)
    return out


def asg_viterbi(outputs, W, num_replabels, garbage_idx=None):
    """asg.py:211-237."""
    res = []
    for y in np.asarray(outputs, dtype=np.float64):
        path = G.viterbi_path(G.intersect(emissions_graph(y, False), asg_transitions_graph(W))).labels_to_list()
        path = [p for p, _ in itertools.groupby(path)]
        if garbage_idx is not None:
            path = [p for p in path if p != garbage_idx]
        res.append(unpack_replabels(path, num_replabels))
    return res


# --------------------------------------------------------------------------------------------------
# STC  (criterions/stc.py)
# --------------------------------------------------------------------------------------------------
STC_BLANK = 0


def stc_graph(target, star_idx, prob):
    """stc.py:23-64."""
    g = G.Graph(False)
    L = len(target)
    S = 2 * L + 1
    for s in range(S):
        g.add_node(s == 0, s >= S - 2)
        lab = target[(s - 1) // 2] if s % 2 else STC_BLANK
        if lab == STC_BLANK:
            g.add_arc(s, s, lab)
        if s > 0:
            g.add_arc(s - 1, s, lab)
        if s % 2 and s > 1:
            g.add_arc(s - 2, s, lab)
    lp = math.log(prob)
    for l in range(L + 1):
        p_tok, p_blank = 2 * l - 1, 2 * l
        c = g.add_node(False, l == L)
        star = star_idx if l == L else star_idx + target[l]
        if p_tok >= 0:
            g.add_arc(p_tok, c, star, star, lp)
        g.add_arc(p_blank, c, star, star, lp)
        g.add_arc(c, c, star, star, lp)
        if l < L:
            g.add_arc(c, 2 * l + 1, target[l])
        g.add_arc(c, p_blank, STC_BLANK)
    return g


def stc_function(x, targets, prob, reduction="none"):
    """stc.py:67-129 on the already star-augmented [B,T,2C'] input.  Returns (loss, dx)."""
    x = np.asarray(x, dtype=np.float64)
    B, T, Cstar = x.shape
    losses, dx = np.zeros(B), np.zeros_like(x)
    for b in range(B):
        em = emissions_graph(x[b])
        crit = stc_graph(targets[b], Cstar // 2, prob)
        crit.arc_sort(False)
        loss = G.negate(G.forward_score(G.compose(crit, em)))
        sc = _scale(reduction, T)
        losses[b] = loss.item() * sc
        G.backward(loss)
        dx[b] = em.grad64().reshape(x[b].shape) * sc / B
    return float(np.mean(losses)), dx


def stc_augment(log_probs_btc, targets):
    """stc.py:199-220 (torch-side preprocessing) in numpy.

    Returns (augmented [B,T,2C'], remapped targets, select_idx).  The `set()` iteration order of the
    reference is replaced by sorted order (any order gives the same loss: it only renames columns).
    """
    lp = np.asarray(log_probs_btc, dtype=np.float64)
    with np.errstate(all="ignore"):
        rest = lp[:, :, 1:]
        m = np.max(rest, axis=2, keepdims=True)
        msafe = np.where(np.isfinite(m), m, 0.0)
        lse = msafe + np.log(np.sum(np.exp(rest - msafe), axis=2, keepdims=True))
        select = [STC_BLANK] + sorted(set(t for tg in targets for t in tg))
        remap = {t: i for i, t in enumerate(select)}
        sel = lp[:, :, select]
        neglse = lse + np.log1p(1e-7 - np.exp(sel[:, :, 1:] - lse))
    aug = np.concatenate([sel, lse, neglse], axis=2)
    return aug, [[remap[t] for t in tg] for tg in targets], select


def stc_prob(p0, plast, thalf, nstep):
    """stc.py:193-195."""
    return plast + (p0 - plast) * math.exp(-nstep * math.log(2) / thalf)


# --------------------------------------------------------------------------------------------------
# Transducer  (criterions/transducer.py)
# --------------------------------------------------------------------------------------------------
def make_chain_graph(seq):
    """transducer.py:23-29."""
    g = G.Graph(False)
    g.add_node(True)
    for i, s in enumerate(seq):
        g.add_node(False, i == len(seq) - 1)
        g.add_arc(i, i + 1, int(s))
    return g


def make_transitions_graph(ngram, num_tokens, calc_grad=False):
    """transducer.py:32-58."""
    g = G.Graph(calc_grad)
    g.add_node(True, ngram == 1)
    ids = {(): 0}
    for n in range(1, ngram):
        for st in itertools.product(range(num_tokens), repeat=n):
            ids[st] = g.add_node(False, ngram == 1)
            g.add_arc(ids[st[:-1]], ids[st], st[-1])
    for st in itertools.product(range(num_tokens), repeat=ngram):
        g.add_arc(ids[st[:-1]], ids[st[1:]], st[-1])
    if ngram > 1:
        end = g.add_node(False, True)
        for n in range(end):
            g.add_arc(n, end, G.epsilon)
    return g


def make_lexicon_graph(word_pieces, graphemes_to_idx):
    """transducer.py:61-75."""
    g = G.Graph(False)
    g.add_node(True, True)
    for i, wp in enumerate(word_pieces):
        prev = 0
        for ch in wp[:-1]:
            n = g.add_node()
            g.add_arc(prev, n, graphemes_to_idx[ch], G.epsilon)
            prev = n
        g.add_arc(prev, 0, graphemes_to_idx[wp[-1]], i)
    g.arc_sort()
    return g


def make_token_graph(token_list, blank="none", allow_repeats=True):
    """transducer.py:78-123."""
    if not allow_repeats and blank != "optional":
        raise ValueError("Must use blank='optional' if disallowing repeats.")
    n = len(token_list)
    g = G.Graph(False)
    g.add_node(True, True)
    for _ in range(n):
        g.add_node(False, blank != "forced")
    if blank != "none":
        g.add_node()
        g.add_arc(0, n + 1, n, G.epsilon)
        g.add_arc(n + 1, 0, G.epsilon)
    entry = n + 1 if blank == "forced" else 0
    for i in range(n):
        g.add_arc(entry, i + 1, i)
        g.add_arc(i + 1, i + 1, i, G.epsilon)
        if allow_repeats:
            if blank == "forced":
                g.add_arc(i + 1, n + 1, n, G.epsilon)
            else:
                g.add_arc(i + 1, 0, G.epsilon)
        else:
            g.add_arc(i + 1, n + 1, n, G.epsilon)
            for j in range(n):
                if j != i:
                    g.add_arc(i + 1, j + 1, j, j)
    return g


def make_kernel_graph(x, blank_idx, blank_optional, spike=False, calc_grad=False):
    """transducer.py:351-367."""
    g = G.Graph(calc_grad)
    g.add_node(True, len(x) == 0)
    g.add_arc(0, 0, blank_idx)
    for i, c in enumerate(x):
        last = i + 1 == len(x)
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


class TransducerOracle:
    """transducer.py:126-348 (module + autograd Function), numpy in / numpy out."""

    def __init__(self, tokens, graphemes_to_idx, ngram=0, transitions=None, blank="none",
                 allow_repeats=True, reduction="none"):
        if blank not in ("optional", "forced", "none"):
            raise ValueError("Invalid value specificed for blank. Must be in ['optional', 'forced', 'none']")
        self.tokens = make_token_graph(tokens, blank=blank, allow_repeats=allow_repeats)
        self.lexicon = make_lexicon_graph(tokens, graphemes_to_idx)
        if ngram > 0 and transitions is not None:
            raise ValueError("Only one of ngram and transitions may be specified")
        if ngram > 0:
            transitions = make_transitions_graph(ngram, len(tokens) + int(blank != "none"), True)
        self.transitions = transitions
        self.transition_params = None
        if transitions is not None:
            transitions.arc_sort()
            self.transition_params = np.zeros(transitions.num_arcs())
        self.reduction = reduction

    def alignment_graph(self, target):
        """transducer.py:265-276: all alignments of all decompositions of `target` into tokens."""
        tgt = make_chain_graph(target)
        tgt.arc_sort(True)
        tokens_target = G.remove(G.project_output(G.compose(tgt, self.lexicon)))
        tokens_target.arc_sort()
        self.tokens.arc_sort(True)
        ali = G.project_input(G.remove(G.compose(self.tokens, tokens_target)))
        ali.arc_sort()
        return ali

    def loss(self, x, targets):
        """Returns (loss, dx, dparams or None).  transducer.py:185-197,239-348."""
        x = np.asarray(x, dtype=np.float64)
        B = x.shape[0]
        tr = self.transitions
        if tr is None:
            x_in = log_softmax(x, 2)
        else:
            x_in = x
            tr.set_weights(self.transition_params)
            tr.calc_grad = True
            tr.zero_grad()
        losses, dx_in = np.zeros(B), np.zeros_like(x)
        for b in range(B):
            em = emissions_graph(x_in[b])
            ali = self.alignment_graph(targets[b])
            if tr is not None:
                ali = G.intersect(tr, ali)
                ali.arc_sort()
            score = G.forward_score(G.intersect(em, ali))
            if tr is not None:
                score = G.subtract(score, G.forward_score(G.intersect(em, tr)))
            loss = G.negate(score)
            sc = _scale(self.reduction, len(targets[b])) if self.reduction == "mean" else 1.0
            losses[b] = loss.item() * sc
            G.backward(loss, G.scalar_graph(sc))
            dx_in[b] = em.grad64().reshape(x[b].shape) / B
        dparams = None
        if tr is not None:
            dparams = tr.grad64() / B
            dx = dx_in
        else:  # chain rule through log_softmax (transducer.py:186-187)
            p = np.exp(x_in)
            dx = dx_in - p * dx_in.sum(axis=2, keepdims=True)
        return float(np.mean(losses)), dx, dparams

    def viterbi(self, outputs):
        """transducer.py:199-234."""
        tr = self.transitions
        if tr is not None:
            tr.set_weights(self.transition_params)
            tr.calc_grad = False
        self.tokens.arc_sort()
        res = []
        for y in np.asarray(outputs, dtype=np.float64):
            em = emissions_graph(y, False)
            full = G.intersect(em, tr) if tr is not None else em
            path = G.remove(G.viterbi_path(full))
            path = G.compose(path, self.tokens)
            path = G.viterbi_path(path)
            path = G.remove(G.project_output(path))
            res.append(path.labels_to_list())
        return res


def conv_transduce_1d_grad(x, lexicon, blank_idx, kernel_size, stride, deltas, blank_optional=True, spike=False,
                           kernel_params=None, viterbi=False):
    """transducer.py:461-552 forward + backward for x [B,T,C] (already padded) and upstream
    gradients deltas [B,Tout,K]: returns (out [B,Tout,K], dx [B,T,C], dparams [num_arcs] or None)."""
    x = np.asarray(x, dtype=np.float64)
    B, T, C = x.shape
    if T < kernel_size:
        raise ValueError(f"Input ({T}) too short for kernel ({kernel_size})")
    kernels = [make_kernel_graph(l, blank_idx, blank_optional, spike) for l in lexicon]
    if kernel_params is not None:  # transducer.py:474-483: consecutive slices of kernel_params
        kp = np.asarray(kernel_params, dtype=np.float64)
        s = 0
        for k in kernels:
            na = k.num_arcs()
            k.set_weights(kp[s:s + na])
            k.calc_grad = True
            k.zero_grad()
            s += na
    score = G.viterbi_score if viterbi else G.forward_score
    starts = list(range(0, T - kernel_size + 1, stride))
    out = np.zeros((B, len(starts), len(kernels)))
    dx = np.zeros_like(x)
    for b in range(B):
        for w, t in enumerate(starts):
            em = emissions_graph(x[b, t:t + kernel_size], True)
            for c, k in enumerate(kernels):
                o = score(G.intersect(em, k))
                out[b, w, c] = o.item()
                G.backward(o, G.scalar_graph(float(deltas[b][w][c])))
            dx[b, t:t + kernel_size] += em.grad64().reshape(kernel_size, C)
    dparams = None
    if kernel_params is not None:
        dparams = np.concatenate([k.grad64() for k in kernels])
    return out, dx, dparams


def conv_layer(x, lexicon, kernel_size, stride, blank_idx, out_weights, blank_optional=True, learn_params=False,
               scale="none", normalize="none", viterbi=False, spike=False, kernel_params=None):
    """The ConvTransduce1D module (transducer.py:438-457) around conv_transduce_1d_grad, with the
    scalar objective sum(out * out_weights): returns (out, d objective / d x, d / d kernel_params)."""
    x = np.asarray(x, dtype=np.float64)
    W = np.asarray(out_weights, dtype=np.float64)
    pad = kernel_size // 2
    xp = np.pad(x, ((0, 0), (pad, pad), (0, 0)))
    xin = log_softmax(xp, 2) if normalize == "pre" else xp
    sc = {"none": 1.0, "sqrt": math.sqrt(kernel_size), "linear": float(kernel_size)}[scale]
    ones = np.ones((x.shape[0], (xp.shape[1] - kernel_size) // stride + 1, len(lexicon)))
    raw, _, _ = conv_transduce_1d_grad(xin, lexicon, blank_idx, kernel_size, stride, 0.0 * ones, blank_optional, spike,
                                       kernel_params if learn_params else None, viterbi)
    y = raw / sc
    if normalize == "post":
        e = np.exp(y - y.max(axis=2, keepdims=True))
        out = e / e.sum(axis=2, keepdims=True)
        dy = out * (W - (W * out).sum(axis=2, keepdims=True))
    elif normalize == "pre":
        out = np.exp(y)
        dy = W * out
    else:
        out, dy = y, W
    _, dxin, dparams = conv_transduce_1d_grad(xin, lexicon, blank_idx, kernel_size, stride, dy / sc, blank_optional,
                                              spike, kernel_params if learn_params else None, viterbi)
    if normalize == "pre":
        dxp = dxin - np.exp(xin) * dxin.sum(axis=2, keepdims=True)
    else:
        dxp = dxin
    return out, dxp[:, pad:xp.shape[1] - pad], dparams


def conv_transduce_1d(x, kernels, kernel_size, stride, viterbi=False):
    """transducer.py:461-524 forward only: [B,T,C] -> [B,Tout,len(kernels)] window scores."""
    x = np.asarray(x, dtype=np.float64)
    B, T, C = x.shape
    if T < kernel_size:
        raise ValueError(f"Input ({T}) too short for kernel ({kernel_size})")
    score = G.viterbi_score if viterbi else G.forward_score
    out = []
    for b in range(B):
        rows = []
        for t in range(0, T - kernel_size + 1, stride):
            em = emissions_graph(x[b, t:t + kernel_size], False)
            rows.append([score(G.intersect(em, k)).item() for k in kernels])
        out.append(rows)
    return np.asarray(out)
