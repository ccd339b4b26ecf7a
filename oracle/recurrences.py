"""
ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/minigtn.py header).

Vectorised float64 numpy restatements of the time-synchronous recurrences that
`forward_score(intersect(emissions, A))` reduces to when the emissions graph is a chain
(criterions/ctc.py:15-69, criterions/asg.py:54-139).  They exist so that
the checker finishes in seconds at BASELINE sizes (T=1000), where the graph-based oracle
(oracle/criteria.py) would take minutes.  tests/test_oracle.py checks them against the graph-based
oracle on small shapes, so they inherit its pinning.
"""
import numpy as np

NEG = -np.inf


def _lse(a, axis=None):
    with np.errstate(all="ignore"):
        m = np.max(a, axis=axis, keepdims=True)
        ms = np.where(np.isfinite(m), m, 0.0)
        out = ms + np.log(np.sum(np.exp(a - ms), axis=axis, keepdims=True))
    return np.squeeze(out, axis=axis) if axis is not None else float(out.reshape(()))


def lattice_forward_backward(x, src, dst, lab, w, start, accept, num_states):
    """Generic epsilon-free acceptor A composed with a [T,C] emissions chain.

    alpha_{t+1}[q'] = LSE_{arcs q->q'} alpha_t[q] + x[t,lab] + w ;  logZ = LSE_{q in accept} alpha_T[q]
    Returns (logZ, dx[T,C] = sum of arc posteriors per (t,label), darc[num_arcs] = sum_t posterior).
    """
    x = np.asarray(x, dtype=np.float64)
    T, C = x.shape
    src, dst, lab = (np.asarray(v, dtype=np.int64) for v in (src, dst, lab))
    w = np.asarray(w, dtype=np.float64)
    w = np.where(np.isnan(w), NEG, w)
    x = np.where(np.isnan(x), NEG, x)
    alpha = np.full((T + 1, num_states), NEG)
    alpha[0, list(start)] = 0.0
    with np.errstate(all="ignore"):
        for t in range(T):
            v = alpha[t, src] + x[t, lab] + w
            np.logaddexp.at(alpha[t + 1], dst, v)
        beta = np.full((T + 1, num_states), NEG)
        beta[T, list(accept)] = 0.0
        for t in range(T - 1, -1, -1):
            v = beta[t + 1, dst] + x[t, lab] + w
            np.logaddexp.at(beta[t], src, v)
        logz = _lse(alpha[T, list(accept)]) if len(accept) else NEG
        dx = np.zeros((T, C))
        darc = np.zeros(len(src))
        if np.isfinite(logz):
            for t in range(T):
                g = np.exp(alpha[t, src] + x[t, lab] + w + beta[t + 1, dst] - logz)
                g = np.where(np.isfinite(g), g, 0.0)
                np.add.at(dx[t], lab, g)
                darc += g
    return float(logz), dx, darc


def _eps_depths(num_states, esrc, edst):
    """longest epsilon path ending at / starting from every node (the epsilon arcs must not form a cycle)"""
    din, dout = np.zeros(num_states, np.int64), np.zeros(num_states, np.int64)
    for _ in range(num_states + 1):
        nin, nout = din.copy(), dout.copy()
        np.maximum.at(nin, edst, din[esrc] + 1)
        np.maximum.at(nout, esrc, dout[edst] + 1)
        if (nin == din).all() and (nout == dout).all():
            return din, dout
        din, dout = nin, nout
    raise ValueError("epsilon arcs form a cycle")


def lattice_forward_backward_eps(x, src, dst, lab, w, start, accept, num_states):
    """lattice_forward_backward for an acceptor WITH epsilon arcs (lab < 0: back-off arcs of a transition model,
    the </s> arcs of make_transitions_graph -- criterions/transducer.py:32-58,279-288): after every frame (and before
    the first) the epsilon arcs are followed, in topological order of their acyclic subgraph,
        alpha_t[q] = LSE(direct_t[q], LSE_{eps arcs p->q} alpha_t[p] + w),   beta_t likewise backwards.
    Returns (logZ, dx[T,C], darc[num_arcs]); an epsilon arc's entry sums its posteriors over the T + 1 closures.
    """
    x = np.asarray(x, dtype=np.float64)
    T, C = x.shape
    src, dst, lab = (np.asarray(v, dtype=np.int64) for v in (src, dst, lab))
    w = np.asarray(w, dtype=np.float64)
    w = np.where(np.isnan(w), NEG, w)
    x = np.where(np.isnan(x), NEG, x)
    ie, il = np.flatnonzero(lab < 0), np.flatnonzero(lab >= 0)
    ls, ld, ll, lw = src[il], dst[il], lab[il], w[il]
    es, ed, ew = src[ie], dst[ie], w[ie]
    din, dout = _eps_depths(num_states, es, ed)
    fwd_groups = [np.flatnonzero(din[es] == d) for d in range(int(din.max()) + 1)] if len(ie) else []
    bwd_groups = [np.flatnonzero(dout[ed] == d) for d in range(int(dout.max()) + 1)] if len(ie) else []

    def close_fwd(a):
        for g in fwd_groups:
            if len(g):
                np.logaddexp.at(a, ed[g], a[es[g]] + ew[g])

    def close_bwd(b):
        for g in bwd_groups:
            if len(g):
                np.logaddexp.at(b, es[g], b[ed[g]] + ew[g])

    alpha = np.full((T + 1, num_states), NEG)
    beta = np.full((T + 1, num_states), NEG)
    with np.errstate(all="ignore"):
        alpha[0, list(start)] = 0.0
        close_fwd(alpha[0])
        for t in range(T):
            np.logaddexp.at(alpha[t + 1], ld, alpha[t, ls] + x[t, ll] + lw)
            close_fwd(alpha[t + 1])
        beta[T, list(accept)] = 0.0
        close_bwd(beta[T])
        for t in range(T - 1, -1, -1):
            np.logaddexp.at(beta[t], ls, beta[t + 1, ld] + x[t, ll] + lw)
            close_bwd(beta[t])
        logz = _lse(alpha[T, list(accept)]) if len(accept) else NEG
        dx = np.zeros((T, C))
        darc = np.zeros(len(src))
        if np.isfinite(logz):
            for t in range(T):
                g = np.exp(alpha[t, ls] + x[t, ll] + lw + beta[t + 1, ld] - logz)
                g = np.where(np.isfinite(g), g, 0.0)
                np.add.at(dx[t], ll, g)
                darc[il] += g
            if len(ie):
                g = np.exp(alpha[:, es] + ew[None, :] + beta[:, ed] - logz)
                darc[ie] += np.where(np.isfinite(g), g, 0.0).sum(axis=0)
    return float(logz), dx, darc


def ctc_arcs(target, blank):
    """Arc list of the CTC label graph (criterions/ctc.py:15-29)."""
    L = len(target)
    S = 2 * L + 1
    src, dst, lab = [], [], []
    for s in range(S):
        c = target[(s - 1) // 2] if s % 2 else blank
        src.append(s), dst.append(s), lab.append(c)
        if s > 0:
            src.append(s - 1), dst.append(s), lab.append(c)
        if s % 2 and s > 1 and c != target[(s - 1) // 2 - 1]:
            src.append(s - 2), dst.append(s), lab.append(c)
    accept = [S - 1] if S == 1 else [S - 1, S - 2]
    return src, dst, lab, [0], accept, S


def ctc_loss_grad(x, targets, blank, reduction="none"):
    """criterions/ctc.py:32-94 for x [B,T,C]: returns (mean loss, dx [B,T,C])."""
    x = np.asarray(x, dtype=np.float64)
    B = x.shape[0]
    losses, dx = np.zeros(B), np.zeros_like(x)
    for b in range(B):
        src, dst, lab, st, acc, S = ctc_arcs(list(targets[b]), blank)
        logz, g, _ = lattice_forward_backward(x[b], src, dst, lab, np.zeros(len(src)), st, acc, S)
        sc = 1.0
        if reduction == "mean" and len(targets[b]) > 0:
            sc = 1.0 / len(targets[b])
        losses[b] = -logz * sc
        dx[b] = -g * sc / B
    return float(losses.mean()), dx


def dense_forward_backward(x, W):
    """Fully connected transitions (criterions/asg.py:54-69 layout): W[0,i] = start->i,
    W[1+i, j] = score of (prev=j -> cur=i).  Returns (logZ, dx[T,C] state posteriors, dW)."""
    x = np.asarray(x, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    T, C = x.shape
    M = W[1:]  # [cur, prev]
    with np.errstate(all="ignore"):
        alpha = np.empty((T, C))
        alpha[0] = x[0] + W[0]
        for t in range(1, T):
            alpha[t] = x[t] + _lse(alpha[t - 1][None, :] + M, axis=1)
        beta = np.zeros((T, C))
        for t in range(T - 2, -1, -1):
            beta[t] = _lse((beta[t + 1] + x[t + 1])[:, None] + M, axis=0)
        logz = _lse(alpha[T - 1])
        dx = np.exp(alpha + beta - logz)
        dW = np.zeros_like(W)
        dW[0] = dx[0]
        for t in range(1, T):
            dW[1:] += np.exp(alpha[t - 1][None, :] + M + (x[t] + beta[t])[:, None] - logz)
    return float(logz), dx, dW


def asg_loss_grad(x, W, targets, reduction="none"):
    """criterions/asg.py:84-185 for x [B,T,C], W [(C+1),C]: (mean loss, dx, dW)."""
    x = np.asarray(x, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    B, T, C = x.shape
    losses, dx, dW = np.zeros(B), np.zeros_like(x), np.zeros_like(W)
    for b in range(B):
        y = list(targets[b])
        L = len(y)
        # force-align lattice: states 0..L (0 = start), arc l-1 -> l and self loop l -> l, label y[l-1]
        src, dst, lab, wid = [], [], [], []
        for l in range(1, L + 1):
            c = y[l - 1]
            src.append(l - 1), dst.append(l), lab.append(c)
            wid.append(c if l == 1 else (1 + c) * C + y[l - 2])
            src.append(l), dst.append(l), lab.append(c)
            wid.append((1 + c) * C + c)
        wflat = W.reshape(-1)
        fal, gx, garc = lattice_forward_backward(
            x[b], src, dst, lab, wflat[wid] if wid else np.zeros(0), [0], [L], L + 1
        )
        fcc, px, pW = dense_forward_backward(x[b], W)
        sc = 1.0
        if reduction == "mean" and L > 0:
            sc = 1.0 / L
        losses[b] = (fcc - fal) * sc
        dx[b] = (px - gx) * sc / B
        gW = np.zeros(W.size)
        np.add.at(gW, np.asarray(wid, dtype=np.int64), garc)
        dW += (pW - gW.reshape(W.shape)) * sc / B
    return float(losses.mean()), dx, dW


def dense_viterbi(x, W):
    """max-plus analogue of dense_forward_backward; ties -> lowest previous index, then lowest
    final index (strict '>' relaxation in index order).  Returns the label path of length T."""
    x = np.asarray(x, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    T, C = x.shape
    M = W[1:]
    score = x[0] + W[0]
    back = np.zeros((T, C), dtype=np.int64)
    for t in range(1, T):
        cand = score[None, :] + M  # [cur, prev]
        back[t] = np.argmax(cand, axis=1)  # first maximum = lowest prev index
        score = x[t] + cand[np.arange(C), back[t]]
    cur = int(np.argmax(score))
    path = [cur]
    for t in range(T - 1, 0, -1):
        cur = int(back[t, cur])
        path.append(cur)
    return path[::-1]


# --------------------------------------------------------------------------------------------------
# Batched variants for BASELINE-size parity checks (tests only).  Same recurrences as above, vectorised
# over the batch so that B=128, T=1000..2000 finishes in seconds; tests/test_oracle.py checks them against
# the per-utterance functions above, so they inherit their pinning.
# --------------------------------------------------------------------------------------------------
def ctc_loss_grad_batched(x, targets, blank, reduction="none", out_dtype=np.float64, batch_size=None):
    """criterions/ctc.py:32-94, all utterances at once (ragged targets are padded; padded states are
    unreachable).  Returns (per-utterance scaled losses [B], dx [B,T,C]).  `batch_size`: the B of the 1/B factor
    when x is a slice of a larger batch (ctc.py:87)."""
    x = np.asarray(x, dtype=np.float64)
    x = np.where(np.isnan(x), NEG, x)
    B, T, C = x.shape
    lens = np.array([len(t) for t in targets], dtype=np.int64)
    Lm = int(lens.max()) if B else 0
    S = 2 * Lm + 1
    lab = np.full((B, S), blank, dtype=np.int64)
    for b, t in enumerate(targets):
        if len(t):
            lab[b, 1:2 * len(t):2] = np.asarray(t, dtype=np.int64)
    s_idx = np.arange(S)[None, :]
    valid = s_idx < (2 * lens[:, None] + 1)
    skip = np.zeros((B, S), dtype=bool)  # arc s-2 -> s exists: s odd, s > 1, label differs from the previous label
    skip[:, 3::2] = lab[:, 3::2] != lab[:, 1:-2:2]
    skip &= valid
    bi = np.arange(B)[:, None]
    e = np.transpose(x[bi, :, lab], (0, 2, 1))  # [B,T,S] emission of each state's label per frame
    e = np.where(valid[:, None, :], e, NEG)

    def shift(a, k):
        out = np.full_like(a, NEG)
        out[:, k:] = a[:, :-k]
        return out

    with np.errstate(all="ignore"):
        alpha = np.full((B, T, S), NEG)
        alpha[:, 0, 0] = e[:, 0, 0]
        if S > 1:
            alpha[:, 0, 1] = e[:, 0, 1]
        for t in range(1, T):
            p = alpha[:, t - 1]
            acc = np.logaddexp(p, shift(p, 1))
            if S > 2:
                acc = np.logaddexp(acc, np.where(skip, shift(p, 2), NEG))
            alpha[:, t] = acc + e[:, t]
        last = 2 * lens  # final blank state; the last label state (last-1) also accepts when L > 0
        fin = np.full((B, S), NEG)
        fin[np.arange(B), last] = 0.0
        has = lens > 0
        fin[np.arange(B)[has], last[has] - 1] = 0.0
        logz = _lse(alpha[:, T - 1] + fin, axis=1)
        # beta[t][s] = score of finishing from state s after frame t (emission of frame t excluded)
        beta = np.full((B, T, S), NEG)
        beta[:, T - 1] = fin
        skip_from = np.zeros((B, S), dtype=bool)  # arc s -> s+2 exists
        skip_from[:, :-2] = skip[:, 2:]

        def shl(a, k):
            out = np.full_like(a, NEG)
            out[:, :-k] = a[:, k:]
            return out

        for t in range(T - 2, -1, -1):
            q = beta[:, t + 1] + e[:, t + 1]
            acc = np.logaddexp(q, shl(q, 1))
            if S > 2:
                acc = np.logaddexp(acc, np.where(skip_from, shl(q, 2), NEG))
            beta[:, t] = acc
        post = np.exp(alpha + beta - logz[:, None, None])
        post = np.where(np.isfinite(post), post, 0.0)
    post[~np.isfinite(logz)] = 0.0
    sc = np.ones(B)
    if reduction == "mean":
        sc = np.where(lens > 0, 1.0 / np.maximum(lens, 1), 1.0)
    dx = np.zeros((B, T, C), dtype=out_dtype)
    coef = -(sc / (batch_size or B))
    for b in range(B):  # scatter-add the state posteriors into their label columns
        g = np.zeros((T, C))
        np.add.at(g, (np.arange(T)[:, None], lab[b][None, :]), post[b])
        dx[b] = g * coef[b]
    return -logz * sc, dx


def asg_loss_grad_batched(x, W, targets, reduction="none"):
    """criterions/asg.py:84-185 for a whole batch: the fully connected sweeps run as float64 scaled
    matrix products over the batch (finite scores only), the force-aligned numerator through
    lattice_forward_backward.  Returns (scaled per-utterance losses [B], dx [B,T,C], dW)."""
    x = np.asarray(x, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    if not (np.isfinite(x).all() and np.isfinite(W).all()):
        raise ValueError("asg_loss_grad_batched: finite scores only (use asg_loss_grad)")
    B, T, C = x.shape
    M = W[1:]  # [cur, prev]
    mrow = M.max()
    P = np.exp(M - mrow)  # [cur, prev], entries in (0, 1]
    xm = x.max(axis=2, keepdims=True)
    ex = np.exp(x - xm)  # [B,T,C]
    # alpha~_t = ex_t * (alpha~_{t-1} @ P^T), normalised; log scale in la[b,t]
    a = np.empty((B, T, C))
    la = np.empty((B, T))
    v = ex[:, 0] * np.exp(W[0] - W[0].max())
    s = v.sum(axis=1)
    a[:, 0] = v / s[:, None]
    la[:, 0] = np.log(s) + xm[:, 0, 0] + W[0].max()
    for t in range(1, T):
        v = ex[:, t] * (a[:, t - 1] @ P.T)
        s = v.sum(axis=1)
        a[:, t] = v / s[:, None]
        la[:, t] = la[:, t - 1] + np.log(s) + xm[:, t, 0] + mrow
    logz = la[:, T - 1]  # sum of alpha~_{T-1} is 1
    bt = np.empty((B, T, C))
    lb = np.empty((B, T))
    bt[:, T - 1] = 1.0 / C
    lb[:, T - 1] = np.log(C)
    for t in range(T - 2, -1, -1):
        v = (bt[:, t + 1] * ex[:, t + 1]) @ P
        s = v.sum(axis=1)
        bt[:, t] = v / s[:, None]
        lb[:, t] = lb[:, t + 1] + np.log(s) + xm[:, t + 1, 0] + mrow
    post = a * bt * np.exp(la + lb - logz[:, None])[:, :, None]  # state posteriors [B,T,C]
    lens = np.array([len(t) for t in targets])
    sc = np.ones(B)
    if reduction == "mean":
        sc = np.where(lens > 0, 1.0 / np.maximum(lens, 1), 1.0)
    cf = sc / B
    # transition posteriors: xi_t[i,j] = a_{t-1}[j] P[i,j] ex_t[i] bt_t[i] * exp(la_{t-1} + lb_t + xm_t + mrow - logz)
    wgt = np.exp(la[:, :-1] + lb[:, 1:] + xm[:, 1:, 0] + mrow - logz[:, None]) * cf[:, None]  # [B,T-1]
    left = (ex[:, 1:] * bt[:, 1:] * wgt[:, :, None]).reshape(-1, C)  # rows (b,t): cur side
    right = a[:, :-1].reshape(-1, C)  # prev side
    dW = np.zeros_like(W)
    dW[1:] = P * (left.T @ right)
    dW[0] = (post[:, 0] * cf[:, None]).sum(axis=0)
    dx = post * cf[:, None, None]
    losses = np.empty(B)
    wflat = W.reshape(-1)
    for b in range(B):
        y = list(targets[b])
        L = len(y)
        src, dst, lab, wid = [], [], [], []
        for l in range(1, L + 1):
            c = y[l - 1]
            src.append(l - 1), dst.append(l), lab.append(c)
            wid.append(c if l == 1 else (1 + c) * C + y[l - 2])
            src.append(l), dst.append(l), lab.append(c)
            wid.append((1 + c) * C + c)
        fal, gx, garc = lattice_forward_backward(x[b], src, dst, lab, wflat[wid] if wid else np.zeros(0), [0], [L], L + 1)
        losses[b] = (logz[b] - fal) * sc[b]
        dx[b] -= gx * cf[b]
        gW = np.zeros(W.size)
        np.add.at(gW, np.asarray(wid, dtype=np.int64), garc)
        dW -= gW.reshape(W.shape) * cf[b]
    return losses, dx, dW


def transducer_transitions_loss_grad(x, numerators, transitions, params, scales):
    """The Transducer criterion WITH a transition model (criterions/transducer.py:279-290,312-348) through the
    epsilon-aware recurrence above, for arc lists instead of graphs:
        loss_b = scale_b * (logZ(emissions_b o transitions) - logZ(emissions_b o numerator_b)),  loss = mean_b loss_b
    x [B,T,C] raw scores; numerators[b] = (src, dst, lab, wid, start, accept, num_states) -- the acceptor
    transitions o alignments_b with, per arc, the arc of `transitions` behind it (wid: its weight is params[wid]);
    transitions = (src, dst, lab, start, accept, num_states), arc a weighted params[a].
    Returns (loss, per-utterance losses, dx [B,T,C], dparams, counts) -- counts[k] = the sum of the two expected arc
    counts (normaliser + numerator, scaled like dparams) whose DIFFERENCE dparams[k] is: what a relative accuracy of
    the two forward_score gradients translates to when they nearly cancel (a blank arc that both use all the time)."""
    x = np.asarray(x, dtype=np.float64)
    params = np.asarray(params, dtype=np.float64)
    B = x.shape[0]
    tsrc, tdst, tlab, tstart, taccept, tn = transitions
    losses, dx, dp = np.zeros(B), np.zeros_like(x), np.zeros_like(params)
    counts = np.zeros_like(params)
    for b in range(B):
        nsrc, ndst, nlab, nwid, nstart, naccept, nn = numerators[b]
        nwid = np.asarray(nwid, dtype=np.int64)
        zn, gn, an = lattice_forward_backward_eps(x[b], nsrc, ndst, nlab, params[nwid], nstart, naccept, nn)
        zd, gd, ad = lattice_forward_backward_eps(x[b], tsrc, tdst, tlab, params, tstart, taccept, tn)
        sc = scales[b]
        losses[b] = sc * (zd - zn)
        dx[b] = sc * (gd - gn) / B
        dp += sc * ad / B
        np.add.at(dp, nwid, -sc * an / B)
        counts += sc * ad / B
        np.add.at(counts, nwid, sc * an / B)
    return float(losses.mean()), losses, dx, dp, counts
