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
