"""scratch: CPU simulation of the CTC chain's lane exponents when they are predicted from a checkpoint `lag` blocks old
(the helper wave of ctc_mitm.h) -- true alpha in log2 per lane and frame, the exponents the helper would choose, and
the mantissa range 2^(true - e) of the lanes that matter."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
kGap, EMPTY = 5, -(1 << 28)

def model_scores(rs, T, C, L, boost, noise, wrong):
    x = (noise * rs.randn(T, C))
    y = rs.randint(0, C - 1, size=L)
    cuts = np.sort(rs.choice(np.arange(1, T), size=2 * L, replace=False))
    lab = np.full(T, C - 1)
    for i in range(L):
        lab[cuts[2 * i]:cuts[2 * i + 1]] = y[i]
    flip = rs.rand(T) < wrong
    lab = np.where(flip, rs.randint(0, C, size=T), lab)
    x[np.arange(T), lab] += boost
    x = x - np.log(np.exp(x).sum(1, keepdims=True))
    return x, y

def lse2(*a):
    a = np.stack(a); m = a.max(0); m2 = np.where(np.isfinite(m), m, 0)
    return m2 + np.log2(np.exp2(a - m2).sum(0)) if True else None

def run(x, y, blank, lag=2, clip_lo=-40, clip_hi=16, trend=True, verbose=False, stay=False, margin=0):
    T, C = x.shape; L = len(y)
    NL = 64
    xs = x * np.log2(np.e)
    # per-frame reference as the kernel's: rint(max over target labels (and the blank? -> labels only) of the score)
    cols = np.full(NL, -1); cols[:L] = y
    fl = np.full((T, NL), -np.inf); fl[:, :L] = xs[:, y]
    fb = np.full((T, NL), -np.inf); fb[:, :L + 1] = xs[:, [blank]]
    r = np.rint(np.maximum(fl[:, :L].max(1), fb[:, 0]))
    fl -= r[:, None]; fb -= r[:, None]
    skip = np.zeros(NL, bool); skip[1:L] = y[1:] != y[:-1]
    # true recursion in log2 (relative to the references)
    pb = np.full(NL, -np.inf); pl = np.full(NL, -np.inf); pb[0] = 0.0
    NB = (T + 15) // 16
    alpha_b = np.zeros((T + 1, NL)); alpha_l = np.zeros((T + 1, NL))
    alpha_b[0], alpha_l[0] = pb, pl
    for t in range(T):
        q = np.concatenate([[-np.inf], pl[:-1]])
        nb = fb[t] + np.logaddexp2(pb, q)
        nl = fl[t] + np.logaddexp2(np.logaddexp2(pl, pb), np.where(skip, q, -np.inf))
        pb, pl = nb, nl
        alpha_b[t + 1], alpha_l[t + 1] = pb, pl
    # per-block stay-path sums (log2 of the product of a lane's factors over the block's frames)
    NBk = (T + 15) // 16
    sbk = np.zeros((NBk, NL)); slk = np.zeros((NBk, NL))
    for mm in range(NBk):
        sbk[mm] = np.maximum(fb[16 * mm:16 * mm + 16].sum(0), -1e4); slk[mm] = np.maximum(fl[16 * mm:16 * mm + 16].sum(0), -1e4)
    def ownbl_of(t):
        ob = np.where(np.isfinite(alpha_b[t]), np.floor(alpha_b[t]) + 1, EMPTY); ol = np.where(np.isfinite(alpha_l[t]), np.floor(alpha_l[t]) + 1, EMPTY)
        return ob, ol
    def own_of(t):  # exponent of max(pb, pl) as frexp gives it
        v = np.maximum(alpha_b[t], alpha_l[t])
        return np.where(np.isfinite(v), np.floor(v) + 1, EMPTY).astype(np.int64)
    def clamp_prefix(pred):
        lane = np.arange(NL)
        return np.maximum.accumulate(pred + kGap * lane) - kGap * lane
    # helper schedule
    e = np.zeros((NB, NL), np.int64)
    own = np.full(NL, EMPTY, np.int64); own[0] = 1
    dpb = np.zeros(NL, np.int64); src_last = -1
    def look(src):
        nonlocal own, dpb, src_last
        now = own_of(src * 16) if src > 0 else own_of(0)
        span = src - src_last
        ok = (now > EMPTY) & (own > EMPTY) & (src_last >= 0)
        d = np.where(span == 2, (now - own) >> 1, now - own)
        dpb = np.where(ok, d, 0) if trend else np.zeros(NL, np.int64)
        own, src_last = now, src
    def emit(m, lg):
        if stay:
            ob, ol = ownbl_of(src_last * 16) if src_last >= 0 else ownbl_of(0)
            sb = sbk[max(src_last, 0):m].sum(0); sl = slk[max(src_last, 0):m].sum(0)
            pb_ = np.where(ob > EMPTY, ob + np.floor(sb), EMPTY); pl_ = np.where(ol > EMPTY, ol + np.floor(sl), EMPTY)
            pred = np.maximum(pb_, pl_)
            pred = np.where(pred > EMPTY / 2, pred + margin * lg, EMPTY).astype(np.int64)
        else:
            pred = np.where(own > EMPTY, own + np.clip(dpb * lg, clip_lo * lg, clip_hi * lg), EMPTY)
        e[m] = clamp_prefix(pred)
    emit(0, 0)
    m = 1
    while m < NB and m <= 3:
        look(m - 1); emit(m, 1); m += 1
    while m < NB:
        look(m - lag); emit(m, lag)
        if m + 1 < NB: emit(m + 1, lag + 1)
        m += 2
    # mantissa range over the frames of every block, lanes within 60 bits of the frame's maximum
    worst_lo, worst_hi = 0.0, 0.0
    bad = []
    for m in range(NB):
        for t in range(16 * m, min(16 * m + 17, T + 1)):
            v = np.maximum(alpha_b[t], alpha_l[t])
            top = v.max()
            sig = v > top - 60
            mant = v[sig] - e[m][sig]
            lo, hi = mant.min(), mant.max()
            # every finite lane: overflow is fatal wherever it happens
            allm = v[np.isfinite(v)] - e[m][np.isfinite(v)]
            hi = max(hi, allm.max())
            worst_lo, worst_hi = min(worst_lo, lo), max(worst_hi, hi)
            if lo < -120 or hi > 120: bad.append((m, t, lo, hi))
    return worst_lo, worst_hi, bad, e

if __name__ == "__main__":
    C, L, T = 100, 44, 1000
    for boost, noise, wrong in [(0, 1, 0), (3, 1, .3), (8, 1, .1), (12, 2, .1), (20, 1, .02)]:
        rs = np.random.RandomState(int(boost * 10 + wrong * 100))
        for lag, trend in [(1, False), (2, "stay"), (3, "stay"), (2, "stay8"), (3, "stay8")]:
            res = []
            for u in range(6):
                x, y = model_scores(rs, T, C, L, boost, noise, wrong)
                if trend in ("stay", "stay8"):
                    lo, hi, bad, _ = run(x, y, C - 1, lag=lag, stay=True, margin=8 if trend == "stay8" else 0)
                else:
                    lo, hi, bad, _ = run(x, y, C - 1, lag=lag, trend=trend)
                res.append((lo, hi, len(bad)))
            print("boost %4.1f noise %3.1f wrong %.2f | lag %d trend %s | mantissa log2 range: lo %7.1f hi %7.1f | frames out of range %s" % (
                boost, noise, wrong, lag, trend, min(r[0] for r in res), max(r[1] for r in res), [r[2] for r in res]))
