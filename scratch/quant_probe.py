"""CPU: what float32 storage of the log-domain state vector (relative to its running maximum) costs on the posteriors of
the n-gram benchmark's numerator graph -- float64 arithmetic throughout, only the per-frame store is quantised."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from oracle import recurrences as OR
from gtn_applications_amd.criterions import transducer as TR
N, T, L = 81, 250, 44
rs = np.random.RandomState(120)
tokens = [(i,) for i in range(N)]
crit = TR.Transducer(tokens, {i: i for i in range(N)}, ngram=2, reduction="mean", blank="optional", allow_repeats=False)
C = N + 1
x = rs.randn(T, C).astype(np.float32).astype(np.float64)
target = rs.randint(0, N, size=L).tolist()
params = (0.3 * rs.randn(crit.transition_params.numel())).astype(np.float32)
crit.tokens.arc_sort(True)
ali, wid = TR._alignment_graph(target, crit.tokens, crit.lexicon, crit.transitions)
a = ali.arrays()
st, ac = np.flatnonzero(a["start"]).tolist(), np.flatnonzero(a["accept"]).tolist()
wid = np.asarray(wid)
w = np.where(wid >= 0, params[np.maximum(wid, 0)].astype(np.float64), 0.0) + a["weight"] if "weight" in a else np.where(wid >= 0, params[np.maximum(wid, 0)].astype(np.float64), 0.0)
Q = len(a["start"])
print("states", Q, "arcs", len(a["src"]), "eps", int((a["ilabel"] < 0).sum()))
z, dx, _ = OR.lattice_forward_backward_eps(x, a["src"], a["dst"], a["ilabel"], w, st, ac, Q)
# quantised variant: monkeypatch by wrapping np.logaddexp.at?  simpler: re-implement the two sweeps here with a store hook
src, dst, lab = a["src"], a["dst"], a["ilabel"]
ie, il = np.flatnonzero(lab < 0), np.flatnonzero(lab >= 0)
ls, ld, ll, lw = src[il], dst[il], lab[il], w[il]
es, ed, ew = src[ie], dst[ie], w[ie]
din, dout = OR._eps_depths(Q, es, ed)
fg = [np.flatnonzero(din[es] == d) for d in range(int(din.max()) + 1)] if len(ie) else []
bg = [np.flatnonzero(dout[ed] == d) for d in range(int(dout.max()) + 1)] if len(ie) else []
NEG = -np.inf
def run(quant):
    alpha = np.full((T + 1, Q), NEG); beta = np.full((T + 1, Q), NEG)
    def store(v):
        if quant is None: return v
        m = v.max()
        return quant(v - m) + m
    with np.errstate(all="ignore"):
        alpha[0, st] = 0.0
        for g in fg: np.logaddexp.at(alpha[0], ed[g], alpha[0][es[g]] + ew[g])
        for t in range(T):
            np.logaddexp.at(alpha[t + 1], ld, alpha[t, ls] + x[t, ll] + lw)
            for g in fg: np.logaddexp.at(alpha[t + 1], ed[g], alpha[t + 1][es[g]] + ew[g])
            alpha[t + 1] = store(alpha[t + 1])
        beta[T, ac] = 0.0
        for g in bg: np.logaddexp.at(beta[T], es[g], beta[T][ed[g]] + ew[g])
        for t in range(T - 1, -1, -1):
            np.logaddexp.at(beta[t], ls, beta[t + 1, ld] + x[t, ll] + lw)
            for g in bg: np.logaddexp.at(beta[t], es[g], beta[t][ed[g]] + ew[g])
            beta[t] = store(beta[t])
        lz = np.logaddexp.reduce(alpha[T, ac])
        d = np.zeros((T, C))
        for t in range(T):
            gg = np.exp(alpha[t, ls] + x[t, ll] + lw + beta[t + 1, ld] - lz)
            np.add.at(d[t], ll, np.where(np.isfinite(gg), gg, 0.0))
    return lz, d, alpha, beta
z0, d0, al, be = run(None)
print("exact replica vs oracle", abs(z0 - z), np.abs(d0 - dx).max())
z1, d1, _, _ = run(lambda v: v.astype(np.float32).astype(np.float64))
print("float32 store rel. to max: logZ err %.3g  posterior err %.3g" % (abs(z1 - z), np.abs(d1 - dx).max()))
# how far below the frame maximum are the states that carry the posteriors?
post = al + be - z0
rel = al - al.max(axis=1, keepdims=True)
mask = post > np.log(1e-3)
print("alpha - max(alpha) of states with posterior > 1e-3: median %.1f  min %.1f" % (np.median(rel[mask]), rel[mask].min()))
