"""scratch: where does the back-off transducer's parameter gradient go wrong at T = 250?  Denominator (transitions alone)
and numerator (transitions o alignments) separately, against the float64 epsilon-aware recurrence, for several T."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd import engine as E, graph as G
from gtn_applications_amd.criterions import transducer as TR
from oracle import recurrences as OR

lit = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_literals.json")))["backoff_transitions"]
N = lit["N"]
g = G.Graph(True)
for n in range(8):
    g.add_node(n in lit["start"], n in lit["accept"])
for a in lit["arcs"]:
    g.add_arc(*a)
toks = [(n,) for n in range(N)]
crit = TR.Transducer(toks, {n: n for n in range(N)}, blank="optional", allow_repeats=False, transitions=g, reduction="mean")
dev = torch.device("cuda")
for T, B in ((6, 1), (14, 2), (32, 2), (64, 4), (250, 1), (250, 16)):
    rs = np.random.RandomState(5)
    x = rs.randn(B, T, N + 1).astype(np.float32)
    L = max(2, min(40, T // 3))
    targets = [rs.randint(0, N, size=L).tolist() for _ in range(B)]
    params = (0.3 * rs.randn(crit.transition_params.numel())).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    pt = torch.from_numpy(params).to(dev)
    ones = torch.ones(B, device=dev)
    # denominator: the transition model alone
    pack = TR._transitions_pack(crit.transitions, B, N + 1, dev)
    st = E.lattice_forward(xt, pack, weights=pt, need_beta=True)
    dx = torch.zeros_like(xt); dW = torch.zeros_like(pt)
    E.lattice_grad(st, ones, coef_w=ones, gout=None, dx=dx, accumulate=False, dW=dW)
    a = crit.transitions.arrays()
    st_n, ac_n = np.flatnonzero(a["start"]).tolist(), np.flatnonzero(a["accept"]).tolist()
    zs, gx, gw = [], np.zeros_like(x, dtype=np.float64), np.zeros(len(params))
    for b in range(B):
        z, g1, a1 = OR.lattice_forward_backward_eps(x[b], a["src"], a["dst"], a["ilabel"], params, st_n, ac_n, len(a["start"]))
        zs.append(z); gx[b] = g1; gw += a1
    print("T=%d B=%d denominator: logz rel %.2e | dx max abs %.2e | dW max rel %.2e (at %d: got %.6g want %.6g)" % (
        T, B, np.abs(st.logz.cpu().numpy() - zs).max() / np.abs(zs).max(), np.abs(dx.cpu().numpy() - gx).max(),
        (np.abs(dW.cpu().numpy() - gw) / np.maximum(np.abs(gw), 1e-9)).max(), int(np.argmax(np.abs(dW.cpu().numpy() - gw) / np.maximum(np.abs(gw), 1e-9))),
        dW.cpu().numpy()[int(np.argmax(np.abs(dW.cpu().numpy() - gw) / np.maximum(np.abs(gw), 1e-9)))], gw[int(np.argmax(np.abs(dW.cpu().numpy() - gw) / np.maximum(np.abs(gw), 1e-9)))]))
    eps_idx = np.flatnonzero(a["ilabel"] < 0)
    rel = np.abs(dW.cpu().numpy() - gw) / np.maximum(np.abs(gw), 1e-9)
    print("    eps arcs max rel %.2e | labelled arcs max rel %.2e" % (rel[eps_idx].max(), np.delete(rel, eps_idx).max()))
    # numerator
    crit.tokens.arc_sort(True)
    graphs, wids = zip(*[TR._alignment_graph(t, crit.tokens, crit.lexicon, crit.transitions) for t in targets])
    npack = E.PackedLattice.from_graphs(list(graphs), N + 1, dev, wids=[np.asarray(w, np.int32) for w in wids])
    st = E.lattice_forward(xt, npack, weights=pt, need_beta=True)
    dx = torch.zeros_like(xt); dW = torch.zeros_like(pt)
    E.lattice_grad(st, ones, coef_w=ones, gout=None, dx=dx, accumulate=False, dW=dW)
    zs, gx, gw = [], np.zeros_like(x, dtype=np.float64), np.zeros(len(params))
    for b in range(B):
        aa = graphs[b].arrays()
        wid = np.asarray(wids[b], np.int64)
        z, g1, a1 = OR.lattice_forward_backward_eps(x[b], aa["src"], aa["dst"], aa["ilabel"], params[wid], np.flatnonzero(aa["start"]).tolist(),
                                                    np.flatnonzero(aa["accept"]).tolist(), len(aa["start"]))
        zs.append(z); gx[b] = g1; np.add.at(gw, wid, a1)
    rel = np.abs(dW.cpu().numpy() - gw) / np.maximum(np.abs(gw), 1e-9)
    print("T=%d B=%d numerator:   logz rel %.2e | dx max abs %.2e | dW max rel %.2e | eps arcs %.2e labelled %.2e" % (
        T, B, np.abs(st.logz.cpu().numpy() - zs).max() / np.abs(zs).max(), np.abs(dx.cpu().numpy() - gx).max(), rel.max(),
        rel[eps_idx].max(), np.delete(rel, eps_idx).max()))
