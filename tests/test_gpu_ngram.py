"""GPU parity of the epsilon / high-degree transition path AT THE SIZE THE REFERENCE BENCHMARKS IT (-m gpu):
benchmarks/transducer_benchmark.py:55-119 -- N = 81 tokens, T = 250, L = 44, n-gram transition models of order 0, 1, 2
under a CTC-style token graph (blank optional, no repeats) and an ASG-style one (no blank, repeats) -- plus the pruned
back-off model of tests/transducer_test.py:534-566 (tests/trans_backoff_test.txt, epsilon arcs between inner nodes)
at T = 250.  Loss, emission gradient AND transition-parameter gradient of every utterance against the float64
epsilon-aware recurrence (oracle/recurrences.py::lattice_forward_backward_eps, tied to the graph oracle by
tests/test_oracle.py), at the bar every other path is held to:  |got - want| <= 1e-4 |want| + 5e-5 scale,
scale = the factor a posterior is multiplied with (scale_b / B); a transition parameter's gradient -- the difference
of two sums of thousands of posteriors -- gets the allowance check_dparams states.  Viterbi decodes of the same batches
against the max-plus recurrence.  Worst cases go to gpurun_out/parity_r04_ngram.json.
criterions/transducer.py:32-58,279-288.  Nothing here reads /root/reference."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import recurrences as OR  # noqa: E402

from test_gpu_configs import check, STATS, RTOL  # noqa: E402,F401  (the same bar, the same bookkeeping)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _gpu_and_stats():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    yield
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_r04_ngram.json"), "w") as f:
            json.dump({k: v for k, v in STATS.items() if k.startswith("ngram") or k.startswith("backoff")}, f, indent=1)
    except OSError:
        pass


COUNT_RTOL = 2e-5


def check_dparams(name, got, want, counts, scale):
    """A transition parameter's gradient is the DIFFERENCE of two expected arc counts -- the normaliser's and the
    numerator's (transducer.py:286-290,329-336), each summed over B (T + 1) posteriors of scale `scale`.  The bar of
    every other gradient, 1e-4 |want| + 5e-5 scale, plus what a relative accuracy of 2e-5 of the two counts (5x inside
    the north star's 1e-4 on a forward_score's gradient) leaves of a difference when they nearly cancel: a blank arc
    that both use in every frame has counts of ~600 scales and a difference of one.  Measured: 3e-6 .. 6e-6 of the
    counts (float32 alpha / beta relative to a per-chunk double offset; scratch/backoff_probe.py), which is also all the
    error there is -- against 5e-5 scale ALONE the back-off model's blank arcs are 20x over, and float32 outputs
    cannot be otherwise (the counts are 2.0, float32 resolves 1.2e-7 of them = 4e-5 scale)."""
    got, want, counts = (np.asarray(v, dtype=np.float64) for v in (got, want, counts))
    assert got.shape == want.shape and np.isfinite(got).all(), name
    err = np.abs(got - want)
    tol = RTOL * np.abs(want) + 5e-5 * scale + COUNT_RTOL * counts
    k = int(np.argmax(err / tol))
    rec = STATS.setdefault(name, dict(max_err_over_tol=0.0, max_abs_err=0.0, max_abs_err_over_scale=0.0,
                                      max_err_over_counts=0.0, elements=0))
    rec["max_err_over_tol"] = max(rec["max_err_over_tol"], float((err / tol).max()))
    rec["max_abs_err"] = max(rec["max_abs_err"], float(err.max()))
    rec["max_abs_err_over_scale"] = max(rec["max_abs_err_over_scale"], float(err.max() / scale))
    rec["max_err_over_counts"] = max(rec["max_err_over_counts"], float((err / np.maximum(counts, 1e-300)).max()))
    rec["elements"] += int(err.size)
    assert err[k] <= tol[k], f"{name}: |{got[k]:.9g} - {want[k]:.9g}| = {err[k]:.3g} > {tol[k]:.3g} at {k} (counts {counts[k]:.3g})"


def _graph_arrays(g):
    a = g.arrays()
    return a, np.flatnonzero(a["start"]).tolist(), np.flatnonzero(a["accept"]).tolist()


def _oracle(crit, x, targets, params):
    """float64 loss / dx / dparams of Transducer(transitions) from the acceptors the HOST LIBRARY builds (its graph
    algebra is pinned against the oracle's in tests/test_host_library.py), swept by the oracle's recurrence."""
    from gtn_applications_amd.criterions import transducer as TR

    crit.tokens.arc_sort(True)
    ta, tstart, taccept = _graph_arrays(crit.transitions)
    nums = []
    for t in targets:
        ali, wid = TR._alignment_graph(list(t), crit.tokens, crit.lexicon, crit.transitions)
        a, st, ac = _graph_arrays(ali)
        nums.append((a["src"], a["dst"], a["ilabel"], np.asarray(wid), st, ac, len(a["start"])))
    trans = (ta["src"], ta["dst"], ta["ilabel"], tstart, taccept, len(ta["start"]))
    return OR.transducer_transitions_loss_grad(x, nums, trans, params, [1.0 / len(t) for t in targets])


def _run(crit, x, targets, params):
    with torch.no_grad():
        crit.transition_params.copy_(torch.from_numpy(params))
    crit.cuda()
    crit.transition_params.grad = None
    xg = torch.from_numpy(x).cuda().requires_grad_(True)
    loss = crit(xg, [torch.tensor(t) for t in targets])
    loss.backward()
    return loss.item(), xg.grad.cpu().numpy(), crit.transition_params.grad.cpu().numpy()


@pytest.mark.parametrize("kind", ["ctc", "asg"])
@pytest.mark.parametrize("ngram", [0, 1, 2])
def test_ngram_transitions_at_the_reference_benchmark_size(kind, ngram):
    """benchmarks/transducer_benchmark.py:55-119 at B = 16.  (ngram = 0: no transition model -- the log_softmax route,
    checked against the epsilon-free recurrence.)"""
    from gtn_applications_amd.criterions import transducer as TR

    N, T, L, B = 81, 250, 44, 16
    rs = np.random.RandomState(100 + 10 * ngram + (kind == "asg"))
    tokens = [(i,) for i in range(N)]
    g2i = {i: i for i in range(N)}
    kw = dict(blank="optional", allow_repeats=False) if kind == "ctc" else {}
    C = N + (1 if kind == "ctc" else 0)
    x = rs.randn(B, T, C).astype(np.float32)
    targets = rs.randint(0, N, size=(B, L)).tolist()
    crit = TR.Transducer(tokens, g2i, ngram=ngram, reduction="mean", **kw)
    name = f"ngram{ngram}_{kind}"
    if ngram == 0:
        from oracle import criteria as OC

        xg = torch.from_numpy(x).cuda().requires_grad_(True)
        loss = crit(xg, [torch.tensor(t) for t in targets])
        loss.backward()
        dx = xg.grad.cpu().numpy()
        crit.tokens.arc_sort(True)
        lp = OC.log_softmax(x.astype(np.float64), 2)
        losses = []
        for b in range(B):
            a, st, ac = _graph_arrays(TR._alignment_graph(targets[b], crit.tokens, crit.lexicon, None)[0])
            logz, g, _ = OR.lattice_forward_backward(lp[b], a["src"], a["dst"], a["ilabel"], np.zeros(len(a["src"])), st, ac,
                                                     len(a["start"]))
            din = -g / (L * B)
            losses.append(-logz / L)
            check(name + "_dx", dx[b], din - np.exp(lp[b]) * din.sum(axis=1, keepdims=True), 1.0 / (L * B))
        check(name + "_loss", [loss.item()], [np.mean(losses)], 0.0)
        return
    params = (0.3 * rs.randn(crit.transition_params.numel())).astype(np.float32)
    want_loss, _, want_dx, want_dp, counts = _oracle(crit, x, targets, params)
    loss, dx, dp = _run(crit, x, targets, params)
    check(name + "_loss", [loss], [want_loss], 0.0)
    check(name + "_dx", dx, want_dx, 1.0 / (L * B))
    check_dparams(name + "_dparams", dp, want_dp, counts, 1.0 / (L * B))
    # Transducer.viterbi (transducer.py:199-234) of the same batch: the best frame path under the transition model by
    # the max-plus recurrence, then the token graph's collapse (blank = C - 1 for the CTC-style graph)
    import itertools

    got = crit.viterbi(torch.from_numpy(x).cuda())
    xd = x.astype(np.float64)
    if ngram == 1:
        frames = (xd + params[None, None, :C]).argmax(axis=2)
    else:
        W = np.zeros((C + 1, C))
        W[0] = params[:C]
        W[1:] = params[C:C + C * C].reshape(C, C).T
        xd = xd.copy()
        xd[:, -1, :] += params[C + C * C + 1:]
        frames = np.array([OR.dense_viterbi(xd[b], W) for b in range(B)])
    blank = C - 1 if kind == "ctc" else -1
    for b in range(B):
        assert got[b].tolist() == [k for k, _ in itertools.groupby(frames[b].tolist()) if k != blank], (name, b)


def _numerator_formats(loss, B, T):
    """fmt[b] of the numerator sweeps behind a Transducer loss: 0 log domain, 1 probability domain"""
    import ctypes

    from gtn_applications_amd import _native as N

    num = loss.grad_fn.aux[2]
    off = ctypes.c_int64()
    N.check(N.lib.wfl_lattice_formats_offset(ctypes.byref(num.pack.desc), T, ctypes.byref(off)))
    torch.cuda.synchronize()
    return num.alpha[off.value:off.value + B].view(torch.int32).cpu().tolist()


@pytest.mark.parametrize("dense_den", [True, False])
@pytest.mark.parametrize("kind", ["ctc", "asg"])
def test_epsilon_acceptors_are_swept_in_the_probability_domain(kind, dense_den, monkeypatch):
    """csrc/lattice_kernels.hip run_chain_prob / eps_closure: the bigram model's alignment acceptors as the reference
    builds them (transducer.py:262-290: alignments o make_transitions_graph(2, N), epsilon arcs into and out of the
    back-off state) through the GENERAL lattice path -- TransducerLossFunction.apply, none of the module's short cuts.
    Their numerator sweeps run in the fp64 probability domain with an in-frame epsilon closure (formats say so), pass
    the certificate (alpha before the closure x beta = Z at every 8th slot), and loss, emission gradient and
    transition-parameter gradient -- epsilon arcs' included -- meet the float64 epsilon-aware recurrence.
    dense_den = False: the DENOMINATOR -- the bigram model itself, 83 states with 81 arcs into each and the back-off
    state's epsilon arcs -- takes the lattice engine as well instead of the dense short cut: run_chain_prob_general
    (rows of 16 lanes per state, more such states than the workgroup has rows)."""
    from gtn_applications_amd import engine as E
    from gtn_applications_amd.criterions import transducer as TR

    monkeypatch.setattr(TR, "_DENSE_NGRAM", dense_den)
    N, T, L, B = 81, 250, 44, 8
    rs = np.random.RandomState(300 + (kind == "asg"))
    kw = dict(blank="optional", allow_repeats=False) if kind == "ctc" else {}
    C = N + (1 if kind == "ctc" else 0)
    x = rs.randn(B, T, C).astype(np.float32)
    targets = rs.randint(0, N, size=(B, L)).tolist()
    crit = TR.Transducer([(i,) for i in range(N)], {i: i for i in range(N)}, ngram=2, reduction="mean", **kw)
    params = (0.3 * rs.randn(crit.transition_params.numel())).astype(np.float32)
    want_loss, _, want_dx, want_dp, counts = _oracle(crit, x, targets, params)
    with torch.no_grad():
        crit.transition_params.copy_(torch.from_numpy(params))
    crit.cuda()
    crit.transition_params.grad = None
    crit.tokens.arc_sort(True)
    xg = torch.from_numpy(x).cuda().requires_grad_(True)
    loss = TR.TransducerLossFunction.apply(xg, [torch.tensor(t) for t in targets], crit.tokens, crit.lexicon,
                                           crit.transition_params, crit.transitions, crit.reduction)
    assert _numerator_formats(loss, B, T) == [1] * B
    if not dense_den:
        assert E.lattice_formats(loss.grad_fn.aux[3]).tolist() == [1] * B
    loss.backward()
    name = f"ngram2_{kind}_general" + ("" if dense_den else "_lattice_den")
    check(name + "_loss", [loss.item()], [want_loss], 0.0)
    check(name + "_dx", xg.grad.cpu().numpy(), want_dx, 1.0 / (L * B))
    check_dparams(name + "_dparams", crit.transition_params.grad.cpu().numpy(), want_dp, counts, 1.0 / (L * B))


def test_backoff_transitions_at_benchmark_length(golden_dir):
    """tests/transducer_test.py:534-566's pruned back-off model (8 nodes, 36 arcs, epsilon back-off arcs between inner
    nodes: in-frame epsilon closure over several levels) at T = 250, B = 16, with targets of its three tokens."""
    from gtn_applications_amd import graph as G
    from gtn_applications_amd.criterions import transducer as TR

    lit = json.load(open(os.path.join(golden_dir, "reference_literals.json")))["backoff_transitions"]
    N, T, B = lit["N"], 250, 16
    g = G.Graph(True)
    for n in range(8):
        g.add_node(n in lit["start"], n in lit["accept"])
    for a in lit["arcs"]:
        g.add_arc(*a)
    rs = np.random.RandomState(5)
    toks = [(n,) for n in range(N)]
    crit = TR.Transducer(toks, {n: n for n in range(N)}, blank="optional", allow_repeats=False, transitions=g,
                         reduction="mean")
    x = rs.randn(B, T, N + 1).astype(np.float32)
    targets = [rs.randint(0, N, size=rs.randint(20, 45)).tolist() for _ in range(B)]
    params = (0.3 * rs.randn(crit.transition_params.numel())).astype(np.float32)
    want_loss, _, want_dx, want_dp, counts = _oracle(crit, x, targets, params)
    loss, dx, dp = _run(crit, x, targets, params)
    check("backoff_loss", [loss], [want_loss], 0.0)
    scale = max(1.0 / len(t) for t in targets) / B
    check("backoff_dx", dx, want_dx, scale)
    check_dparams("backoff_dparams", dp, want_dp, counts, scale)


@pytest.mark.parametrize("ngram", [1, 2])
@pytest.mark.parametrize("kind", ["ctc", "asg"])
def test_dense_ngram_routes_equal_the_general_path(ngram, kind):
    """TransducerLoss with the unigram / bigram model takes a short cut (criterions/transducer.py::_unigram_route: the
    transition-free step on x + p; ::_bigram_route: the ASG step on emissions that carry the end arcs' scores and an
    alignment acceptor without its final epsilon arc).  TransducerLossFunction.apply called directly takes none: the
    general lattice path with the transition graph as the reference builds it (transducer.py:262-290).  Same loss, same
    gradients -- also through Transducer.prepare()."""
    from gtn_applications_amd.criterions import transducer as TR

    torch.manual_seed(3)
    N, T, L, B = 23, 60, 9, 5
    tokens = [(i,) for i in range(N)]
    kw = dict(blank="optional", allow_repeats=False) if kind == "ctc" else {}
    crit = TR.Transducer(tokens, {i: i for i in range(N)}, ngram=ngram, reduction="mean", **kw).cuda()
    with torch.no_grad():
        crit.transition_params.normal_(0, 0.5)
    C = N + (1 if kind == "ctc" else 0)
    x0 = torch.randn(B, T, C).cuda()
    targets = [torch.randint(N, (L,)) for _ in range(B)]

    def run(how):
        x = x0.clone().requires_grad_(True)
        crit.transition_params.grad = None
        if how == "direct":
            crit.tokens.arc_sort(True)
            loss = TR.TransducerLossFunction.apply(x, targets, crit.tokens, crit.lexicon, crit.transition_params,
                                                   crit.transitions, crit.reduction)
        elif how == "prepared":
            loss = crit(x, crit.prepare(targets))
        else:
            loss = crit(x, targets)
        loss.backward()
        return loss.detach().cpu(), x.grad.cpu(), crit.transition_params.grad.cpu().clone()

    l0, dx0, dp0 = run("direct")
    for how in ("module", "prepared"):
        l1, dx1, dp1 = run(how)
        torch.testing.assert_close(l1, l0, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(dx1, dx0, rtol=1e-4, atol=2e-6)
        torch.testing.assert_close(dp1, dp0, rtol=1e-4, atol=2e-5)
    # one of the two gradients alone (the bigram's end arcs take theirs from the emission gradient's last frame)
    crit.transition_params.grad = None
    crit(x0.clone(), targets).backward()
    torch.testing.assert_close(crit.transition_params.grad.cpu(), dp0, rtol=1e-4, atol=2e-5)
    crit.transition_params.requires_grad_(False)
    try:
        x = x0.clone().requires_grad_(True)
        crit(x, targets).backward()
        torch.testing.assert_close(x.grad.cpu(), dx0, rtol=1e-4, atol=2e-6)
    finally:
        crit.transition_params.requires_grad_(True)


@pytest.mark.parametrize("ngram", [1, 2])
def test_dense_ngram_routes_on_degenerate_batches(ngram):
    """The short cuts of TransducerLoss (unigram / bigram) on the shapes the general path also has to get right: one
    frame, an empty target next to ordinary ones, emissions that live on the CPU."""
    from gtn_applications_amd.criterions import transducer as TR

    torch.manual_seed(11)
    N = 9
    tokens = [(i,) for i in range(N)]
    crit = TR.Transducer(tokens, {i: i for i in range(N)}, ngram=ngram, reduction="mean", blank="optional",
                         allow_repeats=False).cuda()
    with torch.no_grad():
        crit.transition_params.normal_(0, 0.5)
    C = N + 1
    cases = [(1, [[3]]), (1, [[]]), (6, [[], [2, 2, 5], [7]]), (4, [[1, 2, 3, 4]])]
    for T, tg in cases:
        B = len(tg)
        targets = [torch.tensor(t, dtype=torch.long) for t in tg]
        x0 = torch.randn(B, T, C)
        outs = []
        for how in ("direct", "module", "module_cpu_input"):
            x = (x0.clone() if how == "module_cpu_input" else x0.clone().cuda()).requires_grad_(True)
            crit.transition_params.grad = None
            if how == "direct":
                crit.tokens.arc_sort(True)
                loss = TR.TransducerLossFunction.apply(x, targets, crit.tokens, crit.lexicon, crit.transition_params,
                                                       crit.transitions, crit.reduction)
            else:
                loss = crit(x, targets)
            loss.backward()
            assert x.grad.device == x.device
            outs.append((loss.detach().cpu(), x.grad.cpu(), crit.transition_params.grad.cpu().clone()))
        for l1, dx1, dp1 in outs[1:]:
            torch.testing.assert_close(l1, outs[0][0], rtol=2e-5, atol=2e-5)
            torch.testing.assert_close(dx1, outs[0][1], rtol=1e-4, atol=2e-6)
            torch.testing.assert_close(dp1, outs[0][2], rtol=1e-4, atol=2e-5)
