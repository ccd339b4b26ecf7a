"""GPU parity at the sizes BASELINE.json names (-m gpu): every utterance of each single-GPU configuration
against the float64 oracle, at the north-star tolerance.

  cfg2  CTC        T=1000 C=100  B=128 L=44     (benchmarks/ctc_benchmark.py)
  cfg3  ASG        T=1000 C=100  B=128 L=44     (benchmarks/asg_benchmark.py), learned (C+1)xC transitions
  cfg4  Transducer T=800  C=1001 B=64, the 1000 word pieces of benchmarks/word_pieces_tokens_1000.txt
        (benchmarks/transducer_benchmark.py:18-53: blank optional, no repeats, reduction mean)
  cfg5  CTC        T=2000 C=512  B=128 -- one GPU's shard of the 8-GPU configuration

Tolerance (north_star: 1e-4 relative on log-semiring loss / grad): every gradient element within
1e-4 * |expected| + 5e-5 * |coef_b| (the reference's own equivalence bar: tests/transducer_test.py:275-316 uses
atol 1e-5 at B=5, i.e. 5e-5 of a posterior's scale), where coef_b = scale_b / B is the factor the reference multiplies a
posterior (a number in [0,1]) with (ctc.py:87, asg.py:171-179, transducer.py:329-336) -- i.e. posteriors are
right to 5e-5 of their scale; per-utterance losses to 1e-4 relative.  The measured worst cases are written to
gpurun_out/parity_r06.json.  Nothing here reads /root/reference."""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import criteria as OC  # noqa: E402
from oracle import recurrences as OR  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-4
ATOL_SCALE = 2e-5  # (the worst at-size case of round 5 sits at 1.0e-5 of scale: profiles/r05_parity_worst_cases.json)
STATS = {}


@pytest.fixture(scope="module", autouse=True)
def _gpu_and_stats():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    yield
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_r06.json"), "w") as f:
            json.dump(STATS, f, indent=1)
    except OSError:
        pass


def check(name, got, want, scale):
    """|got - want| <= RTOL * |want| + ATOL_SCALE * scale elementwise; records the worst case."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape and np.isfinite(got).all(), name
    err = np.abs(got - want)
    tol = RTOL * np.abs(want) + ATOL_SCALE * scale
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(tol > 0, err / np.where(tol > 0, tol, 1.0), np.where(err > 0, np.inf, 0.0))
    k = int(np.argmax(ratio))
    rec = STATS.setdefault(name, dict(max_err_over_tol=0.0, max_abs_err=0.0, max_abs_err_over_scale=0.0, elements=0))
    rec["max_err_over_tol"] = max(rec["max_err_over_tol"], float(ratio.flat[k]))
    rec["max_abs_err"] = max(rec["max_abs_err"], float(err.max()))
    if scale > 0:
        rec["max_abs_err_over_scale"] = max(rec["max_abs_err_over_scale"], float(err.max() / scale))
    else:  # purely relative check (losses): report the relative error instead
        rel = err / np.maximum(np.abs(want), 1e-300)
        rec["max_abs_err_over_scale"] = max(rec["max_abs_err_over_scale"], float(rel.max()))
    rec["elements"] += int(err.size)
    assert err.flat[k] <= tol.flat[k], (f"{name}: |{got.flat[k]:.9g} - {want.flat[k]:.9g}| = {err.flat[k]:.3g} "
                                        f"> {tol.flat[k]:.3g} at flat index {k}")


def test_cfg2_ctc_every_utterance():
    """BASELINE configs[1] through the default step (lane-exponent pipelined launch + certificate / repair)."""
    from gtn_applications_amd import engine as E
    from gtn_applications_amd.criterions import ctc

    B, T, C, L = 128, 1000, 100, 44
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, C, generator=g)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    want_loss, want_dx = OR.ctc_loss_grad_batched(x.numpy(), targets, C - 1)
    xg = x.cuda().requires_grad_(True)
    loss = ctc.CTCLoss(xg, targets, C - 1)
    loss.backward()
    assert loss.item() == pytest.approx(want_loss.mean(), rel=RTOL)
    check("cfg2_ctc_dx", xg.grad.cpu().numpy(), want_dx, 1.0 / B)
    # per-utterance losses and the repair count of the same launch, through the engine call the criterion makes
    dev = xg.device
    tg = E.targets_on_device(targets, dev)
    scale, _, coef = E.loss_factors(tg, "none")
    dx = torch.full_like(xg.detach(), float("nan"))
    ws, nll = E.ctc_forward_backward(xg.detach(), tg, C - 1, coef, None, dx)
    check("cfg2_ctc_nll", nll.cpu().numpy(), want_loss, 0.0)
    check("cfg2_ctc_dx_engine", dx.cpu().numpy(), want_dx, 1.0 / B)
    STATS["cfg2_ctc_repaired_utterances"] = E.ctc_pipeline_repaired(ws, B, T, tg.max_len)
    # mean reduction (gradient scaled by 1/L) and the upstream scalar
    xg2 = x.cuda().requires_grad_(True)
    (3.0 * ctc.CTCLoss(xg2, targets, C - 1, "mean")).backward()
    check("cfg2_ctc_dx_mean_x3", xg2.grad.cpu().numpy(), want_dx * (3.0 / L), 3.0 / (L * B))


def test_cfg1_ctc_plumbing_shape():
    """BASELINE configs[0] (T=150, C=28, B=8, L=44 -- the reference's CPU plumbing case): the HIP step against the
    float64 oracle.  (The CPU port at this shape: tests/test_oracle.py::test_c_restatement_at_cfg1_shape.)"""
    from gtn_applications_amd.criterions import ctc

    B, T, C, L = 8, 150, 28, 44
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, C, generator=g)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    want_loss, want_dx = OR.ctc_loss_grad_batched(x.numpy(), targets, C - 1)
    xg = x.cuda().requires_grad_(True)
    loss = ctc.CTCLoss(xg, targets, C - 1)
    loss.backward()
    assert loss.item() == pytest.approx(want_loss.mean(), rel=RTOL)
    check("cfg1_ctc_dx", xg.grad.cpu().numpy(), want_dx, 1.0 / B)


def _model_shaped_scores(rs, B, T, C, L, boost, noise, wrong):
    """log-probabilities that look like a trained model's output: noise + boost on the label a random monotone
    alignment of the target puts on each frame (blank elsewhere), a fraction `wrong` of frames confidently wrong."""
    x = (noise * rs.randn(B, T, C)).astype(np.float32)
    targets = []
    for b in range(B):
        y = rs.randint(0, C - 1, size=L)
        targets.append(y.tolist())
        cuts = np.sort(rs.choice(np.arange(1, T), size=2 * L, replace=False))
        lab = np.full(T, C - 1)
        for i in range(L):
            lab[cuts[2 * i]:cuts[2 * i + 1]] = y[i]
        flip = rs.rand(T) < wrong
        lab = np.where(flip, rs.randint(0, C, size=T), lab)
        x[b, np.arange(T), lab] += boost
    return torch.log_softmax(torch.tensor(x), 2), targets


@pytest.mark.parametrize("boost,noise,wrong", [(8.0, 1.0, 0.1), (12.0, 2.0, 0.1), (20.0, 1.0, 0.02), (3.0, 1.0, 0.3)])
def test_cfg2_ctc_model_shaped_scores_stay_on_the_fast_path(boost, noise, wrong):
    """Peaked log-probabilities (spread up to ~20 nats between the aligned label and the rest): parity on every
    utterance AND at most 2 of 128 utterances handed to the log-domain repair launch -- the lane-exponent step is
    the one that runs on data a trained model produces, not only on unit-variance noise."""
    from gtn_applications_amd import engine as E

    B, T, C, L = 128, 1000, 100, 44
    rs = np.random.RandomState(int(boost * 10 + wrong * 100))
    lp, targets = _model_shaped_scores(rs, B, T, C, L, boost, noise, wrong)
    want_loss, want_dx = OR.ctc_loss_grad_batched(lp.numpy(), targets, C - 1)
    xd = lp.cuda()
    tg = E.targets_on_device(targets, xd.device)
    scale, _, coef = E.loss_factors(tg, "none")
    dx = torch.full_like(xd, float("nan"))
    ws, nll = E.ctc_forward_backward(xd, tg, C - 1, coef, None, dx)
    name = f"cfg2_ctc_model_shaped_b{boost:g}_n{noise:g}_w{wrong:g}"
    check(name + "_nll", nll.cpu().numpy(), want_loss, 0.0)
    check(name + "_dx", dx.cpu().numpy(), want_dx, 1.0 / B)
    repaired = E.ctc_pipeline_repaired(ws, B, T, tg.max_len)
    STATS[name + "_repaired_utterances"] = repaired
    assert repaired <= 2, f"{repaired} of {B} utterances left the lane-exponent path"


@pytest.mark.parametrize("spread", [1.5, 2.5])
def test_cfg2_ctc_scores_the_lane_exponent_step_rejects_are_repaired_to_the_same_bar(spread):
    """log_softmax(spread * randn): a quarter (1.5) / all (2.5) of the utterances leave the lane-exponent step through its
    certificate and are recomputed by the repair launch -- the probability-domain chain and gradient blocks on doubles
    (ctc_log_chain_body, ctc_grad_body) -- to the SAME north-star bar as everything else, every utterance against the
    float64 oracle.  (Their fp32 log-add predecessors were at 6e-5 .. 1.9e-4 of the coefficient here.)"""
    from gtn_applications_amd import engine as E

    B, T, C, L = 128, 1000, 100, 44
    g = torch.Generator().manual_seed(int(spread * 10))
    lp = torch.log_softmax(spread * torch.randn(B, T, C, generator=g), 2)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    want_loss, want_dx = OR.ctc_loss_grad_batched(lp.numpy(), targets, C - 1)
    xd = lp.cuda()
    tg = E.targets_on_device(targets, xd.device)
    scale, _, coef = E.loss_factors(tg, "none")
    dx = torch.full_like(xd, float("nan"))
    ws, nll = E.ctc_forward_backward(xd, tg, C - 1, coef, None, dx)
    name = f"cfg2_ctc_spread{spread:g}"
    check(name + "_nll", nll.cpu().numpy(), want_loss, 0.0)
    check(name + "_dx", dx.cpu().numpy(), want_dx, 1.0 / B)
    # (how many went through the repair launch is recorded, not asserted: after a step with many of them the library
    # runs the log-domain pipelined launch -- the same bodies -- for everything, and then nothing is "repaired")
    STATS[name + "_repaired_utterances"] = E.ctc_pipeline_repaired(ws, B, T, tg.max_len)


@pytest.mark.parametrize("L", [100, 150, 200])
def test_ctc_long_targets_at_benchmark_length_every_utterance(L):
    """Targets of 64 .. 255 labels (character targets of real utterances; two to four target positions per lane:
    ctc_long_chain_body / ctc_long_grad_body, probability domain on doubles) at T = 1000, C = 100: every utterance against
    the float64 oracle at the common bar.  (The fp32 log-add version of these kernels was at 0.9 .. 1.3e-4 of the
    coefficient here.)"""
    from gtn_applications_amd.criterions import ctc

    B, T, C = 32, 1000, 100
    g = torch.Generator().manual_seed(L)
    lp = torch.log_softmax(torch.randn(B, T, C, generator=g), 2)
    targets = [torch.randint(C - 2, (int(n),), generator=g).tolist() for n in torch.randint(L - 30, L + 1, (B,), generator=g)]
    want_loss, want_dx = OR.ctc_loss_grad_batched(lp.numpy(), targets, C - 1)
    xg = lp.cuda().requires_grad_(True)
    loss = ctc.CTCLoss(xg, targets, C - 1, "none")
    loss.backward()
    assert loss.item() == pytest.approx(want_loss.mean(), rel=RTOL)
    check(f"ctc_long_L{L}_dx", xg.grad.cpu().numpy(), want_dx, 1.0 / B)


def test_cfg2_ctc_module_raw_scores_every_utterance():
    """The CTC MODULE (ctc.py:99-121, use_pt=False) at BASELINE configs[1]'s shape: raw scores in, log_softmax fused into
    the meet-in-the-middle launch (the emitters form cf (gamma - softmax(x)) from the raw rows while they write the
    tile out), every utterance's gradient w.r.t. the RAW scores against the float64 oracle through log_softmax."""
    from gtn_applications_amd.criterions import ctc

    B, T, C, L = 128, 1000, 100, 44
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, T, C, generator=g)  # (ctc_benchmark.py:21-22: unit-variance scores)
    targets = torch.randint(C - 2, (B, L), generator=g)
    lp = torch.log_softmax(x.double(), 2)
    want_loss, dlp = OR.ctc_loss_grad_batched(lp.numpy(), targets.tolist(), C - 1)  # (dlp = -gamma_b / B)
    # module: mean over the batch of nll_b / L_b; d/dx = dlp - softmax * sum_c dlp
    dlp = torch.tensor(dlp, dtype=torch.float64) / L
    want_dx = (dlp - torch.exp(lp) * dlp.sum(dim=2, keepdim=True)).numpy()
    xg = x.cuda().requires_grad_(True)
    loss = ctc.CTC(C - 1, False)(xg, [t for t in targets])
    loss.backward()
    assert loss.item() == pytest.approx((want_loss / L).mean(), rel=RTOL)
    check("cfg2_ctc_module_dx", xg.grad.cpu().numpy(), want_dx, 1.0 / (L * B))
    rows = xg.grad.sum(dim=2).abs().max().item()  # the gradient through a softmax sums to zero over a row
    assert rows <= 5e-5 / (L * B) * C


def test_cfg3_asg_every_utterance():
    """BASELINE configs[2]: ASGLoss at B=128 with random learned transitions, dx and dW of the whole batch."""
    from gtn_applications_amd.criterions import asg

    B, T, C, L = 128, 1000, 100, 44
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, C, generator=g)
    W = torch.randn(C + 1, C, generator=g)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    want_loss, want_dx, want_dW = OR.asg_loss_grad_batched(x.numpy(), W.numpy(), targets)
    xg, Wg = x.cuda().requires_grad_(True), W.cuda().requires_grad_(True)
    loss = asg.ASGLoss(xg, Wg, targets)
    loss.backward()
    assert loss.item() == pytest.approx(want_loss.mean(), rel=RTOL)
    check("cfg3_asg_dx", xg.grad.cpu().numpy(), want_dx, 1.0 / B)
    # a transition gradient is a sum of B*(T-1) posteriors of scale 1/B each: tolerance relative to the
    # expected value plus 1e-4 of ONE posterior's scale
    check("cfg3_asg_dW", Wg.grad.cpu().numpy(), want_dW, 1.0 / B)
    # mean reduction, gradient of each input alone (asg.py:142-185 honours needs_input_grad)
    xg2 = x.cuda().requires_grad_(True)
    asg.ASGLoss(xg2, W.cuda(), targets, "mean").backward()
    check("cfg3_asg_dx_mean", xg2.grad.cpu().numpy(), want_dx / L, 1.0 / (L * B))


def test_asg_at_benchmark_length_with_a_forbidden_transition():
    """A transition of -1e4 (a forbidden label bigram; -inf likewise) is outside what the probability-domain dense sweeps represent:
    every utterance then takes the log-domain launches (dense_chain_kernel / dense_grad_kernel, csrc/dense_kernels.hip)
    -- doubles in LDS, frames stored relative to a double per frame, alpha + beta - ln Z summed in double.  At the
    benchmark's T = 1000, C = 100, every utterance of a batch of 16 against the float64 oracle, at the common bar
    (asg.py:54-69,100-139; until round 4 these launches were plain fp32 log-adds: 1.0 .. 2.5e-4 of the coefficient)."""
    from gtn_applications_amd import engine as E
    from gtn_applications_amd.criterions import asg

    B, T, C, L = 16, 1000, 100, 44
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, T, C, generator=g)
    W = torch.randn(C + 1, C, generator=g)
    W[1 + 5, 7] = -1.0e4  # label 7 never directly before label 5 (as good as -inf; the batched oracle wants finite scores)
    targets = torch.randint(C - 2, (B, L), generator=g)
    targets[targets == 5] = 9  # (the targets do not ask for it)
    targets = targets.tolist()
    want_loss, want_dx, want_dW = OR.asg_loss_grad_batched(x.numpy(), W.numpy(), targets)
    st = E.dense_forward(x.cuda(), W.cuda())
    assert bool(E.dense_flagged(st).all().item())  # the log-domain launches served the batch
    xg, Wg = x.cuda().requires_grad_(True), W.cuda().requires_grad_(True)
    loss = asg.ASGLoss(xg, Wg, targets)
    loss.backward()
    assert loss.item() == pytest.approx(want_loss.mean(), rel=RTOL)
    check("asg_forbidden_transition_dx", xg.grad.cpu().numpy(), want_dx, 1.0 / B)
    dW = Wg.grad.cpu().numpy()
    assert dW[1 + 5, 7] == 0.0
    check("asg_forbidden_transition_dW", dW, want_dW, 1.0 / B)


def _word_piece_setup():
    with open(os.path.join(ROOT, "tests", "golden", "word_pieces_tokens_1000.txt")) as fid:
        tokens = sorted(l.strip() for l in fid)
    graphemes = sorted(set(c for t in tokens for c in t))
    return tokens, {t: i for i, t in enumerate(graphemes)}


def _lattice_oracle(x_b, arcs, sc, B):
    """loss and gradient w.r.t. RAW scores of -sc * forward_score(log_softmax(x) o A) (transducer.py:186-187,
    283,302-305,329-336) for an epsilon-free acceptor given as arrays."""
    lp = OC.log_softmax(np.asarray(x_b, dtype=np.float64), 1)
    src, dst, lab, start, accept, n = arcs
    logz, g, _ = OR.lattice_forward_backward(lp, src, dst, lab, np.zeros(len(src)), start, accept, n)
    din = -g * sc / B
    return -logz * sc, din - np.exp(lp) * din.sum(axis=1, keepdims=True)


def test_cfg4_transducer_word_pieces():
    """BASELINE configs[3] with the reference's own 1000 word pieces: 10^6-arc token graph, C=1001, T=800,
    B=64.  Every utterance against the float64 recurrence over the alignment acceptor the host library built;
    two utterances against the acceptor built by the ORACLE's graph algebra (compose / remove / project);
    plus the batch properties (rows sum to zero through the fused log_softmax, loss finite)."""
    from gtn_applications_amd.criterions import transducer as TR

    tokens, g2i = _word_piece_setup()
    B, T, Lp = 64, 800, 15
    C = len(tokens) + 1
    random.seed(0)
    targets = [[g2i[c] for wp in (random.choice(tokens) for _ in range(Lp)) for c in wp] for _ in range(B)]
    x = torch.randn(B, T, C, generator=torch.Generator().manual_seed(0))
    crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
    assert crit.tokens.num_arcs() == 1002002 and crit.lexicon.num_arcs() == 4644
    xg = x.cuda().requires_grad_(True)
    loss = crit(xg, [torch.tensor(t) for t in targets])
    loss.backward()
    dx = xg.grad.cpu().numpy()
    assert np.isfinite(loss.item())
    # log_softmax backward: rows sum to zero, i.e. the posteriors of a frame sum to one to 1e-4 of their scale
    assert np.abs(dx.sum(axis=2)).max() <= 1e-4 * (1.0 / (B * min(len(t) for t in targets)))
    crit.tokens.arc_sort(True)
    losses = []
    for b in range(B):
        a = TR._alignment_graph(targets[b], crit.tokens, crit.lexicon, None)[0].arrays()
        assert (a["ilabel"] >= 0).all()
        arcs = (a["src"], a["dst"], a["ilabel"], np.nonzero(a["start"])[0], np.nonzero(a["accept"])[0], len(a["start"]))
        sc = 1.0 / len(targets[b])
        want_loss, want_dx = _lattice_oracle(x[b].numpy(), arcs, sc, B)
        losses.append(want_loss)
        check("cfg4_transducer_dx", dx[b], want_dx, sc / B)
    assert loss.item() == pytest.approx(float(np.mean(losses)), rel=RTOL)
    oracle = OC.TransducerOracle(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
    for b in (0, B - 1):
        ali = oracle.alignment_graph(targets[b])
        arcs = (ali.src, ali.dst, ali.ilab, ali.start_nodes(), ali.accept_nodes(), ali.num_nodes())
        sc = 1.0 / len(targets[b])
        want_loss, want_dx = _lattice_oracle(x[b].numpy(), arcs, sc, B)
        assert want_loss == pytest.approx(losses[b], rel=1e-9)  # host library and oracle built the same acceptor
        check("cfg4_transducer_dx_oracle_graph", dx[b], want_dx, sc / B)


def test_transducer_acceptors_of_more_than_512_states():
    """Targets long enough for the alignment acceptor to need the 1024-thread sweep workgroups (their chunks run as frame
    loops, not as straight-line code; 90 KB of LDS): 300 letters, blank optional -> 601 states.  Every utterance, loss
    and emission gradient, against the float64 recurrence; the last frames form a partial chunk (T % 16 != 0)."""
    from gtn_applications_amd.criterions import transducer as TR

    letters = [chr(ord("a") + i) for i in range(26)]
    g2i = {c: i for i, c in enumerate(letters)}
    B, T, L = 3, 650, 300
    C = len(letters) + 1
    rs = np.random.RandomState(5)
    targets = [[int(v) for v in rs.randint(0, 26, size=L - 7 * b)] for b in range(B)]
    x = torch.randn(B, T, C, generator=torch.Generator().manual_seed(5))
    crit = TR.Transducer(letters, g2i, blank="optional", allow_repeats=False, reduction="mean")
    xg = x.cuda().requires_grad_(True)
    loss = crit(xg, [torch.tensor(t) for t in targets])
    loss.backward()
    dx = xg.grad.cpu().numpy()
    crit.tokens.arc_sort(True)
    losses = []
    for b in range(B):
        a = TR._alignment_graph(targets[b], crit.tokens, crit.lexicon, None)[0].arrays()
        assert len(a["start"]) > 512 or b > 0
        arcs = (a["src"], a["dst"], a["ilabel"], np.nonzero(a["start"])[0], np.nonzero(a["accept"])[0], len(a["start"]))
        sc = 1.0 / len(targets[b])
        want_loss, want_dx = _lattice_oracle(x[b].numpy(), arcs, sc, B)
        losses.append(want_loss)
        check("transducer_601_states_dx", dx[b], want_dx, sc / B)
    assert loss.item() == pytest.approx(float(np.mean(losses)), rel=RTOL)


def test_transducer_word_pieces_reference_golden(golden_dir):
    """the reference's Transducer module itself (run on the oracle's WFST primitives by
    oracle/pin_against_reference.py --write-round2-golden) with the 1000 word pieces, short input"""
    from gtn_applications_amd.criterions import transducer as TR

    gold = np.load(os.path.join(golden_dir, "transducer_wordpieces_1000.npz"))
    tokens, g2i = _word_piece_setup()
    B, T = int(gold["B"]), int(gold["T"])
    flat = gold["targets"].tolist()
    lens, flat = flat[:B], flat[B:]
    targets = [flat[sum(lens[:b]):sum(lens[:b + 1])] for b in range(B)]
    x = torch.randn(B, T, len(tokens) + 1, generator=torch.Generator().manual_seed(int(gold["seed"])))
    assert float(x.double().sum()) == pytest.approx(float(gold["x_checksum"]), abs=1e-6)  # same inputs as the generator
    crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
    xg = x.cuda().requires_grad_(True)
    loss = crit(xg, [torch.tensor(t) for t in targets])
    loss.backward()
    assert loss.item() == pytest.approx(float(gold["loss"]), rel=RTOL)
    check("transducer_wordpieces_reference_golden_dx", xg.grad.cpu().numpy(), gold["grad"], 1.0 / (B * min(lens)))


def test_cfg5_ctc_shard_every_utterance():
    """One GPU's shard of BASELINE configs[4] (T=2000, C=512, B=128, seed = rank 0): wide rows, compact
    gradient tiles; oracle in chunks of 32 utterances."""
    from gtn_applications_amd.criterions import ctc

    B, T, C, L = 128, 2000, 512, 44
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, C, generator=g)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    xg = x.cuda().requires_grad_(True)
    loss = ctc.CTCLoss(xg, targets, C - 1)
    loss.backward()
    dx = xg.grad.cpu().numpy()
    losses = []
    for lo in range(0, B, 32):
        want_loss, want_dx = OR.ctc_loss_grad_batched(x[lo:lo + 32].numpy(), targets[lo:lo + 32], C - 1, batch_size=B)
        losses.append(want_loss)
        check("cfg5_ctc_shard_dx", dx[lo:lo + 32], want_dx, 1.0 / B)
    assert loss.item() == pytest.approx(float(np.concatenate(losses).mean()), rel=RTOL)


def test_criteria_run_on_a_device_that_is_not_the_current_one():
    """Inputs on cuda:k, k != torch.cuda.current_device() (what rank k of a multi-GPU job would see if it did not call
    set_device): the per-device staging rings, workspaces and timing events of the engine must be looked up for the
    inputs' device, results land there, and the values equal the ones cuda:0 gives.  Needs two visible devices."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one device visible")
    from gtn_applications_amd import engine as E
    from gtn_applications_amd.criterions import asg, ctc

    B, T, C, L = 16, 200, 40, 12
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, C, generator=g)
    W = torch.randn(C + 1, C, generator=g)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    res = {}
    assert torch.cuda.current_device() == 0
    for k in (0, 1):
        dev = torch.device("cuda", k)
        xg = x.to(dev).requires_grad_(True)
        E.PHASE_EVENTS = []  # timing events are per device too
        try:
            loss = ctc.CTCLoss(xg, targets, C - 1)
            loss.backward()
        finally:
            E.PHASE_EVENTS = None
        xa, Wa = x.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
        la = asg.ASGLoss(xa, Wa, targets)
        la.backward()
        assert loss.device == dev and xg.grad.device == dev and Wa.grad.device == dev
        res[k] = [t.cpu() for t in (loss.detach(), xg.grad, la.detach(), xa.grad, Wa.grad)]
    assert torch.cuda.current_device() == 0
    for a, b in zip(res[0], res[1]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("C,B,T,L", [(1000, 6, 60, 12), (333, 70, 33, 9)])
def test_asg_beyond_the_on_chip_class_limit(C, B, T, L):
    """asg.py:198-199 sizes `transitions` to any N.  Above wfl_dense_on_chip_classes() the dense sweeps run as one
    tiled matrix product per frame for the whole batch (csrc/dense_wide.h): ASGLoss with 1000 classes (and a size
    with partial tiles on both axes) against the float64 oracle -- loss, emission gradient, transition gradient,
    mean reduction, an upstream scalar -- and ASG.viterbi at that size against the oracle's max-plus recursion."""
    from gtn_applications_amd import _native as N
    from gtn_applications_amd.criterions import asg

    assert C > N.lib.wfl_dense_on_chip_classes()
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, T, C, generator=g)
    W = torch.randn(C + 1, C, generator=g)
    targets = [torch.randint(C, (int(n),), generator=g).tolist() for n in torch.randint(1, L + 1, (B,), generator=g)]
    want_loss, want_dx, want_dW = OR.asg_loss_grad_batched(x.numpy(), W.numpy(), targets)
    xg, Wg = x.cuda().requires_grad_(True), W.cuda().requires_grad_(True)
    loss = asg.ASGLoss(xg, Wg, targets)
    (2.0 * loss).backward()
    assert loss.item() == pytest.approx(want_loss.mean(), rel=RTOL)
    check(f"asg_wide_c{C}_dx", xg.grad.cpu().numpy(), 2.0 * want_dx, 2.0 / B)
    check(f"asg_wide_c{C}_dW", Wg.grad.cpu().numpy(), 2.0 * want_dW, 2.0 / B)
    # the module (garbage + replabels on top: N = C here by construction) and its Viterbi decode
    crit = asg.ASG(C - 2, 1, True).cuda()
    with torch.no_grad():
        crit.transitions.copy_(W.cuda())
    assert crit.N == C
    got = crit.viterbi(x.cuda())
    for b in range(min(B, 3)):
        path = OR.dense_viterbi(x[b].numpy(), W.numpy())
        import itertools

        collapsed = [p for p, _ in itertools.groupby(path)]
        collapsed = [p for p in collapsed if p != crit.garbage_idx]
        assert got[b].tolist() == asg.unpack_replabels(collapsed, 1)


def test_ctc_batches_with_more_sweeps_than_cus():
    """B = 160 at the cfg2 shape: 320 sweeps on 256 CUs -- the meet-in-the-middle step switches to its 8-wave workgroups,
    two per CU (csrc/ctc_mitm.h MitmK<8>).  Every utterance against the float64 oracle, ragged targets included
    (lengths 1 .. 44, one empty), plus the mean reduction."""
    from gtn_applications_amd.criterions import ctc

    B, T, C, L = 160, 1000, 100, 44
    g = torch.Generator().manual_seed(160)
    x = torch.randn(B, T, C, generator=g)
    lens = torch.randint(1, L + 1, (B,), generator=g).tolist()
    lens[7] = 0
    targets = [torch.randint(C - 2, (n,), generator=g).tolist() for n in lens]
    want_loss, want_dx = OR.ctc_loss_grad_batched(x.numpy(), targets, C - 1)
    xg = x.cuda().requires_grad_(True)
    loss = ctc.CTCLoss(xg, targets, C - 1)
    loss.backward()
    assert loss.item() == pytest.approx(want_loss.mean(), rel=RTOL)
    check("ctc_b160_dx", xg.grad.cpu().numpy(), want_dx, 1.0 / B)


def test_cfg4_transducer_viterbi_decodes_the_batch_in_one_native_call():
    """Transducer.viterbi at BASELINE configs[3]'s size (transducer.py:199-234; called in every training step,
    train.py:278-279): frame labels by the row-argmax kernel (first maximum, NaN = impossible), token sequences by
    wfl_transducer_decode_batch.  Every utterance against the collapse the token graph stands for, two utterances
    against the reference's own chain of graph calls on the host library."""
    import itertools

    from gtn_applications_amd import graph as G
    from gtn_applications_amd.criterions import transducer as TR

    tokens, g2i = _word_piece_setup()
    B, T = 64, 800
    C = len(tokens) + 1
    x = torch.randn(B, T, C, generator=torch.Generator().manual_seed(1))
    x[0, 5, :] = x[0, 5, 17]          # a frame of equal scores: the first label wins
    x[1, 7, 3] = float("nan")         # a NaN is an impossible arc, not a maximum
    x[2, :40, C - 1] = 9.0            # a stretch of blanks
    x[3, 100:140, 11] = 9.0           # a stretch of one token
    crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
    got = crit.viterbi(x.cuda())
    assert len(got) == B and all(p.dtype == torch.int32 and p.device.type == "cpu" for p in got)
    xn = np.nan_to_num(x.numpy(), nan=-np.inf)
    frames = xn.argmax(axis=2)  # (numpy: the first maximum)
    assert frames[0, 5] == 0
    for b in range(B):
        want = [k for k, _ in itertools.groupby(frames[b].tolist()) if k != C - 1]
        assert got[b].tolist() == want, b
    crit.tokens.arc_sort()
    for b in (0, 3, B - 1):
        path = G.viterbi_path(G.compose(TR.make_chain_graph(frames[b].tolist()), crit.tokens))
        assert got[b].tolist() == G.remove(G.project_output(path)).labels_to_list()


def test_row_argmax_first_maximum_any_width():
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(3)
    for C in (1, 7, 64, 65, 100, 129, 300, 513, 1001, 1500):
        x = rs.randint(-3, 4, size=(3, 37, C)).astype(np.float32)  # (many ties)
        x[0, 0, :] = -np.inf
        if C > 2:
            x[1, 1, 1] = np.nan
        got = E.row_argmax(torch.tensor(x).cuda()).cpu().numpy()
        want = np.nan_to_num(x, nan=-np.inf).argmax(axis=2)
        assert (got == want).all(), C


@pytest.mark.parametrize("argv", [["--config", "cfg5"], ["--workload", "asg"]])
def test_bench_collective_path_executes_on_one_gpu(argv):
    """bench.py's multi-GPU branch as far as a 1-GPU box allows (train.py:137-142,201-208): with
    WFL_BENCH_FORCE_DIST=1 the process group is RCCL ("nccl") at world size 1, the barrier / MAX-over-ranks timing runs,
    and the step's all-reduce(mean) -- the [(C+1), C] buffer of cfg5, the transition-weight gradient of the ASG step --
    goes through parallel.all_reduce_mean_ on the device.  No scaling number comes out of this (there is one GPU)."""
    import subprocess
    import sys

    env = dict(os.environ, WFL_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                          "--no-cpu-baseline", "--no-extras"] + argv, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["collective_world_size"] == 1
    assert "all-reduce(mean)" in line["config"]["parallelism"]
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
