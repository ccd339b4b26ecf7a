"""GPU parity tests (-m gpu): the HIP path (through the C ABI of libwfl.so) against the oracle on
the same inputs, against the committed golden fixtures and the reference's literal vectors, and --
at BASELINE sizes -- through size-independent properties.

Tolerances: log-semiring losses and gradients 1e-4 relative (BASELINE.json north_star) with an
absolute floor of 1e-5 on gradients; Viterbi / index outputs bit-exact.
Nothing here reads /root/reference."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import criteria as OC  # noqa: E402
from oracle import recurrences as OR  # noqa: E402

RTOL, ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def crit():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gtn_applications_amd.criterions import asg, ctc, stc, transducer

    return dict(ctc=ctc, asg=asg, stc=stc, transducer=transducer)


@pytest.fixture(scope="module")
def lit(golden_dir):
    with open(os.path.join(golden_dir, "reference_literals.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def cases(golden_dir):
    with open(os.path.join(golden_dir, "criterion_cases.json")) as f:
        return json.load(f)


def dev(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    return t.requires_grad_(True) if grad else t


def close(got, want, rtol=RTOL, atol=ATOL, msg=""):
    got = got.detach().cpu().double().numpy() if hasattr(got, "detach") else np.asarray(got)
    np.testing.assert_allclose(got, np.asarray(want), rtol=rtol, atol=atol, err_msg=msg)


# =================================================================================================
# CTC
# =================================================================================================
@pytest.mark.parametrize("reduction", ["none", "mean"])
def test_ctc_cpp_autograd_node_is_the_python_operator(crit, reduction):
    """The hot case goes through the C++ autograd node (csrc/torch_ops.cpp): same loss and gradient, bit for bit,
    as the Python autograd.Function it stands in for; a non-unit upstream gradient, a second backward over a
    retained graph and the fused log_softmax of the module behave the same."""
    ctc = crit["ctc"]
    assert ctc._native_node() is not None, "_wfl_torch.so is not built"
    rs = np.random.RandomState(7)
    B, T, C = 5, 60, 12
    xs = np.log(rs.dirichlet(np.ones(C), size=(B, T))).astype(np.float32)
    targets = [rs.randint(0, C - 1, size=rs.randint(0, 14)).tolist() for _ in range(B)]
    x1, x2 = dev(xs, grad=True), dev(xs, grad=True)
    l1 = ctc.CTCLoss(x1, targets, C - 1, reduction)
    assert "CtcStep" in l1.grad_fn.name(), l1.grad_fn.name()
    l2 = ctc.CTCLossFunction.apply(x2, targets, C - 1, reduction)
    assert torch.equal(l1, l2)
    w = dev(rs.randn(*l1.shape))
    (l1 * w).sum().backward(retain_graph=True)
    (l2 * w).sum().backward(retain_graph=True)
    assert torch.equal(x1.grad, x2.grad)
    want = OR.ctc_loss_grad(xs, targets, C - 1, reduction)
    close(l1, want[0])
    g1 = x1.grad.clone()
    x1.grad = None
    (l1 * w).sum().backward()  # second backward: the launch runs again
    close(x1.grad, g1.cpu().numpy(), rtol=1e-6, atol=1e-7)
    # targets as 1-D CPU tensors (what train.py hands over): int64, and a strided int32 view
    x3 = dev(xs, grad=True)
    l3 = ctc.CTCLoss(x3, [torch.tensor(t, dtype=torch.long) for t in targets], C - 1, reduction)
    assert "CtcStep" in l3.grad_fn.name() and torch.equal(l3, l1)
    x4 = dev(xs, grad=True)
    wide = [torch.tensor([v for u in t for v in (u, -7)], dtype=torch.int32)[::2] for t in targets]
    assert torch.equal(ctc.CTCLoss(x4, wide, C - 1, reduction), l1)
    # module with the fused log_softmax
    raw = rs.randn(B, T, C).astype(np.float32)
    r1, r2 = dev(raw, grad=True), dev(raw, grad=True)
    tt = [torch.tensor(t, dtype=torch.long) for t in targets]
    m = ctc.CTC(C - 1, False)
    m(r1, tt).backward()
    ctc._FusedLogSoftmaxCTCLoss.apply(r2, tt, C - 1, "mean").backward()
    assert torch.equal(r1.grad, r2.grad)


@pytest.mark.parametrize("nbytes", [1, 15, 16, 17, 4095, 23 * 1024 + 3, 1 << 20])
def test_upload_kernel_copies_pinned_memory_exactly(crit, nbytes):
    """wfl_upload: the staged targets / packed lattices reach the device through a kernel that reads the pinned buffer."""
    from gtn_applications_amd import engine as E

    g = torch.Generator().manual_seed(nbytes)
    src = torch.empty(nbytes + 64, dtype=torch.uint8, pin_memory=True)
    src.copy_(torch.randint(0, 256, (nbytes + 64,), generator=g, dtype=torch.uint8))
    dst = torch.full((nbytes + 64,), 7, dtype=torch.uint8, device="cuda")
    E.upload(dst, src, nbytes)
    torch.cuda.synchronize()
    assert torch.equal(dst[:nbytes].cpu(), src[:nbytes]) and bool((dst[nbytes:] == 7).all())


def test_native_library_is_the_one_loaded(crit):
    from gtn_applications_amd import _native

    assert os.path.basename(_native.LIB_PATH) == "libwfl.so" and _native.lib.wfl_version() >= 1


def test_ctc_reference_literals(crit, lit):
    ctc = crit["ctc"]
    c = lit["ctc_trivial"]
    lp = torch.log(dev(c["probs"]).view(1, c["T"], c["N"]))
    assert ctc.CTCLoss(lp, c["labels"], c["blank"]).item() == pytest.approx(0.0, abs=1e-6)
    c = lit["ctc_uniform"]
    lp = torch.log_softmax(torch.zeros(1, c["T"], c["N"], device="cuda"), 2)
    assert ctc.CTCLoss(lp, c["labels"], c["blank"]).item() == pytest.approx(-math.log(0.25 ** 3 * 5), abs=1e-5)
    for key in ("ctc_5x6", "ctc_5x6_repeat"):
        c = lit[key]
        le = torch.log(dev(c["probs"]).view(1, c["T"], c["N"])).requires_grad_(True)
        loss = ctc.CTCLoss(torch.log_softmax(le, 2), c["labels"], c["blank"])
        assert loss.item() == pytest.approx(c["loss"], abs=5e-5)
        loss.backward()
        close(le.grad.view(-1), c["grad"], rtol=1e-4, atol=1e-6, msg=key)


def test_ctc_golden_cases(crit, cases):
    for name, c in cases.items():
        if c["kind"] != "ctc":
            continue
        x = dev(c["inputs"], grad=True)
        lp = torch.log_softmax(x, 2) if c["log_softmax"] else x
        loss = crit["ctc"].CTCLoss(lp, c["targets"], c["blank"], c["reduction"])
        loss.backward()
        assert loss.item() == pytest.approx(c["loss"], rel=RTOL, abs=1e-5), name
        close(x.grad, c["grad"], msg=name)


@pytest.mark.parametrize("B,T,C,Lmax,reduction", [(4, 50, 11, 9, "none"), (3, 120, 30, 63, "mean"), (2, 33, 7, 0, "mean"),
                                                  (5, 64, 20, 20, "mean")])
def test_ctc_fast_path_vs_oracle(crit, B, T, C, Lmax, reduction):
    rs = np.random.RandomState(B * 1000 + T)
    x = rs.randn(B, T, C).astype(np.float32) * 2.0
    targets = [rs.randint(0, C - 1, size=rs.randint(0, Lmax + 1)).tolist() for _ in range(B)]
    targets[0] = rs.randint(0, C - 1, size=Lmax).tolist()
    if Lmax >= 4:
        targets[-1] = [1, 1, 1, 1][:Lmax]  # forced blanks between repeats
    want_loss, want_dx = OR.ctc_loss_grad(x, targets, C - 1, reduction)
    xt = dev(x, grad=True)
    loss = crit["ctc"].CTCLoss(xt, targets, C - 1, reduction)
    loss.backward()
    assert loss.item() == pytest.approx(want_loss, rel=RTOL)
    close(xt.grad, want_dx)


@pytest.mark.parametrize("lens,T", [((70, 64, 5), 180), ((127, 128, 0), 300), ((129, 191, 192), 420),
                                    ((255, 193, 17), 530), ((256, 300, 12), 640)])
def test_ctc_long_targets(crit, lens, T):
    """targets longer than a wavefront: two to four positions per lane on the CTC fast path (pipelined
    step through CTCLoss, two-kernel step through the engine calls), the generic lattice engine
    beyond 255 labels -- all against the oracle, with repeated labels and ragged lengths"""
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(sum(lens))
    B, C = len(lens), 12
    x = rs.randn(B, T, C).astype(np.float32)
    targets = [rs.randint(0, C - 1, size=n).tolist() for n in lens]
    want_loss, want_dx = OR.ctc_loss_grad(x, targets, C - 1, "mean")
    xt = dev(x, grad=True)
    loss = crit["ctc"].CTCLoss(xt, targets, C - 1, "mean")
    loss.backward()
    assert loss.item() == pytest.approx(want_loss, rel=RTOL)
    close(xt.grad, want_dx)
    if max(lens) <= E.CTC_FAST_MAX_LEN:
        tg = E.targets_on_device(targets, xt.device)
        scale, _, coef = E.loss_factors(tg, "mean")
        dx = torch.full_like(xt, float("nan"))
        ws, nll = E.ctc_forward(xt.detach(), tg, C - 1)
        E.ctc_grad(xt.detach(), tg, C - 1, ws, nll, coef, None, dx)
        assert float(E.reduce_loss(nll, scale, 1.0)) == pytest.approx(want_loss, rel=RTOL)
        close(dx, want_dx)


def test_ctc_three_hip_paths_agree(crit):
    """the same batch through the generic lattice kernels and through both CTC chains (default
    log-domain chain; lane-exponent chain with certificate), including block-boundary cases of
    the checkpoint/recompute scheme (T = 16k, 16k+1, 16k+8, 16k+15, < 16)"""
    from gtn_applications_amd import _native as N
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(11)
    for T in (90, 96, 97, 104, 111, 16, 7, 1, 300):
        B, C = 5, 17
        x = dev(rs.randn(B, T, C))
        targets = [rs.randint(0, C - 1, size=rs.randint(0, min(30, T) + 1)).tolist() for _ in range(B)]
        tg = E.CtcTargets(targets, x.device)
        pack = E.PackedLattice.ctc(tg.flat, tg.offsets, C - 1, C, x.device)
        st = E.lattice_forward(x, pack)
        coef = torch.full((B,), -1.0 / B, device="cuda")
        d2 = torch.empty_like(x)
        E.lattice_grad(st, coef, dx=d2)
        ws, nll = E.ctc_forward(x, tg, C - 1, 0)
        close(-st.logz, nll.cpu().double().numpy(), rtol=1e-5, atol=1e-4, msg=f"T={T}")
        d1 = torch.full_like(x, float("nan"))
        E.ctc_grad(x, tg, C - 1, ws, nll, coef, None, d1)
        close(d1, d2.cpu().double().numpy(), rtol=1e-3, atol=1e-6, msg=f"T={T}")


def test_ctc_step_on_an_utterance_with_a_40_nat_outlier_segment(crit):
    """the training step (wfl_ctc_forward_backward: lane-exponent sweeps that emit the gradient) on an utterance the
    round-1 wave-uniform scaling could not represent (one target label 40 nats above all others inside a long unlikely
    segment): served by the sweeps or rejected by the certificate and recomputed by the repair launch in the same call
    -- either way loss and gradient of both utterances must match the oracle"""
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(1)
    T, C, L = 400, 40, 30
    y = rs.randint(0, C - 1, size=L)
    logits = rs.randn(2, T, C).astype(np.float32)
    pos = np.sort(rs.choice(np.arange(5, T - 5), size=L, replace=False))
    logits[:, :, C - 1] += 12
    for i, p in enumerate(pos):
        logits[:, p:p + 2, y[i]] += 25
    lp = torch.log_softmax(torch.tensor(logits), 2).numpy()
    other = [c for c in range(C - 1) if c not in y][0]
    lp[1, 200:215, :] = -40.0
    lp[1, 200:215, other] = 0.0
    # uniform shifts of a frame do not matter; a frame where the target labels differ by 40 nats does
    lp[1, 203, y[0]] = 0.0
    targets = [y.tolist(), y.tolist()]
    xt = dev(lp)
    tg = E.CtcTargets(targets, xt.device)
    want_loss, want_dx = OR.ctc_loss_grad(lp, targets, C - 1, "none")
    dx = torch.full_like(xt, float("nan"))
    coef = torch.full((2,), -0.5, device="cuda")
    ws, nll = E.ctc_forward_backward(xt, tg, C - 1, coef, None, dx)
    torch.cuda.synchronize()
    assert float(nll.mean()) == pytest.approx(want_loss, rel=RTOL)
    close(dx, want_dx)
    # (whether the damaged utterance went to the repair launch is the certificate's call: the per-lane exponents of the
    # meet-in-the-middle sweeps hold this case themselves; what counts is that what comes out is right)
    assert E.ctc_pipeline_repaired(ws, 2, T, tg.max_len) in (0, 1, 2)


def test_ctc_pipelined_step_matches_split_step_and_oracle():
    """wfl_ctc_forward_backward: chains and gradient waves in one launch (gradient waves wait on
    device-side flags); same loss, gradient within tolerance of the oracle and of the two-kernel
    step; ragged targets, T not a multiple of 16, an infeasible utterance, one 16-frame block only"""
    from gtn_applications_amd import engine as E

    for (B, T, C, Lmax, seed) in [(5, 83, 13, 11, 1), (3, 16, 7, 3, 2), (4, 250, 40, 30, 3), (2, 7, 5, 2, 4)]:
        rs = np.random.RandomState(seed)
        x = rs.randn(B, T, C).astype(np.float32)
        targets = [rs.randint(0, C - 1, size=rs.randint(0, Lmax + 1)).tolist() for _ in range(B)]
        targets[0] = rs.randint(0, C - 1, size=Lmax).tolist()
        if T < 20:
            targets[-1] = [1] * (T + 1)  # cannot be aligned: loss inf, zero gradient
        want_loss, want_dx = OR.ctc_loss_grad(x, targets, C - 1, "none")
        xt = dev(x)
        tg = E.targets_on_device(targets, xt.device)
        scale, _, coef = E.loss_factors(tg, "none")
        gout = torch.full((1,), 0.75, device="cuda")
        dx = torch.full_like(xt, float("nan"))
        ws, nll, loss = E.ctc_forward_backward(xt, tg, C - 1, coef, gout, dx, loss_scale=scale, want_loss=True)
        torch.cuda.synchronize()
        assert not E.ctc_pipeline_gave_up(ws, B, T, tg.max_len)
        ref = E.reduce_loss(nll, scale, 1.0)
        assert torch.equal(loss, ref) or torch.allclose(loss, ref, rtol=1e-6)  # (inf == inf for the infeasible case)
        dx2 = torch.full_like(xt, float("nan"))
        ws2, nll2 = E.ctc_forward(xt, tg, C - 1)
        E.ctc_grad(xt, tg, C - 1, ws2, nll2, coef, gout, dx2)
        # (the pipelined step runs the lane-exponent chains, the split step the log-domain ones)
        assert torch.equal(torch.isfinite(nll), torch.isfinite(nll2))
        assert torch.allclose(torch.nan_to_num(nll, posinf=0.0), torch.nan_to_num(nll2, posinf=0.0), rtol=1e-5, atol=1e-4)
        fin = np.isfinite(nll.cpu().numpy())
        wl, _ = OR.ctc_loss_grad(x[fin], [t for t, f in zip(targets, fin) if f], C - 1, "none")
        assert float(nll[torch.from_numpy(fin).cuda()].mean()) == pytest.approx(wl, rel=RTOL)
        close(dx2, 0.75 * np.nan_to_num(want_dx))
        close(dx, 0.75 * np.nan_to_num(want_dx))


def test_ctc_pipelined_step_at_baseline_size_and_under_graph_replay():
    """cfg2 sizes: the pipelined launch agrees with the two-kernel step; captured in a hipGraph and
    replayed (same workspace, same launch token every time) it keeps producing the same gradient --
    the consumers must have cleared the ready flags of the previous replay"""
    from gtn_applications_amd import engine as E

    g = torch.Generator().manual_seed(5)
    B, T, C, L = 128, 1000, 100, 44
    x = torch.randn(B, T, C, generator=g).cuda()
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    tg = E.targets_on_device(targets, x.device)
    scale, _, coef = E.loss_factors(tg, "none")
    gout = torch.ones(1, device="cuda")
    dx_split, dx_pipe = torch.empty_like(x), torch.empty_like(x)
    ws, nll = E.ctc_forward(x, tg, C - 1)
    E.ctc_grad(x, tg, C - 1, ws, nll, coef, gout, dx_split)
    ws2, nll2 = E.ctc_forward_backward(x, tg, C - 1, coef, gout, dx_pipe)
    torch.cuda.synchronize()
    assert not E.ctc_pipeline_gave_up(ws2, B, T, tg.max_len)
    assert E.ctc_pipeline_repaired(ws2, B, T, tg.max_len) == 0  # benchmark data: served by the lane-exponent chains
    assert torch.allclose(nll, nll2, rtol=1e-5, atol=1e-4)  # (log-domain chains vs lane-exponent chains)
    np.testing.assert_allclose(dx_pipe.cpu().numpy(), dx_split.cpu().numpy(), rtol=2e-3, atol=2e-8)
    np.testing.assert_allclose(dx_pipe.sum(dim=2).cpu().numpy(), coef.cpu().numpy()[:, None] * np.ones((1, T)), rtol=1e-4)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    dx_g = torch.empty_like(x)
    with torch.cuda.stream(side):
        E.ctc_forward_backward(x, tg, C - 1, coef, gout, dx_g)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        ws_g, nll_g = E.ctc_forward_backward(x, tg, C - 1, coef, gout, dx_g)
    for _ in range(3):
        dx_g.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert not E.ctc_pipeline_gave_up(ws_g, B, T, tg.max_len)
        assert torch.equal(dx_g, dx_pipe) and torch.equal(nll_g, nll2)


def test_ctc_pipelined_step_more_chains_than_workgroup_slots():
    """B = 900 utterances = 1800 chain workgroups, more than the chip holds at once: the chains are
    dispatched (in order) before any gradient workgroup, so waiting gradient waves can never starve
    them; long utterances make the waits real"""
    from gtn_applications_amd import engine as E

    g = torch.Generator().manual_seed(9)
    B, T, C, L = 900, 400, 30, 20
    x = torch.randn(B, T, C, generator=g).cuda()
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    tg = E.targets_on_device(targets, x.device)
    scale, _, coef = E.loss_factors(tg, "mean")
    gout = torch.ones(1, device="cuda")
    dx_split, dx_pipe = torch.empty_like(x), torch.empty_like(x)
    ws, nll = E.ctc_forward(x, tg, C - 1)
    E.ctc_grad(x, tg, C - 1, ws, nll, coef, gout, dx_split)
    ws2, nll2, loss = E.ctc_forward_backward(x, tg, C - 1, coef, gout, dx_pipe, loss_scale=scale, want_loss=True)
    torch.cuda.synchronize()
    assert not E.ctc_pipeline_gave_up(ws2, B, T, tg.max_len)
    assert E.ctc_pipeline_repaired(ws2, B, T, tg.max_len) == 0
    assert torch.allclose(nll, nll2, rtol=1e-5, atol=1e-4)
    assert float(loss) == pytest.approx(float((scale * nll2).mean()), rel=1e-6)
    np.testing.assert_allclose(dx_pipe.cpu().numpy(), dx_split.cpu().numpy(), rtol=2e-3, atol=1e-9)


def test_ctc_pipelined_step_stress_fresh_data_same_buffers():
    """the checkpoints travel between CUs of different XCDs inside one launch: thirty steps with new
    emissions each time in the SAME buffers (the allocator hands the blocks out again), every element
    of loss and gradient compared with the two-kernel step -- a stale cross-XCD read would show here"""
    from gtn_applications_amd import engine as E

    g = torch.Generator().manual_seed(21)
    B, T, C, L = 128, 640, 64, 30
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    gout = torch.ones(1, device="cuda")
    for it in range(30):
        x = torch.randn(B, T, C, generator=g).cuda() * (1.0 + 0.1 * it)
        tg = E.targets_on_device(targets, x.device)
        scale, _, coef = E.loss_factors(tg, "mean")
        dx_pipe = torch.empty_like(x)
        ws2, nll2, loss = E.ctc_forward_backward(x, tg, C - 1, coef, gout, dx_pipe, loss_scale=scale, want_loss=True)
        dx_split = torch.empty_like(x)
        ws, nll = E.ctc_forward(x, tg, C - 1)
        E.ctc_grad(x, tg, C - 1, ws, nll, coef, gout, dx_split)
        ref = E.reduce_loss(nll, scale, 1.0)
        torch.cuda.synchronize()
        assert not E.ctc_pipeline_gave_up(ws2, B, T, tg.max_len), it
        assert torch.allclose(nll, nll2, rtol=1e-5, atol=1e-4), it
        assert torch.allclose(loss, ref, rtol=1e-5), (it, float(loss), float(ref))
        # (1e-4 of the gradient's scale |coef| = 1/B: the parity bar; the lane-exponent chains prune mass below that)
        np.testing.assert_allclose(dx_pipe.cpu().numpy(), dx_split.cpu().numpy(), rtol=2e-3, atol=1e-4 / B, err_msg=str(it))
        del x, dx_pipe, dx_split, ws, ws2, nll, nll2, loss


def test_ctc_fast_step_accepts_targets_that_cannot_be_aligned_without_repair():
    """T < L + adjacent repeats: Z = 0 exactly in any arithmetic, so the lane-exponent step's inf loss / zero gradient
    needs no repair launch work (one such utterance in a batch must not double the step time)"""
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(0)
    B, T, C = 6, 20, 9
    x = rs.randn(B, T, C).astype(np.float32)
    targets = [[1, 2, 3], [1] * 25, [1] * 11, [2, 3] * 10, [], [4, 4, 5]]  # [1]*11 needs 21 frames, [2,3]*10 exactly 20
    xt = dev(x)
    tg = E.CtcTargets(targets, xt.device)
    scale, _, coef = E.loss_factors(tg, "none")
    dx = torch.full_like(xt, float("nan"))
    ws, nll, loss = E.ctc_forward_backward(xt, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
    torch.cuda.synchronize()
    assert E.ctc_pipeline_repaired(ws, B, T, tg.max_len) == 0
    assert [math.isinf(v) for v in nll.tolist()] == [False, True, True, False, False, False]
    want_loss, want_dx = OR.ctc_loss_grad(x, targets, C - 1, "none")
    close(dx, np.nan_to_num(want_dx))


@pytest.mark.parametrize("C", [513, 1001, 2500, 17001])
def test_ctc_wide_rows_through_the_criterion(crit, C):
    """word-piece sized vocabularies: compact gradient tiles on the CTC fast path (C <= 16384), the generic lattice
    engine beyond -- CTCLoss and the CTC module (fused log_softmax) against the oracle"""
    rs = np.random.RandomState(C)
    B, T = 3, 50
    x = rs.randn(B, T, C).astype(np.float32)
    targets = [rs.randint(0, C - 1, size=n).tolist() for n in (7, 0, 12)]
    targets[2][3] = targets[2][4] = targets[2][9]  # a repeated label: adjacent and apart
    xt = dev(x, grad=True)
    loss = crit["ctc"].CTCLoss(xt, targets, C - 1, "mean")
    loss.backward()
    want_loss, want_dx = OR.ctc_loss_grad(x, targets, C - 1, "mean")
    assert loss.item() == pytest.approx(want_loss, rel=RTOL)
    close(xt.grad, want_dx)
    lp = OC.log_softmax(x.astype(np.float64))
    want_loss, dlp = OR.ctc_loss_grad(lp, targets, C - 1, "mean")
    want_dx = dlp - np.exp(lp) * dlp.sum(axis=2, keepdims=True)
    xt = dev(x, grad=True)
    loss = crit["ctc"].CTC(C - 1, False)(xt, [torch.tensor(t, dtype=torch.long) for t in targets])
    loss.backward()
    assert loss.item() == pytest.approx(want_loss, rel=RTOL)
    close(xt.grad, want_dx)


def test_ctc_pipeline_env_selects_log_domain_launch():
    """WFL_CTC_PIPELINE=log (read once per process): the log-domain pipelined launch serves the step, nothing is
    ever 'repaired', and the result agrees with the default (lane-exponent) step of this process"""
    import subprocess
    import sys

    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from gtn_applications_amd import engine as E\n"
        "g = torch.Generator().manual_seed(4)\n"
        "x = torch.randn(6, 200, 40, generator=g).cuda()\n"
        "targets = torch.randint(38, (6, 17), generator=g).tolist()\n"
        "tg = E.targets_on_device(targets, x.device)\n"
        "scale, _, coef = E.loss_factors(tg, 'mean')\n"
        "dx = torch.empty_like(x)\n"
        "ws, nll, loss = E.ctc_forward_backward(x, tg, 39, coef, None, dx, loss_scale=scale, want_loss=True)\n"
        "torch.cuda.synchronize()\n"
        "print(repr(float(loss)), repr(float(dx.double().abs().sum())), E.ctc_pipeline_repaired(ws, 6, 200, tg.max_len))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env_val in ("log", None):
        env = dict(os.environ)
        env.pop("WFL_CTC_PIPELINE", None)
        if env_val:
            env["WFL_CTC_PIPELINE"] = env_val
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1].split())
    (l0, g0, r0), (l1, g1, r1) = outs
    assert int(r0) == 0 and int(r1) == 0
    assert float(l0) == pytest.approx(float(l1), rel=1e-5)
    assert float(g0) == pytest.approx(float(g1), rel=1e-4)


def test_ctc_lane_exponent_step_random_shapes():
    """lane-exponent step (the meet-in-the-middle launch, with and without the fused log_softmax) against the three-launch log-domain
    step over random shapes: T around the 16-frame block boundaries, 2 <= C <= 1001 (dense and compact gradient tiles), targets
    of 0..63 labels, infeasible utterances, -inf / NaN entries, score spreads that make the certificate reject
    some utterances (repaired in the log domain within the same call)"""
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(7)
    repaired_total = 0
    for it in range(40):
        B = int(rs.choice([1, 2, 5, 17, 64, 130]))
        T = int(rs.choice([1, 5, 16, 17, 31, 32, 33, 100, 257, 640]))
        C = int(rs.choice([2, 3, 8, 29, 100, 130, 255, 300, 513, 1001]))
        Lmax = int(rs.choice([0, 1, 3, 20, 44, 63]))
        sc = float(rs.choice([0.3, 1.0, 1.0, 1.7]))
        lsm = bool(rs.randint(2))
        x = torch.tensor(rs.randn(B, T, C).astype(np.float32) * sc).cuda()
        if rs.rand() < 0.3:
            x[rs.randint(B), rs.randint(T), rs.randint(C)] = float("-inf")
        if rs.rand() < 0.2:
            x[rs.randint(B), rs.randint(T), rs.randint(C)] = float("nan")
        targets = [rs.randint(0, max(C - 1, 1), size=rs.randint(0, Lmax + 1)).tolist() for _ in range(B)]
        tg = E.CtcTargets(targets, x.device)
        scale, _, coef = E.loss_factors(tg, "mean")
        dx = torch.full_like(x, float("nan"))
        ws, nll, loss = E.ctc_forward_backward(x, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True,
                                               lse=E.row_lse(x) if lsm else None)
        xl = torch.log_softmax(torch.nan_to_num(x, nan=float("-inf")), 2) if lsm else x
        ws2, nll2 = E.ctc_forward(xl, tg, C - 1)
        dx2 = torch.empty_like(x)
        E.ctc_grad(xl, tg, C - 1, ws2, nll2, coef, None, dx2)
        if lsm:  # chain rule of log_softmax by hand
            dx2 = torch.nan_to_num(dx2 - torch.exp(xl) * dx2.sum(2, keepdim=True), nan=0.0)
        torch.cuda.synchronize()
        msg = f"it={it} B={B} T={T} C={C} Lmax={Lmax} scale={sc} lsm={lsm}"
        assert not E.ctc_pipeline_gave_up(ws, B, T, tg.max_len), msg
        repaired_total += E.ctc_pipeline_repaired(ws, B, T, tg.max_len)
        fin = torch.isfinite(nll2)
        assert torch.equal(torch.isfinite(nll), fin), msg
        assert torch.allclose(nll[fin], nll2[fin], rtol=2e-5, atol=2e-4), msg
        assert bool(torch.isfinite(dx).all()), msg
        err = float((dx - dx2).abs().max()) / float(coef.abs().max())
        assert err < 2e-4, (msg, err)  # 1e-4 of the gradient's scale is the parity bar; both sides are fp32
    assert repaired_total > 0  # (the sweep must exercise the repair launch as well)


@pytest.mark.parametrize("lens,T,C", [((7, 0, 12, 3), 60, 12), ((100, 64, 5), 230, 9), ((200, 30), 400, 15)])
def test_ctc_module_fused_log_softmax(crit, lens, T, C):
    """CTC(blank, use_pt=False): log_softmax is fused into the pipelined launch (forward at the gather,
    backward as the softmax term of the rows) -- against the oracle's log_softmax + CTC + chain rule,
    and against the same module with the fusion bypassed; raw scores with a -inf and a NaN entry"""
    rs = np.random.RandomState(sum(lens) + T)
    B = len(lens)
    x = (2.0 * rs.randn(B, T, C)).astype(np.float32)
    x[0, 3, 1] = -np.inf
    x[-1, 5, 2] = np.nan
    targets = [rs.randint(0, C - 1, size=n).tolist() for n in lens]
    xo = np.where(np.isnan(x), -np.inf, x).astype(np.float64)
    lp = OC.log_softmax(xo)
    want_loss, dlp = OR.ctc_loss_grad(lp, targets, C - 1, "mean")
    want_dx = dlp - np.exp(lp) * dlp.sum(axis=2, keepdims=True)
    m = crit["ctc"].CTC(C - 1, False)
    xt = dev(x, grad=True)
    loss = m(xt, [torch.tensor(t, dtype=torch.long) for t in targets])
    (1.5 * loss).backward()
    assert loss.item() == pytest.approx(want_loss, rel=RTOL)
    close(xt.grad, 1.5 * want_dx)
    # the unfused route through the same kernels: torch log_softmax + CTCLoss
    x2 = dev(np.where(np.isnan(x), -np.inf, x).astype(np.float32), grad=True)
    loss2 = crit["ctc"].CTCLoss(torch.nn.functional.log_softmax(x2, dim=2), targets, C - 1, "mean")
    assert loss.item() == pytest.approx(loss2.item(), rel=1e-5)


def test_ctc_infeasible_and_minus_inf(crit):
    ctc = crit["ctc"]
    # T < L: no alignment -> loss +inf, zero gradient (documented policy)
    x = dev(np.zeros((1, 2, 4)), grad=True)
    loss = ctc.CTCLoss(x, [[0, 1, 2]], 3)
    assert math.isinf(loss.item()) and loss.item() > 0
    loss.backward()
    assert float(x.grad.abs().max()) == 0.0
    # -inf emissions on a forced path
    lp = torch.log(dev([[[1.0, 0.0], [0.0, 1.0], [1.0, 0.0]]]))
    assert ctc.CTCLoss(lp, [[0, 0]], 1).item() == pytest.approx(0.0, abs=1e-6)


def test_ctc_errors_and_cpu_tensors(crit):
    ctc = crit["ctc"]
    with pytest.raises(ValueError):
        ctc.CTCLoss(dev(np.zeros((1, 3, 3))), [[0]], 2, "sum")
    with pytest.raises(TypeError):
        ctc.CTCLoss(torch.zeros(1, 3, 3, dtype=torch.float64, device="cuda"), [[0]], 2)
    x = torch.randn(2, 10, 5, requires_grad=True)  # CPU tensor in -> CPU loss / grad out (ctc.py:69,85)
    loss = ctc.CTCLoss(x, [[0, 1], [2]], 4, "mean")
    loss.backward()
    assert loss.device.type == "cpu" and x.grad.device.type == "cpu"
    want, dx = OR.ctc_loss_grad(x.detach().numpy(), [[0, 1], [2]], 4, "mean")
    assert loss.item() == pytest.approx(want, rel=RTOL)
    close(x.grad, dx)


def test_ctc_module_and_greedy_viterbi(crit):
    ctc = crit["ctc"]
    rs = np.random.RandomState(3)
    x = dev(rs.randn(3, 25, 6), grad=True)
    targets = [torch.tensor([0, 1, 2]), torch.tensor([4, 4]), torch.tensor([3])]
    loss = ctc.CTC(5, use_pt=False)(x, targets)
    loss_pt = ctc.CTC(5, use_pt=True)(x.detach(), targets)
    lp = OC.log_softmax(x.detach().cpu().double().numpy())
    want, _ = OR.ctc_loss_grad(lp, [t.tolist() for t in targets], 5, "mean")
    assert loss.item() == pytest.approx(want, rel=RTOL)
    assert loss_pt.item() == pytest.approx(want, rel=1e-3)  # torch's own CTC: same quantity (ctc.py:109-121)
    got = [p.tolist() for p in ctc.CTC(5, False).viterbi(x.detach())]
    assert got == OC.ctc_greedy(x.detach().cpu().numpy(), 5)


def test_ctc_baseline_shape_properties(crit):
    """cfg2 of BASELINE.json (T=1000, C=100, B=128, L=44): oracle on two utterances, and for the
    whole batch the properties: every gradient row sums to coef_b (posteriors of a frame sum to 1),
    zero gradient outside the target's label set, loss == torch's independent CTC."""
    ctc = crit["ctc"]
    g = torch.Generator().manual_seed(0)
    B, T, C, L = 128, 1000, 100, 44
    x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
    tgt = torch.randint(C - 2, (B, L), generator=g)
    targets = tgt.tolist()
    loss = ctc.CTCLoss(x, targets, C - 1)
    loss.backward()
    dx = x.grad
    rows = dx.sum(dim=2)
    # fp32 log-domain rounding accumulates over the 1000 dependent frames: ~3e-4 worst case on a row sum
    assert torch.allclose(rows, torch.full_like(rows, -1.0 / B), rtol=1e-3, atol=1e-7)
    mask = torch.ones(B, C, dtype=torch.bool)
    mask[torch.arange(B).unsqueeze(1), tgt] = False
    mask[:, C - 1] = False
    assert float(dx.abs().amax(dim=1).cpu()[mask].max()) == 0.0
    assert float(dx.max()) <= 0.0
    per_utt = torch.nn.functional.ctc_loss(
        x.detach().permute(1, 0, 2), tgt.cuda(), [T] * B, [L] * B, blank=C - 1, reduction="none")
    assert loss.item() == pytest.approx(per_utt.mean().item(), rel=1e-4)
    xs = x.detach()[:2].cpu().double().numpy()
    want_loss, want_dx = OR.ctc_loss_grad(xs, targets[:2], C - 1)
    sub = ctc.CTCLoss(x.detach()[:2].clone().requires_grad_(True), targets[:2], C - 1)
    assert sub.item() == pytest.approx(want_loss, rel=RTOL)
    # gradient of utterances 0,1 inside the B=128 batch differs from the B=2 one only by the 1/B factor
    close(dx[:2] * (B / 2.0), want_dx, rtol=1e-3, atol=2e-6)


# =================================================================================================
# ASG
# =================================================================================================
def test_asg_reference_literals(crit, lit):
    asg = crit["asg"]
    c = lit["asg_3x5x6"]
    x = dev(c["emissions"], grad=True)
    W = torch.zeros(c["N"] + 1, c["N"], device="cuda", requires_grad=True)
    loss = asg.ASGLoss(x, W, c["labels"])
    assert loss.item() == pytest.approx(c["loss"], abs=5e-5)
    loss.backward()
    close(x.grad * c["B"], c["grad_times_B"], rtol=1e-3, atol=2e-4)
    close(W.grad[1:] * c["B"], c["trans_grad_rows1_times_B"], rtol=1e-3, atol=2e-4)
    c = lit["asg_viterbi"]
    n = c["N"] + c["num_replabels"]
    crit_asg = asg.ASG(c["N"], c["num_replabels"], c["use_garbage"]).cuda()
    with torch.no_grad():
        crit_asg.transitions.copy_(dev(c["transitions"]).view(n + 1, n))
    path = crit_asg.viterbi(dev(c["inputs"]).view(1, c["T"], n))[0].tolist()
    assert path == c["path"]


def test_asg_golden_cases(crit, cases):
    asg = crit["asg"]
    for name, c in cases.items():
        if c["kind"] == "asg":
            x, W = dev(c["inputs"], grad=True), dev(c["transitions"], grad=True)
            loss = asg.ASGLoss(x, W, c["targets"], c["reduction"])
            loss.backward()
            assert loss.item() == pytest.approx(c["loss"], rel=RTOL, abs=1e-5), name
            close(x.grad, c["grad"], msg=name)
            close(W.grad, c["trans_grad"], msg=name)
        elif c["kind"] == "asg_module":
            m = asg.ASG(c["num_classes"], c["num_replabels"], c["use_garbage"]).cuda()
            with torch.no_grad():
                m.transitions.copy_(dev(c["transitions"]))
            x = dev(c["inputs"], grad=True)
            loss = m(x, [torch.tensor(t) for t in c["targets"]])
            loss.backward()
            assert loss.item() == pytest.approx(c["loss"], rel=RTOL, abs=1e-5)
            close(x.grad, c["grad"])
            close(m.transitions.grad, c["trans_grad"])
            assert [p.tolist() for p in m.viterbi(x.detach())] == c["viterbi"]


@pytest.mark.parametrize("B,T,C,reduction", [(3, 40, 9, "none"), (2, 75, 28, "mean"), (4, 30, 100, "mean")])
def test_asg_vs_oracle(crit, B, T, C, reduction):
    rs = np.random.RandomState(C)
    x = rs.randn(B, T, C).astype(np.float32)
    W = (0.5 * rs.randn(C + 1, C)).astype(np.float32)
    targets = [rs.randint(0, C, size=rs.randint(1, min(T, 20))).tolist() for _ in range(B)]
    want = OR.asg_loss_grad(x, W, targets, reduction)
    xt, Wt = dev(x, grad=True), dev(W, grad=True)
    loss = crit["asg"].ASGLoss(xt, Wt, targets, reduction)
    loss.backward()
    assert loss.item() == pytest.approx(want[0], rel=RTOL)
    close(xt.grad, want[1])
    close(Wt.grad, want[2], atol=2e-5)
    # only one of the two gradients requested (asg.py:151-154)
    x2, W2 = dev(x, grad=True), dev(W)
    crit["asg"].ASGLoss(x2, W2, targets, reduction).backward()
    close(x2.grad, want[1])
    x3, W3 = dev(x), dev(W, grad=True)
    crit["asg"].ASGLoss(x3, W3, targets, reduction).backward()
    close(W3.grad, want[2], atol=2e-5)
    # a non-unit upstream gradient scales both halves (the numerator's is computed during forward for
    # grad_output = 1 and rescaled in backward), and a second backward over the retained graph repeats it
    x4, W4 = dev(x, grad=True), dev(W, grad=True)
    loss4 = crit["asg"].ASGLoss(x4, W4, targets, reduction)
    (loss4 * -2.5).sum().backward(retain_graph=True)
    close(x4.grad, -2.5 * want[1])
    close(W4.grad, -2.5 * want[2], atol=5e-5)
    x4.grad = W4.grad = None
    (loss4 * 0.5).sum().backward()
    close(x4.grad, 0.5 * want[1])
    close(W4.grad, 0.5 * want[2], atol=2e-5)


@pytest.mark.parametrize("leaf", [True, False])
def test_asg_native_call_is_the_python_sequence(crit, leaf):
    """csrc/torch_ops.cpp::asg_forward issues the launches of ASGLossFunction.forward (asg.py:84-139) in one native
    call; criterions/asg.py keeps the same sequence spelled in Python (no extension, phase timing).  Same kernels, same
    buffers: loss and dx bit for bit, dW up to the order of its atomics -- as leaves (the forward's gradient handed to .grad) and through the engine."""
    asg = crit["asg"]
    if asg._native_node() is None:
        pytest.skip("the torch extension is not built")
    rs = np.random.RandomState(11)
    B, T, C = 5, 60, 28
    x = rs.randn(B, T, C).astype(np.float32)
    W = (0.5 * rs.randn(C + 1, C)).astype(np.float32)
    targets = [rs.randint(0, C, size=rs.randint(1, 20)).tolist() for _ in range(B)]

    def run():
        xt, Wt = dev(x, grad=True), dev(W, grad=True)
        loss = asg.ASGLoss(xt if leaf else xt * 1.0, Wt if leaf else Wt * 1.0, targets, "mean")
        (loss if leaf else loss * 0.75).backward()
        return loss.detach().cpu().numpy(), xt.grad.cpu().numpy(), Wt.grad.cpu().numpy()

    native = run()
    node, asg._NODE = asg._NODE, None
    try:
        python = run()
    finally:
        asg._NODE = node
    assert np.array_equal(native[0], python[0]) and np.array_equal(native[1], python[1])
    # (the numerator's transition gradient is accumulated with float atomics: the order differs from run to run)
    np.testing.assert_allclose(native[2], python[2], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("hard", [False, True])
def test_dense_calls_in_parts_equal_the_whole(crit, hard):
    """wfl_dense_forward_parts / wfl_dense_grad_parts (include/wfl.h): the probability-domain launches, the log-domain
    launches for what those flag, and the reduction of the transition-gradient partials, asked for one after the other,
    leave what wfl_dense_forward / wfl_dense_grad leave -- with a transition matrix every utterance keeps in the
    probability domain, and with a forbidden transition (-inf: every utterance goes to the log-domain launches)."""
    from gtn_applications_amd import _native as N
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(3)
    B, T, C = 6, 50, 28
    x = dev(rs.randn(B, T, C).astype(np.float32))
    Wn = (0.5 * rs.randn(C + 1, C)).astype(np.float32)
    if hard:
        Wn[4, 7] = -np.inf
    W = dev(Wn)
    coef = dev(rs.rand(B).astype(np.float32))
    whole = E.dense_forward(x, W)
    assert bool(E.dense_flagged(whole).all().item()) == hard
    dx0, dW0 = torch.empty_like(x), torch.empty_like(W)
    E.dense_grad(x, W, whole, coef, coef_w=coef, dx=dx0, dW=dW0)
    st = E.DenseState()
    st.B, st.T, st.C = B, T, C
    st.alpha, st.beta = torch.empty_like(whole.alpha), torch.empty_like(whole.beta)
    st.logz, st.ws = torch.empty_like(whole.logz), torch.empty_like(whole.ws)
    p, s = E.ptr, E.stream_ptr()
    for part in (N.DENSE_MAIN, N.DENSE_REPAIR):
        N.check(N.lib.wfl_dense_forward_parts(p(x), p(W), B, T, C, N.SEMIRING_LOG, p(st.alpha), p(st.beta), None, p(st.logz),
                                              p(st.ws), part, s))
    assert torch.equal(st.logz, whole.logz)
    dx1, dW1 = torch.full_like(x, float("nan")), torch.full_like(W, float("nan"))
    partials = torch.empty(E._dense_sizes(B, T, C)[0], dtype=torch.float32, device=x.device)
    for part in (N.DENSE_REPAIR, N.DENSE_MAIN, N.DENSE_REDUCE):
        N.check(N.lib.wfl_dense_grad_parts(p(x), p(W), B, T, C, p(st.alpha), p(st.beta), p(st.logz), p(coef), p(coef), None, 0,
                                           None, None, p(dx1), p(dW1), p(partials), p(st.ws), part, s))
    assert torch.equal(dx1, dx0) and torch.equal(dW1, dW0)


def test_asg_viterbi_vs_oracle_integer_scores(crit):
    rs = np.random.RandomState(2)
    B, T, C = 4, 30, 7
    x = rs.randint(-6, 7, size=(B, T, C)).astype(np.float32)  # exactly representable: ties are real ties
    W = rs.randint(-3, 4, size=(C + 1, C)).astype(np.float32)
    from gtn_applications_amd import engine as E

    got = E.dense_viterbi(dev(x), dev(W)).cpu().tolist()
    assert got == [OR.dense_viterbi(x[b], W) for b in range(B)]



def _dense_check(x, W, expect_flagged, need_dw=True):
    """engine-level check of the dense (ASG denominator) kernels against the float64 recurrences"""
    from gtn_applications_amd import engine as E

    B, T, C = x.shape
    xt, Wt = dev(x), dev(W)
    st = E.dense_forward(xt, Wt, need_beta=True)
    flagged = E.dense_flagged(st).cpu().tolist()
    coef = torch.full((B,), 0.5, device="cuda")
    dx = torch.full_like(xt, float("nan"))
    dW = torch.zeros_like(Wt) if need_dw else None
    E.dense_grad(xt, Wt, st, coef, coef_w=coef, dx=dx, dW=dW)
    xo = np.where(np.isnan(x), -np.inf, x)  # NaN policy: an impossible emission
    want = [OR.dense_forward_backward(xo[b], W) for b in range(B)]
    logz = st.logz.cpu().numpy()
    for b in range(B):
        if np.isfinite(want[b][0]):
            assert logz[b] == pytest.approx(want[b][0], rel=RTOL, abs=1e-4), b
        else:
            assert logz[b] == want[b][0], b
    close(dx, np.stack([0.5 * np.nan_to_num(w[1]) for w in want]))
    if need_dw:
        close(dW, 0.5 * sum(np.nan_to_num(w[2]) for w in want), atol=5e-5)
    assert flagged == expect_flagged
    # forward only (no beta sweep): same logZ
    st1 = E.dense_forward(xt, Wt, need_beta=False)
    np.testing.assert_allclose(st1.logz.cpu().numpy(), logz, rtol=1e-6)


@pytest.mark.parametrize("C,T", [(5, 37), (32, 20), (33, 65), (64, 12), (100, 150), (104, 9), (105, 40), (128, 33),
                                 (129, 40), (150, 130), (160, 33), (161, 21), (187, 64), (192, 45)])
def test_dense_probability_domain_sweeps_every_padding_bucket(C, T):
    """the probability-domain sweeps (C <= 192: four chain waves up to 128 classes, five / six beyond) on well-conditioned
    data: served without fallback, loss / emission gradient / transition gradient match the float64 recurrences"""
    rs = np.random.RandomState(C * 7 + T)
    B = 3
    x = (2.0 * rs.randn(B, T, C)).astype(np.float32)
    W = rs.randn(C + 1, C).astype(np.float32)
    _dense_check(x, W, [False] * B)


@pytest.mark.parametrize("C,T", [(193, 2), (193, 40), (200, 9), (200, 61), (255, 8), (256, 17), (257, 5), (257, 33),
                                 (300, 26), (320, 12), (320, 45), (321, 7)])
def test_dense_register_resident_sweeps_beyond_the_on_chip_limit(C, T):
    """193 .. 320 classes: the frames of an utterance's sweep run inside ONE workgroup whose registers hold the transition
    matrix (csrc/dense_wide.h wide_resident_sweep_kernel: four rows x four column chunks per lane up to 256 classes, five
    x five up to 320; the stored format is the per-frame launches', which 321 classes still take) -- every bucket edge,
    utterance lengths around the emission prefetch depth (whole groups of 8 / 4 steps, a tail, a single step), an
    emission row holding -inf and NaN entries, a dead utterance: loss / emission gradient / transition gradient against
    the float64 recurrences."""
    rs = np.random.RandomState(C * 11 + T)
    B = 3
    x = (2.0 * rs.randn(B, T, C)).astype(np.float32)
    W = rs.randn(C + 1, C).astype(np.float32)
    if T > 4:
        x[1, 2, ::3] = -np.inf
        x[1, 3, 5] = np.nan
        x[2, 1, :] = -np.inf  # no path through frame 1: logZ = -inf, zero gradient
    _dense_check(x, W, [False] * B)


def test_dense_sweep_emissions_that_are_only_four_byte_aligned():
    """With an even class count the sweep's helper wave loads a row's scores as 8-byte pairs; a tensor that starts on an
    odd float (a view into a larger buffer) takes the two-load form instead: same results either way."""
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(5)
    B, T, C = 2, 70, 100
    x = (2.0 * rs.randn(B, T, C)).astype(np.float32)
    W = rs.randn(C + 1, C).astype(np.float32)
    Wt = dev(W)
    buf = torch.zeros(B * T * C + 1, device="cuda")
    odd = buf[1:].view(B, T, C)
    odd.copy_(dev(x))
    assert odd.data_ptr() % 8 == 4 and odd.is_contiguous()
    a, b = E.dense_forward(dev(x), Wt, need_beta=True), E.dense_forward(odd, Wt, need_beta=True)
    np.testing.assert_allclose(b.logz.cpu().numpy(), a.logz.cpu().numpy(), rtol=1e-6)
    want = [OR.dense_forward_backward(x[i], W) for i in range(B)]
    coef = torch.ones(B, device="cuda")
    dx = torch.empty_like(odd)
    E.dense_grad(odd, Wt, b, coef, coef_w=coef, dx=dx, dW=None)
    close(dx, np.stack([w[1] for w in want]))
    assert E.dense_flagged(b).cpu().tolist() == [False] * B


@pytest.mark.parametrize("C", [130, 188, 200])
def test_asg_beyond_128_classes(crit, C):
    """ASG above the 128 classes of the four-wave register-resident sweeps: five / six chain waves up to 192 classes,
    the batched per-frame product (csrc/dense_wide.h) beyond"""
    rs = np.random.RandomState(C)
    B, T = 2, 40
    x = rs.randn(B, T, C).astype(np.float32)
    W = (0.3 * rs.randn(C + 1, C)).astype(np.float32)
    targets = [rs.randint(0, C, size=n).tolist() for n in (5, 9)]
    xt, Wt = dev(x, grad=True), dev(W, grad=True)
    loss = crit["asg"].ASGLoss(xt, Wt, targets, "mean")
    loss.backward()
    want = OR.asg_loss_grad(x, W, targets, "mean")
    assert loss.item() == pytest.approx(want[0], rel=RTOL)
    close(xt.grad, want[1])
    close(Wt.grad, want[2], atol=2e-4)
    # beyond the on-chip limit the same call runs on the batched per-frame product (csrc/dense_wide.h; parity at 333 and
    # 1000 classes: tests/test_gpu_configs.py::test_asg_beyond_the_on_chip_class_limit)
    big = crit["asg"].ASGLoss(dev(rs.randn(1, 10, 300).astype(np.float32)), dev(np.zeros((301, 300), np.float32)), [[1, 2]], "mean")
    assert np.isfinite(big.item())


def test_dense_more_classes_than_the_fast_path_supports():
    rs = np.random.RandomState(5)
    x = rs.randn(2, 25, 230).astype(np.float32)
    W = (0.3 * rs.randn(231, 230)).astype(np.float32)
    _dense_check(x, W, [False, False], need_dw=False)  # beyond 192 classes: the batched per-frame product (dense_wide.h)


def test_dense_range_flags_hand_utterances_to_the_log_domain_kernels():
    """what the probability domain cannot hold must be detected per utterance and recomputed in the
    log domain with the reference's semantics: -inf emissions, a frame of extreme dynamic range,
    and (for the whole batch) hard constraints in W"""
    rs = np.random.RandomState(11)
    B, T, C = 4, 60, 20
    x = rs.randn(B, T, C).astype(np.float32)
    W = (0.5 * rs.randn(C + 1, C)).astype(np.float32)
    x[1, 10, 3] = -np.inf              # an impossible emission
    x[2, 20:24, :] -= 90.0 * np.arange(C, dtype=np.float32) / C  # 90 nats of spread inside a frame
    x[3, 5, :] = np.nan                # NaN policy: an impossible frame -> no path at all
    _dense_check(x, W, [False, True, True, True])
    W2 = W.copy()
    W2[1 + 4, 7] = -np.inf             # a forbidden transition: the whole batch takes the log-domain path
    x2 = rs.randn(2, 30, C).astype(np.float32)
    _dense_check(x2, W2, [True, True])
    W3 = W.copy()
    W3[0, :] = -np.inf                 # only start weights are impossible except one class
    W3[0, 2] = 0.0
    _dense_check(x2, W3, [True, True])


def test_dense_baseline_shape_properties():
    """cfg3 sizes (T=1000, C=100): no oracle at this size in seconds -- size-independent properties:
    posteriors of every frame sum to the gradient coefficient, transition posteriors sum to T-1,
    start posteriors to 1, and the fast sweep agrees with the log-domain kernels"""
    from gtn_applications_amd import engine as E
    from gtn_applications_amd import _native as N

    g = torch.Generator().manual_seed(3)
    B, T, C = 8, 1000, 100
    x = torch.randn(B, T, C, generator=g).cuda()
    W = torch.randn(C + 1, C, generator=g).cuda()
    st = E.dense_forward(x, W)
    assert not E.dense_flagged(st).any()
    coef = torch.ones(B, device="cuda")
    dx, dW = torch.empty_like(x), torch.zeros_like(W)
    E.dense_grad(x, W, st, coef, coef_w=coef, dx=dx, dW=dW)
    np.testing.assert_allclose(dx.sum(dim=2).cpu().numpy(), 1.0, rtol=2e-4)
    assert float(dW[0].sum()) == pytest.approx(B, rel=2e-4)
    assert float(dW[1:].sum()) == pytest.approx(B * (T - 1), rel=2e-4)
    # same batch through the log-domain kernels: make the fast path unavailable with one -inf in a copy of W that
    # no path needs (state 0 can still be reached from every other state)
    Wh = W.clone()
    Wh[1, 0] = float("-inf")
    st_h = E.dense_forward(x, Wh)
    assert E.dense_flagged(st_h).all()
    W_soft = W.clone()
    W_soft[1, 0] = -35.0  # numerically the same constraint (e^-35), still inside the fast path's range check
    st_s = E.dense_forward(x, W_soft)
    assert not E.dense_flagged(st_s).any()
    np.testing.assert_allclose(st_s.logz.cpu().numpy(), st_h.logz.cpu().numpy(), rtol=1e-5)
    dxs, dxh = torch.empty_like(x), torch.empty_like(x)
    dWs, dWh = torch.zeros_like(W), torch.zeros_like(W)
    E.dense_grad(x, W_soft, st_s, coef, coef_w=coef, dx=dxs, dW=dWs)
    E.dense_grad(x, Wh, st_h, coef, coef_w=coef, dx=dxh, dW=dWh)
    # (the log-domain launches carry doubles since round 4: they used to keep plain fp32 log scores, O(5000) at T=1000,
    # and agreed with the fast sweep to 1e-2 only)
    np.testing.assert_allclose(dxs.cpu().numpy(), dxh.cpu().numpy(), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(dWs.cpu().numpy(), dWh.cpu().numpy(), rtol=1e-2, atol=1e-3)


# =================================================================================================
# STC
# =================================================================================================
def test_stc_reference_literals(crit, lit):
    stc = crit["stc"]
    c = lit["stc_trivial"]
    lp = torch.log(dev(c["probs_TBN"]).view(c["T"], 1, c["N"]))
    assert stc.STC(0, 1, 1, 1)(lp, c["labels"]).item() == pytest.approx(0.0, abs=1e-6)
    c = lit["stc_uniform"]
    lp = torch.log_softmax(torch.zeros(c["T"], 1, c["N"], device="cuda"), 2)
    assert stc.STC(0, 1, 1, 1, "none")(lp, c["labels"]).item() == pytest.approx(c["loss"], abs=1e-5)


def test_stc_golden_cases(crit, cases):
    stc = crit["stc"]
    for name, c in cases.items():
        if c["kind"] != "stc":
            continue
        x = dev(c["inputs"], grad=True)
        m = stc.STC(0, c["p0"], c["plast"], c["thalf"], c["reduction"])
        m.eval()
        loss = m(torch.log_softmax(x, 2), c["targets"])
        loss.backward()
        assert loss.item() == pytest.approx(c["loss"], rel=RTOL, abs=1e-5), name
        close(x.grad, c["grad"], msg=name)


def test_stc_module_fused_augmentation_equals_the_torch_spelling(crit):
    """stc.py:199-220 (select the batch's classes, <star> = logsumexp over the non-blank classes, <star>\\token) as one
    launch each way (wfl_stc_augment / _grad: device inputs) against the same steps written with torch ops (host
    inputs take that path): loss and the gradient w.r.t. the (T, B, C) log-probabilities, also with a class count that
    is not a multiple of the wave, a -inf log-probability and a batch that uses few of the classes."""
    stc = crit["stc"]
    rs = np.random.RandomState(21)
    for (T, B, C, lens) in [(37, 3, 11, (4, 1, 7)), (60, 2, 130, (9, 12)), (25, 4, 65, (3, 3, 2, 5))]:
        x = torch.log_softmax(torch.tensor(rs.randn(T, B, C).astype(np.float32)), 2)
        x[3, 0, 2] = float("-inf")
        targets = [rs.randint(1, min(C, 20), size=n).tolist() for n in lens]
        res = []
        for device in ("cpu", "cuda"):
            xi = x.detach().clone().to(device).requires_grad_(True)
            m = stc.STC(0, 0.4, 0.1, 50, "mean")
            m.nstep = 7
            loss = m(xi, targets)
            (3.0 * loss).backward()
            res.append((loss.item(), xi.grad.cpu().numpy()))
        assert res[1][0] == pytest.approx(res[0][0], rel=2e-5)
        np.testing.assert_allclose(res[1][1], res[0][1], rtol=2e-4, atol=2e-6)


def test_stc_module_rejects_bad_labels_and_keeps_the_reference_select_list(crit):
    """Device inputs: a label outside [0, C) raises as the reference's index_select does (stc.py:207) -- negative ones
    too, which a host-side fancy index would wrap around silently -- instead of reaching the augmentation kernel; a
    target that names the blank (column 0 twice in the reference's select list, stc.py:205) and a batch without frames
    take the torch spelling and give what host inputs give."""
    stc = crit["stc"]
    rs = np.random.RandomState(22)
    T, B, C = 20, 2, 9
    x = torch.log_softmax(torch.tensor(rs.randn(T, B, C).astype(np.float32)), 2)
    m = stc.STC(0, 0.5, 0.5, 10, "mean")
    for bad in ([[1, 2], [3, C]], [[1, -1], [2]], [[-C, 1], [2]]):
        with pytest.raises(IndexError):
            m(x.cuda().requires_grad_(True), bad)
    with_blank = [[1, 0, 2], [3]]
    res = []
    for device in ("cpu", "cuda"):
        xi = x.detach().clone().to(device).requires_grad_(True)
        loss = stc.STC(0, 0.5, 0.5, 10, "mean")(xi, with_blank)
        loss.backward()
        res.append((loss.item(), xi.grad.cpu().numpy()))
    assert res[1][0] == pytest.approx(res[0][0], rel=2e-5)
    np.testing.assert_allclose(res[1][1], res[0][1], rtol=2e-4, atol=2e-6)


def test_stc_function_vs_oracle(crit):
    rs = np.random.RandomState(9)
    B, T, Cp = 3, 40, 6  # Cp selected columns -> Cstar = 2*Cp
    x = rs.randn(B, T, 2 * Cp).astype(np.float32)
    targets = [rs.randint(1, Cp, size=n).tolist() for n in (5, 1, 8)]
    want_loss, want_dx = OC.stc_function(x, targets, 0.3, "mean")
    xt = dev(x, grad=True)
    loss = crit["stc"].STCLoss(xt, targets, 0.3, "mean")
    loss.backward()
    assert loss.item() == pytest.approx(want_loss, rel=RTOL)
    close(xt.grad, want_dx)


# =================================================================================================
# Transducer
# =================================================================================================
def _transducer_from_case(tr, c):
    toks = [tuple(t) if isinstance(t, list) else t for t in c["tokens"]]
    g2i = {(int(k) if c["grapheme_keys_are_int"] else k): v for k, v in c["graphemes_to_idx"].items()}
    return tr.Transducer(toks, g2i, **c["kwargs"]).cuda()


def test_transducer_reference_literals(crit, lit):
    tr = crit["transducer"]
    c = lit["transducer_trivial"]
    lp = torch.log(dev(c["probs"]).view(1, c["T"], c["N"]))
    for k in c["cases"]:
        m = tr.Transducer(k["tokens"], k["g2i"], blank=k["blank"], allow_repeats=k["allow_repeats"])
        assert m(lp, k["labels"]).item() == pytest.approx(0.0, abs=1e-6)
    c = lit["ctc_uniform"]
    m = tr.Transducer(["a", "b", "c"], {"a": 0, "b": 1, "c": 2}, blank="optional")
    lp = torch.log_softmax(torch.zeros(1, c["T"], c["N"], device="cuda"), 2)
    assert m(lp, c["labels"]).item() == pytest.approx(-math.log(0.25 ** 3 * 5), abs=1e-5)
    for key, rep in (("ctc_5x6", True), ("ctc_5x6_repeat", False)):
        c = lit[key]
        le = torch.log(dev(c["probs"]).view(1, c["T"], c["N"])).requires_grad_(True)
        m = tr.Transducer(["a", "b", "c", "d", "e"], {k: i for i, k in enumerate("abcde")}, blank="optional",
                          allow_repeats=rep)
        loss = m(le, c["labels"])
        assert loss.item() == pytest.approx(c["loss"], abs=5e-5)
        loss.backward()
        close(le.grad.view(-1), c["grad"], rtol=1e-4, atol=1e-6, msg=key)
    v = lit["transducer_viterbi"]
    em = torch.stack([dev(v["emissions1"]).view(v["T"], v["N"]), dev(v["emissions2"]).view(v["T"], v["N"])])
    toks = v["no_blank"]["tokens"]
    m = tr.Transducer(toks, {t: i for i, t in enumerate(toks)}, blank="none")
    assert [p.tolist() for p in m.viterbi(em)] == v["no_blank"]["labels"]
    toks = v["blank_norepeat"]["tokens"]
    m = tr.Transducer(toks, {t: i for i, t in enumerate(toks)}, blank="optional", allow_repeats=False)
    assert [p.tolist() for p in m.viterbi(em)] == v["blank_norepeat"]["labels"]


def _word_piece_batch(B, T, seed, pieces=15):
    """a batch of the Transducer benchmark's kind (benchmarks/transducer_benchmark.py:18-53): alignment graphs of
    ~260 states, the size at which the sweeps run as 512-thread workgroups and the gradient beside them"""
    import random

    import bench

    tokens, g2i = bench.word_pieces()
    rnd = random.Random(seed)
    x = torch.randn(B, T, len(tokens) + 1, generator=torch.Generator().manual_seed(seed)).cuda()
    tg = [torch.tensor([g2i[ch] for _ in range(pieces) for ch in rnd.choice(tokens)]) for _ in range(B)]
    return tokens, g2i, x, tg


@pytest.mark.parametrize("leaf", [True, False])
def test_transducer_native_call_is_the_python_sequence(crit, monkeypatch, leaf):
    """csrc/torch_ops.cpp::lattice_loss_forward issues the launches of a Transducer step without a transition model
    (transducer.py:239-315) in one native call; criterions/transducer.py keeps the same sequence in Python.  Same
    kernels, same buffers: the loss bit for bit, the gradient to the last digits (the gradient beside the sweeps
    normalises tile by tile) -- as leaves and through the autograd engine with a grad_output that is not 1."""
    tr = crit["transducer"]
    if tr._native_node() is None:
        pytest.skip("the torch extension is not built")
    tokens, g2i, x, tg = _word_piece_batch(6, 120, 3)
    m = tr.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")

    def run():
        xi = x.clone().requires_grad_(True)
        loss = m(xi if leaf else xi * 1.0, tg)
        (loss if leaf else loss * 0.75).backward()
        return loss.detach().cpu().numpy(), xi.grad.cpu().numpy()

    native = run()
    monkeypatch.setattr(tr, "_NODE", None)
    python = run()
    assert np.array_equal(native[0], python[0])
    np.testing.assert_allclose(native[1], python[1], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("B,T,mitm", [(6, 200, "0"), (70, 48, "1"), (6, 208, "2"), (9, 64, "2"), (6, 200, "2"), (3, 330, "1"),
                                      (6, 200, "0/one-phase"), (6, 208, "2/one-phase")])
def test_transducer_gradient_beside_the_sweeps_equals_the_gradient_in_backward(crit, monkeypatch, B, T, mitm):
    """csrc/lattice_kernels.hip wfl_lattice_forward_grad: the emission gradient computed by the persistent workgroups
    that follow the two sweeps (tile-local log Z, L1-bypassing reads of alpha / beta) against the gradient kernel that
    runs after them in backward -- through `loss.backward()` (the buffer becomes .grad as it is) and through the
    autograd engine with a grad_output that is not 1 (the buffer is scaled).  B = 70: more utterances than the gate
    kernel's wave has lanes, not a multiple of the 8 XCDs.  mitm = WFL_LATTICE_MITM: "2" the sweeps meet in the middle
    from 64 frames on and the workgroups read occupancies (csrc/lattice_kernels.hip run_chain_prob, "meeting the
    partner"), "1" (the default) from 320 frames on, "0" never."""
    tr = crit["transducer"]
    # ("/one-phase": WFL_LATTICE_TWO_PHASE=0, whole rows per tile instead of base rows first and occupancies added behind
    # the sweeps -- csrc/lattice_kernels.hip occ_live_kernel)
    monkeypatch.setenv("WFL_LATTICE_MITM", mitm.split("/")[0])
    monkeypatch.setenv("WFL_LATTICE_TWO_PHASE", "0" if mitm.endswith("one-phase") else "1")
    tokens, g2i, x, tg = _word_piece_batch(B, T, 11)
    m = tr.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")

    def run(scale):
        xi = x.clone().requires_grad_(True)
        loss = m(xi, tg)
        (loss if scale is None else loss * scale).backward()
        return loss.detach().clone(), xi.grad.clone(), type(loss).__name__

    monkeypatch.setattr(tr, "_IN_LAUNCH_GRAD", False)
    ref_loss, ref_dx, kind = run(None)
    assert kind == "Tensor"
    monkeypatch.setattr(tr, "_IN_LAUNCH_GRAD", True)
    loss, dx, kind = run(None)
    assert kind == "EagerLoss"
    assert loss.item() == pytest.approx(ref_loss.item(), rel=1e-6)
    close(dx, ref_dx.cpu().numpy(), rtol=1e-4, atol=1e-6, msg="loss.backward()")
    loss, dx, _ = run(2.5)
    close(dx, 2.5 * ref_dx.cpu().numpy(), rtol=1e-4, atol=2.5e-6, msg="autograd engine, grad_output 2.5")
    # .grad accumulates; a second backward through the same graph raises, as after a pass of the engine (the graph is spent)
    xi = x.clone().requires_grad_(True)
    m(xi, tg).backward()
    loss = m(xi, tg)
    loss.backward()
    close(xi.grad, 2 * ref_dx.cpu().numpy(), rtol=1e-4, atol=2e-6, msg="accumulated .grad")
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
    assert loss.grad_fn.aux is None  # (the forward's buffers went with the graph)
    # hooks on the loss or on its node only run inside the engine: backward() must go there, same numbers
    for how in ("tensor hook", "retain_grad", "node hook", "node prehook"):
        xi = x.clone().requires_grad_(True)
        loss = m(xi, tg)
        seen = []
        if how == "tensor hook":
            loss.register_hook(lambda g: seen.append(float(g)))
        elif how == "retain_grad":
            loss.retain_grad()
        elif how == "node hook":
            loss.grad_fn.register_hook(lambda gi, go: seen.append(1.0))
        else:
            loss.grad_fn.register_prehook(lambda go: seen.append(1.0))
        loss.backward()
        close(xi.grad, ref_dx.cpu().numpy(), rtol=1e-4, atol=1e-6, msg=how)
        if how == "retain_grad":
            assert float(loss.grad) == 1.0
        else:
            assert seen == [1.0], how
    # through the engine: the buffer scaled in place on the first pass, recomputed on the second (retained graph)
    xi = x.clone().requires_grad_(True)
    loss = m(xi, tg) * 1.5
    loss.backward(retain_graph=True)
    loss.backward()
    close(xi.grad, 3 * ref_dx.cpu().numpy(), rtol=1e-4, atol=3e-6, msg="retain_graph")


def test_eager_loss_starts_the_engine_at_inputs_that_are_not_plain_leaves(crit, monkeypatch):
    """engine.EagerLoss.backward when an input is a model's output (train.py:262-266), an nn.Parameter (the ASG module;
    DistributedDataParallel hangs its all-reduce on the parameter's AccumulateGrad node) or a leaf with a hook: the
    gradients the forward launches computed are handed to ONE pass of the autograd engine that starts at those inputs
    (torch.autograd.backward(inputs, grads)) -- the producer's backward, tensor hooks, retain_grad and the parameter's
    accumulation behave as under torch.Tensor.backward from the loss, with the same numbers; ASG and Transducer."""
    from gtn_applications_amd import engine as E

    asg, tr = crit["asg"], crit["transducer"]
    rs = np.random.RandomState(31)
    B, T, C = 5, 60, 9
    x = torch.tensor(rs.randn(B, T, C).astype(np.float32))
    W = torch.tensor((0.3 * rs.randn(C + 1, C)).astype(np.float32))
    wcol = torch.tensor(rs.rand(C).astype(np.float32) + 0.5).cuda()
    targets = [rs.randint(0, C, size=n).tolist() for n in (4, 7, 1, 9, 3)]
    started = []
    real = torch.autograd.backward

    def spy(tensors, grad_tensors=None, *a, **k):
        started.append(len(tensors) if isinstance(tensors, (list, tuple)) else 1)
        return real(tensors, grad_tensors, *a, **k)

    def asg_run(eager, how):
        monkeypatch.setattr(asg, "_EARLY_GRAD", eager)
        del started[:]
        xi = x.cuda().requires_grad_(True)
        par = torch.nn.Parameter(W.cuda().clone())
        seen = {}
        em = xi * wcol if how != "leaf_hook" else xi
        if how == "em_hook":
            em.register_hook(lambda g: seen.setdefault("em", g.clone()))
        if how == "leaf_hook":
            xi.register_hook(lambda g: seen.setdefault("leaf", g.clone()))
        if how == "retain_grad":
            em.retain_grad()
        loss = asg.ASGLoss(em, par, targets, "mean")
        monkeypatch.setattr(torch.autograd, "backward", spy)
        try:
            loss.backward()
        finally:
            monkeypatch.setattr(torch.autograd, "backward", real)
        if how == "retain_grad":
            seen["em_grad"] = em.grad.clone()
        return xi.grad.clone(), par.grad.clone(), seen, type(loss).__name__, list(started)

    for how in ("product", "em_hook", "leaf_hook", "retain_grad"):
        want = asg_run(False, how)
        got = asg_run(True, how)
        assert want[3] == "Tensor" and got[3] == "EagerLoss", how
        # (torch.Tensor.backward calls torch.autograd.backward once with the loss; the eager route with both inputs)
        assert got[4] == [2], (how, got[4])
        close(got[0], want[0].cpu().numpy(), rtol=1e-5, atol=1e-7, msg=how + " dx")
        close(got[1], want[1].cpu().numpy(), rtol=1e-5, atol=1e-7, msg=how + " dW")
        assert sorted(got[2]) == sorted(want[2]), how
        for k in want[2]:
            close(got[2][k], want[2][k].cpu().numpy(), rtol=1e-5, atol=1e-7, msg=how + " " + k)
    # plain leaves still get .grad without any engine pass
    monkeypatch.setattr(asg, "_EARLY_GRAD", True)
    xi, Wi = x.cuda().requires_grad_(True), W.cuda().requires_grad_(True)
    del started[:]
    monkeypatch.setattr(torch.autograd, "backward", spy)
    try:
        asg.ASGLoss(xi, Wi, targets, "mean").backward()
    finally:
        monkeypatch.setattr(torch.autograd, "backward", real)
    assert started == [] and xi.grad is not None and Wi.grad is not None
    # Transducer: the gradient beside the sweeps handed to the producer of non-leaf emissions
    tokens, g2i, xt, tg = _word_piece_batch(6, 120, 5)
    m = tr.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")

    def tr_run(in_launch):
        monkeypatch.setattr(tr, "_IN_LAUNCH_GRAD", in_launch)
        xi = xt.clone().requires_grad_(True)
        em = xi * 1.25
        got_hook = []
        em.register_hook(lambda g: got_hook.append(g.clone()))
        loss = m(em, tg)
        loss.backward()
        return xi.grad.clone(), got_hook[0], type(loss).__name__

    want, got = tr_run(False), tr_run(True)
    assert want[2] == "Tensor" and got[2] == "EagerLoss"
    close(got[0], want[0].cpu().numpy(), rtol=1e-4, atol=1e-6, msg="transducer dx")
    close(got[1], want[1].cpu().numpy(), rtol=1e-4, atol=1e-6, msg="transducer hook")
    assert E.may_hand_over(torch.nn.Parameter(torch.zeros(2, device="cuda"))) and not E.may_hand_over(torch.zeros(2, requires_grad=True))


def test_transducer_gradient_beside_the_sweeps_under_cu_contention(crit):
    """The gradient workgroups wait on progress words of the sweeps (another launch, another stream), the gate kernel on
    the sweeps' announcement: safe only while everything gets CUs eventually.  A third stream keeps every CU busy with
    GEMMs and device-wide copies while 12 steps run: every step must give the uncontended loss and gradient (to the
    bar between two orders of the same float32 sums: a tile's own log Z), and none may fall back wholesale (the
    probability-domain formats stay)."""
    tr = crit["transducer"]
    tokens, g2i, x, tg = _word_piece_batch(24, 320, 13)
    m = tr.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")

    def run():
        xi = x.clone().requires_grad_(True)
        loss = m(xi, tg)
        loss.backward()
        return loss.detach().clone(), xi.grad.clone()

    ref_loss, ref_dx = run()
    torch.cuda.synchronize()
    side = _stream_that_runs_beside_the_current_one()
    a = torch.randn(8192, 8192, device="cuda")
    big = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
    stop = torch.zeros((), device="cuda")
    with torch.cuda.stream(side):  # ~1 s of all-CU work queued ahead
        for _ in range(60):
            c = a @ a
            big.copy_(big.roll(1)[: big.numel()])
            stop += c[0, 0] * 0
    for step in range(12):
        loss, dx = run()
        torch.cuda.current_stream().synchronize()
        assert loss.item() == pytest.approx(ref_loss.item(), rel=1e-6), f"step {step}"
        close(dx, ref_dx.cpu().numpy(), rtol=1e-4, atol=1e-6, msg=f"step {step}")
    busy_during = not side.query()
    torch.cuda.synchronize()
    assert busy_during, "the competing stream did not outlast the steps: the test did not exercise contention"


_GATE_SCRIPT = r"""
import json, sys, torch
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(tests)r)
import numpy as np
from gtn_applications_amd import engine as E
from gtn_applications_amd.criterions import transducer as tr
from test_gpu_parity import _word_piece_batch
tokens, g2i, x, tg = _word_piece_batch(6, %(T)d, 11)
m = tr.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
out = []
for step in range(14):
    xi = x.clone().requires_grad_(True)
    loss = m(xi, tg)
    loss.backward()
    torch.cuda.synchronize()
    out.append((loss.item(), float(xi.grad.double().abs().sum()), float(xi.grad.double().sum(dim=2).abs().max()), E.lattice_diagnostics()))
np.save(sys.argv[1], xi.grad.cpu().numpy())
print("RESULT " + json.dumps(out))
"""


@pytest.mark.parametrize("T,mitm", [(200, "0"), (200, "2"), (208, "2")])  # ("2": the sweeps meet in the middle and leave occupancies)
def test_transducer_gradient_beside_the_sweeps_gate_gives_up_cleanly_and_reports_it(crit, tmp_path, T, mitm):
    """A stack that runs the kernels of different streams one after the other (a counter-collecting profiler, a
    debugger): the gate kernel in front of the gradient workgroups cannot see the sweeps.  WFL_LATTICE_FUSED_SERIAL=1
    puts it in front of them on the caller's stream (what such a stack does to the launch order).  It must give up
    within its bound (microseconds, not seconds), write nothing into the sweeps' buffers, and the call must fall back
    to the plain gradient for every row -- same loss, same gradient as the overlapped step -- and SAY so:
    wfl_lattice_diagnostics counts the give-ups, the following calls back off (1, 2, 4, ... calls on the plain path
    between attempts).  In a process of its own (the switches are read once)."""
    import json
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _GATE_SCRIPT % dict(root=root, tests=os.path.join(root, "tests"), T=T)

    def run(env_extra, name):
        env = dict(os.environ)
        env.update(env_extra)
        env["WFL_LATTICE_MITM"] = mitm
        t0 = time.time()
        res = subprocess.run([sys.executable, "-c", script, str(tmp_path / name)], capture_output=True, text=True, env=env,
                             timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1]
        return json.loads(line[7:]), np.load(tmp_path / (name + ".npy")), time.time() - t0

    ref, ref_dx, _ = run({}, "ref")
    d = ref[-1][3]
    # (the very first launch of a process may load the sweeps' code for longer than the gate waits: one give-up, one
    # call on the plain path, at most)
    assert d["launched"] >= 13 and d["gate_gave_up"] <= 1 and d["gate_ok"] >= 12 and d["skipped_in_backoff"] <= 1, d
    got, dx, _ = run({"WFL_LATTICE_FUSED_SERIAL": "1"}, "serial")
    for step, (a, b) in enumerate(zip(ref, got)):
        assert b[0] == pytest.approx(a[0], rel=1e-6), step
        assert b[1] == pytest.approx(a[1], rel=1e-5), step
        assert b[2] <= 1e-6, step  # rows through the fused log_softmax sum to zero: no row was left unwritten
    close(dx, ref_dx, rtol=1e-4, atol=1e-6, msg="gate gave up")
    d = got[-1][3]
    # steps 0, 2, 5 and 10 launch and give up; the 1, 2, 4 and (so far 3 of) 8 calls behind them back off
    assert d["gate_gave_up"] == 4 and d["gate_ok"] == 0 and d["launched"] == 4 and d["skipped_in_backoff"] == 10, d
    assert d["gate_spins"] <= 1 << 12, d  # the bound: milliseconds per give-up, not seconds


@pytest.mark.parametrize("T,mitm", [(150, "0"), (150, "2"), (160, "2")])  # ("2": occupancies; the rest launch forms the rows from them)
def test_transducer_gradient_beside_the_sweeps_falls_back_through_the_certificate(crit, monkeypatch, T, mitm):
    """WFL_LATTICE_FUSED_BADXCD=1 makes the gate kernel report every utterance as swept on two XCDs (what a different
    workgroup-to-XCD dealing would look like): no gradient workgroup may touch them, the certificate sends them to the
    log-domain sweeps and wfl_lattice_grad_rest writes their rows -- same loss, same gradient."""
    tr = crit["transducer"]
    monkeypatch.setenv("WFL_LATTICE_MITM", mitm)
    tokens, g2i, x, tg = _word_piece_batch(5, T, 12)
    m = tr.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")

    def run():
        xi = x.clone().requires_grad_(True)
        loss = m(xi, tg)
        loss.backward()
        return loss.item(), xi.grad.clone()

    ref_loss, ref_dx = run()
    monkeypatch.setenv("WFL_LATTICE_FUSED_BADXCD", "1")
    loss, dx = run()
    assert loss == pytest.approx(ref_loss, rel=2e-5)
    close(dx, ref_dx.cpu().numpy(), rtol=2e-3, atol=2e-6, msg="fall-back")  # (fp32 log-domain sweeps against fp64 probabilities)


def test_transducer_two_phase_gradient_over_many_steps_back_to_back(crit, monkeypatch):
    """csrc/lattice_kernels.hip occ_live_kernel: the gradient workgroups beside the sweeps write the rows' base values
    while the sweeps are in their first half and add the label columns' occupancies behind them -- tiles of one launch
    wait for words (`based`) that workgroups of the same launch write, launches follow each other without a host
    synchronisation, buffers come back from the caching allocator.  120 steps back to back on two alternating batches
    against WFL_LATTICE_TWO_PHASE=0 (whole rows per tile): the loss bit for bit, the gradient to the order of the LDS
    float additions."""
    tr = crit["transducer"]
    batches = [_word_piece_batch(12, 416, 31 + k) for k in range(2)]
    m = tr.Transducer(batches[0][0], batches[0][1], blank="optional", allow_repeats=False, reduction="mean")

    def run(k):
        x = batches[k][2].clone().requires_grad_(True)
        loss = m(x.view_as(x), batches[k][3])
        loss.backward()
        return loss.detach(), x.grad

    monkeypatch.setenv("WFL_LATTICE_TWO_PHASE", "0")
    ref = [run(k) for k in range(2)]
    torch.cuda.synchronize()
    monkeypatch.setenv("WFL_LATTICE_TWO_PHASE", "1")
    got = [(it & 1,) + run(it & 1) for it in range(120)]
    torch.cuda.synchronize()
    for k, loss, dx in got:
        assert float(loss) == float(ref[k][0])
        close(dx, ref[k][1].cpu().numpy(), rtol=1e-5, atol=1e-9, msg="two phases")


def _sweep_formats(loss, B, T):
    """fmt[b] of the numerator sweeps behind a Transducer loss (0 log domain, 1 probability domain, 2 met in the middle)"""
    import ctypes

    from gtn_applications_amd import _native as N

    num = loss.grad_fn.aux[2]
    off = ctypes.c_int64()
    N.check(N.lib.wfl_lattice_formats_offset(ctypes.byref(num.pack.desc), T, ctypes.byref(off)))
    torch.cuda.synchronize()
    return num.alpha[off.value:off.value + B].view(torch.int32).cpu().tolist()


@pytest.mark.parametrize("B,T", [(5, 64), (7, 160), (3, 400), (5, 65), (6, 79), (4, 97), (7, 203), (3, 250), (2, 401),
                                 (3, 175), (4, 72), (4, 137), (4, 119)])
def test_transducer_sweeps_that_meet_in_the_middle_equal_the_full_sweeps(crit, monkeypatch, B, T):
    """csrc/lattice_kernels.hip run_chain_prob: from 64 frames on the two sweeps of an utterance store their own vector up
    to the middle slot and state occupancies (floats, normalised by the Z formed at the middle) beyond it, reading the
    partner's vectors from L2 -- WFL_LATTICE_MITM=0 keeps both vectors everywhere.  The forward sweep's arithmetic is
    untouched: the loss bit for bit; the gradient to the rounding of float occupancies; the formats say which one ran.
    T not a multiple of 16: the sweeps' chunk boundaries differ by T % 16, the slots both hold are converted at the
    forward sweep's crossing and each sweep's last, partial chunk after it was stored (gamma_rows).  Shorter
    utterances keep the full sweeps."""
    tr = crit["transducer"]
    tokens, g2i, x, tg = _word_piece_batch(B, T, 17)
    m = tr.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")

    def run():
        xi = x.clone().requires_grad_(True)
        loss = m(xi, tg)
        fm = _sweep_formats(loss, B, T)
        loss.backward()
        return loss.item(), xi.grad.clone(), fm

    monkeypatch.setenv("WFL_LATTICE_MITM", "0")
    ref = run()
    monkeypatch.setenv("WFL_LATTICE_MITM", "2")  # (from 64 frames on; by default from 320)
    got = run()
    assert ref[2] == [1] * B and got[2] == [2] * B, (ref[2], got[2])
    monkeypatch.delenv("WFL_LATTICE_MITM")
    assert run()[2] == [2 if T >= 320 else 1] * B
    assert got[0] == ref[0]
    close(got[1], ref[1].cpu().numpy(), rtol=1e-4, atol=1e-7, msg="occupancies")
    xo = x[:, :63].contiguous()
    xi = xo.clone().requires_grad_(True)
    assert _sweep_formats(m(xi, tg), B, 63) == [1] * B


def test_transducer_equals_ctc(crit, lit):
    """tests/transducer_test.py:275-316: CTC == Transducer(blank optional, no repeats)."""
    c = lit["ctc_compare_targets"]
    T, N, B, tgt = c["T"], c["N"], c["B"], c["targets"]
    x = torch.randn(B, T, N, generator=torch.Generator().manual_seed(4)).cuda().requires_grad_(True)
    toks = [(t,) for t in range(N - 1)]
    for reduction in ("none", "mean"):
        m = crit["transducer"].Transducer(toks, {t: t for t in range(N - 1)}, blank="optional", allow_repeats=False,
                                          reduction=reduction)
        a = crit["ctc"].CTCLoss(torch.log_softmax(x, 2), tgt, N - 1, reduction)
        a.backward()
        ga, x.grad = x.grad, None
        b = m(x, tgt)
        b.backward()
        gb, x.grad = x.grad, None
        assert a.item() == pytest.approx(b.item(), abs=1e-4)
        assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-5)


def test_transducer_golden_cases(crit, cases):
    for name, c in cases.items():
        if c["kind"] != "transducer":
            continue
        m = _transducer_from_case(crit["transducer"], c)
        if "transition_params" in c:
            with torch.no_grad():
                m.transition_params.copy_(dev(c["transition_params"]))
        x = dev(c["inputs"], grad=True)
        loss = m(x, c["targets"])
        loss.backward()
        assert loss.item() == pytest.approx(c["loss"], rel=RTOL, abs=1e-5), name
        close(x.grad, c["grad"], msg=name)
        if "transition_grad" in c:
            close(m.transition_params.grad, c["transition_grad"], atol=2e-5, msg=name)
        assert [p.tolist() for p in m.viterbi(x.detach())] == c["viterbi"], name


def test_transducer_asg_transitions(crit, lit):
    """tests/transducer_test.py:420-532: ASG == Transducer(transitions = ASG graph)."""
    c = lit["asg_3x5x6"]
    N = c["N"]
    trans = crit["asg"].ASGLossFunction.create_transitions_graph(torch.zeros(N + 1, N))
    m = crit["transducer"].Transducer([(n,) for n in range(N)], {n: n for n in range(N)}, transitions=trans).cuda()
    x = dev(c["emissions"], grad=True)
    loss = m(x, c["labels"])
    assert loss.item() == pytest.approx(c["loss"], abs=5e-5)
    loss.backward()
    close(x.grad * c["B"], c["grad_times_B"], rtol=1e-3, atol=2e-4)
    close(m.transition_params.grad[N:].view(N, N) * c["B"], c["trans_grad_rows1_times_B"], rtol=1e-2, atol=2e-4)
    v = lit["asg_viterbi_transducer"]
    trans = crit["asg"].ASGLossFunction.create_transitions_graph(torch.zeros(v["N"] + 1, v["N"]))
    m = crit["transducer"].Transducer([(n,) for n in range(v["N"])], {n: n for n in range(v["N"])},
                                      transitions=trans).cuda()
    with torch.no_grad():
        m.transition_params.copy_(dev(v["transitions"]))
    assert m.viterbi(dev(v["inputs"]).view(1, v["T"], v["N"]))[0].tolist() == v["path"]


def test_transducer_backoff_transitions(crit, lit, tmp_path):
    """tests/transducer_test.py:534-566: epsilon (back-off) arcs; analytic vs oracle gradient and
    vs central differences."""
    from gtn_applications_amd import graph as G
    from oracle import minigtn as MG

    c = lit["backoff_transitions"]
    lines = [" ".join(map(str, c["start"])), " ".join(map(str, c["accept"]))] + [" ".join(map(str, a)) for a in c["arcs"]]
    path = tmp_path / "backoff.txt"
    path.write_text("\n".join(lines) + "\n")
    N = c["N"]
    toks = [(n,) for n in range(N)]
    m = crit["transducer"].Transducer(toks, {n: n for n in range(N)}, blank="optional", allow_repeats=False,
                                      transitions=G.loadtxt(path)).cuda()
    og = MG.loadtxt(str(path))
    oracle = OC.TransducerOracle(toks, {n: n for n in range(N)}, blank="optional", allow_repeats=False, transitions=og)
    rs = np.random.RandomState(0)
    x = rs.randn(1, c["T"], N + 1).astype(np.float32)
    params = (0.3 * rs.randn(len(c["arcs"]))).astype(np.float32)
    oracle.transition_params = params.astype(np.float64)
    want_loss, want_dx, want_dp = oracle.loss(x, c["labels"])
    with torch.no_grad():
        m.transition_params.copy_(dev(params))
    xt = dev(x, grad=True)
    loss = m(xt, c["labels"])
    loss.backward()
    assert loss.item() == pytest.approx(want_loss, rel=RTOL)
    close(xt.grad, want_dx)
    close(m.transition_params.grad, want_dp, atol=2e-5)



@pytest.mark.parametrize("ngram,blank,route", [(1, "optional", "lattice"), (2, "none", "lattice"), (2, "optional", "lattice"),
                                               (2, "none", "dense"), (2, "optional", "dense")])
def test_transducer_dense_ngram_transitions(crit, ngram, blank, route, monkeypatch):
    """30 tokens: the n-gram transition graph has 30-way in-degree.  route "lattice": above the threshold at which a
    whole wavefront relaxes a state (lattice engine, general path); route "dense" (the default for ngram = 2): the
    normaliser forward_score(intersect(emissions, transitions)) (transducer.py:286-288) through the dense transition
    engine, start / bigram / end-arc parameters mapped to its W and back.  Loss, emission gradient,
    transition-parameter gradient (random non-zero parameters, end arcs included) and Viterbi against the graph oracle"""
    tr = crit["transducer"]
    monkeypatch.setattr(tr, "_DENSE_NGRAM", route == "dense")
    ntok = 30
    tokens = [(i,) for i in range(ntok)]
    g2i = {i: i for i in range(ntok)}
    kw = dict(ngram=ngram, blank=blank, allow_repeats=(blank == "none"), reduction="mean")
    rs = np.random.RandomState(40 + ngram)
    B, T = 2, 14
    C = ntok + int(blank != "none")
    x = rs.randn(B, T, C).astype(np.float32)
    targets = [rs.randint(0, ntok, size=5).tolist(), rs.randint(0, ntok, size=3).tolist()]
    m = tr.Transducer(tokens, g2i, **kw)
    assert tr._dense_bigram(m.transitions, C) == (ngram == 2)
    params = (0.3 * rs.randn(m.transition_params.numel())).astype(np.float32)
    with torch.no_grad():
        m.transition_params.copy_(torch.from_numpy(params))
    m.cuda()
    orc = OC.TransducerOracle(tokens, g2i, **kw)
    orc.transition_params = params.astype(np.float64)
    want_loss, want_dx, want_dp = orc.loss(x, targets)
    xt = dev(x, grad=True)
    loss = m(xt, [torch.tensor(t) for t in targets])
    loss.backward()
    assert loss.item() == pytest.approx(want_loss, rel=RTOL)
    close(xt.grad, want_dx)
    close(m.transition_params.grad, want_dp, atol=2e-5)
    assert [p.tolist() for p in m.viterbi(xt.detach())] == orc.viterbi(x)

# =================================================================================================
# ConvTransduce1D
# =================================================================================================
def test_conv_transduce_golden_cases(crit, cases):
    """the reference module (run on the oracle primitives) vs the HIP layer: outputs, input gradient
    and kernel-parameter gradient for the objective sum(out * w), forward and Viterbi scores"""
    tr = crit["transducer"]
    n = 0
    for name, c in cases.items():
        if c["kind"] != "conv":
            continue
        n += 1
        layer = tr.ConvTransduce1D([tuple(l) for l in c["lexicon"]], c["kernel_size"], c["stride"], c["blank_idx"],
                                   **c["kwargs"])
        if "kernel_params" in c:
            with torch.no_grad():
                layer.kernel_params.copy_(torch.tensor(c["kernel_params"]))
            layer.cuda()
        x = dev(np.array(c["inputs"], dtype=np.float32), grad=True)
        out = layer(x)
        close(out, c["outputs"], msg=name)
        (out * dev(np.array(c["out_weights"], dtype=np.float32))).sum().backward()
        close(x.grad, c["grad"], msg=name)
        if "kernel_grad" in c:
            close(layer.kernel_params.grad, c["kernel_grad"], msg=name)
    assert n >= 6


@pytest.mark.parametrize("viterbi,learn,spike,bo", [(False, False, False, True), (False, True, True, False),
                                                    (True, True, False, True), (False, True, False, True)])
def test_conv_transduce_vs_oracle(crit, viterbi, learn, spike, bo):
    """random lexicon with repeats, overlapping windows (stride < kernel), more entries than one
    workgroup pass (K > 16), CPU-resident input"""
    tr = crit["transducer"]
    rs = np.random.RandomState(17 + 2 * viterbi + learn)
    C, blank, ks, stride, B, T = 6, 5, 7, 2, 2, 11
    lexicon = [tuple(rs.randint(0, 5, size=rs.randint(1, 4)).tolist()) for _ in range(21)] + [(1, 1, 1), (2, 2, 3)]
    layer = tr.ConvTransduce1D(lexicon, ks, stride, blank, blank_optional=bo, learn_params=learn, viterbi=viterbi,
                               spike=spike)
    params = None
    if learn:
        params = (0.5 * rs.randn(layer.kernel_params.numel())).astype(np.float32)
        with torch.no_grad():
            layer.kernel_params.copy_(torch.from_numpy(params))
    x = rs.randn(B, T, C).astype(np.float32)
    w = rs.randn(B, (T + 2 * (ks // 2) - ks) // stride + 1, len(lexicon)).astype(np.float32)
    want_out, want_dx, want_dp = OC.conv_layer(x, lexicon, ks, stride, blank, w, blank_optional=bo, learn_params=learn,
                                               viterbi=viterbi, spike=spike, kernel_params=params)
    xt = torch.from_numpy(x).requires_grad_(True)  # CPU tensor in, CPU tensors out
    out = layer(xt)
    assert out.device.type == "cpu" and tuple(out.shape) == want_out.shape
    close(out, want_out)
    (out * torch.from_numpy(w)).sum().backward()
    close(xt.grad, want_dx)
    if learn:
        close(layer.kernel_params.grad, want_dp, atol=5e-5)


def test_conv_transduce_shapes_and_errors(crit):
    """tests/transducer_test.py:57-96: output shapes for every input length, ValueError on T = 0"""
    tr = crit["transducer"]
    lexicon = [(0, 0), (0, 1), (1, 0), (1, 1)]
    conv = tr.ConvTransduce1D(lexicon, 5, 3, 2)
    with pytest.raises(ValueError):
        conv(torch.randn(2, 0, 3))
    for Tin in (1, 2, 3, 4):
        conv(torch.randn(2, Tin, 3))
    for Ti, To in zip((1, 3, 4, 6, 7, 8), (1, 1, 2, 2, 3, 3)):
        x = torch.randn(2, Ti, 3, requires_grad=True)
        out = conv(x)
        assert tuple(out.shape) == (2, To, len(lexicon))
        out.backward(torch.ones_like(out))
        assert torch.isfinite(x.grad).all()
    # an entry that cannot be aligned inside the window (three repeats need tok,blank,tok,blank,tok and, without
    # the optional blank, a final blank: 6 frames > 5): score -inf like forward_score of an empty intersection,
    # and no NaN in the gradient of the other entries
    conv2 = tr.ConvTransduce1D([(0, 0, 0), (1,)], 5, 5, 2, blank_optional=False)
    x = torch.randn(1, 5, 3, requires_grad=True)
    out = conv2(x)
    assert out[0, 0, 0] == float("-inf") and torch.isfinite(out[0, 0, 1])
    out[0, 0, 1].backward()
    assert torch.isfinite(x.grad).all()
    with pytest.raises(ValueError):
        tr.ConvTransduce1D([(0, 0, 0)], 3, 1, 2)  # kernel too small for the entry (transducer.py:425-426)
    with pytest.raises(ValueError):
        tr.ConvTransduce1D(lexicon, 5, 1, 2, scale="cubic")
    with pytest.raises(ValueError):
        tr.ConvTransduce1D(lexicon, 5, 1, 2, normalize="both")


# =================================================================================================
# lattice engine, probability domain (round 2)
# =================================================================================================
def test_lattice_probability_domain_is_the_default_and_matches_log_domain(crit):
    """lean acceptors (CTC-like, force alignment, STC, Transducer alignments) are swept in the fp64 probability
    domain; WFL_LATTICE_DOMAIN=log (read at the first launch of a process) is covered by the subprocess below"""
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(7)
    B, T, C = 3, 300, 9
    x = dev(rs.randn(B, T, C) * 1.5)
    targets = [rs.randint(0, C - 1, size=n).tolist() for n in (300, 0, 17)]  # 300 labels: lattice engine (L > 255), Q = 601
    tg = E.targets_on_device(targets, x.device)
    pack = E.PackedLattice.ctc(tg.flat, tg.offsets, C - 1, C, x.device)
    st = E.lattice_forward(x, pack)
    assert E.lattice_formats(st).tolist() == [1, 1, 1]
    coef = torch.ones(B, device="cuda")
    dx = torch.full_like(x, float("nan"))
    E.lattice_grad(st, coef, dx=dx)
    want_loss, want_dx = OR.ctc_loss_grad_batched(x.cpu().numpy(), targets, C - 1)
    got = st.logz.cpu().numpy()
    assert np.isinf(want_loss[0]) and got[0] == -np.inf and float(dx[0].abs().max()) == 0.0  # 300 labels need > 300 frames
    np.testing.assert_allclose(-got[1:], want_loss[1:], rtol=1e-6)
    close(dx[1:], -want_dx[1:] * B, rtol=1e-5, atol=1e-6)
    import subprocess
    import sys

    code = (
        "import os, sys, numpy as np, torch\n"
        "os.environ['WFL_LATTICE_DOMAIN'] = 'log'\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "from gtn_applications_amd import engine as E\n"
        "rs = np.random.RandomState(7)\n"
        "x = torch.tensor(rs.randn(3, 300, 9) * 1.5, dtype=torch.float32, device='cuda')\n"
        "targets = [rs.randint(0, 8, size=n).tolist() for n in (300, 0, 17)]\n"
        "tg = E.targets_on_device(targets, x.device)\n"
        "st = E.lattice_forward(x, E.PackedLattice.ctc(tg.flat, tg.offsets, 8, 9, x.device))\n"
        "assert E.lattice_formats(st).tolist() == [0, 0, 0]\n"
        "print(' '.join('%.6f' % v for v in st.logz.cpu().tolist()))\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    logd = [float(v) for v in out.stdout.split()[-3:]]
    assert logd[0] == -np.inf
    np.testing.assert_allclose(logd[1:], got[1:], rtol=2e-6)


def _random_acceptor(rs, Q, C, n_eps, hubs):
    """(Graph, src, dst, lab) of an acceptor the lean sweeps do not take: a self-loop and one to three arcs from earlier
    states into every state, `hubs` states that collect (and send) 100 arcs, `n_eps` epsilon arcs p -> q with p < q
    (acyclic: chains of them give several closure levels, one state collects twelve)."""
    from gtn_applications_amd import graph as G

    src, dst, lab = [], [], []
    for q in range(Q):
        src.append(q), dst.append(q), lab.append(int(rs.randint(C)))
        for _ in range(int(rs.randint(1, 4))):
            if q:
                src.append(int(rs.randint(max(0, q - 40), q))), dst.append(q), lab.append(int(rs.randint(C)))
    for h in hubs:
        for p_ in rs.randint(0, h, size=100).tolist():
            src.append(p_), dst.append(h), lab.append(int(rs.randint(C)))
        for q_ in rs.randint(h + 1, Q, size=100).tolist():
            src.append(h), dst.append(q_), lab.append(int(rs.randint(C)))
    eps_into = min(Q - 1, 60)
    for p_ in sorted(set(rs.randint(0, eps_into, size=12).tolist())):  # (one state closed by a row of lanes)
        src.append(p_), dst.append(eps_into), lab.append(-1)
    for _ in range(n_eps):
        p_ = int(rs.randint(0, Q - 1))
        src.append(p_), dst.append(int(rs.randint(p_ + 1, min(Q, p_ + 30)))), lab.append(-1)
    g = G.Graph(True)
    for q in range(Q):
        g.add_node(q % 50 == 0, q % 7 == 3)
    for s_, d_, l_ in zip(src, dst, lab):
        g.add_arc(s_, d_, G.epsilon if l_ < 0 else l_)
    return g, src, dst, lab


@pytest.mark.parametrize("Q,T", [(1100, 10), (40, 70), (300, 33)])
def test_general_probability_sweep_on_random_acceptors(crit, Q, T):
    """csrc/lattice_kernels.hip run_chain_prob_general: acceptors outside the lean sweeps' shape -- states of 100 arcs (a
    row of 16 lanes each, more of them than the workgroup has rows, more than 64 arcs a row), epsilon arcs over several
    levels (one state with twelve), more states than the workgroup has threads (1100), learnable weights on every arc
    -- are swept in the fp64 probability domain too: formats say so, and log Z, the emission gradient and every arc's
    weight gradient (epsilon arcs' included) meet the float64 epsilon-aware recurrence."""
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(Q + T)
    C, B = 12, 2
    graphs, arcs = [], []
    for b in range(B):
        g, src, dst, lab = _random_acceptor(rs, Q - 3 * b, C, n_eps=25, hubs=[Q // 3, Q // 2] + list(range(Q // 2 + 1, Q // 2 + 9)))
        graphs.append(g)
        arcs.append((src, dst, lab))
    n = [len(a[0]) for a in arcs]
    W = (0.4 * rs.randn(sum(n))).astype(np.float32)
    wids = [np.arange(n[0]), n[0] + np.arange(n[1])]
    x = rs.randn(B, T, C).astype(np.float32)
    xd, Wd = dev(x), dev(W)
    pack = E.PackedLattice.from_graphs(graphs, C, xd.device, wids=wids)
    st = E.lattice_forward(xd, pack, weights=Wd, need_beta=True)
    assert E.lattice_formats(st).tolist() == [1] * B
    coef = torch.ones(B, device="cuda")
    dx, dW = torch.full_like(xd, float("nan")), torch.zeros_like(Wd)
    E.lattice_grad(st, coef, coef_w=coef, dx=dx, dW=dW)
    got = st.logz.cpu().numpy()
    for b in range(B):
        src, dst, lab = arcs[b]
        Qb = Q - 3 * b
        start = [q for q in range(Qb) if q % 50 == 0]
        accept = [q for q in range(Qb) if q % 7 == 3]
        score, gx, garc = OR.lattice_forward_backward_eps(x[b].astype(np.float64), src, dst, lab,
                                                          W[wids[b]].astype(np.float64), start, accept, Qb)
        assert np.isfinite(score)
        assert got[b] == pytest.approx(score, rel=RTOL, abs=1e-5)
        close(dx[b], gx, msg=f"utterance {b}")
        close(dW[wids[b][0]:wids[b][-1] + 1], garc, atol=5e-5, msg=f"arc weights of utterance {b}")


@pytest.mark.parametrize("T", [1, 2, 15, 16, 17, 33, 130])
def test_banded_sweep_awkward_sizes(crit, T):
    """The ASG force-alignment lattices run the register-resident banded sweep (chain wave + loader wave, chunks of
    16 frames, four chunks of lookahead): frame counts around the chunk size, targets from empty to 63 labels (64
    states: the whole wave), label sets that are not a multiple of four (the compact rows are padded), repeated
    labels (two arcs of one slot) -- numerator scores and gradients against the oracle, every utterance."""
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(T)
    C = 11
    lens = [0, 1, 2, 3, 5, 9, 16, 31, 63]
    B = len(lens)
    targets = [rs.randint(0, C, size=n).tolist() for n in lens]
    x = rs.randn(B, T, C).astype(np.float32)
    W = (0.3 * rs.randn(C + 1, C)).astype(np.float32)
    xd, Wd = dev(x), dev(W)
    tg = E.targets_on_device(targets, xd.device)
    pack = E.PackedLattice.asg_force_align(tg.flat, tg.offsets, C, xd.device)
    assert pack.desc.max_labels % 4 == 0 and pack.desc.max_states == 64
    st = E.lattice_forward(xd, pack, weights=Wd)
    fmt = E.lattice_formats(st).tolist()  # 1: probability domain (the banded path); utterances without any accepting
    assert all(f == 1 for f, n in zip(fmt, lens) if 1 <= n <= T), fmt  # path may be handed to the log-domain repair
    coef = torch.ones(B, device="cuda")
    dx, dW = torch.full_like(xd, float("nan")), torch.zeros_like(Wd)
    E.lattice_grad(st, coef, coef_w=coef, dx=dx, dW=dW)
    got = st.logz.cpu().numpy()
    wflat = W.astype(np.float64).reshape(-1)
    dW_want = np.zeros(W.size)
    for b, y in enumerate(targets):
        L = len(y)
        src, dst, lab, wid = [], [], [], []  # asg.py:72-81 with the weight indices of asg.py:54-69
        for l in range(1, L + 1):
            c = y[l - 1]
            src += [l - 1, l]
            dst += [l, l]
            lab += [c, c]
            wid += [c if l == 1 else (1 + c) * C + y[l - 2], (1 + c) * C + c]
        score, gx, garc = OR.lattice_forward_backward(x[b].astype(np.float64), src, dst, lab,
                                                      wflat[wid] if wid else np.zeros(0), [0], [L], L + 1)
        if L == 0 or L > T or not np.isfinite(score):  # (asg.py:75-77: no accepting node for an empty target)
            assert got[b] == -np.inf and float(dx[b].abs().max()) == 0.0, (b, L, got[b])
            continue
        assert got[b] == pytest.approx(score, rel=RTOL, abs=1e-5)
        close(dx[b], gx, msg=f"utterance {b} (L={L})")
        np.add.at(dW_want, np.asarray(wid, dtype=np.int64), garc)
    close(dW, dW_want.reshape(W.shape), atol=5e-5)


@pytest.mark.parametrize("T,accumulate", [(5, False), (47, True), (200, False)])
def test_banded_gradient_kernel_matches_the_general_kernel(crit, T, accumulate, monkeypatch):
    """band_grad_kernel (one wave per 16 frames, lane = state) against grad_kernel (LDS tiles, arc lists) on the same
    sweeps: per-utterance factors, an upstream scalar, accumulation into an existing gradient, repeated labels,
    targets from one label to the whole wave, utterances without an accepting path.  WFL_LATTICE_BAND_GRAD=0 sends
    every utterance to the general kernel."""
    from gtn_applications_amd import engine as E

    rs = np.random.RandomState(100 + T)
    C = 13
    lens = [1, 2, 4, 4, 7, 20, 40, 63, T + 1]
    targets = [rs.randint(0, C, size=n).tolist() for n in lens]
    targets[3] = [5, 5, 5, 5]
    B = len(lens)
    xd, Wd = dev(rs.randn(B, T, C).astype(np.float32)), dev((0.5 * rs.randn(C + 1, C)).astype(np.float32))
    tg = E.targets_on_device(targets, xd.device)
    pack = E.PackedLattice.asg_force_align(tg.flat, tg.offsets, C, xd.device)
    st = E.lattice_forward(xd, pack, weights=Wd)
    coef = dev((rs.rand(B) + 0.5).astype(np.float32))
    gout = dev(np.array([0.37], dtype=np.float32))
    seed = dev(rs.randn(B, T, C).astype(np.float32))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("WFL_LATTICE_BAND_GRAD", mode)
        dx, dW = seed.clone(), torch.zeros_like(Wd)
        E.lattice_grad(st, coef, coef_w=coef, gout=gout, dx=dx, accumulate=accumulate, dW=dW)
        out[mode] = (dx.cpu().numpy(), dW.cpu().numpy())
    monkeypatch.delenv("WFL_LATTICE_BAND_GRAD")
    base = seed.cpu().numpy() if accumulate else 0.0
    assert np.abs(out["0"][0] - base).max() > 0.05  # (the gradient is not trivially zero)
    close(out["1"][0], out["0"][0], atol=2e-6)
    close(out["1"][1], out["0"][1], atol=2e-5)
    assert float(np.abs(out["1"][0][-1] - (base[-1] if accumulate else 0.0)).max()) == 0.0  # T + 1 labels: no path


def test_lattice_certificate_sends_what_a_double_cannot_hold_to_the_log_domain(crit):
    """Monotone chains whose alpha mass sits 2^2900 above the states that carry the posteriors (scores that reward the
    late states early and the early states late): the forward sweep's double underflows there, the two sweeps disagree
    about Z, the repair launch re-runs the utterance in the log domain -- next to a harmless utterance that stays in
    the probability domain.  Both against the float64 oracle."""
    from gtn_applications_amd import engine as E

    T, L, C, c = 400, 20, 24, 10.0
    x = np.zeros((2, T, C), dtype=np.float32)
    y = list(range(L))
    x[0, : T // 2, L // 2:L] = c  # early frames reward the late labels ...
    x[0, T // 2:, : L // 2] = c   # ... late frames the early labels: no monotone path collects either
    rs = np.random.RandomState(0)
    x[1] = rs.randn(T, C)
    targets = [y, y]
    xt = dev(x)
    tg = E.targets_on_device(targets, xt.device)
    W = torch.zeros(C + 1, C, device="cuda")
    pack = E.PackedLattice.asg_force_align(tg.flat, tg.offsets, C, xt.device)
    st = E.lattice_forward(xt, pack, weights=W)
    assert E.lattice_formats(st).tolist() == [0, 1]  # utterance 0 was repaired in the log domain
    dx = torch.zeros_like(xt)
    E.lattice_grad(st, torch.ones(2, device="cuda"), dx=dx)
    for b in range(2):
        src, dst, lab = [], [], []
        for l in range(1, L + 1):
            src += [l - 1, l]
            dst += [l, l]
            lab += [y[l - 1], y[l - 1]]
        lz, gx, _ = OR.lattice_forward_backward(x[b], src, dst, lab, np.zeros(2 * L), [0], [L], L + 1)
        assert float(st.logz[b]) == pytest.approx(lz, rel=1e-5)
        close(dx[b], gx, rtol=2e-3 if b == 0 else 1e-4, atol=2e-3 if b == 0 else 1e-5, msg=f"utterance {b}")


def _stream_that_runs_beside_the_current_one():
    """HIP streams share a few hardware queues (round-robin over the streams a process has created): a new stream may sit
    on the CURRENT stream's queue, and work queued on it then runs in front of the current stream's instead of beside it
    (seen when this file runs inside the whole suite: the 'competing' work simply delayed the steps).  Try streams until
    a kernel of the current stream finishes while a long one on the candidate is still running."""
    probe = torch.zeros((), device="cuda")
    for _ in range(8):
        cand = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(cand):
            torch.cuda._sleep(20_000_000)  # (10 ms at the shader clock, 0.2 s if the counter is the 100 MHz one)
        probe += 1
        torch.cuda.current_stream().synchronize()
        beside = not cand.query()
        torch.cuda.synchronize()
        if beside:
            return cand
    pytest.skip("no stream that runs beside the current one (hardware queues shared)")


def test_ctc_pipelined_step_keeps_its_forward_progress_under_cu_contention():
    """The gradient workgroups of the pipelined launch wait on flags raised by the chain workgroups of the SAME launch:
    safe only while the chains get CUs.  In data-parallel training another stream (RCCL all-reduce, the model's
    backward GEMMs) competes for them -- exactly the 8-GPU situation this box cannot show.  Here a second stream keeps
    every CU busy with large GEMMs and device-wide copies while 30 steps run: no wave may give up waiting (status
    word), and every step must reproduce the uncontended result bit for bit (same kernels, same order of operations)."""
    from gtn_applications_amd import engine as E

    B, T, C, L = 128, 1000, 100, 44
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, C, generator=g).cuda()
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    tg = E.targets_on_device(targets, x.device)
    scale, _, coef = E.loss_factors(tg, "none")
    ref_dx = torch.empty_like(x)
    ws, ref_nll, ref_loss = E.ctc_forward_backward(x, tg, C - 1, coef, None, ref_dx, loss_scale=scale, want_loss=True)
    torch.cuda.synchronize()
    assert not E.ctc_pipeline_gave_up(ws, B, T, tg.max_len)
    ref_nll, ref_loss = ref_nll.clone(), ref_loss.clone()
    # (GEMMs of ~1.5 ms: a step's workgroups get onto the chip where the competing stream's kernels hand over; with
    # 8192^3 GEMMs of 7 ms the steps were seen to take seconds each on a bad day -- most of the step's workgroups
    # resident and spinning for the last few, the GEMM crawling on the CUs they left -- and the suite minutes)
    a = torch.randn(4096, 4096, device="cuda")
    big = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
    stop = torch.zeros((), device="cuda")
    beside = 0
    for attempt in range(4):
        # (HIP streams share a few hardware queues and the mapping is the runtime's: when the competing stream lands on
        # this stream's queue its work runs IN FRONT of the steps, not beside them -- seen inside the whole suite -- and
        # the attempt says nothing; another stream then)
        side = _stream_that_runs_beside_the_current_one()
        with torch.cuda.stream(side):  # ~0.4 s of all-CU work queued ahead: GEMMs (compute) and fills / copies (bandwidth)
            for _ in range(240):
                c = a @ a
                big.copy_(big.roll(1)[: big.numel()])
                stop += c[0, 0] * 0
        beside = 0
        for step in range(30):
            dx = torch.full_like(x, float("nan"))
            ws, nll, loss = E.ctc_forward_backward(x, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
            torch.cuda.current_stream().synchronize()
            beside += not side.query()  # the step ended while the competing work was still running
            assert not E.ctc_pipeline_gave_up(ws, B, T, tg.max_len), f"step {step}: a gradient wave gave up waiting"
            same = (torch.equal(nll, ref_nll), torch.equal(loss, ref_loss), torch.equal(dx, ref_dx))
            assert all(same), (f"step {step}: nll / loss / dx equal: {same}; max |dx - ref| "
                               f"{(dx - ref_dx).abs().nan_to_num(float('inf')).max().item():.3g}, utterances that differ "
                               f"{(dx != ref_dx).flatten(1).any(1).nonzero().flatten().tolist()[:8]}")
        torch.cuda.synchronize()
        if beside >= 10:
            break
    assert beside >= 10, f"only {beside} of 30 steps ran beside the competing stream in 4 attempts: contention not exercised"


def test_ctc_loss_backward_without_the_engine_equals_the_engine(monkeypatch):
    """`CTCLoss(x, targets, blank).backward()` on leaf emissions takes csrc/torch_ops.cpp's ctc_fast_backward (the gradient
    of the forward launch handed to x.grad, no autograd engine); everything else takes the engine.  Same numbers bit
    for bit, same .grad semantics: first gradient assigned, later ones added; a second backward raises; retain_graph,
    a gradient argument, hooks on the emissions and non-leaf emissions all go through the engine and agree."""
    from gtn_applications_amd.criterions import ctc

    g = torch.Generator().manual_seed(11)
    B, T, C, L = 16, 120, 40, 9
    x = torch.randn(B, T, C, generator=g)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()

    def grad(fast, how="plain"):
        monkeypatch.setattr(ctc, "_FAST_BACKWARD", fast)
        xg = x.cuda().requires_grad_(True)
        if how == "nonleaf":
            loss = ctc.CTCLoss(xg * 1.0, targets, C - 1)
        else:
            loss = ctc.CTCLoss(xg, targets, C - 1)
        assert type(loss) is ctc._EagerLoss and loss.dim() == 0
        if how == "hook":
            seen = []
            xg.register_hook(lambda gr: seen.append(1))
        if how == "retain":
            loss.backward(retain_graph=True)
        elif how == "gradient":
            loss.backward(torch.ones_like(loss))
        else:
            loss.backward()
        if how == "hook":
            assert seen == [1]
        return xg, loss

    want, _ = grad(False)
    for how in ("plain", "retain", "gradient", "hook", "nonleaf"):
        got, loss = grad(True, how)
        assert torch.equal(got.grad, want.grad), how
        if how == "plain":
            with pytest.raises(RuntimeError, match="second time"):
                loss.backward()
            # accumulation: a second step on the same leaf adds
            ctc.CTCLoss(got, targets, C - 1).backward()
            torch.testing.assert_close(got.grad, 2 * want.grad)
            # the loss is an ordinary tensor in expressions: the engine handles (2 * loss).backward()
            got.grad = None
            (2.0 * ctc.CTCLoss(got, targets, C - 1)).backward()
            torch.testing.assert_close(got.grad, 2 * want.grad)


def test_ctc_targets_recognised_by_object_identity_track_in_place_edits():
    """csrc/torch_ops.cpp LastBatch: a target list handed over again (ctc_benchmark.py:26-31 reuses one list) is recognised
    by the identity of its int objects instead of being flattened and hashed again.  Ints are immutable, so the only way
    to change such a batch is to put OTHER objects into the lists -- which the comparison sees: after every in-place
    edit (a label replaced, a row replaced by an equal / a different one, a row shortened, the outer list edited) the
    loss must equal what a fresh copy of the edited batch gives."""
    from gtn_applications_amd.criterions import ctc

    g = torch.Generator().manual_seed(13)
    B, T, C, L = 6, 80, 300, 9  # (labels beyond CPython's cached small ints too)
    x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
    tg = [[int(v) for v in row] for row in torch.randint(C - 1, (B, L), generator=g).tolist()]

    def loss_of(t):
        return ctc.CTCLoss(x, t, C - 1, "none").item()

    def fresh():  # the same labels as tuples of new int objects: nothing for the identity check to recognise
        return loss_of(tuple(tuple(int(str(v)) for v in row) for row in tg))

    for _ in range(3):  # first call stages, second hits the content cache and remembers the objects, third is recognised
        assert loss_of(tg) == fresh()
    tg[2][4] = (tg[2][4] + 7) % (C - 1)
    assert loss_of(tg) == fresh()
    assert loss_of(tg) == fresh()
    tg[1] = list(tg[1])  # an equal row, another list object
    assert loss_of(tg) == fresh()
    tg[3] = [5, 299 - 1, 17]
    assert loss_of(tg) == fresh()
    assert loss_of(tg) == fresh()
    tg[0].pop()
    assert loss_of(tg) == fresh()
    tg[5], tg[4] = tg[4], tg[5]
    assert loss_of(tg) == fresh()
    assert loss_of(tg) == fresh()


def test_ctc_loss_backward_started_at_non_leaf_emissions_equals_the_engine(monkeypatch):
    """Emissions that are a producer's output (train.py:262-266; ctc_benchmark.py:22): `loss.backward()` starts the
    autograd engine AT the emissions' edge with the forward launch's gradient (csrc/torch_ops.cpp ctc_fast_backward),
    skipping the criterion's own node.  Everything under the emissions must behave as under torch.Tensor.backward: the
    producer's backward, tensor hooks and retain_grad on the emissions, accumulation into the leaf from two losses; a hook or retain_grad on the LOSS, and a torch other than the one the extension was built for, fall
    back to the engine proper."""
    from gtn_applications_amd.criterions import ctc

    g = torch.Generator().manual_seed(12)
    B, T, C, L = 8, 90, 30, 7
    x = torch.randn(B, T, C, generator=g)
    w = torch.rand(C, generator=g) + 0.5
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    taken = []
    real = ctc._native_node().ctc_fast_backward

    class Spy:  # (records whether the short cut handled the call)
        def __getattr__(self, name):
            return getattr(ctc._NODE_REAL, name)

        def ctc_fast_backward(self, loss):
            ok = real(loss)
            taken.append(ok)
            return ok

    monkeypatch.setattr(ctc, "_NODE_REAL", ctc._native_node(), raising=False)
    monkeypatch.setattr(ctc, "_NODE", Spy())

    def run(fast, how):
        monkeypatch.setattr(ctc, "_FAST_BACKWARD", fast)
        del taken[:]
        xg = x.cuda().requires_grad_(True)
        wd = w.cuda()
        seen = {}
        if how == "view":
            em = xg.view_as(xg)
        else:
            em = xg * wd  # a producer with a backward of its own
        if how == "hook":
            em.register_hook(lambda gr: seen.setdefault("hook", gr.clone()))
        if how == "retain_grad":
            em.retain_grad()
        loss = ctc.CTCLoss(em, targets, C - 1)
        assert type(loss) is ctc._EagerLoss
        if how == "loss_hook":
            loss.register_hook(lambda gr: seen.setdefault("loss_hook", gr.clone()))
        if how == "loss_retain_grad":
            loss.retain_grad()
        if how == "two_losses":  # (each with a producer of its own: their gradients add up in the leaf)
            other = ctc.CTCLoss(xg * (2.0 * wd), targets[::-1], C - 1, "mean")
            loss.backward()
            other.backward()
        else:
            loss.backward()
        if how == "retain_grad":
            seen["em_grad"] = em.grad.clone()
        if how == "loss_retain_grad":
            seen["loss_grad"] = loss.grad.clone()
        return xg.grad.clone(), seen, list(taken)

    for how, shortcut in (("mul", True), ("view", True), ("hook", True), ("retain_grad", True), ("two_losses", True),
                          ("loss_hook", False), ("loss_retain_grad", False)):
        want, want_seen, _ = run(False, how)
        got, got_seen, took = run(True, how)
        assert torch.equal(got, want), how
        assert sorted(got_seen) == sorted(want_seen), how
        for k in want_seen:
            assert torch.equal(got_seen[k], want_seen[k]), (how, k)
        assert took and all(t is shortcut for t in took), (how, took)
    assert "hook" in run(True, "hook")[1] and "em_grad" in run(True, "retain_grad")[1]
    # a second backward raises torch's error on both routes
    monkeypatch.setattr(ctc, "_FAST_BACKWARD", True)
    xg = x.cuda().requires_grad_(True)
    loss = ctc.CTCLoss(xg * 1.0, targets, C - 1)
    loss.backward()
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()
    # another torch than the extension's: the short cut is refused, the engine gives the same gradient
    want = run(True, "mul")[0]
    monkeypatch.setattr(ctc, "_FAST_BACKWARD_OK", None)
    monkeypatch.setattr(ctc, "torch_release", lambda v: "0.0.0")
    assert not ctc.fast_backward_enabled()
    got, _, took = run(True, "mul")
    assert took == [] and torch.equal(got, want)
    monkeypatch.setattr(ctc, "_FAST_BACKWARD_OK", None)


def test_ctc_step_picks_the_log_domain_launch_while_the_certificate_keeps_rejecting():
    """Scores without structure and a spread of 3 nats: the lane-exponent step's certificate rejects every utterance
    (fast launch + log-domain repair launch).  The repair launch leaves its count in a pinned host word of the
    workspace; the following calls on that workspace go straight to the log-domain step (no repairs reported, ~2/3
    of the time), every 16th one tries the lane-exponent step again, and benign data brings the step back to it.
    Every call -- whichever launch served it -- is within the parity bar of the float64 oracle."""
    from gtn_applications_amd import engine as E

    B, T, C, L = 16, 200, 40, 9
    g = torch.Generator().manual_seed(21)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    wild = 3.0 * torch.randn(B, T, C, generator=g)
    calm = torch.randn(B, T, C, generator=g)
    want = {id(t): OR.ctc_loss_grad_batched(t.numpy(), targets, C - 1) for t in (wild, calm)}

    def step(x):
        xd = x.cuda()
        tg = E.targets_on_device(targets, xd.device)
        scale, _, coef = E.loss_factors(tg, "none")
        dx = torch.full_like(xd, float("nan"))
        ws, nll = E.ctc_forward_backward(xd, tg, C - 1, coef, None, dx)
        torch.cuda.synchronize()
        np.testing.assert_allclose(nll.cpu().numpy(), want[id(x)][0], rtol=1e-4)
        np.testing.assert_allclose(dx.cpu().numpy(), want[id(x)][1], rtol=1e-4, atol=1e-4 / B)
        return E.ctc_pipeline_repaired(ws, B, T, tg.max_len)

    first = step(wild)
    assert first * 8 > B, "unstructured scores with a spread of 3 nats are expected to fail the certificate"
    later = [step(wild) for _ in range(20)]
    assert later.count(0) >= 17 and max(later) * 8 > B, later  # log-domain launches, with a lane-exponent probe in between
    back = [step(calm) for _ in range(20)]
    assert back[-1] == 0 and max(back) == 0, back
    # ... and the step is on the lane-exponent launch again: wild data is caught by the certificate at once
    assert step(wild) * 8 > B
