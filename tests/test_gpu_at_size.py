"""STC and ConvTransduce1D at a size (-m gpu): the two criteria the BASELINE configurations do not name, at the shape
of the reference's benchmarks / configurations instead of toy shapes.

  STC              T=1000, 100 selected classes (200 augmented columns: stc.py:199-220), B=64, L=44, reduction "mean"
                   -- /root/reference/criterions/stc.py:67-129 -- sampled utterances against the float64 lattice recurrence
                   (oracle/recurrences.py) run on the graph oracle/criteria.py::stc_graph builds (stc.py:23-64)
  ConvTransduce1D  configs/iamdb/convtrans.json: kernel_size 7, stride 4, scale "sqrt", normalize "none", 200 word pieces
                   over 78 graphemes + blank, batch 8, 400 frames -- transducer.py:370-556 -- sampled output windows (all 200
                   entries of each) and the input gradient of an objective that weights exactly those windows, against the
                   graph oracle window by window

Same tolerance as tests/test_gpu_configs.py.  Nothing here reads /root/reference."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import criteria as OC  # noqa: E402
from oracle import recurrences as OR  # noqa: E402
from test_gpu_configs import STATS, check  # noqa: E402,F401  (worst cases land in gpurun_out/parity_r05.json)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _graph_arcs(g):
    return g.src, g.dst, g.ilab, g.w, g.start_nodes(), g.accept_nodes(), g.num_nodes()


def test_stc_at_benchmark_length():
    from gtn_applications_amd.criterions import stc

    B, T, Cp, L, prob = 64, 1000, 100, 44, 0.3
    rs = np.random.RandomState(5)
    x = torch.log_softmax(torch.tensor(rs.randn(B, T, 2 * Cp).astype(np.float32)), 2)
    targets = [rs.randint(1, Cp, size=L).tolist() for _ in range(B)]
    xd = x.cuda().requires_grad_(True)
    # reduction "none": no division by the number of frames; the loss is still the batch's MEAN and a posterior enters the
    # gradient with 1 / B (stc.py:116-129)
    loss = stc.STCLoss(xd, targets, prob, "none")
    loss.backward()
    got_dx = xd.grad.cpu().numpy()
    xn = x.numpy()
    nll = np.zeros(B)
    for b in range(B):
        src, dst, lab, w, start, accept, ns = _graph_arcs(OC.stc_graph(targets[b], Cp, prob))
        logz, dxb, _ = OR.lattice_forward_backward(xn[b], src, dst, lab, w, start, accept, ns)
        nll[b] = -logz
        check("stc_T1000_dx", got_dx[b], -dxb / B, 1.0 / B)
    check("stc_T1000_loss", np.array([loss.item()]), np.array([nll.mean()]), 0.0)
    # reduction "mean" divides by the number of FRAMES (stc.py:90-91)
    m = stc.STCLoss(x.cuda(), targets, prob, "mean").item()
    assert m == pytest.approx(float(nll.mean()) / T, rel=1e-4)


def _word_pieces(n):
    with open(os.path.join(ROOT, "benchmarks", "word_pieces_tokens_1000.txt")) as f:
        tokens = sorted(l.strip() for l in f)
    graphemes = sorted(set(c for t in tokens for c in t))
    g2i = {c: i for i, c in enumerate(graphemes)}
    short = sorted((t for t in tokens if len(t) <= 3), key=lambda t: (len(t), t))[:n]
    assert len(short) == n
    return [tuple(g2i[c] for c in t) for t in short], len(graphemes)


def test_conv_transduce_at_the_iamdb_configuration():
    from gtn_applications_amd.criterions import transducer as tr

    lexicon, ngraph = _word_pieces(200)
    blank, ks, stride, B, T = ngraph, 7, 4, 8, 400
    C = ngraph + 1
    layer = tr.ConvTransduce1D(lexicon, ks, stride, blank, scale="sqrt", normalize="none").cuda()
    rs = np.random.RandomState(11)
    x = torch.log_softmax(torch.tensor(rs.randn(B, T, C).astype(np.float32)), 2)
    pad = ks // 2
    Tout = (T + 2 * pad - ks) // stride + 1
    xd = x.cuda().requires_grad_(True)
    out = layer(xd)
    assert tuple(out.shape) == (B, Tout, len(lexicon))
    # an objective that weights a sample of windows (all entries of each), zero elsewhere
    windows = [(0, 0), (0, 1), (0, Tout - 1), (3, 17), (3, 18), (5, 50), (7, Tout // 2), (7, Tout - 1)]
    wts = np.zeros((B, Tout, len(lexicon)), np.float32)
    for b, wi in windows:
        wts[b, wi] = rs.randn(len(lexicon)).astype(np.float32)
    (out * torch.tensor(wts).cuda()).sum().backward()
    got_out, got_dx = out.detach().cpu().numpy(), xd.grad.cpu().numpy()
    xp = np.pad(x.numpy().astype(np.float64), ((0, 0), (pad, pad), (0, 0)))
    want_dx = np.zeros_like(xp)
    sc = math.sqrt(ks)
    for b, wi in windows:
        t = wi * stride
        crop = xp[b:b + 1, t:t + ks]
        o, dxc, _ = OC.conv_transduce_1d_grad(crop, lexicon, blank, ks, stride, wts[b:b + 1, wi:wi + 1].astype(np.float64) / sc)
        check("conv_iamdb_out", got_out[b, wi], o[0, 0] / sc, 1.0)
        want_dx[b, t:t + ks] += dxc[0]
    check("conv_iamdb_dx", got_dx, want_dx[:, pad:pad + T], 1.0)
