"""The product kernels under world_size 2 (train.py:137-142,201-208, utils.py:62-74).

The pool's boxes have one GPU, so both ranks use cuda:0 and the process group is gloo (RCCL refuses two ranks on one
device); everything else is the multi-GPU path as it stands: every rank deals itself a contiguous shard of ONE global
batch (parallel.shard_batch), runs the HIP criterion on it -- ASG with learned transitions, a Transducer with a learned
bigram, CTC -- and the transition-weight gradients meet in ONE all-reduce (parallel.sync_transition_grads, or
DistributedDataParallel's reducer when the criterion is wrapped as train.py:205-208 wraps it).  Checked on every rank:
the averaged transition gradient and the global mean loss equal the single-process HIP result on the whole batch AND the
float64 oracle; the rank's emission gradient equals its rows of the single-process gradient (x B_global / B_local).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

TOKENS = ["a", "b", "ab", "ba", "aba"]
G2I = {"a": 0, "b": 1}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _asg_batch():
    rs = np.random.RandomState(5)
    B, T, C = 8, 120, 12
    x = rs.randn(B, T, C).astype(np.float32)
    W = (0.3 * rs.randn(C + 1, C)).astype(np.float32)
    targets = [rs.randint(0, C, size=n).tolist() for n in (5, 9, 1, 7, 12, 3, 8, 6)]
    return x, W, targets


def _ctc_batch():
    rs = np.random.RandomState(6)
    B, T, C = 8, 150, 20
    x = rs.randn(B, T, C).astype(np.float32)
    targets = [rs.randint(0, C - 1, size=n).tolist() for n in (11, 4, 0, 9, 17, 2, 6, 13)]
    return x, targets, C - 1


def _transducer_batch():
    rs = np.random.RandomState(7)
    B, T, C = 6, 40, len(TOKENS) + 1
    x = rs.randn(B, T, C).astype(np.float32)
    targets = [[0, 1, 0], [1, 0], [0, 0, 1, 0], [1], [1, 1, 0], [0, 1]]
    params = None
    return x, targets, params


def _single_process(which):
    """The whole global batch on one process: the HIP result the shards must reproduce."""
    from gtn_applications_amd.criterions import asg, ctc, transducer

    if which == "asg":
        x, W, targets = _asg_batch()
        xd = torch.tensor(x, device="cuda", requires_grad=True)
        Wd = torch.tensor(W, device="cuda", requires_grad=True)
        loss = asg.ASGLoss(xd, Wd, targets, "mean")
        loss.backward()
        return loss.item(), xd.grad.cpu().numpy(), Wd.grad.cpu().numpy()
    if which == "ctc":
        x, targets, blank = _ctc_batch()
        xd = torch.tensor(x, device="cuda", requires_grad=True)
        loss = ctc.CTCLoss(xd, targets, blank, "mean")
        loss.backward()
        return loss.item(), xd.grad.cpu().numpy(), None
    x, targets, _ = _transducer_batch()
    crit = _make_transducer()
    xd = torch.tensor(x, device="cuda", requires_grad=True)
    loss = crit(xd, [torch.tensor(t) for t in targets])
    loss.backward()
    return loss.item(), xd.grad.cpu().numpy(), crit.transition_params.grad.cpu().numpy()


def _make_transducer():
    from gtn_applications_amd.criterions import transducer

    crit = transducer.Transducer(TOKENS, G2I, ngram=2, blank="optional", allow_repeats=False, reduction="mean").cuda()
    rs = np.random.RandomState(8)
    with torch.no_grad():
        crit.transition_params.copy_(torch.tensor(0.2 * rs.randn(crit.transition_params.numel()).astype(np.float32)))
    return crit


def _worker(rank, world, port, out_dir, want):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    torch.cuda.set_device(0)  # (both ranks: the box has one GPU)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP

        from gtn_applications_amd import _native
        from gtn_applications_amd import parallel as P
        from gtn_applications_amd.criterions import asg, ctc
        from oracle import recurrences as OR

        assert _native.lib.wfl_version() >= 1  # the HIP library, not a stand-in

        # ---- ASG: functional form + sync_transition_grads (what a hand-written loop does) ----
        x, W, targets = _asg_batch()
        B = x.shape[0]
        xs, tg = P.shard_batch(torch.tensor(x), targets)
        lo, hi = P.shard_bounds(B, rank, world)
        assert xs.shape[0] == hi - lo == B // world and tg == targets[lo:hi]
        xd = xs.cuda().requires_grad_(True)

        class Crit(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.transitions = torch.nn.Parameter(torch.tensor(W, device="cuda"))

            def forward(self, inputs, tgs):
                return asg.ASGLoss(inputs, self.transitions, tgs, "mean")

        crit = Crit()
        loss = crit(xd, tg)
        loss.backward()
        P.sync_transition_grads(crit)
        gl = float(P.global_mean_loss(loss.detach(), hi - lo))
        want_loss, want_dx, want_dw = want["asg"]
        orc_loss, orc_dx, orc_dw = OR.asg_loss_grad(x.astype(np.float64), W.astype(np.float64), targets, "mean")
        got_dw = crit.transitions.grad.cpu().numpy()
        np.testing.assert_allclose(got_dw, want_dw, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(got_dw, orc_dw, rtol=1e-4, atol=2e-5)
        assert abs(gl - want_loss) <= 2e-6 * abs(want_loss) and abs(gl - orc_loss) <= 1e-4 * abs(orc_loss)
        got_dx = xd.grad.cpu().numpy() * (hi - lo) / B
        np.testing.assert_allclose(got_dx, want_dx[lo:hi], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(got_dx, orc_dx[lo:hi], rtol=1e-4, atol=1e-5)

        # ---- the same criterion wrapped in DistributedDataParallel (train.py:205-208): the reducer's all-reduce ----
        ddp = DDP(Crit(), device_ids=[0])
        xd2 = xs.cuda().requires_grad_(True)
        ddp(xd2, tg).backward()
        np.testing.assert_allclose(ddp.module.transitions.grad.cpu().numpy(), want_dw, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(xd2.grad.cpu().numpy() * (hi - lo) / B, want_dx[lo:hi], rtol=2e-5, atol=1e-7)

        # ---- Transducer with a learned bigram under DDP ----
        xt, tt, _ = _transducer_batch()
        Bt = xt.shape[0]
        xts, tts = P.shard_batch(torch.tensor(xt), tt)
        lo_t, hi_t = P.shard_bounds(Bt, rank, world)
        tr = DDP(_make_transducer(), device_ids=[0])
        xtd = xts.cuda().requires_grad_(True)
        tl = tr(xtd, [torch.tensor(t) for t in tts])
        tl.backward()
        wl, wdx, wdp = want["transducer"]
        np.testing.assert_allclose(tr.module.transition_params.grad.cpu().numpy(), wdp, rtol=5e-5, atol=2e-6)
        np.testing.assert_allclose(xtd.grad.cpu().numpy() * (hi_t - lo_t) / Bt, wdx[lo_t:hi_t], rtol=5e-5, atol=1e-7)
        assert abs(float(P.global_mean_loss(tl.detach(), hi_t - lo_t)) - wl) <= 5e-6 * abs(wl)

        # ---- CTC (cfg5's shape of step: utterance shards + the forced exchange of a [(C+1), C] buffer) ----
        xc, tc, blank = _ctc_batch()
        Bc, C = xc.shape[0], xc.shape[2]
        xcs, tcs = P.shard_batch(torch.tensor(xc), tc)
        lo_c, hi_c = P.shard_bounds(Bc, rank, world)
        xcd = xcs.cuda().requires_grad_(True)
        cl = ctc.CTCLoss(xcd, tcs, blank, "mean")
        cl.backward()
        exchange = torch.full((C + 1, C), float(rank + 1), device="cuda")
        P.all_reduce_mean_([exchange])
        assert torch.equal(exchange.cpu(), torch.full((C + 1, C), 1.5))
        wl, wdx, _ = want["ctc"]
        orc_l, orc_d = OR.ctc_loss_grad(xc.astype(np.float64), tc, blank, "mean")
        assert abs(float(P.global_mean_loss(cl.detach(), hi_c - lo_c)) - wl) <= 2e-6 * abs(wl)
        got = xcd.grad.cpu().numpy() * (hi_c - lo_c) / Bc
        np.testing.assert_allclose(got, wdx[lo_c:hi_c], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(got, orc_d[lo_c:hi_c], rtol=1e-4, atol=1e-5)
        torch.cuda.synchronize()
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_the_hip_kernels_equal_the_single_process_result(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp

    want = {k: _single_process(k) for k in ("asg", "ctc", "transducer")}
    torch.cuda.synchronize()
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), want), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
