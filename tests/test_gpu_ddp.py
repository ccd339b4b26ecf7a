"""The reference's multi-GPU shape as far as ONE GPU allows: train.py:137-142 / 201-208 wraps a
criterion that has parameters in DistributedDataParallel, train.py:116-120 and utils.py:276-283 round-trip its
state_dict.  Here: a process group on RCCL ("nccl") of world size one, DDP around ASG and around a Transducer with a
learned bigram -- forward / backward equal to the unwrapped criterion, the parameter gradient arrives through DDP's
reducer (its all-reduce hangs on the parameter's AccumulateGrad node: engine.EagerLoss must stay out of its way) --
the checkpoint round trip, and parallel.all_reduce_mean_(force=True) issuing the RCCL collective on the [(C+1), C]
buffer of cfg5."""
import io
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world_of_one():
    import torch.distributed as dist

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    created = not dist.is_initialized()
    if created:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
    yield dist
    if created:
        dist.destroy_process_group()


def _asg_case(rs, B=6, T=40, C=9):
    x = torch.tensor(rs.randn(B, T, C).astype(np.float32))
    targets = [torch.tensor(rs.randint(0, C - 2, size=n)) for n in (5, 3, 7, 1, 4, 6)][:B]
    return x, targets


def test_ddp_wrapped_asg_equals_the_unwrapped_criterion_and_round_trips_its_state(nccl_world_of_one):
    from torch.nn.parallel import DistributedDataParallel as DDP

    from gtn_applications_amd.criterions import asg

    rs = np.random.RandomState(0)
    # (ASG's class count: tokens + replabels + garbage -- the emissions have that many columns)
    plain = asg.ASG(6, num_replabels=2, use_garbage=True).cuda()
    with torch.no_grad():
        plain.transitions.copy_(torch.tensor(0.3 * rs.randn(*plain.transitions.shape).astype(np.float32)))
    x, targets = _asg_case(rs, C=plain.N)
    x1 = x.cuda().requires_grad_(True)
    loss1 = plain(x1, targets)
    loss1.backward()
    want_dx, want_dw = x1.grad.clone(), plain.transitions.grad.clone()

    wrapped = asg.ASG(6, num_replabels=2, use_garbage=True).cuda()
    wrapped.load_state_dict(plain.state_dict())
    ddp = DDP(wrapped, device_ids=[0])  # train.py:205-208
    x2 = x.cuda().requires_grad_(True)
    loss2 = ddp(x2, targets)
    loss2.backward()
    assert loss2.item() == pytest.approx(loss1.item(), rel=1e-6)
    torch.testing.assert_close(x2.grad, want_dx, rtol=1e-6, atol=1e-7)
    assert wrapped.transitions.grad is not None, "DDP's reducer never saw the parameter's gradient"
    torch.testing.assert_close(wrapped.transitions.grad, want_dw, rtol=1e-6, atol=1e-7)
    # a second step through the same wrapper (the reducer re-arms itself after every backward)
    x3 = x.cuda().requires_grad_(True)
    ddp.zero_grad()
    ddp(x3, targets).backward()
    torch.testing.assert_close(wrapped.transitions.grad, want_dw, rtol=1e-6, atol=1e-7)

    # checkpoint round trip (train.py:116-120 saves criterion.state_dict(), utils.py:276-283 loads it): through bytes,
    # from the DDP wrapper's module, onto another device-less instance
    buf = io.BytesIO()
    torch.save(ddp.module.state_dict(), buf)
    buf.seek(0)
    fresh = asg.ASG(6, num_replabels=2, use_garbage=True)
    fresh.load_state_dict(torch.load(buf, map_location="cpu"))
    assert list(fresh.state_dict()) == ["transitions"]
    torch.testing.assert_close(fresh.transitions.detach(), plain.transitions.detach().cpu(), rtol=0, atol=0)
    # ... and the reloaded criterion decodes like the original
    out = x.cuda()
    assert [p.tolist() for p in fresh.cuda().viterbi(out)] == [p.tolist() for p in plain.viterbi(out)]


def test_ddp_wrapped_transducer_with_a_learned_bigram(nccl_world_of_one):
    from torch.nn.parallel import DistributedDataParallel as DDP

    from gtn_applications_amd.criterions import transducer

    rs = np.random.RandomState(1)
    tokens = ["a", "b", "ab", "ba", "aba"]
    g2i = {"a": 0, "b": 1}

    def make():
        return transducer.Transducer(tokens, g2i, ngram=2, blank="optional", allow_repeats=False, reduction="mean")

    plain = make().cuda()
    with torch.no_grad():
        plain.transition_params.copy_(torch.tensor(0.2 * rs.randn(plain.transition_params.numel()).astype(np.float32)))
    B, T, C = 4, 30, len(tokens) + 1
    x = torch.tensor(rs.randn(B, T, C).astype(np.float32))
    targets = [torch.tensor(t) for t in ([0, 1, 0], [1, 0], [0, 0, 1, 0], [1])]
    x1 = x.cuda().requires_grad_(True)
    loss1 = plain(x1, targets)
    loss1.backward()
    want_dx, want_dp = x1.grad.clone(), plain.transition_params.grad.clone()

    wrapped = make().cuda()
    wrapped.load_state_dict(plain.state_dict())
    ddp = DDP(wrapped, device_ids=[0])
    x2 = x.cuda().requires_grad_(True)
    loss2 = ddp(x2, targets)
    loss2.backward()
    assert loss2.item() == pytest.approx(loss1.item(), rel=1e-6)
    torch.testing.assert_close(x2.grad, want_dx, rtol=1e-6, atol=1e-7)
    assert wrapped.transition_params.grad is not None
    torch.testing.assert_close(wrapped.transition_params.grad, want_dp, rtol=1e-6, atol=1e-7)

    buf = io.BytesIO()
    torch.save(ddp.module.state_dict(), buf)
    buf.seek(0)
    fresh = make()
    fresh.load_state_dict(torch.load(buf, map_location="cpu"))
    torch.testing.assert_close(fresh.transition_params.detach(), plain.transition_params.detach().cpu(), rtol=0, atol=0)
    out = x.cuda()
    assert [p.tolist() for p in fresh.cuda().viterbi(out)] == [p.tolist() for p in plain.viterbi(out)]


def test_forced_all_reduce_issues_the_rccl_collective_on_one_rank(nccl_world_of_one):
    """parallel.all_reduce_mean_ returns at once for a group of one -- unless force=True: then dist.all_reduce (RCCL) runs
    on the flattened payload; at world size one the mean of one rank is the input.  Payload: cfg5's [(C+1), C] buffer."""
    from gtn_applications_amd import parallel

    dist = nccl_world_of_one
    C = 512
    buf = torch.randn(C + 1, C, device="cuda")
    other = torch.randn(7, device="cuda")
    want = (buf.clone(), other.clone())
    calls = []
    real = dist.all_reduce

    def spy(t, *a, **k):
        calls.append((t.numel(), t.device.type))
        return real(t, *a, **k)

    dist.all_reduce = spy
    try:
        parallel.all_reduce_mean_([buf, None, other])  # world 1, not forced: no collective
        assert calls == []
        parallel.all_reduce_mean_([buf, None, other], force=True)
    finally:
        dist.all_reduce = real
    assert calls == [((C + 1) * C + 7, "cuda")], calls
    torch.cuda.synchronize()
    torch.testing.assert_close(buf, want[0], rtol=0, atol=0)
    torch.testing.assert_close(other, want[1], rtol=0, atol=0)


def test_transducer_prepare_packs_the_next_batch_on_a_side_thread():
    """Transducer.prepare(targets) -> PreparedTargets: the batch's alignment acceptors built, packed and uploaded on a
    side thread and stream; forward(inputs, prepared) gives what forward(inputs, targets) gives (the loss bit for bit,
    the gradient to the rounding of its atomic adds), also when the handle was prepared several steps ahead and under a
    transition model; a handle for another batch size is an error like a list of the wrong length."""
    from gtn_applications_amd.criterions import transducer

    rs = np.random.RandomState(2)
    tokens = ["a", "b", "ab", "ba", "aba", "bb"]
    g2i = {"a": 0, "b": 1}
    for kwargs in (dict(blank="optional", allow_repeats=False, reduction="mean"), dict(ngram=2, blank="optional", reduction="none")):
        crit = transducer.Transducer(tokens, g2i, **kwargs).cuda()
        if crit.transition_params is not None:
            with torch.no_grad():
                crit.transition_params.copy_(torch.tensor(0.2 * rs.randn(crit.transition_params.numel()).astype(np.float32)))
        B, T, C = 5, 40, len(tokens) + 1
        batches = [[torch.tensor(rs.randint(0, 2, size=rs.randint(1, 6))) for _ in range(B)] for _ in range(4)]
        x = torch.tensor(rs.randn(B, T, C).astype(np.float32)).cuda()
        want = []
        for tg in batches:
            xr = x.clone().requires_grad_(True)
            loss = crit(xr, tg)
            loss.backward()
            want.append((loss.item(), xr.grad.clone()))
        handles = [crit.prepare(tg) for tg in batches]  # all of them ahead
        assert all(len(h) == B for h in handles)
        for h, (wl, wg) in zip(handles, want):
            xr = x.clone().requires_grad_(True)
            loss = crit(xr, h)
            loss.backward()
            assert loss.item() == wl
            torch.testing.assert_close(xr.grad, wg, rtol=1e-5, atol=1e-8)
        # a handle prepared for a batch of another size than the emissions: an error, as with a plain list
        with pytest.raises(ValueError):
            crit(x[:3], handles[0])
