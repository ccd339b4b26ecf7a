"""CPU tests (-m "not gpu") of the N>1 path on the gloo backend, world_size 2: shard + one
all-reduce reproduces the single-process result.  The per-shard gradients come from the oracle
(stand-in for the kernels, which need a GPU); what is under test is the sharding / scaling /
collective logic of gtn_applications_amd/parallel.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gtn_applications_amd import parallel as P
        from oracle import recurrences as OR

        rs = np.random.RandomState(0)  # same global batch on every rank
        B, T, C = 6, 12, 5
        x = rs.randn(B, T, C)
        W = 0.3 * rs.randn(C + 1, C)
        targets = [rs.randint(0, C, size=rs.randint(1, 5)).tolist() for _ in range(B)]
        xs, tg = P.shard_batch(torch.tensor(x), targets)
        lo, hi = P.shard_bounds(B, rank, world)
        assert xs.shape[0] == hi - lo == B // world
        loss, dx, dW = OR.asg_loss_grad(xs.numpy(), W, tg, "mean")

        class Crit(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.transitions = torch.nn.Parameter(torch.tensor(W))

        crit = Crit()
        crit.transitions.grad = torch.tensor(dW)
        P.sync_transition_grads(crit)
        gl = P.global_mean_loss(torch.tensor(loss), hi - lo)
        want_loss, want_dx, want_dW = OR.asg_loss_grad(x, W, targets, "mean")
        np.testing.assert_allclose(crit.transitions.grad.numpy(), want_dW, rtol=1e-6, atol=1e-9)
        assert abs(float(gl) - want_loss) < 1e-6
        # emission gradients never cross ranks: rank-local rows equal the global ones up to the
        # 1/B_local vs 1/B_global normalisation
        np.testing.assert_allclose(dx * (hi - lo) / B, want_dx[lo:hi], rtol=1e-6, atol=1e-9)
        # several tensors, one collective
        a, b = torch.full((3,), float(rank)), torch.full((2, 2), float(rank + 1))
        P.all_reduce_mean_([a, None, b])
        assert torch.allclose(a, torch.full((3,), 0.5)) and torch.allclose(b, torch.full((2, 2), 1.5))
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    from gtn_applications_amd import parallel as P

    for n in (0, 1, 7, 128, 1000):
        for world in (1, 2, 3, 8):
            spans = [P.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_single_process_helpers_are_noops():
    from gtn_applications_amd import parallel as P

    t = torch.ones(3)
    assert P.all_reduce_mean_([t])[0] is t and torch.equal(t, torch.ones(3))
    assert float(P.global_mean_loss(torch.tensor(2.5), 4)) == 2.5


def test_bench_starts_its_own_ranks_and_reports_the_slowest():
    """`python bench.py --gpus 2` with no launcher (train.py:344-347 spawns its own ranks too): two gloo ranks on the
    CPU with a stub step (--stub-cpu) -- the JSON line says n_gpus == 2, the collective saw two ranks, and the
    reported time is the MAX over ranks (rank 1's stub step is slower than rank 0's)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-cpu", "--steps", "5",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["collective_world_size"] == 2 and rec["steps"] == 5
    assert rec["ms_per_step"] >= rec["rank0_ms_per_step"] and rec["ms_per_step"] >= 2.0  # rank 1 sleeps 2 ms per step
    # a launcher that started a different number of ranks than --gpus says is an error, not a silent 1-rank run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--stub-cpu"],
                         env=dict(env, RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr
