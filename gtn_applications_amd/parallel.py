"""Data-parallel plumbing of the loss engine.

Utterances are independent, so the hot path shards over the batch dimension with no data-path
collective.  The single exchange step is the all-reduce of learned transition-weight gradients
(ASG `transitions`, Transducer `transition_params`) -- what DistributedDataParallel does when the
reference wraps a criterion that has parameters (train.py:205-208) -- plus the scalar loss /
metrics (utils.py:107-126).  On ROCm `backend="nccl"` is RCCL over xGMI; the same code runs on
`gloo` for the CPU tests.  Payloads are tiny (40 KB .. 1 MB), i.e. latency-bound: one flat
all-reduce per step, no bucketing.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n utterances for `rank`; sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(inputs, targets, rank=None, world=None):
    """Slice a global batch ([B,T,C] emissions + list of targets) for this rank."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(inputs.shape[0], rank, world)
    return inputs[lo:hi], targets[lo:hi]


def _all_reduce_sum_(flat, group=None):
    """dist.all_reduce(SUM) of one flat tensor.  RCCL ("nccl") reduces device buffers in place over xGMI; a gloo group
    (CPU tests; two ranks sharing one GPU in tests/test_gpu_world2.py) gets device payloads staged through the host."""
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def all_reduce_mean_(tensors, group=None, force=False):
    """In-place average over ranks of a list of tensors with ONE collective (flattened).  A group of one rank has
    nothing to exchange and returns at once -- unless `force`: the collective is then issued all the same (the only
    way a one-GPU box can run the RCCL call of the multi-GPU path: tests, bench.py under WFL_BENCH_FORCE_DIST=1)."""
    tensors = [t for t in tensors if t is not None]
    if not tensors or not dist.is_available() or not dist.is_initialized():
        return tensors
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return tensors
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    _all_reduce_sum_(flat, group)
    flat /= world
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return tensors


def sync_transition_grads(criterion, group=None):
    """Average the gradients of a criterion's learned transition weights across ranks (equal shard
    sizes assumed, as DDP does).  No-op for criteria without parameters (CTC, STC)."""
    grads = [p.grad for p in criterion.parameters() if p.grad is not None]
    all_reduce_mean_(grads, group)
    return grads


def global_mean_loss(local_mean, n_local, group=None):
    """Mean loss over the GLOBAL batch from per-rank means of possibly unequal shards."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_mean
    buf = torch.stack([local_mean.detach().to(torch.float32) * n_local,
                       torch.tensor(float(n_local), device=local_mean.device)])
    _all_reduce_sum_(buf, group)
    return buf[0] / buf[1]
