"""Where the host time of the operator path goes (run on the GPU box): per-call wall times of the pieces of
`CTCLoss(x, targets, blank).backward()` with fresh targets, GPU work left asynchronous."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtn_applications_amd import engine as E
from gtn_applications_amd.criterions import ctc

B, T, C, L, N = 128, 1000, 100, 44, 200
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
batches = [torch.randint(C - 2, (B, L), generator=g).tolist() for _ in range(N + 20)]
dev = x.device


def timed(fn, n=N, skip=20):
    for i in range(skip):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(skip, skip + n):
        fn(i)
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return host, (time.perf_counter() - t0) / n * 1e6


E._TARGET_CACHE.data.clear()
print("targets_on_device (fresh)      host %.1f us  (with sync %.1f)" % timed(lambda i: E.targets_on_device(batches[i], dev)))
print("targets_on_device (cached)     host %.1f us  (with sync %.1f)" % timed(lambda i: E.targets_on_device(batches[5], dev)))
tg = E.targets_on_device(batches[0], dev)
xd = x.detach()
dx = torch.empty_like(xd)
coef, scale = tg.addr("cneg_none"), tg.addr("scale_none")
print("engine call (shared ws)        host %.1f us  (with sync %.1f)" % timed(
    lambda i: E.ctc_forward_backward(xd, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True, shared_ws=True)))
print("torch.empty_like(x)            host %.1f us  (with sync %.1f)" % timed(lambda i: torch.empty_like(xd)))
E._TARGET_CACHE.data.clear()
losses = [None]


def fwd(i):
    losses[0] = ctc.CTCLoss(x, batches[i], C - 1)


print("CTCLoss forward (fresh)        host %.1f us  (with sync %.1f)" % timed(fwd))


def fwd_bwd(i):
    x.grad = None
    ctc.CTCLoss(x, batches[i], C - 1).backward()


E._TARGET_CACHE.data.clear()
print("CTCLoss fwd+bwd (fresh)        host %.1f us  (with sync %.1f)" % timed(fwd_bwd))
print("CTCLoss fwd+bwd (same targets) host %.1f us  (with sync %.1f)" % timed(lambda i: fwd_bwd(3)))
tens = [[torch.tensor(r) for r in b] for b in batches]
E._TARGET_CACHE.data.clear()


def fwd_bwd_t(i):
    x.grad = None
    ctc.CTCLoss(x, tens[i], C - 1).backward()


print("CTCLoss fwd+bwd (fresh, tensor targets) host %.1f us  (with sync %.1f)" % timed(fwd_bwd_t))

# ---- Transducer (cfg4): where the cold (fresh targets) host time goes
import ctypes
import random
import numpy as np
from gtn_applications_amd import _native as N
from gtn_applications_amd.criterions import transducer as TR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tokens = sorted(l.strip() for l in open(os.path.join(ROOT, "benchmarks", "word_pieces_tokens_1000.txt")))
graphemes = sorted(set(c for t in tokens for c in t))
g2i = {t: i for i, t in enumerate(graphemes)}
crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
crit.tokens.arc_sort(True)
rnd = random.Random(0)
NB = 40
tb = [[torch.tensor([g2i[ch] for _ in range(15) for ch in rnd.choice(tokens)]) for _ in range(64)] for _ in range(NB)]
flats = [E.flatten_any(t) for t in tb]
Cc = len(tokens) + 1


def tt(fn, n=NB - 5, skip=5):
    for i in range(skip):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(skip, skip + n):
        fn(i)
    h = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return h, (time.perf_counter() - t0) / n * 1e6


print("flatten_any (64 tensors)            host %.1f us (sync %.1f)" % tt(lambda i: E.flatten_any(tb[i])))


def native_only(i, nthreads=0):
    flat, off, _ = flats[i]
    h = N.lib.wfl_transducer_pack_batch(crit.tokens._h, crit.lexicon._h, None, flat.ctypes.data, off.ctypes.data, 64, Cc, nthreads)
    N.lib.wfl_lattice_host_free(h)


print("wfl_transducer_pack_batch (pool)    host %.1f us (sync %.1f)" % tt(native_only))
print("wfl_transducer_pack_batch (serial)  host %.1f us (sync %.1f)" % tt(lambda i: native_only(i, 1), n=5, skip=1))
print("PackedLattice.transducer_batch      host %.1f us (sync %.1f)" % tt(
    lambda i: E.PackedLattice.transducer_batch(crit.tokens, crit.lexicon, None, flats[i][0], flats[i][1], Cc, dev)))
xt = torch.randn(64, 800, Cc, device="cuda", requires_grad=True)
TR._PACK_CACHE.data.clear()


def tfwd(i):
    xt.grad = None
    crit(xt, tb[i]).backward()


print("Transducer fwd+bwd (fresh)          host %.1f us (sync %.1f)" % tt(tfwd))
print("Transducer fwd+bwd (same)           host %.1f us (sync %.1f)" % tt(lambda i: tfwd(7)))
# ---- ASG (cfg3)
from gtn_applications_amd.criterions import asg as AS
Wt = torch.zeros(C + 1, C, device="cuda", requires_grad=True)
E._TARGET_CACHE.data.clear()
tgs = [E.targets_on_device(b, dev) for b in batches[:60]]
print("asg_force_align pack+upload         host %.1f us (sync %.1f)" % tt(
    lambda i: E.PackedLattice.asg_force_align(tgs[i].flat, tgs[i].offsets, C, dev)))


def native_fal(i):
    h = N.lib.wfl_lattice_pack_asg_fal(tgs[i].flat.ctypes.data, tgs[i].offsets.ctypes.data, B, C)
    N.lib.wfl_lattice_host_free(h)


print("wfl_lattice_pack_asg_fal            host %.1f us (sync %.1f)" % tt(native_fal))
E._TARGET_CACHE.data.clear()


def afwd(i):
    x.grad = None
    Wt.grad = None
    AS.ASGLoss(x, Wt, batches[i], "mean").backward()


print("ASGLoss fwd+bwd (fresh)             host %.1f us (sync %.1f)" % timed(afwd))
print("ASGLoss fwd+bwd (same)              host %.1f us (sync %.1f)" % timed(lambda i: afwd(3)))


def afwd_only(i):
    AS.ASGLoss(x, Wt, batches[i], "mean")


E._TARGET_CACHE.data.clear()
print("ASGLoss forward (fresh)             host %.1f us (sync %.1f)" % timed(afwd_only))
print("ASGLoss forward (same)              host %.1f us (sync %.1f)" % timed(lambda i: afwd_only(3)))
print("host cores", os.cpu_count())
