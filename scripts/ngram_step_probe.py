"""Where the time of the Transducer step WITH a transition model goes (the reference's n-gram scenario,
benchmarks/transducer_benchmark.py: N = 81, T = 250, L = 44): wall time per step with the host running ahead, with a
synchronise per step, and the host's own time per step (no synchronise inside the timed loop, stream drained first)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gtn_applications_amd.criterions import transducer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, T, L = 81, 250, 44
torch.manual_seed(0)
tokens = [(i,) for i in range(N)]
g2i = {i: i for i in range(N)}
x = torch.randn(B, T, N + 1).cuda().requires_grad_(True)
targets = [t.squeeze() for t in torch.randint(N, size=(B, L)).split(1)]
crit = transducer.Transducer(tokens, g2i, ngram=n, reduction="mean", blank="optional", allow_repeats=False).cuda()


def step():
    x.grad = None
    for p in crit.parameters():
        p.grad = None
    crit(x, targets).backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
K = 100
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B} ngram={n}: host {(t1 - t0) / K * 1e3:.3f} ms per step, with the drain {(t2 - t0) / K * 1e3:.3f} ms")
t0 = time.perf_counter()
for _ in range(K):
    step()
    torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"   synchronised every step: {(t1 - t0) / K * 1e3:.3f} ms")
if len(sys.argv) > 3:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)
