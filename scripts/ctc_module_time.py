"""Step time of the CTC MODULE (criterions/ctc.py CTC(blank, use_pt=False): raw scores in, log_softmax fused into the
step) at the cfg2 / cfg5-shard shapes."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from gtn_applications_amd.criterions import ctc
for (B, T, C, L) in ((128, 1000, 100, 44), (128, 2000, 512, 44)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
    targets = [t for t in torch.randint(C - 2, (B, L), generator=g)]
    crit = ctc.CTC(C - 1, False)
    def step():
        x.grad = None
        crit(x, targets).backward()
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for _ in range(n): step()
    torch.cuda.synchronize()
    print(f"CTC module fwd+bwd T={T} C={C} B={B}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms")
