"""Dense (ASG denominator) engine alone: forward sweeps and gradient timed separately (torch events), per class count.
Usage: python scripts/dense_split_time.py B,T,C [B,T,C ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd import engine as E  # noqa: E402

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(128, 1000, 200)]
for (B, T, C) in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, C, generator=g).cuda()
    W = torch.randn(C + 1, C, generator=g).cuda()
    coef = torch.ones(B, device="cuda")
    dx, dW = torch.empty_like(x), torch.empty_like(W)

    def fwd():
        return E.dense_forward(x, W, need_beta=True)

    def grad(st):
        E.dense_grad(x, W, st, coef, coef_w=coef, dx=dx, dW=dW)

    st = fwd()
    grad(st)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 3
    tf = tg = 0.0
    for _ in range(n):
        ev[0].record()
        st = fwd()
        ev[1].record()
        grad(st)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tg += ev[1].elapsed_time(ev[2])
    print(f"B={B} T={T} C={C}: dense forward (alpha + beta) {tf / n:.3f} ms   dense grad (dx + dW) {tg / n:.3f} ms")
