"""Host and wall microseconds per step of the CTC operator by how loss.backward() reaches the emissions' producer
(cfg2 shape): leaf, non-leaf through the short cut (engine started at the emissions' edge), non-leaf through
torch.Tensor.backward from the loss, and the bare engine floor (y = x.view_as(x); y.backward(g))."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd.criterions import ctc  # noqa: E402

B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
tg = torch.randint(C - 2, (B, L), generator=g).tolist()
dx = torch.randn(B, T, C, generator=g).cuda()


def leaf():
    x.grad = None
    ctc.CTCLoss(x, tg, C - 1).backward()


def view_fast():
    x.grad = None
    ctc.CTCLoss(x.view_as(x), tg, C - 1).backward()


def view_engine():
    x.grad = None
    old, ctc._FAST_BACKWARD = ctc._FAST_BACKWARD, False
    try:
        ctc.CTCLoss(x.view_as(x), tg, C - 1).backward()
    finally:
        ctc._FAST_BACKWARD = old


def fwd_only():
    ctc.CTCLoss(x.view_as(x), tg, C - 1)


def engine_floor():
    x.grad = None
    x.view_as(x).backward(dx)


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return host / n * 1e6, wall / n * 1e6


for name, fn in (("leaf emissions", leaf), ("non-leaf, engine started at the emissions", view_fast),
                 ("non-leaf, torch.Tensor.backward from the loss", view_engine), ("forward only (non-leaf)", fwd_only),
                 ("engine floor: x.view_as(x).backward(g)", engine_floor)):
    h, w = timeit(fn)
    print(f"{name:50s} host {h:7.1f} us   wall {w:7.1f} us")
print("host cores", os.cpu_count())
