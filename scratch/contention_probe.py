"""How long does a cfg2 CTC step take beside a stream of large GEMMs + copies?  (tests/test_gpu_parity.py::...under_cu_contention)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd import engine as E
B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(5)
x = torch.randn(B, T, C, generator=g).cuda()
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
tg = E.targets_on_device(targets, x.device)
scale, _, coef = E.loss_factors(tg, "none")
dx = torch.empty_like(x)
for _ in range(3): E.ctc_forward_backward(x, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
a = torch.randn(8192, 8192, device="cuda"); big = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
for _ in range(3): c = a @ a; big.copy_(big.roll(1)[: big.numel()])
torch.cuda.synchronize()
for waves in (os.environ.get("WFL_CTC_MITM_WAVES", "default"),):
    side = torch.cuda.Stream()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(40): c = a @ a; big.copy_(big.roll(1)[: big.numel()])
    for step in range(10):
        E.ctc_forward_backward(x, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
        torch.cuda.current_stream().synchronize()
        print(f"waves={waves} step {step} done at {time.perf_counter() - t0:.3f} s, side done {side.query()}")
    torch.cuda.synchronize(); print(f"all done at {time.perf_counter() - t0:.3f} s")
