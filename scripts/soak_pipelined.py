"""Soak test of the pipelined CTC step: N launches on the same inputs must reproduce the first result
bit for bit (a stale cross-XCD read of a checkpoint or flag would show up as a difference).
usage (GPU box): python scripts/soak_pipelined.py [N [check_every]]"""
import sys

import torch

sys.path.insert(0, "/root/repo")
from gtn_applications_amd import engine as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
check_every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator().manual_seed(0)
B, T, C, L = 128, 1000, 100, 44
x = torch.randn(B, T, C, generator=g).cuda()
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
tg = E.targets_on_device(targets, x.device)
scale, _, coef = E.loss_factors(tg, "none")
gout = torch.ones(1, device="cuda")
ref_dx = torch.empty_like(x)
_, ref_nll, ref_loss = E.ctc_forward_backward(x, tg, C - 1, coef, gout, ref_dx, loss_scale=scale, want_loss=True)
ref_dx, ref_nll, ref_loss = ref_dx.clone(), ref_nll.clone(), ref_loss.clone()
bad = 0
dx = torch.empty_like(x)
for i in range(n):
    dx.fill_(float("nan"))
    _, nll, loss = E.ctc_forward_backward(x, tg, C - 1, coef, gout, dx, loss_scale=scale, want_loss=True)
    if i % check_every == 0 or i == n - 1:
        ok = torch.equal(dx, ref_dx) and torch.equal(nll, ref_nll) and torch.equal(loss, ref_loss)
        bad += not ok
torch.cuda.synchronize()
print(f"{n} launches, {bad} mismatching checks")
sys.exit(1 if bad else 0)
