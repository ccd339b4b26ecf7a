"""scratch: probe the criteria over unusual shapes for errors / mismatches against the oracle."""
import sys, os, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd.criterions import asg, ctc, stc
from oracle import recurrences as OR
rs = np.random.RandomState(0)
def report(name, fn):
    try:
        print(name, fn())
    except Exception as e:
        print(name, "ERROR", repr(e)[:200])
for (B, T, C, L) in [(2, 300, 100, 64), (2, 300, 700, 100), (1, 600, 30, 255), (2, 40, 603, 10), (1, 1, 5, 0), (1, 1, 5, 1), (3, 20, 17000, 4), (2, 700, 603, 130)]:
    def run():
        x = rs.randn(B, T, C).astype(np.float32)
        targets = [rs.randint(0, C - 1, size=rs.randint(max(L - 3, 0), L + 1)).tolist() for _ in range(B)]
        xt = torch.tensor(x, device="cuda", requires_grad=True)
        loss = ctc.CTCLoss(xt, targets, C - 1, "mean"); loss.backward()
        wl, wdx = OR.ctc_loss_grad(x, targets, C - 1, "mean")
        return ("loss ok" if (np.isinf(wl) and np.isinf(loss.item())) or abs(loss.item() - wl) <= 1e-4 * abs(wl) else f"LOSS {loss.item()} vs {wl}",
                "grad err %.2e" % float(np.abs(xt.grad.cpu().numpy() - np.nan_to_num(wdx)).max()))
    report(f"CTC B={B} T={T} C={C} L~{L}:", run)
for (B, T, C, L) in [(2, 60, 130, 7), (2, 60, 260, 7), (1, 30, 5, 29), (2, 50, 40, 45)]:
    def run():
        x = rs.randn(B, T, C).astype(np.float32)
        W = (0.3 * rs.randn(C + 1, C)).astype(np.float32)
        targets = [rs.randint(0, C, size=rs.randint(1, L + 1)).tolist() for _ in range(B)]
        xt = torch.tensor(x, device="cuda", requires_grad=True); Wt = torch.tensor(W, device="cuda", requires_grad=True)
        loss = asg.ASGLoss(xt, Wt, targets, "mean"); loss.backward()
        want = OR.asg_loss_grad(x, W, targets, "mean")
        return ("loss ok" if abs(loss.item() - want[0]) <= 1e-4 * abs(want[0]) else f"LOSS {loss.item()} vs {want[0]}",
                "dx err %.2e dW err %.2e" % (float(np.abs(xt.grad.cpu().numpy() - want[1]).max()), float(np.abs(Wt.grad.cpu().numpy() - want[2]).max())))
    report(f"ASG B={B} T={T} C={C} L<={L}:", run)
