"""scratch: randomized comparison of the fast pipelined CTC step with the three-launch log-domain step."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import engine as E
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nbad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    B = int(rs.choice([1, 2, 5, 17, 64, 130]))
    T = int(rs.choice([1, 5, 16, 17, 31, 32, 33, 100, 257, 640]))
    C = int(rs.choice([2, 3, 8, 29, 100, 130, 255, 300, 513, 1001]))
    Lmax = int(min(63, rs.choice([0, 1, 3, 20, 44, 63])))
    sc = float(rs.choice([0.3, 1.0, 1.0, 1.7]))
    lsm = bool(rs.randint(2))
    x = torch.tensor(rs.randn(B, T, C).astype(np.float32) * sc).cuda()
    if rs.rand() < 0.3:
        x[rs.randint(B), rs.randint(T), rs.randint(C)] = float("-inf")
    if rs.rand() < 0.2:
        x[rs.randint(B), rs.randint(T), rs.randint(C)] = float("nan")
    targets = [rs.randint(0, max(C - 1, 1), size=rs.randint(0, Lmax + 1)).tolist() for _ in range(B)]
    tg = E.CtcTargets(targets, x.device)
    scale, _, coef = E.loss_factors(tg, "mean")
    lse = E.row_lse(x) if lsm else None
    xin = x
    dx = torch.full_like(x, float("nan"))
    ws, nll, loss = E.ctc_forward_backward(xin, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True, lse=lse)
    # reference: log-domain three-launch step on log_softmax'ed input (chain rule by hand for lsm)
    xl = torch.log_softmax(torch.nan_to_num(x, nan=float("-inf")), 2) if lsm else x
    ws2, nll2 = E.ctc_forward(xl, tg, C - 1)
    dx2 = torch.empty_like(x)
    E.ctc_grad(xl, tg, C - 1, ws2, nll2, coef, None, dx2)
    if lsm:
        dx2 = dx2 - torch.exp(xl) * dx2.sum(2, keepdim=True)
        dx2 = torch.nan_to_num(dx2, nan=0.0)
    torch.cuda.synchronize()
    rep = E.ctc_pipeline_repaired(ws, B, T, tg.max_len)
    fin = torch.isfinite(nll2)
    ok = torch.equal(torch.isfinite(nll), fin) and torch.allclose(nll[fin], nll2[fin], rtol=2e-5, atol=2e-4)
    scale_g = float(coef.abs().max())
    err = float((torch.nan_to_num(dx) - dx2).abs().max()) / max(scale_g, 1e-30)
    okg = bool(torch.isfinite(dx).all()) and err < 2e-4
    if not (ok and okg):
        nbad += 1
    print(f"{'ok ' if ok and okg else 'BAD'} B={B} T={T} C={C} Lmax={Lmax} scale={sc} lsm={lsm} repaired={rep} nll_ok={ok} grad_err/scale={err:.2e}")
print("bad:", nbad)
