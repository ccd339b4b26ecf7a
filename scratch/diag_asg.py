"""cfg3 parity diagnosis: which half of the ASG step (dense denominator / force-aligned numerator) loses accuracy
at T=1000, and where along the utterance."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtn_applications_amd import engine as E
from oracle import recurrences as OR

B, T, C, L = 4, 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(128, T, C, generator=g)[:B].contiguous()
W0 = torch.randn(C + 1, C, generator=g)
targets = torch.randint(C - 2, (128, L), generator=g).tolist()[:B]
dev = torch.device("cuda")
out = {}
for wscale in (1.0, 0.0, 0.3):
    W = (W0 * wscale).contiguous()
    xg, Wg = x.to(dev), W.to(dev)
    coef = torch.ones(B, device=dev)
    # denominator alone
    st = E.dense_forward(xg, Wg)
    dx = torch.zeros_like(xg); dW = torch.zeros_like(Wg)
    E.dense_grad(xg, Wg, st, coef, coef_w=coef, gout=None, dx=dx, accumulate=False, dW=dW)
    flagged = E.dense_flagged(st).cpu().numpy().tolist()
    res = dict(flagged=flagged)
    wdW = np.zeros((C + 1, C))
    for b in range(B):
        lz, px, pW = OR.dense_forward_backward(x[b].numpy(), W.numpy())
        wdW += pW
        err = np.abs(dx[b].cpu().numpy() - px)
        res.setdefault("fcc_logz_abs_err", []).append(abs(float(st.logz[b]) - lz))
        res.setdefault("fcc_post_max_err", []).append(float(err.max()))
        res.setdefault("fcc_post_err_by_t", []).append([float(err[k:k + 100].max()) for k in range(0, T, 100)])
        res.setdefault("fcc_rowsum_dev", []).append(float(np.abs(dx[b].sum(1).cpu().numpy() - 1).max()))
    e = np.abs(dW.cpu().numpy() - wdW)
    res["fcc_dW_max_rel"] = float((e / (np.abs(wdW) + 1e-3)).max())
    # numerator alone
    tg = E.targets_on_device(targets, dev)
    pack = E.PackedLattice.asg_force_align(tg.flat, tg.offsets, C, dev)
    fal = E.lattice_forward(xg, pack, weights=Wg)
    dx2 = torch.zeros_like(xg)
    E.lattice_grad(fal, coef, coef_w=coef, gout=None, dx=dx2)
    for b in range(B):
        y = targets[b]
        src, dst, lab, wid = [], [], [], []
        for l in range(1, L + 1):
            c = y[l - 1]
            src.append(l - 1), dst.append(l), lab.append(c); wid.append(c if l == 1 else (1 + c) * C + y[l - 2])
            src.append(l), dst.append(l), lab.append(c); wid.append((1 + c) * C + c)
        lz, gx, _ = OR.lattice_forward_backward(x[b].numpy(), src, dst, lab, W.numpy().reshape(-1)[wid], [0], [L], L + 1)
        err = np.abs(dx2[b].cpu().numpy() - gx)
        res.setdefault("fal_logz_abs_err", []).append(abs(float(fal.logz[b]) - lz))
        res.setdefault("fal_logz", []).append(lz)
        res.setdefault("fal_post_max_err", []).append(float(err.max()))
        res.setdefault("fal_post_err_by_t", []).append([float(err[k:k + 100].max()) for k in range(0, T, 100)])
    out[f"wscale_{wscale}"] = res
print(json.dumps(out, indent=1))
