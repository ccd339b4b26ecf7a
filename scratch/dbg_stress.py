import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import engine as E
from oracle import recurrences as OR

g = torch.Generator().manual_seed(21)
B, T, C, L = 128, 640, 64, 30
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
gout = torch.ones(1, device="cuda")
for it in range(14):
    x = torch.randn(B, T, C, generator=g).cuda() * (1.0 + 0.1 * it)
tg = E.targets_on_device(targets, x.device)
scale, _, coef = E.loss_factors(tg, "mean")
dx_pipe = torch.empty_like(x)
ws2, nll2, loss = E.ctc_forward_backward(x, tg, C - 1, coef, gout, dx_pipe, loss_scale=scale, want_loss=True)
dx_split = torch.empty_like(x)
ws, nll = E.ctc_forward(x, tg, C - 1)
E.ctc_grad(x, tg, C - 1, ws, nll, coef, gout, dx_split)
torch.cuda.synchronize()
print("repaired", E.ctc_pipeline_repaired(ws2, B, T, tg.max_len))
from gtn_applications_amd import _native as N
z2 = E.ctc_workspace_field(ws2, B, T, tg.max_len, N.CTC_WS_LOG2Z).view(torch.float64).cpu().numpy()
zmm = E.ctc_workspace_field(ws2, B, T, tg.max_len, N.CTC_WS_ZRANGE).view(torch.int64).cpu().numpy().reshape(B, 2) / 65536.0
pb = np.zeros((B, 2), int)
for u in range(B):
    if abs(zmm[u, 0] - z2[u]) > 1.4e-3 or abs(zmm[u, 1] - z2[u]) > 1.4e-3 or pb[u].any():
        print("cert", u, z2[u], zmm[u] - z2[u], pb[u])
a, s = dx_pipe.cpu().numpy(), dx_split.cpu().numpy()
bad = np.abs(a - s) > 2e-3 * np.abs(s) + 1e-9
ub = np.unique(np.nonzero(bad)[0])
print("utterances with mismatches", ub, "count", bad.sum())
for u in ub[:3]:
    xs = x[u:u + 1].cpu().numpy().astype(np.float64)
    wl, wdx = OR.ctc_loss_grad(xs, [targets[u]], C - 1, "none")
    wdx = wdx[0] * float(coef[u]) * -1.0 if False else wdx[0]
    # oracle gradient is d(mean loss)/dx for a batch of 1 with reduction none -> scale to coef
    k = float(coef[u]) / -1.0
    ref = wdx * (-k)
    print("u", u, "nll pipe/split/oracle", float(nll2[u]), float(nll[u]), wl)
    for name, arr in (("pipe", a[u]), ("split", s[u])):
        d = np.abs(arr - ref)
        print("  ", name, "max abs err", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "ref there", ref[np.unravel_index(d.argmax(), d.shape)], "scale", np.abs(ref).max())
    idx = np.nonzero(bad[u])
    t0 = idx[0][:8]; c0 = idx[1][:8]
    print("   frames", t0, "cols", c0)
    print("   pipe ", a[u][t0, c0]); print("   split", s[u][t0, c0]); print("   ref  ", ref[t0, c0])
