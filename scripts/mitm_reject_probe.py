"""scratch: which certificate rejects utterances of the meet-in-the-middle CTC step on model-shaped scores
(per-utterance log2 Z of the chain after repair against the range the first emitted blocks reproduced)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import engine as E, _native as N
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_gpu_configs import _model_shaped_scores

boost, noise, wrong = [float(v) for v in (sys.argv[1:4] or (12.0, 2.0, 0.1))]
B, T, C, L = 128, 1000, 100, 44
rs = np.random.RandomState(int(boost * 10 + wrong * 100))
lp, targets = _model_shaped_scores(rs, B, T, C, L, boost, noise, wrong)
xd = lp.cuda()
tg = E.targets_on_device(targets, xd.device)
scale, _, coef = E.loss_factors(tg, "none")
dx = torch.full_like(xd, float("nan"))
ws, nll = E.ctc_forward_backward(xd, tg, C - 1, coef, None, dx)
torch.cuda.synchronize()
rep = E.ctc_pipeline_repaired(ws, B, T, tg.max_len)
z2 = E.ctc_workspace_field(ws, B, T, tg.max_len, N.CTC_WS_LOG2Z).view(torch.float64).cpu().numpy()
zr = E.ctc_workspace_field(ws, B, T, tg.max_len, N.CTC_WS_ZRANGE).view(torch.int64).cpu().numpy().reshape(B, 2)
print("repaired", rep)
zq = np.rint(z2 * 65536).astype(np.int64)
for b in range(B):
    lo, hi = zr[b]
    if lo < zq[b] - 10 or hi > zq[b] + 10:
        print("b %3d  log2Z %.4f  block range [%s, %s] (diff %s, %s)" % (
            b, z2[b], "DEAD" if lo < -(1 << 61) else "%.4f" % (lo / 65536), "%.4f" % (hi / 65536),
            "-" if lo < -(1 << 61) else "%.5f" % ((lo - zq[b]) / 65536), "%.5f" % ((hi - zq[b]) / 65536)))
