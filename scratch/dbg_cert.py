import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import engine as E

def cert(ws2, B, T, max_len):
    from gtn_applications_amd import _native as N
    z2 = E.ctc_workspace_field(ws2, B, T, max_len, N.CTC_WS_LOG2Z).view(torch.float64).cpu().numpy()
    zmm = E.ctc_workspace_field(ws2, B, T, max_len, N.CTC_WS_ZRANGE).view(torch.int64).cpu().numpy().reshape(B, 2) / 65536.0
    return z2, zmm

if __name__ == "__main__":
    g = torch.Generator().manual_seed(3)
    for (B, T, C, L, sc) in [(128, 1000, 100, 44, 1.0), (128, 1000, 100, 44, 1.5), (128, 1000, 100, 44, 2.0), (128, 2000, 512, 44, 1.0), (128, 640, 64, 30, 2.3), (128, 640, 64, 30, 3.0)]:
        if 8 * 17 * C * 4 > 160 * 1024:
            continue
        x = torch.randn(B, T, C, generator=g).cuda() * sc
        for lsm in (False, True):
            xx = torch.log_softmax(x, 2) if lsm else x
            targets = torch.randint(C - 2, (B, L), generator=g).tolist()
            tg = E.targets_on_device(targets, x.device)
            scale, _, coef = E.loss_factors(tg, "mean")
            dx = torch.empty_like(x)
            ws2, nll2, loss = E.ctc_forward_backward(xx, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
            torch.cuda.synchronize()
            z2, zmm = cert(ws2, B, T, tg.max_len)
            dev = np.maximum(np.abs(zmm[:, 0] - z2), np.abs(zmm[:, 1] - z2))
            print(f"T={T} C={C} scale={sc} logsoftmax={lsm}: repaired {E.ctc_pipeline_repaired(ws2, B, T, tg.max_len)}; |zk - z2| median {np.median(dev):.2e} p90 {np.percentile(dev, 90):.2e} max {dev.max():.2e}")
