"""scratch: does data that looks like a trained model's output stay on the lane-exponent path?
logits = noise + boost * onehot(label of a random monotone alignment of the target), then log_softmax."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import engine as E
rs = np.random.RandomState(0)
B, T, C, L = 128, 1000, 100, 44
for boost, noise, wrong in [(8.0, 1.0, 0.0), (8.0, 1.0, 0.1), (12.0, 2.0, 0.1), (5.0, 1.0, 0.2), (15.0, 3.0, 0.05), (3.0, 1.0, 0.3), (20.0, 1.0, 0.02)]:
    x = (noise * rs.randn(B, T, C)).astype(np.float32)
    targets = []
    for b in range(B):
        y = rs.randint(0, C - 1, size=L)
        targets.append(y.tolist())
        # random monotone alignment: each label gets >= 1 frame, blanks in between
        cuts = np.sort(rs.choice(np.arange(1, T), size=2 * L, replace=False))
        lab = np.full(T, C - 1)
        for i in range(L):
            lab[cuts[2 * i]:cuts[2 * i + 1]] = y[i]
        flip = rs.rand(T) < wrong  # frames where the model is confidently wrong
        lab = np.where(flip, rs.randint(0, C, size=T), lab)
        x[b, np.arange(T), lab] += boost
    xt = torch.log_softmax(torch.tensor(x), 2).cuda()
    tg = E.CtcTargets(targets, xt.device)
    scale, _, coef = E.loss_factors(tg, "mean")
    dx = torch.empty_like(xt)
    ws, nll, loss = E.ctc_forward_backward(xt, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
    torch.cuda.synchronize()
    print(f"boost {boost} noise {noise} wrong-frame rate {wrong}: loss {float(loss):.3f}, repaired {E.ctc_pipeline_repaired(ws, B, T, tg.max_len)} / {B}")
