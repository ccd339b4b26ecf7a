"""scratch: timeline of the fast pipelined CTC launch (needs libwfl built with -DWFL_DBG_FAST=512 copied over libwfl.so)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import engine as E
B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda()
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
tg = E.targets_on_device(targets, x.device)
scale, _, coef = E.loss_factors(tg, "mean")
dx = torch.empty_like(x)
for _ in range(3):
    ws, nll, loss = E.ctc_forward_backward(x, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
torch.cuda.synchronize()
P, nb = tg.max_len + 1, (T + 15) // 16
o = B * 2 * nb * P * 2; o = (o + 1) & ~1
o += 2 * B * 2 * nb + 2 * B + B + 2 * B; o = (o + 1) & ~1
o += 2 * B * 2 * nb + 2 * B + 2 + 2 * B + 4 * B; o = (o + 1) & ~1
d = ws[o:o + 2 * 4 * (B * nb + 2 * B)].view(torch.int64).cpu().numpy().reshape(-1, 4).astype(np.float64) / 100.0  # us (100 MHz)
items, chains = d[:B * nb], d[B * nb:]
t0 = min(items[:, 0].min(), chains[:, 0].min())
items -= t0; chains -= t0
print("chain waves: start %.1f..%.1f us, end %.1f..%.1f us" % (chains[:, 0].min(), chains[:, 0].max(), chains[:, 1].min(), chains[:, 1].max()))
it = items.reshape(B, nb, 4)
print("items: start min %.1f max %.1f; end min %.1f max %.1f" % (it[:, :, 0].min(), it[:, :, 0].max(), it[:, :, 2].min(), it[:, :, 2].max()))
pre = it[:, :, 1] - it[:, :, 0]; post = it[:, :, 2] - it[:, :, 1]
print("start->flags seen: median %.1f p10 %.1f p90 %.1f us;  flags seen->end: median %.1f p10 %.1f p90 %.1f max %.1f us" % (np.median(pre), np.percentile(pre, 10), np.percentile(pre, 90), np.median(post), np.percentile(post, 10), np.percentile(post, 90), post.max()))
mid = (nb - 1) // 2
for k in (mid, mid + 8, mid + 16, mid + 20, mid + 24, mid + 25, mid + 26, nb - 2, nb - 1, 0):
    print("block %2d: start %.1f flags %.1f end %.1f (medians over utterances)" % (k, np.median(it[:, k, 0]), np.median(it[:, k, 1]), np.median(it[:, k, 2])))
