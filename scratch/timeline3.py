"""scratch: timeline of the fast pipelined CTC launch with persistent gradient waves (libwfl built with -DWFL_DBG_FAST=512,
selected through WFL_LIB_PATH)."""
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
o += 2 * B * 2 * nb + 2 * B + 2 + 2 * B + 64 * B + 4 * B; o = (o + 1) & ~1
d = ws[o:o + 2 * 4 * (B * nb + 2 * B)].view(torch.int64).cpu().numpy().reshape(-1, 4).astype(np.float64) / 100.0  # us (100 MHz)
items, chains = d[:B * nb], d[B * nb:]
t0 = min(items[:, 0].min(), chains[:, 0].min())
items = items - t0; chains = chains - t0
print("chain waves: start %.1f..%.1f us, end %.1f..%.1f us" % (chains[:, 0].min(), chains[:, 0].max(), chains[:, 1].min(), chains[:, 1].max()))
it = items.reshape(B, nb, 4)
print("items: start min %.1f max %.1f; end min %.1f max %.1f" % (it[:, :, 0].min(), it[:, :, 0].max(), it[:, :, 2].min(), it[:, :, 2].max()))
pre = it[:, :, 1] - it[:, :, 0]; post = it[:, :, 2] - it[:, :, 1]
print("start->flags seen: median %.1f p10 %.1f p90 %.1f;  flags seen->end: median %.1f p10 %.1f p90 %.1f max %.1f us" % (np.median(pre), np.percentile(pre, 10), np.percentile(pre, 90), np.median(post), np.percentile(post, 10), np.percentile(post, 90), post.max()))
end = it[:, :, 2].reshape(-1); st = it[:, :, 0].reshape(-1); fl = it[:, :, 1].reshape(-1)
print("bucket(us): items started / flags seen / finished")
for lo in range(0, 80, 4):
    print("%2d-%2d: %5d %5d %5d" % (lo, lo + 4, ((st >= lo) & (st < lo + 4)).sum(), ((fl >= lo) & (fl < lo + 4)).sum(), ((end >= lo) & (end < lo + 4)).sum()))
mid = (nb - 1) // 2
# when COULD an item have started its post-flag part: both checkpoints published = chain progress
for k in (mid, mid + 8, mid + 16, mid + 24, mid + 28, mid + 30, nb - 1, 0, 1, 2):
    print("block %2d: start %.1f flags %.1f end %.1f (medians over utterances; p90 end %.1f)" % (k, np.median(it[:, k, 0]), np.median(it[:, k, 1]), np.median(it[:, k, 2]), np.percentile(it[:, k, 2], 90)))
