"""scratch: per-wave cycle counters of the meet-in-the-middle CTC launch (libwfl built with -DWFL_MITM_STATS=1, selected
through WFL_LIB_PATH).  d[wave] = total, wait0, wait1, wait2, polls, hw_id, end wall clock."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import engine as E
B, T, C, L = int(os.environ.get("B", 128)), int(os.environ.get("T", 1000)), 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda()
targets = torch.randint(C - 2, (B, L), generator=g).tolist()
tg = E.targets_on_device(targets, x.device)
scale, _, coef = E.loss_factors(tg, "mean")
dx = torch.empty_like(x)
for _ in range(5):
    ws, nll, loss = E.ctc_forward_backward(x, tg, C - 1, coef, None, dx, loss_scale=scale, want_loss=True)
torch.cuda.synchronize()
raw = E.ctc_workspace_field(ws, B, T, tg.max_len, 4).view(torch.int64).cpu().numpy()
d = raw[:B * 2 * 16 * 8].reshape(B, 2, 16, 8).astype(np.float64)
blkraw = raw[B * 2 * 16 * 8:].reshape(B, 2, 256)
blk = (blkraw & ((1 << 48) - 1)).astype(np.float64)
blkpoll = (blkraw >> 48) & 0xfff
names = {0: "chain", 1: "stager0", 2: "stager1", 3: "stager2", 5: "stager3", 4: "flusher", 8: "fetcher",
         6: "emit0", 7: "emit1", 9: "emit2", 10: "emit3", 11: "emit4", 13: "emit5", 14: "emit6", 15: "emit7", 12: "emit8"}
what = {"chain": ("first block", "staged polls", "offdone"), "stager": ("slot wait", "stage compute", "-"),
        "flusher": ("ckready wait", "-", "-"), "fetcher": ("pck slot wait", "partner flag wait", "-"),
        "emit": ("ckdone wait", "pready wait", "compute+store")}
cyc = 1.0 / 2400.0  # us per cycle (nominal)
for dirn in (0, 1):
    print("dir %d  (medians over %d utterances, us at 2.4 GHz; polls = count)" % (dirn, B))
    for wv in sorted(names):
        r = d[:, dirn, wv, :]
        nm = names[wv]
        k = "emit" if nm.startswith("emit") else "stager" if nm.startswith("stager") else nm
        simd = int(np.median((r[:, 5].astype(np.int64) >> 4) & 3))
        print("  %-8s simd %d total %6.1f | %s %6.1f | %s %6.1f | %s %6.1f | polls %7.0f" % (
            nm, simd, np.median(r[:, 0]) * cyc, what[k][0], np.median(r[:, 1]) * cyc, what[k][1], np.median(r[:, 2]) * cyc,
            what[k][2], np.median(r[:, 3]) * cyc, np.median(r[:, 4])) + (
            " | of compute+store: forward %5.1f, rows->HBM %5.1f" % (np.median(r[:, 4]) * cyc, np.median(r[:, 7]) * cyc) if k == "emit" else ""))
ex = blkraw[:, :, 128:128 + 72].reshape(B, 2, 9, 8).astype(np.float64)
nblk = (NB_ := (T + 15) // 16) / 2 / 9.0
print("emitters, cumulative cycles from block entry (median over utterances and emitters, per block = total / %.2f blocks):" % nblk)
for i, nm in ((0, "forward done"), (2, "K done"), (3, "backward done"), (4, "fold + tile writes done"), (5, "certificate done"), (1, "(rows -> HBM part)")):
    print("   %-26s %7.0f cycles/block" % (nm, np.median(ex[:, :, :, i]) / nblk))
ends = d[:, :, :, 6]
starts = d[:, :, 0, 7]  # chain wave's entry clock per workgroup
t0 = starts.min()
wg_end = ends.max(axis=2)
print("workgroup entry (us after the first): median %.1f p90 %.1f max %.1f" % tuple(np.percentile((starts - t0) / 100, [50, 90, 100])))
print("chain wave end: median %.1f p90 %.1f max %.1f" % tuple(np.percentile((ends[:, :, 0] - t0) / 100, [50, 90, 100])))
print("workgroup end:  median %.1f p90 %.1f max %.1f" % tuple(np.percentile((wg_end - t0) / 100, [50, 90, 100])))
print("workgroup span (entry -> last wave end): median %.1f p90 %.1f max %.1f" % tuple(np.percentile((wg_end - starts) / 100, [50, 90, 100])))
pro = (d[:, :, 0, 6] - d[:, :, 0, 7]) / 100 - d[:, :, 0, 0] / 2400
print("chain wave: entry -> counters start (prologue) us: alpha median %.1f, beta median %.1f" % (np.median(pro[:, 0]), np.median(pro[:, 1])))
hw = d[0, 0, :, 5].astype(np.int64)
print("utterance 0 alpha workgroup: wave -> simd", [int((h >> 4) & 3) for h in hw], "cu", [int((h >> 8) & 15) for h in hw], "se", [int((h >> 13) & 7) for h in hw])
NB = (T + 15) // 16
bt = np.median(blk[:, :, :NB], axis=0) / 2400.0  # us since the chain wave's start
for dirn in (0, 1):
    print("dir %d chain block start times (us): " % dirn + " ".join("%d:%.1f" % (k, bt[dirn, k]) for k in range(0, NB, 4)))
    dts = np.diff(bt[dirn, :NB])
    h = NB // 2
    print("   pace us/block: first half median %.3f, second half median %.3f" % (np.median(dts[2:h - 2]), np.median(dts[h + 2:])))
    print("   per block us: " + " ".join("%.2f" % v for v in dts))
    print("   polls/block (mean over utterances): " + " ".join("%.0f" % v for v in blkpoll[:, dirn, :NB].mean(axis=0)))

# stragglers: which workgroups end last?
hwc = d[:, :, 0, 5].astype(np.int64)
xcc = (hwc >> 32) & 15; cu = (hwc >> 8) & 15; se = (hwc >> 13) & 7
order = np.argsort(wg_end.reshape(-1))[::-1]
print("slowest workgroups: (b, dir) xcc se cu | chain end, wg end (us after first entry) | chain staged-poll us")
for o in order[:12]:
    bb, dd = divmod(o, 2)
    print("  (%3d,%d) xcc %d se %d cu %2d | %.1f %.1f | %.1f" % (bb, dd, xcc[bb, dd], se[bb, dd], cu[bb, dd], (ends[bb, dd, 0] - t0) / 100, (wg_end[bb, dd] - t0) / 100, d[bb, dd, 0, 2] / 2400))
print("fastest:")
for o in order[-5:]:
    bb, dd = divmod(o, 2)
    print("  (%3d,%d) xcc %d se %d cu %2d | %.1f %.1f | %.1f" % (bb, dd, xcc[bb, dd], se[bb, dd], cu[bb, dd], (ends[bb, dd, 0] - t0) / 100, (wg_end[bb, dd] - t0) / 100, d[bb, dd, 0, 2] / 2400))
for x in range(8):
    m = xcc == x
    if m.any(): print("  xcc %d: %3d workgroups, end median %.1f max %.1f" % (x, m.sum(), np.median((wg_end[m] - t0) / 100), ((wg_end[m] - t0) / 100).max()))

# three-stamp pace (round 5: the chain wave stamps blocks 4, NB/2 and NB-2 only)
st = blk[:, :, :3]
print("THREE-STAMP pace, cycles/block (clock64 units x 1): first half median %.0f, second half median %.0f  | us: first %.2f second %.2f (at 2.4 GHz nominal)" % (
    np.median((st[:, :, 1] - st[:, :, 0]) / (NB // 2 - 4)), np.median((st[:, :, 2] - st[:, :, 1]) / (NB - 2 - NB // 2)),
    np.median((st[:, :, 1] - st[:, :, 0])) / 2400.0, np.median((st[:, :, 2] - st[:, :, 1])) / 2400.0))
