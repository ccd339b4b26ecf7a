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
d = E.ctc_workspace_field(ws, B, T, tg.max_len, 4).view(torch.int64).cpu().numpy().reshape(B, 2, 12, 8).astype(np.float64)
names = {0: "chain", 1: "stager0", 2: "stager1", 3: "stager2", 4: "flusher", 5: "stager3", 6: "emit0", 7: "emit1", 8: "fetcher",
         9: "emit2", 10: "emit3", 11: "emit4"}
what = {"chain": ("first block", "staged polls", "offdone"), "stager": ("slot wait", "stage compute", "-"),
        "flusher": ("ckready wait", "-", "-"), "fetcher": ("pck slot wait", "partner flag wait", "-"),
        "emit": ("ckdone wait", "pready wait", "compute+store")}
cyc = 1.0 / 2400.0  # us per cycle (nominal)
for dirn in (0, 1):
    print("dir %d  (medians over %d utterances, us at 2.4 GHz; polls = count)" % (dirn, B))
    for wv in range(12):
        r = d[:, dirn, wv, :]
        nm = names[wv]
        k = "emit" if nm.startswith("emit") else "stager" if nm.startswith("stager") else nm
        simd = int(np.median((r[:, 5].astype(np.int64) >> 4) & 3))
        print("  %-8s simd %d total %6.1f | %s %6.1f | %s %6.1f | %s %6.1f | polls %7.0f" % (
            nm, simd, np.median(r[:, 0]) * cyc, what[k][0], np.median(r[:, 1]) * cyc, what[k][1], np.median(r[:, 2]) * cyc,
            what[k][2], np.median(r[:, 3]) * cyc, np.median(r[:, 4])))
ends = d[:, :, :, 6]
t0 = ends[ends > 0].min()
print("wave end wall clock (us after the first wave to end): chain a %.1f b %.1f, last emitter a %.1f b %.1f" % (
    np.median(ends[:, 0, 0] - t0) / 100, np.median(ends[:, 1, 0] - t0) / 100, np.median(ends[:, 0, 6:].max(axis=1) - t0) / 100,
    np.median(ends[:, 1, 6:].max(axis=1) - t0) / 100))
