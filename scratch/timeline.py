import ctypes, sys, numpy as np, torch, subprocess, os
os.system("cp /root/repo/scratch/libwfl_TL.so /root/repo/gtn_applications_amd/libwfl.so")
sys.path.insert(0, "/root/repo"); sys.argv = ["bench.py", "--workload", sys.argv[1], "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
import runpy
try:
    runpy.run_path("/root/repo/bench.py", run_name="__main__")
except SystemExit:
    pass
from gtn_applications_amd import _native as N
n = 3 * 4096
buf = (ctypes.c_ulonglong * n)()
print("rc", N.lib.wfl_debug_timeline(buf, n))
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 3)
a = a[a[:, 1] > 0]
t0, t1, hw = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2]
base = t0.min()
print("WGs", len(a), "span(ticks)", t1.max() - base, "mean life", (t1 - t0).mean(), "max life", (t1 - t0).max())
# concurrency: sample
ev = sorted([(t, 1) for t in t0] + [(t, -1) for t in t1])
c = m = 0
for _, d in ev:
    c += d; m = max(m, c)
print("max concurrent WGs", m)
cu = (hw & 0xffffffff).astype(np.int64); xcc = (hw >> 32).astype(np.int64)
cuid = ((cu >> 8) & 0xf) | (((cu >> 12) & 0x3) << 4) | (((cu >> 13) & 0x7) << 6)
print("distinct (xcc, hwid cu/sh/se):", len(set(zip(xcc.tolist(), ((cu >> 8) & 0xfff).tolist()))))
starts = np.sort(t0 - base); print("start times percentiles", np.percentile(starts, [0, 25, 50, 75, 100]))
