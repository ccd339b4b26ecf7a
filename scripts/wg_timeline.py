"""Per-workgroup timeline of the lattice gradient kernel (diagnostic build only).

Build libwfl.so with -DWFL_DBG_TIMELINE (csrc/lattice_kernels.hip records wall_clock64 at entry / exit
and HW_ID per workgroup), run one bench workload, then print lifetime and concurrency statistics.
This is how the "grid slightly above one round of resident workgroups costs a whole second round"
effect of DESIGN.md section 3.2 was found (1856 workgroups on 1536 slots: 2 x 250 us).

usage (GPU box):  python scripts/wg_timeline.py transducer
"""
import ctypes
import runpy
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
workload = sys.argv[1] if len(sys.argv) > 1 else "transducer"
sys.argv = ["bench.py", "--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
try:
    runpy.run_path("/root/repo/bench.py", run_name="__main__")
except SystemExit:
    pass
from gtn_applications_amd import _native as N  # noqa: E402

n = 3 * 4096
buf = (ctypes.c_ulonglong * n)()
if not hasattr(N.lib, "wfl_debug_timeline"):
    raise SystemExit("libwfl.so was not built with -DWFL_DBG_TIMELINE")
N.lib.wfl_debug_timeline(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 3)
a = a[a[:, 1] > 0]
t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
print("workgroups", len(a), "span (100 MHz ticks)", t1.max() - t0.min(), "mean lifetime", (t1 - t0).mean())
ev = sorted([(t, 1) for t in t0] + [(t, -1) for t in t1])
cur = peak = 0
for _, d in ev:
    cur += d
    peak = max(peak, cur)
print("max concurrent workgroups", peak)
