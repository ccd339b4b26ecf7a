import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gtn_applications_amd import _native as N
B, L, C = 128, 44, 100
rng = np.random.default_rng(0)
flats = [rng.integers(0, C - 2, B * L).astype(np.int32) for _ in range(50)]
off = (np.arange(B + 1) * L).astype(np.int64)
for name, fn in (("asg_fal", lambda f: N.lib.wfl_lattice_pack_asg_fal(f.ctypes.data, off.ctypes.data, B, C)),
                 ("ctc", lambda f: N.lib.wfl_lattice_pack_ctc(f.ctypes.data, off.ctypes.data, B, C - 1, C))):
    for f in flats[:10]: N.lib.wfl_lattice_host_free(fn(f))
    t0 = time.perf_counter()
    for f in flats[10:]: N.lib.wfl_lattice_host_free(fn(f))
    print(name, "threads", os.environ.get("WFL_HOST_THREADS", "default"), "%.1f us" % ((time.perf_counter() - t0) / 40 * 1e6))
