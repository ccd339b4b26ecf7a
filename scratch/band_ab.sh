python scratch/band_ab.py gpurun_out/band_on.npz > gpurun_out/band_ab.log 2>&1
WFL_LATTICE_BAND_GRAD=0 python scratch/band_ab.py gpurun_out/band_off.npz >> gpurun_out/band_ab.log 2>&1
python - >> gpurun_out/band_ab.log 2>&1 <<'PY'
import numpy as np
a = np.load("gpurun_out/band_on.npz"); b = np.load("gpurun_out/band_off.npz")
for k in ("dx", "dW", "logz"):
    e = np.abs(a[k] - b[k]); print(k, "max abs diff", e.max(), "max ref", np.abs(b[k]).max())
e = np.abs(a["dx"] - b["dx"])
bad = np.argwhere(e > 1e-5 * np.abs(b["dx"]).max())
print("bad elements", len(bad))
for r in bad[:40]: print(r, a["dx"][tuple(r)], b["dx"][tuple(r)])
import collections
print("bad by b", collections.Counter(bad[:, 0].tolist()))
print("bad by t%16", collections.Counter((bad[:, 1] % 16).tolist()))
PY
rm -f gpurun_out/band_on.npz gpurun_out/band_off.npz
