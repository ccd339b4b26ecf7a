"""How much do the Transducer's sweeps and its gradient kernel disturb each other when they run side by side?
(cfg4 shapes; the gradient of the PREVIOUS forward pass on a second stream next to this pass's gather + sweeps)"""
import sys, time, random
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from gtn_applications_amd import engine as E
from gtn_applications_amd.criterions import transducer as TR

B, T, Lp = 64, 800, 15
tokens, g2i = bench.word_pieces()
C = len(tokens) + 1
rnd = random.Random(0)
x = torch.randn(B, T, C, generator=torch.Generator().manual_seed(0)).cuda()
tg = [torch.tensor([g2i[ch] for _ in range(Lp) for ch in rnd.choice(tokens)]) for _ in range(B)]
crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
crit.tokens.arc_sort(True)
dev = x.device
flat, offsets, lens = E.flatten_any(tg)
sc = np.array([1.0 / n for n in lens], dtype=np.float32)
pack = E.PackedLattice.transducer_batch(crit.tokens, crit.lexicon, None, flat, offsets, C, dev, extra=np.concatenate([sc, sc / B, -sc / B]))
cneg = pack.extra[2 * B:]
dx = torch.empty_like(x)
side = torch.cuda.Stream()

def fwd():
    return E.lattice_forward(x, pack, need_beta=True, log_softmax=True)

def grad(st):
    E.lattice_grad(st, cneg, coef_w=cneg, gout=None, dx=dx, accumulate=False)

prev = fwd(); torch.cuda.synchronize()
def serial():
    st = fwd(); grad(prev); return st
def overlapped():
    st = fwd2(lambda: grad(prev))
    torch.cuda.current_stream().wait_stream(side)
    return st

import ctypes
from gtn_applications_amd import _native as N
from gtn_applications_amd.engine import ptr, stream_ptr, LatticeState
_F32 = torch.float32
tiny = torch.zeros(64, device=dev)
def fwd2(beside):
    """lattice_forward with `beside()` launched on the side stream once the gather is done (next to the sweeps)"""
    d = pack.desc
    n_xg, n_ab = ctypes.c_int64(), ctypes.c_int64()
    N.check(N.lib.wfl_lattice_workspace(pack._desc_ref, T, ctypes.byref(n_xg), ctypes.byref(n_ab)))
    st = LatticeState()
    st.pack, st.T, st.C, st.weights = pack, T, C, None
    st.xg = torch.empty(max(n_xg.value, 1), dtype=_F32, device=dev)
    st.alpha = torch.empty(max(n_ab.value, 1), dtype=_F32, device=dev)
    st.beta = torch.empty(max(n_ab.value, 1), dtype=_F32, device=dev)
    st.bptr = None
    st.logz = torch.empty(B, dtype=_F32, device=dev)
    st.x = x
    st.row_lse = torch.empty((B, T), dtype=_F32, device=dev)
    s = stream_ptr()
    N.check(N.lib.wfl_lattice_gather(pack._desc_ref, ptr(pack.ints), ptr(x), T, C, ptr(st.xg), ptr(st.row_lse), s))
    ev = torch.cuda.Event(); ev.record()
    N.check(N.lib.wfl_lattice_forward(pack._desc_ref, ptr(pack.ints), ptr(pack.floats), ptr(st.xg), T, None, N.SEMIRING_LOG,
                                      ptr(st.alpha), ptr(st.beta), None, ptr(st.logz), s))
    with torch.cuda.stream(side):
        side.wait_event(ev)
        for _ in range(DELAY): tiny.add_(1.0)
        beside()
    return st
DELAY = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def only_fwd():
    return fwd()
def only_grad():
    grad(prev); return prev
import os
for name, fn in [c for c in (("fwd", only_fwd), ("grad", only_grad), ("serial", serial), ("overlapped", overlapped)) if os.environ.get("ONLY", c[0]) == c[0]]:
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f"{name:12s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us")
