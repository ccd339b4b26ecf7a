"""Timeline of the gradient workgroups beside the Transducer benchmark's sweeps (a -DWFL_LIVE_STATS build through
WFL_LIB_PATH, WFL_TRANSDUCER_NATIVE=0): when the sweeps end, and what the tiles that end after them spent their time on."""
import ctypes, os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from gtn_applications_amd import _native as N
from gtn_applications_amd.criterions import transducer as TR
B, T, Lp = int(sys.argv[2]) if len(sys.argv) > 2 else 64, int(sys.argv[1]) if len(sys.argv) > 1 else 800, 15
tokens, g2i = bench.word_pieces()
C = len(tokens) + 1
rnd = random.Random(0)
x = torch.randn(B, T, C, generator=torch.Generator().manual_seed(0)).cuda().requires_grad_(True)
tg = [torch.tensor([g2i[ch] for _ in range(Lp) for ch in rnd.choice(tokens)]) for _ in range(B)]
crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
lib = ctypes.CDLL(N.LIB_PATH)
tiles = (ctypes.c_ulonglong * (4096 * 5))()
sweeps = (ctypes.c_ulonglong * (512 * 2))()
n = ctypes.c_uint()
marks = (ctypes.c_ulonglong * 16)()
for it in range(5):
    x.grad = None
    lib.wfl_debug_marks(marks)
    crit(x, tg).backward()
    torch.cuda.synchronize()
    lib.wfl_debug_live_timeline(tiles, ctypes.byref(n), sweeps)
logbuf = (ctypes.c_ulonglong * 1024)()
nlog = ctypes.c_uint()
lib.wfl_debug_log(logbuf, ctypes.byref(nlog))
NSTEPS = 60
for it in range(NSTEPS):  # (back to back: the host runs ahead, the GPU in its sustained state)
    x.grad = None
    crit(x, tg).backward()
torch.cuda.synchronize()
lib.wfl_debug_log(logbuf, ctypes.byref(nlog))
lg = np.array(list(logbuf), dtype=np.int64).reshape(512, 2)[: min(nlog.value, 512)]
lg = lg[np.argsort(lg[:, 1])][-22:]
lg = lg[np.flatnonzero(lg[:, 0] == 1)[0]:]
names = {1: "gather_lse begins", 2: "sweeps begin (workgroup 0)", 3: "sweep of workgroup 0 ends", 10: "certificate begins", 11: "log-domain launch begins",
         12: "gradient kernel of the rest begins", 13: "loss reduction begins"}
print(f"launch log of the last steps of {NSTEPS} back to back (us since the first entry shown; workgroup (0,0) of each kernel):")
for k, t in lg:
    print(f"  {(t - lg[0, 1]) / 100.0:8.1f}  {names.get(int(k), k)}")
lib.wfl_debug_marks(marks)
mk = np.array([min(int(v), 2**62) for v in marks], dtype=np.int64).reshape(8, 2)
tl = np.array(list(tiles), dtype=np.int64).reshape(4096, 5)[: n.value]
sw = np.array(list(sweeps), dtype=np.int64).reshape(512, 2)[: 2 * ((B + 7) & ~7)]
sw = sw[sw[:, 1] > 0]
t0 = sw[:, 0].min()
us = lambda v: (v - t0) / 100.0
print(f"{len(tl)} tiles; sweeps begin {us(sw[:, 0].min()):.1f}..{us(sw[:, 0].max()):.1f} us, end {us(sw[:, 1].min()):.1f}..{us(sw[:, 1].max()):.1f} us")
end = sw[:, 1].max()
print(f"last tile ends {us(tl[:, 4].max()):.1f} us = {(tl[:, 4].max() - end) / 100.0:.1f} us behind the last sweep")
late = tl[tl[:, 4] > end]
late = late[np.argsort(late[:, 4])]
print(f"{len(late)} tiles end behind the sweeps; the last twelve (utterance, first frame: wait began, wait over, occupancies done, rows done; us behind the last sweep's end):")
for r in late[-12:]:
    print(f"  b={r[0] >> 16:2d} t={r[0] & 0xffff:3d}: " + "  ".join(f"{(v - end) / 100.0:7.1f}" for v in r[1:]))
wait_over = (late[:, 2] - end) / 100.0
print(f"of those: wait over at {np.median(wait_over):.1f} us (median) / {wait_over.max():.1f} (max) behind the sweeps' end; occupancies {np.median((late[:, 3] - late[:, 2]) / 100.0):.1f} us, rows {np.median((late[:, 4] - late[:, 3]) / 100.0):.1f} us (medians)")
wgs = (ctypes.c_ulonglong * (1024 * 4))()
lib.wfl_debug_live_workgroups(wgs)
wg = np.array(list(wgs), dtype=np.int64).reshape(1024, 4)
wg = wg[wg[:, 0] > 0]
print(f"{len(wg)} gradient workgroups: began {us(wg[:, 0].min()):.1f}..{us(wg[:, 0].max()):.1f} us ({int((wg[:, 0] > end).sum())} of them behind the last sweep's end), "
      f"their CU free at {us(wg[:, 1].min()):.1f}..{us(wg[:, 1].max()):.1f}, last draw (= exit) at {us(wg[:, 2].min()):.1f}..{us(wg[:, 2].max()):.1f}; jobs per workgroup {wg[:, 3].min()}..{wg[:, 3].max()}")
late_wg = wg[np.argsort(wg[:, 2])][-6:]
for r in late_wg:
    print(f"  a late one: began {us(r[0]):.1f}, CU free {us(r[1]):.1f}, exit {us(r[2]):.1f}, jobs {r[3]}")
for k, name in enumerate(("certificate", "log-domain launch (nothing to do)", "gradient kernel of the rest (backward)")):
    print(f"first workgroup of the {name} begins at {us(mk[k, 0]):.1f} us")
Bp = (B + 7) & ~7
num = crit(x, tg).grad_fn.aux[2]
so = num.pack.field("state_off", B + 1)
Q = np.diff(np.asarray(so))
full = np.array(list(sweeps), dtype=np.int64).reshape(512, 2)
print("utterance: states, forward sweep us, backward sweep us (XCD = utterance % 8)")
for b in range(B):
    fa, fb = full[b], full[Bp + b]
    print(f"  b={b:2d} xcd={b % 8} Q={int(Q[b]):3d}  fwd {(fa[1] - fa[0]) / 100.0:6.1f}  bwd {(fb[1] - fb[0]) / 100.0:6.1f}")
