import os, sys, ctypes, random
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd import engine as E, _native as N
from gtn_applications_amd.criterions import transducer
import bench
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 160
tokens, g2i = bench.word_pieces()
C = len(tokens) + 1
rnd = random.Random(0)
x = torch.randn(B, T, C, generator=torch.Generator().manual_seed(0)).cuda().requires_grad_(True)
tg = [torch.tensor([g2i[ch] for _ in range(15) for ch in rnd.choice(tokens)]) for _ in range(B)]
crit = transducer.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
loss = crit(x, tg)
torch.cuda.synchronize()
node = loss.grad_fn
num = node.aux[2]
d = num.pack.desc
al, be = num.alpha, num.beta
off = ctypes.c_int64()
N.check(N.lib.wfl_lattice_formats_offset(ctypes.byref(d), T, ctypes.byref(off)))
fmt = al[off.value:off.value + B].view(torch.int32).cpu().numpy()
nch1 = T + 1
tail = off.value - 2 * (B * nch1 + B)
verdict = be[tail + 2 * (B * nch1 + B): tail + 2 * (B * nch1 + B) + 2 * B].view(torch.float64).cpu().numpy()
prog_off = tail + 2 * (B * nch1 + 2 * B + 1024 + 1)
prog_a = al[prog_off:prog_off + 2 * B].view(torch.int64).cpu().numpy()
prog_b = be[prog_off:prog_off + 2 * B].view(torch.int64).cpu().numpy()
mitm_b = be[prog_off + 2 * B: prog_off + 3 * B].view(torch.int32).cpu().numpy()
print("fmt", fmt, "\nverdict", verdict, "\nmitm_b", mitm_b)
print("prog_a", [hex(int(v) & 0xffffffff) for v in prog_a], "\nprog_b", [hex(int(v) & 0xffffffff) for v in prog_b])
# utterance 0: its state count, gamma sums per slot
so = num.pack.field("state_off", B + 1)
Q0 = int(so[1] - so[0])
mid = 16 * ((T // 16) // 2)
a64 = al[:2 * (T + 1) * Q0].view(torch.float64).view(T + 1, Q0).cpu().numpy()
b64 = be[:2 * (T + 1) * Q0].view(torch.float64).view(T + 1, Q0).cpu().numpy()
a32 = al[:2 * (T + 1) * Q0].view(T + 1, 2 * Q0)[:, :Q0].cpu().numpy()
b32 = be[:2 * (T + 1) * Q0].view(T + 1, 2 * Q0)[:, :Q0].cpu().numpy()
print("Q0", Q0, "mid", mid)
for s_ in (0, 1, 8, mid - 8, mid - 1, mid, mid + 1, mid + 8, T - 8, T):
    src = b32 if s_ <= mid else a32
    print("slot", s_, "gamma sum", float(src[s_].sum()), "max", float(src[s_].max()))
offs_a = al[tail:tail + 2 * nch1].view(torch.float64).cpu().numpy()
offs_b = be[tail:tail + 2 * nch1].view(torch.float64).cpu().numpy()
za = al[tail + 2 * B * nch1: tail + 2 * B * nch1 + 2 * B].view(torch.float64).cpu().numpy()
zb = be[tail + 2 * B * nch1: tail + 2 * B * nch1 + 2 * B].view(torch.float64).cpu().numpy()
print("za", za[:4], "zb", zb[:4])
print("alpha_mid . beta_mid (doubles):", float((a64[mid] * b64[mid]).sum()) if fmt[0] != 2 else "n/a (beta row overwritten)")
print("log2 sum alpha_mid*? offs", offs_a[mid], offs_b[mid])
