import os, sys, random
sys.path.insert(0, os.getcwd())
import torch, bench
from gtn_applications_amd.criterions import transducer as TR
B, T, Lp = 64, 800, 15
tokens, g2i = bench.word_pieces()
C = len(tokens) + 1
rnd = random.Random(3)
xs = [torch.randn(B, T, C, generator=torch.Generator().manual_seed(s)).cuda() for s in (1, 2)]
tgs = [[torch.tensor([g2i[ch] for _ in range(Lp) for ch in rnd.choice(tokens)]) for _ in range(B)] for _ in range(2)]
crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
def run(k):
    x = xs[k].clone().requires_grad_(True)
    loss = crit(x.view_as(x), tgs[k])
    loss.backward()
    return loss.detach(), x.grad
os.environ["WFL_LATTICE_TWO_PHASE"] = "0"
ref = [run(k) for k in (0, 1)]
torch.cuda.synchronize()
os.environ["WFL_LATTICE_TWO_PHASE"] = "1"
worst = 0.0
keep = []
for it in range(400):
    k = it & 1
    l, g = run(k)
    keep.append((k, l, g))
    if len(keep) == 40:  # (checked in bursts: the steps themselves run back to back)
        for kk, ll, gg in keep:
            assert float(ll) == float(ref[kk][0]), (it, float(ll), float(ref[kk][0]))
            worst = max(worst, float((gg - ref[kk][1]).abs().max()))
        keep = []
scale = float(ref[0][1].abs().max())
print(f"400 steps back to back, two batches alternating: losses bit-identical to the one-phase build, max |dx - dx_one_phase| = {worst:.3e} (largest |dx| {scale:.3e})")
assert worst <= 1e-6 * max(scale, 1e-3) + 1e-9
