import gc, os, sys, collections, random
sys.path.insert(0, os.getcwd())
import torch, bench
from gtn_applications_amd.criterions import transducer as TR, asg
which = sys.argv[1] if len(sys.argv) > 1 else "transducer"
if which == "transducer":
    B, T, Lp = 16, 400, 15
    tokens, g2i = bench.word_pieces()
    C = len(tokens) + 1
    rnd = random.Random(0)
    x = torch.randn(B, T, C).cuda().requires_grad_(True)
    tg = [torch.tensor([g2i[ch] for _ in range(Lp) for ch in rnd.choice(tokens)]) for _ in range(B)]
    crit = TR.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
    def step():
        x.grad = None
        crit(x.view_as(x), tg).backward()
else:
    B, T, C, L = 32, 300, 100, 20
    x = torch.randn(B, T, C).cuda().requires_grad_(True)
    W = torch.randn(C + 1, C).cuda().requires_grad_(True)
    tg = torch.randint(C, (B, L)).tolist()
    def step():
        x.grad = None; W.grad = None
        asg.ASGLoss(x.view_as(x), W, tg, "mean").backward()
for _ in range(5): step()
torch.cuda.synchronize()
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
for _ in range(10): step()
torch.cuda.synchronize()
found = gc.collect()
print(which, "unreachable found by the collector after 10 steps:", found)
print(collections.Counter(type(o).__name__ for o in gc.garbage).most_common(12))
for o in gc.garbage[:40]:
    if type(o).__name__ not in ("tuple", "cell", "list", "dict"):
        print("  ", type(o).__name__, repr(o)[:140])
