import gc, os, sys, collections
sys.path.insert(0, os.getcwd())
import torch
from gtn_applications_amd.criterions import ctc
B, T, C, L = 128, 1000, 100, 44
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
tg = torch.randint(C - 2, (B, L), generator=g).tolist()
def step():
    x.grad = None
    ctc.CTCLoss(x.view_as(x), tg, C - 1).backward()
for _ in range(5): step()
torch.cuda.synchronize()
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
n0 = len(gc.get_objects())
for _ in range(10): step()
torch.cuda.synchronize()
n1 = len(gc.get_objects())
found = gc.collect()
print("objects tracked before / after 10 steps:", n0, n1, "; unreachable found by the collector:", found)
print(collections.Counter(type(o).__name__ for o in gc.garbage).most_common(12))
for o in gc.garbage[:6]:
    print("  ", type(o).__name__, repr(o)[:160])
