"""The general lattice gradient kernel on the n-gram benchmark's numerator (N = 81, T = 250, L = 44, bigram): time of
wfl_lattice_grad with dx only, dW only, both (events around 20 calls)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gtn_applications_amd import engine as E
from gtn_applications_amd.criterions import transducer as TR

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, T, L = 81, 250, 44
torch.manual_seed(0)
tokens = [(i,) for i in range(N)]
crit = TR.Transducer(tokens, {i: i for i in range(N)}, ngram=2, reduction="mean", blank="optional", allow_repeats=False).cuda()
C = N + 1
x = torch.randn(B, T, C).cuda()
targets = [t.squeeze() for t in torch.randint(N, size=(B, L)).split(1)]
dev = x.device
params = crit.transition_params.detach()
crit.tokens.arc_sort(True)
nb, entry = TR._pack_entry(targets, crit.tokens, crit.lexicon, TR._numerator_transitions(crit.transitions, C), C, dev, "mean")
pack, scale, cpos, cneg, _ = entry
num = E.lattice_forward(x, pack, weights=params, need_beta=True)
torch.cuda.synchronize()
print("formats:", E.lattice_formats(num).cpu().tolist()[:8], "states", pack.desc.max_states, "arcs", pack.desc.max_arcs, "labels",
      pack.desc.max_labels, "eps", pack.desc.max_eps)
for what in ("dx", "dW", "both"):
    dx = torch.zeros_like(x) if what != "dW" else None
    dW = torch.zeros_like(params) if what != "dx" else None
    for _ in range(3):
        E.lattice_grad(num, cneg, coef_w=cneg, dx=dx, accumulate=False, dW=dW)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        E.lattice_grad(num, cneg, coef_w=cneg, dx=dx, accumulate=False, dW=dW)
    e1.record()
    torch.cuda.synchronize()
    print(f"lattice_grad {what}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
