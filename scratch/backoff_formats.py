import json, os, sys, ctypes
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gtn_applications_amd import graph as G, _native as N
from gtn_applications_amd.criterions import transducer as TR
lit = json.load(open("tests/golden/reference_literals.json"))["backoff_transitions"]
Nn, T, B = lit["N"], 250, 16
g = G.Graph(True)
for n in range(8): g.add_node(n in lit["start"], n in lit["accept"])
for a in lit["arcs"]: g.add_arc(*a)
rs = np.random.RandomState(5)
crit = TR.Transducer([(n,) for n in range(Nn)], {n: n for n in range(Nn)}, blank="optional", allow_repeats=False, transitions=g, reduction="mean").cuda()
x = torch.from_numpy(rs.randn(B, T, Nn + 1).astype(np.float32)).cuda().requires_grad_(True)
targets = [torch.tensor(rs.randint(0, Nn, size=rs.randint(20, 45)).tolist()) for _ in range(B)]
loss = crit(x, targets)
num = loss.grad_fn.aux[2]
off = ctypes.c_int64()
N.check(N.lib.wfl_lattice_formats_offset(ctypes.byref(num.pack.desc), T, ctypes.byref(off)))
torch.cuda.synchronize()
print("backoff formats", num.alpha[off.value:off.value + B].view(torch.int32).cpu().tolist(), "states", num.pack.desc.max_states, "arcs", num.pack.desc.max_arcs, "eps", num.pack.desc.max_eps, "levels", num.pack.desc.max_levels)
den = TR._transitions_pack(crit.transitions, B, Nn + 1, x.device)
print("denominator pack: states", den.desc.max_states, "arcs", den.desc.max_arcs, "eps", den.desc.max_eps, "levels", den.desc.max_levels, "labels", den.desc.max_labels)
import numpy as np
a = crit.transitions.arrays()
indeg = np.bincount(a["dst"], minlength=den.desc.max_states)
print("in-degrees (labelled + eps)", indeg.tolist())
