"""scratch: shape of the packed alignment lattices of the cfg4 Transducer workload (host side only)."""
import sys, os, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import engine as E
from gtn_applications_amd.criterions import transducer
rnd = random.Random(0)
letters = "abcdefghijklmnopqrstuvwxyz"
pieces = set(letters)
while len(pieces) < 1000:
    pieces.add("".join(rnd.choice(letters) for _ in range(rnd.choice([2, 3, 4, 5, 6, 7]))))
tokens = sorted(pieces)
g2i = {c: i for i, c in enumerate(letters)}
targets = [torch.tensor([g2i[ch] for _ in range(15) for ch in rnd.choice(tokens)]) for _ in range(8)]
crit = transducer.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
orig = E.PackedLattice.from_graphs
def spy(graphs, C, device, **kw):
    p = orig(graphs, C, None, **kw)
    d = p.desc
    print({k: getattr(d, k) for k in ("B", "max_states", "max_arcs", "max_labels", "max_eps", "max_levels") if hasattr(d, k)})
    for g in graphs[:3]:
        print("states", g.num_nodes(), "arcs", g.num_arcs())
    raise SystemExit
E.PackedLattice.from_graphs = classmethod(lambda cls, *a, **k: spy(*a, **k))
E.require_gpu = lambda: torch.device("cpu")
try:
    crit(torch.zeros(8, 10, len(tokens) + 1), targets)
except SystemExit:
    pass
