"""Step of a Transducer with a `transitions=` back-off model (the pruned model of tests/transducer_test.py:534-566 from
tests/golden/reference_literals.json; T = 250, B = 16 as tests/test_gpu_ngram.py::test_backoff_transitions_at_benchmark_length):
wall time of loss + backward, pipelined and one step at a time, and the host time of the call."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gtn_applications_amd import graph as G
from gtn_applications_amd.criterions import transducer as TR

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
lit = json.load(open(os.path.join(root, "tests/golden/reference_literals.json")))["backoff_transitions"]
N, T, B = lit["N"], 250, int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = G.Graph(True)
for n in range(8):
    g.add_node(n in lit["start"], n in lit["accept"])
for a in lit["arcs"]:
    g.add_arc(*a)
rs = np.random.RandomState(5)
crit = TR.Transducer([(n,) for n in range(N)], {n: n for n in range(N)}, blank="optional", allow_repeats=False,
                     transitions=g, reduction="mean").cuda()
x = torch.from_numpy(rs.randn(B, T, N + 1).astype(np.float32)).cuda().requires_grad_(True)
targets = [torch.tensor(rs.randint(0, N, size=rs.randint(20, 45)).tolist()) for _ in range(B)]


def step():
    x.grad = None
    crit.transition_params.grad = None
    crit(x, targets).backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
host = 0.0
for _ in range(n):
    a = time.perf_counter()
    step()
    host += time.perf_counter() - a
torch.cuda.synchronize()
pipelined = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    step()
    torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / n
print(f"back-off transitions, B={B} T={T}: step {pipelined * 1e3:.3f} ms pipelined (host {host / n * 1e3:.3f} ms of it), {serial * 1e3:.3f} ms one at a time")
