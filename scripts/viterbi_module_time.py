"""What train.py:279 pays per step for `criterion.viterbi(outputs)` at the benchmark shapes: the MODULE call (device
decode + copy to the host + collapse / unpack), and beside it the row-by-row spelling of the host part alone (the
reference's asg.py:228-234 / ctc.py:130-134), for comparison."""
import itertools, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gtn_applications_amd.criterions import asg, ctc


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
B, T, C = 128, 1000, 100
x = torch.randn(B, T, C).cuda()
crit = asg.ASG(C - 2, 1, True).cuda()  # 98 tokens + 1 replabel + garbage = 100 classes
with torch.no_grad():
    crit.transitions.normal_()
print(f"ASG.viterbi  B={B} T={T} C={C}: {timed(lambda: crit.viterbi(x)):.2f} ms per call")
from gtn_applications_amd import engine as E
paths = E.dense_viterbi(x, crit.transitions.detach()).cpu()


def rowwise():
    out = []
    for path in paths.tolist():
        col = [p for p, _ in itertools.groupby(path)]
        col = [p for p in col if p != crit.garbage_idx]
        out.append(torch.IntTensor(asg.unpack_replabels(col, crit.num_replabels)))
    return out


print(f"   its host part row by row (asg.py:228-234): {timed(rowwise, 5):.2f} ms")
c = ctc.CTC(C - 1, True)
print(f"CTC.viterbi  B={B} T={T} C={C}: {timed(lambda: c.viterbi(x)):.2f} ms per call")
best = torch.argmax(x, dim=2).cpu()


def ctc_rowwise():
    res = []
    for row in best:
        keep = torch.ones_like(row, dtype=torch.bool)
        keep[1:] = row[1:] != row[:-1]
        row = row[keep]
        res.append(row[row != C - 1])
    return res


print(f"   its host part row by row (ctc.py:130-134): {timed(ctc_rowwise, 5):.2f} ms")
