import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd.criterions import transducer as TR
N, T, L, B = 81, 250, 44, 16
toks = [(i,) for i in range(N)]
g2 = {i: i for i in range(N)}
x = torch.randn(B, T, N + 1).cuda()
crit = TR.Transducer(toks, g2, ngram=2, reduction="mean", blank="optional", allow_repeats=False).cuda()
for _ in range(5):
    crit.viterbi(x)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    crit.viterbi(x)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
