"""Host profile of Transducer.viterbi at the word-piece benchmark's shape (B = 64, T = 800, 1000 word pieces)."""
import os, sys, time, random, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gtn_applications_amd.criterions import transducer

random.seed(0); torch.manual_seed(0)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "benchmarks", "word_pieces_tokens_1000.txt")) as f:
    tokens = sorted(l.strip() for l in f)
graphemes = sorted(set(c for t in tokens for c in t))
g2i = {t: i for i, t in enumerate(graphemes)}
B, T = 64, 800
x = torch.randn(B, T, len(tokens) + 1).cuda()
crit = transducer.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")
for _ in range(5): crit.viterbi(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): crit.viterbi(x)
torch.cuda.synchronize()
print(f"Transducer.viterbi B={B} T={T}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): crit.viterbi(x)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
