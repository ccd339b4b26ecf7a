"""Counterpart of the reference's benchmarks/transducer_benchmark.py: the same three scenarios
(word-piece decompositions; CTC-like and ASG-like token graphs with n-gram transitions, n = 0, 1, 2),
same shapes, same 20-iteration protocol, the reference's own token list
(benchmarks/word_pieces_tokens_1000.txt: 1000 word pieces over 78 graphemes -- a data file, shipped as a fixture).
Usage: python benchmarks/transducer_benchmark.py [B]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd import compat  # noqa: E402

compat.install()
import transducer  # noqa: E402  (the reference's import line, benchmarks/transducer_benchmark.py:13)

from time_utils import time_func  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
random.seed(0)
torch.manual_seed(0)


def word_decompositions():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "word_pieces_tokens_1000.txt"), "r") as fid:
        tokens = sorted([l.strip() for l in fid])
    graphemes = sorted(set(c for t in tokens for c in t))
    graphemes_to_index = {t: i for i, t in enumerate(graphemes)}
    N, T, L = len(tokens) + 1, 100, 15
    inputs = torch.randn(B, T, N, dtype=torch.float).cuda().requires_grad_(True)
    targets = [torch.tensor([graphemes_to_index[c] for _ in range(L) for c in random.choice(tokens)]) for _ in range(B)]
    crit = transducer.Transducer(tokens, graphemes_to_index, blank="optional", allow_repeats=False, reduction="mean")

    def fwd_bwd():
        inputs.grad = None
        crit(inputs, targets).backward()

    time_func(fwd_bwd, 20, "word decomps fwd + bwd")
    time_func(lambda: crit.viterbi(inputs), 20, "word decomps viterbi")


def ngram(kind):
    N, T, L = 81, 250, 44
    tokens = [(i,) for i in range(N)]
    graphemes_to_index = {i: i for i in range(N)}
    extra = 1 if kind == "ctc" else 0
    inputs = torch.randn(B, T, N + extra, dtype=torch.float).cuda().requires_grad_(True)
    targets = [t.squeeze() for t in torch.randint(N, size=(B, L)).split(1)]
    for n in (0, 1, 2):
        kw = dict(blank="optional", allow_repeats=False) if kind == "ctc" else {}
        crit = transducer.Transducer(tokens, graphemes_to_index, ngram=n, reduction="mean", **kw).cuda()

        def fwd_bwd():
            inputs.grad = None
            crit(inputs, targets).backward()

        time_func(fwd_bwd, 20, f"{kind} fwd + bwd, ngram={n}")
        time_func(lambda: crit.viterbi(inputs), 20, f"{kind} viterbi, ngram={n}")


if __name__ == "__main__":
    word_decompositions()
    ngram("ctc")
    ngram("asg")
