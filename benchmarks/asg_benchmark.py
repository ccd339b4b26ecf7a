"""Counterpart of the reference's benchmarks/asg_benchmark.py.  Usage: python benchmarks/asg_benchmark.py B [T] [N] [L]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd import compat  # noqa: E402

compat.install()
from utils import ASGLoss  # noqa: E402

from time_utils import time_func  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
N = int(sys.argv[3]) if len(sys.argv) > 3 else 80
L = int(sys.argv[4]) if len(sys.argv) > 4 else 44
torch.manual_seed(0)
inputs = torch.randn(B, T, N, dtype=torch.float).cuda().requires_grad_(True)
transitions = torch.randn(N + 1, N, dtype=torch.float).cuda().requires_grad_(True)
tgt = [t.tolist()[0] for t in torch.randint(N - 2, (B, L)).split(1)]


def func():
    inputs.grad = None
    transitions.grad = None
    op = ASGLoss(inputs, transitions, tgt)
    op.backward()


ms = time_func(func, name="asg fwd + bwd")
print("utterances/s: %.0f" % (B / (ms * 1e-3)))
