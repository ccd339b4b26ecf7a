"""Counterpart of the reference's benchmarks/ctc_benchmark.py (same inputs and protocol, flat-layout
import spelling included).  Usage: python benchmarks/ctc_benchmark.py B [T] [N] [L]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gtn_applications_amd import compat  # noqa: E402

compat.install()
from utils import CTCLoss  # noqa: E402  (the reference's import line, benchmarks/ctc_benchmark.py:13)

from time_utils import time_func  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
N = int(sys.argv[3]) if len(sys.argv) > 3 else 80
L = int(sys.argv[4]) if len(sys.argv) > 4 else 44
torch.manual_seed(0)
inputs = torch.randn(B, T, N, dtype=torch.float).cuda().requires_grad_(True)
tgt = [t.tolist()[0] for t in torch.randint(N - 2, (B, L)).split(1)]


def func():
    inputs.grad = None
    op = CTCLoss(inputs, tgt, N - 1)
    op.backward()


ms = time_func(func, name="ctc fwd + bwd")
print("utterances/s: %.0f" % (B / (ms * 1e-3)))

# the criterion module of the training loop (criterions/ctc.py:41-63 in the reference): raw scores in,
# log_softmax + loss + gradient -- fused into the pipelined launch here vs torch's log_softmax in front
from criterions.ctc import CTC  # noqa: E402

module = CTC(N - 1, False)
targets = [torch.tensor(t) for t in tgt]


def fused():
    inputs.grad = None
    module(inputs, targets).backward()


def unfused():
    inputs.grad = None
    CTCLoss(torch.nn.functional.log_softmax(inputs, dim=2), tgt, N - 1).backward()


time_func(fused, name="CTC module (log_softmax fused) fwd + bwd")
time_func(unfused, name="torch log_softmax + CTCLoss fwd + bwd")
