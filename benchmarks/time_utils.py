"""Timing protocol of the reference's micro-benchmarks (benchmarks/time_utils.py:11-21): 5 warm-up
calls, then `iterations` timed calls, wall clock, ms per call -- plus the device synchronisation the
reference did not need because its op was synchronous on the CPU."""
import time

import torch


def time_func(func, iterations=100, name=None):
    for _ in range(5):
        func()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(iterations):
        func()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - start) * 1e3 / iterations
    print('"{}" took {:.3f} (ms)'.format("function" if name is None else name, ms))
    return ms
