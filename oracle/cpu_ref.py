"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/cpu_ref.c (see its header)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcpu_ref.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_ctc_cpu.restype = ctypes.c_int
        _LIB.oracle_ctc_cpu.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p]
    return _LIB


def ctc_cpu(x, targets, blank, reduction="none", nthreads=1, want_grad=True):
    """Graph-faithful float32 CPU CTC (criterions/ctc.py:32-94): returns (mean loss, dx or None)."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, T, C = x.shape
    lens = [len(t) for t in targets]
    flat = np.ascontiguousarray([v for t in targets for v in t], dtype=np.int32)
    if flat.size == 0:
        flat = np.zeros(1, np.int32)
    off = np.zeros(B + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    sc = np.array([(1.0 / n if (reduction == "mean" and n > 0) else 1.0) for n in lens], dtype=np.float32)
    losses = np.zeros(B, np.float32)
    grad = np.zeros_like(x) if want_grad else None
    gscale = (sc / B).astype(np.float32)
    lib.oracle_ctc_cpu(x.ctypes.data, B, T, C, flat.ctypes.data, off.ctypes.data, int(blank), gscale.ctypes.data,
                       int(nthreads), losses.ctypes.data, None if grad is None else grad.ctypes.data)
    return float(np.mean(losses * sc)), grad
