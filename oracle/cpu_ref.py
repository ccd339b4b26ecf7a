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
        P = ctypes.c_void_p
        _LIB.oracle_asg_cpu.restype = ctypes.c_int
        _LIB.oracle_asg_cpu.argtypes = [P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P, ctypes.c_int, P, P, P]
        _LIB.oracle_lattice_cpu.restype = ctypes.c_int
        _LIB.oracle_lattice_cpu.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P, P, P, P, P,
                                            P, ctypes.c_int, P, P]
    return _LIB


def _flat_targets(targets):
    lens = [len(t) for t in targets]
    flat = np.ascontiguousarray([v for t in targets for v in t], dtype=np.int32)
    if flat.size == 0:
        flat = np.zeros(1, np.int32)
    off = np.zeros(len(targets) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    return flat, off, lens


def asg_cpu(x, W, targets, reduction="none", nthreads=1, want_grad=True):
    """Graph-faithful float32 CPU ASG (criterions/asg.py:84-185): (mean loss, dx, dW) (grads None if not wanted)."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    B, T, C = x.shape
    flat, off, lens = _flat_targets(targets)
    sc = np.array([(1.0 / n if (reduction == "mean" and n > 0) else 1.0) for n in lens], dtype=np.float32)
    gscale = (sc / B).astype(np.float32)
    losses = np.zeros(B, np.float32)
    gx = np.zeros_like(x) if want_grad else None
    gW = np.zeros_like(W) if want_grad else None
    lib.oracle_asg_cpu(x.ctypes.data, W.ctypes.data, B, T, C, flat.ctypes.data, off.ctypes.data, gscale.ctypes.data,
                       int(nthreads), losses.ctypes.data, None if gx is None else gx.ctypes.data,
                       None if gW is None else gW.ctypes.data)
    return float(np.mean(losses * sc)), gx, gW


def lattice_cpu(x, acceptors, scales, log_softmax=False, nthreads=1, want_grad=True):
    """-scale_b * forward_score(intersect(emissions_b, A_b)) with A_b = dict(src, dst, lab, start, accept) of an
    epsilon-free acceptor (start / accept: node masks), optionally through log_softmax
    (criterions/transducer.py:186-187,283,302-305,321-336).  Returns (mean scaled loss, dx or None)."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, T, C = x.shape
    node_off = np.zeros(B + 1, np.int64)
    arc_off = np.zeros(B + 1, np.int64)
    np.cumsum([len(a["start"]) for a in acceptors], out=node_off[1:])
    np.cumsum([len(a["src"]) for a in acceptors], out=arc_off[1:])
    cat = {k: np.ascontiguousarray(np.concatenate([np.asarray(a[k]) for a in acceptors]),
                                   dtype=np.uint8 if k in ("start", "accept") else np.int32)
           for k in ("src", "dst", "lab", "start", "accept")}
    sc = np.ascontiguousarray(scales, dtype=np.float32)
    gscale = (sc / B).astype(np.float32)
    losses = np.zeros(B, np.float32)
    gx = np.zeros_like(x) if want_grad else None
    lib.oracle_lattice_cpu(x.ctypes.data, B, T, C, int(bool(log_softmax)), node_off.ctypes.data, arc_off.ctypes.data,
                           cat["src"].ctypes.data, cat["dst"].ctypes.data, cat["lab"].ctypes.data,
                           cat["start"].ctypes.data, cat["accept"].ctypes.data, gscale.ctypes.data, int(nthreads),
                           losses.ctypes.data, None if gx is None else gx.ctypes.data)
    return float(np.mean(losses * sc)), gx


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
