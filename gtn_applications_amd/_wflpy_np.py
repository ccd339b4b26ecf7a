"""Pure-Python stand-in for the CPython helper `_wflpy` (csrc/wflpy.c), used only when that extension was not built:
the same four functions over raw addresses (ctypes / numpy views), slower, same results.  Host-side target staging
only -- no arithmetic of the path happens here."""
import ctypes
import hashlib

import numpy as np


def _view(addr, dtype, n):
    buf = (ctypes.c_char * (np.dtype(dtype).itemsize * max(int(n), 1))).from_address(int(addr))
    return np.frombuffer(buf, dtype=dtype, count=int(n))


def flatten_into(targets, flat_addr, capacity, off_addr):
    """targets: list / tuple of lists / tuples of ints -> int32 labels at flat_addr (at most `capacity`), int64 offsets
    [B+1] at off_addr.  Returns (total, max_len, lo, hi), or None when the labels do not fit (the offsets are written
    anyway: their last entry tells the caller how much room to come back with).  TypeError for anything else."""
    if not isinstance(targets, (list, tuple)):
        raise TypeError("targets must be a list or tuple")
    B = len(targets)
    lens = []
    for r in targets:
        if not isinstance(r, (list, tuple)):
            raise TypeError("targets must be lists or tuples of ints")
        lens.append(len(r))
    off = _view(off_addr, np.int64, B + 1)
    off[0] = 0
    if B:
        np.cumsum(lens, out=off[1:])
    total = int(off[B])
    if total > capacity:
        return None
    if total:
        rows = [v for r in targets for v in r]
        if not all(type(v) is int for v in rows):
            raise TypeError("targets must be lists or tuples of ints")
        arr = np.asarray(rows, dtype=np.int64)
        if arr.min() < -(1 << 31) or arr.max() >= (1 << 31):
            raise ValueError("target label does not fit int32")
        _view(flat_addr, np.int32, total)[:] = arr
        lo, hi = int(arr.min()), int(arr.max())
    else:
        lo, hi = 0, -1
    return total, (max(lens) if lens else 0), lo, hi


def factors_into(off_addr, B, fac_addr):
    """six float32 arrays [B] behind each other: scale_none, scale_mean, then both times +1/B and -1/B"""
    off = _view(off_addr, np.int64, B + 1)
    fac = _view(fac_addr, np.float32, 6 * B)
    ln = np.diff(off).astype(np.float32)
    mean = np.where(ln > 0, np.float32(1.0) / np.maximum(ln, np.float32(1.0)), np.float32(1.0)).astype(np.float32)
    inv_b = np.float32(1.0) / np.float32(B if B > 0 else 1)
    fac[0 * B:1 * B] = 1.0
    fac[1 * B:2 * B] = mean
    fac[2 * B:3 * B] = inv_b
    fac[3 * B:4 * B] = mean * inv_b
    fac[4 * B:5 * B] = -inv_b
    fac[5 * B:6 * B] = mean * -inv_b


def content_key(addr, n):
    """128-bit hash of n staged bytes (a cache key within this process: need not equal the C helper's)"""
    d = hashlib.blake2b(bytes(_view(addr, np.uint8, n)), digest_size=16).digest()
    return int.from_bytes(d[:8], "little"), int.from_bytes(d[8:], "little")


def same_bytes(addr, b):
    return bytes(_view(addr, np.uint8, len(b))) == b
