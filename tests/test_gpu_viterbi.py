"""ASG.viterbi (asg.py:211-236) on the max-plus sweeps without back-pointers + the back-traces that re-derive the ones
they follow (C <= 256: csrc/dense_kernels.hip dense_viterbi_sweep_kernel / dense_viterbi_backtrace_kernel, transition rows
in registers; beyond: csrc/dense_wide.h wide_viterbi_max_kernel / dense_viterbi_walk_kernel) against

  * the launch it replaced -- wfl_dense_forward in the tropical semiring, which stores a back-pointer per (frame, state):
    the stored vectors must be IDENTICAL (same additions in the same order) and the path the one its back-pointers give,
    ties included;
  * the oracle's float64-free max-plus recurrence (oracle/recurrences.py::dense_viterbi) on integer scores, where ties
    are real ties.
"""
import numpy as np
import pytest
import torch

from oracle import recurrences as OR

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def old_launch(x, W):
    """(alpha [B,T,C], path [B,T]) of the back-pointer launch, through the C ABI"""
    from gtn_applications_amd import _native as N
    from gtn_applications_amd import engine as E

    B, T, C = x.shape
    alpha = torch.empty((B, T, C), dtype=torch.float32, device=x.device)
    bptr = torch.empty((B, T, C), dtype=torch.int32, device=x.device)
    N.check(N.lib.wfl_dense_forward(E.ptr(x), E.ptr(W), B, T, C, N.SEMIRING_TROPICAL, E.ptr(alpha), None, E.ptr(bptr), None,
                                    None, E.stream_ptr()))
    torch.cuda.synchronize()
    a, bp = alpha.cpu().numpy(), bptr.cpu().numpy()
    paths = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        cur = int(np.argmax(a[b, T - 1]))  # (first maximum: lowest final label)
        for t in range(T - 1, -1, -1):
            paths[b, t] = cur
            if t > 0:
                cur = max(int(bp[b, t, cur]), 0)
    return a, paths


def new_launch(x, W):
    from gtn_applications_amd import _native as N
    from gtn_applications_amd import engine as E

    B, T, C = x.shape
    alpha = torch.full((B, T, C), float("nan"), dtype=torch.float32, device=x.device)
    path = torch.full((B, T), -7, dtype=torch.int32, device=x.device)
    N.check(N.lib.wfl_dense_viterbi(E.ptr(x), E.ptr(W), B, T, C, E.ptr(alpha), None, E.ptr(path), E.stream_ptr()))
    torch.cuda.synchronize()
    return alpha.cpu().numpy(), path.cpu().numpy().astype(np.int64)


def same_values(a, b):
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


SHAPES = [(3, 17, 5), (4, 50, 32), (4, 33, 33), (2, 40, 64), (2, 41, 65), (3, 64, 100), (2, 30, 104), (2, 25, 105),
          (2, 20, 128), (2, 18, 129), (1, 19, 192), (2, 21, 193), (1, 16, 256), (2, 1, 7), (2, 2, 3), (1, 300, 82),
          # beyond 256 classes: one tiled launch per frame (csrc/dense_wide.h), 16-byte loads when C % 4 == 0
          # (257 .. 320: the frames of an utterance inside one workgroup, the matrix in its registers -- wide_resident_viterbi_kernel:
          # whole groups of its four-frame emission prefetch, a tail, two frames, the largest class count it takes, one more)
          (2, 12, 257), (3, 9, 300), (2, 2, 258), (3, 23, 320), (2, 40, 301), (1, 10, 321),
          (17, 7, 333), (2, 6, 1000), (1, 5, 1031)]


@pytest.mark.parametrize("B,T,C", SHAPES)
@pytest.mark.parametrize("kind", ["integers", "gaussian", "holes"])
def test_viterbi_equals_the_back_pointer_launch(B, T, C, kind):
    rs = np.random.RandomState(B * 1000 + T * 7 + C)
    if kind == "integers":  # exactly representable sums: many ties
        x = rs.randint(-3, 4, size=(B, T, C)).astype(np.float32)
        W = rs.randint(-2, 3, size=(C + 1, C)).astype(np.float32)
    else:
        x = rs.randn(B, T, C).astype(np.float32) * 3
        W = rs.randn(C + 1, C).astype(np.float32)
    if kind == "holes":  # impossible arcs and emissions: -inf, NaN (an impossible arc by the NaN policy), a dead frame
        x[rs.rand(B, T, C) < 0.1] = -np.inf
        x[rs.rand(B, T, C) < 0.02] = np.nan
        W[rs.rand(C + 1, C) < 0.2] = -np.inf
        W[rs.rand(C + 1, C) < 0.02] = np.nan
        if T > 5:
            x[0, T // 2, :] = -np.inf  # no path survives this frame
    xd, Wd = dev(x), dev(W)
    a_old, p_old = old_launch(xd, Wd)
    a_new, p_new = new_launch(xd, Wd)
    assert same_values(a_new, a_old)
    assert np.array_equal(p_new, p_old)


@pytest.mark.parametrize("C", [7, 82, 100, 150, 256, 300])
def test_viterbi_vs_oracle_integer_scores(C):
    rs = np.random.RandomState(C)
    B, T = 3, 40
    x = rs.randint(-6, 7, size=(B, T, C)).astype(np.float32)
    W = rs.randint(-3, 4, size=(C + 1, C)).astype(np.float32)
    _, got = new_launch(dev(x), dev(W))
    assert got.tolist() == [OR.dense_viterbi(x[b], W) for b in range(B)]


def test_viterbi_at_the_benchmark_shape_equals_the_back_pointer_launch():
    """asg_benchmark.py's shape (T = 1000, C = 100, B = 128): every utterance, vectors and paths"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(128, 1000, 100, generator=g).cuda()
    W = torch.randn(101, 100, generator=g).cuda()
    a_old, p_old = old_launch(x, W)
    a_new, p_new = new_launch(x, W)
    assert same_values(a_new, a_old)
    assert np.array_equal(p_new, p_old)


def test_collapse_on_the_device_equals_the_host_spelling():
    """ASG.viterbi collapses its paths on the device before they travel (criterions/asg.py::collapse_and_unpack with a
    tensor); the same function on a numpy array is pinned to the reference's row-by-row spelling on the CPU
    (tests/test_host_library.py).  Same lists."""
    from gtn_applications_amd.criterions import asg

    rs = np.random.RandomState(4)
    for trial in range(12):
        B, T, R = rs.randint(1, 9), rs.randint(1, 70), rs.randint(1, 4)
        C = R + rs.randint(1, 6) + 1
        garbage = None if trial % 3 == 0 else C - 1
        paths = np.repeat(rs.randint(0, C, size=(B, (T + 2) // 3)).astype(np.int32), 3, axis=1)[:, :T]
        want = [t.tolist() for t in asg.collapse_and_unpack(paths, garbage, R)]
        got = [t.tolist() for t in asg.collapse_and_unpack(torch.from_numpy(paths).cuda(), garbage, R)]
        assert got == want
