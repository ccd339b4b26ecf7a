"""ASG criterion -- counterpart of /root/reference/criterions/asg.py.

loss_b = forward_score(emissions o transitions) - forward_score(force_align o transitions o emissions)
(asg.py:111-115).  The denominator runs on the dense-transition kernels (csrc/dense_kernels.hip),
the numerator on the lattice engine with arcs that index the (C+1) x C transition matrix; the
reference's per-sample rebuild of the C^2-arc transitions graph (asg.py:54-69,103) disappears: the
transitions tensor is the graph.
"""
import ctypes
import itertools
import os

import torch

from .. import engine as E
from .. import graph as G


def pack_replabels(tokens, num_replabels):
    """asg.py:13-32: replace runs of a repeated token by a "repeat k times" label."""
    if all(isinstance(t, list) for t in tokens):
        return [pack_replabels(t, num_replabels) for t in tokens]
    assert isinstance(tokens, list)
    out, run, prev = [], 0, -1
    for tok in tokens:
        if tok == prev and run < num_replabels:
            run += 1
            continue
        if run > 0:
            out.append(run - 1)
            run = 0
        out.append(tok + num_replabels)
        prev = tok
    if run > 0:
        out.append(run - 1)
    return out


def unpack_replabels(tokens, num_replabels):
    """asg.py:35-49: inverse of pack_replabels."""
    if all(isinstance(t, list) for t in tokens):
        return [unpack_replabels(t, num_replabels) for t in tokens]
    assert isinstance(tokens, list)
    out, prev = [], -1
    for tok in tokens:
        if tok >= num_replabels:
            out.append(tok - num_replabels)
            prev = tok
        elif prev != -1:
            out.extend([prev - num_replabels] * (tok + 1))
            prev = -1
    return out


def pack_targets_batch(targets, num_replabels, garbage_idx):
    """ASG.forward's target preparation (asg.py:201-208: pack_replabels per target, then a garbage label between and
    around the labels) for a whole batch of 1-D CPU tensors as array operations; returns a list of B int64 tensors
    (views of one).  A run of n equal labels becomes groups of up to 1 + num_replabels: the label (shifted by
    num_replabels), then -- for a group of s >= 2 -- the replabel s - 2."""
    import numpy as np

    B = len(targets)
    lens = np.fromiter((t.numel() for t in targets), np.int64, B)
    total = int(lens.sum())
    flat = torch.cat([t.reshape(-1) for t in targets]).to(torch.int64).numpy() if total else np.zeros(0, np.int64)
    R, P = num_replabels, num_replabels + 1
    starts = np.zeros(B, dtype=np.int64)
    np.cumsum(lens[:-1], out=starts[1:])
    rows = np.repeat(np.arange(B), lens)
    if total:
        brk = np.ones(total, dtype=bool)
        brk[1:] = flat[1:] != flat[:-1]
        brk[starts[lens > 0]] = True
        run_start = np.flatnonzero(brk)
        run_id = np.cumsum(brk) - 1
        pos = (np.arange(total) - run_start[run_id]) % P          # position inside the group
        run_end = np.empty(total, dtype=bool)                      # last element of its run
        run_end[:-1] = brk[1:]
        run_end[-1] = True
        is_label = pos == 0
        is_rep = (pos >= 1) & (run_end | (pos == P - 1))            # last element of a group of >= 2
        keep = is_label | is_rep
        vals = np.where(is_label, flat + R, pos - 1)[keep]
        out_rows = rows[keep]
        out_lens = np.bincount(out_rows, minlength=B).astype(np.int64)
    else:
        vals, out_lens = np.zeros(0, np.int64), np.zeros(B, np.int64)
    if garbage_idx is not None:  # [g, t0, g, t1, ..., g]
        new_lens = 2 * out_lens + 1
        new_starts = np.zeros(B, dtype=np.int64)
        np.cumsum(new_lens[:-1], out=new_starts[1:])
        res = np.full(int(new_lens.sum()), garbage_idx, dtype=np.int64)
        if len(vals):
            old_starts = np.zeros(B, dtype=np.int64)
            np.cumsum(out_lens[:-1], out=old_starts[1:])
            r = np.repeat(np.arange(B), out_lens)
            res[new_starts[r] + 2 * (np.arange(len(vals)) - old_starts[r]) + 1] = vals
        vals, out_lens = res, new_lens
    return list(torch.split(torch.from_numpy(np.ascontiguousarray(vals, dtype=np.int64)), out_lens.tolist()))


def collapse_and_unpack(paths, garbage_idx, num_replabels):
    """asg.py:228-234 for a whole batch (numpy [B,T] label paths): repeats collapsed, the garbage label dropped,
    replabels unpacked (unpack_replabels above: a replabel r right behind a label repeats that label r + 1 times, any
    other replabel is dropped) -- as array operations instead of a Python loop over 128 x 1000 labels.
    Returns a list of B torch.IntTensor."""
    import numpy as np

    B = paths.shape[0]
    if isinstance(paths, torch.Tensor):
        # on the device: the same collapse as five small launches, and only what survives it travels to the host
        keep = torch.ones_like(paths, dtype=torch.bool)
        if paths.shape[1] > 1:
            keep[:, 1:] = paths[:, 1:] != paths[:, :-1]
        if garbage_idx is not None:
            keep &= paths != garbage_idx
        lens = keep.sum(dim=1).cpu().numpy()
        flat = paths[keep].cpu().numpy()
    else:
        flat, lens = E.collapse_rows(paths, drop=garbage_idx)
    R = num_replabels
    is_lab = flat >= R
    prev_lab = np.zeros(len(flat), dtype=bool)
    prev_lab[1:] = is_lab[:-1]
    starts = np.zeros(B, dtype=np.int64)
    np.cumsum(lens[:-1], out=starts[1:])
    prev_lab[starts[lens > 0]] = False  # (a row's first token has no predecessor)
    prev_tok = np.roll(flat, 1)
    counts = np.where(is_lab, 1, np.where(prev_lab, flat + 1, 0)).astype(np.int64)
    values = np.where(is_lab, flat, prev_tok) - R
    out = np.repeat(values, counts).astype(np.int32)
    rows = np.repeat(np.arange(B), lens)
    out_lens = np.bincount(rows, weights=counts, minlength=B).astype(np.int64)
    return E.split_rows(out, out_lens, torch.int32)


_NODE = False
_PHASES = ("lattice_gather", "lattice_chain", "lattice_grad", "dense_chain", "dense_grad")


def _native_node():
    """csrc/torch_ops.cpp (the step's launches in one native call), or None if the extension was not built /
    WFL_ASG_NATIVE=0 (A/B, tests: the Python spelling of the same sequence below)."""
    global _NODE
    if _NODE is False:
        _NODE = None
        if os.environ.get("WFL_ASG_NATIVE", "1") != "0":
            try:
                from .. import _wfl_torch as mod
                _NODE = mod if hasattr(mod, "asg_forward") else None
            except ImportError:
                pass
    return _NODE


def max_classes():
    """Largest class count the ASG kernels accept (wfl_dense_max_classes: an index-width bound, 16384 -- up to
    wfl_dense_on_chip_classes() the transition matrix is private to a workgroup, beyond it is streamed from L2 by the
    batched per-frame product of csrc/dense_wide.h).  The reference has no limit (asg.py:198-199)."""
    from .. import _native as N

    return int(N.lib.wfl_dense_max_classes())


_EARLY_GRAD = os.environ.get("WFL_ASG_EARLY_GRAD", "1") != "0"  # (0: the gradient kernels in backward -- A/B, tests)


class _EarlyGrads:
    """The gradients the forward launches wrote for grad_output = 1, until backward claims them."""

    __slots__ = ("dx", "dW", "inputs", "transitions", "taken")

    def __init__(self, dx, dW, inputs, transitions):
        self.dx, self.dW, self.inputs, self.transitions, self.taken = dx, dW, inputs, transitions, False

    def fresh(self):
        return not self.taken

    def claim(self):
        self.taken = True
        dx, dW, self.dx, self.dW = self.dx, self.dW, None, None
        return dx, dW

    def take(self):
        """E.EagerLoss.backward: the buffers become the leaves' .grad"""
        if self.taken or not all(E.takes_grad(t, g) for t, g in ((self.inputs, self.dx), (self.transitions, self.dW))
                                 if g is not None):
            return None
        dx, dW = self.claim()
        return [(t, g) for t, g in ((self.inputs, dx), (self.transitions, dW)) if g is not None]


class PackedNumerator:
    """A numerator lattice packed elsewhere, in the place of ASGLoss's `targets`: acceptors whose learnable arc weights
    index `transitions` row-major, with the loss factors (scale, +scale/B, -scale/B) that travel with them.  What the
    Transducer with the bigram transition model hands over (criterions/transducer.py::_bigram_route): its loss IS
    forward_score(emissions o transitions) - forward_score(emissions o alignments o transitions), the ASG step."""

    __slots__ = ("pack", "scale", "cpos", "cneg", "B")

    def __init__(self, pack, scale, cpos, cneg, B):
        self.pack, self.scale, self.cpos, self.cneg, self.B = pack, scale, cpos, cneg, B


class ASGLossFunction(torch.autograd.Function):
    @staticmethod
    def create_transitions_graph(transitions, calc_grad=False):
        """asg.py:54-69 as a host graph: arc ids are the row-major index into transitions
        [(C+1), C] (row 0: start -> i, row 1+i: j -> i).  Used when a Transducer is given ASG
        transitions (tests/transducer_test.py:474-481); the ASG loss itself never builds it."""
        import numpy as np

        C = transitions.shape[1]
        assert transitions.shape == (C + 1, C)
        g = G.Graph(calc_grad)
        g.add_nodes([1] + [0] * C, [0] + [1] * C)
        i = np.repeat(np.arange(C, dtype=np.int32), C)
        j = np.tile(np.arange(C, dtype=np.int32), C)
        src = np.concatenate([np.zeros(C, np.int32), j + 1])
        dst = np.concatenate([np.arange(1, C + 1, dtype=np.int32), i + 1])
        lab = np.concatenate([np.arange(C, dtype=np.int32), i])
        g.add_arcs(src, dst, lab, lab, transitions.detach().cpu().contiguous().view(-1).numpy())
        return g

    @staticmethod
    def create_force_align_graph(target):
        """asg.py:72-81 as a host graph (API parity)."""
        g = G.Graph(False)
        g.add_node(True)
        L = len(target)
        for l in range(1, L + 1):
            g.add_node(False, l == L)
            g.add_arc(l - 1, l, target[l - 1])
            g.add_arc(l, l, target[l - 1])
        g.arc_sort(True)
        return g

    @staticmethod
    @E.on_input_device
    def forward(ctx, inputs, transitions, targets, reduction="none"):
        B, T, C = inputs.shape
        if reduction not in ("none", "mean"):  # asg.py:120-121
            raise ValueError("invalid value for reduction '" + str(reduction) + "'")
        if tuple(transitions.shape) != (C + 1, C):
            raise ValueError(f"transitions must be [{C + 1}, {C}], got {tuple(transitions.shape)}")
        if T == 0:
            raise ValueError("ASGLoss: empty emissions (T == 0)")
        dev = E.require_gpu()
        x = E.as_device_f32(inputs.detach(), dev)
        W = E.as_device_f32(transitions.detach(), dev)
        if isinstance(targets, PackedNumerator):
            if targets.B != B:
                raise ValueError(f"got {targets.B} targets for a batch of {B}")
            tg, pack, scale, cpos, cneg = None, targets.pack, targets.scale, targets.cpos, targets.cneg
        else:
            tg = E.targets_on_device(targets, dev)
            if tg.B != B:
                raise ValueError(f"got {tg.B} targets for a batch of {B}")
            scale, cpos, cneg = E.loss_factors(tg, reduction)
            pack = tg.cache.get(("asg_fal", C))
            if pack is None:
                pack = tg.cache[("asg_fal", C)] = E.PackedLattice.asg_force_align(tg.flat, tg.offsets, C, dev)
        need_dx, need_dw = inputs.requires_grad, transitions.requires_grad
        need_grad = need_dx or need_dw
        node = _native_node()
        timed = node is not None and E.phase_due(_PHASES)  # (a step whose launch groups bench.py brackets with events)
        if node is not None and not timed and not torch.cuda.is_current_stream_capturing():
            # every launch below in one native call (csrc/torch_ops.cpp::asg_forward): the same sequence, ~60 us of host
            # time instead of ~200 (what a step costs at a training batch of 8, where the kernels take less than that)
            early = need_grad and _EARLY_GRAD and all(E.may_hand_over(t) and t.device == x.device
                                                      for t in (inputs, transitions) if t.requires_grad)
            fork = E.side_stream(dev)
            up = getattr(pack, "_uploaded", None)
            if up is not None and up[0] not in (E.stream_ptr(), fork.side.cuda_stream):
                fork.side.wait_event(up[1])  # (a cached pack uploaded on a third stream)
            loss, da, db, dz, ws, dx_num, dw_num, dx, dW = node.asg_forward(
                x, W, ctypes.addressof(pack.desc), pack.ints, pack.floats, scale, cpos, cneg, need_dx, need_dw, early,
                fork.side.cuda_stream)
            fcc = E.DenseState()
            fcc.B, fcc.T, fcc.C, fcc.alpha, fcc.beta, fcc.logz, fcc.ws = B, T, C, da, db, dz, ws
            ctx.aux = (x, W, fcc, cpos, dx_num, dw_num, fork if need_grad else None)
            ctx.devices = (inputs.device, transitions.device)
            ctx.early = None
            if early:
                ctx.early = _EarlyGrads(dx, dW, inputs, transitions)
                ctx.eager_take = ctx.early.take
                E.watch_node_hooks(ctx)
            return loss if inputs.is_cuda else loss.cpu()
        # numerator (force-aligned lattice) and denominator (fully connected) sweeps are independent and
        # both latency-bound: fork the numerator onto a second stream so that they overlap.  The numerator is
        # the shorter of the two, so its gradient (for grad_output = 1) is computed right behind its sweeps, still
        # under the denominator's; backward adds it, scaled by grad_output, inside the denominator's gradient kernel.
        # (its buffers first, on this stream: the numerator's stream is the longer one, a fill there is a fill on the
        # step's critical path)
        E._PHASE_FORCE = timed
        try:
            return ASGLossFunction._forward_launches(ctx, inputs, transitions, x, W, tg, pack, scale, cpos, cneg, need_dx,
                                                     need_dw, dev)
        finally:
            E._PHASE_FORCE = False

    @staticmethod
    def _forward_launches(ctx, inputs, transitions, x, W, tg, pack, scale, cpos, cneg, need_dx, need_dw, dev):
        """The step's launches, one engine call after the other (asg_forward of csrc/torch_ops.cpp is the same sequence)."""
        need_grad = need_dx or need_dw
        dx_num = torch.empty_like(x) if need_dx else None
        dw_num = torch.zeros_like(W) if need_dw else None
        with E.side_stream(dev) as fork:
            for t in (dx_num, dw_num):
                if t is not None:
                    t.record_stream(fork.side)
            fal = E.lattice_forward(x, pack, weights=W, need_beta=need_grad)
            swept = fork.mark()
            if need_grad:
                E.lattice_grad(fal, cneg, coef_w=cneg, gout=None, dx=dx_num, accumulate=False, dW=dw_num)
        fcc = E.dense_forward(x, W, need_beta=need_grad)
        # the loss only needs the numerator's sweeps; its gradient keeps running and is joined in backward
        fork.join_at(swept, fal.xg, fal.alpha, fal.beta, fal.logz)
        loss = E.reduce_loss(fcc.logz, scale, 1.0, minus=fal.logz)
        ctx.aux = (x, W, fcc, cpos, dx_num, dw_num, fork if need_grad else None)
        ctx.devices = (inputs.device, transitions.device)
        ctx.early = None
        if need_grad and _EARLY_GRAD and all(E.may_hand_over(t) and t.device == x.device
                                             for t in (inputs, transitions) if t.requires_grad):
            # The denominator's gradient right behind its sweeps, for grad_output = 1 (as the numerator's): between the
            # forward and the backward kernels of a step the GPU otherwise waits ~30 us for the host to come back
            # through the autograd engine.  `loss.backward()` takes the two buffers as they are (E.EagerLoss); should
            # the engine run this node after all, backward scales them by grad_output.  Plain leaf tensors (the reference's
            # asg_benchmark.py:19-31 protocol on device tensors) get them as .grad; a model's output, an nn.Parameter
            # (the ASG module, anything under DistributedDataParallel) get them as the root gradients of an engine
            # pass that starts at those tensors (E.EagerLoss.backward).
            dx = torch.empty_like(x) if need_dx else None
            dW = torch.empty_like(W) if need_dw else None
            fork.join(dx_num, dw_num)
            E.dense_grad(x, W, fcc, cpos, coef_w=cpos, gout=None, dx=dx, accumulate=False, dW=dW, addend=dx_num,
                         dW_addend=dw_num)
            ctx.early = _EarlyGrads(dx, dW, inputs, transitions)
            ctx.eager_take = ctx.early.take
            E.watch_node_hooks(ctx)
        return loss if inputs.is_cuda else loss.cpu()

    @staticmethod
    @E.on_input_device
    def backward(ctx, grad_output):
        E.check_not_released(ctx)
        if ctx.early is not None and ctx.early.fresh():
            # the buffers of the forward launches, scaled in place (wfl_scale returns at once when grad_output is 1); a
            # second pass over a retained graph finds them used and recomputes below
            dx, dW = ctx.early.claim()
            gout = E.as_device_f32(grad_output.detach().reshape(1), (dx if dx is not None else dW).device)
            dx = E.scale_inplace(dx, gout) if dx is not None and ctx.needs_input_grad[0] else None
            dW = E.scale_inplace(dW, gout) if dW is not None and ctx.needs_input_grad[1] else None
            if dx is not None and ctx.devices[0].type != "cuda":
                dx = dx.to(ctx.devices[0])
            if dW is not None and ctx.devices[1].type != "cuda":
                dW = dW.to(ctx.devices[1])
            return dx, dW, None, None
        x, W, fcc, cpos, dx_num, dw_num, fork = ctx.aux
        gout = E.as_device_f32(grad_output.detach().reshape(1), x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] and dx_num is not None else None
        dW = torch.empty_like(W) if ctx.needs_input_grad[1] and dw_num is not None else None
        if dx is not None or dW is not None:
            fork.join(dx_num, dw_num)
            # + posteriors of the fully connected graph, - posteriors of the force-aligned one (asg.py:158-168)
            E.dense_grad(x, W, fcc, cpos, coef_w=cpos, gout=gout, dx=dx, accumulate=False, dW=dW,
                         addend=dx_num if dx is not None else None, dW_addend=dw_num if dW is not None else None)
        if dx is not None and ctx.devices[0].type != "cuda":
            dx = dx.to(ctx.devices[0])
        if dW is not None and ctx.devices[1].type != "cuda":
            dW = dW.to(ctx.devices[1])
        return dx, dW, None, None


def ASGLoss(*args):
    """asg.py:183 (`ASGLoss = ASGLossFunction.apply`): same call, same result."""
    return E.make_eager(ASGLossFunction.apply(*args))


class ASG(torch.nn.Module):
    def __init__(self, num_classes, num_replabels=1, use_garbage=True):
        super(ASG, self).__init__()
        self.num_classes = num_classes
        self.num_replabels = num_replabels
        assert self.num_replabels > 0
        self.garbage_idx = (num_classes + num_replabels) if use_garbage else None
        self.N = num_classes + num_replabels + int(use_garbage)
        limit = max_classes()
        if self.N > limit:  # (16384: the (N+1) x N matrix alone would be a gigabyte)
            raise NotImplementedError(
                f"ASG with {self.N} classes (tokens + replabels + garbage): the dense-transition kernels take at most {limit}")
        self.transitions = torch.nn.Parameter(torch.zeros(self.N + 1, self.N))

    @E.on_input_device
    def forward(self, inputs, targets):
        if len(targets) and all(type(t) is torch.Tensor and not t.is_cuda for t in targets):
            # (what train.py hands over: the whole batch as array operations -- row by row in Python this preparation cost
            # more host time than the step's kernels take at a training batch)
            targets = pack_targets_batch(targets, self.num_replabels, self.garbage_idx)
            return ASGLoss(inputs, self.transitions, targets, "mean")
        targets = [pack_replabels(t.tolist(), self.num_replabels) for t in targets]
        if self.garbage_idx is not None:  # a garbage token between (and around) the labels: asg.py:203-208
            for idx, tgt in enumerate(targets):
                interleaved = [self.garbage_idx] * (2 * len(tgt) + 1)
                interleaved[1::2] = tgt
                targets[idx] = interleaved
        return ASGLoss(inputs, self.transitions, targets, "mean")

    def viterbi(self, outputs):
        """asg.py:211-237: best label sequence under emissions + transitions, repeats collapsed,
        garbage dropped, replabels unpacked."""
        B, T, C = outputs.shape
        assert C == self.N, "Wrong number of classes in output."
        dev = E.require_gpu()
        x = E.as_device_f32(outputs.detach(), dev)
        W = E.as_device_f32(self.transitions.detach(), dev)
        return collapse_and_unpack(E.dense_viterbi(x, W), self.garbage_idx, self.num_replabels)
