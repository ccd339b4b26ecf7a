"""Device orchestration for the WFST loss engine: owns the torch-allocated buffers, the packed
lattices and the calls into libwfl.so.  torch is plumbing here (device memory + current stream);
all arithmetic happens in the HIP kernels behind the C ABI (include/wfl.h).
"""
import collections
import ctypes
import os
import itertools
import threading
from collections import OrderedDict

import numpy as np
import torch

from . import _native as N

_F32 = torch.float32


_HAVE_GPU = None
_DEVICES = {}


def require_gpu():
    global _HAVE_GPU
    if _HAVE_GPU is None:  # (a visible GPU does not go away within a process)
        _HAVE_GPU = bool(torch.cuda.is_available())
    if not _HAVE_GPU:
        raise RuntimeError(
            "gtn_applications_amd: no ROCm GPU visible. The criteria run on HIP kernels only; "
            "there is deliberately no CPU fallback."
        )
    idx = torch.cuda.current_device()
    dev = _DEVICES.get(idx)
    if dev is None:
        dev = _DEVICES[idx] = torch.device("cuda", idx)
    return dev


def stream_ptr():
    # raw handle of torch's current stream on the current device (what current_stream().cuda_stream
    # returns, without building the Stream object: this sits on every launch).  Plain ints: ctypes converts them
    # for c_void_p parameters without a Python-level c_void_p object per argument.
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()) or None


def ptr(t):
    """device pointer argument: a tensor, a raw address (int) or None"""
    if t is None:
        return None
    return t if type(t) is int else t.data_ptr()


def as_device_f32(t, device):
    """float32, contiguous, on `device` (the reference reads raw float32 too: ctc.py:43-44)."""
    if t.dtype != _F32:
        raise TypeError(f"expected a float32 tensor, got {t.dtype}")
    if t.device != device:
        t = t.to(device)
    return t.contiguous()


def flatten_targets(targets):
    """list of int sequences (or 1-D tensors) -> (flat int32, offsets int64, lengths list)."""
    rows = [t.tolist() if hasattr(t, "tolist") else list(t) for t in targets]
    lens = [len(r) for r in rows]
    flat = np.fromiter(itertools.chain.from_iterable(rows), dtype=np.int32, count=sum(lens))
    offsets = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    return flat, offsets, lens


# bench.py sets this to a list to time the native launches of a step with HIP events recorded on the stream
# each launch goes to: entries are (phase name, start event, end event).  None (the default) costs one test.
PHASE_EVENTS = None
PHASE_ONLY = None  # optional set of phase names: the only ones recorded (bench.py's timed region records one)
PHASE_STRIDE = 1   # record every PHASE_STRIDE-th launch of a phase (an event pair costs the step ~10 us)
_PHASE_COUNT = {}


_EVENT_POOLS = {}  # device index -> timing events free for reuse (an event belongs to the device it was first recorded on)


def _event_pool():
    idx = torch.cuda.current_device()
    pool = _EVENT_POOLS.get(idx)
    if pool is None:
        pool = _EVENT_POOLS[idx] = []
    return pool


def prealloc_events(n):
    """bench.py: create the timing events of a timed region up front (hipEventCreate costs more than the record)"""
    pool = _event_pool()
    while len(pool) < n:
        pool.append(torch.cuda.Event(enable_timing=True))


def _event():
    pool = _event_pool()
    ev = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def on_input_device(fn):
    """Run a criterion entry point with the device of its first CUDA tensor argument current (the reference's
    criteria return results on the inputs' device whatever the current device is: ctc.py:69, asg.py:139).  Every
    per-device resource of the engine -- staging rings, workspaces, streams, timing events -- is looked up through
    the current device, so the switch is all it takes; nothing is done when it already is the current one."""
    def wrapped(*args, **kwargs):
        for a in args:
            if type(a) is torch.Tensor or isinstance(a, torch.Tensor):
                if a.is_cuda and a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)

    wrapped.__name__, wrapped.__doc__ = getattr(fn, "__name__", "wrapped"), fn.__doc__
    return wrapped


_PHASE_FORCE = False


def phase_due(names):
    """Will a launch group of `names` be bracketed in THIS step?  For operator paths that issue all their launches in
    one native call (criterions/asg.py): they take the Python spelling of the same sequence in the steps whose groups
    are being timed -- under PHASE_STRIDE every PHASE_STRIDE-th step -- with `_PHASE_FORCE` set around it."""
    if PHASE_EVENTS is None or (PHASE_ONLY is not None and not any(n in PHASE_ONLY for n in names)):
        return False
    if PHASE_STRIDE > 1:
        key = "step:" + names[0]
        k = _PHASE_COUNT[key] = _PHASE_COUNT.get(key, 0) + 1
        if k % PHASE_STRIDE:
            return False
    return True


def _mark(name):
    if PHASE_EVENTS is None or (PHASE_ONLY is not None and name not in PHASE_ONLY):
        return None
    if PHASE_STRIDE > 1 and not _PHASE_FORCE:
        k = _PHASE_COUNT[name] = _PHASE_COUNT.get(name, 0) + 1
        if k % PHASE_STRIDE:
            return None
    return name, _event()


def _done(tok):
    if tok is not None:
        PHASE_EVENTS.append((tok[0], tok[1], _event()))


_WFLPY = None


def _staging_helper():
    """the CPython helper of the target staging (csrc/wflpy.c), or its numpy stand-in when the extension is missing"""
    global _WFLPY
    if _WFLPY is None:
        try:
            from . import _wflpy as mod
        except ImportError:
            from . import _wflpy_np as mod
        _WFLPY = mod
    return _WFLPY


def flatten_any(targets):
    """Targets as the criteria receive them -- a list of int sequences or of 1-D LongTensors (train.py hands over
    tensors, the benchmarks lists) -> (flat int32 array, offsets int64 [B+1], lengths).  Tensors are flattened
    with one torch.cat instead of B tolist() calls."""
    if len(targets) and all(type(t) is torch.Tensor and t.dim() == 1 and not t.is_cuda for t in targets):
        lens = [t.numel() for t in targets]
        flat = torch.cat(targets).to(torch.int32).numpy() if sum(lens) else np.zeros(0, np.int32)
        offsets = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        return flat, offsets, lens
    return flatten_targets(targets)


class LRU:
    def __init__(self, capacity=32):
        self.capacity = capacity
        self.data = OrderedDict()

    def get(self, key, make):
        hit = self.data.get(key)
        if hit is not None:
            self.data.move_to_end(key)
            return hit
        val = make()
        self.data[key] = val
        if len(self.data) > self.capacity:
            self.data.popitem(last=False)
        return val


class PackedLattice:
    """B acceptors in the flat device format of `wfl_lattice_desc`, resident in HBM."""

    def __init__(self, host_handle, device, extra=None, staged=None):
        """`extra`: optional float32 host array uploaded behind the float blob in the same copy (per-utterance loss
        factors); `self.extra` is its device view.  `staged` = (slot, buf, view): the packer already wrote the blobs
        into that staging slot (wfl_transducer_pack_batch_into)."""
        N.check_handle(host_handle)
        cuda = device is not None and device.type == "cuda"
        try:
            d = N.lib.wfl_lattice_host_desc(host_handle).contents
            self.desc = N.LatticeDesc.from_buffer_copy(d)
            ni, nf = int(d.int_words), int(d.float_words)
            ne = 0 if extra is None else int(extra.size)
            off_i = (4 * (nf + ne) + 15) & ~15  # [floats | extra | pad to 16 B | ints]: ONE buffer, one upload
            nbytes = off_i + 4 * max(ni, 1)
            external = staged is not None and N.lib.wfl_lattice_host_external(host_handle) == off_i
            if external:
                slot, buf, view = staged
                ring = _lattice_ring(device)
            elif cuda:
                # through a ring of reusable PINNED buffers and one asynchronous copy: a pageable copy is synchronous
                # (it would wait for everything queued on the stream before it -- the previous step's kernels), and
                # a fresh pinned allocation per batch costs a hipHostMalloc
                ring = _lattice_ring(device)
                if staged is not None:
                    ring.i -= 1  # (the slot offered to the packer was too small: take it again, grown)
                slot, buf, view = ring.next(nbytes, True)
            else:
                buf = torch.empty(nbytes, dtype=torch.uint8)
                view = buf.numpy()
            if nf and not external:
                ctypes.memmove(buf.data_ptr(), N.lib.wfl_lattice_host_floats(host_handle), 4 * nf)
            if ne:
                view[4 * nf:4 * (nf + ne)].view(np.float32)[:] = np.asarray(extra, dtype=np.float32).reshape(-1)
            if ni and not external:
                ctypes.memmove(buf.data_ptr() + off_i, N.lib.wfl_lattice_host_ints(host_handle), 4 * ni)
        finally:
            N.lib.wfl_lattice_host_free(host_handle)
        self.device = device
        self._host = None
        self._uploaded = None
        if cuda:
            blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
            upload(blob, buf, nbytes)
            ev = ring.events[slot]
            if ev is None:
                ev = ring.events[slot] = torch.cuda.Event()
            ev.record()  # (the staging slot: reusable once this upload has read it)
            # the lattice's OWN event: packs are cached and may be used from another stream later, by which time the
            # slot's event may have been re-recorded on some other stream (lattice_forward orders itself behind this one)
            own = torch.cuda.Event()
            own.record()
            self._uploaded = (stream_ptr(), own)
        elif device is not None:
            blob = buf.to(device)
        else:
            blob = None
            self._host = (view[:4 * nf].view(np.float32).copy(), view[off_i:off_i + 4 * ni].view(np.int32).copy())
        self._blob = blob
        self.floats = blob[:4 * (nf + ne)].view(_F32) if blob is not None else None
        self.ints = blob[off_i:off_i + 4 * max(ni, 1)].view(torch.int32) if blob is not None else None
        self.extra = self.floats[nf:nf + ne] if (ne and blob is not None) else None
        self._n = (ni, nf)
        self._desc_ref = ctypes.byref(self.desc)

    # host copies of the blobs (tests, diagnostics): kept when there is no device, fetched back otherwise
    @property
    def host_ints(self):
        return self._host[1] if self._host is not None else self.ints[:self._n[0]].cpu().numpy()

    @property
    def host_floats(self):
        return self._host[0] if self._host is not None else self.floats[:self._n[1]].cpu().numpy()

    # -- constructors ---------------------------------------------------------------------------
    @classmethod
    def from_graphs(cls, graphs, C, device, wids=None, B=None, shared=False):
        n = len(graphs)
        B = n if B is None else B
        handles = (ctypes.c_void_p * n)(*[g._h for g in graphs])
        keep = []
        wid_ptrs = None
        if wids is not None:
            arr = []
            for g, w in zip(graphs, wids):
                if w is None:
                    arr.append(None)
                    continue
                w = np.ascontiguousarray(w, dtype=np.int32)
                if w.size != g.num_arcs():
                    raise ValueError("weight-id array must have one entry per arc")
                keep.append(w)
                arr.append(w.ctypes.data)
            wid_ptrs = (ctypes.c_void_p * n)(*arr)
        h = N.lib.wfl_lattice_pack(handles, wid_ptrs, n, B, int(bool(shared)), int(C))
        return cls(h, device)

    @classmethod
    def transducer_batch(cls, tokens, lexicon, transitions, flat, offsets, C, device, nthreads=0, extra=None):
        """Alignment acceptors of a whole batch, built and packed on the library's host thread pool
        (wfl_transducer_pack_batch: transducer.py:262-281 under gtn.parallel_for)."""
        flat = np.ascontiguousarray(flat, dtype=np.int32)
        if flat.size == 0:
            flat = np.zeros(1, np.int32)
        tr = None if transitions is None else transitions._h
        staged = None
        if device is not None and device.type == "cuda":
            # the packer writes the blobs straight into the pinned staging slot they are uploaded from
            ring = _lattice_ring(device)
            staged = ring.next(ring.nbytes, True)
            h = N.lib.wfl_transducer_pack_batch_into(tokens._h, lexicon._h, tr, flat.ctypes.data, offsets.ctypes.data,
                                                     len(offsets) - 1, int(C), int(nthreads), staged[1].data_ptr(),
                                                     staged[1].numel(), 0 if extra is None else int(extra.size))
        else:
            h = N.lib.wfl_transducer_pack_batch(tokens._h, lexicon._h, tr, flat.ctypes.data, offsets.ctypes.data,
                                                len(offsets) - 1, int(C), int(nthreads))
        return cls(h, device, extra, staged)

    @classmethod
    def ctc(cls, flat, offsets, blank, C, device):
        return cls(N.lib.wfl_lattice_pack_ctc(flat.ctypes.data, offsets.ctypes.data, len(offsets) - 1, blank, C), device)

    @classmethod
    def asg_force_align(cls, flat, offsets, C, device):
        return cls(N.lib.wfl_lattice_pack_asg_fal(flat.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C), device)

    @classmethod
    def stc(cls, flat, offsets, star_idx, log_prob, C, device):
        return cls(
            N.lib.wfl_lattice_pack_stc(flat.ctypes.data, offsets.ctypes.data, len(offsets) - 1, star_idx, log_prob, C),
            device,
        )

    # -- host-side views (tests, Viterbi post-processing) -----------------------------------------
    def field(self, name, count):
        off = getattr(self.desc, name)
        src = self.host_floats if name in ("arc_w", "eps_w", "start_w", "accept_w") else self.host_ints
        return src[off:off + count]


class LatticeState:
    """Everything the backward pass needs from a forward pass of the lattice engine."""

    __slots__ = ("pack", "T", "C", "xg", "alpha", "beta", "logz", "weights", "bptr", "x", "row_lse", "in_launch")


def lattice_diagnostics():
    """Counters of the gradient beside the sweeps (wfl_lattice_diagnostics): how often it was launched, how often its
    gate gave up (the call then fell back to the plain gradient for every row), the back-off state."""
    out = (ctypes.c_uint64 * 8)()
    N.check(N.lib.wfl_lattice_diagnostics(out, 8))
    names = ("launched", "gate_gave_up", "gate_ok", "skipped_in_backoff", "backoff_left", "env_serialised", "gate_spins", "fork")
    return dict(zip(names, (int(v) for v in out)))


def lattice_side_join():
    """The current stream waits for the gradient workgroups that ran beside the sweeps (lattice_forward with
    defer_join=True)."""
    N.check(N.lib.wfl_lattice_side_join(stream_ptr()))


def lattice_forward(x, pack, weights=None, need_beta=True, semiring=N.SEMIRING_LOG, log_softmax=False, grad_into=None,
                    defer_join=False):
    """forward_score(intersect(emissions, A_b)) for every b: returns a LatticeState whose `logz`
    holds the per-utterance score (gtn call sites: ctc.py:50, asg.py:111, stc.py:86,
    transducer.py:283,287).

    grad_into = (coef, dx): ask the sweeps' launch to compute the emission gradient for grad_output = 1 as well
    (wfl_lattice_forward_grad); `st.in_launch` says whether it did -- lattice_grad_rest finishes the job.
    defer_join=True: if it did, the caller MUST call lattice_side_join() before anything else touches dx, alpha or
    beta (it queues the loss reduction first, which then runs under the gradient's tail)."""
    B, T, C = x.shape
    d = pack.desc
    up = getattr(pack, "_uploaded", None)
    if up is not None and up[0] != stream_ptr():  # a cached pack uploaded on another stream
        torch.cuda.current_stream().wait_event(up[1])
    if d.B != B:
        raise ValueError(f"lattice batch has {d.B} utterances, emissions have {B}")
    n_xg, n_ab = ctypes.c_int64(), ctypes.c_int64()
    N.check(N.lib.wfl_lattice_workspace(pack._desc_ref, T, ctypes.byref(n_xg), ctypes.byref(n_ab)))
    dev = x.device
    st = LatticeState()
    st.pack, st.T, st.C, st.weights = pack, T, C, weights
    st.xg = torch.empty(max(n_xg.value, 1), dtype=_F32, device=dev)
    st.alpha = torch.empty(max(n_ab.value, 1), dtype=_F32, device=dev)
    tropical = semiring == N.SEMIRING_TROPICAL
    st.beta = torch.empty(max(n_ab.value, 1), dtype=_F32, device=dev) if (need_beta and not tropical) else None
    st.bptr = torch.empty(max(n_ab.value, 1), dtype=torch.int32, device=dev) if tropical else None
    st.logz = torch.empty(B, dtype=_F32, device=dev)
    # log_softmax=True: x holds raw scores; the gather subtracts each row's log-sum-exp and the
    # gradient kernel differentiates through it (ctc.py:107, transducer.py:186-187 fused into the path)
    st.x = x if log_softmax else None
    st.row_lse = torch.empty((B, T), dtype=_F32, device=dev) if log_softmax else None
    s = stream_ptr()
    tag = "/shared" if d.shared else ""
    tok = _mark("lattice_gather" + tag)
    N.check(N.lib.wfl_lattice_gather(pack._desc_ref, ptr(pack.ints), ptr(x), T, C, ptr(st.xg), ptr(st.row_lse), s))
    _done(tok)
    tok = _mark("lattice_chain" + tag)
    st.in_launch = False
    if grad_into is not None and st.beta is not None:
        coef, dx = grad_into
        flag = ctypes.c_int(2 if defer_join else 0)
        N.check(
            N.lib.wfl_lattice_forward_grad(
                pack._desc_ref, ptr(pack.ints), ptr(pack.floats), ptr(st.xg), T, C, ptr(weights), ptr(st.alpha),
                ptr(st.beta), ptr(st.logz), ptr(coef), ptr(st.x), ptr(st.row_lse), ptr(dx), ctypes.byref(flag), s,
            )
        )
        st.in_launch = bool(flag.value)
    else:
        N.check(
            N.lib.wfl_lattice_forward(
                pack._desc_ref, ptr(pack.ints), ptr(pack.floats), ptr(st.xg), T, ptr(weights), semiring,
                ptr(st.alpha), ptr(st.beta), ptr(st.bptr), ptr(st.logz), s,
            )
        )
    _done(tok)
    return st


def lattice_formats(st):
    """int32 [B] (device): how each utterance of a log-semiring forward pass was swept -- 1: fp64 probability domain
    (2: its sweeps met in the middle and left occupancies), 0: log domain (the certificate's repair: the two
    probability-domain sweeps disagreed about Z; an acceptor outside the lean sweeps' shape in the launch that has the
    gradient beside the sweeps; WFL_LATTICE_DOMAIN=log).  Diagnostics and tests (wfl_lattice_formats_offset)."""
    B = st.pack.desc.B
    pos = ctypes.c_int64()
    N.check(N.lib.wfl_lattice_formats_offset(st.pack._desc_ref, st.T, ctypes.byref(pos)))
    off = pos.value
    return st.alpha[off:off + B].view(torch.int32)


def lattice_grad(st, coef, coef_w=None, gout=None, dx=None, accumulate=False, dW=None):
    """Posteriors of the lattice -> dense emission gradient rows (and learnable-weight grads)."""
    p = st.pack
    tok = _mark("lattice_grad" + ("/shared" if p.desc.shared else ""))
    N.check(
        N.lib.wfl_lattice_grad(
            p._desc_ref, ptr(p.ints), ptr(p.floats), ptr(st.xg), st.T, st.C, ptr(st.weights), ptr(st.alpha),
            ptr(st.beta), ptr(st.logz), ptr(coef), ptr(coef_w), ptr(gout), int(bool(accumulate)), ptr(st.x),
            ptr(st.row_lse), ptr(dx), ptr(dW), stream_ptr(),
        )
    )
    _done(tok)


def lattice_grad_rest(st, coef, gout, dx):
    """After a forward pass with st.in_launch: dx (already scaled by grad_output) gets the rows of the utterances the
    launch did not serve (wfl_lattice_grad_rest)."""
    p = st.pack
    tok = _mark("lattice_grad" + ("/shared" if p.desc.shared else ""))
    N.check(
        N.lib.wfl_lattice_grad_rest(
            p._desc_ref, ptr(p.ints), ptr(p.floats), ptr(st.xg), st.T, st.C, ptr(st.weights), ptr(st.alpha),
            ptr(st.beta), ptr(st.logz), ptr(coef), ptr(gout), ptr(st.x), ptr(st.row_lse), ptr(dx), stream_ptr(),
        )
    )
    _done(tok)


def lattice_viterbi(x, pack, weights=None):
    """viterbi_path(intersect(emissions, A_b)): list of numpy arrays of the caller's arc ids along
    the best path of each utterance (None if no accepting path), plus the best scores tensor."""
    B, T, C = x.shape
    st = lattice_forward(x, pack, weights, need_beta=False, semiring=N.SEMIRING_TROPICAL)
    stride = (T + 1) * max(1, pack.desc.max_levels) + 1
    path = torch.empty((B, stride), dtype=torch.int32, device=x.device)
    plen = torch.empty(B, dtype=torch.int32, device=x.device)
    N.check(
        N.lib.wfl_lattice_backtrace(
            pack._desc_ref, ptr(pack.ints), ptr(pack.floats), ptr(st.alpha), ptr(st.bptr), T, ptr(path), ptr(plen),
            stride, stream_ptr(),
        )
    )
    path, plen = path.cpu().numpy(), plen.cpu().numpy()
    return [None if plen[b] < 0 else path[b, :plen[b]].copy() for b in range(B)], st.logz


_SIDE_STREAMS = {}
_ORDER = False


def _order_module():
    """csrc/torch_ops.cpp's stream-ordering entry points (a ring of device-scope events), or None if it is not built."""
    global _ORDER
    if _ORDER is False:
        try:
            from . import _wfl_torch as mod
            _ORDER = mod if hasattr(mod, "order_mark") else None
        except ImportError:
            _ORDER = None
    return _ORDER


def _order_after(waiter, signaller):
    """`waiter`'s later work after `signaller`'s earlier work.  torch's Stream.wait_stream creates and records an event
    per call; between the criteria's forked streams that record held the host for ~0.4 ms a call whenever the GPU had
    work queued (a Transducer step with a back-off model: 1.17 ms, 1.1 of them host, against 0.41 with the ring of
    device-scope events csrc/torch_ops.cpp keeps: order_event_flags) -- the host could not run ahead of the GPU."""
    mod = _order_module()
    if mod is None:
        waiter.wait_stream(signaller)
    else:
        mod.order_after(waiter.cuda_stream, signaller.cuda_stream, waiter.device.index)


class side_stream:
    """Run a block of launches on a second HIP stream that forks from / joins the current one: two
    independent latency-bound sweeps (ASG numerator and denominator) then share the GPU instead of
    queueing behind each other.  Tensors allocated inside must be passed to `keep()` so that the
    caching allocator knows the current stream uses them later."""

    def __init__(self, device):
        self.cur = torch.cuda.current_stream(device)
        key = device.index
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device)
        self.side = _SIDE_STREAMS[key]
        self.ctx = None

    def __enter__(self):
        _order_after(self.side, self.cur)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        """The current stream waits for everything launched on the side stream so far."""
        cur = torch.cuda.current_stream(self.side.device)
        _order_after(cur, self.side)
        for t in tensors:
            if t is not None:
                t.record_stream(cur)

    def mark(self):
        """Event after what has been launched on the side stream so far (call inside the block)."""
        mod = _order_module()
        if mod is not None:
            return mod.order_mark(self.side.cuda_stream, self.side.device.index)
        ev = torch.cuda.Event()
        ev.record(self.side)
        return ev

    def join_at(self, ev, *tensors):
        """The current stream waits for the side stream only up to `ev` (work launched after it keeps overlapping)."""
        if isinstance(ev, int):
            _order_module().order_wait(self.cur.cuda_stream, ev)
        else:
            self.cur.wait_event(ev)
        for t in tensors:
            if t is not None:
                t.record_stream(self.cur)


class EagerLoss(torch.Tensor):
    """A scalar loss whose gradients the forward launches have already computed (for grad_output = 1).  A plain tensor
    in every respect but one: `loss.backward()` with no arguments -- the call of the reference's benchmarks
    (transducer_benchmark.py:47-49) and of a training loop that uses the criterion's output as its loss -- hands those
    buffers to the leaves' .grad without a trip through the autograd engine (its two thread hand-overs, the ones_like
    fill and the scale passes over [B, T, C] leave the GPU idle between the forward and the backward kernels) -- or,
    when an input is somebody's output, a parameter or a hooked leaf, starts the engine AT the inputs with those
    buffers as the root gradients (no ones_like, no trip through the criterion's node, no scale launches).  The
    autograd node offers them through `eager_take()` -> [(tensor, grad), ...] or None.  Anything else -- a gradient argument, retain_graph, create_graph, inputs=, anomaly mode, the loss
    inside a larger expression -- goes through torch.Tensor.backward / the engine, where the node scales the buffers
    by grad_output (in place, a launch that returns at once when grad_output is 1; a second pass over a retained graph
    recomputes)."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        node = self.grad_fn
        take = getattr(node, "eager_take", None)
        # (hooks on the loss itself -- register_hook, retain_grad -- and on its node only run inside the engine)
        if (take is not None and gradient is None and not retain_graph and not create_graph and inputs is None
                and self.dim() == 0 and not torch.is_anomaly_enabled() and not self._backward_hooks
                and not self.retains_grad and not getattr(node, "_wfl_hooked", False)):
            pairs = take()
            if pairs is not None:
                # the graph is spent, as after a pass of the engine without retain_graph: its buffers go, and a second
                # backward() raises instead of accumulating a recomputed gradient
                release_node(node)
                if all(plain_leaf(t) for t, _ in pairs):
                    for leaf, g in pairs:
                        if leaf.grad is None:
                            leaf.grad = g
                        else:
                            leaf.grad.add_(g)
                else:
                    # somebody's output (a model's, train.py:262-266), an nn.Parameter (DistributedDataParallel's
                    # reducer hangs on its AccumulateGrad node), a leaf with hooks: the autograd engine, started AT
                    # these tensors with the gradients the forward launches computed -- everything below them runs as
                    # under torch.Tensor.backward, only this node, the ones_like fill in front of it and its scale
                    # launches do not
                    torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])
                return None
        return torch.Tensor.backward(self, gradient, retain_graph, create_graph, inputs=inputs)


def watch_node_hooks(ctx):
    """forward() of a criterion that offers `eager_take`: remember (ctx._wfl_hooked) whether somebody registers a hook on
    the autograd node -- the engine runs those, EagerLoss.backward does not, so it must stand back.  The node offers no
    way to ask afterwards; the wrappers hold the node weakly (no cycle that would keep the forward's buffers alive)."""
    import weakref

    ref = weakref.ref(ctx)
    ctx._wfl_hooked = False
    for name in ("register_hook", "register_prehook"):
        orig = getattr(type(ctx), name)

        def wrapped(fn, _orig=orig):
            node = ref()
            node._wfl_hooked = True
            return _orig(node, fn)

        setattr(ctx, name, wrapped)


def release_node(ctx):
    """After EagerLoss.backward handed the forward's gradients over: drop what backward would recompute from."""
    ctx._wfl_freed = True
    ctx.aux = None
    ctx.early = None
    ctx.eager_take = None


def check_not_released(ctx):
    """first line of backward(): the error torch raises for a second pass over a freed graph"""
    if getattr(ctx, "_wfl_freed", False):
        raise RuntimeError(
            "Trying to backward through the graph a second time (or directly access saved tensors after they have "
            "already been freed). Saved intermediate values of the graph are freed when you call .backward() or "
            "autograd.grad(). Specify retain_graph=True if you need to backward through the graph a second time.")


def plain_leaf(t):
    """May EagerLoss.backward write t.grad itself?  A plain leaf tensor that requires grad, on the device, without
    hooks -- and NOT an nn.Parameter: DistributedDataParallel (train.py:205-208 wraps criteria that have parameters)
    hangs its gradient all-reduce on the parameter's AccumulateGrad node, which only the autograd engine runs."""
    return (type(t) is torch.Tensor and t.is_leaf and t.requires_grad and t.is_cuda
            and not t._backward_hooks and not getattr(t, "_post_accumulate_grad_hooks", None))


def may_hand_over(t):
    """May EagerLoss.backward pass a gradient of t on itself -- to .grad if plain_leaf(t), else by starting the autograd
    engine at t?  Any float32 device tensor that requires grad (a model's output, an nn.Parameter, a leaf with hooks)."""
    return isinstance(t, torch.Tensor) and t.requires_grad and t.is_cuda and t.dtype == torch.float32


def takes_grad(t, g):
    """may_hand_over(t), and the gradient buffer g lives where t does (ASG transitions may sit on another GPU than the
    inputs: the engine running the criterion's node copies across; neither `t.grad = g` nor a root gradient may)"""
    return may_hand_over(t) and g.device == t.device and g.shape == t.shape and g.dtype == t.dtype


def make_eager(loss):
    """-> loss, as an EagerLoss if its autograd node offers `eager_take`"""
    if type(loss) is torch.Tensor and getattr(loss.grad_fn, "eager_take", None) is not None:
        loss.__class__ = EagerLoss
    return loss


def scale_inplace(v, s):
    """v *= s[0] on the device without a host sync (skipped by the kernel when s[0] == 1)."""
    N.check(N.lib.wfl_scale(ptr(v), v.numel(), ptr(s), stream_ptr()))
    return v


def reduce_loss(vals, scale, sign=1.0, out=None, minus=None):
    """out = (1/B) sum_b sign * scale[b] * (vals[b] - minus[b])   (ctc.py:68-69 and twins), on the device."""
    B = vals.numel()
    accumulate = out is not None
    if out is None:
        out = torch.empty((), dtype=_F32, device=vals.device)
    N.check(N.lib.wfl_reduce_loss(ptr(vals), ptr(minus), ptr(scale), B, float(sign), int(accumulate), ptr(out),
                                  stream_ptr()))
    return out


# -------------------------------------------------------------------------------------------------
# dense transitions
# -------------------------------------------------------------------------------------------------
class DenseState:
    __slots__ = ("alpha", "beta", "logz", "ws", "B", "T", "C")


def _dense_sizes(B, T, C):
    n, w = ctypes.c_int64(), ctypes.c_int64()
    N.check(N.lib.wfl_dense_workspace(B, T, C, ctypes.byref(n), ctypes.byref(w)))
    return n.value, w.value


def dense_forward(x, W, need_beta=True):
    B, T, C = x.shape
    st = DenseState()
    st.B, st.T, st.C = B, T, C
    st.alpha = torch.empty((B, T, C), dtype=_F32, device=x.device)
    st.beta = torch.empty((B, T, C), dtype=_F32, device=x.device) if need_beta else None
    st.logz = torch.empty(B, dtype=_F32, device=x.device)
    st.ws = torch.empty(_dense_sizes(B, T, C)[1], dtype=torch.uint8, device=x.device)
    tok = _mark("dense_chain")
    N.check(
        N.lib.wfl_dense_forward(ptr(x), ptr(W), B, T, C, N.SEMIRING_LOG, ptr(st.alpha), ptr(st.beta), None,
                                ptr(st.logz), ptr(st.ws), stream_ptr())
    )
    _done(tok)
    return st


def dense_flagged(st):
    """[B] bool: utterances the probability-domain sweep handed to the log-domain kernels (none beyond the on-chip class
    count: there the frames of the whole batch are one product per frame, csrc/dense_wide.h, with no second arithmetic).
    (the workspace's layout is the library's: wfl_dense_workspace_field)"""
    B, T = st.B, st.T
    if st.C > N.lib.wfl_dense_on_chip_classes():
        return torch.zeros(B, dtype=torch.bool, device=st.ws.device)
    off, n = ctypes.c_int64(), ctypes.c_int64()
    N.check(N.lib.wfl_dense_workspace_field(B, T, N.DENSE_WS_FLAGS, ctypes.byref(off), ctypes.byref(n)))
    return st.ws[off.value:off.value + n.value].view(torch.int32).view(B, 2).ne(0).any(dim=1)


def dense_grad(x, W, st, coef, coef_w=None, gout=None, dx=None, accumulate=False, dW=None, addend=None, dW_addend=None):
    """dx / dW are OVERWRITTEN unless `accumulate`.  `addend` [B,T,C], `dW_addend` [(C+1),C]: terms added to dx / dW
    scaled by gout (gradients computed before gout was known)."""
    part = None
    if dW is not None:
        part = torch.empty(_dense_sizes(st.B, st.T, st.C)[0], dtype=_F32, device=x.device)
    tok = _mark("dense_grad")
    N.check(
        N.lib.wfl_dense_grad(ptr(x), ptr(W), st.B, st.T, st.C, ptr(st.alpha), ptr(st.beta), ptr(st.logz), ptr(coef),
                             ptr(coef_w), ptr(gout), int(bool(accumulate)), ptr(addend), ptr(dW_addend), ptr(dx),
                             ptr(dW), ptr(part), ptr(st.ws), stream_ptr())
    )
    _done(tok)


def dense_viterbi(x, W):
    B, T, C = x.shape
    alpha = torch.empty((B, T, C), dtype=_F32, device=x.device)
    bptr = None  # (no back-pointer buffer: include/wfl.h)
    path = torch.empty((B, T), dtype=torch.int32, device=x.device)
    N.check(N.lib.wfl_dense_viterbi(ptr(x), ptr(W), B, T, C, ptr(alpha), ptr(bptr), ptr(path),
                                    stream_ptr()))
    return path


# -------------------------------------------------------------------------------------------------
# CTC fast path
# -------------------------------------------------------------------------------------------------
def upload(dst, pinned, nbytes):
    """dst[:nbytes] (device uint8 tensor) = pinned[:nbytes] (pinned host uint8 tensor) on the current stream, by a
    kernel that reads the pinned buffer directly (wfl_upload): `copy_(non_blocking=True)` -- hipMemcpyAsync -- from
    pinned memory intermittently blocks the host until the stream has drained, which serialises host and GPU."""
    N.check(N.lib.wfl_upload(dst.data_ptr(), pinned.data_ptr(), int(nbytes), stream_ptr()))


class _StagingRing:
    """Pinned host buffers through which a batch's targets reach the device in ONE asynchronous copy.  A slot is
    reused only after the copy that last read it has completed (event per slot)."""

    def __init__(self, slots=8, nbytes=1 << 18):
        self.bufs, self.views, self.events = [None] * slots, [None] * slots, [None] * slots
        self.i, self.nbytes = 0, nbytes

    def next(self, need, pinned):
        i = self.i = (self.i + 1) % len(self.bufs)
        if self.events[i] is not None:
            self.events[i].synchronize()
        buf = self.bufs[i]
        if buf is None or buf.numel() < need:
            n = max(self.nbytes, 1 << (int(need) - 1).bit_length())
            # every slot at once: a pinned allocation costs hundreds of microseconds -- slot by slot they would be
            # spread over the first steps of a run instead of being paid in the first one
            for k in range(len(self.bufs)):
                if self.bufs[k] is None or self.bufs[k].numel() < n:
                    if self.events[k] is not None:
                        self.events[k].synchronize()
                    self.bufs[k] = torch.empty(n, dtype=torch.uint8, pin_memory=pinned)
                    self.views[k] = self.bufs[k].numpy()
            buf = self.bufs[i]
        return i, buf, self.views[i]


_STAGING = {}
_LATTICE_STAGING = {}


def _lattice_ring(device):
    """The pinned staging ring of the lattice packers on `device` -- one PER HOST THREAD: Transducer.prepare packs the
    next batch on a side thread while the caller's thread packs (ASG force alignment, Viterbi, CTC lattices) right
    after the loss; a ring's cursor, its `ring.i -= 1` retake and its per-slot events are not atomic, and two threads
    handed the same pinned slot would upload each other's bytes.  (The C++ operator's ring, csrc/torch_ops.cpp
    PinnedRing, is per device and only touched under the GIL without releasing it.)"""
    key = (device.index, threading.get_ident())
    ring = _LATTICE_STAGING.get(key)
    if ring is None:
        ring = _LATTICE_STAGING[key] = _StagingRing(slots=4, nbytes=1 << 22)
    return ring
_FACTORS = ("scale_none", "scale_mean", "cpos_none", "cpos_mean", "cneg_none", "cneg_mean")


class CtcTargets:
    """Targets of a batch as the kernels want them, resident on the device: int64 offsets [B+1], int32 flat labels
    and the per-utterance loss / gradient factors of both reductions (scale_b = 1/len_b for "mean", 1 for "none";
    +-scale_b/B: ctc.py:53-58,87, asg.py:116-121,171-179) in ONE buffer filled on the host and uploaded with one
    asynchronous copy from pinned memory.  Lists of int lists are flattened by the CPython helper (_wflpy), lists
    of 1-D tensors by one torch.cat."""

    __slots__ = ("max_len", "B", "dev_buf", "cache", "label_min", "label_max", "n", "_off_flat", "_off_fac", "_key",
                 "_lens", "_uploaded")

    def __init__(self, targets, device, flat=None, lens=None, _staged=None):
        self.cache = {}  # derived device objects (factor views, packed lattices), keyed by the caller
        st = _staged if _staged is not None else _stage_targets(targets, device, flat, lens)
        slot, host, nbytes, B, n, self.max_len, self.label_min, self.label_max, self._off_flat, self._off_fac, ck = st
        self._key = (host[1][:ck[2]].tobytes(), ck)  # host copy of [offsets | labels]
        self.B, self.n, self._lens = B, n, None
        view = host[1]
        if device.type == "cuda":
            ring = _STAGING[(device.index, threading.get_ident())]  # (the ring _stage_targets filled: same thread)
            self.dev_buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            upload(self.dev_buf, host[0], nbytes)
            ev = ring.events[slot]
            if ev is None:
                ev = ring.events[slot] = torch.cuda.Event()
            ev.record()  # (the staging slot: reusable once this upload has read it)
            # the targets' OWN event: a later user on ANOTHER stream orders itself behind the upload (targets_on_device);
            # the slot's event may have been re-recorded by then, possibly on a different stream
            own = torch.cuda.Event()
            own.record()
            self._uploaded = (stream_ptr(), own)
        else:  # host-only uses (tests of the packers): same layout, no device
            self.dev_buf = torch.from_numpy(view[:nbytes].copy())
            self._uploaded = None

    # host copies of the staged content (the key bytes hold offsets and labels back to back)
    @property
    def offsets(self):
        return np.frombuffer(self._key[0], dtype=np.int64, count=self.B + 1)

    @property
    def flat(self):
        return np.frombuffer(self._key[0], dtype=np.int32, count=self.n, offset=self._off_flat)

    @property
    def lens(self):
        if self._lens is None:
            self._lens = np.diff(self.offsets).tolist()
        return self._lens

    # raw device addresses (what the C ABI takes) and tensor views of the same memory
    def addr(self, what):
        base = self.dev_buf.data_ptr()
        if what == "offsets":
            return base
        if what == "flat":
            return base + self._off_flat
        return base + self._off_fac + 4 * self.B * _FACTORS.index(what)

    def factor(self, what):
        k = _FACTORS.index(what)
        lo = self._off_fac + 4 * self.B * k
        return self.dev_buf[lo:lo + 4 * self.B].view(_F32)

    @property
    def dev_offsets(self):
        return self.dev_buf[:8 * (self.B + 1)].view(torch.int64)

    @property
    def dev_flat(self):
        return self.dev_buf[self._off_flat:self._off_flat + 4 * max(self.n, 1)].view(torch.int32)


def _stage_targets(targets, device, flat=None, lens=None):
    """Fill a staging slot with [offsets | flat | factors]; returns what CtcTargets needs plus a content key."""
    _wflpy = _staging_helper()

    key = device.index if device.type == "cuda" else -1
    ring = _STAGING.get((key, threading.get_ident()))  # (per host thread, as _lattice_ring)
    if ring is None:
        ring = _STAGING[(key, threading.get_ident())] = _StagingRing()
    pinned = device.type == "cuda"
    B = len(lens) if flat is not None else len(targets)
    off_flat = 8 * (B + 1)
    tail = 4 * B * len(_FACTORS) + 16
    slot, buf, view = ring.next(off_flat + tail + 4096, pinned)
    if flat is None and len(targets) and all(type(t) is torch.Tensor and t.dim() == 1 and not t.is_cuda for t in targets):
        lens = [t.numel() for t in targets]
        flat = torch.cat(targets).to(torch.int32).numpy() if sum(lens) else np.zeros(0, np.int32)
    if flat is not None:
        n = int(flat.size)
        if off_flat + 4 * n + tail > buf.numel():
            ring.i -= 1
            slot, buf, view = ring.next(off_flat + 4 * n + tail, pinned)
        offs = view[:off_flat].view(np.int64)
        offs[0] = 0
        np.cumsum(lens, out=offs[1:])
        view[off_flat:off_flat + 4 * n].view(np.int32)[:] = flat
        max_len = max(lens) if lens else 0
        lo, hi = (int(flat.min()), int(flat.max())) if n else (0, -1)
    else:
        rows = targets
        while True:
            cap = (buf.numel() - off_flat - tail) // 4
            try:
                res = _wflpy.flatten_into(rows, buf.data_ptr() + off_flat, cap, buf.data_ptr())
            except TypeError:  # rows of tensors / ranges / numpy ints: normalise once and retry
                rows = [t.tolist() if hasattr(t, "tolist") else [int(v) for v in t] for t in rows]
                continue
            if res is not None:
                break
            need = off_flat + 4 * int(view[:off_flat].view(np.int64)[B]) + tail
            ring.i -= 1
            slot, buf, view = ring.next(need, pinned)
        n, max_len, lo, hi = res
    off_fac = (off_flat + 4 * max(n, 1) + 7) & ~7
    nbytes = off_fac + 4 * B * len(_FACTORS)
    _wflpy.factors_into(buf.data_ptr(), B, buf.data_ptr() + off_fac)  # (the six _FACTORS arrays, in that order)
    # content key of the batch: a 128-bit hash of [offsets | labels] (hashing the bytes object itself costs Python more
    # than the staging); a cache hit is confirmed byte for byte before it is used (targets_on_device)
    nkey = off_flat + 4 * n
    content = _wflpy.content_key(buf.data_ptr(), nkey) + (nkey, key)
    return slot, (buf, view), nbytes, B, n, max_len, lo, hi, off_flat, off_fac, content


_CTC_WS_SIZES = {}
CTC_FAST_MAX_LEN = 255  # longest target of the CTC fast path (four positions per lane); beyond: lattice engine
CTC_FAST_MAX_CLASSES = 16000  # widest emission row of the CTC fast path (compact gradient tiles: 8 x (4 KB + C bytes) of LDS) ...
CTC_FAST_MAX_CLASSES_LONG = 602  # ... and for targets of more than 63 labels (dense row tiles [17][C] per wave)


def check_labels(tg, C, what):
    """Labels must index the emission row: the reference's intersect would find no path (loss +inf);
    here an out-of-range label is a caller error and is rejected before any kernel indexes with it."""
    if tg.label_min < 0 or tg.label_max >= C:
        bad = tg.label_min if tg.label_min < 0 else tg.label_max
        raise ValueError(f"{what}: target label {bad} is outside [0, {C}) (emissions have {C} classes)")


def ctc_fast_path_ok(max_len, C):
    """Does the CTC fast path (csrc/ctc_kernels.hip) take this shape?  Otherwise: the generic lattice engine."""
    return max_len <= CTC_FAST_MAX_LEN and C <= (CTC_FAST_MAX_CLASSES if max_len <= 63 else CTC_FAST_MAX_CLASSES_LONG)
CTC_DEFAULT_FLAGS = 0  # (wfl_ctc_forward's flags: must be 0)


def ctc_forward(x, tg, blank, flags=None):
    B, T, C = x.shape
    key = (B, T, C, tg.max_len)
    n_ws = _CTC_WS_SIZES.get(key)
    if n_ws is None:
        n = ctypes.c_int64()
        N.check(N.lib.wfl_ctc_workspace(B, T, C, tg.max_len, ctypes.byref(n)))
        n_ws = _CTC_WS_SIZES[key] = n.value
    ws = torch.empty(n_ws, dtype=_F32, device=x.device)
    nll = torch.empty(B, dtype=_F32, device=x.device)
    if flags is None:
        flags = CTC_DEFAULT_FLAGS
    N.check(
        N.lib.wfl_ctc_forward(ptr(x), B, T, C, ptr(tg.dev_flat), ptr(tg.dev_offsets), tg.max_len, blank, flags,
                              ptr(ws), ptr(nll), stream_ptr())
    )
    return ws, nll


def row_lse(x):
    """[B,T] log-sum-exp over the classes of x [B,T,C] (the forward half of a fused log_softmax)."""
    B, T, C = x.shape
    out = torch.empty((B, T), dtype=_F32, device=x.device)
    N.check(N.lib.wfl_row_lse(ptr(x), B * T, C, ptr(out), stream_ptr()))
    return out


def collapse_rows(paths, drop=None):
    """The collapse every criterion's viterbi ends with (ctc.py:130-134, asg.py:228-233), for a whole batch at once:
    rows of `paths` (numpy [B,T] ints) with runs of equal labels reduced to one label, then the entries equal to `drop`
    removed.  Returns (flat values, row-major; lengths [B]) -- a Python loop over the rows costs milliseconds at a
    benchmark batch (128 rows of 1000 frames), this costs tens of microseconds."""
    import numpy as np

    B, T = paths.shape
    keep = np.ones((B, T), dtype=bool)
    if T > 1:
        np.not_equal(paths[:, 1:], paths[:, :-1], out=keep[:, 1:])
    if drop is not None:
        keep &= paths != drop
    return paths[keep], keep.sum(axis=1)


def split_rows(flat, lens, dtype):
    """list of B tensors (views of one CPU tensor of `dtype`) holding the rows of (flat, lens)"""
    t = torch.from_numpy(flat).to(dtype)
    return list(torch.split(t, [int(n) for n in lens]))


def row_argmax(x):
    """[B,T] int32: per frame the first maximal class of x [B,T,C] -- viterbi_path of the bare emissions graph
    (transducer.py:205-216 without transitions)."""
    B, T, C = x.shape
    out = torch.empty((B, T), dtype=torch.int32, device=x.device)
    N.check(N.lib.wfl_row_argmax(ptr(x), B * T, C, ptr(out), stream_ptr()))
    return out


_CTC_WS_CACHE = {}


def ctc_workspace(x, max_len):
    """Scratch of the pipelined step (checkpoints, flags, certificate words) and the per-utterance nll, one per
    (device, stream, shape): consecutive steps on a stream are ordered, so they can share it -- no allocation per
    call.  (The returned nll is overwritten by the next step of the same shape on the same stream.)"""
    B, T, C = x.shape
    idx = x.device.index
    key = (idx, torch._C._cuda_getCurrentRawStream(idx), B, T, C, max_len)
    hit = _CTC_WS_CACHE.get(key)
    if hit is None:
        n_ws = _CTC_WS_SIZES.get(key[2:])
        if n_ws is None:
            n = ctypes.c_int64()
            N.check(N.lib.wfl_ctc_workspace(B, T, C, max_len, ctypes.byref(n)))
            n_ws = _CTC_WS_SIZES[key[2:]] = n.value
        if len(_CTC_WS_CACHE) >= 16:
            _CTC_WS_CACHE.clear()
        hit = _CTC_WS_CACHE[key] = (torch.empty(n_ws, dtype=_F32, device=x.device),
                                    torch.empty(B, dtype=_F32, device=x.device))
    return hit


_CTC_STATE = collections.OrderedDict()  # (device, stream, shape) -> (pinned word pair, CtcCall), least recently used first
_CTC_STATE_MAX = 256                    # shapes remembered at once (variable-length training sees thousands)
_CTC_PAGES = []                         # pinned int32 pages the pairs are cut from (never freed: a launch may still write)
_CTC_FREE = []                          # word pairs free for reuse: (page tensor, offset)


def _ctc_state_slot():
    if not _CTC_FREE:
        page = torch.zeros(1024, dtype=torch.int32).pin_memory()  # ONE pinned allocation per 512 shapes
        _CTC_PAGES.append(page)
        _CTC_FREE.extend((page, o) for o in range(1022, -1, -2))
    page, o = _CTC_FREE.pop()
    words = page[o:o + 2]
    words.zero_()
    return words


def ctc_host_state(x, max_len):
    """The CTC step's memory between calls (`wfl_ctc_call.host_state`, include/wfl.h): two int32 of pinned host memory
    per (device, stream, shape) -- the repair launch leaves there how many utterances it recomputed, the next call of
    the shape reads it (no synchronisation) and picks its launch.  Owned here, by the caller of the C ABI: the pairs
    are cut from a few pinned pages that are never freed (a launch may still write them), at most _CTC_STATE_MAX shapes
    are remembered (least recently used first out: its pair goes to the next new shape -- a late write of the evicted
    shape's launch can then only mislead that shape's FIRST choice of launch, never a result), zeroed by
    ctc_reset_state().  Returns (pinned words, CtcCall struct)."""
    B, T, C = x.shape
    idx = x.device.index
    key = (idx, torch._C._cuda_getCurrentRawStream(idx), B, T, C, max_len)
    hit = _CTC_STATE.get(key)
    if hit is None:
        if torch.cuda.is_current_stream_capturing():
            # (a step captured into a graph replays ONE choice and must not allocate: no memory, lane-exponent step first)
            return None, N.CtcCall(0, None)
        if len(_CTC_STATE) >= _CTC_STATE_MAX:
            _, (old_words, _call) = _CTC_STATE.popitem(last=False)
            _CTC_FREE.append((old_words._base if old_words._base is not None else old_words, old_words.storage_offset()))
        words = _ctc_state_slot()
        hit = _CTC_STATE[key] = (words, N.CtcCall(0, words.data_ptr()))
    else:
        _CTC_STATE.move_to_end(key)
    return hit


def ctc_reset_state():
    """Forget which launch the CTC steps preferred (tests that run unrelated data through one shape)."""
    for words, _ in _CTC_STATE.values():
        words.zero_()
    try:
        from . import _wfl_torch
    except ImportError:
        return
    _wfl_torch.ctc_reset_host_state()


def ctc_forward_backward(x, tg, blank, coef, gout, dx, loss_scale=None, want_loss=False, lse=None, shared_ws=False):
    """Loss and gradient in one pipelined launch (wfl_ctc_forward_backward): returns (ws, nll) or,
    with want_loss, (ws, nll, mean_b(loss_scale[b] * nll[b]) as a 0-dim device tensor).  `coef` / `loss_scale`
    may be tensors or raw device addresses (CtcTargets.addr).  shared_ws: use the per-stream cached workspace."""
    B, T, C = x.shape
    if shared_ws:
        ws, nll = ctc_workspace(x, tg.max_len)
    else:
        key = (B, T, C, tg.max_len)
        n_ws = _CTC_WS_SIZES.get(key)
        if n_ws is None:
            n = ctypes.c_int64()
            N.check(N.lib.wfl_ctc_workspace(B, T, C, tg.max_len, ctypes.byref(n)))
            n_ws = _CTC_WS_SIZES[key] = n.value
        ws = torch.empty(n_ws, dtype=_F32, device=x.device)
        nll = torch.empty(B, dtype=_F32, device=x.device)
    loss = torch.empty((), dtype=_F32, device=x.device) if want_loss else None
    tok = _mark("ctc_step")
    words, call = ctc_host_state(x, tg.max_len)
    call.n_labels = tg.n
    N.check(
        N.lib.wfl_ctc_forward_backward_call(ptr(x), B, T, C, ptr(tg.addr("flat")), ptr(tg.addr("offsets")), tg.max_len,
                                            blank, ptr(ws), ptr(nll), ptr(coef), ptr(gout), ptr(dx), ptr(loss_scale),
                                            ptr(loss), ptr(lse), ctypes.byref(call), stream_ptr())
    )
    _done(tok)
    return (ws, nll, loss) if want_loss else (ws, nll)


_CTC_WS_FIELDS = {}


def ctc_workspace_field(ws, B, T, max_len, field):
    """Slice of the CTC workspace that holds a bookkeeping field (include/wfl.h, WFL_CTC_WS_*)."""
    key = (B, T, max_len, field)
    span = _CTC_WS_FIELDS.get(key)
    if span is None:
        off, n = ctypes.c_int64(), ctypes.c_int64()
        N.check(N.lib.wfl_ctc_workspace_field(B, T, max_len, field, ctypes.byref(off), ctypes.byref(n)))
        span = _CTC_WS_FIELDS[key] = (off.value, n.value)
    return ws[span[0]:span[0] + span[1]]


def ctc_pipeline_gave_up(ws, B, T, max_len):
    """True if a gradient wave of the pipelined step stopped waiting for a checkpoint (diagnostics)."""
    return bool(ctc_workspace_field(ws, B, T, max_len, N.CTC_WS_STATUS).view(torch.int32)[0].item())


def ctc_pipeline_repaired(ws, B, T, max_len):
    """Number of utterances of the last pipelined step whose lane-exponent chains failed the certificate
    and were recomputed in the log domain by the repair launch (diagnostics; 0 on the log-domain step)."""
    return int(ctc_workspace_field(ws, B, T, max_len, N.CTC_WS_STATUS).view(torch.int32)[1].item())


def ctc_grad(x, tg, blank, ws, nll, coef, gout, dx):
    B, T, C = x.shape
    N.check(
        N.lib.wfl_ctc_grad(ptr(x), B, T, C, ptr(tg.dev_flat), ptr(tg.dev_offsets), tg.max_len, blank, ptr(ws),
                           ptr(nll), ptr(coef), ptr(gout), ptr(dx), stream_ptr())
    )


def loss_factors(tg, reduction, norm_lens=None):
    """Per-utterance loss scale and gradient coefficients as device tensors (scale, +scale/B, -scale/B).

    scale_b = 1/len_b for "mean" (1 if len_b == 0), 1 for "none" (ctc.py:53-58, asg.py:116-121,
    transducer.py:302-305).  Normalised by the target lengths they are views of the buffer uploaded with the
    targets (no kernel, no copy); `norm_lens` (STC normalises by T) builds them separately, cached on `tg`."""
    if reduction not in ("mean", "none"):
        raise ValueError("invalid value for reduction '" + str(reduction) + "'")
    key = ("factors", reduction, None if norm_lens is None else tuple(norm_lens))
    hit = tg.cache.get(key)
    if hit is None:
        if norm_lens is None:
            hit = (tg.factor("scale_" + reduction), tg.factor("cpos_" + reduction), tg.factor("cneg_" + reduction))
        else:
            sc = [1.0 / n if n > 0 else 1.0 for n in norm_lens] if reduction == "mean" else [1.0] * len(norm_lens)
            scale = torch.tensor(sc, dtype=_F32, device=tg.dev_buf.device)
            hit = (scale, scale / len(sc), -scale / len(sc))
        tg.cache[key] = hit
    return hit


_TARGET_CACHE = LRU(64)


def targets_on_device(targets, device):
    """Stage and upload the targets of a batch (once per distinct content: a hash of the staged bytes is the key of a
    small LRU and a hit is confirmed byte for byte, so a repeated batch -- the reference benchmarks reuse one target list -- skips the upload and keeps its
    derived objects).  A CtcTargets built earlier is passed through."""
    if isinstance(targets, CtcTargets):
        return targets
    _wflpy = _staging_helper()

    st = _stage_targets(targets, device)
    data = _TARGET_CACHE.data
    hit = data.get(st[-1])
    if hit is not None and _wflpy.same_bytes(st[1][0].data_ptr(), hit._key[0]):
        data.move_to_end(st[-1])
        if hit._uploaded is not None and hit._uploaded[0] != stream_ptr():
            torch.cuda.current_stream().wait_event(hit._uploaded[1])
        return hit
    val = data[st[-1]] = CtcTargets(None, device, _staged=st)
    data.move_to_end(st[-1])
    if len(data) > _TARGET_CACHE.capacity:
        data.popitem(last=False)
    return val
