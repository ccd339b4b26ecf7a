"""CTC criterion -- counterpart of /root/reference/criterions/ctc.py.

`CTCLossFunction.forward/backward`, `CTCLoss` and `CTC` keep the reference's signatures
(ctc.py:13-135).  Where the reference builds `gtn.intersect(g_emissions, g_criterion)` per sample
on host threads (ctc.py:38-65), this sends the whole batch through the CTC fast-path kernels
(csrc/ctc_kernels.hip; up to four target positions per lane) -- or, for targets longer than 255
labels, through the generic lattice engine (csrc/lattice_kernels.hip).  Both are HIP paths; there is no CPU path.
"""
import os

import torch

from .. import engine as E
from .. import graph as G


class CTCLossFunction(torch.autograd.Function):
    @staticmethod
    def create_ctc_graph(target, blank_idx):
        """ctc.py:15-29 as a host graph (API parity; the kernels never need it)."""
        g = G.Graph(False)
        L = len(target)
        S = 2 * L + 1
        for s in range(S):
            g.add_node(s == 0, s >= S - 2)
            label = target[(s - 1) // 2] if s % 2 else blank_idx
            g.add_arc(s, s, label)
            if s > 0:
                g.add_arc(s - 1, s, label)
            if s % 2 and s > 1 and label != target[(s - 1) // 2 - 1]:
                g.add_arc(s - 2, s, label)
        g.arc_sort(False)
        return g

    @staticmethod
    @E.on_input_device
    def forward(ctx, log_probs, targets, blank_idx=0, reduction="none"):
        B, T, C = log_probs.shape
        if T == 0:
            raise ValueError("CTCLoss: empty emissions (T == 0)")
        if reduction not in ("none", "mean"):  # ctc.py:57-58
            raise ValueError("invalid value for reduction '" + str(reduction) + "'")
        dev = E.require_gpu()
        x = E.as_device_f32(log_probs.detach(), dev)
        tg = E.targets_on_device(targets, dev)
        if tg.B != B:
            raise ValueError(f"got {tg.B} targets for a batch of {B}")
        E.check_labels(tg, C, "CTCLoss")
        if not 0 <= int(blank_idx) < C:
            raise ValueError(f"CTCLoss: blank index {blank_idx} is outside [0, {C})")
        need_grad = log_probs.requires_grad
        if E.ctc_fast_path_ok(tg.max_len, C) and need_grad:
            # loss and gradient in ONE pipelined launch (gradient waves run behind the chains); backward
            # only applies the upstream scalar.  Like torch's own CTC, the gradient is produced eagerly.
            # The per-utterance factors (loss scale; gradient coefficient -scale/B) were uploaded with the targets.
            scale, coef = tg.addr("scale_" + reduction), tg.addr("cneg_" + reduction)
            dx = torch.empty_like(x)
            lse = E.row_lse(x) if ctx_log_softmax(ctx) else None
            _, _, loss = E.ctc_forward_backward(x, tg, int(blank_idx), coef, None, dx, loss_scale=scale, want_loss=True,
                                                lse=lse, shared_ws=True)
            ctx.aux = ("pipelined", x, tg, int(blank_idx), dx, coef, lse)
        elif ctx_log_softmax(ctx):
            raise RuntimeError("fused log_softmax CTC is only used on the pipelined path")
        elif E.ctc_fast_path_ok(tg.max_len, C):
            scale, _, coef = E.loss_factors(tg, reduction)
            ws, nll = E.ctc_forward(x, tg, int(blank_idx))
            loss = E.reduce_loss(nll, scale, 1.0)
            ctx.aux = ("fast", x, tg, int(blank_idx), None, nll, coef)
        else:
            pack = tg.cache.get(("ctc_lattice", int(blank_idx), C))
            if pack is None:
                pack = tg.cache[("ctc_lattice", int(blank_idx), C)] = E.PackedLattice.ctc(
                    tg.flat, tg.offsets, int(blank_idx), C, dev)
            scale, _, coef = E.loss_factors(tg, reduction)
            st = E.lattice_forward(x, pack, need_beta=need_grad)
            loss = E.reduce_loss(st.logz, scale, -1.0)
            ctx.aux = ("lattice", x, st, coef)
        ctx.in_device = log_probs.device
        return loss if log_probs.is_cuda else loss.cpu()

    @staticmethod
    @E.on_input_device
    def backward(ctx, grad_output):
        kind, x = ctx.aux[0], ctx.aux[1]
        gout = E.as_device_f32(grad_output.detach().reshape(1), x.device)
        if kind == "pipelined":
            _, _, tg, blank, dx, coef, lse = ctx.aux
            if dx is None:
                # a second backward through a retained graph: the eager gradient was handed out (and scaled in
                # place) by the first one, so run the same launch again -- with the same row log-sum-exps when
                # the log_softmax is fused -- into a fresh buffer, the upstream scalar applied by the kernel
                dx = torch.empty_like(x)
                E.ctc_forward_backward(x, tg, blank, coef, gout, dx, lse=lse, shared_ws=True)
            else:
                E.scale_inplace(dx, gout)
                ctx.aux = ("pipelined", x, tg, blank, None, coef, lse)
        elif kind == "fast":
            raise RuntimeError("CTCLoss: backward through an input that did not require grad in forward")
        else:
            dx = torch.empty_like(x)
            _, _, st, coef = ctx.aux
            E.lattice_grad(st, coef, gout=gout, dx=dx)
        if ctx.in_device.type != "cuda":
            dx = dx.to(ctx.in_device)
        return dx, None, None, None


def ctx_log_softmax(ctx):
    return getattr(ctx, "fused_log_softmax", False)


class _FusedLogSoftmaxCTCLoss(CTCLossFunction):
    """CTCLoss(log_softmax(inputs), ...) as one operator (ctc.py:107 + ctc.py:122): the row log-sum-exps
    are computed once, subtracted where the chains gather their emissions, and the gradient rows
    start at -cf * softmax(inputs); only for the pipelined path (targets up to 255 labels, input
    requires grad) -- the module falls back to torch's log_softmax otherwise."""

    @staticmethod
    @E.on_input_device
    def forward(ctx, inputs, targets, blank_idx=0, reduction="none"):
        ctx.fused_log_softmax = True
        return CTCLossFunction.forward(ctx, inputs, targets, blank_idx, reduction)


class _EagerLoss(torch.Tensor):
    """The scalar the C++ CTC node returns.  A plain tensor in every respect (no __torch_function__ dispatch) but one:
    `loss.backward()` with no arguments -- the call of ctc_benchmark.py:29-31 and of every training loop that uses the
    criterion's output as its loss -- does not run the criterion's own node on the autograd engine: the forward launch
    already computed the gradient, so for leaf emissions it is handed to their .grad directly, and for emissions that
    are a producer's output (a model's, train.py:262-266) the engine is started at THEIR edge with that gradient
    (csrc/torch_ops.cpp ctc_fast_backward: the ones_like fill, the trip through this node and the scale launch cost
    more host time than the step's kernels take).  Anything else -- a gradient argument, retain_graph, create_graph,
    inputs=, hooks on the loss (or on leaf emissions), anomaly mode, the loss used inside a larger expression, a torch
    other than the one the node was compiled against -- goes through torch.Tensor.backward / the engine."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        if (gradient is None and not retain_graph and not create_graph and inputs is None and fast_backward_enabled()
                and not torch.is_anomaly_enabled() and _native_node().ctc_fast_backward(self)):
            return None
        return torch.Tensor.backward(self, gradient, retain_graph, create_graph, inputs=inputs)


_FAST_BACKWARD = os.environ.get("WFL_CTC_FAST_BACKWARD", "1") != "0"  # (0: always the autograd engine -- A/B, tests)
_FAST_BACKWARD_OK = None


def fast_backward_enabled():
    """The short cut reads autograd structures through torch's C++ headers (Node hooks, AccumulateGrad, Engine::execute):
    it is only taken under the torch release the extension was compiled against -- any other falls back to
    torch.Tensor.backward, which is always correct (tests/test_host_library.py pins the fall-back)."""
    global _FAST_BACKWARD_OK
    if _FAST_BACKWARD_OK is None:
        node = _native_node()
        built = getattr(node, "built_for_torch", None)
        _FAST_BACKWARD_OK = bool(built) and built() == torch_release(torch.__version__)
    return _FAST_BACKWARD and _FAST_BACKWARD_OK


def torch_release(version):
    """'2.10.0+rocm7.0' -> '2.10.0' (TORCH_VERSION in the headers carries no local build tag)."""
    return str(version).split("+")[0]


def _native_node():
    """The C++ autograd node of the pipelined step (csrc/torch_ops.cpp), or None if the extension was not built."""
    global _NODE
    if _NODE is False:
        try:
            from .. import _wfl_torch as _NODE
        except ImportError:
            _NODE = None
    return _NODE


_NODE = False


@E.on_input_device
def _ctc_loss(log_probs, targets, blank_idx, reduction, fused_log_softmax):
    """CTCLossFunction.apply, with the hot case -- float32 device emissions that require grad, targets the fast
    kernels take -- routed through the C++ autograd node: same checks, same staging, same launch, but neither the
    forward nor the backward passes through Python's autograd.Function machinery (which costs more host time than
    the step's kernels take on the GPU at the benchmark shape)."""
    node = _native_node()
    if (node is not None and type(log_probs) is torch.Tensor and log_probs.is_cuda and log_probs.requires_grad
            and log_probs.dtype == torch.float32 and log_probs.dim() == 3 and log_probs.is_contiguous()
            and torch.is_grad_enabled() and log_probs.shape[1] > 0 and reduction in ("none", "mean")):
        B, T, C = log_probs.shape
        dev = log_probs.device
        if type(targets) in (list, tuple):
            # staging, upload, checks and launch in one native call (csrc/torch_ops.cpp); None: not its case after all
            lim = (int(blank_idx), reduction == "mean", fused_log_softmax, E.CTC_FAST_MAX_LEN, E.CTC_FAST_MAX_CLASSES,
                   E.CTC_FAST_MAX_CLASSES_LONG)
            tok = None if E.PHASE_EVENTS is None else E._mark("ctc_step")
            if tok is None:
                loss = node.ctc_loss_lists(log_probs, targets, *lim)
            else:  # (profiling: the event pair brackets the launch, not the staging)
                E._event_pool().append(tok[1])  # (recorded too early: take the start event again after the staging)
                st = node.stage_lists(targets, log_probs)
                tok = (tok[0], E._event())
                loss = node.ctc_loss_staged(log_probs, st, *lim)
                E._done(tok)
            if loss is not None:
                loss.__class__ = _EagerLoss
                return loss
        tg = E.targets_on_device(targets, dev)
        if E.ctc_fast_path_ok(tg.max_len, C):
            if tg.B != B:
                raise ValueError(f"got {tg.B} targets for a batch of {B}")
            E.check_labels(tg, C, "CTCLoss")
            if not 0 <= int(blank_idx) < C:
                raise ValueError(f"CTCLoss: blank index {blank_idx} is outside [0, {C})")
            ws, nll = E.ctc_workspace(log_probs, tg.max_len)
            lse = E.row_lse(log_probs.detach()) if fused_log_softmax else None
            fac = tg._off_fac + 4 * B * (0 if reduction == "none" else 1)  # byte offset of scale_<reduction>
            tok = E._mark("ctc_step")
            words, _ = E.ctc_host_state(log_probs, tg.max_len)
            loss = node.ctc_step(log_probs, tg.dev_buf, 0, tg._off_flat, fac, fac + 16 * B, tg.max_len, int(blank_idx),
                                 ws, nll, lse, tg.n, 0 if words is None else words.data_ptr())
            E._done(tok)
            return loss
    fn = _FusedLogSoftmaxCTCLoss if fused_log_softmax else CTCLossFunction
    return fn.apply(log_probs, targets, blank_idx, reduction)


def CTCLoss(log_probs, targets, blank_idx=0, reduction="none"):
    """ctc.py:96 (`CTCLoss = CTCLossFunction.apply`): same call, same result."""
    return _ctc_loss(log_probs, targets, blank_idx, reduction, False)


class CTC(torch.nn.Module):
    def __init__(self, blank, use_pt):
        super(CTC, self).__init__()
        self.blank = blank  # index of blank label
        self.use_pt = use_pt  # use torch.nn.functional.ctc_loss instead of the WFST engine

    @E.on_input_device
    def forward(self, inputs, targets):
        if not self.use_pt and inputs.requires_grad and inputs.dtype == torch.float32 and \
                E.ctc_fast_path_ok(max((t.numel() for t in targets), default=0), inputs.shape[2]):
            return _ctc_loss(inputs, targets, self.blank, "mean", True)
        log_probs = torch.nn.functional.log_softmax(inputs, dim=2)
        if self.use_pt:  # ctc.py:109-121
            return torch.nn.functional.ctc_loss(
                log_probs.permute(1, 0, 2), torch.cat(targets), [inputs.shape[1]] * inputs.shape[0],
                [t.numel() for t in targets], blank=self.blank, zero_infinity=True,
            )
        return CTCLoss(log_probs, [t.tolist() for t in targets], self.blank, "mean")

    def viterbi(self, outputs):
        """Greedy decode (ctc.py:126-135): argmax, collapse repeats, drop blank."""
        best = torch.argmax(outputs, dim=2).to("cpu").numpy()
        flat, lens = E.collapse_rows(best, drop=self.blank)  # (the whole batch at once: no loop over the rows)
        return E.split_rows(flat, lens, torch.int64)
