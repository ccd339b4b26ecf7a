"""Star Temporal Classification -- counterpart of /root/reference/criterions/stc.py.

The label graph of stc.py:23-64 (self-loop-less CTC graph plus <star> states carrying
log(prob) penalties) is packed by the native builder and scored by the generic lattice engine;
the alphabet augmentation (<star>, <star>\\token columns) stays in torch as in stc.py:174-221.
"""
import math

import torch

from .. import _native as N
from .. import engine as E
from .. import graph as G

# blank idx is REQUIRED to be zero (stc.py:13)
STC_BLANK_IDX = 0


class STCLossFunction(torch.autograd.Function):
    """STC with autograd; assumes <star>, <star>\\token columns are appended to the input."""

    @staticmethod
    def create_stc_graph(target, star_idx, prob):
        """stc.py:23-64 as a host graph (API parity; the device path uses the bulk builder)."""
        g = G.Graph(False)
        L = len(target)
        S = 2 * L + 1
        for s in range(S):
            g.add_node(s == 0, s >= S - 2)
            label = target[(s - 1) // 2] if s % 2 else STC_BLANK_IDX
            if label == STC_BLANK_IDX:
                g.add_arc(s, s, label)
            if s > 0:
                g.add_arc(s - 1, s, label)
            if s % 2 and s > 1:
                g.add_arc(s - 2, s, label)
        lp = math.log(prob)
        for l in range(L + 1):
            p_tok, p_blank = 2 * l - 1, 2 * l
            c = g.add_node(False, l == L)
            star = star_idx if l == L else star_idx + target[l]
            if p_tok >= 0:
                g.add_arc(p_tok, c, star, star, lp)
            g.add_arc(p_blank, c, star, star, lp)
            g.add_arc(c, c, star, star, lp)
            if l < L:
                g.add_arc(c, 2 * l + 1, target[l])
            g.add_arc(c, p_blank, STC_BLANK_IDX)
        return g

    @staticmethod
    @E.on_input_device
    def forward(ctx, inputs, targets, prob, reduction="none"):
        B, T, Cstar = inputs.shape
        if reduction not in ("none", "mean"):  # stc.py:92-93
            raise ValueError("invalid value for reduction '" + str(reduction) + "'")
        if T == 0:
            raise ValueError("STCLoss: empty emissions (T == 0)")
        dev = E.require_gpu()
        x = E.as_device_f32(inputs.detach(), dev)
        tg = E.targets_on_device(targets, dev)
        if tg.B != B:
            raise ValueError(f"got {tg.B} targets for a batch of {B}")
        # "mean" divides by the number of frames, not the target length (stc.py:90-91)
        scale, _, cneg = E.loss_factors(tg, reduction, norm_lens=[T] * B)
        key = ("stc", Cstar, float(prob))
        pack = tg.cache.get(key)
        if pack is None:
            pack = tg.cache[key] = E.PackedLattice.stc(tg.flat, tg.offsets, Cstar // 2, math.log(prob), Cstar, dev)
        st = E.lattice_forward(x, pack, need_beta=inputs.requires_grad)
        loss = E.reduce_loss(st.logz, scale, -1.0)
        ctx.aux = (x, st, cneg)
        ctx.in_device = inputs.device
        return loss if inputs.is_cuda else loss.cpu()

    @staticmethod
    @E.on_input_device
    def backward(ctx, grad_output):
        x, st, cneg = ctx.aux
        gout = E.as_device_f32(grad_output.detach().reshape(1), x.device)
        dx = torch.empty_like(x)
        E.lattice_grad(st, cneg, gout=gout, dx=dx)
        if ctx.in_device.type != "cuda":
            dx = dx.to(ctx.in_device)
        return dx, None, None, None


STCLoss = STCLossFunction.apply


class _StcAugment(torch.autograd.Function):
    """stc.py:199-220 -- the batch's classes selected, <star> = logsumexp over the non-blank classes and <star>\\token
    appended -- as ONE launch each way (wfl_stc_augment / wfl_stc_augment_grad) instead of six torch ops over [B, T, C]
    whose intermediates autograd keeps: (T, B, C) log-probabilities in, (B, T, 2K) out."""

    @staticmethod
    def forward(ctx, inputs, select, inv):
        T, B, C = inputs.shape
        K = select.numel()
        x = inputs.detach().contiguous()
        out = torch.empty((B, T, 2 * K), dtype=torch.float32, device=x.device)
        lse = torch.empty(T * B, dtype=torch.float32, device=x.device)
        N.check(N.lib.wfl_stc_augment(E.ptr(x), T, B, C, E.ptr(select), K, E.ptr(out), E.ptr(lse), E.stream_ptr()))
        ctx.aux = (x, select, inv, lse)
        return out

    @staticmethod
    def backward(ctx, g):
        x, select, inv, lse = ctx.aux
        T, B, C = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x)
        N.check(N.lib.wfl_stc_augment_grad(E.ptr(x), T, B, C, E.ptr(select), E.ptr(inv), select.numel(), E.ptr(lse),
                                           E.ptr(g), E.ptr(dx), E.stream_ptr()))
        return dx, None, None


class STC(torch.nn.Module):
    """The Star Temporal Classification loss (stc.py:135-221).

    p0 / plast / thalf: initial and final token-insertion penalty (before the log) and the number
    of training steps after which it reaches their midpoint."""

    def __init__(self, blank_idx, p0=1, plast=1, thalf=1, reduction="none"):
        super(STC, self).__init__()
        assert blank_idx == STC_BLANK_IDX
        self.p0 = p0
        self.plast = plast
        self.thalf = thalf
        self.nstep = 0
        self.reduction = reduction

    @staticmethod
    def logsubexp(a, b):
        """log(exp(a) - exp(b)) for a [M,N,1], b [M,N,O] (stc.py:158-172)."""
        with torch.set_grad_enabled(a.requires_grad):
            a = a.tile((1, 1, b.shape[2]))
            return a + torch.log1p(1e-7 - torch.exp(b - a))

    @E.on_input_device
    def forward(self, inputs, targets):
        """inputs: (T, B, C) log-probabilities; targets: list of B label lists."""
        if self.training:
            self.nstep += 1
        prob = self.plast + (self.p0 - self.plast) * math.exp(-self.nstep * math.log(2) / self.thalf)
        labels = set(t for target in targets for t in target)
        # (one launch each way for the alphabet augmentation -- unless a target names the blank itself: the reference's
        # select list then holds column 0 twice (stc.py:205), which the torch spelling below reproduces as it stands;
        # an empty batch / zero frames also go there)
        if (inputs.is_cuda and inputs.dtype == torch.float32 and inputs.dim() == 3 and inputs.shape[2] > 1
                and inputs.shape[0] * inputs.shape[1] > 0 and STC_BLANK_IDX not in labels):
            C = inputs.shape[2]
            bad = [t for t in labels if not 0 <= int(t) < C]
            if bad:  # (index_select of stc.py:207 raises; the kernel would read x[t, b, label] out of bounds)
                raise IndexError(f"index out of range in self: target label {bad[0]} for {C} classes")
            # keep only blank and the tokens present in this batch (stc.py:205-209: the same list, in the same order,
            # as the torch spelling below), then the augmentation in one launch
            select_idx = [STC_BLANK_IDX] + list(labels)
            target_map = {t: i for i, t in enumerate(select_idx)}
            inv_host = torch.full((C,), -1, dtype=torch.int32)
            sel_host = torch.tensor(select_idx, dtype=torch.int32)
            inv_host[sel_host.long()] = torch.arange(len(select_idx), dtype=torch.int32)
            both = torch.cat([sel_host, inv_host]).to(inputs.device, non_blocking=True)
            mapped = [[target_map[t] for t in target] for target in targets]
            with torch.cuda.device(inputs.device):
                log_probs = _StcAugment.apply(inputs, both[:len(select_idx)], both[len(select_idx):])
            return STCLoss(log_probs, mapped, prob, self.reduction)
        log_probs = inputs.permute(1, 0, 2)  # (T, B, C) -> (B, T, C)
        with torch.set_grad_enabled(log_probs.requires_grad):
            lse = torch.logsumexp(log_probs[:, :, 1:], 2, keepdim=True)  # <star>
            # keep only blank and the tokens present in this batch
            select_idx = [STC_BLANK_IDX] + list(labels)
            target_map = {t: i for i, t in enumerate(select_idx)}
            select = torch.IntTensor(select_idx).to(log_probs.device)
            log_probs = log_probs.index_select(2, select)
            targets = [[target_map[t] for t in target] for target in targets]
            neglse = STC.logsubexp(lse, log_probs[:, :, 1:])  # <star>\token
            log_probs = torch.cat([log_probs, lse, neglse], dim=2)
        return STCLoss(log_probs, targets, prob, self.reduction)
