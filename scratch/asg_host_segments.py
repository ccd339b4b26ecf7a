"""Host time of the pieces of ASGLoss fwd+bwd with fresh targets (perf_counter around the engine calls)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gtn_applications_amd import engine as E
from gtn_applications_amd.criterions import asg as AS
B, T, C, L, N = 128, 1000, 100, 44, 200
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
Wt = torch.zeros(C + 1, C, device="cuda", requires_grad=True)
batches = [torch.randint(C - 2, (B, L), generator=g).tolist() for _ in range(N + 20)]
acc = collections.defaultdict(float)
def wrap(mod, name, label=None):
    fn = getattr(mod, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t0
    setattr(mod, name, w)
for n in ("targets_on_device", "loss_factors", "lattice_forward", "lattice_grad", "dense_forward", "dense_grad", "reduce_loss", "as_device_f32"):
    wrap(E, n)
wrap(E, "_stage_targets")
wrap(E._StagingRing, "next", "ring.next")
_oc = torch.Tensor.copy_
def _copy(self, *a, **k):
    t0 = time.perf_counter(); r = _oc(self, *a, **k); acc["Tensor.copy_"] += time.perf_counter() - t0; return r
torch.Tensor.copy_ = _copy
_oe = torch.empty
def _empty(*a, **k):
    t0 = time.perf_counter(); r = _oe(*a, **k); acc["torch.empty"] += time.perf_counter() - t0; return r
torch.empty = _empty
wrap(E.N.lib, "wfl_lattice_pack_asg_fal") if hasattr(E.N.lib, "__dict__") else None
orig_fal = E.PackedLattice.asg_force_align.__func__
def fal(cls, *a):
    t0 = time.perf_counter(); r = orig_fal(cls, *a); acc["asg_force_align"] += time.perf_counter() - t0; return r
E.PackedLattice.asg_force_align = classmethod(fal)
of, ob = AS.ASGLossFunction.forward, AS.ASGLossFunction.backward
def f2(ctx, *a):
    t0 = time.perf_counter(); r = of(ctx, *a); acc["Function.forward"] += time.perf_counter() - t0; return r
def b2(ctx, *a):
    t0 = time.perf_counter(); r = ob(ctx, *a); acc["Function.backward"] += time.perf_counter() - t0; return r
AS.ASGLossFunction.forward, AS.ASGLossFunction.backward = staticmethod(f2), staticmethod(b2)
def step(i):
    x.grad = None; Wt.grad = None
    t0 = time.perf_counter()
    loss = AS.ASGLossFunction.apply(x, Wt, batches[i], "mean")
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    acc["apply()"] += t1 - t0; acc["loss.backward()"] += t2 - t1
for mode in ("fresh", "same"):
    for i in range(20): step(i if mode == "fresh" else 3)
    torch.cuda.synchronize(); acc.clear()
    t0 = time.perf_counter()
    for i in range(20, 20 + N): step(i if mode == "fresh" else 3)
    tot = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(mode, "total host %.1f us/iter" % (tot / N * 1e6))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("   %-22s %7.1f us" % (k, v / N * 1e6))
