#!/usr/bin/env python
"""Benchmarks of the MI355X WFST loss engine on the BASELINE.json configurations.

Headline (default, what the driver runs): BASELINE.json's metric -- utterances/sec, forward+backward -- on
configs[1], the reference's CTC benchmark (benchmarks/ctc_benchmark.py:17-31: randn "log_probs", targets
randint(C-2), blank C-1, reduction "none") at T=1000, C=100, B=128, L=44 on one MI355X.

A "step" = one pass of the hot path over one batch whose inputs -- emissions AND targets -- are resident in HBM:
  --mode api (default)  the drop-in operator exactly as the reference's benchmark scripts call it --
                        `CTCLoss(x, targets, blank).backward()` (`ASGLoss(...)`, `Transducer(...)(x, targets)`),
                        autograd and host-side target handling included (benchmarks/ctc_benchmark.py:26-31).  This
                        is `value`;
  --emissions output (default)  the emissions handed to the criterion are NOT a leaf of the autograd graph: that is what
                        the reference's own benchmark hands over (ctc_benchmark.py:22: `randn(..., requires_grad=True)
                        .cuda()` is the output of a copy) and what every training loop does (train.py:262-266: a
                        model's output), so `loss.backward()` has a graph below the emissions to run.  The producer
                        here is `x.view_as(x)` of a device-resident leaf: a graph node with no kernel of its own, so
                        the step's bytes stay the criterion's;
  --emissions leaf      the emissions ARE the leaf (round 1-5's headline; reported as `leaf_emissions` next to `value`);
  --mode abi (ctc only; reported as `abi_kernels_only` next to the headline)  `wfl_ctc_forward_backward` through
                        the C ABI of include/wfl.h, targets staged on the device before the timed region: what the
                        GPU does.
  --targets same (default)  the reference benchmarks' own protocol (benchmarks/ctc_benchmark.py:26-31: one target
                        list reused by every iteration): after the first call the targets -- like the emissions --
                        are resident in HBM when a timed step starts (the engine's content-keyed staging cache);
  --targets fresh       every step (warm-up included) gets targets never seen before, so no content-keyed cache
                        of the engine can hit: per-batch host work (flattening, staging, the upload; for the
                        Transducer the whole graph algebra) is inside the timed region.
`value` is the operator with the reference benchmarks' protocol; the CTC module on raw scores (`module_raw_scores`), the
C-ABI step (`abi_kernels_only`, ctc) and the
cold-cache operator (`fresh_targets`) are reported next to it.

Per-kernel times come from HIP events recorded on the stream each launch goes to, inside the timed region
(engine.PHASE_EVENTS); `roofline.frac` is STEP level -- the batch's algorithmic bytes over the sum of all kernel
groups of a step (for CTC that is one launch + the repair launch) -- with the dominant kernel's own figure under
`roofline.dominant_kernel`; `traffic` from the committed PMC passes (the newest profiles/rNN_pmc_traffic.json,
collected with scripts/collect_round.sh on the same commands).

  python bench.py                                   # cfg2 CTC, 1 GPU, finishes in about a minute
  python bench.py --workload asg                    # cfg3
  python bench.py --workload transducer             # cfg4 (the reference's 1000 word pieces)
  python bench.py --config cfg5                     # one GPU's shard of cfg5 (T=2000, C=512, 128 utterances per GPU)
  python bench.py --gpus 8 --config cfg5            # cfg5 itself: spawns its own 8 ranks (train.py:344-347 does too)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W   # ... or under a launcher (RANK / WORLD_SIZE set)

Multi-GPU: utterances shard over ranks (same per-GPU batch: weak scaling), no emissions cross GPUs.  CTC has no
learnable transition weights, hence no data-path collective at cfg2; --workload asg averages the transition-weight
gradient over ranks each step (parallel.sync_transition_grads, RCCL all-reduce); --config cfg5 all-reduces a
[(C+1), C] fp32 buffer per step through parallel.all_reduce_mean_ although pure CTC has nothing to exchange
(BASELINE configs[4] names "RCCL all-reduce of transition grads", so the harness exercises the
collective with the payload an ASG criterion of that size would have) and says so in config.parallelism.  Timing:
barrier + synchronize on both sides, MAX over ranks.  Without a launcher `--gpus N` (N > 1) starts N worker
processes itself, one device each, rendezvous on 127.0.0.1.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FORCE_DIST = os.environ.get("WFL_BENCH_FORCE_DIST") == "1"  # one-GPU boxes: the collective code path at world size 1
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
VALU_F32_PEAK_TFLOPS = 157.3  # fp32 vector peak (MI355X_MICROARCH.md)

# which kernels run inside each timed phase of engine.PHASE_EVENTS (short names as rocprofv3 reports them)
PHASE_KERNEL_NAMES = {
    "ctc_step": ["ctc_mitm_kernel", "ctc_repair_kernel"],
    "ctc_chains": ["ctc_log_chain_kernel"],
    "ctc_grad": ["reduce_loss_kernel", "ctc_grad_kernel"],
    "lattice_gather": ["gather_lse_kernel", "gather_kernel"],
    "lattice_chain": ["prob_chain_kernel", "prob_chain_pub_kernel", "occ_gate_kernel", "occ_live_kernel",
                      "prob_certify_kernel", "chain_kernel"],  # (occ_*: the gradient that runs beside the sweeps)
    "lattice_grad": ["occ_grad_kernel", "band_grad_kernel", "grad_kernel"],
    "lattice_gather/shared": ["gather_kernel"],
    "lattice_chain/shared": ["prob_chain_kernel", "prob_certify_kernel", "chain_kernel"],
    "lattice_grad/shared": ["grad_kernel"],
    "dense_chain": ["dense_fast_chain_kernel", "dense_chain_kernel", "wide_resident_sweep_kernel", "wide_frame_mfma_kernel",
                    "wide_rows_kernel", "wide_prep_kernel", "wide_scan_kernel"],
    "dense_grad": ["dense_mfma_grad_kernel", "dense_fast_grad_kernel", "dense_grad_kernel", "dense_reduce_kernel",
                   "wide_grad_x_kernel", "wide_grad_w_kernel", "wide_reduce_w_kernel"],
}
PHASE_KERNELS = {k: " + ".join(v) + (" (transitions graph)" if k.endswith("/shared") else "")
                 for k, v in PHASE_KERNEL_NAMES.items()}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ctc", choices=["ctc", "asg", "transducer"])
    ap.add_argument("--config", default=None, choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="a BASELINE.json configuration by name (sets workload and shape; cfg5 adds the all-reduce)")
    ap.add_argument("--mode", default="api", choices=["abi", "api"],
                    help="api: the drop-in operator (headline); abi: the C-ABI call with pre-staged targets (ctc only)")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="tests only: gloo ranks on the CPU and a stub step -- exercises the rank / timing plumbing, measures nothing")
    ap.add_argument("--targets", default="same", choices=["fresh", "same"])
    ap.add_argument("--emissions", default="output", choices=["output", "leaf"],
                    help="output: the criterion's input is a producer's output (x.view_as(x); headline); leaf: it is the leaf itself")
    ap.add_argument("--B", type=int, default=None)
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--C", type=int, default=None)
    ap.add_argument("--L", type=int, default=44)
    ap.add_argument("--ctc-step", default="pipelined", choices=["split", "pipelined"],
                    help="abi CTC step: forward and gradient kernels back to back, or one pipelined launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the labelled extra measurements (profiling runs)")
    ap.add_argument("--cpu-sample-utts", type=int, default=None)
    args = ap.parse_args()
    if args.config:
        shape = {"cfg2": ("ctc", 128, 1000, 100), "cfg3": ("asg", 128, 1000, 100), "cfg4": ("transducer", 64, 800, None),
                 "cfg5": ("ctc", 128, 2000, 512)}[args.config]
        args.workload = shape[0]
        args.B, args.T, args.C = args.B or shape[1], args.T or shape[2], args.C or shape[3]
    return args


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: start N copies of this command, one rank per device, and wait
    (the reference spawns its own ranks too: train.py:344-347).  Rank 0 prints the JSON line."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def dist_setup(n, stub_cpu=False):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(n, 1) and "WORLD_SIZE" in os.environ:
        raise SystemExit(f"bench.py: --gpus {n} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1 or os.environ.get("WFL_BENCH_FORCE_DIST"):  # (the variable: RCCL plumbing on a 1-GPU box, world size 1)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if stub_cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            return rank, world, local, dist
        if torch.cuda.device_count() <= local:
            raise SystemExit(f"bench.py: rank {rank} wants cuda:{local}, {torch.cuda.device_count()} device(s) visible")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        return rank, world, local, dist
    if not stub_cpu:
        torch.cuda.set_device(0)
    return 0, 1, 0, None


# --------------------------------------------------------------------------------------------------
# workloads: each returns dict(step=fn(i), meta=..., payload=for the CPU baseline, [abi_step]).
# step(i) runs forward+backward of batch i; with fresh targets batch i has targets no earlier step saw.
# --------------------------------------------------------------------------------------------------
def make_ctc(args, rank, n_batches, dist=None):
    from gtn_applications_amd import _native as N
    from gtn_applications_amd import engine as E
    from gtn_applications_amd import parallel
    from gtn_applications_amd.criterions import ctc

    B, T, C, L = args.B or 128, args.T or 1000, args.C or 100, args.L
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(B, T, C, generator=g).cuda()
    blank = C - 1
    # benchmarks/ctc_benchmark.py:23-24: randint(N - 2, (B, L)) as a list of int lists
    batches = [torch.randint(C - 2, (B, L), generator=g).tolist() for _ in range(n_batches)]
    xr = x.clone().requires_grad_(True)
    # cfg5 (BASELINE configs[4]): "RCCL all-reduce of transition grads".  Pure CTC has none, so the harness averages the
    # buffer an ASG criterion of this size would exchange -- [(C+1), C] fp32, 1.05 MB at C = 512 -- once per step
    exchange = torch.zeros(C + 1, C, device=x.device) if args.config == "cfg5" else None

    # the headline protocol: the emissions are a producer's OUTPUT (ctc_benchmark.py:22, train.py:262-266), so
    # loss.backward() has the graph below them to run on the autograd engine
    def step(i):
        xr.grad = None
        ctc.CTCLoss(xr.view_as(xr), batches[i % n_batches], blank).backward()
        if exchange is not None:
            parallel.all_reduce_mean_([exchange], force=FORCE_DIST)

    def leaf_step(i):  # (the emissions ARE the leaf: the gradient of the forward launch becomes x.grad, no engine)
        xr.grad = None
        ctc.CTCLoss(xr, batches[i % n_batches], blank).backward()
        if exchange is not None:
            parallel.all_reduce_mean_([exchange], force=FORCE_DIST)

    # ... with a producer that has kernels of its own (x * 1.0 and its backward: two elementwise passes over [B,T,C])
    def engine_step(i):
        xr.grad = None
        ctc.CTCLoss(xr * 1.0, batches[i % n_batches], blank).backward()

    def engine_proper_step(i):  # torch.Tensor.backward from the loss down (the criterion's short cut switched off)
        xr.grad = None
        old, ctc._FAST_BACKWARD = ctc._FAST_BACKWARD, False
        try:
            ctc.CTCLoss(xr.view_as(xr), batches[i % n_batches], blank).backward()
        finally:
            ctc._FAST_BACKWARD = old

    # the CTC MODULE (criterions/ctc.py:99-121, use_pt=False): raw scores in, log_softmax fused into the step -- what a
    # training loop calls; reported next to the headline as `module_raw_scores`
    module = ctc.CTC(blank, False)
    module_targets = [torch.tensor(t) for t in batches[0]]

    def module_step(i):
        xr.grad = None
        module(xr, module_targets).backward()

    # ... and the step train.py:262-279 runs: the MODULE on a model's output (raw scores that are not a leaf), targets it
    # has never seen (a new batch every step, as tensors): nothing the reference benchmark's protocol lets a cache hit
    # (as a DataLoader hands them over: built outside the timed region)
    fresh_tensors = [[torch.tensor(t) for t in bt] for bt in batches]

    def training_step(i):
        xr.grad = None
        module(xr * 1.0, fresh_tensors[(1 + i % max(1, n_batches - 1)) % n_batches]).backward()

    def viterbi_step(i):  # (train.py:279 decodes every training batch for its error rate)
        module.viterbi(xr.detach())

    # the C-ABI call underneath, targets pre-staged (kernels only)
    dev = x.device
    tg = E.targets_on_device(batches[0], dev)
    scale, _, coef = E.loss_factors(tg, "none")
    gout = torch.ones(1, device=dev)
    dx = torch.empty_like(x)
    chain_flags = E.CTC_DEFAULT_FLAGS
    last = [None]
    if args.ctc_step == "pipelined":
        def abi_step(i):
            last[0] = E.ctc_forward_backward(x, tg, blank, coef, gout, dx, loss_scale=scale, want_loss=True)
            if exchange is not None:
                parallel.all_reduce_mean_([exchange], force=FORCE_DIST)
    else:
        def abi_step(i):
            tok = E._mark("ctc_chains")
            ws, nll = E.ctc_forward(x, tg, blank, chain_flags)
            E._done(tok)
            tok = E._mark("ctc_grad")
            E.reduce_loss(nll, scale, 1.0)
            E.ctc_grad(x, tg, blank, ws, nll, coef, gout, dx)
            E._done(tok)

    def launch_clock(reps=24):
        """Duration of the meet-in-the-middle launch ALONE, measured on the device: the constant 100 MHz clock at the entry
        of every workgroup and at the exit of its last wave (workspace field WFL_CTC_WS_CLOCK; the HIP-event bracket of
        the step also holds the repair launch behind it).  Median over `reps` launches, each synchronised."""
        if args.ctc_step != "pipelined":
            return None
        spans = []
        for _ in range(reps):
            ws, _nll, _loss = E.ctc_forward_backward(x, tg, blank, coef, gout, dx, loss_scale=scale, want_loss=True)
            torch.cuda.synchronize()
            clk = E.ctc_workspace_field(ws, B, T, tg.max_len, N.CTC_WS_CLOCK).view(torch.int64).view(2 * B, 2).cpu().numpy()
            if (clk[:, 1] <= clk[:, 0]).any():
                return None  # (a launch that did not go through the meet-in-the-middle kernel)
            spans.append((int(clk[:, 1].max()) - int(clk[:, 0].min())) * 1e-5)  # 10 ns ticks -> ms
        return float(np.median(spans))

    def repaired():  # utterances of the last abi step that the certificate sent to the log-domain repair launch
        return E.ctc_pipeline_repaired(last[0][0], B, T, tg.max_len) if last[0] is not None else None

    which = {(1000, 100, 128, 44): " (BASELINE configs[1])", (2000, 512, 128, 44): " (BASELINE configs[4], one GPU's shard)",
             (150, 28, 8, 44): " (BASELINE configs[0])"}.get((T, C, B, L), "")
    key = {(1000, 100, 128, 44): "cfg2", (2000, 512, 128, 44): "cfg5", (1000, 100, 1024, 44): "cfg2_B1024"}.get((T, C, B, L))
    meta = dict(workload=f"ctc fwd+bwd T={T} C={C} B={B} L={L}{which}", B=B, T=T, C=C, L=L, key=key, repaired=repaired,
                exchange_bytes=0 if exchange is None else exchange.numel() * 4,
                metric=f"utterances/sec fwd+bwd (ctc_benchmark T={T},C={C},B={B}); HBM GB/s vs peak",
                call="CTCLoss(x, targets, blank).backward()", algorithmic_bytes_per_utt=8 * T * C)
    return dict(step=step, leaf_step=leaf_step, abi_step=abi_step, module_step=module_step, engine_step=engine_step,
                engine_proper_step=engine_proper_step, meta=meta, launch_clock=launch_clock, training_step=training_step, viterbi_step=viterbi_step,
                payload=("ctc", x, batches[0], blank))


def make_asg(args, rank, n_batches, dist):
    from gtn_applications_amd import parallel
    from gtn_applications_amd.criterions import asg

    B, T, C, L = args.B or 128, args.T or 1000, args.C or 100, args.L
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)

    # asg_benchmark.py:20-22: a free (C+1) x C transitions tensor with requires_grad (here a leaf ON the device)
    transitions = torch.randn(C + 1, C, generator=torch.Generator().manual_seed(7)).cuda().requires_grad_(True)
    batches = [torch.randint(C - 2, (B, L), generator=g).tolist() for _ in range(n_batches)]

    def step(i):  # (emissions that are a producer's output: asg_benchmark.py:20 hands over `.cuda()` of a host leaf)
        x.grad = None
        transitions.grad = None
        asg.ASGLoss(x.view_as(x), transitions, batches[i % n_batches]).backward()
        if dist is not None:  # the one exchange step of the path (train.py:205-208: DDP averages criterion grads)
            parallel.all_reduce_mean_([transitions.grad], force=FORCE_DIST)

    def leaf_step(i):
        x.grad = None
        transitions.grad = None
        asg.ASGLoss(x, transitions, batches[i % n_batches]).backward()
        if dist is not None:
            parallel.all_reduce_mean_([transitions.grad], force=FORCE_DIST)

    # a training loop's shape (train.py:205-208,262-266): emissions that are a model's output (non-leaf) and transitions
    # that are the ASG module's nn.Parameter -- both keep loss.backward() on the autograd engine
    par = torch.nn.Parameter(transitions.detach().clone())

    def engine_step(i):
        x.grad = None
        par.grad = None
        asg.ASGLoss(x * 1.0, par, batches[i % n_batches]).backward()

    def training_step(i):  # non-leaf emissions, nn.Parameter transitions, a batch of targets never seen before
        x.grad = None
        par.grad = None
        asg.ASGLoss(x * 1.0, par, batches[(1 + i % max(1, n_batches - 1)) % n_batches]).backward()

    vit_module = asg.ASG(C - 2, 1, True) if C > 2 else None  # (C classes: tokens + one replabel + garbage)
    if vit_module is not None:
        vit_module.transitions.data = transitions.detach().clone()

    def viterbi_step(i):  # (train.py:279 decodes every training batch for its error rate)
        vit_module.viterbi(x.detach())

    which = " (BASELINE configs[2])" if (T, C, B, L) == (1000, 100, 128, 44) else ""
    meta = dict(workload=f"asg fwd+bwd T={T} C={C} B={B} L={L}{which}", B=B, T=T, C=C, L=L,
                key="cfg3" if which else None,
                metric=f"utterances/sec fwd+bwd (asg_benchmark T={T},C={C},B={B}); HBM GB/s vs peak",
                call="ASGLoss(x, transitions, targets).backward()",
                algorithmic_bytes_per_utt=8 * T * C, algorithmic_bytes_per_batch=8 * (C + 1) * C)
    wl = dict(step=step, leaf_step=leaf_step, engine_step=engine_step, training_step=training_step, meta=meta,
              payload=("asg", x.detach(), transitions.detach(), batches[0]))
    if vit_module is not None:
        wl["viterbi_step"] = viterbi_step
    return wl


def word_pieces():
    """benchmarks/word_pieces_tokens_1000.txt (the reference's data file, shipped as a fixture): 1000 pieces,
    78 graphemes (transducer_benchmark.py:19-23)."""
    with open(os.path.join(ROOT, "benchmarks", "word_pieces_tokens_1000.txt"), "r") as fid:
        tokens = sorted(l.strip() for l in fid)
    graphemes = sorted(set(c for t in tokens for c in t))
    return tokens, {t: i for i, t in enumerate(graphemes)}


def make_transducer(args, rank, n_batches):
    from gtn_applications_amd.criterions import transducer

    B, T, Lp = args.B or 64, args.T or 800, 15
    tokens, g2i = word_pieces()
    C = len(tokens) + 1
    rnd = random.Random(rank)
    x = torch.randn(B, T, C, generator=torch.Generator().manual_seed(rank)).cuda().requires_grad_(True)
    # transducer_benchmark.py:36-40: 15 random pieces per sample, spelled out in graphemes, as tensors
    batches = [[torch.tensor([g2i[ch] for _ in range(Lp) for ch in rnd.choice(tokens)]) for _ in range(B)]
               for _ in range(n_batches)]
    crit = transducer.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")

    def step(i):  # (emissions that are a producer's output: transducer_benchmark.py:33 hands over `.cuda()` of a host leaf)
        x.grad = None
        crit(x.view_as(x), batches[i % n_batches]).backward()

    def leaf_step(i):
        x.grad = None
        crit(x, batches[i % n_batches]).backward()

    def engine_step(i):  # (a producer with kernels of its own, as under train.py:262-266)
        x.grad = None
        crit(x * 1.0, batches[i % n_batches]).backward()

    def training_step(i):  # non-leaf emissions, a batch of targets never seen before
        x.grad = None
        crit(x * 1.0, batches[(1 + i % max(1, n_batches - 1)) % n_batches]).backward()

    # ... and the same with the NEXT batch's targets handed to criterion.prepare() while this step is queued: their graph
    # algebra, packing and upload run on a side thread / stream (what a prefetching loader does for the inputs)
    pending = {}

    def fresh_prepared_step(i):  # (the `fresh_targets` protocol -- leaf emissions, batch 1 + i -- with prepare() one step ahead)
        x.grad = None
        k = (1 + i) % n_batches
        cur = pending.pop(("f", k), None) or crit.prepare(batches[k])
        loss = crit(x, cur)
        pending[("f", (2 + i) % n_batches)] = crit.prepare(batches[(2 + i) % n_batches])
        loss.backward()

    def training_step_prepared(i):
        x.grad = None
        k = (1 + i % max(1, n_batches - 1)) % n_batches
        cur = pending.pop(k, None) or crit.prepare(batches[k])
        loss = crit(x * 1.0, cur)
        kn = (1 + (i + 1) % max(1, n_batches - 1)) % n_batches
        pending[kn] = crit.prepare(batches[kn])
        loss.backward()

    which = " (BASELINE configs[3])" if (T, B) == (800, 64) else ""
    meta = dict(workload=f"transducer fwd+bwd, 1000 word pieces (word_pieces_tokens_1000.txt) T={T} C={C} B={B}{which}",
                B=B, T=T, C=C, L=Lp, key="cfg4" if which else None,
                metric=f"utterances/sec fwd+bwd (transducer_benchmark word decompositions T={T},C={C},B={B}); HBM GB/s vs peak",
                call="Transducer(tokens, ..., blank='optional', allow_repeats=False, reduction='mean')(x, targets).backward()",
                algorithmic_bytes_per_utt=8 * T * C)
    def viterbi_step(i):  # (train.py:279 decodes every training batch for its error rate)
        crit.viterbi(x.detach())

    return dict(step=step, leaf_step=leaf_step, engine_step=engine_step, training_step=training_step,
                training_step_prepared=training_step_prepared, fresh_prepared_step=fresh_prepared_step, viterbi_step=viterbi_step,
                meta=meta, payload=("transducer", x.detach(), crit, batches[0]))


# --------------------------------------------------------------------------------------------------
# CPU baselines (oracle/cpu_ref.c, "port"), bounded samples, rank 0 at N=1 only
# --------------------------------------------------------------------------------------------------
def _timed_reps(fn, budget_s):
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            return reps, el


def cpu_baseline(payload, n_utts):
    from oracle import cpu_ref

    cores = os.cpu_count() or 1
    kind = payload[0]
    if kind == "ctc":
        _, x, targets, blank = payload
        n = min(n_utts or 128, x.shape[0])
        xs, tg = x[:n].cpu().numpy(), targets[:n]
        run = lambda: cpu_ref.ctc_cpu(xs, tg, blank, "none", cores)  # noqa: E731
        what = "graph-faithful CTC (materialised emissions x label-graph lattice)"
    elif kind == "asg":
        _, x, W, targets = payload
        n = min(n_utts or 128, x.shape[0])
        xs, Ws, tg = x[:n].cpu().numpy(), W.cpu().numpy(), targets[:n]
        run = lambda: cpu_ref.asg_cpu(xs, Ws, tg, "none", cores)  # noqa: E731
        what = "graph-faithful ASG (per-sample transitions graph, T*C^2-arc denominator lattice)"
    else:
        from gtn_applications_amd.criterions import transducer as TR

        _, x, crit, targets = payload
        n = min(n_utts or 64, x.shape[0])
        xs = x[:n].cpu().numpy()
        crit.tokens.arc_sort(True)
        accs, scales = [], []
        for t in targets[:n]:  # alignment acceptors: built ONCE outside the timed region (not part of the port)
            a = TR._alignment_graph(t.tolist(), crit.tokens, crit.lexicon, None)[0].arrays()
            accs.append(dict(src=a["src"], dst=a["dst"], lab=a["ilabel"], start=a["start"], accept=a["accept"]))
            scales.append(1.0 / max(1, t.numel()))
        run = lambda: cpu_ref.lattice_cpu(xs, accs, scales, True, cores)  # noqa: E731
        what = ("log_softmax + forward_score/backward over emissions x alignment acceptor; the per-sample graph algebra "
                "(compose/remove/project, transducer.py:265-276) is EXCLUDED from the port, i.e. it is faster than the real path")
    run()  # warm-up (page in, thread start)
    reps, el = _timed_reps(run, 10.0)
    return dict(value=n * reps / el, unit="utt/s", cores=cores, kind="port",
                sample=f"{reps} x fwd+bwd of {n} utterances of the same workload with oracle/cpu_ref.c on {cores} threads, "
                       f"{el:.1f} s: {what}")


def cpu_torch_ctc(payload):
    """torch.nn.functional.ctc_loss on the host cores -- the reference's own alternative CTC path
    (criterions/ctc.py:109-121) -- on the same batch, forward + backward, for context."""
    _, x, targets, blank = payload
    xc = x.detach().cpu().requires_grad_(True)
    B, T, _ = xc.shape
    tg = torch.tensor(targets, dtype=torch.long)
    lens = torch.full((B,), tg.shape[1], dtype=torch.long)

    def run():
        xc.grad = None
        torch.nn.functional.ctc_loss(xc.permute(1, 0, 2), tg, torch.full((B,), T, dtype=torch.long), lens,
                                     blank=blank, reduction="sum").backward()

    run()
    reps, el = _timed_reps(run, 3.0)
    return dict(value=B * reps / el, unit="utt/s", threads=torch.get_num_threads(),
                what=f"torch.nn.functional.ctc_loss fwd+bwd on CPU, {reps} x {B} utterances in {el:.1f} s")


def pmc_traffic(key, phases):
    """HBM bytes per step of the kernels of `phases` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate `--pmc` runs of this same command, corrected with the factors measured by
    scripts/pmc_calib.hip; see profiles/README.md).  Only for the configurations they were measured on."""
    import glob
    # the newest round's passes first (profiles/rNN_pmc_traffic.json)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True):
        if key is None:
            continue
        with open(path) as f:
            rec = json.load(f).get("configs", {}).get(key, {}).get("kernels", {})
        names = {n for ph in phases for n in PHASE_KERNEL_NAMES.get(ph.split("/")[0], [])}
        tot = [rec[n]["hbm_bytes"] for n in names if n in rec]
        if tot:
            PMC_SOURCE[0] = os.path.basename(path)
            return float(sum(tot))
    return None


PMC_SOURCE = [None]  # which committed file `traffic` came from (printed in the line)


# --------------------------------------------------------------------------------------------------
def timed_loop(step, steps, warmup, fence, collect_events):
    """-> (seconds of the timed region, ms per step of the dominant kernel group measured INSIDE it, ms per step of every
    group measured during warm-up).  HIP events around every launch perturb the step (each record is a packet on the
    stream, ~5 us on the critical path), so the timed region brackets only the dominant group -- found during the
    warm-up steps, which bracket them all -- and only three of its launches, a third of the run apart (measured at the
    CTC benchmark, 20 steps: every 4th launch bracketed 0.0472-0.0486 ms a step, every 10th 0.0455-0.0458).  (The
    cyclic garbage collector is left alone: a collection right before the timed region made the 20 steps after it 15 us
    slower each, switching it off for the region leaves every step's cycles and their device buffers alive.)"""
    from gtn_applications_amd import engine as E

    def collect(events, n):
        phases = {}
        for name, a, b in events or []:
            phases.setdefault(name, []).append(a.elapsed_time(b))
        # ms per step: a phase may launch more than once per step (numerator + denominator)
        return {k: float(np.sum(v)) / n for k, v in phases.items()}

    cal = min(warmup, 3) if collect_events else 0
    for i in range(warmup - cal):
        step(i)
    fence()
    warm = {}
    if cal:
        E.prealloc_events(16 * (steps + cal))
        E.PHASE_EVENTS, E.PHASE_ONLY = [], None
        for i in range(warmup - cal, warmup):
            step(i)
        fence()
        warm = collect(E.PHASE_EVENTS, cal)
    # (--warmup 0: nothing to calibrate on, every group is bracketed inside the timed region)
    E.PHASE_EVENTS = [] if collect_events else None
    E.PHASE_ONLY = {max(warm, key=warm.get)} if warm else None
    # bracket three launches of the dominant group in the timed region (every launch of a run of fewer than four steps)
    E.PHASE_STRIDE = max(1, steps // 3) if warm else 1
    E._PHASE_COUNT.clear()
    if collect_events and not warm:
        E.prealloc_events(16 * steps)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    events, E.PHASE_EVENTS, E.PHASE_ONLY = E.PHASE_EVENTS, None, None
    stride, E.PHASE_STRIDE = E.PHASE_STRIDE, 1
    timed = collect(events, max(1, steps // stride))
    return elapsed, {**warm, **timed}


def stub_main(args):
    """--stub-cpu (tests/test_parallel.py): the rank / barrier / MAX-over-ranks plumbing of main() with gloo ranks
    and a stub step.  Measures nothing."""
    rank, world, local, dist = dist_setup(args.gpus, stub_cpu=True)
    buf = torch.zeros(8)

    def fence():
        if dist is not None:
            dist.barrier()

    def step(i):
        buf.add_(1.0)
        if dist is not None:
            from gtn_applications_amd import parallel

            parallel.all_reduce_mean_([buf])
        time.sleep(0.001 * (1 + rank))  # ranks differ: the reported time must be the slowest rank's

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = own = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": world * 1 * args.steps / elapsed, "unit": "utt/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
                          "rank0_ms_per_step": own * 1e3 / args.steps, "config": {"workload": "stub (cpu, gloo)"},
                          "collective_world_size": dist.get_world_size() if dist is not None else 1}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    if args.stub_cpu:
        return stub_main(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    rank, world, local, dist = dist_setup(args.gpus)
    extras_steps = max(10, args.steps // 2)
    fresh_extra = world == 1 and not args.no_extras and (args.mode == "abi" or args.targets == "same")
    n_batches = (args.steps + args.warmup) if args.targets == "fresh" else 1 + (extras_steps + 3 if fresh_extra else 0)
    if args.workload == "ctc":
        wl = make_ctc(args, rank, n_batches, dist)
    elif args.workload == "asg":
        wl = make_asg(args, rank, n_batches, dist)
    else:
        wl = make_transducer(args, rank, n_batches)
    meta = wl["meta"]
    if args.mode == "abi" and "abi_step" not in wl:
        raise SystemExit("--mode abi exists for --workload ctc only")

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    api_step = wl["leaf_step"] if args.emissions == "leaf" else wl["step"]
    step = wl["abi_step"] if args.mode == "abi" else api_step if args.targets == "fresh" else (lambda i: api_step(0))
    elapsed, phase_ms = timed_loop(step, args.steps, args.warmup, fence, True)
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    B = meta["B"]
    ms = elapsed * 1e3 / args.steps
    value = world * B * args.steps / elapsed
    repaired = meta["repaired"]() if args.mode == "abi" else None
    if meta.get("exchange_bytes"):
        par = (f"dp{world} (utterance shards; all-reduce(mean) of a [(C+1), C] fp32 buffer per step, {meta['exchange_bytes']} bytes, "
               f"through parallel.all_reduce_mean_ -- pure CTC has no transition weights: the payload is the one an ASG "
               f"criterion of this size would exchange, SURVEY.md 8(e))")
    elif args.workload == "asg":
        par = f"dp{world} (utterance shards; all-reduce(mean) of the transition-weight gradient per step)"
    else:
        par = f"dp{world} (utterance shards, no data-path collective)"
    out = {
        "metric": meta["metric"], "value": value, "unit": "utt/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": meta["workload"], "mode": args.mode, "targets": ("pre-staged" if args.mode != "api" else "fresh (new targets every step)" if args.targets == "fresh" else
                               "same list every step (the reference benchmark's protocol: resident on the device after the first call)"),
                   "timed_call": meta["call"] if args.mode == "api" else "wfl_ctc_forward_backward (C ABI, targets pre-staged)",
                   "emissions": ("n/a (C ABI)" if args.mode != "api" else
                                 "the leaf itself (loss.backward() hands the gradient to x.grad)" if args.emissions == "leaf" else
                                 "a producer's output, not a leaf (x.view_as(x) of a device-resident leaf; ctc_benchmark.py:22 and "
                                 "train.py:262-266 hand over non-leaf emissions): loss.backward() runs the graph below them on the "
                                 "autograd engine"),
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": par,
                   "collective_world_size": dist.get_world_size() if dist is not None else 1},
    }
    if repaired is not None:
        out["config"]["utterances_repaired_in_log_domain"] = repaired
    alg_bytes = meta["algorithmic_bytes_per_utt"] * B + meta.get("algorithmic_bytes_per_batch", 0)
    if phase_ms:
        dom = max(phase_ms, key=phase_ms.get)
        sum_ms = float(sum(phase_ms.values()))
        # groups on forked streams overlap (ASG: numerator under the denominator's sweeps; Transducer: the gradient
        # beside the sweeps): their sum exceeds the step.  The step's GPU time is then bounded by its wall time.
        forked = args.workload in ("asg", "transducer")
        gpu_ms = ms if forked else (min(sum_ms, ms) if args.mode == "api" else sum_ms)
        basis = ("wall time of the step (its kernel groups run on forked streams and overlap: neither their sum nor any one "
                 "of them is the step)" if forked else
                 "wall time of the step (smaller than the sum of its bracketed kernel groups: the brackets cost)" if gpu_ms < sum_ms
                 else "sum of the step's kernel groups (HIP events)")
        step_achieved = alg_bytes / (gpu_ms * 1e-3) / 1e9
        dom_achieved = alg_bytes / (phase_ms[dom] * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": " + ".join(PHASE_KERNELS.get(k, k) for k in sorted(phase_ms, key=lambda k: -phase_ms[k])),
            "achieved": step_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": step_achieved / HBM_PEAK_GBPS,
            "traffic": pmc_traffic(meta["key"], list(phase_ms)),
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms": {PHASE_KERNELS.get(k, k): v for k, v in sorted(phase_ms.items(), key=lambda kv: -kv[1])},
            "step_kernels_ms": sum_ms, "time_basis": basis, "step_ms_used": gpu_ms,
            "dominant_kernel": {"kernel": PHASE_KERNELS.get(dom, dom), "ms": phase_ms[dom], "achieved": dom_achieved,
                                "frac": dom_achieved / HBM_PEAK_GBPS},
            "traffic_source": PMC_SOURCE[0],
            "note": "STEP level: achieved = algorithmic bytes of the batch (8*T*C per utterance, + 8*(C+1)*C for ASG's W and "
                    "dW) / the sum of the average durations of ALL kernel groups of a step, HIP events on the launch stream "
                    "(the dominant group inside the timed region, the others during the last warm-up steps only: an event "
                    "pair per launch costs the step ~10 us each; groups on forked streams overlap, so the sum is an upper "
                    "bound of the step's GPU time); dominant_kernel divides the same bytes by that group alone; traffic = "
                    "HBM bytes of all the step's kernels from the committed PMC passes",
        }
        # (not under --no-extras: the profiling runs want the timed steps' launches and nothing else in their statistics)
        if args.workload == "ctc" and rank == 0 and "launch_clock" in wl and not args.no_extras:
            # the single longest kernel by itself: the group's bracket also holds the (normally empty) repair launch
            lc = wl["launch_clock"]()
            if lc:
                out["roofline"]["dominant_kernel"] = {
                    "kernel": "ctc_mitm_kernel", "ms": lc, "achieved": alg_bytes / (lc * 1e-3) / 1e9,
                    "frac": alg_bytes / (lc * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "how": "device clock (100 MHz) from the first workgroup's entry to the last wave's exit, median of 24 "
                           "synchronised launches after the timed region; the step's HIP-event bracket above also holds "
                           "ctc_repair_kernel"}
        if args.workload == "asg":
            # the dense sweep is 2*B*T*C^2 multiply-adds (alpha and beta) + as many for the gradient
            fma = 2.0 * B * meta["T"] * meta["C"] ** 2
            out["roofline"]["valu"] = {"bound": "valu_f32", "achieved": 2 * fma / (gpu_ms * 1e-3) / 1e12, "peak": VALU_F32_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": 2 * fma / (gpu_ms * 1e-3) / 1e12 / VALU_F32_PEAK_TFLOPS,
                                       "note": "2*B*T*C^2 fused multiply-adds of the dense forward + backward sweeps (2 flop each) "
                                               "over the step's kernel time, against the fp32 vector peak: the sweeps are "
                                               "dependent chains of small matrix-vector products in a scaled semiring on the "
                                               "vector pipes"}
            # the transition gradient's outer products run on the matrix cores (dense_mfma_grad_kernel:
            # v_mfma_f32_16x16x4_f32, fp32 in / fp32 accumulate -- exact fp32, at the vector rate on this chip): a
            # deviation from north_star's "no MFMA", stated here because it IS a contraction (dW = sum_t a_t u_t^T)
            mfma_ms = next((v for k, v in phase_ms.items() if k == "dense_grad"), None)
            out["roofline"]["mfma"] = {
                "bound": "mfma_f32", "kernel": "dense_mfma_grad_kernel (dW = sum over frames of outer products, 2*B*T*C^2 flop) "
                "+ dense_reduce_kernel", "achieved": (2 * fma / (mfma_ms * 1e-3) / 1e12) if mfma_ms else None,
                "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (2 * fma / (mfma_ms * 1e-3) / 1e12 / VALU_F32_PEAK_TFLOPS) if mfma_ms else None,
                "note": "fp32-input MFMA peak = the fp32 vector peak on MI355X (157.3 TFLOP/s); the kernel is bound by its "
                        "A / U / dx streams, not by the matrix cores"}
    single = rank == 0 and world == 1
    if single and not args.no_extras:
        if args.mode == "abi":
            # the drop-in operator on the same workload, the reference benchmark's protocol (same target list every step)
            el, _ = timed_loop(lambda i: api_step(0), extras_steps, 3, fence, False)
            out["python_api"] = {"value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                                 "what": meta["call"] + ": autograd operator, eager, host overhead included"}
        if fresh_extra:
            # cold cache: targets never seen before in every step (batches 1.. of the workload; batch 0 was the main run's)
            el, _ = timed_loop(lambda i: api_step(1 + i), extras_steps, 3, fence, False)
            out["fresh_targets"] = {"value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                                    "what": "operator path, new targets in every step: per-batch host work (flattening, "
                                            "staging and upload; Transducer: the graph algebra) inside the timed region, "
                                            "no content-keyed cache can hit"}
        if fresh_extra and "training_step" in wl and args.mode == "api":
            el, _ = timed_loop(wl["training_step"], extras_steps, 3, fence, False)
            out["training_step"] = {
                "value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                "what": "what a training loop gets (train.py:262-279): emissions that are a model's OUTPUT (x * 1.0: not a "
                        "leaf, so loss.backward() runs the autograd engine, incl. that product and its backward) AND a batch "
                        "of targets never seen before in every step (no content-keyed cache can hit)" +
                        ("; the CTC module on raw scores (log_softmax fused), targets as tensors" if args.workload == "ctc" else
                         "; transitions as an nn.Parameter" if args.workload == "asg" else "")}
        if "viterbi_step" in wl and args.mode == "api":
            nv = max(3, min(extras_steps, 20))
            el, _ = timed_loop(wl["viterbi_step"], nv, 2, fence, False)
            out["viterbi"] = {
                "ms_per_call": el * 1e3 / nv,
                "what": "criterion.viterbi(outputs) on the same batch, as train.py:279 calls it in every training step: "
                        "device decode, copy to the host, collapse / unpack there (not part of `value`)"}
        if fresh_extra and "fresh_prepared_step" in wl and args.mode == "api":
            el, _ = timed_loop(wl["fresh_prepared_step"], extras_steps, 3, fence, False)
            out["fresh_targets_prepared"] = {
                "value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                "what": "fresh_targets with criterion.prepare() called one step ahead (Transducer.prepare: the batch's graph "
                        "algebra, packing and upload on a side thread and stream)"}
        if fresh_extra and "training_step_prepared" in wl and args.mode == "api":
            el, _ = timed_loop(wl["training_step_prepared"], extras_steps, 3, fence, False)
            out["training_step_prepared"] = {
                "value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                "what": "training_step with criterion.prepare(next batch's targets) called while the current step is queued: "
                        "the per-batch graph algebra, packing and upload on a side thread and stream (Transducer.prepare)"}
        if args.mode == "api" and args.targets == "fresh":
            # the reference benchmarks' own protocol: the same target list every iteration
            el, _ = timed_loop(lambda i: api_step(0), extras_steps, 3, fence, False)
            out["same_targets"] = {"value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                                   "what": "operator path, one target list reused by every iteration (the reference "
                                           "benchmark scripts' protocol; content-keyed host caches hit)"}
        if args.mode == "api" and "leaf_step" in wl:
            other = wl["step"] if args.emissions == "leaf" else wl["leaf_step"]
            el, _ = timed_loop(lambda i: other(0), extras_steps, 3, fence, False)
            out["leaf_emissions" if args.emissions != "leaf" else "output_emissions"] = {
                "value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                "what": ("the same call with the emissions being the LEAF itself (rounds 1-5's headline): loss.backward() hands "
                         "the forward launch's gradient to x.grad, no graph below the emissions, no autograd engine"
                         if args.emissions != "leaf" else
                         "the same call on emissions that are a producer's output (x.view_as(x)): the default headline protocol")}
        if args.mode == "api" and "engine_proper_step" in wl:
            el, _ = timed_loop(lambda i: wl["engine_proper_step"](0), extras_steps, 3, fence, False)
            out["autograd_engine_from_the_loss"] = {
                "value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                "what": "the headline call with the criterion's short cut switched off (WFL_CTC_FAST_BACKWARD=0): "
                        "torch.Tensor.backward from the loss down -- ones_like fill, the criterion's node, the scale launch, "
                        "then the graph below the emissions"}
        if args.mode == "api" and "engine_step" in wl:
            el, _ = timed_loop(lambda i: wl["engine_step"](0), extras_steps, 3, fence, False)
            out["through_autograd_engine"] = {
                "value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                "what": "the same call with a producer that has kernels of its own (x * 1.0: a model's output, train.py:262-266" +
                        ("; transitions as an nn.Parameter, train.py:205-208" if args.workload == "asg" else "") +
                        "): incl. the x * 1.0 and its backward, two elementwise passes over [B,T,C]"}
        if args.workload == "ctc" and args.mode == "api":
            el, ph = timed_loop(wl["abi_step"], extras_steps, 3, fence, True)
            out["abi_kernels_only"] = {"value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                                       "kernel_ms": ph, "utterances_repaired_in_log_domain": meta["repaired"](),
                                       "what": "wfl_ctc_forward_backward through the C ABI, targets pre-staged: kernels only"}
        if args.workload == "ctc" and args.mode == "api" and args.config != "cfg5":
            el, _ = timed_loop(wl["module_step"], extras_steps, 3, fence, False)
            out["module_raw_scores"] = {"value": B * extras_steps / el, "unit": "utt/s", "ms_per_step": el * 1e3 / extras_steps,
                                        "what": "CTC(blank, use_pt=False)(x, targets).backward() on RAW scores: log_softmax "
                                                "fused into the step (one row_lse pass + the same launch), same targets every step"}
    if single and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl["payload"], args.cpu_sample_utts)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if args.workload == "ctc":
            out["cpu_torch_ctc_loss"] = cpu_torch_ctc(wl["payload"])
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
