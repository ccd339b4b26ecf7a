#!/usr/bin/env python
"""Headline benchmark of the MI355X WFST loss engine.

Metric (BASELINE.json): utterances/sec, forward+backward, on the CTC benchmark of the reference
(benchmarks/ctc_benchmark.py:17-31 protocol: randn "log_probs", targets randint(C-2), blank C-1,
reduction "none", fwd + bwd per step) at configs[1]: T=1000, C=100, B=128, L=44, one MI355X.

A "step" = one pass of the hot path over one batch resident in HBM: the call the criterion makes
through the C ABI of libwfl.so -- wfl_ctc_forward_backward, ONE pipelined launch that runs the alpha
and beta chains, the loss reduction and the dense [B,T,C] gradient (--ctc-step split times the same
work as wfl_ctc_forward -> wfl_reduce_loss -> wfl_ctc_grad).  `value` is measured at that boundary
(--mode abi, default); the same step through the Python drop-in operator
(`CTCLoss(x, targets, blank).backward()`, what the reference's benchmark script times) is reported
next to it as `python_api`, a hipGraph replay of it as `hip_graph`, the CPU baselines as
`cpu_baseline` (oracle C port, all host cores) and `cpu_torch_ctc_loss`.

  python bench.py                      # 1 GPU, defaults finish in well under a minute
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: utterances shard over ranks (same per-GPU batch: weak scaling); CTC has no learnable
transition weights, so there is no data-path collective (ASG's transition-gradient all-reduce is
exercised with --workload asg).  Timing: barrier + synchronize on both sides, MAX over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="ctc", choices=["ctc", "asg", "transducer"])
    ap.add_argument("--mode", default="abi", choices=["abi", "api"])
    ap.add_argument("--B", type=int, default=None)
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--C", type=int, default=None)
    ap.add_argument("--L", type=int, default=44)
    ap.add_argument("--ctc-chain", default="default", choices=["default", "log", "fast"],
                    help="CTC chain kernel of the split step: library default, log-domain, or the lane-exponent chain + certificate")
    ap.add_argument("--ctc-step", default="pipelined", choices=["split", "pipelined"],
                    help="CTC step: forward and gradient kernels back to back, or one pipelined launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-utts", type=int, default=128)
    return ap.parse_args()


def dist_setup(n):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if n > 1 or world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        return rank, world, local, dist
    torch.cuda.set_device(0)
    return 0, 1, 0, None


# --------------------------------------------------------------------------------------------------
# workloads: each returns (step_fn, phase_names, meta).  step_fn(events) runs one fwd+bwd and, if
# `events` is a list, appends torch.cuda.Event markers between the kernels on the launch stream.
# --------------------------------------------------------------------------------------------------
def make_ctc(args, rank, mode):
    from gtn_applications_amd import engine as E
    from gtn_applications_amd.criterions import ctc

    B, T, C, L = args.B or 128, args.T or 1000, args.C or 100, args.L
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(B, T, C, generator=g).cuda()
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()
    blank = C - 1
    dev = x.device
    tg = E.targets_on_device(targets, dev)
    scale, _, coef = E.loss_factors(tg, "none")
    gout = torch.ones(1, device=dev)
    dx = torch.empty_like(x)

    def mark(events):
        if events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            events.append(e)

    from gtn_applications_amd import _native as N
    chain_flags = {"default": E.CTC_DEFAULT_FLAGS, "log": 0, "fast": N.CTC_FAST_CHAIN}[args.ctc_chain]
    last = [None]
    if mode == "abi" and args.ctc_step == "pipelined":
        def step(events=None):
            mark(events)
            # chains + gradient waves + loss reduction, one launch
            last[0] = E.ctc_forward_backward(x, tg, blank, coef, gout, dx, loss_scale=scale, want_loss=True)
            mark(events)
        # (the event bracket holds both launches of the step: lane-exponent pipelined launch + certificate / repair launch)
        phases = ["ctc_fast_pipelined_kernel (+ctc_repair_kernel)" if os.environ.get("WFL_CTC_PIPELINE") != "log" and L <= 63
                  else "ctc_pipelined_kernel"]
    elif mode == "abi":
        def step(events=None):
            mark(events)
            ws, nll = E.ctc_forward(x, tg, blank, chain_flags)  # alpha || beta chains
            mark(events)
            E.reduce_loss(nll, scale, 1.0)
            E.ctc_grad(x, tg, blank, ws, nll, coef, gout, dx)  # posteriors -> dense gradient
            mark(events)
        phases = ["ctc_log_chain_kernel", "ctc_grad_kernel(+reduce_loss)"]
    else:
        xr = x.clone().requires_grad_(True)

        def step(events=None):
            xr.grad = None
            ctc.CTCLoss(xr, targets, blank).backward()
        phases = []
    which = {(1000, 100, 128, 44): " (BASELINE configs[1])", (2000, 512, 128, 44): " (BASELINE configs[4], one GPU's shard)",
             (150, 28, 8, 44): " (BASELINE configs[0])"}.get((T, C, B, L), "")
    def repaired():  # utterances of the last step that the certificate sent to the log-domain repair launch
        return E.ctc_pipeline_repaired(last[0][0], B, T, tg.max_len) if last[0] is not None else None

    meta = dict(
        workload=f"ctc fwd+bwd T={T} C={C} B={B} L={L}{which}", B=B, T=T, C=C, L=L, repaired=repaired,
        algorithmic_bytes_per_utt=8 * T * C,
    )
    return step, phases, meta, (x, targets, blank)


def make_asg(args, rank, mode, dist):
    from gtn_applications_amd.criterions import asg

    B, T, C, L = args.B or 128, args.T or 1000, args.C or 100, args.L
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
    W = torch.randn(C + 1, C, generator=g).cuda().requires_grad_(True)
    targets = torch.randint(C - 2, (B, L), generator=g).tolist()

    def step(events=None):
        x.grad = None
        W.grad = None
        asg.ASGLoss(x, W, targets).backward()
        if dist is not None:  # the one exchange step of the path: transition-weight gradient
            dist.all_reduce(W.grad)

    meta = dict(workload=f"asg fwd+bwd T={T} C={C} B={B} L={L} (BASELINE configs[2])", B=B, T=T, C=C, L=L,
                algorithmic_bytes_per_utt=8 * T * C)
    return step, [], meta, None


def make_transducer(args, rank, mode):
    import random

    from gtn_applications_amd.criterions import transducer

    B, T = args.B or 64, args.T or 800
    rnd = random.Random(rank)
    # 1000 synthetic word pieces over 26 graphemes (the reference's token file is not shipped):
    # same size and length statistics as benchmarks/word_pieces_tokens_1000.txt (mean ~4.6 letters)
    letters = "abcdefghijklmnopqrstuvwxyz"
    pieces = set(letters)
    while len(pieces) < 1000:
        pieces.add("".join(rnd.choice(letters) for _ in range(rnd.choice([2, 3, 4, 5, 6, 7]))))
    tokens = sorted(pieces)
    g2i = {c: i for i, c in enumerate(letters)}
    C = len(tokens) + 1
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(B, T, C, generator=g).cuda().requires_grad_(True)
    targets = [torch.tensor([g2i[ch] for _ in range(15) for ch in rnd.choice(tokens)]) for _ in range(B)]
    crit = transducer.Transducer(tokens, g2i, blank="optional", allow_repeats=False, reduction="mean")

    def step(events=None):
        x.grad = None
        crit(x, targets).backward()

    meta = dict(workload=f"transducer 1000 word pieces fwd+bwd T={T} C={C} B={B} (BASELINE configs[3])", B=B, T=T, C=C,
                L=15, algorithmic_bytes_per_utt=8 * T * C)
    return step, [], meta, None


def cpu_baseline(payload, n_utts):
    """The oracle's graph-faithful C restatement (oracle/cpu_ref.c, "port") on this box's host
    cores, on a bounded sample of the same workload."""
    from oracle import cpu_ref

    x, targets, blank = payload
    n = min(n_utts, x.shape[0])
    xs = x[:n].cpu().numpy()
    tg = targets[:n]
    cores = os.cpu_count() or 1
    cpu_ref.ctc_cpu(xs[:max(1, n // 8)], tg[:max(1, n // 8)], blank, "none", cores)  # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        cpu_ref.ctc_cpu(xs, tg, blank, "none", cores)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0:  # a bounded sample: about 10 s of wall clock on all host cores
            break
    return dict(
        value=n * reps / el, unit="utt/s", cores=cores, kind="port",
        sample=f"{reps} x fwd+bwd of {n} utterances (same T,C,L) with oracle/cpu_ref.c on {cores} threads, {el:.1f} s",
    )


def cpu_torch_ctc(payload):
    """torch.nn.functional.ctc_loss on the host cores -- the reference's own alternative CTC path
    (criterions/ctc.py:109-121) -- on the same batch, forward + backward, for context."""
    x, targets, blank = payload
    xc = x.detach().cpu().requires_grad_(True)
    B, T, _ = xc.shape
    tg = torch.tensor(targets, dtype=torch.long)
    lens = torch.full((B,), tg.shape[1], dtype=torch.long)

    def run():
        xc.grad = None
        torch.nn.functional.ctc_loss(xc.permute(1, 0, 2), tg, torch.full((B,), T, dtype=torch.long), lens,
                                     blank=blank, reduction="sum").backward()

    run()
    reps, t0 = 0, time.perf_counter()
    while True:
        run()
        reps += 1
        el = time.perf_counter() - t0
        if el > 3.0:
            break
    return dict(value=B * reps / el, unit="utt/s", threads=torch.get_num_threads(),
                what=f"torch.nn.functional.ctc_loss fwd+bwd on CPU, {reps} x {B} utterances in {el:.1f} s")


def pmc_traffic(kernel, meta):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate `--pmc` runs of this same command; see profiles/README.md).
    Only valid for the configuration they were measured on; otherwise null."""
    path = os.path.join(ROOT, "profiles", "r01_ctc_cfg2_pmc_traffic.json")
    if not os.path.exists(path) or (meta["B"], meta["T"], meta["C"], meta["L"]) != (128, 1000, 100, 44):
        return None
    with open(path) as f:
        rec = json.load(f)["kernels"]
    for name, v in rec.items():
        if kernel.startswith(name):
            return v["hbm_bytes"]
    return None


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    rank, world, local, dist = dist_setup(args.gpus)
    if args.workload == "ctc":
        step, phases, meta, payload = make_ctc(args, rank, args.mode)
    elif args.workload == "asg":
        step, phases, meta, payload = make_asg(args, rank, args.mode, dist)
    else:
        step, phases, meta, payload = make_transducer(args, rank, args.mode)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    events = [] if phases else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    B = meta["B"]
    ms = elapsed * 1e3 / args.steps
    value = world * B * args.steps / elapsed
    out = {
        "metric": "utterances/sec fwd+bwd (ctc_benchmark T=1000,C=100,B=128); HBM GB/s vs peak",
        "value": value, "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": meta["workload"], "mode": args.mode, "per_gpu_batch": B,
                   "global_batch": B * world, "parallelism": f"dp{world} (utterance shards, no data-path collective)"
                   if args.workload == "ctc" else f"dp{world} (all-reduce of transition grads)"},
    }
    if callable(meta.get("repaired")):
        out["config"]["utterances_repaired_in_log_domain"] = meta["repaired"]()
    alg_bytes = meta["algorithmic_bytes_per_utt"] * B  # per launch: every launch processes the whole batch
    if events:
        per = len(phases) + 1
        durs = np.zeros(len(phases))
        for s in range(args.steps):
            ev = events[s * per:(s + 1) * per]
            for k in range(len(phases)):
                durs[k] += ev[k].elapsed_time(ev[k + 1])
        durs /= args.steps  # ms, average per launch
        dom = int(np.argmax(durs))
        achieved = alg_bytes / (durs[dom] * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": phases[dom], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(phases[dom], meta),
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms": {p: float(d) for p, d in zip(phases, durs)},
            "step_achieved": alg_bytes / (float(durs.sum()) * 1e-3) / 1e9,
            "step_frac": alg_bytes / (float(durs.sum()) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "note": "achieved = 8*T*C*B algorithmic bytes / average duration of the dominant kernel (HIP events on "
                    "the launch stream, over the timed region); step_* divides by the sum of all kernels of a step",
        }
    else:
        ach = alg_bytes / (ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "whole step (host-timed)", "achieved": ach,
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None}
    if rank == 0 and world == 1 and args.workload == "ctc" and args.mode == "abi":
        # the same three launches captured once in a hipGraph and replayed (no host launch gaps, no event records)
        try:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(graph):
                step()
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            n_rep = max(args.steps, 20)
            t0 = time.perf_counter()
            for _ in range(n_rep):
                graph.replay()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            out["hip_graph"] = {"value": B * n_rep / el, "unit": "utt/s", "ms_per_step": el * 1e3 / n_rep,
                                "what": "the step's kernels captured in one hipGraph and replayed"}
        except Exception as exc:  # capture is an extra, never fail the bench on it
            out["hip_graph"] = {"error": str(exc)[:200]}
    if rank == 0 and world == 1 and args.workload == "ctc":
        if args.mode == "abi":  # the same step through the Python drop-in operator, for the record
            step_api, _, _, _ = make_ctc(args, rank, "api")
            for _ in range(3):
                step_api()
            torch.cuda.synchronize()
            n_api = max(10, args.steps // 4)
            t0 = time.perf_counter()
            for _ in range(n_api):
                step_api()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            out["python_api"] = {"value": B * n_api / el, "unit": "utt/s", "ms_per_step": el * 1e3 / n_api,
                                 "what": "CTCLoss(x, targets, blank).backward() eager, incl. host overhead"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(payload, args.cpu_sample_utts)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            out["cpu_torch_ctc_loss"] = cpu_torch_ctc(payload)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
