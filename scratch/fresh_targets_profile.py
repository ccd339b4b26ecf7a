import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
sys.argv = ["bench.py", "--workload", sys.argv[1] if len(sys.argv) > 1 else "transducer", "--targets", "fresh"]
args = bench.parse()
mk = {"transducer": bench.make_transducer, "ctc": bench.make_ctc, "asg": bench.make_asg}[args.workload]
n = 60
wl = mk(args, 0, n + 10) if args.workload == "transducer" else mk(args, 0, n + 10, None)
step = wl["step"]
for i in range(8): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); host = 0.0
pr = cProfile.Profile(); pr.enable()
for i in range(8, 8 + n):
    a = time.perf_counter(); step(i); host += time.perf_counter() - a
pr.disable()
torch.cuda.synchronize()
print(f"{args.workload} fresh targets: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step, host {host / n * 1e3:.3f}")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(10); print(s.getvalue()[:2200])
