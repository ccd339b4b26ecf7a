import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, bench
sys.argv = ["bench.py", "--workload", "transducer"]
args = bench.parse()
wl = bench.make_transducer(args, 0, 1)
for name in ("step", "leaf_step"):
    step = wl[name]
    for i in range(10): step(i)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter(); host = 0.0
    for i in range(n):
        a = time.perf_counter(); step(i); host += time.perf_counter() - a
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: {1e3 * (t2 - t0) / n:.4f} ms/step wall; host time in the calls {1e3 * host / n:.4f} ms/step; the GPU was {1e3 * (t2 - t1):.3f} ms behind the host at the end")
