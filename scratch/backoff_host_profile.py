import cProfile, pstats, sys, os, io
sys.argv = ["x"]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts", "backoff_step_probe.py")).read()
src = src.split("for _ in range(5):")[0]
exec(src)
import torch
torch.cuda.set_sync_debug_mode(1)
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])
