import time, torch
x = torch.randn(8192, 8192, device="cuda")
s2 = torch.cuda.Stream()
def busy(n=6):
    for _ in range(n):
        y = x @ x
torch.cuda.synchronize()
for name, mk in (("default Event", lambda: torch.cuda.Event()), ("timing Event", lambda: torch.cuda.Event(enable_timing=True)),
                 ("blocking Event", lambda: torch.cuda.Event(blocking=True))):
    for depth in (0, 6):
        busy(depth)
        evs = [mk() for _ in range(4)]
        t = []
        for ev in evs:
            a = time.perf_counter(); ev.record(); t.append((time.perf_counter() - a) * 1e6)
        a = time.perf_counter(); s2.wait_event(evs[-1]); tw = (time.perf_counter() - a) * 1e6
        a = time.perf_counter(); s2.wait_stream(torch.cuda.current_stream()); tws = (time.perf_counter() - a) * 1e6
        a = time.perf_counter(); torch.cuda.synchronize(); ts = (time.perf_counter() - a) * 1e6
        print(f"{name}, {depth} matmuls queued: record {[round(v) for v in t]} us, wait_event {tw:.0f}, wait_stream {tws:.0f}, then sync {ts:.0f}")
