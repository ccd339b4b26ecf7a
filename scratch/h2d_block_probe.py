"""Does a non_blocking H2D copy from pinned memory block the host while the stream is busy?"""
import time, torch
dev = torch.device("cuda")
a = torch.randn(8192, 8192, device=dev)
pin = torch.empty(1 << 20, dtype=torch.uint8, pin_memory=True)
dst = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
def busy():
    for _ in range(3): (a @ a)
for nbytes in (1024, 23 * 1024, 640 * 1024):
    for label, fn in (("copy_ non_blocking", lambda: dst[:nbytes].copy_(pin[:nbytes], non_blocking=True)),
                      ("device-side copy of mapped pinned", None)):
        if fn is None:
            # a device kernel reads the pinned buffer through its device-visible address
            src_dev = None
            try:
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                p = ctypes.c_void_p()
                rc = hip.hipHostGetDevicePointer(ctypes.byref(p), ctypes.c_void_p(pin.data_ptr()), 0)
                print("   hipHostGetDevicePointer rc", rc, hex(p.value or 0), hex(pin.data_ptr()))
            except Exception as e:
                print("   (no hip)", e)
            continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            busy()
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
            torch.cuda.synchronize()
        idle = []
        for _ in range(5):
            t0 = time.perf_counter(); fn(); idle.append((time.perf_counter() - t0) * 1e6)
            torch.cuda.synchronize()
        print(f"{nbytes:8d} B  {label}: busy stream {min(ts):8.1f} .. {max(ts):8.1f} us   idle stream {min(idle):6.1f} .. {max(idle):6.1f} us")
# matmul duration for reference
torch.cuda.synchronize(); t0 = time.perf_counter(); busy(); torch.cuda.synchronize(); print("busy() =", (time.perf_counter() - t0) * 1e3, "ms")
