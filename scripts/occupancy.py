import ctypes, torch
lib = ctypes.CDLL("/root/repo/gtn_applications_amd/libwfl.so")  # after import torch (one HIP runtime)
torch.zeros(1).cuda()
for lds in (0, 8192, 16384, 24000, 32768, 49152, 65536, 100000, 160000):
    print(lds, lib.wfl_debug_grad_occupancy(lds))
p = torch.cuda.get_device_properties(0)
print(p.multi_processor_count, getattr(p, "shared_memory_per_multiprocessor", None), getattr(p, "max_threads_per_multi_processor", None))
