#!/bin/bash
# per-wave stats build of the CTC meet-in-the-middle launch + scripts/mitm_stats.py on the GPU box (no tests)
set -e
cd /root/repo/gtn_applications_amd/csrc
mkdir -p /tmp/dbg /root/repo/scripts/_build
/opt/rocm/bin/hipcc -DWFL_MITM_STATS=1 $EXTRA -O3 -std=c++17 -fPIC -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c ctc_kernels.hip -o /tmp/dbg/ctc_stats.o 2>&1 | grep -A3 error && exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/graph.cpp.o build/pack.cpp.o build/lattice_kernels.hip.o build/dense_kernels.hip.o build/conv_kernels.hip.o /tmp/dbg/ctc_stats.o -o /root/repo/scripts/_build/libwfl_stats.so
cd /root/repo
timeout 2400 /usr/local/graft/bin/gpurun --timeout 600 -- 'WFL_LIB_PATH=$PWD/scripts/_build/libwfl_stats.so timeout 300 python scripts/mitm_stats.py 2>&1 | tail -70' 2>&1 | tail -72
