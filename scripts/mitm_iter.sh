#!/bin/bash
# build libwfl.so + the stats variant, then (GPU box) CTC tests, cfg2 kernel stats, per-wave stats
set -e
cd /root/repo
make -C gtn_applications_amd/csrc -j8 2>&1 | grep -E "error|Error" && exit 1
cd gtn_applications_amd/csrc
/opt/rocm/bin/hipcc -DWFL_MITM_STATS=1 -O3 -std=c++17 -fPIC -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c ctc_kernels.hip -o /tmp/dbg/ctc_stats.o 2>&1 | grep error && exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/graph.cpp.o build/pack.cpp.o build/lattice_kernels.hip.o build/dense_kernels.hip.o build/conv_kernels.hip.o /tmp/dbg/ctc_stats.o -o /root/repo/scripts/_build/libwfl_stats.so
cd /root/repo
timeout 2400 /usr/local/graft/bin/gpurun --timeout 900 -- 'mkdir -p gpurun_out/r3c; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "ctc" 2>&1 | tail -3; scripts/kstat_one.sh r3c_cfg2 --workload ctc | head -3; WFL_LIB_PATH=$PWD/scripts/_build/libwfl_stats.so timeout 300 python scripts/mitm_stats.py 2>&1 | tail -60' 2>&1 | tail -52
