#!/bin/bash
# usage: scripts/mitm_abl.sh "<abl values>"  -- builds libwfl variants with -DWFL_MITM_ABL=<v> (scratch ablations of the
# meet-in-the-middle CTC launch, results are WRONG by construction) and times the cfg2 kernel of each on the GPU box
set -e
cd /root/repo/gtn_applications_amd/csrc
mkdir -p /tmp/dbg /root/repo/scripts/_build
for v in $1; do (
  /opt/rocm/bin/hipcc -DWFL_MITM_ABL=${v%%s*} $EXTRA -O3 -std=c++17 -fPIC -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c ctc_kernels.hip -o /tmp/dbg/ctc_abl$v.o 2>&1 | grep error && exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/graph.cpp.o build/pack.cpp.o build/lattice_kernels.hip.o build/dense_kernels.hip.o build/conv_kernels.hip.o /tmp/dbg/ctc_abl$v.o -o /root/repo/scripts/_build/libwfl_abl$v.so ) &
done; wait
cd /root/repo
timeout 2400 /usr/local/graft/bin/gpurun --timeout 900 -- "scripts/mitm_abl_gpu.sh $1" 2>&1 | tail -40
