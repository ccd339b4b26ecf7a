#!/bin/bash
# usage: scripts/lat_ab.sh "<flagsA>" "<flagsB>" [bench args]  -- two builds of the lattice kernels with different -D flags,
# timed on the SAME GPU box, alternating (A B A B); default workload: --config cfg4
set -e
cd /root/repo/gtn_applications_amd/csrc
mkdir -p /tmp/dbg /root/repo/scripts/_build
i=0
for f in "$1" "$2"; do
  i=$((i+1))
  ( /opt/rocm/bin/hipcc $f -O3 -std=c++17 -fPIC -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c lattice_kernels.hip -o /tmp/dbg/lat_ab$i.o 2>&1 | grep error && exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/graph.cpp.o build/pack.cpp.o /tmp/dbg/lat_ab$i.o build/dense_kernels.hip.o build/conv_kernels.hip.o build/ctc_kernels.hip.o -o /root/repo/scripts/_build/libwfl_latab$i.so ) &
done; wait
cd /root/repo
shift 2
timeout 2400 /usr/local/graft/bin/gpurun --timeout 900 -- "scripts/lat_ab_gpu.sh ${*:---config cfg4}" 2>&1 | grep -vE "^\[gpurun\] sending|amdgpu.ids"
