#!/bin/bash
# usage: scripts/mitm_ab.sh "<flagsA>" "<flagsB>"  -- two builds of the CTC kernels with different -D flags, timed on the
# SAME GPU box, alternating (A B A B): box-to-box differences are as large as the effects being measured
set -e
cd /root/repo/gtn_applications_amd/csrc
mkdir -p /tmp/dbg /root/repo/scripts/_build
i=0
for f in "$1" "$2"; do
  i=$((i+1))
  ( /opt/rocm/bin/hipcc $f -O3 -std=c++17 -fPIC -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c ctc_kernels.hip -o /tmp/dbg/ctc_ab$i.o 2>&1 | grep error && exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/graph.cpp.o build/pack.cpp.o build/lattice_kernels.hip.o build/dense_kernels.hip.o build/conv_kernels.hip.o /tmp/dbg/ctc_ab$i.o -o /root/repo/scripts/_build/libwfl_ablab$i.so ) &
done; wait
cd /root/repo
timeout 2400 /usr/local/graft/bin/gpurun --timeout 900 -- "scripts/mitm_abl_gpu.sh ab1 ab2 ab1 ab2" 2>&1 | grep -iE "ABL|mitm"
