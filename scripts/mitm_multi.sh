#!/bin/bash
# usage: scripts/mitm_multi.sh "name1:flags" "name2:flags" ...
# For every variant of the CTC kernels (-D flags): the cfg2 kernel time (rocprofv3 kernel trace, bench.py --mode abi) AND
# the per-wave statistics of a -DWFL_MITM_STATS=1 build of the same flags (pace per block, prologue), all on ONE GPU box.
set -e
cd /root/repo/gtn_applications_amd/csrc
mkdir -p /tmp/dbg /root/repo/scripts/_build
names=""
for spec in "$@"; do
  n="${spec%%:*}"; f="${spec#*:}"; names="$names $n"
  for kind in t s; do (
    extra=""; [ $kind = s ] && extra="-DWFL_MITM_STATS=1"
    /opt/rocm/bin/hipcc $f $extra -O3 -std=c++17 -fPIC -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -c ctc_kernels.hip -o /tmp/dbg/ctc_mm_$n$kind.o 2>&1 | grep -A3 "error" && exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/graph.cpp.o build/pack.cpp.o build/lattice_kernels.hip.o build/dense_kernels.hip.o build/conv_kernels.hip.o /tmp/dbg/ctc_mm_$n$kind.o -o /root/repo/scripts/_build/libwfl_mm_$n$kind.so ) &
  done
done; wait
cd /root/repo
timeout 2400 /usr/local/graft/bin/gpurun --timeout 900 -- "scripts/mitm_multi_gpu.sh $names" 2>&1 | tail -120
