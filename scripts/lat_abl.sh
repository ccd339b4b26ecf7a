#!/bin/bash
# usage: scripts/lat_abl.sh "<values of WFL_LAT_ABL>"  -- timing-only ablations of the lattice sweep at cfg4 (results wrong)
set -e
cd /root/repo/gtn_applications_amd/csrc
mkdir -p /tmp/dbg /root/repo/scripts/_build
for v in $1; do (
  /opt/rocm/bin/hipcc -DWFL_LAT_ABL=$v -O3 -std=c++17 -fPIC -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -c lattice_kernels.hip -o /tmp/dbg/lat_abl$v.o 2>&1 | grep error && exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/graph.cpp.o build/pack.cpp.o /tmp/dbg/lat_abl$v.o build/dense_kernels.hip.o build/conv_kernels.hip.o build/ctc_kernels.hip.o -o /root/repo/scripts/_build/libwfl_lat$v.so ) &
done; wait
cd /root/repo
cmd=""
for v in $1; do cmd="$cmd echo == LAT_ABL $v; rm -rf /tmp/l$v; WFL_LIB_PATH=\$PWD/scripts/_build/libwfl_lat$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/l$v -- python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/l$v.log 2>&1; python scripts/kstats.py /tmp/l$v | grep -E 'prob_chain|occ_grad';"; done
timeout 2400 /usr/local/graft/bin/gpurun --timeout 900 -- "cd /tmp && export TMPDIR=/tmp; cd \$GRAFT_REPO_ROOT; $cmd" 2>&1 | grep -E "LAT_ABL|prob_chain|occ_grad"
