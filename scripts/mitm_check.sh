#!/bin/bash
# build, then on the GPU box: the CTC parity tests and the cfg2 kernel times (ABI step)
set -e
cd /root/repo
make -C gtn_applications_amd/csrc -j16 2>&1 | grep -E "error" -A3 && exit 1
cp gtn_applications_amd/libwfl.so scripts/_build/libwfl_abl0.so
timeout 2400 /usr/local/graft/bin/gpurun --timeout 900 -- 'timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "ctc" 2>&1 | tail -3; scripts/mitm_abl_gpu.sh 0' 2>&1 | tail -8
