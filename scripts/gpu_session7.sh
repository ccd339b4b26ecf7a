#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s7
cd /tmp && export TMPDIR=/tmp
cd "$R"; rm -rf "$O"; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu > $O/pytest_parity.log 2>&1; tail -30 $O/pytest_parity.log
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -m gpu > $O/pytest_configs.log 2>&1; tail -5 $O/pytest_configs.log
