#!/bin/bash
# (GPU box) bench steps with several builds of libwfl.so (WFL_LIB_PATH), alternating on the same box.
# usage: scripts/lib_ab_gpu.sh "<bench args>" <lib or "-" for the in-tree one> ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
args=$1; shift
for pass in 1 2 3; do
  for l in "$@"; do
    p=$l; [ "$l" = "-" ] && p=""
    v=$(WFL_LIB_PATH=$p python bench.py $args --no-cpu-baseline --no-extras --steps 300 --warmup 30 2>/dev/null |
        python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
    echo "pass $pass lib=$l ms_per_step $v"
  done
done
