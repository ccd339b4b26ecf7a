#!/bin/bash
# (GPU box) cfg3 / cfg4 steps with the stream-ordering events recorded with (1) and without (0) the system-scope fence,
# alternating on the same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/event_fence_ab.txt
: > $out
for pass in 1 2 3; do
  for cfg in cfg3 cfg4; do
    for f in 1 0; do
      v=$(WFL_EVENT_FENCE=$f python bench.py --config $cfg --no-cpu-baseline --no-extras --steps 300 --warmup 30 2>/dev/null |
          python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
      echo "pass $pass $cfg fence=$f ms_per_step $v" | tee -a $out
    done
  done
done
