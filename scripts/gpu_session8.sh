#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s8
cd /tmp && export TMPDIR=/tmp
cd "$R"; rm -rf "$O"; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu > $O/pytest_parity.log 2>&1; tail -30 $O/pytest_parity.log
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -m gpu > $O/pytest_configs.log 2>&1; tail -5 $O/pytest_configs.log
cp gpurun_out/parity_r02.json $O/
for w in asg transducer; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; tail -2 $O/bench_$w.err; cat $O/bench_$w.json
done
