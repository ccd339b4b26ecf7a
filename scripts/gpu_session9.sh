#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s9
cd /tmp && export TMPDIR=/tmp
cd "$R"; rm -rf "$O"; mkdir -p "$O"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python scripts/host_overhead.py 2>/dev/null | tail -8 > $O/host_overhead.txt; cat $O/host_overhead.txt
for w in asg transducer; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; cat $O/bench_$w.json
done
