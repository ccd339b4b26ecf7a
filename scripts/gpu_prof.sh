#!/bin/bash
# usage: scripts/gpu_prof.sh <tag> : rocprofv3 kernel stats of the three workloads (+ cfg5 shard) with pre-staged targets
tag=${1:-p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
cd "$R"; rm -rf "$O"; mkdir -p "$O"
run() {  # name, bench args
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -- python bench.py "$@" --no-cpu-baseline --no-extras --targets same > $O/stats_$name.log 2>&1
  cp $(find $O/stats_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  rm -rf $O/stats_$name
  head -12 $O/${name}_kernel_stats.csv | cut -d, -f1-8
}
run ctc_cfg2 --workload ctc --steps 50 --warmup 5
run asg_cfg3 --workload asg --steps 20 --warmup 5
run transducer_cfg4 --workload transducer --steps 20 --warmup 5
run ctc_cfg5_shard --workload ctc --T 2000 --C 512 --steps 20 --warmup 5
