#!/bin/bash
# usage (GPU box, repo root): scripts/kstat_one.sh <tag> <bench args...>   -> gpurun_out/<tag>/kernel_stats.csv + bench line
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
cd "$R"; mkdir -p "$O"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline --no-extras --targets same > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
timeout 600 python bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_line.json
head -8 $O/kernel_stats.csv
cat $O/bench_line.json
