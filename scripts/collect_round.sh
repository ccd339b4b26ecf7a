#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/collect_round.sh <tag>
# Everything profiles/ quotes for a round, under gpurun_out/<tag>/:
#   <cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py <cfg args> --targets same`
#   pmc_<cfg>_{FETCH,WRITE}_SIZE.csv -> pmc_traffic.json (scripts/pmc_traffic.py; separate --pmc passes)
#   bench_lines.jsonl        the default bench.py line of every configuration (operator path; `fresh_targets` beside it)
tag=${1:-r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
cd "$R"; rm -rf "$O"; mkdir -p "$O"
declare -A ARGS=( [cfg2]="--workload ctc" [cfg3]="--workload asg" [cfg4]="--workload transducer" [cfg5_shard]="--workload ctc --T 2000 --C 512" )
for cfg in cfg2 cfg3 cfg4 cfg5_shard; do
  a=${ARGS[$cfg]}
  steps=50; [ $cfg = cfg2 ] || steps=20
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- python bench.py $a --steps $steps --warmup 5 --no-cpu-baseline --no-extras --targets same > $O/stats_$cfg.log 2>&1
  cp $(find $O/stats_$cfg -name "*kernel_stats.csv" | head -1) $O/${cfg}_kernel_stats.csv
  rm -rf $O/stats_$cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${cfg}_$c -- python bench.py $a --steps 10 --warmup 2 --no-cpu-baseline --no-extras --targets same > $O/pmc_${cfg}_$c.log 2>&1
    cp $(find $O/pmc_${cfg}_$c -name "*counter_collection.csv" | head -1) $O/pmc_${cfg}_$c.csv
    rm -rf $O/pmc_${cfg}_$c
  done
done
python scripts/pmc_traffic.py $O > $O/pmc_traffic.json
for cfg in cfg2 cfg3 cfg4 cfg5_shard; do
  python bench.py ${ARGS[$cfg]} --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/bench_lines.jsonl
done
python scripts/kstats.py $O
rm -f $O/pmc_*_SIZE.csv.bak
