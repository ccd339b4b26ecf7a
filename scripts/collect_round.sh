#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/collect_round.sh <tag>
# kernel-trace stats for the three BASELINE single-GPU workloads, PMC traffic passes for cfg2 and the
# bench lines; everything lands under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
tag=${1:-r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
cd "$R"; rm -rf "$O"; mkdir -p "$O"
for w in ctc asg transducer; do
  steps=50; [ $w = ctc ] || steps=10
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline > $O/stats_$w.log 2>&1
  cp $(find $O/stats_$w -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
  python bench.py --workload $w --steps $steps --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/bench_lines.jsonl
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/pmc_$c.log 2>&1
  cp $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) $O/pmc_$c.csv
done
python bench.py --T 2000 --C 512 --no-cpu-baseline --steps 20 2>/dev/null | tail -1 >> $O/bench_lines.jsonl
python scripts/kstats.py $O/stats_ctc $O/stats_asg $O/stats_transducer
python scripts/pmc_traffic.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv > $O/pmc_traffic.json; cat $O/pmc_traffic.json
rm -rf $O/stats_* $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
