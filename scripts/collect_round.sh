#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/collect_round.sh <tag>
# Everything profiles/ quotes for a round, under gpurun_out/<tag>/:
#   <cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --config <cfg>` (the operator path,
#                            same targets every step: the reference benchmarks' protocol)
#   pmc_<cfg>_{FETCH,WRITE}_SIZE.csv -> pmc_traffic.json (scripts/pmc_traffic.py; separate --pmc passes)
#   bench_lines.jsonl        the default bench.py line of every configuration (+ cfg2 through the C ABI, B = 1024)
#   ngram_lines.txt          benchmarks/transducer_benchmark.py (N = 81, T = 250, L = 44, n-gram 0 / 1 / 2), bigram
#                            normaliser on the dense engine and, for comparison, on the general lattice sweep
#   ngram_kernel_stats.csv   rocprofv3 kernel stats of the same script (B = 16)
#   asg_wide_line.json       ASG with 1000 classes (csrc/dense_wide.h): python bench.py --workload asg --C 1000 --B 32 --T 250
#   stc_conv_lines.jsonl, asg_129_to_200_classes.txt, <cfg>_step_timeline.txt, host_overhead.txt, viterbi_times.txt,
#   ctc_long_probe.txt, ctc_module_time.txt   (round 5: the scripts of the same names)
tag=${1:-r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
cd "$R"; rm -rf "$O"; mkdir -p "$O"
for cfg in cfg2 cfg3 cfg4 cfg5; do
  steps=50; [ $cfg = cfg2 ] || steps=20
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- python bench.py --config $cfg --steps $steps --warmup 5 --no-cpu-baseline --no-extras > $O/stats_$cfg.log 2>&1
  cp $(find $O/stats_$cfg -name "*kernel_stats.csv" | head -1) $O/${cfg}_kernel_stats.csv
  rm -rf $O/stats_$cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${cfg}_$c -- python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_${cfg}_$c.log 2>&1
    cp $(find $O/pmc_${cfg}_$c -name "*counter_collection.csv" | head -1) $O/pmc_${cfg}_$c.csv
    rm -rf $O/pmc_${cfg}_$c
  done
done
# round 6: the CTC launch at B = 1024 (eight utterance pairs per CU: no chain-latency excuse) through the C ABI, kernel stats + PMC
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2_B1024 -- python bench.py --config cfg2 --mode abi --B 1024 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/stats_cfg2_B1024.log 2>&1 </dev/null
f=$(find $O/stats_cfg2_B1024 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg2_B1024_kernel_stats.csv
rm -rf $O/stats_cfg2_B1024
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_cfg2_B1024_$c -- python bench.py --config cfg2 --mode abi --B 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_cfg2_B1024_$c.log 2>&1 </dev/null
  f=$(find $O/pmc_cfg2_B1024_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_cfg2_B1024_$c.csv
  rm -rf $O/pmc_cfg2_B1024_$c
done
python scripts/pmc_traffic.py $O > $O/pmc_traffic.json
for cfg in cfg2 cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 30 --warmup 5 2>/dev/null | tail -1 >> $O/bench_lines.jsonl
done
python bench.py --config cfg2 --mode abi --B 1024 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 >> $O/bench_lines.jsonl
python bench.py --workload asg --C 1000 --B 32 --T 250 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/asg_wide_line.json
for n in 1 0; do
  echo "== WFL_DENSE_NGRAM=$n (B=16)" >> $O/ngram_lines.txt
  WFL_DENSE_NGRAM=$n timeout 600 python benchmarks/transducer_benchmark.py 16 2>/dev/null | grep -i "ngram" >> $O/ngram_lines.txt
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ngram -- python benchmarks/transducer_benchmark.py 16 > $O/stats_ngram.log 2>&1
cp $(find $O/stats_ngram -name "*kernel_stats.csv" | head -1) $O/ngram_kernel_stats.csv
rm -rf $O/stats_ngram
python scripts/kstats.py $O
# round 5: STC / ConvTransduce1D at a size, ASG beyond 128 classes, step timelines, host overhead, resource table input
python scripts/at_size_lines.py > $O/stc_conv_lines.jsonl 2> $O/stc_conv_lines.log
python scripts/asg_classes_time.py > $O/asg_129_to_200_classes.txt 2>/dev/null
python scripts/asg_classes_time.py 128,1000,190 128,1000,200 128,1000,256 128,1000,320 128,1000,330 > $O/asg_193_to_320_classes.txt 2>/dev/null
python scripts/dense_split_time.py 128,1000,190 128,1000,200 128,1000,256 128,1000,320 32,250,1000 >> $O/asg_193_to_320_classes.txt 2>/dev/null
python scripts/engine_path_probe.py > $O/engine_path_probe.txt 2>/dev/null
for cfg in cfg2 cfg3 cfg4; do
  bash scripts/step_timeline.sh --config $cfg > $O/${cfg}_step_timeline.txt 2>/dev/null
done
python scripts/host_overhead.py > $O/host_overhead.txt 2>/dev/null
python scripts/viterbi_time.py > $O/viterbi_times.txt 2>/dev/null
python scripts/ctc_long_probe.py > $O/ctc_long_probe.txt 2>/dev/null
python scripts/ctc_module_time.py > $O/ctc_module_time.txt 2>/dev/null
# round 5, later: what train.py calls per step besides the loss, the n-gram Transducer's step, the Viterbi kernels
python scripts/viterbi_module_time.py > $O/viterbi_module_times.txt 2>/dev/null
for b in 16 32; do for n in 1 2; do python scripts/ngram_step_probe.py $b $n 2>/dev/null | grep -v amdgpu >> $O/ngram_step_probe.txt; done; done
bash scripts/cmd_timeline.sh gather python scripts/ngram_step_probe.py 32 2 > $O/ngram_bigram_step_timeline.txt 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_vit -- python scripts/asg_classes_time.py 128,1000,100 128,1000,150 > $O/stats_vit.log 2>&1
f=$(find $O/stats_vit -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "viterbi\|Name" $f > $O/viterbi_kernel_stats.csv
rm -rf $O/stats_vit
rm -f $O/pmc_*_SIZE.csv.bak
