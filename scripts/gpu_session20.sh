cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
rm -rf $O/stats_cfg3; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3 -- python bench.py --workload asg --steps 20 --warmup 5 --no-cpu-baseline --no-extras --targets same > $O/stats_cfg3.log 2>&1
cp $(find $O/stats_cfg3 -name "*kernel_stats.csv" | head -1) $O/cfg3_kernel_stats.csv; rm -rf $O/stats_cfg3
bash scratch/timeline.sh asg > $O/timeline_cfg3.txt 2>&1
bash scratch/run_fal_alone.sh > $O/fal_alone.txt 2>&1
python bench.py --workload asg --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_cfg3.json
rm -rf gpurun_out/tl gpurun_out/fal_alone
