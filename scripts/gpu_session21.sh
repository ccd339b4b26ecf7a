cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
rm -rf $O/stats_cfg4; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- python bench.py --workload transducer --steps 20 --warmup 5 --no-cpu-baseline --no-extras --targets same > $O/stats_cfg4.log 2>&1
cp $(find $O/stats_cfg4 -name "*kernel_stats.csv" | head -1) $O/cfg4_kernel_stats.csv; rm -rf $O/stats_cfg4
python bench.py --workload transducer --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_cfg4.json
