cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pt; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_cfg2_$c -- python bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-extras --targets same > $O/pmc_$c.log 2>&1
  cp $(find $O/pmc_cfg2_$c -name "*counter_collection.csv" | head -1) $O/pmc_cfg2_$c.csv
done
python scripts/pmc_traffic.py $O | python -c "
import sys, json
j = json.load(sys.stdin)
for k, v in j['configs'].items():
    print(k, {n[:30]: round(r['hbm_bytes'] / 1e6, 1) for n, r in v['kernels'].items() if r['hbm_bytes'] > 1e6})"
