#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_one.sh <tag> <cfg key> <bench args...>  -> gpurun_out/<tag>/pmc_<cfg>_{FETCH,WRITE}_SIZE.csv
# + pmc_traffic.json (separate --pmc passes, no trace domains besides --kernel-trace)
tag=$1; cfg=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
cd "$R"; mkdir -p "$O"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${cfg}_$c -- python bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-extras --targets same > $O/pmc_${cfg}_$c.log 2>&1
  cp $(find $O/pmc_${cfg}_$c -name "*counter_collection.csv" | head -1) $O/pmc_${cfg}_$c.csv
  rm -rf $O/pmc_${cfg}_$c
done
python scripts/pmc_traffic.py $O > $O/pmc_traffic.json
python - $O/pmc_traffic.json <<'PY'
import json, sys
for cfg, rec in json.load(open(sys.argv[1]))["configs"].items():
    for k, v in rec["kernels"].items():
        print(cfg, k, "fetch %.1f MB write %.1f MB total %.1f MB" % (v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6, v["hbm_bytes"] / 1e6))
PY
rm -f $O/pmc_*_SIZE.csv.bak
