#!/bin/bash
# GPU-box half of scripts/mitm_abl.sh: kernel times of the cfg2 ABI step for each libwfl_abl<v>.so
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  echo "== ABL $v"
  WFL_LIB_PATH=$PWD/scripts/_build/libwfl_abl$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$v -- python bench.py --mode abi --steps 30 --warmup 3 --no-cpu-baseline --no-extras > /tmp/abl$v.log 2>&1
  python - "$(find /tmp/abl$v -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mitm" in r["Name"] or "repair" in r["Name"]:
        print("  %-28s calls %4s avg %8.2f us min %8.2f max %8.2f" % (r["Name"].split("(")[0].split("::")[-1][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
