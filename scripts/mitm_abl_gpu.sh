#!/bin/bash
# GPU-box half of scripts/mitm_abl.sh: kernel times of the cfg2 ABI step for each libwfl_abl<v>.so
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  echo "== ABL $v"
  rm -rf /tmp/abl$v
  WFL_LIB_PATH=$PWD/scripts/_build/libwfl_abl$v.so timeout 60 rocprofv3 --kernel-trace --output-format csv -d /tmp/abl$v -- python bench.py --mode abi --steps 30 --warmup 3 --no-cpu-baseline --no-extras > /tmp/abl$v.log 2>&1
  python - "$(find /tmp/abl$v -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "mitm" in n.lower() or "repair" in n:
        d[n.split("(")[0].split("wfl::")[-1][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[3:]  # (warm-up)
    print("  %-40s calls %3d median %7.2f us  mean %7.2f  min %7.2f  max %7.2f" % (k, len(v), statistics.median(v), sum(v) / len(v), min(v), max(v)))
PY
done
