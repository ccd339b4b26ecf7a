#!/bin/bash
# GPU-box half of scripts/mitm_multi.sh
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2; do
for v in "$@"; do
  echo "== $v (round $round)"
  rm -rf /tmp/mm$v
  WFL_LIB_PATH=$PWD/scripts/_build/libwfl_mm_${v}t.so timeout 60 rocprofv3 --kernel-trace --output-format csv -d /tmp/mm$v -- python bench.py --mode abi --steps 30 --warmup 3 --no-cpu-baseline --no-extras > /tmp/mm$v.log 2>&1
  python - "$(find /tmp/mm$v -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "mitm" in n.lower() or "repair" in n:
        d[n.split("(")[0].split("wfl::")[-1][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[3:]
    print("  %-40s calls %3d median %7.2f us  mean %7.2f  min %7.2f  max %7.2f" % (k, len(v), statistics.median(v), sum(v) / len(v), min(v), max(v)))
PY
  if [ $round = 1 ]; then
    WFL_LIB_PATH=$PWD/scripts/_build/libwfl_mm_${v}s.so timeout 60 python scripts/mitm_stats.py 2>&1 | grep -E "THREE|prologue|chain wave end|workgroup end" | head -20
  fi
done
done
