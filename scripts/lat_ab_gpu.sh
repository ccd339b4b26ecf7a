#!/bin/bash
# GPU-box half of scripts/lat_ab.sh: per-kernel medians and the step time of a bench configuration for libwfl_latab{1,2}.so
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 1 2 1 2; do
  echo "== build $v"
  rm -rf /tmp/latab$v
  WFL_LIB_PATH=$PWD/scripts/_build/libwfl_latab$v.so timeout 300 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  ms_per_step %.4f' % d['ms_per_step'])"
  WFL_LIB_PATH=$PWD/scripts/_build/libwfl_latab$v.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/latab$v -- python bench.py "$@" --steps 30 --warmup 3 --no-cpu-baseline --no-extras > /tmp/latab$v.log 2>&1
  python - "$(find /tmp/latab$v -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "wfl::" in n:
        d[n.split("(")[0].split("wfl::")[-1][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[3:] or v
    print("  %-40s calls %3d median %7.2f us  mean %7.2f" % (k, len(v), statistics.median(v), sum(v) / len(v)))
PY
done
