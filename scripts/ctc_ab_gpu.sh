#!/bin/bash
# GPU box: kernel durations of the cfg2 CTC step for several builds (scripts/_build/libwfl_mm_<name>t.so), alternating
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2 3; do
for v in "$@"; do
  rm -rf /tmp/ab$v
  WFL_LIB_PATH=$PWD/scripts/_build/libwfl_mm_${v}t.so timeout 60 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab$v -- python scripts/ctc_step_loop.py 60 > /tmp/ab$v.log 2>&1
  python - "$v" "$round" "$(find /tmp/ab$v -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[3])):
    n = r["Kernel_Name"]
    if "mitm" in n.lower() or "repair" in n:
        d["mitm" if "mitm" in n.lower() else "repair"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-6s round %s: " % (sys.argv[1], sys.argv[2]) + "  ".join("%s median %.2f min %.2f (n=%d)" % (k, statistics.median(v[5:]), min(v[5:]), len(v) - 5) for k, v in d.items()))
PY
done
done
