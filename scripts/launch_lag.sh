#!/bin/bash
# GPU box: for one steady-state step, when the HOST called each launch and when the GPU started it (is a gap in front of
# a kernel the host's or the GPU's?)   usage: scripts/launch_lag.sh <bench args>
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf /tmp/ll
timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/ll -- python bench.py "$@" --steps 12 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ll.log 2>&1
ls /tmp/ll/*/ 2>/dev/null | head
python - "$(find /tmp/ll -name '*kernel_trace.csv' | head -1)" "$(find /tmp/ll -name '*hip_api_trace.csv' | head -1)" <<'PY'
import csv, sys
k = list(csv.DictReader(open(sys.argv[1])))
a = list(csv.DictReader(open(sys.argv[2])))
api = {r["Correlation_Id"]: r for r in a if "Launch" in r["Function"]}
rows = []
for r in k:
    h = api.get(r["Correlation_Id"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("wfl::")[-1][:40], int(h["Start_Timestamp"]) if h else 0))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith(("gather", "ctc_mitm", "upload"))]
a0, b0 = starts[-3], starts[-2]
t0 = rows[a0][0]
print("   GPU start   duration   host call (all relative to the step's first kernel start)")
for s, e, n, h in rows[a0 - 3:b0 + 2]:
    print("  +%8.1f us  %8.1f us   host %+9.1f us   %s" % ((s - t0) / 1e3, (e - s) / 1e3, (h - t0) / 1e3, n))
PY
