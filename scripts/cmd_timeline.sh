#!/bin/bash
# GPU box: kernel timeline of one steady-state step of an arbitrary command (start offset, duration, queue, kernel).
# usage: scripts/cmd_timeline.sh <name of the step's first kernel (prefix)> <command...>
first=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf /tmp/tlc
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlc -- "$@" > /tmp/tlc.log 2>&1
python - "$(find /tmp/tlc -name '*kernel_trace.csv' | head -1)" "$first" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("wfl::")[-1][:60], r.get("Queue_Id", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith(sys.argv[2])]
a, b = starts[-3], starts[-2]
t0 = rows[a][0]
for s, e, n, q in rows[a:b]:
    print("  +%8.1f us  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
print("  step span %.1f us (start of this step to start of the next)" % ((rows[b][0] - t0) / 1e3))
PY
