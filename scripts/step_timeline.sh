#!/bin/bash
# GPU box: kernel timeline of one steady-state step of a bench configuration (start offset, duration, kernel)
# usage: scripts/step_timeline.sh <bench args>
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python bench.py "$@" --steps 12 --warmup 3 --no-cpu-baseline --no-extras > /tmp/tl.log 2>&1
python - "$(find /tmp/tl -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("wfl::")[-1][:44], r.get("Queue_Id", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# a step starts at each gather kernel (or the first kernel containing "mitm" / "dense_fast_chain")
starts = [i for i, r in enumerate(rows) if r[2].startswith(("gather", "ctc_mitm", "upload"))]
if len(starts) < 4:
    starts = list(range(0, len(rows), max(1, len(rows) // 12)))
a, b = starts[-3], starts[-2]
t0 = rows[a][0]
for s, e, n, q in rows[a:b]:
    print("  +%8.1f us  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
print("  step span %.1f us (start of this step to start of the next)" % ((rows[b][0] - t0) / 1e3))
PY
