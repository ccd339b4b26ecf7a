# usage: timeline.sh <workload> : prints one steady-state iteration's kernel timeline (start offset, duration, stream)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --workload $1 --targets same --steps 20 --no-extras --no-cpu-baseline > gpurun_out/tl.log 2>&1
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last occurrence of the first kernel name pattern of an iteration
names = [r["Kernel_Name"] for r in rows]
key = "gather" if any("gather" in n for n in names) else names[0]
idx = [i for i, n in enumerate(names) if key in n]
i0, i1 = idx[-3], idx[-2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?'):>3s}  {r['Kernel_Name'][:70]}")
print("iteration:", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us")
PY
