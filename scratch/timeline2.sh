# usage: timeline2.sh <workload> <fresh|same> : one steady-state iteration's kernel + copy timeline
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- python bench.py --workload $1 --targets $2 --steps 30 --no-extras --no-cpu-baseline > gpurun_out/tl.log 2>&1
python - <<'PY'
import csv, glob
k = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("gpurun_out/tl/**/*memory_copy_trace.csv", recursive=True)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"][:60]) for r in csv.DictReader(open(k))]
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "?"))))
rows.sort()
names = [r[3] for r in rows]
idx = [i for i, n in enumerate(names) if "gather" in n]
i0, i1 = idx[-4], idx[-3]
t0 = rows[i0][0]
for s, e, q, n in rows[i0 - 4:i1]:
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  {q:>5s}  {n}")
print("iteration:", (rows[i1][0] - t0) / 1e3, "us")
PY
