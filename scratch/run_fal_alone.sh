cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fal_alone; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fal_alone -- python scratch/fal_alone.py > gpurun_out/fal_alone.log 2>&1
f=$(find gpurun_out/fal_alone -name "*kernel_stats.csv" | head -1); cut -c1-60,200- $f | head -12; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])): print(r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3)
PY
