# usage: kstats.sh <bench args...> : per-kernel averages of a bench run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python bench.py "$@" --targets same --steps 30 --no-extras --no-cpu-baseline > gpurun_out/ks.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ks/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)): print("%-70s %5s %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
