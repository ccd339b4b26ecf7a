"""Print the per-kernel averages of a rocprofv3 --kernel-trace --stats --output-format csv run."""
import csv, glob, sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        print(d)
        for r in list(csv.DictReader(open(f)))[:10]:
            print(f"  {r['Name'][:64]:64s} n={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):5.1f}%")
