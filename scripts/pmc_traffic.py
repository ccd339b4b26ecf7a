"""Average HBM-side bytes per launch of the wfl kernels from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, KiB units, collected separately as MI355X_MICROARCH.md prescribes).
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv>  -> JSON on stdout"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        per_dispatch[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (name, _), v in per_dispatch.items():
        tot[name] += v
        cnt[name] += 1
    return {k: tot[k] / cnt[k] * 1024.0 for k in tot}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for name in fetch:
    if "wfl::" not in name:
        continue
    short = name.split("wfl::")[1].split("(")[0].split("<")[0]
    out[short] = dict(fetch_bytes_raw=fetch[name], fetch_bytes_corrected=2 * fetch[name],
                      write_bytes=write.get(name, 0.0), hbm_bytes=fetch[name] + write.get(name, 0.0))
print(json.dumps(dict(
    source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py cfg2",
    unit_note="counters are KiB; fetch_bytes_corrected applies the x2 gfx950 correction that "
              "MI355X_MICROARCH.md derives for 16-B/lane coalesced streams; hbm_bytes = raw fetch + write "
              "(these kernels gather 4 B/lane, for which the correction is uncalibrated)",
    kernels=out), indent=1))
