"""Average memory-side bytes per launch of the wfl kernels from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE,
KiB units, collected in separate passes as MI355X_MICROARCH.md prescribes), per configuration.
usage: pmc_traffic.py <dir with pmc_<cfg>_FETCH_SIZE.csv / pmc_<cfg>_WRITE_SIZE.csv>  -> JSON on stdout

Corrections (measured on this box with scripts/pmc_calib.hip, profiles/r02_pmc_calibration.json): FETCH_SIZE reports
0.50 of the bytes for 16 B/lane streams, 4 B/lane streams AND 4 B/lane gathers (per 64-B line touched), i.e. x2
for every read pattern of these kernels; WRITE_SIZE is exact (1.00) for 16 B and 4 B streams and for 400-B rows.
hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE.  These are L2 <-> fabric bytes: reads served by the 256 MiB Infinity Cache
are counted too (the guide: "Infinity-Cache hits appear to be counted"), so re-reads of a tensor that fits it show up
here although they never reach HBM."""
import collections
import csv
import glob
import json
import os
import re
import sys


def per_kernel(path, counter):
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            per_dispatch[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for (name, _), v in per_dispatch.items():
        tot[name] += v
        cnt[name] += 1
    return {k: tot[k] / cnt[k] * 1024.0 for k in tot}


out = {}
for fpath in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*_FETCH_SIZE.csv"))):
    cfg = re.match(r"pmc_(.*)_FETCH_SIZE\.csv", os.path.basename(fpath)).group(1)
    wpath = fpath.replace("FETCH_SIZE", "WRITE_SIZE")
    fetch = per_kernel(fpath, "FETCH_SIZE")
    write = per_kernel(wpath, "WRITE_SIZE") if os.path.exists(wpath) else {}
    kern = {}
    for name in fetch:
        if "wfl::" not in name:
            continue
        short = name.split("wfl::")[1].split("(")[0].split("<")[0]
        k = kern.setdefault(short, dict(fetch_counter_bytes=0.0, write_bytes=0.0))
        k["fetch_counter_bytes"] += fetch[name]
        k["write_bytes"] += write.get(name, 0.0)
    for k in kern.values():
        k["fetch_bytes"] = 2.0 * k["fetch_counter_bytes"]
        k["hbm_bytes"] = k["fetch_bytes"] + k["write_bytes"]
    out[cfg] = dict(kernels=kern)
print(json.dumps(dict(
    source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py <cfg> --targets same`",
    correction="fetch x2 (scripts/pmc_calib.hip: ratio 0.50 for streams and for 4-B gathers per 64-B line), write x1",
    configs=out), indent=1))
