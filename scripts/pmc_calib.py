"""Ratios counter / known bytes for scripts/pmc_calib.hip: usage
  pmc_calib.py <stdout of pmc_calib> <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv>"""
import collections
import csv
import json
import re
import sys

known = dict(re.findall(r"(\w+)=(\d+)", open(sys.argv[1]).read()))
known = {k: int(v) for k, v in known.items()}


def per_kernel(path, counter):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            per[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])] += float(r["Counter_Value"])
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for (name, _), v in per.items():
        tot[name] += v
        cnt[name] += 1
    return {k: tot[k] / cnt[k] * 1024.0 for k in tot}


f, w = per_kernel(sys.argv[2], "FETCH_SIZE"), per_kernel(sys.argv[3], "WRITE_SIZE")
B = known["bytes"]
out = {
    "read16_stream": dict(counter=f.get("read16_stream"), known=B),
    "read4_stream": dict(counter=f.get("read4_stream"), known=B),
    "read4_gather_vs_useful": dict(counter=f.get("read4_gather"), known=known["gather_useful"]),
    "read4_gather_vs_lines64": dict(counter=f.get("read4_gather"), known=known["gather_lines64"]),
    "read4_gather_vs_lines128": dict(counter=f.get("read4_gather"), known=known["gather_lines128"]),
    "write16_stream": dict(counter=w.get("write16_stream"), known=B),
    "write4_stream": dict(counter=w.get("write4_stream"), known=B),
    "write_rows_400B": dict(counter=w.get("write_rows"), known=known["rowwrite_bytes"]),
}
for v in out.values():
    v["ratio"] = None if not v["counter"] else v["counter"] / v["known"]
print(json.dumps(out, indent=1))
