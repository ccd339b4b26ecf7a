#!/usr/bin/env python3
"""Where a kernel's scratch / v_writelane / v_readlane instructions come from: counts per source line, from a -g1 -S build.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -g1 <file>.hip -o /tmp/k.s
    python scripts/spill_sites.py /tmp/k.s <substring of the mangled kernel name>"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().splitlines()
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[m.group(1)] = (m.group(3) or m.group(2)).split("/")[-1]
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if not m or sys.argv[2] not in m.group(1):
        continue
    end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
    loc, kinds = None, {"scratch": Counter(), "v_writelane": Counter(), "v_readlane": Counter()}
    for j in range(i, end):
        t = lines[j].strip()
        mm = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if mm:
            loc = (files.get(mm.group(1), mm.group(1)), int(mm.group(2)))
        for k in kinds:
            if t.startswith(k):
                kinds[k][loc] += 1
    print(m.group(1)[:100])
    for k, c in kinds.items():
        print("  %s: %d  %s" % (k, sum(c.values()), sorted(c.items(), key=lambda kv: -kv[1])[:8]))
