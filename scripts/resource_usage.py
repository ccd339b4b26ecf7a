#!/usr/bin/env python3
"""Kernel resource table (registers, scratch, spills, LDS, occupancy) of one .hip file as the gfx950 compiler
reports it (-Rpass-analysis=kernel-resource-usage), plus the number of scratch_* / v_writelane / v_readlane
instructions in each kernel's device assembly.

    python scripts/resource_usage.py ctc_kernels.hip [filter] [-- extra hipcc flags]

Prints one line per kernel whose demangled name contains `filter`.  Used for profiles/r04_resource_usage.txt."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "gtn_applications_amd", "csrc")


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    src = argv[0]
    flt = argv[1] if len(argv) > 1 else ""
    tmp = tempfile.mkdtemp(prefix="wflres")
    asm = os.path.join(tmp, "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "--offload-arch=gfx950",
           "-Wno-unused-function", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", asm,
           "-Rpass-analysis=kernel-resource-usage"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
    if res.returncode != 0:
        sys.stderr.write(res.stderr[-4000:])
        sys.exit(1)
    # per-kernel instruction counts from the assembly
    counts = {}
    cur = None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            counts[cur] = {"scratch": 0, "writelane": 0, "readlane": 0, "lines": 0}
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        s = line.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        c = counts[cur]
        c["lines"] += 1
        if s.startswith("scratch_"):
            c["scratch"] += 1
        elif s.startswith("v_writelane"):
            c["writelane"] += 1
        elif s.startswith("v_readlane"):
            c["readlane"] += 1
    blocks = re.split(r"remark: Function Name: ", res.stderr)[1:]
    print("%-86s %5s %5s %5s %8s %4s %6s %6s %7s | %7s %8s %9s %7s" % (
        "kernel", "VGPR", "AGPR", "SGPR", "scratchB", "occ", "sSpill", "vSpill", "LDS", "instr", "scratch*", "writelane", "readlane"))
    for b in blocks:
        name = b.split()[0]
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("wfl::", "").replace("(anonymous namespace)::", "")
        dn = re.sub(r"\(.*$", "", dn)
        if dn.startswith("void "):
            dn = dn[5:]
        if flt and flt not in dn:
            continue

        def g(k):
            m = re.search(k + r": (\S+)", b)
            return m.group(1) if m else "?"
        c = counts.get(name, {"scratch": -1, "writelane": -1, "readlane": -1, "lines": -1})
        print("%-86s %5s %5s %5s %8s %4s %6s %6s %7s | %7d %8d %9d %7d" % (
            dn[:86], g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
            g("SGPRs Spill"), g("VGPRs Spill"), g(r"LDS Size \[bytes/block\]"), c["lines"], c["scratch"], c["writelane"], c["readlane"]))
    if os.environ.get("WFL_KEEP_ASM"):
        print("asm kept:", asm)


if __name__ == "__main__":
    main()
