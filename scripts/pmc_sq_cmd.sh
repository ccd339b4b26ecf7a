#!/bin/bash
# usage (GPU box): scripts/pmc_sq_cmd.sh <tag> <kernel-substring> <command...>
# SQ counter passes (one rocprofv3 --pmc run per group) for one kernel of an arbitrary command; per-launch averages.
tag=$1; kern=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; cd "$R"
O=gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_INSTS_VALU_TRANS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- "$@" > $O/g$i.log 2>&1
  f=$(find $O/g$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$kern" <<'PY'
import csv, sys, collections
tot=collections.defaultdict(float); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in tot: print(f"  {k:28s} {tot[k]/len(n[k]):16.0f}  (per launch, {len(n[k])} launches)")
PY
done
