#!/bin/bash
# usage (on the GPU box): scripts/prof_cmd.sh <tag> <command...>   -> per-kernel averages of any command
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- "$@" > gpurun_out/prof_$tag.log 2>&1
python scripts/kstats.py gpurun_out/prof_$tag
