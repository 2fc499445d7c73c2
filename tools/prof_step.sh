#!/bin/bash
# rocprofv3 kernel trace of the headline bench + per-step breakdown: bash tools/prof_step.sh <tag> [extra env assignments...]
# writes gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_step_breakdown.txt
tag=$1; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_$tag
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b32 --no-other > gpurun_out/${tag}_bench.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
s=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $s gpurun_out/${tag}_kernel_stats.csv
python tools/step_breakdown.py $f 8 adamw_kernel 18 > gpurun_out/${tag}_step_breakdown.txt 2>&1
head -60 gpurun_out/${tag}_step_breakdown.txt
