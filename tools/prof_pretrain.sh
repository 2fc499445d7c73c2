#!/bin/bash
# rocprofv3 kernel trace of the MAE pretraining bench + per-step breakdown: bash tools/prof_pretrain.sh [tag] [ENV=...]
tag=${1:-r3}; shift
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -rf /tmp/prof_pre
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pre -o p -- python bench.py --workload pretrain --steps 10 --warmup 3 > gpurun_out/pre_bench.log 2>&1
f=$(find /tmp/prof_pre -name "*kernel_trace.csv" | head -1)
python tools/step_breakdown.py $f 8 adamw_kernel 0 > gpurun_out/${tag}_step_breakdown_pretrain.txt 2>&1
head -${LINES_SHOWN:-30} gpurun_out/${tag}_step_breakdown_pretrain.txt
