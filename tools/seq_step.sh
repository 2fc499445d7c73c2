#!/bin/bash
# One headline step's launch sequence (kernel, workgroups, duration, gap to the previous launch) from a rocprofv3 kernel trace of the
# graph-replayed bench: bash tools/seq_step.sh <tag> [env assignments...]  ->  gpurun_out/<tag>_step_seq.txt, <tag>_step_breakdown.txt
tag=$1; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/seq_$tag
EXTRA=("$@"); if [ ${#EXTRA[@]} -eq 0 ]; then EXTRA=("SEQ_STEP=1"); fi
env "${EXTRA[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/seq_$tag -o p -- python bench.py --steps 10 --warmup 3 --reps 1 --plain > gpurun_out/${tag}_bench.log 2>&1
f=$(find /tmp/seq_$tag -name "*kernel_trace.csv" | head -1)
s=$(find /tmp/seq_$tag -name "*kernel_stats.csv" | head -1)
cp $s gpurun_out/${tag}_kernel_stats.csv
python tools/step_breakdown.py $f 8 adamw_kernel 0 seq > gpurun_out/${tag}_step_seq_full.txt 2>&1
grep -n "wgs=" gpurun_out/${tag}_step_seq_full.txt | grep "gap" > gpurun_out/${tag}_step_seq.txt
grep -v "gap" gpurun_out/${tag}_step_seq_full.txt > gpurun_out/${tag}_step_breakdown.txt
rm gpurun_out/${tag}_step_seq_full.txt
head -5 gpurun_out/${tag}_step_breakdown.txt
