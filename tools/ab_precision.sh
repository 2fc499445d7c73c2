#!/bin/bash
# per-kernel step breakdown of the headline step in bf16 and in fp16 mode, same box: bash tools/ab_precision.sh -> gpurun_out/prec_{bf16,fp16}_breakdown.txt
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for p in bf16 fp16; do
  rm -rf /tmp/prec_$p
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prec_$p -o p -- python bench.py --steps 10 --warmup 3 --reps 1 --plain --precision $p > /dev/null 2>&1
  python tools/step_breakdown.py $(find /tmp/prec_$p -name "*kernel_trace.csv" | head -1) 8 adamw_kernel 0 > gpurun_out/prec_${p}_breakdown.txt 2>&1
  head -1 gpurun_out/prec_${p}_breakdown.txt
done
