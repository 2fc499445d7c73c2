#!/bin/bash
# End-of-round evidence on HEAD, one gpurun call: the three bench lines, the rocprofv3 kernel stats + per-step breakdown of the headline
# command, and the A/B of the round's switches.  bash tools/collect_final.sh  -> gpurun_out/final/ (copy into profiles/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; o=gpurun_out/final; mkdir -p $o
python bench.py > $o/bench.log 2>&1; grep '^{' $o/bench.log | tail -1 > $o/r3_bench_line.json
python bench.py --workload pretrain 2>&1 | grep '^{' | tail -1 > $o/r3_bench_pretrain_line.json
python bench.py --workload infer 2>&1 | grep '^{' | tail -1 > $o/r3_bench_infer_line.json
bash tools/prof_step.sh r3 > /dev/null 2>&1; cp gpurun_out/r3_step_breakdown.txt $o/; cp gpurun_out/r3_kernel_stats.csv $o/r3_bench_kernel_stats.csv
(echo "# bash tools/ab_step.sh <configs>: python bench.py --steps 50 --warmup 10 --reps 3 --no-cpu-baseline per configuration, every configuration twice, interleaved, ONE gpurun call (one box)"
 bash tools/ab_step.sh "COUNTR_LEAN=0 COUNTR_LN_FOLD=0 COUNTR_DGRAD_T=0" "COUNTR_LN_FOLD=0" "COUNTR_LEAN_WGRAD=0" "COUNTR_DGRAD_T=0" "COUNTR_NOOP=1") > $o/r3_step_ab.txt 2>&1
ls -la $o
