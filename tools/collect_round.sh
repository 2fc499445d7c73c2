#!/bin/bash
# End-of-round evidence from ONE box on HEAD (run through gpurun): bash tools/collect_round.sh [tag]  ->  gpurun_out/<tag>_*
# bench lines of the three single-GPU workloads, kernel stats + per-step breakdown + one step's launch sequence of the headline step,
# the counter passes over the eager step (tools/pmc_step.sh), per-step breakdowns of pretraining / inference, the 256x256 kernel's stamps.
tag=${1:-r6}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
python bench.py --precision fp16 --no-other --no-cpu-baseline --no-families --no-b32 > gpurun_out/${tag}_bench_line_fp16.json 2>> gpurun_out/${tag}_bench.err
python bench.py --workload pretrain --no-cpu-baseline > gpurun_out/${tag}_bench_pretrain_line.json 2>> gpurun_out/${tag}_bench.err
python bench.py --workload infer --no-cpu-baseline > gpurun_out/${tag}_bench_infer_line.json 2>> gpurun_out/${tag}_bench.err
bash tools/prof_step.sh ${tag}_bench > /dev/null 2>&1            # kernel stats of the driver-style command (incl. roofline passes)
bash tools/seq_step.sh ${tag} > /dev/null 2>&1                   # plain headline steps: breakdown + sequence
bash tools/pmc_step.sh ${tag} > /dev/null 2>&1
rm -rf /tmp/prof_pre /tmp/prof_inf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pre -o p -- python bench.py --workload pretrain --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python tools/step_breakdown.py $(find /tmp/prof_pre -name "*kernel_trace.csv" | head -1) 8 adamw_kernel 0 > gpurun_out/${tag}_step_breakdown_pretrain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o p -- python bench.py --workload infer --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_infer_kernel_stats.csv
python -c "
import json
for n in ('bench_line', 'bench_line_fp16', 'bench_pretrain_line', 'bench_infer_line'):
    d = json.load(open('gpurun_out/${tag}_%s.json' % n)); print(n, round(d['ms_per_step'], 4), round(d['value'], 1), d['unit'])
d = json.load(open('gpurun_out/${tag}_bench_line.json')); print('attention', d['roofline']['us_per_launch'], d['roofline']['frac'], 'b32', d['roofline_b32']['frac']); print(d['other_workloads']['pretrain']['ms_per_step'], d['other_workloads']['infer']['ms_per_32_windows'])"
head -4 gpurun_out/${tag}_step_breakdown.txt
