#!/bin/bash
# rocprofv3 kernel stats of the zero-shot inference bench: bash tools/prof_infer.sh
export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -rf /tmp/prof_inf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o p -- python bench.py --workload infer --steps 10 --warmup 3 > gpurun_out/inf_bench.log 2>&1
s=$(find /tmp/prof_inf -name "*kernel_stats.csv" | head -1)
cp $s gpurun_out/r3_infer_kernel_stats.csv
python - << 'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3_infer_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print("%-90s calls %6s avg %8.1f us  %5.1f%%" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
