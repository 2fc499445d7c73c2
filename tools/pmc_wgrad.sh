#!/bin/bash
# PMC passes over the conv weight-gradient microbench (tools/bench_wgrad.py 1: the 192 x 192 layer in every form), one rocprofv3 run per
# counter group, csv output; prints the per-kernel averages.  bash tools/pmc_wgrad.sh <tag>
tag=${1:-cwg}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  rm -rf /tmp/pw_$1
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pw_$1 -o p -- python tools/bench_wgrad.py 1 > /tmp/pw_$1.log 2>&1
  f=$(find /tmp/pw_$1 -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' >> gpurun_out/${tag}_pmc_wgrad.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "cwg" not in k and "gemm_kernel" not in k:
        continue
    acc[k[:70] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("    %-28s n %3d  avg %16.1f" % (c, len(v), sum(v) / len(v)))
PY
}
: > gpurun_out/${tag}_pmc_wgrad.txt
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
run lds "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES"
cat gpurun_out/${tag}_pmc_wgrad.txt
