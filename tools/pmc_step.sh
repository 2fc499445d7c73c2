#!/bin/bash
# PMC passes over the whole finetune step (eager launches: python bench.py --steps 2 --warmup 2 --no-graph), one rocprofv3 run per counter
# group (--kernel-trace only; FETCH_SIZE and WRITE_SIZE do not fit one TCC pass), summarised per (kernel, grid) and per family by
# tools/pmc_report.py.  bash tools/pmc_step.sh <tag> [env assignments...]  ->  gpurun_out/<tag>_pmc_groups.json, <tag>_pmc_report.txt
tag=${1:-pmc}; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
dbs=""
run() {
  rm -rf /tmp/pmcs_$1
  env "${EXTRA[@]}" rocprofv3 --kernel-trace --pmc $2 -d /tmp/pmcs_$1 -o p -- python bench.py --steps 3 --warmup 2 --reps 1 --no-graph --plain > /tmp/pmcs_$1.log 2>&1
  f=$(find /tmp/pmcs_$1 -name "*.db" | head -1)
  dbs="$dbs $f"
}
EXTRA=("$@"); if [ ${#EXTRA[@]} -eq 0 ]; then EXTRA=("PMC_STEP=1"); fi
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
run lds "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python tools/pmc_report.py gpurun_out/${tag} $dbs > gpurun_out/${tag}_pmc_report.txt 2>&1
head -80 gpurun_out/${tag}_pmc_report.txt
