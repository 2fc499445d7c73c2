#!/bin/bash
# PMC passes for the encoder Linear GEMMs INSIDE the finetune step: qkv / fc1 (plain 128x128 two-stage kernel, two workgroups per CU)
# and proj / fc2 (wave-specialised).  Matrix pipe and LDS port activity against the busy cycles.
# bash tools/pmc_gemm.sh <outdir>; separate passes per counter group, --kernel-trace only.
out=${1:-gpurun_out/pmc_gemm}; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { rm -rf /tmp/pmc_$1; rocprofv3 --kernel-trace --pmc $2 -d /tmp/pmc_$1 -o p -- python bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > /dev/null 2>&1; f=$(find /tmp/pmc_$1 -name "*.db" | head -1); echo "## pass $1: $2"; python tools/pmc_summary.py $f | grep -A10 -E "gemm_kernelItLi0ELi0ELi2ELi2ELi2ELi4ELi0E|gemm_kernelItLi0ELi0ELi3ELi2ELi2ELi4ELi4E"; }
(run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
 run lds "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_BF SQ_INSTS_MFMA") > $out/pmc.txt 2>&1   # (a TA_* pass did not finish within 15 minutes on this stack: do not add one)
cat $out/pmc.txt
