#!/bin/bash
# PMC passes for the fused attention forward INSIDE the finetune step (run on the GPU box: bash tools/pmc_attn.sh <outdir>).
# Separate passes per counter group (SQ: 8 slots; FETCH_SIZE and WRITE_SIZE do not fit one TCC pass), --kernel-trace only.
out=${1:-gpurun_out/pmc_attn}; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { rm -rf /tmp/pmc_$1; rocprofv3 --kernel-trace --pmc $2 -d /tmp/pmc_$1 -o p -- python bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > /dev/null 2>&1; f=$(find /tmp/pmc_$1 -name "*.db" | head -1); echo "## pass $1: $2"; python tools/pmc_summary.py $f | grep -A9 "fa_fwd_pipe_kernelILi64"; }
(run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
 run sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
 run fetch "FETCH_SIZE"
 run write "WRITE_SIZE") > $out/pmc.txt 2>&1
cat $out/pmc.txt
