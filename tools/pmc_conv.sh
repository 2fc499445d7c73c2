#!/bin/bash
# PMC passes for the 192x192 density-head convolution (implicit GEMM, 128x256 tile, 8 compute + 4 loader waves) INSIDE the finetune
# step (run on the GPU box: bash tools/pmc_conv.sh <outdir>).  Separate passes per counter group, --kernel-trace only.
out=${1:-gpurun_out/pmc_conv}; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { rm -rf /tmp/pmc_$1; rocprofv3 --kernel-trace --pmc $2 -d /tmp/pmc_$1 -o p -- python bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > /dev/null 2>&1; f=$(find /tmp/pmc_$1 -name "*.db" | head -1); echo "## pass $1: $2"; python tools/pmc_summary.py $f | grep -A12 "gemm_kernelItLi2ELi0ELi3ELi2ELi4ELi4ELi4"; }
(run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
 run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 run fetch "FETCH_SIZE"
 run write "WRITE_SIZE") > $out/pmc.txt 2>&1
cat $out/pmc.txt
