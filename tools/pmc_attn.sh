#!/bin/bash
# PMC passes for the attention forward (run on the GPU box).  Usage: bash tools/pmc_attn.sh <impl 1|2> <outfile>
impl=$1; out=$2
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmcA /tmp/pmcB /tmp/pmcC
COUNTR_ATTN_IMPL=$impl rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/pmcA -o a -- python tools/bench_attn.py --one > /dev/null 2>&1
COUNTR_ATTN_IMPL=$impl rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d /tmp/pmcB -o b -- python tools/bench_attn.py --one > /dev/null 2>&1
COUNTR_ATTN_IMPL=$impl rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL -d /tmp/pmcC -o c -- python tools/bench_attn.py --one > /dev/null 2>&1
(for d in /tmp/pmcA /tmp/pmcB /tmp/pmcC; do f=$(find $d -name "*.db" | head -1); echo "## $d $f"; python tools/pmc_summary.py $f | grep -A12 -i "fa_fwd\|flash_attn_fwd"; done) > $out 2>&1
