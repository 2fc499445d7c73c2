#!/bin/bash
# Ablation timings of the pipelined attention forward (COUNTR_FA_ABL, see flash_attn_fwd.hip); results of ablated runs are wrong by design.
cd "$(dirname "$0")/.."
export BENCH_ATTN_SHAPES="8,576,12,64;32,576,12,64;1,576,12,64"
for a in 0 1 2 3 4 5 6; do echo "== COUNTR_FA_ABL=$a"; COUNTR_FA_ABL=$a COUNTR_ATTN_IMPL=2 python tools/bench_attn.py --one 2>&1 | grep "^B"; done
