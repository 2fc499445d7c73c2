#!/bin/bash
# A/B of the headline step inside ONE gpurun call (box-to-box variance is +-3 %): bash tools/ab_step.sh "ENV=a" "ENV=b" ... (each run twice, interleaved)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 50 --warmup 10 --reps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$cfg', 'ms/step', round(d['ms_per_step'],4), 'attn_us', round(d['roofline']['us_per_launch'],2))"
  done
done
