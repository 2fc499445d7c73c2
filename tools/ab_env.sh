#!/bin/bash
# interleaved unprofiled A/B of environment settings on the headline step, one gpurun call:
#   bash tools/ab_env.sh "ENV=a" "ENV=b" [bench flags...]   (each configuration three times)
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for cfg in "$A" "$B"; do
    env $cfg python bench.py --plain --reps 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$cfg', round(d['ms_per_step'],4))"
  done
done
