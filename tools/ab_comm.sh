#!/bin/bash
# One-rank RCCL (COUNTR_FORCE_COMM=1): step time with the bucket all-reduces captured into the step graph vs issued by the host between
# per-phase graphs vs no communication at all.  bash tools/ab_comm.sh
cd $GRAFT_REPO_ROOT
run() { env "$@" COUNTR_BENCH_INIT_PG=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + RANDOM % 200)) python bench.py --gpus 1 --steps 50 --warmup 10 --reps 3 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', 'ms/step', round(d['ms_per_step'],4))"; }
for rep in 1 2; do
  run COUNTR_FORCE_COMM=0
  run COUNTR_FORCE_COMM=1 COUNTR_GRAPH_COMM=1
  run COUNTR_FORCE_COMM=1 COUNTR_GRAPH_COMM=0
done
