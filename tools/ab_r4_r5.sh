#!/bin/bash
# Same-box, interleaved, unprofiled A/B of an older tree against the working tree (profiles/r5_ab_r4_tree.txt).  Prepare once in the build
# container:  mkdir _r4tree && git archive b61323b | tar -x -C _r4tree && (cd _r4tree && python -m countr_amd.build)   (_r4tree/ is not tracked)
# then:       gpurun -- 'bash tools/ab_r4_r5.sh'
cd $GRAFT_REPO_ROOT
run() { (cd $1; python bench.py --plain --reps 3 $2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 $2', round(d['ms_per_step'],4))"); }
for i in 1 2 3; do run _r4tree ""; run . ""; done
(cd _r4tree; python bench.py --workload pretrain --batch 16 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('r4 pretrain16', d['ms_per_step'])")
python bench.py --workload pretrain --batch 16 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('r5 pretrain16', d['ms_per_step'])"
(cd _r4tree; python bench.py --workload infer --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('r4 infer', d['ms_per_step'])")
python bench.py --workload infer --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('r5 infer', d['ms_per_step'])"
