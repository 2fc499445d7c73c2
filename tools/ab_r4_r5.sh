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
