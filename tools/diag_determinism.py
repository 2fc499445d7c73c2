"""Run-to-run determinism of the finetune step's gradients: which tensors differ between two identical runs?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_trainer_gpu import make, W
from countr_amd.trainer import FinetuneStep
graph = os.environ.get("GRAPH", "1") == "1"
B = int(os.environ.get("BATCH", "2"))
runs = []
for r in range(3):
    m, _ = make("bf16")
    step = FinetuneStep(m, batch=B, lr=1e-4, use_graph=graph)
    gs = []
    for it in range(3):
        arrs = W.make_inputs(batch=B, shots=3, seed=30 + it)
        step.load(*[torch.from_numpy(a).cuda() for a in arrs], 3)
        step.step(3)
        torch.cuda.synchronize()
        gs.append(step.eng.G.clone())
    runs.append((gs, step))
gs0, step = runs[0]
lay = step.eng.layout
for r in (1, 2):
    for it in range(3):
        d = (runs[r][0][it] != gs0[it])
        if d.any():
            print("run %d step %d: %d elements differ" % (r, it, int(d.sum())))
            names = []
            for name, off in lay.off.items():
                o = off - lay.train_start
                if o < 0:
                    continue
                n = 1
                for q in lay.shapes[name]:
                    n *= q
                if d[o:o + n].any():
                    names.append((name, int(d[o:o + n].sum()), n))
            print(names[:40])
            break
print("done")
