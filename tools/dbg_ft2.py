import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import models_mae_cross
from countr_amd.trainer import FinetuneStep
from countr_amd.synthetic import make_batch
from countr_amd.parallel import shared_shot_num
graph = sys.argv[1] != "nograph"
sync_every = int(sys.argv[2])
torch.manual_seed(0)
m = models_mae_cross.mae_vit_base_patch16(precision="bf16").to("cuda").train()
step = FinetuneStep(m, batch=8, lr=3e-6, use_graph=graph)
batches = [make_batch(8, shots=3, seed=it, device="cuda") for it in range(8)]
for it in range(60):
    S = shared_shot_num(it, seed=0)
    step.load(*batches[it % 8], S)
    sums = step.step(S)
    if (it + 1) % sync_every == 0:
        s = sums.float().cpu().numpy()
        print(it, "S", S, "loss %.5f" % s[0], flush=True)
