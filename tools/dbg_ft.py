import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import models_mae_cross
from countr_amd.trainer import FinetuneStep
from countr_amd.synthetic import make_batch
from countr_amd.parallel import shared_shot_num
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
graph = (sys.argv[2] != "nograph") if len(sys.argv) > 2 else True
torch.manual_seed(0)
m = models_mae_cross.mae_vit_base_patch16(precision=prec).to("cuda").train()
step = FinetuneStep(m, batch=8, lr=3e-6, use_graph=graph)
for it in range(16):
    S = shared_shot_num(it, seed=0)
    imgs, boxes, gt, mask = make_batch(8, shots=3, seed=it, device="cuda")
    step.load(imgs, boxes, gt, mask, S)
    s = step.step(S).float().cpu().numpy()
    print(it, "S", S, "loss %.5f" % s[0], "cnt", s[1:3], flush=True)
