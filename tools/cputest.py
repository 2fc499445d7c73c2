import time, torch, os, sys
sys.path.insert(0,'.')
from oracle import countr_ref as R, weights as W
sd = W.make_state_dict("mae_vit_base_patch16", seed=0)
imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=0)
p = R.Params(sd)
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    R.forward(p, imgs[:1], boxes[:1], 3)
    t0=time.time(); R.forward(p, imgs, boxes, 3); dt=time.time()-t0
    print(th, "threads fwd B=2: %.2fs" % dt, flush=True)
    if dt > 30: break
