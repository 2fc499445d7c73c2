"""Per-tensor gradient error of the bf16 engine against the fp32 oracle gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import countr_ref as R, weights as W
from test_trainer_gpu import make, NAME
m, sd = make("fp32")
imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=10)
m.train(); m.zero_grad()
out = m(torch.from_numpy(imgs).cuda(), torch.from_numpy(boxes).cuda(), 3)
loss = R.masked_mse_loss(out, torch.from_numpy(gt).cuda(), torch.from_numpy(mask).cuda()); loss.backward()
_, rloss, rg = R.loss_and_grads(sd, imgs, boxes, gt, mask, 3, NAME)
_, rloss64, rg64 = R.loss_and_grads(sd, imgs, boxes, gt, mask, 3, NAME, dtype=torch.float64)
for k, p in m.named_parameters():
    if p.grad is None or rg.get(k) is None: continue
    ref = rg64[k].double(); got = p.grad.detach().cpu().double(); r32 = rg[k].double()
    rms = ref.pow(2).mean().sqrt().item()
    print("%-42s rms %.2e  hip-vs-f64 max %.2e (%.3f rms)   cpu32-vs-f64 max %.2e (%.3f rms)" % (k, rms, (got-ref).abs().max().item(), (got-ref).abs().max().item()/rms, (r32-ref).abs().max().item(), (r32-ref).abs().max().item()/rms))
