"""Where does the bf16 exemplar-CNN gradient error (cos ~0.96 vs the fp32 oracle) come from?  The same backward in the fp32 and the bf16
engine, cosine of the intermediate gradient buffers of the exemplar branch: dy_tok (gradient of the exemplar tokens, produced by the
cross-attention backward), dc4 .. dc1 (gradient of each conv output), and of the four conv weight gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import countr_ref as R, weights as W
import models_mae_cross as mm
MODEL = "mae_vit_base_patch16"
sd = W.make_state_dict(MODEL, seed=0)
B, S = 2, 3
imgs, boxes, gt, mask = W.make_inputs(batch=B, shots=3, seed=1)
res = {}
for prec in ("fp32", "bf16"):
    m = mm.__dict__[MODEL](precision=prec); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.to("cuda")
    m.train(); m.zero_grad()
    out = m(torch.from_numpy(imgs).cuda(), torch.from_numpy(boxes).cuda(), S)
    loss = R.masked_mse_loss(out, torch.from_numpy(gt).cuda(), torch.from_numpy(mask).cuda()); loss.backward()
    torch.cuda.synchronize()
    p = m._engine().plan(B, S, True)
    d = {k: p.buf[k].detach().double().cpu().clone() for k in ("dy_tok", "dc4", "dc3", "dc2", "dc1", "ytok", "c4", "c1") if k in p.buf}
    for k, q in m.named_parameters():
        if k.startswith("decoder_proj") and q.grad is not None:
            d["grad:" + k] = q.grad.detach().double().cpu().clone()
    res[prec] = d
for k in res["fp32"]:
    a, b = res["fp32"][k].reshape(-1), res["bf16"][k].reshape(-1)
    print("%-32s cos %.5f  norm ratio %.4f" % (k, (a @ b / (a.norm() * b.norm())).item(), (b.norm() / a.norm()).item()))
