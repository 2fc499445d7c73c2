"""Where does the bf16 forward lose magnitude at shot_num = 0 (counts ~6 % low, gradient norms 0.84-0.96 of fp32)?  Same forward in the fp32
and the bf16 engine: relative error / magnitude ratio of the decoder output and of every density-head buffer, and |group mean| / sigma of
the conv outputs in front of the GroupNorms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import weights as W
import models_mae_cross as mm
MODEL = "mae_vit_base_patch16"
sd = W.make_state_dict(MODEL, seed=0)
B = 2
S = int(sys.argv[1]) if len(sys.argv) > 1 else 0
imgs, boxes, gt, mask = W.make_inputs(batch=B, shots=3, seed=2)
res = {}
for prec in ("fp32", "bf16"):
    m = mm.__dict__[MODEL](precision=prec); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.to("cuda")
    m.eval()
    with torch.no_grad():
        out = m(torch.from_numpy(imgs).cuda(), torch.from_numpy(boxes).cuda(), S)
    torch.cuda.synchronize()
    p = m._engine().plan(B, S, False)
    keys = [k for k in p.buf if k.startswith(("hc", "hu", "o1", "out", "latent", "dx"))]
    res[prec] = {k: p.buf[k].detach().double().cpu().clone() for k in keys}
print("shot_num", S)
for k in sorted(res["fp32"]):
    a, b = res["fp32"][k].reshape(-1), res["bf16"][k].reshape(-1)
    if a.numel() != b.numel():
        continue
    line = "%-8s rel err %.3e  cos %.6f  sum ratio %.4f" % (k, ((a - b).norm() / a.norm()).item(), (a @ b / (a.norm() * b.norm())).item(), (b.sum() / a.sum()).item())
    if k.startswith("hc"):
        x = res["fp32"][k].reshape(B, -1, 8, 32)                 # [B, HW, G, C/G]
        mu = x.mean((1, 3)); sg = x.permute(0, 2, 1, 3).reshape(B, 8, -1).std(2)
        line += "   max |group mean| / sigma %.2f" % (mu.abs() / sg).max().item()
        xc = res["fp32"][k].reshape(B, -1, 256)
        line += "   max |channel mean| / channel sigma %.1f" % (xc.mean(1).abs() / xc.std(1)).max().item()
    print(line)
