import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch, torch.nn.functional as F
from oracle import countr_ref as R, weights as W
from test_trainer_gpu import make, NAME
m, sd = make("fp32")
B, S = 2, 3
imgs, boxes, gt, mask = W.make_inputs(batch=B, shots=3, seed=10)
m.train(); m.zero_grad()
out = m(torch.from_numpy(imgs).cuda(), torch.from_numpy(boxes).cuda(), 3)
loss = R.masked_mse_loss(out, torch.from_numpy(gt).cuda(), torch.from_numpy(mask).cuda()); loss.backward()
eng = m._eng; buf = eng.plans[(B, S, True)].buf
dt = torch.float64
p = R.Params(sd, dt)
cfg = W.CONFIGS[NAME]
with torch.no_grad():
    latent = R.forward_encoder(p, torch.from_numpy(imgs).to(dt), cfg)
bx = torch.from_numpy(boxes).to(dt)
# CNN with intermediates as leaves
inter = {}
ys = []
x0 = bx.reshape(B * S, 3, 64, 64)   # row b*S+s
y = x0
for li, pool in ((1, True), (2, True), (3, True), (4, False)):
    c = F.conv2d(y, p["decoder_proj%d.0.weight" % li], p["decoder_proj%d.0.bias" % li], padding=1)
    c.retain_grad() if c.requires_grad else None
    c = c.detach().requires_grad_(True); inter["c%d" % li] = c
    a = R.instance_norm_relu(c)
    y = R.max_pool2(a) if pool else a.mean((2, 3), keepdim=True)
    if pool:
        y = y.detach().requires_grad_(True); inter["p%d" % li] = y
tok = y.reshape(B, S, -1).detach().requires_grad_(True)
o = R.forward_decoder(p, latent, bx, S, cfg, y=tok)
l = R.masked_mse_loss(o, torch.from_numpy(gt).to(dt), torch.from_numpy(mask).to(dt))
l.backward()
def cmp(name, got, ref):
    got = got.double().cpu().flatten(); ref = ref.double().flatten()
    print("%-10s rel-max %.3e  rms-rel %.3e" % (name, ((got-ref).abs().max()/ref.abs().max()).item(), ((got-ref).norm()/ref.norm()).item()))
cmp("ytok", buf["ytok"], tok.detach().reshape(B*S, -1))
cmp("dy_tok", buf["dy_tok"], tok.grad.reshape(B*S, -1))
# layer 4 backward in the oracle from dy_tok
c4 = inter["c4"]; (R.instance_norm_relu(c4).mean((2,3)) * tok.grad.reshape(B*S,-1)).sum().backward()
cmp("c4", buf["c4"], c4.detach().permute(0,2,3,1))
cmp("dc4", buf["dc4"], c4.grad.permute(0,2,3,1))
p3 = inter["p3"]
F.conv2d(p3, p["decoder_proj4.0.weight"], None, padding=1).backward(c4.grad)
cmp("dp3", buf["dp3"], p3.grad.permute(0,2,3,1))
w4 = p["decoder_proj4.0.weight"].detach().requires_grad_(True)
F.conv2d(p3.detach(), w4, None, padding=1).backward(c4.grad)
cmp("dW4", eng.gview("decoder_proj4.0.weight"), w4.grad)
c3 = inter["c3"]; R.max_pool2(R.instance_norm_relu(c3)).backward(p3.grad)
cmp("dc3", buf["dc3"], c3.grad.permute(0,2,3,1))
# count relu-gate flips between engine (fp32) and oracle (f64)
st = buf["instats4"].double().cpu()            # [BS, C, 2]
c4e = buf["c4"].double().cpu()                 # [BS, 8, 8, C]
xh_e = (c4e - st[:, None, None, :, 0]) * st[:, None, None, :, 1]
c4o = c4.detach().permute(0, 2, 3, 1)
mu = c4o.mean((1, 2), keepdim=True); var = ((c4o - mu) ** 2).mean((1, 2), keepdim=True)
xh_o = (c4o - mu) / torch.sqrt(var + 1e-5)
flips = ((xh_e > 0) != (xh_o > 0))
print("relu gate flips in layer 4:", int(flips.sum()), "of", flips.numel(), " |xhat| at flips:", xh_o[flips].abs().tolist()[:5])
d = (buf["dc4"].double().cpu() - c4.grad.permute(0, 2, 3, 1))
print("dc4 rms-rel excluding flipped elements: %.3e" % (d[~flips].norm() / c4.grad.norm()).item())
