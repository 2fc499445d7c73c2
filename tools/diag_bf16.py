"""bf16 vs fp32 engine: error of the density map and counts per shot_num on the golden inputs (numbers quoted in tests/test_model_gpu.py)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import countr_ref as R, weights as W
sys.path.insert(0, 'tests')
from test_model_gpu import build, case_inputs, CASES, rel, G, MODEL
m, sd = build("bf16")
g = np.load(os.path.join(G, "forward.npz")); meta = json.load(open(os.path.join(G, "meta.json")))
for name in CASES:
    im, bx, s = case_inputs(name)
    with torch.no_grad():
        out = m(torch.from_numpy(im).cuda(), torch.from_numpy(bx).cuda(), s).cpu().numpy()
    rms = np.sqrt(((out.astype(np.float64) - g[name]) ** 2).mean()) / np.sqrt((g[name].astype(np.float64) ** 2).mean())
    cnt = out.reshape(out.shape[0], -1).sum(1) / 60
    print(name, "maxrel %.4f rms %.4f cnt" % (rel(out, g[name]), rms), cnt, meta["count_" + name])
imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=1)
m.train(); m.zero_grad()
out = m(torch.from_numpy(imgs).cuda(), torch.from_numpy(boxes).cuda(), 3)
loss = R.masked_mse_loss(out, torch.from_numpy(gt).cuda(), torch.from_numpy(mask).cuda()); loss.backward()
_, rloss, rg = R.loss_and_grads(sd, imgs, boxes, gt, mask, 3, MODEL)
print("loss", loss.item(), rloss.item())
for k, p in m.named_parameters():
    if p.grad is None or rg.get(k) is None: continue
    ref = rg[k].double(); got = p.grad.detach().cpu().double()
    cos = (got * ref).sum() / (got.norm() * ref.norm() + 1e-30)
    print("%-45s cos %.4f norm %.3e ref %.3e" % (k, cos.item(), got.norm().item(), ref.norm().item()))
