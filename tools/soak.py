"""Soak test: many fused finetune / pretrain steps with the shot_num schedule of the CLI; checks finite losses, a decreasing trend
and a flat memory footprint (no allocation in steady state)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import models_mae_cross, models_mae_noct
from countr_amd.trainer import FinetuneStep, PretrainStep
from countr_amd.parallel import shared_shot_num
from countr_amd.synthetic import make_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
m = models_mae_cross.mae_vit_base_patch16(precision="bf16").to("cuda").train()
st = FinetuneStep(m, batch=8, lr=2e-5)
batches = [make_batch(8, shots=3, seed=i, device="cuda") for i in range(4)]
losses, mem = torch.zeros(steps, device="cuda"), []
for it in range(steps):
    S = shared_shot_num(it, seed=0)
    st.load(*batches[it % 4], S)
    losses[it].copy_(st.step(S)[0])
    if it in (100, steps - 1):
        torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated())
l = losses.cpu()
assert torch.isfinite(l).all(), "non-finite loss"
print("finetune: loss first 20 %.5f  last 20 %.5f   mem %d -> %d MB" % (l[:20].mean(), l[-20:].mean(), mem[0] >> 20, mem[1] >> 20))
assert l[-20:].mean() < l[:20].mean() and mem[0] == mem[1]
del st, m
p = models_mae_noct.mae_vit_base_patch16(precision="bf16").to("cuda").train()
ps = PretrainStep(p, batch=8, lr=1e-4)
imgs = [torch.rand(8, 3, 384, 384, device="cuda") for _ in range(4)]
losses, mem = torch.zeros(steps, device="cuda"), []
for it in range(steps):
    ps.load(imgs[it % 4])
    losses[it].copy_(ps.step()[0])
    if it in (100, steps - 1):
        torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated())
l = losses.cpu()
assert torch.isfinite(l).all(), "non-finite loss"
print("pretrain: loss first 20 %.5f  last 20 %.5f   mem %d -> %d MB" % (l[:20].mean(), l[-20:].mean(), mem[0] >> 20, mem[1] >> 20))
assert l[-20:].mean() < l[:20].mean() and mem[0] == mem[1]
print("soak ok")
