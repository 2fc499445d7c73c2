"""Reproducer kept for the record: a hipMemsetAsync node + float atomics inside a captured region gave wrong loss sums under graph replay (ROCm 7.2); the fix is the deterministic two-stage reduction (see countr_amd/trainer.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import models_mae_cross
from countr_amd.trainer import FinetuneStep
from countr_amd.synthetic import make_batch
from countr_amd.parallel import shared_shot_num
variant = sys.argv[1]
torch.manual_seed(0)
m = models_mae_cross.mae_vit_base_patch16(precision="bf16").to("cuda").train()
step = FinetuneStep(m, batch=8, lr=3e-6, use_graph=(variant != "nograph"))
keep = []
for it in range(50):
    S = shared_shot_num(it, seed=0)
    b = make_batch(8, shots=3, seed=it, device="cuda")
    if variant == "sync_after_make": torch.cuda.synchronize()
    if variant == "keep": keep.append(b)
    step.load(*b, S)
    if variant == "sync_after_load": torch.cuda.synchronize()
    sums = step.step(S)
s = sums.float().cpu().numpy()
print(variant, "final loss %.5f" % s[0], flush=True)
if not np.isfinite(s[0]) or abs(s[0]) > 10:
    eng = m._eng
    def st(name, t):
        t = t.float()
        print("  %-22s finite=%s absmax=%.3e" % (name, bool(torch.isfinite(t).all()), t.abs().max().item()))
    st("gt", step.gt); st("mask", step.mask); st("P(all)", eng.P); st("P(train)", eng.P[eng.layout.train_start:]); st("G", eng.G); st("M", eng.M); st("V", eng.V); st("Wt", eng.Wt)
    for key, p in eng.plans.items():
        for k in ("img", "out", "dout", "latent", "dn"):
            if k in p.buf: st("%s %s" % (key, k), p.buf[k])
    st("hyper", eng.hyper)
    print("  step_count", eng.step_count, "hyper", eng.hyper.cpu().numpy())
