"""Is the host ahead of the GPU in the finetune loop?  Per-iteration host time of load() + step() without any synchronisation,
against the GPU time per step: if the host side takes as long as the GPU step, some call in the loop blocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import models_mae_cross
from countr_amd.trainer import FinetuneStep
from countr_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
model = models_mae_cross.mae_vit_base_patch16(precision="bf16").to(dev).train()
step = FinetuneStep(model, batch=8, lr=1e-5, weight_decay=0.05, use_graph=True)
imgs, boxes, gt, _ = make_batch(8, shots=3, seed=0, device=dev)
mask = (torch.rand(384, 384, device=dev) < 0.8).float()
if "host" in sys.argv[1:]:     # the DataLoader's hand-over: pinned host tensors, staged over PCIe by load()
    imgs, boxes, gt, mask = (t.cpu().pin_memory() for t in (imgs, boxes, gt, mask))
for _ in range(4):
    step.load(imgs, boxes, gt, mask, 3); step.step(3)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter(); ts = []; parts = []
for _ in range(N):
    a = time.perf_counter()
    step.load(imgs, boxes, gt, mask, 3)
    b = time.perf_counter()
    step.step(3)
    c = time.perf_counter()
    ts.append(c - a); parts.append((b - a, c - b))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host loop: %.3f ms per iteration (load %.3f, step %.3f); GPU drained %.3f ms after the loop; total %.3f ms per step"
      % (1e3 * (t1 - t0) / N, 1e3 * sum(p[0] for p in parts) / N, 1e3 * sum(p[1] for p in parts) / N, 1e3 * (t2 - t1), 1e3 * (t2 - t0) / N))
print("per-iteration host ms:", " ".join("%.2f" % (1e3 * t) for t in ts))
