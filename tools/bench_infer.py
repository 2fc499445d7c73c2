"""Forward-only throughput of the HIP engine (bf16): eager launch list vs hipGraph replay, several batch sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import models_mae_cross
torch.manual_seed(0)
m = models_mae_cross.mae_vit_base_patch16(precision="bf16").to("cuda").eval()
eng = m._engine()
for B, S in ((1, 3), (8, 3), (32, 0), (32, 3)):
    imgs = torch.rand(B, 3, 384, 384, device="cuda"); boxes = torch.rand(B, 3, 3, 64, 64, device="cuda")
    p = eng.plan(B, S, False)
    eng._load_inputs(p, imgs, boxes, S)
    for _ in range(3): eng.run(p.fwd_par)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.run(p.fwd_par)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            eng.run(p.fwd_par)
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 20
    print("B=%2d S=%d  eager %.2f ms (%.0f img/s)   graph %.2f ms (%.0f img/s)   %d launches" % (B, S, eager * 1e3, B / eager, gr * 1e3, B / gr, len(p.fwd)), flush=True)
