"""Per-buffer rms error of the bf16 engine against the fp32 engine along the forward pass (shows where the final 256->1 conv amplifies)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch
from test_model_gpu import build, case_inputs
mb, _ = build("bf16"); mf, _ = build("fp32")
for name in ("b1_s0", "b1_s2"):
    im, bx, s = case_inputs(name)
    with torch.no_grad():
        ob = mb(torch.from_numpy(im).cuda(), torch.from_numpy(bx).cuda(), s)
        of = mf(torch.from_numpy(im).cuda(), torch.from_numpy(bx).cuda(), s)
    pb = mb._eng.plans[(1, s, False)].buf; pf = mf._eng.plans[(1, s, False)].buf
    print("==", name)
    for k in pb:
        if k in ("img", "boxes", "patches"): continue
        a = pb[k].double().flatten(); b = pf[k].double().flatten()
        if a.numel() != b.numel(): continue
        d = a - b
        print("%-28s rms_rel %.4e  mean_err/rms %.3e  ref_rms %.3e" % (k, (d.norm() / (b.norm() + 1e-30)).item(), (d.mean() / (b.pow(2).mean().sqrt() + 1e-30)).item(), b.pow(2).mean().sqrt().item()))
