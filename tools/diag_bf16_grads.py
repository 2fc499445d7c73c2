"""bf16-mode gradient quality: cosine / norm ratio per trainable tensor vs the fp32 CPU oracle (B = 2 and B = 8, shot_num 3 and 0).
Backs the bars of tests/test_model_gpu.py::test_bf16_gradients_close_to_oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import countr_ref as R, weights as W
import models_mae_cross as mm
MODEL = "mae_vit_base_patch16"
sd = W.make_state_dict(MODEL, seed=0)
m = mm.__dict__[MODEL](precision="bf16"); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.to("cuda")
torch.set_num_threads(min(os.cpu_count(), 32))
for B, S, seed in ((2, 3, 1), (2, 0, 2), (8, 3, 3)):
    imgs, boxes, gt, mask = W.make_inputs(batch=B, shots=3, seed=seed)
    m.train(); m.zero_grad()
    out = m(torch.from_numpy(imgs).cuda(), torch.from_numpy(boxes).cuda(), S)
    loss = R.masked_mse_loss(out, torch.from_numpy(gt).cuda(), torch.from_numpy(mask).cuda()); loss.backward(); m.eval()
    _, rloss, rg = R.loss_and_grads(sd, imgs, boxes, gt, mask, S, MODEL)
    rows = []
    for k, p in m.named_parameters():
        if p.grad is None or rg.get(k) is None: continue
        ref = rg[k].double(); got = p.grad.detach().cpu().double()
        if ref.norm() < 1e-3: continue
        rows.append((k, ((got * ref).sum() / (got.norm() * ref.norm())).item(), (got.norm() / ref.norm()).item()))
    print("B=%d S=%d loss rel err %.2e" % (B, S, abs(loss.item() - rloss.item()) / rloss.item()))
    for grp in ("decoder_proj", "decoder_blocks", "decode_head", "decoder_norm", "decoder_embed", "shot_token"):
        sel = [r for r in rows if r[0].startswith(grp)]
        if sel:
            print("   %-15s n=%2d  cos min %.4f (%s)  norm ratio %.3f..%.3f" % (grp, len(sel), min(r[1] for r in sel), min(sel, key=lambda r: r[1])[0],
                                                                                 min(r[2] for r in sel), max(r[2] for r in sel)))
