"""Is sample i of a batch bit-identical to the same sample run alone?  python tools/diag_batch_indep.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import weights as W
import models_mae_cross as mm
m = mm.__dict__["mae_vit_base_patch16"](precision="bf16")
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict("mae_vit_base_patch16", seed=0).items()}, strict=True)
m.to("cuda").eval()
for B in [int(a) for a in sys.argv[1:]] or [8, 26]:
    imgs, boxes, _, _ = (torch.from_numpy(a).cuda() for a in W.make_inputs(batch=B, shots=3, seed=6))
    with torch.no_grad():
        full = m(imgs, boxes, 3).clone()
        alone = m(imgs[:1], boxes[:1], 3)[0]
        eng = m._engine()
        pf, pa = eng.plan(B, 3, False), eng.plan(1, 3, False)
        m(imgs, boxes, 3); bf = {k: v.clone() for k, v in pf.buf.items() if torch.is_tensor(v)}
        m(imgs[:1], boxes[:1], 3); ba = {k: v.clone() for k, v in pa.buf.items() if torch.is_tensor(v)}
    print("B=%d: out equal %s, max |diff| %.3e" % (B, torch.equal(alone, full[0]), (alone - full[0]).abs().max().item()))
    for k in ba:
        if k in bf and bf[k].dim() >= 1 and ba[k].numel() > 0 and bf[k].numel() == B * ba[k].numel():
            a, f = ba[k].reshape(-1).float(), bf[k].reshape(B, -1)[0].float()
            if not torch.equal(a, f):
                print("   first differing buffer: %-12s max |diff| %.3e (max |v| %.3e)" % (k, (a - f).abs().max().item(), a.abs().max().item()))
                break
