#!/usr/bin/env python3
"""Golden vectors for the odd shapes of mae_vit_huge_patch14 (models_mae_cross.py:235-239) from the REAL reference (build container only;
same shims as make_golden.py, whose fixtures this script does not touch).

1. The factory itself: state_dict schema of mae_vit_huge_patch14 == oracle/weights.schema, pos-embed tables 729 x 1280 / 729 x 512, and the
   shape of its output on a 384 x 384 image: 432 x 432 (27 tokens per side x 16) -- recorded in patch14_meta.json.
2. The reference's own SupervisedMAE class at an affordable width with the SAME odd shapes (patch 14, head_dim 80: oracle/weights.CONFIGS
   ["tiny_patch14"]) on the deterministic weights / inputs of oracle/weights.py: density maps for shot_num 3 and 0 -> tests/golden/patch14.npz."""
import json
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    import make_golden as mg
    from oracle import weights as W
    mm, _ = mg.import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    meta = {}
    huge = mm.mae_vit_huge_patch14(norm_pix_loss=False)
    assert [(k, list(v.shape)) for k, v in huge.state_dict().items()] == [(n, list(s)) for n, s, _ in W.schema("mae_vit_huge_patch14")]
    meta["huge_n_params"] = int(sum(p.numel() for p in huge.parameters()))
    meta["huge_pos_embed"] = list(huge.pos_embed.shape)
    huge.eval()
    with torch.no_grad():
        y = huge(torch.rand(1, 3, 384, 384), torch.rand(1, 3, 3, 64, 64), 3)
    meta["huge_output_shape"] = list(y.shape)
    del huge
    name = "tiny_patch14"
    p, D, depth, H, Dd, ddepth, Hd = W.CONFIGS[name]
    model = mm.SupervisedMAE(patch_size=p, embed_dim=D, depth=depth, num_heads=H, decoder_embed_dim=Dd, decoder_depth=ddepth,
                             decoder_num_heads=Hd, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    assert [(k, list(v.shape)) for k, v in model.state_dict().items()] == [(n, list(s)) for n, s, _ in W.schema(name)]
    sd = W.make_state_dict(name, seed=5)
    meta["weights_sha256_seed5"] = mg.sha(sd)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    imgs, boxes, _gt, _mask = W.make_inputs(batch=2, shots=3, seed=7)
    out = {}
    with torch.no_grad():
        for tag, (im, bx, s) in {"b2_s3": (imgs, boxes, 3), "b1_s0": (imgs[:1], boxes[:1], 0)}.items():
            y = model(torch.from_numpy(im), torch.from_numpy(bx), s).numpy().astype(np.float32)
            meta["count_" + tag] = [float(v) for v in y.reshape(y.shape[0], -1).sum(1) / 60]
            meta["shape_" + tag] = list(y.shape)
            if tag == "b2_s3":
                out[tag] = y
            else:
                out[tag + "_colsum"] = y.sum(1)
                out[tag + "_rowsum"] = y.sum(2)
    np.savez_compressed(os.path.join(OUT, "patch14.npz"), **out)
    with open(os.path.join(OUT, "patch14_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(meta)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(OUT, "patch14.npz")))


if __name__ == "__main__":
    main()
