#!/usr/bin/env python3
"""Golden vectors for the MAE pretraining model from the REAL reference (models_mae_noct.py), build container only.
Same shims as make_golden.py.  The reference draws its masking permutation with torch.rand inside random_masking; to pin
it we patch torch.rand for that call so that argsort(noise) equals the permutation of oracle/weights.make_mae_inputs."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "oracle"))
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    from make_golden import import_reference
    import_reference()
    import models_mae_noct
    from oracle import weights as W
    torch.set_num_threads(8)
    name = "mae_vit_base_patch16"
    model = models_mae_noct.__dict__[name](norm_pix_loss=False)
    ref_keys = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    assert ref_keys == [(n, list(s)) for n, s, _ in W.schema_mae(name)], "MAE state_dict schema mismatch"
    sd = W.make_state_dict_mae(name, seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.train()
    imgs, ids_shuffle, ids_restore, len_keep = W.make_mae_inputs(batch=2, seed=0, mask_ratio=0.5)
    out = {}
    meta = {"schema": ref_keys, "len_keep": len_keep}
    real_rand = torch.rand
    for tag, npl in (("plain", False), ("normpix", True)):
        model.norm_pix_loss = npl
        model.zero_grad()

        def fake_rand(*a, **k):   # noise whose argsort is ids_shuffle: noise[b, ids_shuffle[b, j]] = j / L
            L = ids_shuffle.shape[1]
            noise = torch.empty(ids_shuffle.shape, dtype=torch.float32)
            noise.scatter_(1, torch.from_numpy(ids_shuffle), torch.arange(L, dtype=torch.float32).repeat(ids_shuffle.shape[0], 1) / L)
            return noise
        torch.rand = fake_rand
        try:
            loss, pred, mask = model(torch.from_numpy(imgs), mask_ratio=0.5)
        finally:
            torch.rand = real_rand
        loss.backward()
        out["loss_" + tag] = np.float64(loss.item())
        out["pred_head_" + tag] = pred.detach().numpy()[:, :8, :].copy()
        out["pred_l2_" + tag] = np.float64(np.sqrt((pred.detach().numpy().astype(np.float64) ** 2).sum()))
        if tag == "plain":
            out["mask"] = mask.numpy().copy()
            assert np.array_equal(np.argsort(ids_shuffle, 1), ids_restore)
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.numpy()
            out["%s/norm/%s" % (tag, k)] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            out["%s/head/%s" % (tag, k)] = g.reshape(-1)[:256].copy()
        meta["grad_tensors_" + tag] = sorted(k for k, p in model.named_parameters() if p.grad is not None)
    np.savez_compressed(os.path.join(OUT, "mae_b2.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "mae_meta.json"), "w"), indent=1)
    print("loss", out["loss_plain"], out["loss_normpix"], os.path.getsize(os.path.join(OUT, "mae_b2.npz")))


if __name__ == "__main__":
    main()
