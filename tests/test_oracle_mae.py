"""CPU: pin the MAE-pretraining oracle (oracle/mae_ref.py) against golden vectors produced by the reference's
models_mae_noct.py (tools/oracle/make_golden_mae.py)."""
import json
import os

import numpy as np
import pytest

from oracle import mae_ref as M
from oracle import weights as W

G = os.path.join(os.path.dirname(__file__), "golden")
NAME = "mae_vit_base_patch16"


@pytest.mark.parametrize("tag,npl", [("plain", False), ("normpix", True)])
def test_mae_oracle_matches_reference(tag, npl):
    g = np.load(os.path.join(G, "mae_b2.npz"))
    meta = json.load(open(os.path.join(G, "mae_meta.json")))
    assert [(n, list(s)) for n, s, _ in W.schema_mae(NAME)] == [(a, b) for a, b in meta["schema"]]
    sd = W.make_state_dict_mae(NAME, seed=0)
    imgs, ids_shuffle, ids_restore, len_keep = W.make_mae_inputs(batch=2, seed=0, mask_ratio=0.5)
    assert len_keep == meta["len_keep"] == 288
    loss, pred, mask, grads = M.loss_and_grads(sd, imgs, ids_shuffle, ids_restore, len_keep, NAME, norm_pix_loss=npl)
    assert abs(loss.item() - float(g["loss_" + tag])) <= 1e-5 * float(g["loss_" + tag])
    assert np.array_equal(mask.numpy(), g["mask"])
    ph = g["pred_head_" + tag]
    assert np.abs(pred.numpy()[:, :8] - ph).max() <= 2e-5 * np.abs(ph).max()
    assert sorted(grads) == meta["grad_tensors_" + tag]
    for k, gr in grads.items():
        gn = float(g["%s/norm/%s" % (tag, k)])
        n = np.sqrt((gr.numpy().astype(np.float64) ** 2).sum())
        assert abs(n - gn) <= 5e-4 * gn + 1e-9, k
