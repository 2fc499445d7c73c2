"""CPU: pin the oracle (oracle/countr_ref.py) against golden vectors produced by the reference itself
(tools/oracle/make_golden.py).  fp32 noise floor of the reference is ~1e-6 rel."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import countr_ref as R
from oracle import weights as W

G = os.path.join(os.path.dirname(__file__), "golden")
MODEL = "mae_vit_base_patch16"


@pytest.fixture(scope="module")
def sd():
    return W.make_state_dict(MODEL, seed=0)


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(G, "meta.json")))


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_schema_matches_reference(meta):
    assert [(n, list(s)) for n, s, _ in W.schema(MODEL)] == [tuple(x) if False else (x[0], x[1]) for x in meta["schema"]]
    assert len(meta["schema"]) == 225
    assert meta["n_params"] == 99690625


def test_pos_embed_rows():
    g = np.load(os.path.join(G, "pos_embed_rows.npz"))
    rows = g["rows"]
    assert np.abs(W.sincos_2d(768, 24)[rows] - g["pe768"]).max() < 1e-12
    assert np.abs(W.sincos_2d(512, 24)[rows] - g["pe512"]).max() < 1e-12


def test_forward_cases_match_reference(sd, meta):
    g = np.load(os.path.join(G, "forward.npz"))
    imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=0)
    cases = {
        "b2_s3": (imgs, boxes, 3),
        "b1_s0": (imgs[:1], boxes[:1], 0),
        "b1_s1": (imgs[1:2], boxes[1:2], 1),
        "b1_s2": (imgs[:1], boxes[:1], 2),
        "b1_zero_empty": (imgs[1:2], np.zeros((1, 0), np.float32), 0),
    }
    p = R.Params(sd)
    for name, (im, bx, s) in cases.items():
        out = R.forward(p, im, bx, s, MODEL).numpy()
        assert out.shape == g[name].shape
        assert rel(out, g[name]) < 2e-5, name
        cnt = out.reshape(out.shape[0], -1).sum(1) / 60
        assert np.abs(cnt - np.array(meta["count_" + name])).max() < 0.02, name


def test_probes_match_reference(sd):
    g = np.load(os.path.join(G, "probes_b2_s3.npz"))
    imgs, boxes, _, _ = W.make_inputs(batch=2, shots=3, seed=0)
    probes = {}
    R.forward(R.Params(sd), imgs, boxes, 3, MODEL, probes=probes)
    for mine, ref in (("enc_block0", "enc_block0"), ("enc_block11", "enc_block11"), ("latent", "latent"),
                      ("dec_block0", "dec_block0"), ("dec_block1", "dec_block1"), ("dec_norm", "dec_norm")):
        v = probes[mine].detach().numpy()
        assert list(v.shape) == list(g[ref + "_shape"])
        assert rel(v.reshape(v.shape[0], -1)[:, :256], g[ref + "_head"]) < 2e-5, mine
        l2 = np.sqrt((v.astype(np.float64) ** 2).sum())
        assert abs(l2 - g[ref + "_l2"]) / g[ref + "_l2"] < 1e-5, mine


@pytest.mark.parametrize("tag,shots", [("s3", 3), ("s0", 0)])
def test_gradients_match_reference(sd, meta, tag, shots):
    g = np.load(os.path.join(G, "grads_b2.npz"))
    imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=0)
    out, loss, grads = R.loss_and_grads(sd, imgs, boxes, gt, mask, shots, MODEL)
    assert abs(loss.item() - float(g["loss_" + tag])) / float(g["loss_" + tag]) < 1e-5
    have = sorted(k for k, v in grads.items() if v is not None)
    assert have == meta["grad_tensors_" + tag]
    for k in have:
        gn = float(g["%s/norm/%s" % (tag, k)])
        mine = grads[k].numpy()
        n = np.sqrt((mine.astype(np.float64) ** 2).sum())
        assert abs(n - gn) <= 2e-4 * gn + 2e-8, k  # conv biases before InstanceNorm and wk.bias have exactly-zero gradients (noise only)
        fk = "%s/full/%s" % (tag, k)
        if fk in g:
            assert np.abs(mine - g[fk]).max() <= 2e-4 * np.abs(g[fk]).max() + 2e-8, k
        else:
            hk = g["%s/head/%s" % (tag, k)]
            rms = gn / np.sqrt(mine.size)  # fp32-vs-fp32 backward: elementwise noise scales with the tensor's rms
            assert np.abs(mine.reshape(-1)[:512] - hk).max() <= 2e-4 * np.abs(hk).max() + 5e-3 * rms + 2e-8, k


def test_lr_schedule(meta):
    for e, lr in meta["lr_table"]:
        assert abs(R.adjust_learning_rate(e, 1e-5, 0.0, 10, 1000) - lr) < 1e-18


def test_stitch_matches_reference(sd):
    g = np.load(os.path.join(G, "stitch.npz"))
    _, boxes, _, _ = W.make_inputs(batch=2, shots=3, seed=0)
    p = R.Params(sd)
    rs = np.random.RandomState(77)
    for width in (672, 512, 384):
        wide = rs.uniform(0, 1, size=(1, 3, 384, width)).astype(np.float32)
        starts = []

        def fn(start):
            starts.append(start)
            return R.forward(p, wide[:, :, :, start:start + 384], boxes[:1], 3, MODEL)[0]
        dm = R.stitch_windows(fn, width)
        assert starts == list(g["starts_%d" % width])
        assert abs(dm.sum().item() / 60 - float(g["count_%d" % width])) < 0.02
        assert rel(dm.sum(0).numpy(), g["colsum_%d" % width]) < 5e-5


def test_adamw_matches_torch():
    torch.manual_seed(0)
    p = torch.randn(1000, dtype=torch.float64)
    w = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([w], lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step in range(1, 4):
        gsteps = torch.randn(1000, dtype=torch.float64)
        w.grad = gsteps.clone()
        opt.step()
        p, m, v = R.adamw_step(p, gsteps, m, v, step, 1e-3)
        assert (p - w.detach()).abs().max() < 1e-12


def test_oracle_patch14_head80_matches_reference_golden():
    """The odd shapes of mae_vit_huge_patch14 (models_mae_cross.py:235-239) -- patch 14 on 384 pixels: timm's PatchEmbed conv leaves
    27 x 27 = 729 tokens and the head a 432 x 432 map; head_dim 80 -- pinned through the reference's own SupervisedMAE class at an
    affordable width (tools/oracle/make_golden_patch14.py -> tests/golden/patch14.npz, patch14_meta.json)."""
    import json
    import os
    import numpy as np
    import torch
    from oracle import countr_ref as R, weights as W
    G = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(G, "patch14.npz"))
    meta = json.load(open(os.path.join(G, "patch14_meta.json")))
    assert meta["huge_output_shape"] == [1, 432, 432] and meta["huge_pos_embed"] == [1, 729, 1280]
    sd = W.make_state_dict("tiny_patch14", seed=5)
    imgs, boxes, _gt, _mask = W.make_inputs(batch=2, shots=3, seed=7)
    torch.set_num_threads(min(os.cpu_count(), 8))
    out = R.forward(sd, imgs, boxes, 3, "tiny_patch14").numpy()
    assert out.shape == (2, 432, 432)
    assert np.abs(out - g["b2_s3"]).max() <= 2e-5 * np.abs(g["b2_s3"]).max()
    out0 = R.forward(sd, imgs[:1], boxes[:1], 0, "tiny_patch14").numpy()
    assert np.abs(out0.sum(1) - g["b1_s0_colsum"]).max() <= 2e-5 * np.abs(g["b1_s0_colsum"]).max()
    assert abs(out0.sum() / 60 - meta["count_b1_s0"][0]) < 1e-2
