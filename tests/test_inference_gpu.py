"""GPU: sliding-window stitching path (countr_amd/inference.py) against the reference's own stitch loop goldens
(tests/golden/stitch.npz, produced by tools/oracle/make_golden.py from FSC_test_cross(few-shot).py:322-351)."""
import os

import numpy as np
import pytest
import torch

from oracle import weights as W

G = os.path.join(os.path.dirname(__file__), "golden")


def test_window_schedule_cpu():
    from countr_amd.inference import window_starts
    g = np.load(os.path.join(G, "stitch.npz"))
    for width in (672, 512, 384):
        assert window_starts(width) == list(g["starts_%d" % width])
    assert window_starts(383) == []            # narrower than a window: the reference loop never runs
    assert window_starts(672) == [0, 128, 256, 288]


def test_blend_weights_sum_to_one_cpu():
    from countr_amd.inference import blend_windows, window_starts
    for width in (384, 512, 640, 672, 944):
        st = window_starts(width)
        dm = blend_windows(torch.ones(len(st), 384, 384), st, width)
        assert torch.allclose(dm, torch.ones(384, width))   # sequential 1/2-blend is a convex combination


@pytest.mark.gpu
def test_stitched_density_matches_reference_golden():
    import models_mae_cross
    from countr_amd import inference
    g = np.load(os.path.join(G, "stitch.npz"))
    m = models_mae_cross.mae_vit_base_patch16(precision="fp32")
    sd = W.make_state_dict("mae_vit_base_patch16", seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.to("cuda").eval()
    _, boxes, _, _ = W.make_inputs(batch=2, shots=3, seed=0)
    bx = torch.from_numpy(boxes[:1]).cuda()
    rs = np.random.RandomState(77)
    for width in (672, 512, 384):
        wide = torch.from_numpy(rs.uniform(0, 1, size=(1, 3, 384, width)).astype(np.float32)).cuda()
        dm = inference.density_map(m, wide, bx, 3)
        assert abs(dm.sum().item() / 60 - float(g["count_%d" % width])) < 0.5
        ref = g["colsum_%d" % width]
        assert np.abs(dm.sum(0).cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
    pred, _ = inference.count_image(m, wide, bx, 3, pos=[(0, 0, 50, 50), (10, 10, 80, 80), (5, 5, 60, 60)])
    assert np.isfinite(pred)
    z = inference.density_map(m, torch.rand(1, 3, 384, 300).cuda(), bx, 3)
    assert z.shape == (384, 300) and float(z.abs().sum()) == 0.0
