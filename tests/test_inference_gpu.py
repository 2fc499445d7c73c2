"""GPU: sliding-window stitching path (countr_amd/inference.py) against the reference's own stitch loop goldens
(tests/golden/stitch.npz, produced by tools/oracle/make_golden.py from FSC_test_cross(few-shot).py:322-351)."""
import os

import numpy as np
import pytest
import torch

from oracle import countr_ref as R
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


@pytest.fixture(scope="module")
def fp32_model():
    import models_mae_cross
    m = models_mae_cross.mae_vit_base_patch16(precision="fp32")
    sd = W.make_state_dict("mae_vit_base_patch16", seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to("cuda").eval()


@pytest.mark.gpu
def test_zero_shot_stitch_matches_demo_zero_golden(fp32_model):
    """demo_zero.py:41-74 run unchanged by tools/oracle/make_golden_infer.py on two 384 x 672 frames (1920x1080 resized: config 5)
    with boxes = torch.Tensor([]) and shot_num = 0; here both frames go through ONE forward of 8 windows."""
    from countr_amd import inference
    g = np.load(os.path.join(G, "infer.npz"))
    imgs = [torch.from_numpy(W.make_wide_inputs(100 + k, 672, 0)[0]).cuda() for k in range(2)]
    empty = torch.zeros(1, 0, device="cuda")
    dms = inference.density_maps(fp32_model, imgs, [empty, empty], 0)
    for k, dm in enumerate(dms):
        assert abs(dm.sum().item() / 60 - float(g["zero%d_count" % k])) < 0.5
        for axis, key in ((0, "colsum"), (1, "rowsum")):
            ref = g["zero%d_%s" % (k, key)]
            assert np.abs(dm.sum(axis).cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max(), (k, key)


@pytest.mark.gpu
def test_count_image_paths_match_reference_golden(fp32_model):
    """FSC_test_cross(few-shot).py:261-359 run unchanged by the generator: (a) plain windows + test-time normalisation, (b) two
    exemplars below 10 px -> the 3x3 crop-and-upscale path (nine crops, counts summed, normalised with the last crop's map)."""
    from countr_amd import inference
    g = np.load(os.path.join(G, "infer.npz"))
    img, bx, pos = W.make_wide_inputs(200, 512, 3)
    pred, dm = inference.count_image(fp32_model, torch.from_numpy(img).cuda(), torch.from_numpy(bx).cuda(), 3, pos=pos)
    assert int(g["plain_s_cnt"]) == 0 and [tuple(r) for r in g["plain_pos"]] == pos
    assert abs(pred - float(g["plain_count"])) <= 2e-3 * float(g["plain_count"])
    assert np.abs(dm.sum(0).cpu().numpy() - g["plain_colsum"]).max() <= 1e-3 * np.abs(g["plain_colsum"]).max()
    img, bx, _ = W.make_wide_inputs(201, 400, 3)
    pos = [tuple(int(v) for v in r) for r in g["split_pos"]]
    assert int(g["split_s_cnt"]) == 2
    pred, dm = inference.count_image(fp32_model, torch.from_numpy(img).cuda(), torch.from_numpy(bx).cuda(), 3, pos=pos)
    assert abs(pred - float(g["split_count"])) <= 2e-3 * float(g["split_count"])
    assert np.abs(dm.sum(0).cpu().numpy() - g["split_colsum"]).max() <= 1e-3 * np.abs(g["split_colsum"]).max()
    # without normalisation the prediction is the plain sum of the nine crop counts
    pred_raw, _ = inference.count_image(fp32_model, torch.from_numpy(img).cuda(), torch.from_numpy(bx).cuda(), 3, pos=pos, normalization=False)
    assert abs(pred_raw - float(g["split_crop_counts"].sum())) <= 1e-3 * float(g["split_crop_counts"].sum())


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config5_batch_of_32_windows(precision, fp32_model):
    """BASELINE config 5: zero-shot, 8 frames of 1920x1080 (-> 384 x 672, 4 windows each) = ONE forward of 32 windows.  The batched
    result equals the per-image path window for window, and mixed widths / shot counts keep their input order through
    count_images."""
    import models_mae_cross
    from countr_amd import inference
    if precision == "fp32":
        m = fp32_model
    else:
        m = models_mae_cross.mae_vit_base_patch16(precision="bf16")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict("mae_vit_base_patch16", seed=0).items()})
        m.to("cuda").eval()
    imgs = [torch.from_numpy(W.make_wide_inputs(300 + k, 672, 0)[0]).cuda() for k in range(8)]
    empty = torch.zeros(1, 0, device="cuda")
    calls, seen = [], []

    def spy(a, b, c):
        calls.append(a.shape[0])
        out = orig(a, b, c)
        if a.shape[0] == 32:
            seen.append((a.detach().clone(), out.detach().clone()))
        return out
    orig, native = m.forward, inference._native_maps
    m.forward = spy
    inference._native_maps = lambda *a, **k: None        # the torch path goes through model.forward (what the spy records) ...
    try:
        dms_torch = inference.density_maps(m, imgs, [empty] * 8, 0)
        assert calls == [32]
        single = [inference.density_map(m, im, empty, 0) for im in imgs]
    finally:
        m.forward = orig
        inference._native_maps = native
    # ... and the default path cuts the windows straight into the engine's input batch and stitches with countr_window_gather /
    # countr_window_blend: the same maps bit for bit, the per-image sums of the blend kernel = the maps' sums
    dms, sums = inference.density_maps(m, imgs, [empty] * 8, 0, return_sums=True)
    assert inference._native_maps(m, imgs, [empty] * 8, 0, 32, False) is not None
    for a, b, s_ in zip(dms, dms_torch, sums):
        assert torch.equal(a, b)
        assert abs(s_.item() - a.double().sum().item()) <= 1e-5 * a.abs().double().sum().item() + 1e-6
    # the batch-of-32 forward itself against the ORACLE (not only against the engine's own per-image path): windows 0, 5 (second
    # window of frame 1), 18 and 31 of the batch, same bars as the single-image tests (fp32: 1e-3 / +-0.5 counts; bf16, shot_num 0: 6e-2 / 6 %)
    (win, wout), = seen
    sd = W.make_state_dict("mae_vit_base_patch16", seed=0)
    torch.set_num_threads(min(os.cpu_count(), 32))
    for j in (0, 5, 18, 31):
        ref = R.forward(sd, win[j:j + 1].cpu().numpy(), np.zeros((1, 0), np.float32), 0).numpy()[0]
        got = wout[j].cpu().numpy()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        cnt, rc = got.sum() / 60, ref.sum() / 60
        if precision == "fp32":
            assert err < 1e-3 and abs(cnt - rc) < 0.5, (j, err, cnt, rc)
        else:
            assert err < 6e-2 and abs(cnt - rc) <= 6e-2 * abs(rc), (j, err, cnt, rc)
    for a, b in zip(dms, single):
        if precision == "fp32":      # same kernels, other batch size (tile / split-K choices may differ): rounding-level agreement
            assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
        else:
            assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
        assert torch.isfinite(a).all() and a.shape == (384, 672)
    # count_images: mixed widths and shot counts, results in input order
    items = []
    for k, (w, S) in enumerate(((672, 0), (512, 3), (384, 0), (640, 3), (300, 3))):
        img, bx, pos = W.make_wide_inputs(400 + k, w, max(S, 1))
        items.append((torch.from_numpy(img).cuda(), torch.from_numpy(bx).cuda() if S else empty, pos if S else None))
    res = inference.count_images(m, items)
    for (samples, bx, pos), (pred, dm) in zip(items, res):
        S = bx.shape[1] if bx.nelement() > 0 else 0
        p1, d1 = inference.count_image(m, samples, bx, S, pos=pos)
        assert dm.shape == d1.shape and abs(pred - p1) <= 1e-3 * max(abs(p1), 1.0)
    assert float(res[4][1].abs().sum()) == 0.0      # 300 px wide: no window, all-zero map


@pytest.mark.gpu
def test_panorama_with_more_than_16_windows(fp32_model):
    """A 2432-px-wide image has 17 window positions -- more than countr_window_blend takes (MAX_STARTS 16 in csrc/window.hip); the
    reference loop has no such limit (FSC_test_cross(few-shot).py:322-351).  The call must take the torch stitch path and give the
    sequential blend of the 17 single-window forwards (round 4 ran the forward and then raised CountrError)."""
    from countr_amd import inference
    m = fp32_model
    w = 384 + 128 * 16
    st = inference.window_starts(w)
    assert len(st) == 17 > inference.MAX_BLEND_WINDOWS
    img = torch.from_numpy(np.random.RandomState(5).uniform(0, 1, size=(1, 3, 384, w)).astype(np.float32)).cuda()
    empty = torch.zeros(1, 0, device="cuda")
    assert inference._native_maps(m, [img], [empty], 0, 32, False) is None
    (dm,), (s_,) = inference.density_maps(m, [img], [empty], 0, max_batch=32, return_sums=True)
    with torch.no_grad():
        wins = torch.stack([m(img[:, :, :, a:a + 384], empty, 0)[0].clone() for a in st])
    ref = inference.blend_windows(wins, st, w)
    assert dm.shape == (384, w)
    assert (dm - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert abs(float(s_) - ref.double().sum().item()) <= 1e-4 * ref.double().abs().sum().item()
    # 16 windows (width 2304) still go through the window kernels
    img16 = img[:, :, :, :2304].contiguous()
    assert len(inference.window_starts(2304)) == 16 and inference._native_maps(m, [img16], [empty], 0, 32, False) is not None


@pytest.mark.gpu
@pytest.mark.parametrize("precision,S", [("bf16", 0), ("fp16", 3), ("fp32", 0)])
def test_stream_with_pipelined_encoder_is_bit_identical(precision, S):
    """inference.density_maps_stream: while group k's decoder and density head run, group k + 1's windows are already cut and their
    frozen-encoder forward runs on a lane of its own; group k + 1 starts at decoder_embed (every forward is independent:
    models_mae_cross.py:201-207 under no_grad).  Same launches, same data: maps and per-image sums BIT-identical to one density_maps
    call per group -- over groups of different content, a group with another forward batch size in the middle (no look-ahead across
    it), a group that does not fit the window kernels (17 window positions: torch path), zero-shot and with exemplars.  fp32 has no
    pipelined form (its unfused attention shares scratch): the stream must simply equal the plain calls.  The engine's forward calls
    are recorded, so the test cannot pass by never pipelining."""
    import models_mae_cross
    from countr_amd import inference
    m = models_mae_cross.mae_vit_base_patch16(precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict("mae_vit_base_patch16", seed=0).items()})
    m.to("cuda").eval()
    empty = torch.zeros(1, 0, device="cuda")

    def group(seed, n, width):
        imgs, bxs = [], []
        for k in range(n):
            img, bx, _pos = W.make_wide_inputs(seed + k, width, max(S, 1))
            imgs.append(torch.from_numpy(img).cuda())
            bxs.append(torch.from_numpy(bx).cuda() if S else empty)
        return imgs, bxs
    pano = torch.from_numpy(np.random.RandomState(9).uniform(0, 1, size=(1, 3, 384, 384 + 128 * 16)).astype(np.float32)).cuda()
    _i, pbx, _p = W.make_wide_inputs(1, 672, max(S, 1))
    groups = [group(500, 4, 672), group(510, 4, 672), group(520, 4, 672), group(530, 2, 672), group(540, 4, 672), group(550, 4, 672),
              ([pano], [torch.from_numpy(pbx).cuda() if S else empty]), group(560, 4, 672), group(570, 4, 672)]
    ref = [inference.density_maps(m, g[0], g[1], S, max_batch=32, return_sums=True) for g in groups]
    eng = m._engine()
    calls = []
    orig_p, orig = eng.forward_loaded_pipelined, eng.forward_loaded
    eng.forward_loaded_pipelined = lambda B, s_, have, ahead: (calls.append((B, have, ahead)), orig_p(B, s_, have, ahead))[1]
    eng.forward_loaded = lambda B, s_: (calls.append((B, None, None)), orig(B, s_))[1]
    try:
        got = list(inference.density_maps_stream(m, iter(groups), S, max_batch=32, return_sums=True))
    finally:
        eng.forward_loaded_pipelined, eng.forward_loaded = orig_p, orig
    assert len(got) == len(ref)
    for (dms, sums), (rd, rs) in zip(got, ref):
        assert len(dms) == len(rd)
        for a, b, x, y in zip(dms, rd, sums, rs):
            assert torch.equal(a, b) and torch.equal(x, y)
    if precision == "fp32":
        assert all(h is None for _B, h, _a in calls)
    else:
        # 16 windows | 16 | 16 (no look-ahead: the next group is 8 windows) | 8 on its own | 16 | 16 (next: the panorama, torch path) |
        # the panorama's 17 single... forwards go through model.forward, not these two entry points | 16 | 16
        assert calls == [(16, False, True), (16, True, True), (16, True, False), (8, None, None), (16, False, True), (16, True, False),
                         (16, False, True), (16, True, False)], calls
