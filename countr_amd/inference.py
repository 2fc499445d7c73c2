"""Sliding-window inference around SupervisedMAE.forward.

Reference behaviour: FSC_test_cross(few-shot).py:264-359 and demo_zero.py:41-74 -- images are resized to height 384,
the width is covered by 384-px windows with stride 128 (last window snapped to w-384), overlapping columns are blended
sequentially (previously covered columns: 1/2 old + 1/2 new), count = sum/60, optional 3x3 crop-and-upscale for tiny
exemplars and test-time normalisation by the mean exemplar-box count.  Images narrower than 384 give an all-zero map
(the reference's loop never runs).  Every window is an independent forward, so all windows (of all crops) go through
the engine as ONE batch -- the stitching itself is a few elementwise ops on [384, w].
"""
import torch
import torch.nn.functional as F


def window_starts(width):
    """Start columns produced by the reference loop (FSC_test_cross(few-shot).py:326-349)."""
    starts, start = [], 0
    while start + 383 < width:
        starts.append(start)
        start += 128
        if start + 383 >= width:
            if start == width - 384 + 128:
                break
            start = width - 384
    return starts


def blend_windows(outputs, starts, width, height=384):
    """Sequential blend of per-window densities [n, height, 384] exactly as the reference does it."""
    dm = torch.zeros(height, width, device=outputs.device, dtype=outputs.dtype)
    prev = -1
    for out, start in zip(outputs, starts):
        ov = prev - start + 1
        if ov > 0:
            dm[:, start:prev + 1] = dm[:, start:prev + 1] / 2 + out[:, :ov] / 2
        dm[:, prev + 1:start + 384] = out[:, max(ov, 0):]
        prev = start + 383
    return dm


@torch.no_grad()
def density_map(model, samples, boxes, shot_num, max_batch=32):
    """samples [1, 3, 384, w] -> stitched density [384, w]; all windows run as batched forwards."""
    _, _, h, w = samples.shape
    starts = window_starts(w)
    if not starts:
        return torch.zeros(h, w, device=samples.device)
    wins = torch.cat([samples[:, :, :, s:s + 384] for s in starts], 0)
    bx = boxes.expand(len(starts), *boxes.shape[1:]) if boxes.nelement() > 0 else boxes.new_zeros((len(starts), 0))
    outs = []
    for i in range(0, len(starts), max_batch):
        outs.append(model(wins[i:i + max_batch].contiguous(), bx[i:i + max_batch].contiguous(), shot_num))
    return blend_windows(torch.cat(outs, 0), starts, w, h)


@torch.no_grad()
def count_image(model, samples, boxes, shot_num, pos=None, normalization=True, max_s_cnt=1):
    """Full per-image test path: returns (pred_cnt, density_map).  pos: exemplar rectangles [(y1, x1, y2, x2), ...]."""
    _, _, h, w = samples.shape
    s_cnt = 0
    for rect in (pos or [])[:3]:
        if rect[2] - rect[0] < 10 and rect[3] - rect[1] < 10:
            s_cnt += 1
    if pos is not None and s_cnt >= max_s_cnt:
        # 3x3 split: each crop is upscaled back to (h, w) and counted on its own (FSC_test_cross(few-shot).py:273-320)
        pred, dm = 0.0, None
        for (top, left) in ((0, 0), (h // 3, 0), (0, w // 3), (h // 3, w // 3), (h * 2 // 3, 0), (h * 2 // 3, w // 3),
                            (0, w * 2 // 3), (h // 3, w * 2 // 3), (h * 2 // 3, w * 2 // 3)):
            crop = samples[:, :, top:top + h // 3, left:left + w // 3]
            crop = F.interpolate(crop, size=(h, w), mode="bilinear", align_corners=False)
            dm = density_map(model, crop, boxes, shot_num)
            pred += (dm.sum() / 60).item()
    else:
        dm = density_map(model, samples, boxes, shot_num)
        pred = (dm.sum() / 60).item()
    if normalization and pos:
        e_cnt = sum((dm[r[0]:r[2] + 1, r[1]:r[3] + 1].sum() / 60).item() for r in pos) / 3
        if e_cnt > 1.8:
            pred /= e_cnt
    return pred, dm
