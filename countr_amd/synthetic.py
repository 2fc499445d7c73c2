"""Synthetic FSC147-shaped inputs (no dataset is available offline): SURVEY.md section 8d.

imgs, boxes ~ U[0,1); density = K ~ U{5..200} random dots -> gaussian(sigma=1) x 60 (as util/FSC147.py:275-278);
loss mask ~ Bernoulli(0.8) over [384,384] (FSC_finetune_cross.py:290)."""
import numpy as np
import torch


def make_batch(batch, shots=3, seed=0, img_size=384, device="cpu"):
    rs = np.random.RandomState(1000 + seed)
    imgs = rs.uniform(0, 1, size=(batch, 3, img_size, img_size)).astype(np.float32)
    boxes = rs.uniform(0, 1, size=(batch, shots, 3, 64, 64)).astype(np.float32)
    gt = np.zeros((batch, img_size, img_size), dtype=np.float32)
    from scipy.ndimage import gaussian_filter
    for b in range(batch):
        k = rs.randint(5, 201)
        np.add.at(gt[b], (rs.randint(0, img_size, size=k), rs.randint(0, img_size, size=k)), 1.0)
        gt[b] = gaussian_filter(gt[b], sigma=(1, 1), order=0) * 60.0
    mask = rs.binomial(1, 0.8, size=(img_size, img_size)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(device)
    return t(imgs), t(boxes), t(gt), t(mask)


def wide_frames(n, width=672, device="cpu", seed=0, height=384):
    """n test-time frames [1, 3, 384, width] ~ U[0,1): a 1920x1080 image resized as the reference's test loader does
    (height 384, width 16 * int(1920 / 1080 * 384 / 16) = 672: FSC_test_cross(few-shot).py:150-153, demo_zero.py:28-31)."""
    rs = np.random.RandomState(7000 + seed)
    return [torch.from_numpy(rs.uniform(0, 1, size=(1, 3, height, width)).astype(np.float32)).to(device) for _ in range(n)]
