"""CPU: the PIL/scipy FSC147 loaders (countr_amd/data/fsc147.py) on a synthetic on-disk dataset: shapes and dtypes of the
reference's sample tuple, count preservation of the density maps, exemplar crop geometry, resize rules, DataLoader collation."""
import argparse
import json
import os
import random

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def fake_fsc(tmp_path_factory):
    from PIL import Image
    root = tmp_path_factory.mktemp("fsc")
    im_dir = root / "images_384_VarV2"
    im_dir.mkdir()
    rs = np.random.RandomState(0)
    anno, names = {}, []
    for k, (w, h) in enumerate([(640, 384), (384, 600), (500, 400), (900, 384)]):
        name = "%d.jpg" % k
        yy, xx = np.mgrid[0:h, 0:w]
        arr = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx + yy) % 256)], -1).astype(np.uint8)
        Image.fromarray(arr).save(im_dir / name, quality=95)
        pts = np.stack([rs.uniform(5, w - 5, 20), rs.uniform(5, h - 5, 20)], 1).tolist()
        boxes = []
        for b in range(3):
            x1, y1 = int(rs.uniform(0, w - 80)), int(rs.uniform(0, h - 80))
            x2, y2 = x1 + int(rs.uniform(20, 70)), y1 + int(rs.uniform(20, 70))
            boxes.append([[x1, y1], [x1, y2], [x2, y2], [x2, y1]])
        anno[name] = {"points": pts, "box_examples_coordinates": boxes}
        names.append(name)
    json.dump(anno, open(root / "annotation_FSC147_384.json", "w"))
    json.dump({"train": names, "val": names[:2], "test": names[2:]}, open(root / "Train_Test_Val_FSC_147.json", "w"))
    return argparse.Namespace(data_path=str(root), anno_file="annotation_FSC147_384.json",
                              data_split_file="Train_Test_Val_FSC_147.json", im_dir="images_384_VarV2")


def test_resize_rules():
    from countr_amd.data import fsc147 as D
    assert D.flex_resize(384, 640) == (384, 640)           # already >= 384 on both sides: multiples of 16
    assert D.flex_resize(300, 500) == (384, 640)           # smaller side brought to 384
    assert D.flex_resize(600, 200) == (1152, 384)
    assert D.flex_resize(391, 401) == (384, 400)
    rng = random.Random(0)
    for _ in range(50):
        i, j, h, w = D.random_resized_crop_params(640, 384, rng=rng)
        assert 0 <= i and i + h <= 384 and 0 <= j and j + w <= 640 and 0.2 * 0.95 <= h * w / (640 * 384) <= 1.0 and 0.74 <= w / h <= 1.34


def test_train_and_val_samples(fake_fsc):
    from countr_amd.data import fsc147 as D
    assert D.available(fake_fsc)
    random.seed(0)
    ds = D.TrainData(fake_fsc, split="train", do_aug=False)
    assert len(ds) == 4
    for idx in range(4):
        img, dens, n, boxes, pos, m_flag, im_id = ds[idx]
        assert img.shape == (3, 384, 384) and img.dtype == torch.float32 and 0 <= img.min() and img.max() <= 1
        assert dens.shape == (384, 384) and boxes.shape == (3, 3, 64, 64) and pos.shape == (3, 4) and n == 20 and m_flag == 0
        # the gaussian is normalised: the density integrates to 60 x (dots inside the crop); never more than all dots
        cnt = dens.sum().item() / 60
        assert -1e-3 <= cnt <= 20 + 1e-3 and abs(cnt - round(cnt)) < 1e-3
    dv = D.TrainData(fake_fsc, split="val")
    img, dens, n, boxes, pos, _, _ = dv[0]
    assert img.shape == (3, 384, 384) and abs(dens.sum().item() / 60 - 20) < 0.2      # sigma 4 / radius 7: all dots kept
    y1, x1, y2, x2 = [int(v) for v in pos[0]]
    ref = torch.nn.functional.interpolate(img[:, y1:y2 + 1, x1:x2 + 1].unsqueeze(0), size=(64, 64), mode="bilinear", align_corners=False)[0]
    assert torch.equal(boxes[0], ref)                                                   # exemplars are crops of the resized image


def test_pretrain_samples_and_collation(fake_fsc):
    from countr_amd.data import fsc147 as D
    random.seed(1)
    ds = D.PretrainData(fake_fsc)
    dl = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0, drop_last=False)
    batches = list(dl)
    assert len(batches) == 2 and batches[0].shape == (2, 3, 384, 384) and batches[0].dtype == torch.float32
    random.seed(0)
    dt = D.TrainData(fake_fsc, split="train", do_aug=False)
    imgs, dens, n, boxes, pos, m_flag, ids = next(iter(torch.utils.data.DataLoader(dt, batch_size=4)))
    assert imgs.shape == (4, 3, 384, 384) and dens.shape == (4, 384, 384) and boxes.shape == (4, 3, 3, 64, 64) and len(ids) == 4


def test_transforms_match_reference_golden():
    """The non-augmented train transform and the val transform against util/FSC147.py ITSELF: tools/oracle/make_golden_data.py
    imports the reference module (torchvision / cv2 / imgaug replaced by thin stand-ins with their documented behaviour) and runs
    ResizeTrainImage(do_aug=False) and ResizeValImage unchanged on the synthetic items of oracle/weights.make_fsc_item."""
    from countr_amd.data import fsc147 as D
    from oracle import weights as W
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "data.npz"))
    for k, (w, h) in enumerate(W.DATA_CASES):
        image, rects, dots = W.make_fsc_item(k, w, h)
        assert tuple(g["train%d_flex" % k]) == D.flex_resize(h, w)
        s = D.transform_train_noaug(image, rects, dots, rng=random.Random(100 + k))   # the generator seeds `random` the same way
        v = D.transform_val(image, rects, dots)
        for tag, got in (("train", s), ("val", v)):
            img = got["image"]
            assert img.shape == (3, 384, 384)
            assert np.abs(img.numpy()[:, ::4, ::4] - g["%s%d_image" % (tag, k)]).max() <= 1e-6, (tag, k)
            assert np.abs(img.double().sum(dim=(1, 2)).numpy() - g["%s%d_image_sum" % (tag, k)]).max() <= 1e-6 * 384 * 384
            assert np.abs(got["gt_density"].numpy() - g["%s%d_density" % (tag, k)]).max() <= 1e-6, (tag, k)
            assert np.abs(got["boxes"].numpy() - g["%s%d_boxes" % (tag, k)]).max() <= 1e-6, (tag, k)
            assert np.array_equal(np.asarray(got["pos"]), g["%s%d_pos" % (tag, k)]), (tag, k)
