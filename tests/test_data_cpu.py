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


def _identity_params():
    from countr_amd.data import fsc147 as D
    p = D.AugParams(np.random.RandomState(0))
    p.order, p.brightness, p.contrast, p.saturation, p.hue = [0, 1, 2, 3], 1.0, 1.0, 1.0, 0.0
    p.sigma, p.rotate, p.scale, p.shear, p.tx, p.ty = 1e-4, 0.0, 1.0, 0.0, 0.0, 0.0
    return p


def test_augmented_transform_matches_reference_golden(tmp_path):
    """ResizeTrainImage(do_aug=True) of util/FSC147.py run unchanged by tools/oracle/make_golden_data.py with IDENTITY stand-ins for the
    three third-party ops (ColorJitter, GaussianBlur, imgaug affine): 8 mosaic cases (self-mosaic for the 80-dot image, cross-image
    mosaic with the class test otherwise) and 4 noise + flip + random-crop cases.  transform_train_aug with identity parameters
    must reproduce them: same `random` draw order, same noise, dot maps, blending and crops."""
    from countr_amd.data import fsc147 as D
    from oracle import weights as W
    import types
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_aug.npz"))
    anno_f, split_f, class_f, im_dir, ids = W.write_aug_dataset(str(tmp_path))
    args = types.SimpleNamespace(data_path=str(tmp_path), anno_file=anno_f, data_split_file=split_f, im_dir=im_dir, class_file=class_f)
    ds = D.TrainData(args, split="train", do_aug=True)
    kinds = set()
    for n in range(int(g["ncases"])):
        k, seed, is_mosaic, m_flag, npos = (int(v) for v in g["c%d_meta" % n])
        im_id = ids[k]
        a = ds.annotations[im_id]
        rects = [[b[0][1], b[0][0], b[2][1], b[2][0]] for b in a["box_examples_coordinates"]]
        s = D.transform_train_aug(ds.open_image(im_id), rects, np.array(a["points"]), im_id, ds, rng=random.Random(seed),
                                  nprng=np.random.RandomState(seed), params=_identity_params())
        kinds.add((is_mosaic, m_flag))
        assert s["m_flag"] == m_flag and s["pos"].numel() == npos == 0
        img = s["image"].double()
        assert img.shape == (3, 384, 384)
        tol = 2e-6 if is_mosaic else 2e-5      # the crop path carries the blur / warp identities (fp32 round trips)
        assert np.abs(img.numpy()[:, ::4, ::4] - g["c%d_image" % n]).max() <= tol, n
        assert np.abs(img.sum(dim=(1, 2)).numpy() - g["c%d_image_sum" % n]).max() <= tol * 384 * 384
        assert np.abs(img.sum(dim=(0, 2)).numpy() - g["c%d_rowsum" % n]).max() <= tol * 3 * 384
        assert np.abs(s["gt_density"].numpy() - g["c%d_density" % n]).max() <= 1e-5, n
        assert np.abs(s["boxes"].numpy() - g["c%d_boxes" % n]).max() <= 1e-6, n
    assert kinds == {(1, 0), (1, 1), (0, 0)}     # self-mosaic, cross-image mosaic, plain crop all covered


def test_colour_jitter_blur_and_affine_properties():
    """The re-implemented third-party ops (no reference output exists offline): closed-form properties."""
    from countr_amd.data import fsc147 as D
    rs = np.random.RandomState(5)
    img = torch.from_numpy(rs.uniform(0, 1, (3, 40, 52)).astype(np.float32))
    # colour jitter: neutral factors are the identity in any order; saturation 0 = grayscale; brightness scales; hue keeps value
    assert torch.allclose(D.color_jitter(img, [3, 1, 0, 2], 1.0, 1.0, 1.0, 0.0), img, atol=2e-6)
    gray = D.color_jitter(img, [2], 1.0, 1.0, 0.0, 0.0)
    assert torch.allclose(gray[0], gray[1]) and torch.allclose(gray[0], 0.2989 * img[0] + 0.587 * img[1] + 0.114 * img[2], atol=1e-6)
    assert torch.allclose(D.color_jitter(img, [0], 0.5, 1.0, 1.0, 0.0), img * 0.5, atol=1e-6)
    hue = D.color_jitter(img, [3], 1.0, 1.0, 1.0, 0.1)
    assert torch.allclose(hue.max(0).values, img.max(0).values, atol=1e-5)          # V of HSV is untouched by a hue rotation
    assert torch.allclose(D.color_jitter(hue, [3], 1.0, 1.0, 1.0, -0.1), img, atol=1e-4)
    flat = torch.full((3, 20, 20), 0.3)
    assert torch.allclose(D.color_jitter(flat, [1], 1.0, 1.7, 1.0, 0.0), flat, atol=1e-4)   # contrast about the (0.9999-weighted) gray mean
    # blur: constant images are fixed points (reflect padding, kernel sums to 1); sigma -> 0 is the identity; energy shrinks
    assert torch.allclose(D.gaussian_blur(flat, (7, 9), 1.3), flat, atol=1e-6)
    assert torch.allclose(D.gaussian_blur(img, (7, 9), 1e-4), img, atol=1e-6)
    b = D.gaussian_blur(img, (7, 9), 1.5)
    assert b.shape == img.shape and b.var() < img.var()
    from scipy import ndimage
    ref = ndimage.gaussian_filter(img[0].numpy().astype(np.float64), sigma=1.5, mode="mirror", truncate=100)  # same gaussian, untruncated
    assert np.abs(b[0, 12:-12, 12:-12].numpy() - ref[12:-12, 12:-12]).max() < 0.02
    # affine: identity parameters are the identity; a pure translation moves pixels and key points alike; 90-degree-free check
    M = D.affine_matrix(40, 52, 0.0, 1.0, 0.0, 0.0, 0.0)
    assert np.allclose(M, np.eye(3)) and torch.allclose(D.warp_affine(img, M), img, atol=1e-6)
    M = D.affine_matrix(40, 52, 0.0, 1.0, 0.0, 0.25, -0.1)            # +13 px in x, -4 px in y
    w = D.warp_affine(img, M)
    assert torch.allclose(w[:, 0:36, 13:52], img[:, 4:40, 0:39], atol=1e-5) and float(w[:, :, :13].abs().max()) == 0.0
    assert np.allclose(M @ np.array([5.0, 10.0, 1.0]), [18.0, 6.0, 1.0])
    M = D.affine_matrix(41, 41, 90.0, 1.0, 0.0, 0.0, 0.0)              # rotation about the centre pixel maps the grid onto itself
    sq = torch.from_numpy(rs.uniform(0, 1, (1, 41, 41)).astype(np.float32))
    r = D.warp_affine(sq, M)
    inner = lambda t: t[:, 1:-1, 1:-1]        # border samples land a rounding error outside the image -> zero fill
    assert torch.allclose(inner(r), inner(torch.rot90(sq, -1, (1, 2))), atol=1e-5) or torch.allclose(inner(r), inner(torch.rot90(sq, 1, (1, 2))), atol=1e-5)
    M = D.affine_matrix(41, 41, 0.0, 1.2, 0.0, 0.0, 0.0)
    assert np.allclose(M @ np.array([20.0, 20.0, 1.0]), [20.0, 20.0, 1.0])   # the centre is the fixed point of scale / shear / rotate


def test_augmented_dataset_end_to_end(tmp_path):
    """TrainData(do_aug=True): random parameters, 30 draws over all images -- shapes, finiteness, count bookkeeping (a non-mosaic sample
    can only lose dots to the warp / crop; the density integrates to 60 per surviving dot) and the class-file requirement."""
    from countr_amd.data import fsc147 as D
    from oracle import weights as W
    import types
    anno_f, split_f, class_f, im_dir, ids = W.write_aug_dataset(str(tmp_path))
    args = types.SimpleNamespace(data_path=str(tmp_path), anno_file=anno_f, data_split_file=split_f, im_dir=im_dir, class_file=class_f)
    random.seed(3); np.random.seed(3)
    ds = D.TrainData(args, split="train", do_aug=True)
    flags = set()
    for it in range(30):
        image, dens, ndots, boxes, pos, m_flag, im_id = ds[it % len(ds)]
        assert image.shape == (3, 384, 384) and image.dtype == torch.float32 and boxes.shape == (3, 3, 64, 64)
        assert torch.isfinite(image).all() and -1e-5 <= float(image.min()) and float(image.max()) <= 1.0 + 1e-5
        cnt = float(dens.sum()) / 60
        assert abs(cnt - round(cnt)) < 1e-3 or cnt > 0          # gaussian of unit dots (mass leaks only at the border)
        flags.add(int(m_flag))
        if m_flag == 0 and ndots < 70:
            assert cnt <= ndots + 1e-3
    assert flags == {0, 1}
    with pytest.raises(FileNotFoundError):
        D.TrainData(types.SimpleNamespace(data_path=str(tmp_path), anno_file=anno_f, data_split_file=split_f, im_dir=im_dir,
                                          class_file="missing.txt"), split="train", do_aug=True)


def test_test_time_loader_matches_reference_testdata(tmp_path):
    """countr_amd/data/fsc147.py::test_item / external_exemplars against the reference's own TestData class
    (FSC_test_cross(few-shot).py:82-190, exec'd unchanged by tools/oracle/make_golden_data.py on the same six-image dataset):
    image, exemplar crops, box positions, gaussian ground-truth map; few-shot, --box_bound 2 / 0 and the --external branch."""
    from countr_amd.data import fsc147 as D
    from oracle import weights as W
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_test.npz"))
    anno_f, _split_f, _class_f, im_dir, ids = W.write_aug_dataset(str(tmp_path))
    annotations = json.load(open(anno_f))
    splits = {0: ids, 1: ids[:3]}
    ext_cache = {}
    full_boxes = {}
    assert int(g["ncases"]) == 30
    for n in range(int(g["ncases"])):
        external, bound, k, ndots, npos, split_id = (int(v) for v in g["t%d_meta" % n])
        ext = None
        if external:
            key = (bound, split_id)
            if key not in ext_cache:
                ext_cache[key] = D.external_exemplars(annotations, splits[split_id], im_dir, bound)
            ext = ext_cache[key]
        img, dots, boxes, pos, gt = D.test_item(annotations, im_dir, ids[k], bound, ext)
        assert dots.shape[0] == ndots and len(pos) == npos
        ref_img = g["t%d_image" % n]
        assert img.shape[1] == 384 and img.shape[2] % 16 == 0 and -(-img.shape[2] // 4) == ref_img.shape[2]
        assert np.abs(img.numpy()[:, ::4, ::4] - ref_img).max() <= 1e-6
        assert np.abs(img.double().sum(dim=(1, 2)).numpy() - g["t%d_image_sum" % n]).max() <= 1e-6 * img[0].numel()
        assert np.array_equal(np.asarray(pos, np.int64).reshape(-1, 4), g["t%d_pos" % n])
        bsum = boxes.reshape(boxes.shape[0], -1).double().sum(1).numpy() if boxes.numel() else np.zeros(0)
        assert bsum.shape == g["t%d_boxes_sum" % n].shape and np.abs(bsum - g["t%d_boxes_sum" % n]).max(initial=0.0) <= 1e-3
        if "t%d_boxes" % n in g.files:
            assert np.abs(boxes.numpy() - g["t%d_boxes" % n]).max() <= 1e-6
        assert abs(gt.double().sum().item() - float(g["t%d_gt_sum" % n])) <= 1e-3
        assert np.abs(gt.numpy()[::4, ::4] - g["t%d_gt" % n]).max() <= 1e-5
        if external:
            assert pos == [] and boxes.shape[0] == (bound if bound >= 0 else sum(len(annotations[i]["box_examples_coordinates"]) for i in splits[split_id]))
