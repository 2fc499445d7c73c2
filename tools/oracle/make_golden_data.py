#!/usr/bin/env python3
"""Golden vectors of the reference's dataset transforms (runs only in the build container).

util/FSC147.py imports torchvision, cv2 and imgaug, none of which are installed.  The reference module itself is imported and its
classes run UNCHANGED (ResizeTrainImage with do_aug=False, ResizeValImage, ResizePreTrainImage's resize rule); only the missing
third-party names are supplied by thin stand-ins with the documented behaviour of the pinned versions (torchvision 0.14.1):
  transforms.Resize(size)   PIL image -> img.resize((w, h), BILINEAR) ; tensor -> F.interpolate(bilinear, align_corners=False),
                            no antialias (tensors default to antialias=None -> False in 0.14)
  transforms.ToTensor       uint8 HWC -> float CHW / 255
  transforms.Compose        function composition
  TF.crop / TF.hflip        tensor slicing / flip
  RandomResizedCrop, RandomHorizontalFlip, Normalize: constructed at import time by the reference module but never CALLED on the
                            paths exercised here -> placeholders that raise if called
  cv2                       imported at module level only; not used on these paths -> empty module
Augmented path (ResizeTrainImage with do_aug=True, second half of this script): ColorJitter, GaussianBlur and the imgaug affine are
replaced by IDENTITY stand-ins (ColorJitter / GaussianBlur return their input, iaa.Sequential returns image and keypoints unchanged,
Keypoint / KeypointsOnImage carry x, y and the is_out_of_image test).  What the goldens then pin is everything else the reference
does around them, unchanged: the draw order from `random`, np.random.normal noise, dot maps, flip, random crop, and the whole mosaic
(whose output never depends on the three replaced ops -- it is built from the clean resized image).  The product's own colour
jitter / blur / affine are tested separately (identity parameters reproduce these goldens; closed-form properties otherwise).
Output: tests/golden/data.npz, tests/golden/data_aug.npz; the tests (tests/test_data_cpu.py) rebuild the same synthetic images /
annotations from seeds."""
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def install_stand_ins():
    from PIL import Image

    class Resize:
        def __init__(self, size, interpolation=2):
            self.size = size

        def __call__(self, img):
            if isinstance(img, torch.Tensor):
                return F.interpolate(img.unsqueeze(0), size=tuple(self.size), mode="bilinear", align_corners=False)[0]
            return img.resize((self.size[1], self.size[0]), Image.BILINEAR)

    class ToTensor:
        def __call__(self, img):
            if isinstance(img, np.ndarray) and img.dtype != np.uint8:      # torchvision: non-uint8 arrays are only transposed
                return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
            return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    def placeholder(name):
        class P:
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                raise RuntimeError("%s stand-in called: this path is not covered by the golden generator" % name)
        P.__name__ = name
        return P

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tr.Resize, tr.ToTensor, tr.Compose = Resize, ToTensor, Compose
    for n in ("RandomResizedCrop", "RandomHorizontalFlip", "Normalize"):
        setattr(tr, n, placeholder(n))

    class Identity:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x
    tr.ColorJitter = tr.GaussianBlur = Identity
    tf = types.ModuleType("torchvision.transforms.functional")
    tf.crop = lambda img, top, left, h, w: img[..., top:top + h, left:left + w]
    tf.hflip = lambda img: img.flip(-1)
    tr.functional = tf
    tv.transforms = tr
    ia = types.ModuleType("imgaug")
    iaa = types.ModuleType("imgaug.augmenters")
    iab = types.ModuleType("imgaug.augmentables")

    class Keypoint:
        def __init__(self, x, y):
            self.x, self.y = x, y

        def is_out_of_image(self, image):
            h, w = image.shape[0:2]
            return self.x < 0 or self.x >= w or self.y < 0 or self.y >= h

    class KeypointsOnImage:
        def __init__(self, keypoints, shape):
            self.keypoints, self.shape = keypoints, shape

    class Sequential:
        def __init__(self, children):
            pass

        def __call__(self, image, keypoints):
            return image, keypoints
    iab.Keypoint, iab.KeypointsOnImage = Keypoint, KeypointsOnImage
    iaa.Sequential = Sequential
    iaa.Affine = lambda **k: None
    ia.augmenters, ia.augmentables = iaa, iab
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": tf,
                        "cv2": types.ModuleType("cv2"), "imgaug": ia, "imgaug.augmenters": iaa, "imgaug.augmentables": iab})


def main():
    from oracle import weights as W
    install_stand_ins()
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import util.FSC147 as ref
    out = {}
    # the transform classes read annotation / split files in __init__: hand them a tiny on-disk pair
    import json
    import tempfile
    tmp = tempfile.mkdtemp()
    json.dump({}, open(os.path.join(tmp, "a.json"), "w"))
    json.dump({"train": []}, open(os.path.join(tmp, "s.json"), "w"))
    args = types.SimpleNamespace(im_dir=tmp, anno_file=os.path.join(tmp, "a.json"), data_split_file=os.path.join(tmp, "s.json"),
                                 do_aug=False, class_file=None)
    train_t = ref.ResizeTrainImage(args, do_aug=False)
    val_t = ref.ResizeValImage(args)
    for k, (w, h) in enumerate(W.DATA_CASES):
        image, rects, dots = W.make_fsc_item(k, w, h)
        random.seed(100 + k)
        s = train_t({"image": image, "lines_boxes": rects, "dots": dots, "id": "x.jpg", "m_flag": 0})
        out["train%d_image" % k] = s["image"].numpy()[:, ::4, ::4].copy()         # every 4th pixel ...
        out["train%d_image_sum" % k] = s["image"].double().sum(dim=(1, 2)).numpy()  # ... plus exact channel sums of the full image
        out["train%d_density" % k] = s["gt_density"].numpy()
        out["train%d_boxes" % k] = s["boxes"].numpy()
        out["train%d_pos" % k] = s["pos"].numpy()
        out["train%d_flex" % k] = np.array(train_t.flex_resize(h, w))
        v = val_t({"image": image, "lines_boxes": rects, "dots": dots, "m_flag": 0})
        out["val%d_image" % k] = v["image"].numpy()[:, ::4, ::4].copy()
        out["val%d_image_sum" % k] = v["image"].double().sum(dim=(1, 2)).numpy()
        out["val%d_density" % k] = v["gt_density"].numpy()
        out["val%d_boxes" % k] = v["boxes"].numpy()
        out["val%d_pos" % k] = v["pos"].numpy()
    np.savez_compressed(os.path.join(OUT, "data.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, float(np.asarray(v, np.float64).sum()))

    # ---- augmented path (do_aug=True) on the six-image on-disk dataset
    from PIL import Image
    root = tempfile.mkdtemp()
    anno_f, split_f, class_f, im_dir, ids = W.write_aug_dataset(root)
    aargs = types.SimpleNamespace(im_dir=im_dir, anno_file=anno_f, data_split_file=split_f, do_aug=True, class_file=class_f)
    aug_t = ref.ResizeTrainImage(aargs, do_aug=True)
    annos = json.load(open(anno_f))
    out = {}
    cases = []
    for k in (0, 2, 3, 5):
        want = {"mosaic": 2, "crop": 1}
        seed = 0
        while any(want.values()):
            seed += 1
            kind = "mosaic" if random.Random(seed).random() < 0.25 else "crop"
            if want[kind]:
                want[kind] -= 1
                cases.append((k, seed, kind))
    for n, (k, seed, kind) in enumerate(cases):
        im_id = ids[k]
        image = Image.open(os.path.join(im_dir, im_id)); image.load()
        a = annos[im_id]
        dots = np.array(a["points"])
        rects = [[b[0][1], b[0][0], b[2][1], b[2][0]] for b in a["box_examples_coordinates"]]
        random.seed(seed); np.random.seed(seed)
        s = aug_t({"image": image, "lines_boxes": rects, "dots": dots, "id": im_id, "m_flag": 0})
        img = s["image"].double()
        out["c%d_image" % n] = img.numpy()[:, ::4, ::4].astype(np.float32)
        out["c%d_image_sum" % n] = img.sum(dim=(1, 2)).numpy()
        out["c%d_rowsum" % n] = img.sum(dim=(0, 2)).numpy()
        out["c%d_density" % n] = s["gt_density"].numpy().astype(np.float32)
        out["c%d_boxes" % n] = s["boxes"].numpy()
        out["c%d_meta" % n] = np.array([k, seed, int(kind == "mosaic"), int(s["m_flag"]), int(s["pos"].numel())])
    out["ncases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "data_aug.npz"), **out)
    for k, v in out.items():
        print(k, np.asarray(v).shape, float(np.asarray(v, np.float64).sum()))


def main_test_loader():
    """tests/golden/data_test.npz: the reference's TEST-TIME loader.  `class TestData` (FSC_test_cross(few-shot).py:82-190, its
    external-exemplar branch :96-129 included) is read from /root/reference at run time and exec'd UNCHANGED against the module globals
    it expects (annotations, data_split, im_dir; torchvision's Resize / ToTensor / Compose are the stand-ins documented above,
    misc.measure_time a no-op context manager) on the six-image on-disk dataset of the augmentation fixtures."""
    import contextlib
    import json
    import tempfile
    from PIL import Image
    from scipy import ndimage
    from torch.utils.data import Dataset
    from oracle import weights as W
    root = tempfile.mkdtemp()
    anno_f, split_f, _class_f, im_dir, ids = W.write_aug_dataset(root)
    src = open(os.path.join(REF, "FSC_test_cross(few-shot).py")).read().split("\n")
    first = next(i for i, l in enumerate(src) if l.startswith("class TestData"))
    last = next(i for i in range(first + 1, len(src)) if src[i].startswith("def main"))
    tr = sys.modules["torchvision.transforms"]

    class _MT:
        duration = 0.0

    @contextlib.contextmanager
    def measure_time():
        yield _MT()
    ns = {"Dataset": Dataset, "Image": Image, "transforms": tr, "np": np, "torch": torch, "ndimage": ndimage,
          "misc": types.SimpleNamespace(measure_time=measure_time), "annotations": json.load(open(anno_f)),
          "data_split": {"test": ids, "val": ids[:3]}, "im_dir": im_dir}
    exec(compile("\n".join(src[first:last]), "TestData", "exec"), ns)
    out = {}
    n = 0
    for external, bound, split in ((False, -1, "test"), (False, 2, "test"), (False, 0, "test"), (True, 3, "val"), (True, -1, "val"), (True, 2, "test")):
        ds = ns["TestData"](external=external, box_bound=bound, split=split)
        for idx in range(len(ds)):
            image, dots, boxes, pos, gt_map, name, _dur = ds[idx]
            out["t%d_meta" % n] = np.array([int(external), bound, ids.index(name), dots.shape[0], len(pos), 0 if split == "test" else 1])
            out["t%d_image" % n] = image.numpy()[:, ::4, ::4].copy()
            out["t%d_image_sum" % n] = image.double().sum(dim=(1, 2)).numpy()
            bx = np.asarray(boxes.numpy() if isinstance(boxes, torch.Tensor) else boxes, np.float32)
            # full crops once per distinct exemplar set (bound 2 / 0 are prefixes of bound -1; the external list is the same for every
            # image of a configuration); a float64 checksum per box everywhere
            if (not external and bound == -1) or (external and idx == 0):
                out["t%d_boxes" % n] = bx
            out["t%d_boxes_sum" % n] = bx.reshape(bx.shape[0], -1).astype(np.float64).sum(1) if bx.size else np.zeros(0)
            out["t%d_pos" % n] = np.asarray(pos, np.int64).reshape(-1, 4)
            out["t%d_gt_sum" % n] = np.float64(gt_map.double().sum().item())
            out["t%d_gt" % n] = gt_map.numpy()[::4, ::4].copy()
            n += 1
    out["ncases"] = np.int64(n)
    np.savez_compressed(os.path.join(OUT, "data_test.npz"), **out)
    print("data_test.npz: %d cases" % n, {k: np.asarray(v).shape for k, v in out.items() if k.startswith("t0_") or k.startswith("t%d_" % (n - 1))})


if __name__ == "__main__":
    if "--test-loader-only" not in sys.argv:
        main()
    else:
        install_stand_ins()
    main_test_loader()
