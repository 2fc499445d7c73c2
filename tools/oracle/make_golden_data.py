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
  ColorJitter, GaussianBlur, RandomResizedCrop, RandomHorizontalFlip, Normalize: constructed at import time by the reference
                            module but never CALLED on the non-augmented paths exercised here -> placeholders that raise if called
  cv2, imgaug               imported at module level only; not used on these paths -> empty modules
Output: tests/golden/data.npz; the test (tests/test_data_cpu.py) rebuilds the same synthetic images / annotations from seeds."""
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
    for n in ("ColorJitter", "GaussianBlur", "RandomResizedCrop", "RandomHorizontalFlip", "Normalize"):
        setattr(tr, n, placeholder(n))
    tf = types.ModuleType("torchvision.transforms.functional")
    tf.crop = lambda img, top, left, h, w: img[..., top:top + h, left:left + w]
    tf.hflip = lambda img: img.flip(-1)
    tr.functional = tf
    tv.transforms = tr
    ia = types.ModuleType("imgaug")
    iaa = types.ModuleType("imgaug.augmenters")
    iab = types.ModuleType("imgaug.augmentables")
    iab.Keypoint = iab.KeypointsOnImage = placeholder("imgaug")
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


if __name__ == "__main__":
    main()
