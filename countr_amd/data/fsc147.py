"""FSC147 datasets with PIL + scipy only: the reference's loaders need torchvision / cv2 / imgaug,
none of which exist offline.  Restated here: the NON-augmented train transform, the val transform and the pretrain transform of
util/FSC147.py, and the dataset classes of FSC_finetune_cross.py:113-154 / FSC_pretrain.py:114-143.  The augmentation branch
(Gaussian noise, colour jitter, blur, affine, flip, mosaic: util/FSC147.py:130-253) is out of scope; `do_aug=True` falls back to
the non-augmented transform with a one-time warning.  Host-side code: it produces the [B,3,384,384] / [B,3,3,64,64] / [B,384,384]
batches that FinetuneStep.load() / PretrainStep.load() stage into the engine.

PIL's Image.resize(BILINEAR / BICUBIC) is what torchvision.transforms.Resize calls for PIL inputs; exemplar crops are resized
as TENSORS by the reference (torchvision 0.14.1: bilinear, no antialias) == F.interpolate(mode="bilinear", align_corners=False)."""
import json
import math
import os
import random
import warnings

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

MAX_HW = 384


def to_tensor(img):
    """transforms.ToTensor() on an RGB PIL image."""
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)


def resize_box(t, size=64):
    return F.interpolate(t.unsqueeze(0), size=(size, size), mode="bilinear", align_corners=False)[0]


def dot_map(dots, new_h, new_w, sh, sw):
    m = np.zeros((new_h, new_w), dtype="float32")
    for d in dots:
        m[min(new_h - 1, int(d[1] * sh))][min(new_w - 1, int(d[0] * sw))] = 1
    return m


def exemplar_crops(img_t, rects, sh, sw, limit=3):
    """util/FSC147.py:274-293 / 345-360: first `limit` boxes, scaled with int() truncation, inclusive crop, 64x64 bilinear."""
    boxes, scaled = [], []
    for box in rects[:limit]:
        b = [int(k) for k in box]
        y1, x1, y2, x2 = int(b[0] * sh), int(b[1] * sw), int(b[2] * sh), int(b[3] * sw)
        scaled.append((y1, x1, y2, x2))
        boxes.append(resize_box(img_t[:, y1:y2 + 1, x1:x2 + 1]))
    return torch.stack(boxes), scaled


def flex_resize(h, w, max_hw=MAX_HW):
    """ResizeTrainImage.flex_resize (util/FSC147.py:102-115)."""
    if h < max_hw <= w or h <= w < max_hw:
        new_h = max_hw
        new_w = round(w * new_h / h)
    elif w < max_hw <= h or w < h < max_hw:
        new_w = max_hw
        new_h = round(h * new_w / w)
    else:
        new_w = 16 * int(w / 16)
        new_h = 16 * int(h / 16)
    return new_h, new_w


def transform_train_noaug(image, rects, dots, rng=random):
    """ResizeTrainImage.__call__ with aug_flag False (util/FSC147.py:117-128, 255-300)."""
    from PIL import Image
    from scipy import ndimage
    W, H = image.size
    new_h, new_w = flex_resize(H, W)
    sh, sw = float(new_h) / H, float(new_w) / W
    img_t = to_tensor(image.resize((new_w, new_h), Image.BILINEAR))
    dens = dot_map(dots, new_h, new_w, sh, sw)
    rng.random()     # the reference draws its mosaic coin before looking at do_aug (util/FSC147.py:127): same random stream
    start = rng.randint(0, new_w - MAX_HW)
    crop = img_t[:, 0:MAX_HW, start:start + MAX_HW]
    dens = ndimage.gaussian_filter(dens[0:MAX_HW, start:start + MAX_HW], sigma=(1, 1), order=0) * 60
    boxes, scaled = exemplar_crops(img_t, rects, sh, sw)
    pos = torch.tensor([[y1, max(0, x1 - start), y2, min(MAX_HW, x2 - start)] for (y1, x1, y2, x2) in scaled])
    return {"image": crop.contiguous(), "boxes": boxes, "pos": pos, "gt_density": torch.from_numpy(dens), "m_flag": 0}


def transform_val(image, rects, dots):
    """ResizeValImage.__call__ (util/FSC147.py:316-366): 384x384, gaussian sigma 4 radius 7, x60."""
    from PIL import Image
    from scipy import ndimage
    W, H = image.size
    sh, sw = float(MAX_HW) / H, float(MAX_HW) / W
    img_t = to_tensor(image.resize((MAX_HW, MAX_HW), Image.BILINEAR))
    dens = ndimage.gaussian_filter(dot_map(dots, MAX_HW, MAX_HW, sh, sw), sigma=4, radius=7, order=0)
    boxes, scaled = exemplar_crops(img_t, rects, sh, sw)
    return {"image": img_t, "boxes": boxes, "pos": torch.tensor(scaled), "gt_density": torch.from_numpy(dens) * 60, "m_flag": 0}


def random_resized_crop_params(w, h, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), rng=random):
    """torchvision RandomResizedCrop.get_params."""
    area = h * w
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        ar = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < cw <= w and 0 < ch <= h:
            return rng.randint(0, h - ch), rng.randint(0, w - cw), ch, cw
    in_ratio = float(w) / float(h)
    if in_ratio < min(ratio):
        cw, ch = w, int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        ch, cw = h, int(round(h * max(ratio)))
    else:
        cw, ch = w, h
    return (h - ch) // 2, (w - cw) // 2, ch, cw


def transform_pretrain(image, rng=random):
    """ResizePreTrainImage + PreTrainNormalize (util/FSC147.py:58-83, 369-374): resize to multiples of 16, RandomResizedCrop(384,
    scale 0.2-1, bicubic), horizontal flip p = 0.5, ToTensor.  Only the image is used by FSC_pretrain.py:143."""
    from PIL import Image
    W, H = image.size
    image = image.resize((16 * int(W / 16), 16 * int(H / 16)), Image.BILINEAR)
    i, j, ch, cw = random_resized_crop_params(image.size[0], image.size[1], rng=rng)
    image = image.crop((j, i, j + cw, i + ch)).resize((MAX_HW, MAX_HW), Image.BICUBIC)
    if rng.random() < 0.5:
        image = image.transpose(Image.FLIP_LEFT_RIGHT)
    return to_tensor(image)


def _paths(args):
    j = lambda p: p if os.path.isabs(p) else os.path.join(args.data_path, p)
    return j(args.anno_file), j(args.data_split_file), j(args.im_dir)


def _open_rgb(path):
    from PIL import Image
    image = Image.open(path)
    image.load()
    return image.convert("RGB") if image.mode != "RGB" else image


class TrainData(Dataset):
    """FSC_finetune_cross.py:113-154 -> (image, gt_density, n_dots, boxes, pos, m_flag, im_id)."""
    _warned = False

    def __init__(self, args, split="train", do_aug=True):
        anno, split_file, self.im_dir = _paths(args)
        self.annotations = json.load(open(anno))
        self.img = list(json.load(open(split_file))[split])
        random.shuffle(self.img)
        self.split = split
        if do_aug and split == "train" and not TrainData._warned:
            warnings.warn("FSC147 augmentations (imgaug / cv2 / torchvision) are not available: using the reference's "
                          "non-augmented train transform")
            TrainData._warned = True

    def __len__(self):
        return len(self.img)

    def __getitem__(self, idx):
        im_id = self.img[idx]
        anno = self.annotations[im_id]
        dots = np.array(anno["points"])
        rects = [[b[0][1], b[0][0], b[2][1], b[2][0]] for b in anno["box_examples_coordinates"]]
        image = _open_rgb(os.path.join(self.im_dir, im_id))
        s = transform_train_noaug(image, rects, dots) if self.split == "train" else transform_val(image, rects, dots)
        return s["image"], s["gt_density"], len(dots), s["boxes"], s["pos"], s["m_flag"], im_id


class PretrainData(Dataset):
    """FSC_pretrain.py:114-143 -> image [3, 384, 384] (the density file it also opens is never used downstream)."""

    def __init__(self, args):
        _, split_file, self.im_dir = _paths(args)
        self.img = list(json.load(open(split_file))["train"])
        random.shuffle(self.img)

    def __len__(self):
        return len(self.img)

    def __getitem__(self, idx):
        return transform_pretrain(_open_rgb(os.path.join(self.im_dir, self.img[idx])))


def available(args):
    """True when the annotation, split and image directory named by the CLI flags exist."""
    return all(os.path.exists(p) for p in _paths(args))
