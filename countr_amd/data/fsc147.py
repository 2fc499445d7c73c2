"""FSC147 datasets with PIL + scipy only: the reference's loaders need torchvision / cv2 / imgaug,
none of which exist offline.  Restated here: the train transform of util/FSC147.py with and without augmentation, the val transform
and the pretrain transform, and the dataset classes of FSC_finetune_cross.py:113-154 / FSC_pretrain.py:114-143.  Host-side code: it
produces the [B,3,384,384] / [B,3,3,64,64] / [B,384,384] batches that FinetuneStep.load() / PretrainStep.load() stage into the engine.

Augmentation branch (util/FSC147.py:130-253).  What is pinned against the reference and what is not:
  * the control flow, the order of draws from the `random` module, the mosaic (crop sizes, self / other-image branches, class test,
    dot placement, the three blending passes) and the random crop are the reference's own arithmetic and are checked against golden
    vectors produced by running util/FSC147.py itself (tools/oracle/make_golden_data.py: the mosaic output does not depend on the
    third-party ops, which only feed the branch the mosaic discards);
  * Gaussian noise is np.random.normal(0, 0.1) exactly as in the reference;
  * ColorJitter / GaussianBlur (torchvision 0.14.1) and the random affine (imgaug 0.4) are re-implemented from their documented
    semantics with this module's own parameter draws (same distributions, not the same random streams): "parity unpinned" for
    those three ops -- neither library is installed, so no reference output exists to compare with.

PIL's Image.resize(BILINEAR / BICUBIC) is what torchvision.transforms.Resize calls for PIL inputs; exemplar crops are resized
as TENSORS by the reference (torchvision 0.14.1: bilinear, no antialias) == F.interpolate(mode="bilinear", align_corners=False)."""
import json
import math
import os
import random

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


# ------------------------------------------------------------------------------------------------------------------
# Train-time augmentation (util/FSC147.py:117-300 with aug_flag True)
# ------------------------------------------------------------------------------------------------------------------
def _gray(img):
    return (0.2989 * img[0] + 0.587 * img[1] + 0.114 * img[2]).unsqueeze(0)


def _blend(a, b, f):
    return (f * a + (1.0 - f) * b).clamp(0, 1)


def _rgb2hsv(img):
    r, g, b = img[0], img[1], img[2]
    maxc, minc = img.max(0).values, img.min(0).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc))


def _hsv2rgb(img):
    h, s, v = img[0], img[1], img[2]
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = (v * (1.0 - s)).clamp(0, 1)
    q = (v * (1.0 - s * f)).clamp(0, 1)
    t = (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
    sel = [(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)]
    out = torch.zeros_like(img)
    for k, (rr, gg, bb) in enumerate(sel):
        m = i == k
        out[0] += m * rr
        out[1] += m * gg
        out[2] += m * bb
    return out


def color_jitter(img, order, brightness, contrast, saturation, hue):
    """transforms.ColorJitter on a [3, H, W] tensor in [0, 1] (torchvision 0.14.1 functional_tensor semantics): the four ops in the
    drawn `order` (0 brightness, 1 contrast, 2 saturation, 3 hue); factor 1 (hue 0) is the identity."""
    for op in order:
        if op == 0:
            img = _blend(img, torch.zeros_like(img), brightness)
        elif op == 1:
            img = _blend(img, _gray(img).mean(), contrast)
        elif op == 2:
            img = _blend(img, _gray(img), saturation)
        else:
            hsv = _rgb2hsv(img)
            hsv[0] = torch.remainder(hsv[0] + hue, 1.0)
            img = _hsv2rgb(hsv)
    return img


def gaussian_blur(img, kernel_size=(7, 9), sigma=1.0):
    """transforms.GaussianBlur on a [C, H, W] tensor: separable kernel (kx, ky) of one sigma, reflect padding."""
    def k1d(n):
        x = torch.linspace(-(n - 1) * 0.5, (n - 1) * 0.5, n, dtype=img.dtype)
        k = torch.exp(-0.5 * (x / sigma) ** 2)
        return k / k.sum()
    kx, ky = k1d(kernel_size[0]), k1d(kernel_size[1])
    k2 = (ky[:, None] * kx[None, :]).expand(img.shape[0], 1, kernel_size[1], kernel_size[0])
    pad = (kernel_size[0] // 2, kernel_size[0] // 2, kernel_size[1] // 2, kernel_size[1] // 2)
    x = F.pad(img.unsqueeze(0), pad, mode="reflect")
    return F.conv2d(x, k2.contiguous(), groups=img.shape[0])[0]


def affine_matrix(h, w, rotate_deg, scale, shear_deg, tx_frac, ty_frac):
    """Forward 3x3 matrix (x, y, 1) of iaa.Affine(rotate, scale, shear, translate_percent) about the image centre."""
    cx, cy = w / 2.0 - 0.5, h / 2.0 - 0.5
    r, sh = math.radians(rotate_deg), math.radians(shear_deg)
    to0 = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    sc = np.diag([scale, scale, 1.0])
    shm = np.array([[1, -math.tan(sh), 0], [0, 1, 0], [0, 0, 1.0]])
    rot = np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1.0]])
    back = np.array([[1, 0, cx + tx_frac * w], [0, 1, cy + ty_frac * h], [0, 0, 1.0]])
    return back @ rot @ shm @ sc @ to0


def warp_affine(img, M):
    """Bilinear warp of a [C, H, W] tensor by the forward matrix M, zero fill (imgaug order=1, cval=0, mode='constant')."""
    from scipy import ndimage
    Mi = np.linalg.inv(M)
    # ndimage works in (row, col) = (y, x): swap the axes of the inverse map
    A = np.array([[Mi[1, 1], Mi[1, 0]], [Mi[0, 1], Mi[0, 0]]])
    off = np.array([Mi[1, 2], Mi[0, 2]])
    a = img.numpy()
    out = np.stack([ndimage.affine_transform(a[c], A, offset=off, order=1, mode="constant", cval=0.0) for c in range(a.shape[0])])
    return torch.from_numpy(out)


def _scaled_dot(d, sh, sw, new_h, new_w):
    return min(new_h - 1, int(d[1] * sh)), min(new_w - 1, int(d[0] * sw))   # (y, x) as util/FSC147.py:147,188


def _resize_t(t, size):
    return F.interpolate(t.unsqueeze(0), size=(size, size), mode="bilinear", align_corners=False)[0]


def _mosaic_piece(img_t, dots, sh, sw, new_h, new_w, length, start_w, start_h, resize_l, count_dots):
    """One quadrant source: a length x length crop of the CLEAN resized image, resized to resize_l, with its dot map
    (util/FSC147.py:186-195, 222-233)."""
    piece = _resize_t(img_t[:, start_h:start_h + length, start_w:start_w + length], resize_l)
    dm = np.zeros((resize_l, resize_l), dtype="float32")
    if count_dots:
        for d in dots:
            y, x = _scaled_dot(d, sh, sw, new_h, new_w)
            if start_h <= y < start_h + length and start_w <= x < start_w + length:
                dm[min(resize_l - 1, int((y - start_h) * resize_l / length))][min(resize_l - 1, int((x - start_w) * resize_l / length))] = 1
    return piece, torch.from_numpy(dm)


def _blend_pair(a, b, bl, resize_l, dim):
    """Join two pieces along `dim` (1 = rows, 2 = columns of a [C, H, W] tensor) with the reference's 2*bl-wide linear cross-fade
    (util/FSC147.py:235-253): the kept 192-wide cores are concatenated, then the bl lines either side of the seam are re-mixed
    with the neighbour's overhanging lines."""
    sl = lambda t, lo, hi: t.narrow(dim, lo, hi - lo)
    out = torch.cat((sl(a, bl, resize_l - bl), sl(b, bl, resize_l - bl)), dim)
    i = torch.arange(bl)
    shape = [1, 1, 1]
    shape[dim] = bl
    w_new = ((bl - i).to(out.dtype) / (2 * bl)).view(shape)
    w_old = ((i + bl).to(out.dtype) / (2 * bl)).view(shape)
    hi = out.index_select(dim, 192 + i) * w_old + a.index_select(dim, resize_l - 1 - bl + i) * w_new
    lo = out.index_select(dim, 191 - i) * w_old + b.index_select(dim, bl - i) * w_new
    out.index_copy_(dim, 192 + i, hi)
    out.index_copy_(dim, 191 - i, lo)
    return out.clamp(0, 1)


def mosaic(img_t, dots, sh, sw, im_id, ctx, rng):
    """Random self / cross-image mosaic (util/FSC147.py:177-253) -> image [3, 384, 384], dot map [384, 384], m_flag."""
    new_h, new_w = img_t.shape[1], img_t.shape[2]
    bl = rng.randint(10, 20)
    resize_l = 192 + 2 * bl
    pieces, maps, m_flag = [], [], 0
    if dots.shape[0] >= 70:
        for _ in range(4):
            length = rng.randint(150, 384)
            start_w = rng.randint(0, new_w - length)
            start_h = rng.randint(0, new_h - length)
            p, m = _mosaic_piece(img_t, dots, sh, sw, new_h, new_w, length, start_w, start_h, resize_l, True)
            pieces.append(p); maps.append(m)
    else:
        m_flag = 1
        prob = rng.random()
        gt_pos = rng.randint(0, 3) if prob > 0.25 else rng.randint(0, 4)   # 5 %: none of the four quadrants is the image itself
        for i in range(4):
            if i == gt_pos:
                t_id, t_img, t_dots, t_sh, t_sw = im_id, img_t, dots, sh, sw
            else:
                t_id = ctx.train_set[rng.randint(0, len(ctx.train_set) - 1)]
                t_dots = np.array(ctx.annotations[t_id]["points"])
                timage = ctx.open_image(t_id)
                th, tw = flex_resize(timage.size[1], timage.size[0])
                t_sw, t_sh = float(tw) / timage.size[0], float(th) / timage.size[1]
                from PIL import Image
                t_img = to_tensor(timage.resize((tw, th), Image.BILINEAR))
            th, tw = t_img.shape[1], t_img.shape[2]
            length = rng.randint(250, 384)
            start_w = rng.randint(0, tw - length)
            start_h = rng.randint(0, th - length)
            same = ctx.class_dict[im_id] == ctx.class_dict[t_id]
            p, m = _mosaic_piece(t_img, t_dots, t_sh, t_sw, th, tw, length, start_w, start_h, resize_l, same)
            pieces.append(p); maps.append(m)
    core = lambda m, d: m.narrow(d, bl, resize_l - 2 * bl)
    left = _blend_pair(pieces[0], pieces[1], bl, resize_l, 1)
    right = _blend_pair(pieces[2], pieces[3], bl, resize_l, 1)
    image = _blend_pair(left, right, bl, resize_l, 2)
    dl = torch.cat((core(maps[0], 0), core(maps[1], 0)), 0)
    dr = torch.cat((core(maps[2], 0), core(maps[3], 0)), 0)
    dens = torch.cat((core(dl, 1), core(dr, 1)), 1)
    return image, dens, m_flag


class AugParams:
    """The draws of the three third-party augmentations (distributions of util/FSC147.py:139,153-160,371-374)."""

    def __init__(self, nprng=np.random):
        self.order = [int(k) for k in nprng.permutation(4)]
        self.brightness = float(nprng.uniform(0.75, 1.25))
        self.contrast = float(nprng.uniform(0.85, 1.15))
        self.saturation = float(nprng.uniform(0.85, 1.15))
        self.hue = float(nprng.uniform(-0.15, 0.15))
        self.sigma = float(nprng.uniform(0.1, 2.0))
        self.rotate = float(nprng.uniform(-15, 15))
        self.scale = float(nprng.uniform(0.8, 1.2))
        self.shear = float(nprng.uniform(-10, 10))
        self.tx = float(nprng.uniform(-0.2, 0.2))
        self.ty = float(nprng.uniform(-0.2, 0.2))


def transform_train_aug(image, rects, dots, im_id, ctx, rng=random, nprng=np.random, params=None):
    """ResizeTrainImage.__call__ with aug_flag True (util/FSC147.py:117-300).  `rng` supplies the draws the reference takes from the
    `random` module (same order), `nprng` the Gaussian noise and -- unless `params` is given -- the jitter / blur / affine draws."""
    from PIL import Image
    from scipy import ndimage
    W, H = image.size
    new_h, new_w = flex_resize(H, W)
    sh, sw = float(new_h) / H, float(new_w) / W
    img_t = to_tensor(image.resize((new_w, new_h), Image.BILINEAR))
    mosaic_flag = rng.random() < 0.25
    m_flag = 0
    # noise -> colour jitter -> blur -> affine (image and dots) -> flip: always computed, as in the reference, although the mosaic
    # branch below does not use the result
    noise = torch.from_numpy(nprng.normal(0, 0.1, tuple(img_t.shape)))
    aug = (img_t + noise).clamp(0, 1).float()
    pr = params or AugParams(nprng)
    aug = gaussian_blur(color_jitter(aug, pr.order, pr.brightness, pr.contrast, pr.saturation, pr.hue), (7, 9), pr.sigma)
    M = affine_matrix(new_h, new_w, pr.rotate, pr.scale, pr.shear, pr.tx, pr.ty)
    aug = warp_affine(aug, M)
    dens = np.zeros((new_h, new_w), dtype="float32")
    for d in dots:
        y, x = _scaled_dot(d, sh, sw, new_h, new_w)
        xa, ya, _ = M @ np.array([x, y, 1.0])
        if 0 <= xa < new_w and 0 <= ya < new_h:       # KeypointsOnImage: dropped when it leaves the image
            dens[int(ya)][int(xa)] = 1
    dens = torch.from_numpy(dens)
    if rng.random() > 0.5:
        aug, dens = aug.flip(-1), dens.flip(-1)
    if mosaic_flag:
        out_img, out_dens, m_flag = mosaic(img_t, dots, sh, sw, im_id, ctx, rng)
    else:
        start_w = rng.randint(0, new_w - 1 - 383)
        start_h = rng.randint(0, new_h - 1 - 383)
        out_img = aug[:, start_h:start_h + 384, start_w:start_w + 384]
        out_dens = dens[start_h:start_h + 384, start_w:start_w + 384]
    out_dens = torch.from_numpy(ndimage.gaussian_filter(out_dens.numpy(), sigma=(1, 1), order=0) * 60)
    boxes, _ = exemplar_crops(img_t, rects, sh, sw)     # exemplars come from the CLEAN resized image (util/FSC147.py:283)
    return {"image": out_img.contiguous().float(), "boxes": boxes, "pos": torch.tensor([]), "gt_density": out_dens, "m_flag": m_flag}


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


# ---------------------------------------------------------------------------------------------------------------
# test-time loader: TestData of FSC_test_cross(few-shot).py:82-190 (pinned by tests/golden/data_test.npz, generated by exec'ing that class)
# ---------------------------------------------------------------------------------------------------------------
def _test_resize(image):
    """height 384, width 16 * int(W / H * 384 / 16) (:147-154) -> (tensor CHW in [0, 1], scale_h, scale_w, new_h, new_w)"""
    from PIL import Image
    W, H = image.size
    new_h, new_w = 384, 16 * int((W / H * 384) / 16)
    sw, sh = float(new_w) / W, float(new_h) / H
    return to_tensor(image.resize((new_w, new_h), Image.BILINEAR)), sh, sw, new_h, new_w


def _test_crops(img_t, bboxes, sh, sw):
    """:160-174 -- corner 0 / corner 2 of every box scaled with int() truncation, inclusive crop, 64x64 bilinear"""
    rects = [[int(b[0][1] * sh), int(b[0][0] * sw), int(b[2][1] * sh), int(b[2][0] * sw)] for b in bboxes]
    return [resize_box(img_t[:, y1:y2 + 1, x1:x2 + 1]) for y1, x1, y2, x2 in rects], rects


def external_exemplars(annotations, split_ids, im_dir, box_bound=-1):
    """TestData.__init__ with external=True (:96-129): the exemplar crops of EVERY image of the split, in annotation-file order, as one
    list -- cut to the first `box_bound` crops when box_bound >= 0 -- used as the exemplars of every test image."""
    ids = set(split_ids)
    crops = []
    for im_id in annotations:
        if im_id not in ids:
            continue
        bboxes = annotations[im_id]["box_examples_coordinates"]
        if bboxes:
            img_t, sh, sw, _, _ = _test_resize(_open_rgba_only(os.path.join(im_dir, im_id)))
            crops += _test_crops(img_t, bboxes, sh, sw)[0]
    if box_bound >= 0:
        crops = crops[:box_bound]
    return torch.stack(crops) if crops else torch.zeros(0)


def _open_rgba_only(path):
    """Image.open + the reference's mode rule (:98-100, :142-144): only RGBA is converted"""
    from PIL import Image
    image = Image.open(path)
    if image.mode == "RGBA":
        image = image.convert("RGB")
    image.load()
    return image


def test_item(annotations, im_dir, im_id, box_bound=-1, external_boxes=None):
    """TestData.__getitem__ (:134-190) -> (image [3, 384, W'], dots [n, 2], boxes [k, 3, 64, 64] (or empty), pos [(y1, x1, y2, x2)],
    gt_map [384, W']).  With external exemplars pos is empty (no 3x3 path, no test-time normalisation: there is no box in THIS image)."""
    from scipy import ndimage
    anno = annotations[im_id]
    bboxes = anno["box_examples_coordinates"] if box_bound < 0 else anno["box_examples_coordinates"][:box_bound]
    dots = np.array(anno["points"])
    img_t, sh, sw, new_h, new_w = _test_resize(_open_rgba_only(os.path.join(im_dir, im_id)))
    if external_boxes is not None:
        boxes, pos = external_boxes, []
    else:
        crops, pos = _test_crops(img_t, bboxes, sh, sw)
        boxes = torch.stack(crops) if crops else torch.zeros(0)
    if box_bound >= 0:
        assert len(boxes) <= box_bound
    gt = ndimage.gaussian_filter(dot_map(dots, new_h, new_w, sh, sw), sigma=(1, 1), order=0)
    return img_t, dots, boxes, pos, torch.from_numpy(gt) * 60


def _paths(args):
    j = lambda p: p if os.path.isabs(p) else os.path.join(args.data_path, p)
    return j(args.anno_file), j(args.data_split_file), j(args.im_dir)


def _open_rgb(path):
    from PIL import Image
    image = Image.open(path)
    image.load()
    return image.convert("RGB") if image.mode != "RGB" else image


class TrainData(Dataset):
    """FSC_finetune_cross.py:113-154 -> (image, gt_density, n_dots, boxes, pos, m_flag, im_id).  With do_aug (the reference's
    default) the train split goes through transform_train_aug; it needs args.class_file (ImageClasses_FSC147.txt) for the
    cross-image mosaic, as util/FSC147.py:35-41 does."""

    def __init__(self, args, split="train", do_aug=True):
        anno, split_file, self.im_dir = _paths(args)
        self.annotations = json.load(open(anno))
        splits = json.load(open(split_file))
        self.img = list(splits[split])
        random.shuffle(self.img)
        self.split = split
        self.train_set = list(splits.get("train", []))
        self.do_aug = bool(do_aug) and split == "train"
        self.class_dict = {}
        if self.do_aug:
            cf = getattr(args, "class_file", None)
            cf = cf if cf is None or os.path.isabs(cf) else os.path.join(args.data_path, cf)
            if not cf or not os.path.exists(cf):
                raise FileNotFoundError("--do_aug needs --class_file (ImageClasses_FSC147.txt): %r not found" % (cf,))
            with open(cf) as f:
                for line in f:
                    parts = line.split()
                    if parts:
                        self.class_dict[parts[0]] = parts[1:]

    def open_image(self, im_id):
        return _open_rgb(os.path.join(self.im_dir, im_id))

    def _worker_nprng(self):
        """numpy generator of THIS loader process for the augmentation draws.  DataLoader seeds `random` and torch per worker but
        not numpy: with the global np.random every worker would draw the same noise / jitter / affine sequence (the reference's
        np.random.normal has exactly that flaw; its torchvision / imgaug draws do not)."""
        pid = os.getpid()
        if getattr(self, "_nprng_pid", None) != pid:
            self._nprng_pid = pid
            self._nprng = np.random.RandomState(torch.initial_seed() % (2 ** 32))
        return self._nprng

    def __len__(self):
        return len(self.img)

    def __getitem__(self, idx):
        im_id = self.img[idx]
        anno = self.annotations[im_id]
        dots = np.array(anno["points"])
        rects = [[b[0][1], b[0][0], b[2][1], b[2][0]] for b in anno["box_examples_coordinates"]]
        image = self.open_image(im_id)
        if self.split != "train":
            s = transform_val(image, rects, dots)
        elif self.do_aug:
            s = transform_train_aug(image, rects, dots, im_id, self, nprng=self._worker_nprng())
        else:
            s = transform_train_noaug(image, rects, dots)
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
