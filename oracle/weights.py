"""Deterministic synthetic weights for the CounTR SupervisedMAE (test infrastructure).

No checkpoint or dataset is available offline, so parity is pinned on generated weights: each
tensor is drawn from its own numpy RandomState seeded by crc32(name) ^ seed, scaled like the
reference initialisation (models_mae_cross.py:108-134; torch defaults for Conv2d/GroupNorm) but with
non-zero biases and perturbed norm gains so that every parameter influences the output.
State-dict keys/shapes follow the reference schema.
"""
import zlib
from collections import OrderedDict

import numpy as np

CONFIGS = {
    # name: (patch, embed_dim, depth, heads, dec_dim, dec_depth, dec_heads)   models_mae_cross.py:210-253
    "mae_vit_base_patch16": (16, 768, 12, 12, 512, 2, 16),
    "mae_vit_base4_patch16": (16, 768, 12, 12, 512, 4, 16),
    "mae_vit_base6_patch16": (16, 768, 12, 12, 512, 6, 16),
    "mae_vit_large_patch16": (16, 1024, 24, 16, 512, 2, 16),
    "mae_vit_huge_patch14": (14, 1280, 32, 16, 512, 2, 16),
    # reduced-depth configuration used for fast CPU tests (not a reference factory)
    "tiny_test": (16, 768, 2, 12, 512, 1, 16),
    # the odd shapes of mae_vit_huge_patch14 at a size a test can afford: patch 14 on 384 pixels (27 x 27 = 729 tokens, a 432 x 432 map),
    # head_dim 320 / 4 = 80 (not a reference factory; pinned by tools/oracle/make_golden_patch14.py through the reference's own class)
    "tiny_patch14": (14, 320, 2, 4, 512, 1, 16),
}


def sincos_1d(dim, pos):
    """util/pos_embed.py:49-67 (float64 table)."""
    omega = np.arange(dim // 2, dtype=np.float64) / (dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(dim, grid):
    """util/pos_embed.py:20-46: first half encodes the column (w) index, second half the row."""
    gh = np.arange(grid, dtype=np.float32)
    gw = np.arange(grid, dtype=np.float32)
    g = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid, grid)  # g[0] = w index
    return np.concatenate([sincos_1d(dim // 2, g[0]), sincos_1d(dim // 2, g[1])], axis=1)


def schema(model="mae_vit_base_patch16", img_size=384):
    """Ordered (name, shape, kind) list in the reference's state_dict order."""
    p, D, depth, H, Dd, ddepth, Hd = CONFIGS[model]
    grid = img_size // p
    N = grid * grid
    S = []
    S.append(("pos_embed", (1, N, D), "pos"))
    S.append(("decoder_pos_embed", (1, N, Dd), "pos"))
    S.append(("shot_token", (512,), "token"))
    S.append(("patch_embed.proj.weight", (D, 3, p, p), "patch_w"))
    S.append(("patch_embed.proj.bias", (D,), "bias"))

    def lin(prefix, out_f, in_f):
        S.append((prefix + ".weight", (out_f, in_f), "linear_w"))
        S.append((prefix + ".bias", (out_f,), "bias"))

    def norm(prefix, d):
        S.append((prefix + ".weight", (d,), "norm_w"))
        S.append((prefix + ".bias", (d,), "norm_b"))

    def conv(prefix, co, ci, k):
        S.append((prefix + ".weight", (co, ci, k, k), "conv_w"))
        S.append((prefix + ".bias", (co,), "conv_b:%d" % (ci * k * k)))

    for i in range(depth):
        b = "blocks.%d" % i
        norm(b + ".norm1", D)
        lin(b + ".attn.qkv", 3 * D, D)
        lin(b + ".attn.proj", D, D)
        norm(b + ".norm2", D)
        lin(b + ".mlp.fc1", 4 * D, D)
        lin(b + ".mlp.fc2", D, 4 * D)
    norm("norm", D)
    lin("decoder_embed", Dd, D)
    conv("decoder_proj1.0", 64, 3, 3)
    conv("decoder_proj2.0", 128, 64, 3)
    conv("decoder_proj3.0", 256, 128, 3)
    conv("decoder_proj4.0", Dd, 256, 3)
    for i in range(ddepth):
        b = "decoder_blocks.%d" % i
        norm(b + ".norm0", Dd)
        lin(b + ".selfattn.qkv", 3 * Dd, Dd)
        lin(b + ".selfattn.proj", Dd, Dd)
        norm(b + ".norm1", Dd)
        lin(b + ".attn.wq", Dd, Dd)
        lin(b + ".attn.wk", Dd, Dd)
        lin(b + ".attn.wv", Dd, Dd)
        lin(b + ".attn.proj", Dd, Dd)
        norm(b + ".norm2", Dd)
        lin(b + ".mlp.fc1", 4 * Dd, Dd)
        lin(b + ".mlp.fc2", Dd, 4 * Dd)
    norm("decoder_norm", Dd)
    conv("decode_head0.0", 256, Dd, 3)
    norm("decode_head0.1", 256)
    conv("decode_head1.0", 256, 256, 3)
    norm("decode_head1.1", 256)
    conv("decode_head2.0", 256, 256, 3)
    norm("decode_head2.1", 256)
    conv("decode_head3.0", 256, 256, 3)
    norm("decode_head3.1", 256)
    conv("decode_head3.3", 1, 256, 1)
    return S


def make_state_dict(model="mae_vit_base_patch16", seed=0, img_size=384):
    """OrderedDict name -> float32 numpy array."""
    p = CONFIGS[model][0]
    grid = img_size // p
    sd = OrderedDict()
    for name, shape, kind in schema(model, img_size):
        rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if kind == "pos":
            a = sincos_2d(shape[-1], grid)[None]
        elif kind == "token":
            a = rs.normal(0.0, 0.02, size=shape)
        elif kind in ("linear_w", "patch_w"):
            fan_out = shape[0]
            fan_in = int(np.prod(shape[1:]))
            bound = np.sqrt(6.0 / (fan_in + fan_out))  # xavier uniform (models_mae_cross.py:118-119,129)
            a = rs.uniform(-bound, bound, size=shape)
        elif kind == "bias":
            a = rs.uniform(-0.02, 0.02, size=shape)
        elif kind == "norm_w":
            a = 1.0 + rs.uniform(-0.1, 0.1, size=shape)
        elif kind == "norm_b":
            a = rs.uniform(-0.05, 0.05, size=shape)
        elif kind == "conv_w":
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / np.sqrt(fan_in)  # kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
            a = rs.uniform(-bound, bound, size=shape)
        elif kind.startswith("conv_b"):
            bound = 1.0 / np.sqrt(int(kind.split(":")[1]))
            a = rs.uniform(-bound, bound, size=shape)
        else:
            raise ValueError(kind)
        sd[name] = np.ascontiguousarray(a, dtype=np.float32)
    return sd


def make_inputs(batch, shots=3, seed=0, img_size=384):
    """Synthetic inputs of SURVEY.md section 8d: imgs, boxes ~ U[0,1); gaussian-dot density x60; loss mask."""
    rs = np.random.RandomState(1000 + seed)
    imgs = rs.uniform(0, 1, size=(batch, 3, img_size, img_size)).astype(np.float32)
    boxes = rs.uniform(0, 1, size=(batch, shots, 3, 64, 64)).astype(np.float32)
    gt = np.zeros((batch, img_size, img_size), dtype=np.float32)
    from scipy.ndimage import gaussian_filter
    for b in range(batch):
        k = rs.randint(5, 201)
        ys = rs.randint(0, img_size, size=k)
        xs = rs.randint(0, img_size, size=k)
        np.add.at(gt[b], (ys, xs), 1.0)
        gt[b] = gaussian_filter(gt[b], sigma=(1, 1), order=0) * 60.0  # util/FSC147.py:275-278
    mask = rs.binomial(1, 0.8, size=(img_size, img_size)).astype(np.float32)  # FSC_finetune_cross.py:290
    return imgs, boxes, gt, mask


def make_wide_inputs(seed, width, shots=3, height=384):
    """Test-time inputs (FSC_test_cross(few-shot).py:134-190 shapes): one image [1, 3, 384, width] (height 384, width a multiple
    of 16), `shots` exemplar crops [1, shots, 3, 64, 64] and their rectangles (y1, x1, y2, x2) inside the image."""
    rs = np.random.RandomState(5000 + seed)
    img = rs.uniform(0, 1, size=(1, 3, height, width)).astype(np.float32)
    boxes = rs.uniform(0, 1, size=(1, shots, 3, 64, 64)).astype(np.float32)
    pos = []
    for _ in range(shots):
        y1, x1 = int(rs.randint(0, height - 80)), int(rs.randint(0, width - 80))
        pos.append((y1, x1, y1 + int(rs.randint(20, 70)), x1 + int(rs.randint(20, 70))))
    return img, boxes, pos


DATA_CASES = [(640, 384), (384, 600), (500, 400), (900, 384), (300, 250)]   # (W, H) of the synthetic dataset items


AUG_ITEMS = DATA_CASES + [(700, 500)]     # the augmentation fixtures add one item with 80 dots (self-mosaic branch: >= 70)
AUG_CLASSES = ["apples", "birds", "apples", "birds", "apples", "apples"]


def make_fsc_item(k, w, h, ndots=25):
    """A synthetic FSC147 item (util/FSC147.py sample layout): RGB PIL image, three exemplar rectangles [y1, x1, y2, x2] in
    original pixel coordinates and dot annotations [x, y] -- deterministic from k."""
    from PIL import Image
    rs = np.random.RandomState(9000 + k)
    yy, xx = np.mgrid[0:h, 0:w]
    arr = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx * 3 + yy * 5) % 256)], -1).astype(np.int32)
    arr = np.clip(arr + rs.randint(-20, 21, size=arr.shape), 0, 255).astype(np.uint8)
    dots = np.stack([rs.uniform(2, w - 2, ndots), rs.uniform(2, h - 2, ndots)], 1)
    rects = []
    for _ in range(3):
        x1, y1 = int(rs.uniform(0, w - 80)), int(rs.uniform(0, h - 80))
        rects.append([y1, x1, y1 + int(rs.uniform(20, 70)), x1 + int(rs.uniform(20, 70))])
    return Image.fromarray(arr), rects, dots


def write_aug_dataset(root):
    """The six-image dataset of the augmentation fixtures on disk (lossless PNG), in the FSC147 file layout the loaders read:
    returns (annotation file, split file, class file, image dir, ids)."""
    import json
    import os
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    anno, ids = {}, []
    for k, (w, h) in enumerate(AUG_ITEMS):
        image, rects, dots = make_fsc_item(k, w, h, ndots=80 if k == 5 else 25)
        im_id = "%d.png" % k
        image.save(os.path.join(root, "images", im_id))
        anno[im_id] = {"points": dots.tolist(),
                       "box_examples_coordinates": [[[r[1], r[0]], [r[1], r[2]], [r[3], r[2]], [r[3], r[0]]] for r in rects]}
        ids.append(im_id)
    json.dump(anno, open(os.path.join(root, "anno.json"), "w"))
    json.dump({"train": ids, "val": ids[:2], "test": ids[:2]}, open(os.path.join(root, "split.json"), "w"))
    with open(os.path.join(root, "classes.txt"), "w") as f:
        for im_id, c in zip(ids, AUG_CLASSES):
            f.write("%s\t%s\n" % (im_id, c))
    return os.path.join(root, "anno.json"), os.path.join(root, "split.json"), os.path.join(root, "classes.txt"), os.path.join(root, "images"), ids


# ---------------------------------------------------------------------------------------------------------------
# MAE pretraining model (reference models_mae_noct.py:11-235): schema + deterministic weights
# ---------------------------------------------------------------------------------------------------------------
MAE_CONFIGS = {
    # name: (patch, embed_dim, depth, heads, dec_dim, dec_depth, dec_heads)     models_mae_noct.py:207-235
    "mae_vit_base_patch16": (16, 768, 12, 12, 512, 8, 16),
    "mae_vit_large_patch16": (16, 1024, 24, 16, 512, 8, 16),
    "tiny_test": (16, 768, 2, 12, 512, 2, 16),
}


def schema_mae(model="mae_vit_base_patch16", img_size=384):
    p, D, depth, H, Dd, ddepth, Hd = MAE_CONFIGS[model]
    N = (img_size // p) ** 2
    S = [("pos_embed", (1, N, D), "pos"), ("mask_token", (1, 1, Dd), "token"), ("decoder_pos_embed", (1, N, Dd), "pos"),
         ("patch_embed.proj.weight", (D, 3, p, p), "patch_w"), ("patch_embed.proj.bias", (D,), "bias")]

    def lin(prefix, out_f, in_f):
        S.append((prefix + ".weight", (out_f, in_f), "linear_w"))
        S.append((prefix + ".bias", (out_f,), "bias"))

    def norm(prefix, d):
        S.append((prefix + ".weight", (d,), "norm_w"))
        S.append((prefix + ".bias", (d,), "norm_b"))

    def blocks(prefix, n, d):
        for i in range(n):
            b = "%s.%d" % (prefix, i)
            norm(b + ".norm1", d)
            lin(b + ".attn.qkv", 3 * d, d)
            lin(b + ".attn.proj", d, d)
            norm(b + ".norm2", d)
            lin(b + ".mlp.fc1", 4 * d, d)
            lin(b + ".mlp.fc2", d, 4 * d)
    blocks("blocks", depth, D)
    norm("norm", D)
    lin("decoder_embed", Dd, D)
    blocks("decoder_blocks", ddepth, Dd)
    norm("decoder_norm", Dd)
    lin("decoder_pred", p * p * 3, Dd)
    return S


def make_state_dict_mae(model="mae_vit_base_patch16", seed=0, img_size=384):
    grid = img_size // MAE_CONFIGS[model][0]
    sd = OrderedDict()
    for name, shape, kind in schema_mae(model, img_size):
        rs = np.random.RandomState((zlib.crc32(("mae/" + name).encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if kind == "pos":
            a = sincos_2d(shape[-1], grid)[None]
        elif kind == "token":
            a = rs.normal(0.0, 0.02, size=shape)
        elif kind in ("linear_w", "patch_w"):
            bound = np.sqrt(6.0 / (int(np.prod(shape[1:])) + shape[0]))
            a = rs.uniform(-bound, bound, size=shape)
        elif kind == "bias":
            a = rs.uniform(-0.02, 0.02, size=shape)
        elif kind == "norm_w":
            a = 1.0 + rs.uniform(-0.1, 0.1, size=shape)
        else:
            a = rs.uniform(-0.05, 0.05, size=shape)
        sd[name] = np.ascontiguousarray(a, dtype=np.float32)
    return sd


def make_mae_inputs(batch, seed=0, img_size=384, patch=16, mask_ratio=0.5):
    """imgs ~ U[0,1) and a per-sample random permutation (ids_shuffle) standing in for argsort(rand) of
    models_mae_noct.py:110-135; returns imgs, ids_shuffle, ids_restore, len_keep."""
    rs = np.random.RandomState(2000 + seed)
    imgs = rs.uniform(0, 1, size=(batch, 3, img_size, img_size)).astype(np.float32)
    L = (img_size // patch) ** 2
    ids_shuffle = np.stack([rs.permutation(L) for _ in range(batch)]).astype(np.int64)
    ids_restore = np.argsort(ids_shuffle, axis=1).astype(np.int64)
    return imgs, ids_shuffle, ids_restore, int(L * (1 - mask_ratio))
