"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the CounTR SupervisedMAE hot path.

A functional (state-dict in, tensors out) re-implementation on torch-CPU tensors of
    /root/reference/models_mae_cross.py:136-207  (forward_encoder / forward_decoder / forward)
    /root/reference/models_crossvit.py:46-156    (Mlp, Attention, CrossAttention, CrossAttentionBlock)
    timm==0.4.9 PatchEmbed / Block (not vendored; semantics from the call sites models_mae_cross.py:27-34)
    /root/reference/FSC_finetune_cross.py:290-303 (masked MSE loss, counts)
Norms, GELU, attention, pooling and the bilinear x2 upsampling are written out explicitly; only the
3x3 convolutions and matmuls use torch primitives.  Pinned against the reference itself by
tools/oracle/make_golden.py (fixtures in tests/golden/).  Never imported by countr_amd/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .weights import CONFIGS


def _t(sd, name, dtype):
    v = sd[name]
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(v)
    return v.to(dtype)


class Params:
    """Typed view of a state dict; optionally makes decoder-side tensors autograd leaves."""

    def __init__(self, sd, dtype=torch.float32, requires_grad=False):
        self.t = {}
        for k in sd:
            v = _t(sd, k, dtype).clone()
            if requires_grad and is_trainable_decoder_param(k):
                v.requires_grad_(True)
            self.t[k] = v

    def __getitem__(self, k):
        return self.t[k]


def is_trainable_decoder_param(name):
    """Parameters that receive gradients in finetuning: the encoder runs under no_grad
    (models_mae_cross.py:204-205) and both pos-embeds are requires_grad=False (:30,42)."""
    if name in ("pos_embed", "decoder_pos_embed"):
        return False
    return name.startswith(("decoder_", "decode_head", "shot_token"))


# ---------------------------------------------------------------------------------------------
# elementary ops, written out
# ---------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps=1e-6):
    """nn.LayerNorm(eps=1e-6) (models_mae_cross.py:214 partial)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)  # biased
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu(x):
    """nn.GELU default = exact erf form (models_crossvit.py:49)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x, w, b):
    return x @ w.t() + b


def self_attention(x, p, prefix, heads):
    """models_crossvit.py:82-94 (== timm Attention)."""
    B, N, C = x.shape
    dh = C // heads
    qkv = linear(x, p[prefix + ".qkv.weight"], p[prefix + ".qkv.bias"]).reshape(B, N, 3, heads, dh)
    q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
    s = (q @ k.transpose(-2, -1)) * dh ** -0.5
    s = s - s.max(-1, keepdim=True).values
    e = torch.exp(s)
    a = e / e.sum(-1, keepdim=True)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return linear(o, p[prefix + ".proj.weight"], p[prefix + ".proj.bias"])


def cross_attention(x, y, p, prefix, heads):
    """models_crossvit.py:111-128."""
    B, Nx, C = x.shape
    Ny = y.shape[1]
    dh = C // heads
    q = linear(x, p[prefix + ".wq.weight"], p[prefix + ".wq.bias"]).reshape(B, Nx, heads, dh).transpose(1, 2)
    k = linear(y, p[prefix + ".wk.weight"], p[prefix + ".wk.bias"]).reshape(B, Ny, heads, dh).transpose(1, 2)
    v = linear(y, p[prefix + ".wv.weight"], p[prefix + ".wv.bias"]).reshape(B, Ny, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-2, -1)) * dh ** -0.5
    s = s - s.max(-1, keepdim=True).values
    e = torch.exp(s)
    a = e / e.sum(-1, keepdim=True)
    o = (a @ v).transpose(1, 2).reshape(B, Nx, C)
    return linear(o, p[prefix + ".proj.weight"], p[prefix + ".proj.bias"])


def mlp(x, p, prefix):
    """models_crossvit.py:60-67."""
    h = gelu(linear(x, p[prefix + ".fc1.weight"], p[prefix + ".fc1.bias"]))
    return linear(h, p[prefix + ".fc2.weight"], p[prefix + ".fc2.bias"])


def patch_embed(imgs, w, b, patch):
    """timm PatchEmbed: Conv2d(k=p, s=p) -> flatten(2).transpose(1, 2); token t = i*grid + j,
    feature order (c, py, px)."""
    B, Cc, H, W = imgs.shape
    gh, gw = H // patch, W // patch
    x = imgs[:, :, : gh * patch, : gw * patch].reshape(B, Cc, gh, patch, gw, patch)
    x = x.permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, Cc * patch * patch)
    return x @ w.reshape(w.shape[0], -1).t() + b


def instance_norm_relu(x, eps=1e-5):
    """nn.InstanceNorm2d defaults (affine=False, biased var, eps 1e-5) + ReLU (models_mae_cross.py:49-50)."""
    mu = x.mean((2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean((2, 3), keepdim=True)
    return torch.relu((x - mu) / torch.sqrt(var + eps))


def max_pool2(x):
    """nn.MaxPool2d(2) (models_mae_cross.py:51).  torch's own op is used so that the backward routes the
    gradient to the first maximum of a window exactly like the reference (ties matter for bf16 inputs)."""
    return F.max_pool2d(x, 2)


def group_norm_relu(x, w, b, groups=8, eps=1e-5):
    """nn.GroupNorm(8, 256) + ReLU (models_mae_cross.py:82-83)."""
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, (C // groups) * H * W)
    mu = xg.mean(-1, keepdim=True)
    var = ((xg - mu) ** 2).mean(-1, keepdim=True)
    xn = ((xg - mu) / torch.sqrt(var + eps)).reshape(B, C, H, W)
    return torch.relu(xn * w.view(1, C, 1, 1) + b.view(1, C, 1, 1))


def _up2_1d(x, dim):
    """Bilinear x2, align_corners=False along `dim`: out[2m] = .25 x[m-1] + .75 x[m],
    out[2m+1] = .75 x[m] + .25 x[m+1], indices clamped at the borders (F.interpolate semantics)."""
    n = x.shape[dim]
    idx = torch.arange(n)
    prev = x.index_select(dim, (idx - 1).clamp(min=0))
    nxt = x.index_select(dim, (idx + 1).clamp(max=n - 1))
    even = 0.25 * prev + 0.75 * x
    odd = 0.75 * x + 0.25 * nxt
    out = torch.stack([even, odd], dim=dim + 1)
    shape = list(x.shape)
    shape[dim] = 2 * n
    return out.reshape(shape)


def upsample2x(x):
    """F.interpolate(size=2x, mode='bilinear', align_corners=False) (models_mae_cross.py:189-196)."""
    return _up2_1d(_up2_1d(x, 2), 3)


# ---------------------------------------------------------------------------------------------
# model
# ---------------------------------------------------------------------------------------------
def forward_encoder(p, imgs, cfg, probes=None):
    """models_mae_cross.py:136-148."""
    patch, D, depth, H = cfg[0], cfg[1], cfg[2], cfg[3]
    x = patch_embed(imgs, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], patch)
    x = x + p["pos_embed"]
    if probes is not None:
        probes["patch_pos"] = x
    for i in range(depth):
        b = "blocks.%d" % i
        x = x + self_attention(layer_norm(x, p[b + ".norm1.weight"], p[b + ".norm1.bias"]), p, b + ".attn", H)
        x = x + mlp(layer_norm(x, p[b + ".norm2.weight"], p[b + ".norm2.bias"]), p, b + ".mlp")
        if probes is not None and i in (0, depth - 1):
            probes["enc_block%d" % i] = x
    x = layer_norm(x, p["norm.weight"], p["norm.bias"])
    if probes is not None:
        probes["latent"] = x
    return x


def exemplar_tokens(p, boxes, shot_num, batch):
    """models_mae_cross.py:157-177: per-shot CNN -> y [B, S, 512]; shot_num == 0 -> shot_token."""
    if shot_num == 0:
        return p["shot_token"].reshape(1, 1, -1).expand(batch, 1, -1)
    ys = []
    for s in range(shot_num):
        y = boxes[:, s]
        for li, pool in ((1, True), (2, True), (3, True), (4, False)):
            y = F.conv2d(y, p["decoder_proj%d.0.weight" % li], p["decoder_proj%d.0.bias" % li], padding=1)
            y = instance_norm_relu(y)
            y = max_pool2(y) if pool else y.mean((2, 3), keepdim=True)
        ys.append(y.reshape(y.shape[0], -1))
    return torch.stack(ys, dim=1)


def forward_decoder(p, latent, boxes, shot_num, cfg, probes=None, y=None):
    """models_mae_cross.py:150-199.  `y` (test hook): precomputed exemplar tokens [B, S, 512]."""
    Dd, ddepth, Hd = cfg[4], cfg[5], cfg[6]
    x = linear(latent, p["decoder_embed.weight"], p["decoder_embed.bias"]) + p["decoder_pos_embed"]
    if y is None:
        y = exemplar_tokens(p, boxes, shot_num, latent.shape[0])
    if probes is not None:
        probes["dec_embed"] = x
        probes["exemplar_tokens"] = y
    for i in range(ddepth):
        b = "decoder_blocks.%d" % i
        x = x + self_attention(layer_norm(x, p[b + ".norm0.weight"], p[b + ".norm0.bias"]), p, b + ".selfattn", Hd)
        x = x + cross_attention(layer_norm(x, p[b + ".norm1.weight"], p[b + ".norm1.bias"]), y, p, b + ".attn", Hd)
        x = x + mlp(layer_norm(x, p[b + ".norm2.weight"], p[b + ".norm2.bias"]), p, b + ".mlp")
        if probes is not None:
            probes["dec_block%d" % i] = x
    x = layer_norm(x, p["decoder_norm.weight"], p["decoder_norm.bias"])
    if probes is not None:
        probes["dec_norm"] = x
    n, hw, c = x.shape
    g = int(math.sqrt(hw))
    x = x.transpose(1, 2).reshape(n, c, g, g)
    for i in range(4):
        h = "decode_head%d" % i
        x = F.conv2d(x, p[h + ".0.weight"], p[h + ".0.bias"], padding=1)
        x = group_norm_relu(x, p[h + ".1.weight"], p[h + ".1.bias"])
        if i == 3:
            x = F.conv2d(x, p[h + ".3.weight"], p[h + ".3.bias"])
        x = upsample2x(x)
        if probes is not None:
            probes["head%d" % i] = x
    return x.squeeze(-3)


def forward(sd_or_params, imgs, boxes, shot_num, model="mae_vit_base_patch16", dtype=torch.float32,
            probes=None):
    """SupervisedMAE.forward (models_mae_cross.py:201-207) -> [B, 384, 384]."""
    cfg = CONFIGS[model]
    p = sd_or_params if isinstance(sd_or_params, Params) else Params(sd_or_params, dtype)
    imgs = torch.as_tensor(imgs).to(dtype)
    boxes = torch.as_tensor(boxes).to(dtype)
    with torch.no_grad():
        latent = forward_encoder(p, imgs, cfg, probes)
    return forward_decoder(p, latent, boxes, shot_num, cfg, probes)


def masked_mse_loss(out, gt, mask):
    """FSC_finetune_cross.py:290-295: sum((out-gt)^2 * mask / 384^2) / B, mask shared over the batch."""
    hw = out.shape[-1] * out.shape[-2]
    return (((out - gt) ** 2) * mask / hw).sum() / out.shape[0]


def counts(density):
    """FSC_finetune_cross.py:299: per-image sum / 60."""
    return density.reshape(density.shape[0], -1).sum(1) / 60.0


def loss_and_grads(sd, imgs, boxes, gt, mask, shot_num, model="mae_vit_base_patch16", dtype=torch.float32):
    """One finetune-style backward: returns (out, loss, {name: grad}) for the decoder-side params."""
    p = Params(sd, dtype, requires_grad=True)
    out = forward(p, imgs, boxes, shot_num, model, dtype)
    loss = masked_mse_loss(out, torch.as_tensor(gt).to(dtype), torch.as_tensor(mask).to(dtype))
    names = [k for k in p.t if p.t[k].requires_grad]
    grads = torch.autograd.grad(loss, [p.t[k] for k in names], allow_unused=True)
    return out.detach(), loss.detach(), {k: g for k, g in zip(names, grads)}


def adamw_step(param, grad, m, v, step, lr, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.05):
    """torch.optim.AdamW update (decoupled weight decay), as used at FSC_finetune_cross.py:235."""
    param = param * (1.0 - lr * wd)
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    mhat = m / (1 - beta1 ** step)
    vhat = v / (1 - beta2 ** step)
    param = param - lr * mhat / (vhat.sqrt() + eps)
    return param, m, v


def adjust_learning_rate(epoch, lr, min_lr, warmup_epochs, epochs):
    """util/lr_sched.py:9-21."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def stitch_windows(window_fn, width, height=384):
    """Sliding-window stitch of FSC_test_cross(few-shot).py:322-351 / demo_zero.py:41-74.
    window_fn(start) -> [384(height), 384] density of the window starting at column `start`."""
    dm = torch.zeros(height, width)
    start, prev = 0, -1
    while start + 383 < width:
        out = window_fn(start)
        new = torch.zeros(height, width)
        ov = prev - start + 1  # columns already covered
        if ov > 0:
            new[:, start:prev + 1] = dm[:, start:prev + 1] / 2 + out[:, :ov] / 2
        new[:, :start] = dm[:, :start]
        new[:, prev + 1:start + 384] = out[:, max(ov, 0):]
        dm = new
        prev = start + 383
        start += 128
        if start + 383 >= width:
            if start == width - 384 + 128:
                break
            start = width - 384
    return dm
