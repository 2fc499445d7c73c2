#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (runs only in the build container).

Imports /root/reference (read-only) through the three shims of SURVEY.md section 8c / Appendix A:
  1. np.float alias (util/pos_embed.py:56 uses the removed numpy alias),
  2. stub torchvision / torchvision.utils modules (models_mae_cross.py:11 import is unused),
  3. timm.models.vision_transformer stand-ins {PatchEmbed, Block} assembled from the reference's own
     in-tree duplicates models_crossvit.{Attention, Mlp, DropPath} (timm 0.4.9 is not installed).
It loads the deterministic weights of oracle/weights.py into the reference model (strict=True), runs
it on the synthetic inputs of oracle/weights.make_inputs, and writes small fixtures to tests/golden/.
Nothing from the reference is copied into the repo: only inputs' seeds and output tensors are stored.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    if not hasattr(np, "float"):
        np.float = float
    tv = types.ModuleType("torchvision")
    tv.utils = types.ModuleType("torchvision.utils")
    sys.modules.update({"torchvision": tv, "torchvision.utils": tv.utils})
    import models_crossvit as mc

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            img_size = mc.to_2tuple(img_size)
            patch_size = mc.to_2tuple(patch_size)
            self.img_size, self.patch_size = img_size, patch_size
            self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            B, C, H, W = x.shape
            assert H == self.img_size[0] and W == self.img_size[1]
            return self.proj(x).flatten(2).transpose(1, 2)

    class Block(nn.Module):
        def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                     drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
            super().__init__()
            self.norm1 = norm_layer(dim)
            self.attn = mc.Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                     attn_drop=attn_drop, proj_drop=drop)
            self.drop_path = mc.DropPath(drop_path) if drop_path > 0. else nn.Identity()
            self.norm2 = norm_layer(dim)
            self.mlp = mc.Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

        def forward(self, x):
            x = x + self.drop_path(self.attn(self.norm1(x)))
            return x + self.drop_path(self.mlp(self.norm2(x)))

    timm = types.ModuleType("timm")
    timm.__version__ = "0.4.9"
    timm.models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.PatchEmbed, vt.Block = PatchEmbed, Block
    timm.models.vision_transformer = vt
    sys.modules.update({"timm": timm, "timm.models": timm.models, "timm.models.vision_transformer": vt})
    import models_mae_cross
    import util.lr_sched as lr_sched
    return models_mae_cross, lr_sched


def sha(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def main():
    from oracle import weights as W
    mm, lr_sched = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    meta = {}

    model_name = "mae_vit_base_patch16"
    model = mm.__dict__[model_name](norm_pix_loss=False)
    model.eval()
    ref_keys = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    mine = [(n, list(s)) for n, s, _ in W.schema(model_name)]
    assert ref_keys == mine, "state_dict schema mismatch"
    meta["schema"] = ref_keys
    meta["n_params"] = int(sum(p.numel() for p in model.parameters()))

    # other factories: schema only
    for other in ("mae_vit_base4_patch16", "mae_vit_base6_patch16", "mae_vit_large_patch16"):
        m2 = mm.__dict__[other](norm_pix_loss=False)
        assert [(k, list(v.shape)) for k, v in m2.state_dict().items()] == [(n, list(s)) for n, s, _ in W.schema(other)]
        del m2

    sd = W.make_state_dict(model_name, seed=0)
    meta["weights_sha256_seed0"] = sha(sd)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)

    # pos-embed rows straight from the reference function
    from util.pos_embed import get_2d_sincos_pos_embed
    pe768 = get_2d_sincos_pos_embed(768, 24)
    pe512 = get_2d_sincos_pos_embed(512, 24)
    np.savez_compressed(os.path.join(OUT, "pos_embed_rows.npz"), rows=np.array([0, 1, 24, 575]),
                        pe768=pe768[[0, 1, 24, 575]], pe512=pe512[[0, 1, 24, 575]])

    # ---------------- forward cases
    imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=0)
    cases = {
        "b2_s3": (imgs, boxes, 3),
        "b1_s0": (imgs[:1], boxes[:1], 0),
        "b1_s1": (imgs[1:2], boxes[1:2], 1),
        "b1_s2": (imgs[:1], boxes[:1], 2),
        "b1_zero_empty": (imgs[1:2], np.zeros((1, 0), np.float32), 0),
    }
    probes = {}

    def hook(name):
        def f(mod, inp, out):
            probes[name] = out.detach().numpy().copy()
        return f
    hs = [model.blocks[0].register_forward_hook(hook("enc_block0")),
          model.blocks[11].register_forward_hook(hook("enc_block11")),
          model.norm.register_forward_hook(hook("latent")),
          model.decoder_blocks[0].register_forward_hook(hook("dec_block0")),
          model.decoder_blocks[1].register_forward_hook(hook("dec_block1")),
          model.decoder_norm.register_forward_hook(hook("dec_norm")),
          model.decode_head0.register_forward_hook(hook("head0_pre_up")),
          model.decode_head2.register_forward_hook(hook("head2_pre_up"))]
    fw = {}
    with torch.no_grad():
        for name, (im, bx, s) in cases.items():
            out = model(torch.from_numpy(im), torch.from_numpy(bx), s)
            fw[name] = out.numpy().astype(np.float32)
            meta["count_" + name] = [float(x) for x in (out.reshape(out.shape[0], -1).sum(1) / 60)]
            if name == "b2_s3":
                pr = {k: v for k, v in probes.items()}
    for h in hs:
        h.remove()
    np.savez_compressed(os.path.join(OUT, "forward.npz"), **fw)
    # probes: statistics + a slice of each intermediate (full tensors would be tens of MB)
    ps = {}
    for k, v in pr.items():
        ps[k + "_mean"] = np.float64(v.astype(np.float64).mean())
        ps[k + "_l2"] = np.float64(np.sqrt((v.astype(np.float64) ** 2).sum()))
        ps[k + "_head"] = v.reshape(v.shape[0], -1)[:, :256].copy()
        ps[k + "_shape"] = np.array(v.shape)
    np.savez_compressed(os.path.join(OUT, "probes_b2_s3.npz"), **ps)

    # ---------------- gradients of the finetune loss (FSC_finetune_cross.py:290-295) for B=2, S=3 and S=0
    gr = {}
    for tag, s in (("s3", 3), ("s0", 0)):
        model.zero_grad()
        out = model(torch.from_numpy(imgs), torch.from_numpy(boxes), s)
        loss = (out - torch.from_numpy(gt)) ** 2
        loss = (loss * torch.from_numpy(mask) / (384 * 384)).sum() / out.shape[0]
        loss.backward()
        gr["loss_" + tag] = np.float64(loss.item())
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.numpy()
            gr["%s/norm/%s" % (tag, k)] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            if g.size <= 4096:
                gr["%s/full/%s" % (tag, k)] = g.copy()
            else:
                gr["%s/head/%s" % (tag, k)] = g.reshape(-1)[:512].copy()
        meta["grad_tensors_" + tag] = sorted(k for k, p in model.named_parameters() if p.grad is not None)
    np.savez_compressed(os.path.join(OUT, "grads_b2.npz"), **gr)

    # ---------------- sliding-window stitch (FSC_test_cross(few-shot).py:322-351) on synthetic wide images.  The loop is inline in
    # the reference's main(), so it cannot be imported; it is not restated here either: the lines are read from /root/reference at
    # run time and executed unchanged.  The window starts are recovered from the storage offset of each slice the loop feeds the model.
    import textwrap
    with open(os.path.join(REF, "FSC_test_cross(few-shot).py")) as f:
        stitch_src = textwrap.dedent("".join(f.readlines()[321:351]))
    st = {}
    rs = np.random.RandomState(77)
    for width in (672, 512, 384):
        wide = rs.uniform(0, 1, size=(1, 3, 384, width)).astype(np.float32)
        bx = torch.from_numpy(boxes[:1])
        samples = torch.from_numpy(wide)
        starts = []

        class Rec(nn.Module):
            def forward(self, x, b, n):
                starts.append(x.storage_offset())
                return model(x, b, n)
        ns = {"torch": torch, "nn": nn, "model": Rec(), "device": torch.device("cpu"), "samples": samples, "boxes": bx, "num_boxes": 3,
              "h": 384, "w": width}
        exec(stitch_src, ns)
        density_map = ns["density_map"]
        st["starts_%d" % width] = np.array(starts)
        st["count_%d" % width] = np.float64(density_map.sum().item() / 60)
        st["colsum_%d" % width] = density_map.sum(0).numpy()
        st["seed"] = np.array(77)
    np.savez_compressed(os.path.join(OUT, "stitch.npz"), **st)

    # ---------------- lr schedule table (util/lr_sched.py:9-21)
    class A:
        lr, min_lr, warmup_epochs, epochs = 1e-5, 0.0, 10, 1000

    class Opt:
        param_groups = [{"lr": 0.0}]
    tab = [(e, lr_sched.adjust_learning_rate(Opt, e, A)) for e in (0, 0.5, 5, 9.99, 10, 10.5, 100, 505, 999, 999.9)]
    meta["lr_table"] = tab

    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
