"""Drop-in SupervisedMAE (reference models_mae_cross.py:18-253) executed by the MI355X HIP engine.

Same constructor, factories, state_dict keys and forward(imgs, boxes, shot_num) -> [B, H, W] as the reference.
The nn.Modules below are parameter containers; forward() packs the parameters into the engine's flat
buffers (views, no copies afterwards) and runs hand-written HIP kernels through libcountr_hip.so.  There
is no CPU fallback: calling forward without the built library or without a GPU raises.

Extra (non-reference) constructor keyword: precision = "bf16" (default, the throughput mode: bf16 operands, fp32 accumulation,
statistics and residual stream), "fp16" (the reference-numerics mode: the reference's own autocast dtype, FSC_finetune_cross.py:273-275,286
-- same kernels built with fp16 operands, 8x smaller operand rounding, same MFMA rate) or "fp32" (exact-f32 MFMA parity mode).
"""
from functools import partial

import torch
import torch.nn as nn

from .models_crossvit import Block, CrossAttentionBlock, PatchEmbed
from .util.pos_embed import get_2d_sincos_pos_embed
from .engine import Engine, is_trainable
from ._module import HipModule


class _DecoderFn(torch.autograd.Function):
    """One autograd node for the whole decoder side (the encoder is frozen, models_mae_cross.py:204-205)."""

    @staticmethod
    def forward(ctx, model, imgs, boxes, shot_num, *params):
        eng = model._engine()
        out = eng.forward(imgs, boxes, shot_num, train=True)
        ctx.model, ctx.B, ctx.S, ctx.nparams = model, imgs.shape[0], int(shot_num), len(params)
        ctx.gen = eng.plan(ctx.B, ctx.S, True).fwd_gen
        return out.clone()

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        eng = model._engine()
        # the engine keeps ONE set of activation buffers per (batch, shot_num): a second train-mode forward of the same shape
        # overwrites what this backward needs (the reference module has no such limit) -- refuse instead of returning wrong gradients
        if eng.plan(ctx.B, ctx.S, True).fwd_gen != ctx.gen:
            raise RuntimeError("SupervisedMAE (HIP engine): backward() of a forward whose activations were overwritten by a later "
                               "train-mode forward with the same (batch, shot_num); run backward before the next forward "
                               "(gradient accumulation across forwards is supported by countr_amd.trainer.FinetuneStep)")
        eng.backward(ctx.B, ctx.S, dout.contiguous().float())
        grads = []
        unused = ("decoder_proj",) if ctx.S == 0 else ("shot_token",)
        for name in model._train_names:
            if name.startswith(unused):
                grads.append(None)  # not on the autograd path for this shot_num, as in the reference
            else:
                grads.append(eng.gview(name).clone())
        return (None, None, None, None) + tuple(grads)


class SupervisedMAE(HipModule):
    def __init__(self, img_size=384, patch_size=16, in_chans=3,
                 embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=2, decoder_num_heads=16,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False, precision="bf16"):
        super().__init__()
        assert in_chans == 3 and decoder_embed_dim == 512 and mlp_ratio == 4, "kernels are specialised for the CounTR shapes"
        self.cfg = (patch_size, embed_dim, depth, num_heads, decoder_embed_dim, decoder_depth, decoder_num_heads)
        self.img_size = img_size
        self.precision = precision
        # --- MAE encoder (models_mae_cross.py:25-36)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer, precision=precision)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        # --- decoder (models_mae_cross.py:38-100)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches, decoder_embed_dim), requires_grad=False)
        self.shot_token = nn.Parameter(torch.zeros(512))

        def proj(ci, co, last=False):
            return nn.Sequential(nn.Conv2d(ci, co, kernel_size=3, stride=1, padding=1), nn.InstanceNorm2d(co),
                                 nn.ReLU(inplace=True), nn.AdaptiveAvgPool2d((1, 1)) if last else nn.MaxPool2d(2))
        self.decoder_proj1 = proj(3, 64)
        self.decoder_proj2 = proj(64, 128)
        self.decoder_proj3 = proj(128, 256)
        self.decoder_proj4 = proj(256, decoder_embed_dim, last=True)
        self.decoder_blocks = nn.ModuleList([
            CrossAttentionBlock(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer, precision=precision)
            for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)

        def head(ci, final=False):
            layers = [nn.Conv2d(ci, 256, kernel_size=3, stride=1, padding=1), nn.GroupNorm(8, 256), nn.ReLU(inplace=True)]
            if final:
                layers.append(nn.Conv2d(256, 1, kernel_size=1, stride=1))
            return nn.Sequential(*layers)
        self.decode_head0 = head(decoder_embed_dim)
        self.decode_head1 = head(256)
        self.decode_head2 = head(256)
        self.decode_head3 = head(256, final=True)
        self.norm_pix_loss = norm_pix_loss
        self.initialize_weights()

    # ------------------------------------------------------------------ init (models_mae_cross.py:108-134)
    def initialize_weights(self):
        g = int(self.patch_embed.num_patches ** .5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.pos_embed.shape[-1], g)).float().unsqueeze(0))
        self.decoder_pos_embed.data.copy_(
            torch.from_numpy(get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], g)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.shot_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ engine plumbing (see _module.HipModule)
    def _make_engine(self, shapes, device):
        return Engine(self.cfg, shapes, device, precision=self.precision, img_size=self.img_size, ln_eps=self.norm.eps)

    def _is_trainable(self, name):
        return is_trainable(name)

    # ------------------------------------------------------------------ reference surface
    def forward_encoder(self, x):
        """models_mae_cross.py:136-148 -> latent [B, N, embed_dim] (fp32 copy)."""
        eng = self._engine()
        B = x.shape[0]
        p = eng.plan(B, 0, False)
        p.buf["img"].copy_(x)
        eng.run(p.fwd[:p.enc_ops])
        return p.buf["latent"].float().view(B, -1, self.cfg[1]).clone()

    def forward_decoder(self, x, y_, shot_num=3):
        """models_mae_cross.py:150-199 on a given latent [B, N, embed_dim] (inference path; gradients flow through
        forward(), whose single autograd node covers the whole decoder side)."""
        eng = self._engine()
        B, S = x.shape[0], int(shot_num)
        p = eng.plan(B, S, False)
        p.buf["latent"].copy_(x.reshape(p.buf["latent"].shape))
        if S > 0:
            p.buf["boxes"].view(B, S, 3, 64, 64).copy_(y_[:, :S].float())
        eng.run(p.fwd[p.enc_ops:])
        return p.buf["out"].clone()

    def forward(self, imgs, boxes, shot_num):
        """models_mae_cross.py:201-207."""
        assert imgs.shape[-2] == self.img_size and imgs.shape[-1] == self.img_size, \
            "Input image size (%d*%d) doesn't match model (%d*%d)." % (imgs.shape[-2], imgs.shape[-1], self.img_size, self.img_size)
        shot_num = int(shot_num)
        imgs = imgs.float()
        if shot_num > 0:
            boxes = boxes.float()
            assert boxes.dim() == 5 and boxes.shape[1] >= shot_num, "boxes must be [B, >=shot_num, 3, 64, 64]"
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._train_params) if self._eng else \
            torch.is_grad_enabled()
        eng = self._engine()
        if needs_grad and any(p.requires_grad for p in self._train_params):
            return _DecoderFn.apply(self, imgs, boxes, shot_num, *self._train_params)
        return eng.forward(imgs, boxes, shot_num, train=False).clone()


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return SupervisedMAE(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=2,
                         decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return SupervisedMAE(patch_size=16, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512, decoder_depth=2,
                         decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_huge_patch14_dec512d8b(**kwargs):
    return SupervisedMAE(patch_size=14, embed_dim=1280, depth=32, num_heads=16, decoder_embed_dim=512, decoder_depth=2,
                         decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_base_patch16_fim4(**kwargs):
    return SupervisedMAE(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=4,
                         decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_base_patch16_fim6(**kwargs):
    return SupervisedMAE(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=6,
                         decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


# recommended archs (models_mae_cross.py:248-253)
mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b
mae_vit_base4_patch16 = mae_vit_base_patch16_fim4
mae_vit_base6_patch16 = mae_vit_base_patch16_fim6
mae_vit_large_patch16 = mae_vit_large_patch16_dec512d8b
mae_vit_huge_patch14 = mae_vit_huge_patch14_dec512d8b
