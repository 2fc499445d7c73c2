"""Drop-in MaskedAutoencoderViTNoCT (reference models_mae_noct.py:11-235) executed by the MI355X HIP engine.

Same constructor, factories, state_dict keys and forward(imgs, mask_ratio) -> (loss, pred, mask) as the reference; the
nn.Modules are parameter containers and all math runs in libcountr_hip.so (countr_amd/mae_engine.py).  GPU only, no CPU
fallback.  Extra keyword: precision = "bf16" (default) | "fp16" | "fp32".
"""
from functools import partial

import torch
import torch.nn as nn

from ._module import HipModule
from .mae_engine import MaeEngine, mae_trainable
from .models_crossvit import Block, PatchEmbed
from .util.pos_embed import get_2d_sincos_pos_embed


class _MaeFn(torch.autograd.Function):
    """One autograd node for the whole model: forward + loss in forward(), full backward in backward()."""

    @staticmethod
    def forward(ctx, model, imgs, ids_shuffle, len_keep, *params):
        eng = model._engine()
        loss, pred, mask = eng.forward(imgs, ids_shuffle, len_keep, train=True, norm_pix=model.norm_pix_loss)
        ctx.model, ctx.B, ctx.K = model, imgs.shape[0], len_keep
        mask = mask.clone()
        ctx.mark_non_differentiable(mask)
        return loss[0].clone(), pred.clone(), mask

    @staticmethod
    def backward(ctx, dloss, dpred, dmask):
        model = ctx.model
        eng = model._engine()
        eng.backward(ctx.B, ctx.K)
        # the backward pass is linear in dloss: scale on the device, no host sync (dpred: gradients through the returned
        # prediction are not part of the reference's training path -- loss.backward() only)
        grads = tuple(eng.gview(n) * dloss for n in model._train_names)
        return (None, None, None, None) + grads


class MaskedAutoencoderViTNoCT(HipModule):
    def __init__(self, img_size=384, patch_size=16, in_chans=3,
                 embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False, precision="bf16"):
        super().__init__()
        assert in_chans == 3 and mlp_ratio == 4, "kernels are specialised for 3 input channels and mlp_ratio 4"
        self.cfg = (patch_size, embed_dim, depth, num_heads, decoder_embed_dim, decoder_depth, decoder_num_heads)
        self.img_size = img_size
        self.precision = precision
        # --- encoder (models_mae_noct.py:20-31)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer, precision=precision)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        # --- decoder (models_mae_noct.py:33-47)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer, precision=precision)
                                             for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        self.initialize_weights()

    # ------------------------------------------------------------------ init (models_mae_noct.py:52-80)
    def initialize_weights(self):
        g = int(self.patch_embed.num_patches ** .5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.pos_embed.shape[-1], g)).float().unsqueeze(0))
        self.decoder_pos_embed.data.copy_(
            torch.from_numpy(get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], g)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.mask_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _make_engine(self, shapes, device):
        return MaeEngine(self.cfg, shapes, device, precision=self.precision, img_size=self.img_size, ln_eps=self.norm.eps)

    def _is_trainable(self, name):
        return mae_trainable(name)

    # ------------------------------------------------------------------ reference surface
    def patchify(self, imgs):
        """models_mae_noct.py:82-94 (index shuffle only; the training path patchifies inside countr_patch_mse)."""
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        h = w = imgs.shape[2] // p
        return imgs.reshape(imgs.shape[0], 3, h, p, w, p).permute(0, 2, 4, 3, 5, 1).reshape(imgs.shape[0], h * w, p * p * 3)

    def unpatchify(self, x):
        """models_mae_noct.py:96-108."""
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** .5)
        assert h * w == x.shape[1]
        return x.reshape(x.shape[0], h, w, p, p, 3).permute(0, 5, 1, 3, 2, 4).reshape(x.shape[0], 3, h * p, h * p)

    def len_keep(self, mask_ratio):
        return int(self.patch_embed.num_patches * (1 - mask_ratio))   # models_mae_noct.py:117

    def draw_masking(self, batch, device):
        """Per-sample argsort of uniform noise (models_mae_noct.py:119-121)."""
        noise = torch.rand(batch, self.patch_embed.num_patches, device=device)
        return torch.argsort(noise, dim=1)

    def random_masking(self, x, mask_ratio, ids_shuffle=None):
        """models_mae_noct.py:110-135 on a given token tensor x [N, L, D] -> (x_masked, mask, ids_restore)."""
        from . import _lib
        eng = self._engine()
        N, L, D = x.shape
        K = int(L * (1 - mask_ratio))
        ids_shuffle = self.draw_masking(N, x.device) if ids_shuffle is None else ids_shuffle.to(x.device)
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        src = (ids_shuffle[:, :K] + torch.arange(N, device=x.device).unsqueeze(1) * L).reshape(-1).to(torch.int32)
        xs = x.float().contiguous()
        out = torch.empty(N, K, D, device=x.device, dtype=torch.float32)
        _lib.check(eng.L.countr_gather_rows(xs.data_ptr(), src.data_ptr(), out.data_ptr(), None, None, 0, N * K, D, 0, 0, eng._stream()),
                   "gather_rows")
        mask = (ids_restore >= K).float()
        return out, mask, ids_restore

    def forward_encoder(self, x, mask_ratio, ids_shuffle=None):
        """models_mae_noct.py:137-157 -> (latent [N, len_keep, D] fp32 copy, mask, ids_restore)."""
        eng = self._engine()
        B, K = x.shape[0], self.len_keep(mask_ratio)
        ids_shuffle = self.draw_masking(B, x.device) if ids_shuffle is None else ids_shuffle
        p = eng.plan(B, K, False)
        p.buf["img"].copy_(x.float())
        ids_restore = eng.set_masking(p, ids_shuffle)
        eng.run(p.fwd[:p.enc_ops])
        return p.buf["latent"].float().view(B, K, -1).clone(), p.buf["mask"].clone(), ids_restore

    def forward_decoder(self, x, ids_restore):
        """models_mae_noct.py:159-179 on a latent [N, len_keep, D] -> pred [N, L, p*p*3]."""
        eng = self._engine()
        B, K = x.shape[0], x.shape[1]
        p = eng.plan(B, K, False)
        p.buf["latent"].copy_(x.reshape(p.buf["latent"].shape))
        eng.set_masking(p, torch.argsort(ids_restore.to(x.device), dim=1))
        eng.run(p.fwd[p.enc_ops:])
        return p.buf["pred"].view(B, eng.N, -1).clone()

    def forward_loss(self, imgs, pred, mask):
        """models_mae_noct.py:181-198 (mean over ALL patches; mask is unused by the reference as well)."""
        from . import _lib
        eng = self._engine()
        B = imgs.shape[0]
        im = imgs.float().contiguous()
        pr = pred.float().contiguous()
        loss = torch.empty(1, device=im.device)
        ws = torch.empty(eng.L.countr_patch_mse_workspace_floats(B, eng.img, eng.img, eng.patch), device=im.device)
        _lib.check(eng.L.countr_patch_mse(pr.data_ptr(), im.data_ptr(), None, loss.data_ptr(), ws.data_ptr(), B, eng.img, eng.img,
                                          eng.patch, int(bool(self.norm_pix_loss)), 1.0, 0, eng._stream()), "patch_mse")
        return loss[0]

    def forward(self, imgs, mask_ratio=0.75, ids_shuffle=None):
        """models_mae_noct.py:200-204.  ids_shuffle (optional, [N, L]) replaces the internally drawn permutation."""
        assert imgs.shape[-2] == self.img_size and imgs.shape[-1] == self.img_size, \
            "Input image size (%d*%d) doesn't match model (%d*%d)." % (imgs.shape[-2], imgs.shape[-1], self.img_size, self.img_size)
        imgs = imgs.float()
        eng = self._engine()
        K = self.len_keep(mask_ratio)
        ids_shuffle = self.draw_masking(imgs.shape[0], imgs.device) if ids_shuffle is None else ids_shuffle
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._train_params):
            return _MaeFn.apply(self, imgs, ids_shuffle, K, *self._train_params)
        loss, pred, mask = eng.forward(imgs, ids_shuffle, K, train=False, norm_pix=self.norm_pix_loss)
        return loss[0].clone(), pred.clone(), mask.clone()


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViTNoCT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=8,
                                    decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViTNoCT(patch_size=16, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512, decoder_depth=8,
                                    decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_huge_patch14_dec512d8b(**kwargs):
    return MaskedAutoencoderViTNoCT(patch_size=14, embed_dim=1280, depth=32, num_heads=16, decoder_embed_dim=512, decoder_depth=8,
                                    decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


# recommended archs (models_mae_noct.py:232-235)
mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b
mae_vit_large_patch16 = mae_vit_large_patch16_dec512d8b
mae_vit_huge_patch14 = mae_vit_huge_patch14_dec512d8b
