"""The reference block modules (models_crossvit.py:46-156 and timm 0.4.9 PatchEmbed / Block) on the HIP kernels.

Inside SupervisedMAE / MaskedAutoencoderViTNoCT they are parameter containers: they own nn.Parameters under the reference's attribute
names (state_dict keys match) and the model's engine (countr_amd/engine.py) runs static launch lists over them.  CALLED ON THEIR OWN
-- Mlp(x), Attention(x), CrossAttention(x, y), Block(x), CrossAttentionBlock(x, y): the reference's signatures, [B, N, C] tensors -- they
run through the same C-ABI exports (countr_amd/blocks.py: countr_gemm, countr_layernorm_fwd, countr_attn_fwd,
countr_xattn_fwd ...), so that a maintainer can swap a single module of the reference model for its HIP counterpart -- also under
autograd (every primitive's backward is a C-ABI launch too: blocks.LinearFn / LayerNormFn / SelfAttentionFn / CrossAttentionFn);
`precision` ('bf16' default, 'fp32' = parity mode) is an extra keyword / attribute.  PatchEmbed stays a container (the model's engine
fuses it with the pos-embed add)."""
import torch
import torch.nn as nn

from . import blocks


class _HipModule(nn.Module):
    precision = "bf16"

    def _runner(self):
        r = self.__dict__.get("_hip_runner")
        if r is None or r[0] != self.precision:
            r = (self.precision, blocks.Runner(self.precision))
            self.__dict__["_hip_runner"] = r            # (not a submodule / parameter: stays out of state_dict)
        return r[1]

    @staticmethod
    def _no_dropout(**kw):
        for k, v in kw.items():
            if v:
                raise ValueError("%s = %r: the reference trains and tests with 0 (models_mae_cross.py:32-34,44-46); not implemented" % (k, v))


class _Shell(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("container module: the forward pass runs in countr_amd.engine (HIP kernels)")


class Mlp(_HipModule):
    """models_crossvit.py:46-67: fc2(GELU(fc1(x)))."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0., precision="bf16"):
        super().__init__()
        self._no_dropout(drop=drop)
        if act_layer is not nn.GELU:
            raise ValueError("Mlp: only nn.GELU (the reference's activation) runs on the HIP kernels")
        self.precision = precision
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        r = self._runner()
        if r.check_input(x, module=self):                                   # autograd is recording: differentiable primitives
            x2, B, N = blocks._rows_grad(x)
            return blocks.mlp_autograd(r, self, x2.to(r.tdt)).view(B, N, -1).to(x.dtype)
        x2, B, N = blocks._rows(x)
        return blocks.mlp_forward(r, self, r.to_operand(x2)).view(B, N, -1).to(x.dtype)


class Attention(_HipModule):
    """models_crossvit.py:69-94 (== timm Attention)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., precision="bf16"):
        super().__init__()
        self._no_dropout(attn_drop=attn_drop, proj_drop=proj_drop)
        if qk_scale is not None and abs(qk_scale - (dim // num_heads) ** -0.5) > 1e-12:
            raise ValueError("Attention: qk_scale other than head_dim ** -0.5 is not implemented")
        self.precision = precision
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        r = self._runner()
        if r.check_input(x, module=self):
            x2, B, N = blocks._rows_grad(x)
            return blocks.attention_autograd(r, self, x2.to(r.tdt), B, N).view(B, N, -1).to(x.dtype)
        x2, B, N = blocks._rows(x)
        return blocks.attention_forward(r, self, r.to_operand(x2), B, N).view(B, N, -1).to(x.dtype)


class CrossAttention(_HipModule):
    """models_crossvit.py:96-128: queries from x, keys / values from y (any number of tokens; up to 8 -- the exemplar tokens -- sit in registers)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., precision="bf16"):
        super().__init__()
        self._no_dropout(attn_drop=attn_drop, proj_drop=proj_drop)
        if qk_scale is not None and abs(qk_scale - (dim // num_heads) ** -0.5) > 1e-12:
            raise ValueError("CrossAttention: qk_scale other than head_dim ** -0.5 is not implemented")
        self.precision = precision
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.wq = nn.Linear(dim, dim, bias=qkv_bias)
        self.wk = nn.Linear(dim, dim, bias=qkv_bias)
        self.wv = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, y):
        r = self._runner()
        if r.check_input(x, y, module=self):
            x2, B, N = blocks._rows_grad(x)
            y2, By, S = blocks._rows_grad(y)
            if By != B:
                raise ValueError("CrossAttention: x and y must share the batch dimension")
            return blocks.cross_attention_autograd(r, self, x2.to(r.tdt), y2.to(r.tdt), B, N, S).view(B, N, -1).to(x.dtype)
        x2, B, N = blocks._rows(x)
        y2, By, S = blocks._rows(y)
        if By != B:
            raise ValueError("CrossAttention: x and y must share the batch dimension")
        return blocks.cross_attention_forward(r, self, r.to_operand(x2), r.to_operand(y2), B, N, S).view(B, N, -1).to(x.dtype)


class Block(_HipModule):
    """timm 0.4.9 Block (models_mae_cross.py:32-34): x += attn(norm1(x)); x += mlp(norm2(x))."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, precision="bf16"):
        super().__init__()
        self._no_dropout(drop=drop, attn_drop=attn_drop, drop_path=drop_path)
        self.precision = precision
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, precision=precision)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, precision=precision)

    def forward(self, x):
        r = self._runner()
        if r.check_input(x, module=self):
            x2, B, N = blocks._rows_grad(x)
            x2 = x2 + blocks.attention_autograd(r, self.attn, blocks.layernorm_autograd(r, x2, self.norm1), B, N)
            x2 = x2 + blocks.mlp_autograd(r, self.mlp, blocks.layernorm_autograd(r, x2, self.norm2))
            return x2.view(B, N, -1).to(x.dtype)
        x2, B, N = blocks._rows(x)
        x2 = blocks.attention_forward(r, self.attn, r.layernorm(x2, self.norm1), B, N, resid=x2)
        x2 = blocks.mlp_forward(r, self.mlp, r.layernorm(x2, self.norm2), resid=x2)
        return x2.view(B, N, -1).to(x.dtype)


class CrossAttentionBlock(_HipModule):
    """models_crossvit.py:130-156: x += selfattn(norm0(x)); x += attn(norm1(x), y); x += mlp(norm2(x))."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, precision="bf16"):
        super().__init__()
        self._no_dropout(drop=drop, attn_drop=attn_drop, drop_path=drop_path)
        self.precision = precision
        self.norm0 = norm_layer(dim)
        self.selfattn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, precision=precision)
        self.norm1 = norm_layer(dim)
        self.attn = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, precision=precision)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, precision=precision)

    def forward(self, x, y):
        r = self._runner()
        if r.check_input(x, y, module=self):
            x2, B, N = blocks._rows_grad(x)
            y2, By, S = blocks._rows_grad(y)
            if By != B:
                raise ValueError("CrossAttentionBlock: x and y must share the batch dimension")
            x2 = x2 + blocks.attention_autograd(r, self.selfattn, blocks.layernorm_autograd(r, x2, self.norm0), B, N)
            x2 = x2 + blocks.cross_attention_autograd(r, self.attn, blocks.layernorm_autograd(r, x2, self.norm1), y2.to(r.tdt), B, N, S)
            x2 = x2 + blocks.mlp_autograd(r, self.mlp, blocks.layernorm_autograd(r, x2, self.norm2))
            return x2.view(B, N, -1).to(x.dtype)
        x2, B, N = blocks._rows(x)
        y2, By, S = blocks._rows(y)
        if By != B:
            raise ValueError("CrossAttentionBlock: x and y must share the batch dimension")
        yt = r.to_operand(y2)
        x2 = blocks.attention_forward(r, self.selfattn, r.layernorm(x2, self.norm0), B, N, resid=x2)
        x2 = blocks.cross_attention_forward(r, self.attn, r.layernorm(x2, self.norm1), yt, B, N, S, resid=x2)
        x2 = blocks.mlp_forward(r, self.mlp, r.layernorm(x2, self.norm2), resid=x2)
        return x2.view(B, N, -1).to(x.dtype)


class PatchEmbed(_Shell):
    """timm PatchEmbed: Conv2d(k=p, s=p) -> flatten(2).transpose(1, 2)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
