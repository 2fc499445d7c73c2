"""Parameter containers mirroring the reference block modules (models_crossvit.py:46-156 and timm 0.4.9
PatchEmbed / Block).  They own nn.Parameters under the reference's attribute names so that state_dict keys
match; they are never called -- the math runs in the HIP engine (countr_amd/engine.py)."""
import torch.nn as nn


class _Shell(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("container module: the forward pass runs in countr_amd.engine (HIP kernels)")


class Mlp(_Shell):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(_Shell):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class CrossAttention(_Shell):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.wq = nn.Linear(dim, dim, bias=qkv_bias)
        self.wk = nn.Linear(dim, dim, bias=qkv_bias)
        self.wv = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(_Shell):
    """timm Block: x += attn(norm1(x)); x += mlp(norm2(x))."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class CrossAttentionBlock(_Shell):
    """models_crossvit.py:130-156: self-attn, cross-attn against the exemplar tokens, mlp."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm0 = norm_layer(dim)
        self.selfattn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm1 = norm_layer(dim)
        self.attn = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class PatchEmbed(_Shell):
    """timm PatchEmbed: Conv2d(k=p, s=p) -> flatten(2).transpose(1, 2)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
