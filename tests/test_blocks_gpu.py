"""GPU: the reference's block modules CALLED ON THEIR OWN (countr_amd.models_crossvit: Mlp, Attention, CrossAttention, Block,
CrossAttentionBlock -- models_crossvit.py:46-156, timm Block) against the oracle's restatement of the same modules on the same
weights: fp32 parity mode to 1e-3 of the output's maximum (measured ~1e-6), bf16 mode to bf16 tolerance; the module surface
(constructor keywords, [B, N, C] in / out, dtype preserved, GPU-only errors); and the same modules UNDER AUTOGRAD -- trainable, as
the reference's are -- against fp64 autograd of the oracle's restatement."""
import numpy as np
import pytest
import torch
import torch.nn as nn
from functools import partial

from oracle import countr_ref as R

pytestmark = pytest.mark.gpu


def _init(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / np.sqrt(p.shape[1])))
            elif n.endswith("weight"):          # LayerNorm gamma
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return m


def _params(m, prefix=""):
    return {prefix + k: v.detach().cpu().double() for k, v in m.state_dict().items()}


def _rel(a, b):
    return (a.double().cpu() - b).abs().max().item() / b.abs().max().item()


CASES = [("fp32", 1e-3), ("bf16", 3e-2)]


@pytest.mark.parametrize("precision,tol", CASES)
@pytest.mark.parametrize("dim,heads,B,N", [(768, 12, 2, 576), (512, 16, 3, 576), (256, 4, 1, 200)])
def test_block_matches_oracle(precision, tol, dim, heads, B, N):
    from countr_amd.models_crossvit import Attention, Block, Mlp
    blk = _init(Block(dim, heads, 4.0, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), precision=precision), 5).cuda()
    x = torch.randn(B, N, dim, generator=torch.Generator().manual_seed(6)) * 0.8
    p = _params(blk, "b.")
    xd = x.double()
    with torch.no_grad():
        ref_attn = R.self_attention(xd, p, "b.attn", heads)
        ref_mlp = R.mlp(xd, p, "b.mlp")
        h = xd + R.self_attention(R.layer_norm(xd, p["b.norm1.weight"], p["b.norm1.bias"]), p, "b.attn", heads)
        ref_blk = h + R.mlp(R.layer_norm(h, p["b.norm2.weight"], p["b.norm2.bias"]), p, "b.mlp")
        got_attn, got_mlp, got_blk = blk.attn(x.cuda()), blk.mlp(x.cuda()), blk(x.cuda())
    assert got_blk.shape == (B, N, dim) and got_blk.dtype == torch.float32
    assert _rel(got_attn, ref_attn) < tol and _rel(got_mlp, ref_mlp) < tol and _rel(got_blk, ref_blk) < tol
    assert isinstance(blk.attn, Attention) and isinstance(blk.mlp, Mlp)


@pytest.mark.parametrize("precision,tol", CASES)
@pytest.mark.parametrize("S", [1, 3, 13])
def test_cross_attention_block_matches_oracle(precision, tol, S):
    from countr_amd.models_crossvit import CrossAttentionBlock
    dim, heads, B, N = 512, 16, 2, 576
    blk = _init(CrossAttentionBlock(dim, heads, 4.0, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), precision=precision), 7).cuda()
    g = torch.Generator().manual_seed(8)
    x, y = torch.randn(B, N, dim, generator=g) * 0.8, torch.randn(B, S, dim, generator=g)
    p = _params(blk, "b.")
    xd, yd = x.double(), y.double()
    with torch.no_grad():
        ref_x = R.cross_attention(xd, yd, p, "b.attn", heads)
        h = xd + R.self_attention(R.layer_norm(xd, p["b.norm0.weight"], p["b.norm0.bias"]), p, "b.selfattn", heads)
        h = h + R.cross_attention(R.layer_norm(h, p["b.norm1.weight"], p["b.norm1.bias"]), yd, p, "b.attn", heads)
        ref = h + R.mlp(R.layer_norm(h, p["b.norm2.weight"], p["b.norm2.bias"]), p, "b.mlp")
        got_x, got = blk.attn(x.cuda(), y.cuda()), blk(x.cuda(), y.cuda())
    assert _rel(got_x, ref_x) < tol and _rel(got, ref) < tol


def test_module_surface_and_errors():
    from countr_amd import _lib
    from countr_amd.models_crossvit import Attention, Block, CrossAttention
    a = Attention(128, num_heads=2, qkv_bias=True).cuda()
    x = torch.randn(1, 64, 128, device="cuda")
    assert a(x).requires_grad                            # autograd is recording and the parameters require grad: trainable
    with torch.no_grad():
        out = a(x.half())
        assert out.dtype == torch.float16 and out.shape == x.shape
        with pytest.raises(_lib.CountrError, match="GPU only"):
            a(x.cpu())
        with pytest.raises(_lib.CountrError, match="D == 512"):        # the kernel's layout: 16 heads of 32 (the decoder's width)
            CrossAttention(128, num_heads=4).cuda()(x, torch.randn(1, 9, 128, device="cuda"))
    with pytest.raises(ValueError):
        Block(128, 2, drop_path=0.1)
    # inside the models the same classes are parameter containers under the reference's keys
    assert sorted(k for k in Block(128, 2, qkv_bias=True).state_dict()) == sorted(
        ["norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias",
         "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"])


def test_a_block_of_the_full_model_is_callable():
    """model.blocks[i](x) on the drop-in model equals the oracle's block i on the model's own weights (fp32 mode)."""
    from oracle import weights as W
    import models_mae_cross
    m = models_mae_cross.mae_vit_base_patch16(precision="fp32")
    sd = W.make_state_dict("mae_vit_base_patch16", seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.cuda().eval()
    x = torch.randn(1, 576, 768, generator=torch.Generator().manual_seed(3))
    p = {k: torch.from_numpy(v).double() for k, v in sd.items() if k.startswith("blocks.3.")}
    xd = x.double()
    h = xd + R.self_attention(R.layer_norm(xd, p["blocks.3.norm1.weight"], p["blocks.3.norm1.bias"]), p, "blocks.3.attn", 12)
    ref = h + R.mlp(R.layer_norm(h, p["blocks.3.norm2.weight"], p["blocks.3.norm2.bias"]), p, "blocks.3.mlp")
    with torch.no_grad():
        got = m.blocks[3](x.cuda())
    assert _rel(got, ref) < 1e-3


def _grad_check(mod, ref_fn, inputs, tol, cos_min):
    """mod(*inputs).backward(w) through the C-ABI autograd primitives against fp64 autograd of the oracle's restatement on the same
    weights: input gradients and every parameter gradient by relative error of the tensor (fp32) / direction + norm (bf16)."""
    xs = [t.clone().cuda().requires_grad_(True) for t in inputs]
    out = mod(*xs)
    assert out.requires_grad and out.dtype == torch.float32
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(99)).cuda()
    (out * w).sum().backward()
    p = {"b." + k: v.detach().cpu().double().requires_grad_(True) for k, v in mod.named_parameters()}
    xr = [t.double().requires_grad_(True) for t in inputs]
    ref = ref_fn(p, *xr)
    assert _rel(out.detach(), ref.detach()) < (tol if tol > 1e-3 else 1e-3)
    (ref * w.cpu().double()).sum().backward()
    pairs = [("input%d" % i, a.grad, b.grad) for i, (a, b) in enumerate(zip(xs, xr))] + \
            [(k, v.grad, p["b." + k].grad) for k, v in mod.named_parameters()]
    big = max(want.norm().item() for _n, _g, want in pairs)
    for name, got, want in pairs:
        assert got is not None, name
        got = got.detach().cpu().double()
        if want.norm().item() < 1e-9 * big:          # (attn.wk.bias: a bias on the keys shifts every score of a row alike -- zero gradient)
            assert got.norm().item() < 1e-4 * big, (name, got.norm().item())
            continue
        err = (got - want).norm().item() / want.norm().item()
        cos = ((got * want).sum() / (got.norm() * want.norm())).item()
        assert err < tol and cos > cos_min, (name, err, cos)
    return len(pairs)


GRAD_CASES = [("fp32", 2e-3, 0.999999), ("bf16", 6e-2, 0.998)]


@pytest.mark.parametrize("precision,tol,cos_min", GRAD_CASES)
@pytest.mark.parametrize("dim,heads,B,N", [(512, 16, 2, 576), (768, 12, 1, 576), (256, 4, 1, 200)])
def test_block_is_trainable_and_matches_oracle_autograd(precision, tol, cos_min, dim, heads, B, N):
    """timm Block under autograd (dh 32 / 64: the fused attention forward + backward kernels in bf16; the unfused softmax path in fp32)."""
    from countr_amd.models_crossvit import Block
    blk = _init(Block(dim, heads, 4.0, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), precision=precision), 11).cuda()
    x = torch.randn(B, N, dim, generator=torch.Generator().manual_seed(12)) * 0.8

    def ref(p, xd):
        h = xd + R.self_attention(R.layer_norm(xd, p["b.norm1.weight"], p["b.norm1.bias"]), p, "b.attn", heads)
        return h + R.mlp(R.layer_norm(h, p["b.norm2.weight"], p["b.norm2.bias"]), p, "b.mlp")
    assert _grad_check(blk, ref, [x], tol, cos_min) == 13


@pytest.mark.parametrize("precision,tol,cos_min", GRAD_CASES)
@pytest.mark.parametrize("S", [3, 11])
def test_cross_attention_block_is_trainable_and_matches_oracle_autograd(precision, tol, cos_min, S):
    """models_crossvit.py:130-156 under autograd, gradients to x, y (the exemplar tokens) and all 22 parameter tensors; S = 11 takes the
    many-key cross-attention kernels."""
    from countr_amd.models_crossvit import CrossAttentionBlock
    dim, heads, B, N = 512, 16, 2, 576
    blk = _init(CrossAttentionBlock(dim, heads, 4.0, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), precision=precision), 13).cuda()
    g = torch.Generator().manual_seed(14)
    x, y = torch.randn(B, N, dim, generator=g) * 0.8, torch.randn(B, S, dim, generator=g)

    def ref(p, xd, yd):
        h = xd + R.self_attention(R.layer_norm(xd, p["b.norm0.weight"], p["b.norm0.bias"]), p, "b.selfattn", heads)
        h = h + R.cross_attention(R.layer_norm(h, p["b.norm1.weight"], p["b.norm1.bias"]), yd, p, "b.attn", heads)
        return h + R.mlp(R.layer_norm(h, p["b.norm2.weight"], p["b.norm2.bias"]), p, "b.mlp")
    assert _grad_check(blk, ref, [x, y], tol, cos_min) == 2 + 22


@pytest.mark.parametrize("precision,tol,cos_min", GRAD_CASES)
def test_leaf_modules_are_trainable(precision, tol, cos_min):
    from countr_amd.models_crossvit import Attention, CrossAttention, Mlp
    g = torch.Generator().manual_seed(15)
    x, y = torch.randn(2, 576, 512, generator=g), torch.randn(2, 2, 512, generator=g)
    mlp = _init(Mlp(512, 2048, precision=precision), 16).cuda()
    _grad_check(mlp, lambda p, xd: R.mlp(xd, p, "b"), [x], tol, cos_min)
    att = _init(Attention(512, 16, qkv_bias=True, precision=precision), 17).cuda()
    _grad_check(att, lambda p, xd: R.self_attention(xd, p, "b", 16), [x], tol, cos_min)
    xat = _init(CrossAttention(512, 16, qkv_bias=True, precision=precision), 18).cuda()
    _grad_check(xat, lambda p, xd, yd: R.cross_attention(xd, yd, p, "b", 16), [x, y], tol, cos_min)
