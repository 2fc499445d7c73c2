"""Forward of the reference's standalone block modules through the C ABI (include/countr_hip.h): what makes
countr_amd.models_crossvit.{Mlp, Attention, CrossAttention, Block, CrossAttentionBlock} callable, so that a maintainer can swap ONE
module of the reference model (models_crossvit.py:46-156; timm 0.4.9 Block at models_mae_cross.py:32-34) for its HIP counterpart.

Forward only (the frozen encoder of the reference runs under no_grad, models_mae_cross.py:204-205; inference runs everything so):
with autograd recording and a tensor that requires grad these raise -- training goes through SupervisedMAE / FinetuneStep, whose
backward lists run the same kernels.  Eager launches on torch's current stream, buffers from torch's allocator.  No CPU fallback: a CPU tensor or a missing library raises."""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, BF16, F32, OP_COL, OP_ROW, GemmArgs


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Runner:
    """precision 'bf16': bf16 GEMM operands, fp32 accumulation, fp32 residual stream; 'fp32': exact-fp32 MFMA everywhere (parity)."""

    def __init__(self, precision="bf16"):
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.L = _lib.lib()
        self.code = BF16 if precision == "bf16" else F32
        self.tdt = torch.bfloat16 if precision == "bf16" else torch.float32

    # ---- plumbing
    def check_input(self, *tensors, module=None):
        for t in tensors:
            if not t.is_cuda:
                raise _lib.CountrError("countr_amd block modules run on the GPU only (no CPU fallback): got a %s tensor" % t.device)
        params = list(module.parameters()) if module is not None else []
        if any(not p.is_cuda for p in params):
            raise _lib.CountrError("countr_amd block modules run on the GPU only (no CPU fallback): move the module to the GPU first")
        if torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or any(p.requires_grad for p in params)):
            raise RuntimeError("countr_amd block modules are forward-only: call them under torch.no_grad() (training runs through "
                               "countr_amd SupervisedMAE / FinetuneStep)")
        _lib.check(self.L.countr_init(tensors[0].device.index or 0), "countr_init")

    def weight(self, p):
        """GEMM operand of an nn.Linear weight in the compute dtype (fp32: the parameter itself; bf16: a rounded copy made per call --
        the fused AdamW of FinetuneStep updates parameters behind autograd's version counters, so nothing is cached)."""
        if self.code == F32:
            return p.detach().float().contiguous()
        src = p.detach().float().contiguous()
        w = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
        _lib.check(self.L.countr_cast_permute(src.data_ptr(), w.data_ptr(), src.numel(), 0, 0, 0, 0, BF16, _stream()), "cast")
        return w

    def to_operand(self, x_f32):
        if self.code == F32:
            return x_f32
        y = torch.empty(x_f32.shape, device=x_f32.device, dtype=torch.bfloat16)
        _lib.check(self.L.countr_cast_permute(x_f32.data_ptr(), y.data_ptr(), x_f32.numel(), 0, 0, 0, 0, BF16, _stream()), "cast")
        return y

    # ---- ops
    def linear(self, x, lin, act=ACT_NONE, resid=None, out_f32=False):
        """x [M, K] (compute dtype) -> act(x W^T + b) (+ resid): compute dtype, or fp32 when it carries a residual / out_f32."""
        M, K = x.shape
        N = lin.weight.shape[0]
        W = self.weight(lin.weight)
        f32 = out_f32 or resid is not None or self.code == F32
        out = torch.empty((M, N), device=x.device, dtype=torch.float32 if f32 else torch.bfloat16)
        bias = lin.bias.detach().float().contiguous() if lin.bias is not None else None
        a = GemmArgs()
        a.alpha, a.nbatch, a.nb1, a.splitk = 1.0, 1, 1, 1
        a.A, a.B, a.C = x.data_ptr(), W.data_ptr(), out.data_ptr()
        a.bias = bias.data_ptr() if bias is not None else None
        a.resid = resid.data_ptr() if resid is not None else None
        a.lda, a.ldb, a.ldc, a.ldres = K, K, N, N
        a.M, a.N, a.K = M, N, K
        a.act = act
        a.out_bf16 = int(not f32)
        _lib.check(self.L.countr_gemm(C.byref(a), self.code, OP_ROW, OP_ROW, _stream()), "countr_gemm")
        return out

    def layernorm(self, x_f32, norm):
        rows, D = x_f32.shape
        y = torch.empty((rows, D), device=x_f32.device, dtype=self.tdt)
        g, b = norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous()
        _lib.check(self.L.countr_layernorm_fwd(x_f32.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, rows, D,
                                               float(norm.eps), int(self.code == BF16), _stream()), "countr_layernorm_fwd")
        return y

    def self_attention(self, qkv, B, N, heads):
        """packed qkv [B * N, 3 * D] (compute dtype) -> softmax(q k^T dh^-0.5) v as [B * N, D] (models_crossvit.py:84-91)."""
        D = qkv.shape[1] // 3
        dh = D // heads
        scale = dh ** -0.5
        out = torch.empty((B * N, D), device=qkv.device, dtype=self.tdt)
        if self.code == BF16 and dh in (32, 64):
            _lib.check(self.L.countr_attn_fwd(qkv.data_ptr(), out.data_ptr(), None, B, N, heads, dh, scale, _stream()), "countr_attn_fwd")
            return out
        # parity mode / other head sizes: batched QK^T, row softmax, PV through countr_gemm (the engine's unfused path)
        es = qkv.element_size()
        scores = torch.empty((B * heads, N, N), device=qkv.device, dtype=torch.float32)
        probs = torch.empty((B * heads, N, N), device=qkv.device, dtype=self.tdt)

        def g(**kw):
            a = GemmArgs()
            a.alpha, a.nbatch, a.nb1, a.splitk = 1.0, B * heads, heads, 1
            for k, v in kw.items():
                setattr(a, k, v)
            return a
        a = g(A=qkv.data_ptr(), B=qkv.data_ptr() + D * es, C=scores.data_ptr(), lda=3 * D, ldb=3 * D, ldc=N, M=N, N=N, K=dh,
              sA0=N * 3 * D, sA1=dh, sB0=N * 3 * D, sB1=dh, sC0=heads * N * N, sC1=N * N, alpha=scale, out_bf16=0)
        _lib.check(self.L.countr_gemm(C.byref(a), self.code, OP_ROW, OP_ROW, _stream()), "countr_gemm(QK^T)")
        _lib.check(self.L.countr_softmax_fwd(scores.data_ptr(), probs.data_ptr(), B * heads * N, N, int(self.code == BF16), _stream()), "softmax")
        a = g(A=probs.data_ptr(), B=qkv.data_ptr() + 2 * D * es, C=out.data_ptr(), lda=N, ldb=3 * D, ldc=D, M=N, N=dh, K=N,
              sA0=heads * N * N, sA1=N * N, sB0=N * 3 * D, sB1=dh, sC0=N * D, sC1=dh, out_bf16=int(self.code == BF16))
        _lib.check(self.L.countr_gemm(C.byref(a), self.code, OP_ROW, OP_COL, _stream()), "countr_gemm(PV)")
        return out

    def cross_attention(self, q, k, v, B, N, S, heads):
        D = q.shape[1]
        out = torch.empty((B * N, D), device=q.device, dtype=self.tdt)
        _lib.check(self.L.countr_xattn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, N, S, D, heads, D,
                                           (D // heads) ** -0.5, self.code, _stream()), "countr_xattn_fwd")
        return out


def _rows(x):
    """[B, N, C] (any float dtype) -> fp32 [B * N, C] contiguous."""
    if x.dim() != 3:
        raise ValueError("expected a [B, N, C] tensor, got %s" % (tuple(x.shape),))
    B, N, Cc = x.shape
    return x.detach().reshape(B * N, Cc).float().contiguous(), B, N


def mlp_forward(r, m, xt, resid=None):
    """models_crossvit.py:61-67: fc2(GELU(fc1(x))) on the operand xt [rows, C]; + resid (fp32) when given."""
    h = r.linear(xt, m.fc1, act=ACT_GELU)
    return r.linear(h, m.fc2, resid=resid, out_f32=True)


def attention_forward(r, m, xt, B, N, resid=None):
    """models_crossvit.py:82-94."""
    qkv = r.linear(xt, m.qkv)
    o = r.self_attention(qkv, B, N, m.num_heads)
    return r.linear(o, m.proj, resid=resid, out_f32=True)


def cross_attention_forward(r, m, xt, yt, B, N, S, resid=None):
    """models_crossvit.py:111-128."""
    q = r.linear(xt, m.wq)
    k = r.linear(yt, m.wk)
    v = r.linear(yt, m.wv)
    o = r.cross_attention(q, k, v, B, N, S, m.num_heads)
    return r.linear(o, m.proj, resid=resid, out_f32=True)
