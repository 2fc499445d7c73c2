"""Forward of the reference's standalone block modules through the C ABI (include/countr_hip.h): what makes
countr_amd.models_crossvit.{Mlp, Attention, CrossAttention, Block, CrossAttentionBlock} callable, so that a maintainer can swap ONE
module of the reference model (models_crossvit.py:46-156; timm 0.4.9 Block at models_mae_cross.py:32-34) for its HIP counterpart.

Under torch.no_grad() (or with nothing that requires grad) the forward runs the fused launches of the model's engine (residual in the
GEMM epilogue, rounded copies made per call).  With autograd recording the same modules are TRAINABLE, as the reference's are
(models_crossvit.py:130-156 are plain nn.Modules): every primitive -- nn.Linear (+ GELU), LayerNorm, the self-attention core, the
cross-attention core -- is a torch.autograd.Function whose forward AND backward are C-ABI launches (countr_gemm in its dgrad / wgrad
operand modes, countr_colsum, countr_gelu_bwd, countr_layernorm_bwd, countr_attn_bwd / the unfused softmax path, countr_xattn_bwd);
torch only adds the residuals and carries the tape.  (Whole-model training goes through SupervisedMAE / FinetuneStep, whose static
launch lists run the same kernels without the per-call buffers.)  Eager launches on torch's current stream, buffers from torch's
allocator.  No CPU fallback: a CPU tensor or a missing library raises."""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, BF16, F32, OP_COL, OP_ROW, GemmArgs


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Runner:
    """precision 'bf16' / 'fp16': 16-bit GEMM operands (the two builds of the library, _lib.variant_of), fp32 accumulation, fp32
    residual stream; 'fp32': exact-fp32 MFMA everywhere (parity)."""

    def __init__(self, precision="bf16"):
        if precision not in ("bf16", "fp16", "fp32"):
            raise ValueError("precision must be 'bf16', 'fp16' or 'fp32'")
        self.L = _lib.lib(_lib.variant_of(precision))
        self.code = BF16 if precision in ("bf16", "fp16") else F32
        self.tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[precision]

    # ---- plumbing
    def check_input(self, *tensors, module=None):
        for t in tensors:
            if not t.is_cuda:
                raise _lib.CountrError("countr_amd block modules run on the GPU only (no CPU fallback): got a %s tensor" % t.device)
        params = list(module.parameters()) if module is not None else []
        if any(not p.is_cuda for p in params):
            raise _lib.CountrError("countr_amd block modules run on the GPU only (no CPU fallback): move the module to the GPU first")
        _lib.check(self.L.countr_init(tensors[0].device.index or 0), "countr_init")
        return torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or any(p.requires_grad for p in params))

    def weight(self, p):
        """GEMM operand of an nn.Linear weight in the compute dtype (fp32: the parameter itself; bf16: a rounded copy made per call --
        the fused AdamW of FinetuneStep updates parameters behind autograd's version counters, so nothing is cached)."""
        if self.code == F32:
            return p.detach().float().contiguous()
        src = p.detach().float().contiguous()
        w = torch.empty(src.shape, device=src.device, dtype=self.tdt)
        _lib.check(self.L.countr_cast_permute(src.data_ptr(), w.data_ptr(), src.numel(), 0, 0, 0, 0, BF16, _stream()), "cast")
        return w

    def to_operand(self, x_f32):
        if self.code == F32:
            return x_f32
        y = torch.empty(x_f32.shape, device=x_f32.device, dtype=self.tdt)
        _lib.check(self.L.countr_cast_permute(x_f32.data_ptr(), y.data_ptr(), x_f32.numel(), 0, 0, 0, 0, BF16, _stream()), "cast")
        return y

    # ---- ops
    def linear(self, x, lin, act=ACT_NONE, resid=None, out_f32=False):
        """x [M, K] (compute dtype) -> act(x W^T + b) (+ resid): compute dtype, or fp32 when it carries a residual / out_f32."""
        M, K = x.shape
        N = lin.weight.shape[0]
        W = self.weight(lin.weight)
        f32 = out_f32 or resid is not None or self.code == F32
        out = torch.empty((M, N), device=x.device, dtype=torch.float32 if f32 else self.tdt)
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

    def self_attention(self, qkv, B, N, heads, keep_probs=False):
        """packed qkv [B * N, 3 * D] (compute dtype) -> softmax(q k^T dh^-0.5) v as [B * N, D] (models_crossvit.py:84-91).
        keep_probs: the unfused path, returning (out, P) for a backward."""
        D = qkv.shape[1] // 3
        dh = D // heads
        scale = dh ** -0.5
        out = torch.empty((B * N, D), device=qkv.device, dtype=self.tdt)
        if self.code == BF16 and dh in (32, 64) and not keep_probs:
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
        return (out, probs) if keep_probs else out

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


# ------------------------------------------------------------------------------------------------------------------------------
# autograd: the primitives as torch.autograd.Functions over the C ABI
# ------------------------------------------------------------------------------------------------------------------------------
def _gemm(r, ma, mb, **kw):
    a = GemmArgs()
    a.alpha, a.nbatch, a.nb1, a.splitk = 1.0, 1, 1, 1
    for k, v in kw.items():
        setattr(a, k, v)
    _lib.check(r.L.countr_gemm(C.byref(a), r.code, ma, mb, _stream()), "countr_gemm")


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b); x [M, K] in the compute dtype, W / b the fp32 parameters.  Backward: dx = dy W ((ROW, COL) operand modes),
    dW = dy^T x ((COL, COL)), db = column sums of dy, GELU' through the saved pre-activation (models_crossvit.py:61-67,84,92,115-119)."""

    @staticmethod
    def forward(ctx, r, x, weight, bias, act, out_f32):
        M, K = x.shape
        N = weight.shape[0]
        x = x.contiguous()
        W = r.weight(weight)
        f32 = out_f32 or r.code == F32
        out = torch.empty((M, N), device=x.device, dtype=torch.float32 if f32 else r.tdt)
        pre = torch.empty((M, N), device=x.device, dtype=out.dtype) if act == ACT_GELU else None
        b = bias.detach().float().contiguous() if bias is not None else None
        _gemm(r, OP_ROW, OP_ROW, A=x.data_ptr(), B=W.data_ptr(), C=out.data_ptr(), C2=pre.data_ptr() if pre is not None else None,
              bias=b.data_ptr() if b is not None else None, lda=K, ldb=K, ldc=N, M=M, N=N, K=K, act=act, out_bf16=int(not f32))
        ctx.r, ctx.act, ctx.has_bias = r, act, bias is not None
        ctx.save_for_backward(x, W, pre)
        return out

    @staticmethod
    def backward(ctx, dy):
        r = ctx.r
        x, W, pre = ctx.saved_tensors
        M, K = x.shape
        N = W.shape[0]
        dyt = dy.contiguous().to(r.tdt)
        if ctx.act == ACT_GELU:
            g = torch.empty_like(dyt)
            pt = pre if pre.dtype == r.tdt else pre.to(r.tdt)
            _lib.check(r.L.countr_gelu_bwd(dyt.data_ptr(), pt.data_ptr(), g.data_ptr(), M * N, r.code, _stream()), "countr_gelu_bwd")
            dyt = g
        dx = dW = db = None
        if ctx.needs_input_grad[1]:
            dx = torch.empty((M, K), device=x.device, dtype=x.dtype)
            _gemm(r, OP_ROW, OP_COL, A=dyt.data_ptr(), B=W.data_ptr(), C=dx.data_ptr(), lda=N, ldb=K, ldc=K, M=M, N=K, K=N,
                  out_bf16=int(x.dtype in (torch.bfloat16, torch.float16)))
        if ctx.needs_input_grad[2]:
            dW = torch.empty((N, K), device=x.device, dtype=torch.float32)
            _gemm(r, OP_COL, OP_COL, A=dyt.data_ptr(), B=x.data_ptr(), C=dW.data_ptr(), lda=N, ldb=K, ldc=K, M=N, N=K, K=M, out_bf16=0)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            db = torch.empty(N, device=x.device, dtype=torch.float32)
            ws = torch.empty(r.L.countr_colsum_nparts() * N, device=x.device, dtype=torch.float32)
            _lib.check(r.L.countr_colsum(dyt.data_ptr(), db.data_ptr(), ws.data_ptr(), M, N, r.code, 0, _stream()), "countr_colsum")
        return None, dx, dW, db, None, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension of the fp32 residual stream x [rows, D] -> the compute dtype (models_crossvit.py:153-155)."""

    @staticmethod
    def forward(ctx, r, x, gamma, beta, eps):
        rows, D = x.shape
        x = x.contiguous()
        y = torch.empty((rows, D), device=x.device, dtype=r.tdt)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        _lib.check(r.L.countr_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D,
                                            float(eps), int(r.code == BF16), _stream()), "countr_layernorm_fwd")
        ctx.r = r
        ctx.save_for_backward(x, g, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        r = ctx.r
        x, g, mean, rstd = ctx.saved_tensors
        rows, D = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg, db = torch.empty(D, device=x.device, dtype=torch.float32), torch.empty(D, device=x.device, dtype=torch.float32)
        ws = torch.empty(r.L.countr_layernorm_bwd_nblocks() * 2 * D, device=x.device, dtype=torch.float32)
        _lib.check(r.L.countr_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                            dg.data_ptr(), db.data_ptr(), ws.data_ptr(), rows, D, int(dy.dtype in (torch.bfloat16, torch.float16)), 0, 0, None,
                                            _stream()), "countr_layernorm_bwd")
        return None, dx, dg, db, None


class SelfAttentionFn(torch.autograd.Function):
    """softmax(q k^T dh^-0.5) v on the packed qkv [B * N, 3 D] (models_crossvit.py:84-91).  bf16 with dh 32 / 64: the fused kernels
    (countr_attn_fwd keeps the log-sum-exp, countr_attn_bwd recomputes P); otherwise the unfused path with P kept."""

    @staticmethod
    def forward(ctx, r, qkv, B, N, heads):
        qkv = qkv.contiguous()
        D = qkv.shape[1] // 3
        dh = D // heads
        ctx.r, ctx.dims = r, (B, N, heads, D, dh)
        if r.code == BF16 and dh in (32, 64):
            out = torch.empty((B * N, D), device=qkv.device, dtype=r.tdt)
            lse = torch.empty((B, heads, N), device=qkv.device, dtype=torch.float32)
            _lib.check(r.L.countr_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, heads, dh, dh ** -0.5, _stream()), "countr_attn_fwd")
            ctx.fused = True
            ctx.save_for_backward(qkv, out, lse)
            return out
        out, probs = r.self_attention(qkv, B, N, heads, keep_probs=True)
        ctx.fused = False
        ctx.save_for_backward(qkv, probs)
        return out

    @staticmethod
    def backward(ctx, dout):
        r = ctx.r
        B, N, heads, D, dh = ctx.dims
        dout = dout.contiguous()
        scale = dh ** -0.5
        if ctx.fused:
            qkv, out, lse = ctx.saved_tensors
            dqkv = torch.empty_like(qkv)
            delta = torch.empty((B, heads, N), device=qkv.device, dtype=torch.float32)
            _lib.check(r.L.countr_attn_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(),
                                           B, N, heads, dh, scale, _stream()), "countr_attn_bwd")
            return None, dqkv, None, None, None
        qkv, probs = ctx.saved_tensors
        es = qkv.element_size()
        dqkv = torch.empty_like(qkv)
        dP = torch.empty((B * heads, N, N), device=qkv.device, dtype=torch.float32)
        dS = torch.empty((B * heads, N, N), device=qkv.device, dtype=r.tdt)
        ob = int(r.code == BF16)
        nb = dict(nbatch=B * heads, nb1=heads)
        # dV = P^T dO | dP = dO V^T | dS = softmax' | dQ = dS K | dK = dS^T Q   (engine._attention_bwd)
        _gemm(r, OP_COL, OP_COL, A=probs.data_ptr(), B=dout.data_ptr(), C=dqkv.data_ptr() + 2 * D * es, lda=N, ldb=D, ldc=3 * D, M=N, N=dh, K=N,
              sA0=heads * N * N, sA1=N * N, sB0=N * D, sB1=dh, sC0=N * 3 * D, sC1=dh, out_bf16=ob, **nb)
        _gemm(r, OP_ROW, OP_ROW, A=dout.data_ptr(), B=qkv.data_ptr() + 2 * D * es, C=dP.data_ptr(), lda=D, ldb=3 * D, ldc=N, M=N, N=N, K=dh,
              sA0=N * D, sA1=dh, sB0=N * 3 * D, sB1=dh, sC0=heads * N * N, sC1=N * N, out_bf16=0, **nb)
        _lib.check(r.L.countr_softmax_bwd(probs.data_ptr(), dP.data_ptr(), dS.data_ptr(), B * heads * N, N, scale, r.code, _stream()), "countr_softmax_bwd")
        _gemm(r, OP_ROW, OP_COL, A=dS.data_ptr(), B=qkv.data_ptr() + D * es, C=dqkv.data_ptr(), lda=N, ldb=3 * D, ldc=3 * D, M=N, N=dh, K=N,
              sA0=heads * N * N, sA1=N * N, sB0=N * 3 * D, sB1=dh, sC0=N * 3 * D, sC1=dh, out_bf16=ob, **nb)
        _gemm(r, OP_COL, OP_COL, A=dS.data_ptr(), B=qkv.data_ptr(), C=dqkv.data_ptr() + D * es, lda=N, ldb=3 * D, ldc=3 * D, M=N, N=dh, K=N,
              sA0=heads * N * N, sA1=N * N, sB0=N * 3 * D, sB1=dh, sC0=N * 3 * D, sC1=dh, out_bf16=ob, **nb)
        return None, dqkv, None, None, None


class CrossAttentionFn(torch.autograd.Function):
    """softmax(q k^T dh^-0.5) v with q [B * N, D], k / v [B * S, D] (models_crossvit.py:111-128; D = 512, 16 heads of 32)."""

    @staticmethod
    def forward(ctx, r, q, k, v, B, N, S, heads):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out = r.cross_attention(q, k, v, B, N, S, heads)
        ctx.r, ctx.dims = r, (B, N, S, heads)
        ctx.save_for_backward(q, k, v)
        return out

    @staticmethod
    def backward(ctx, dout):
        r = ctx.r
        B, N, S, heads = ctx.dims
        q, k, v = ctx.saved_tensors
        D = q.shape[1]
        dout = dout.contiguous()
        dq = torch.empty_like(q)
        dk = torch.empty((B * S, D), device=q.device, dtype=torch.float32)
        dv = torch.empty_like(dk)
        ws = torch.empty(int(r.L.countr_xattn_bwd_workspace_floats(B, N, S, D)), device=q.device, dtype=torch.float32)
        _lib.check(r.L.countr_xattn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                        ws.data_ptr(), B, N, S, D, heads, D, (D // heads) ** -0.5, r.code, None, None, _stream()), "countr_xattn_bwd")
        return None, dq, dk.to(k.dtype), dv.to(v.dtype), None, None, None, None


def _rows_grad(x):
    if x.dim() != 3:
        raise ValueError("expected a [B, N, C] tensor, got %s" % (tuple(x.shape),))
    B, N, Cc = x.shape
    return x.reshape(B * N, Cc).float(), B, N


def mlp_autograd(r, m, xt, out_f32=True):
    h = LinearFn.apply(r, xt, m.fc1.weight, m.fc1.bias, ACT_GELU, False)
    return LinearFn.apply(r, h, m.fc2.weight, m.fc2.bias, ACT_NONE, out_f32)


def attention_autograd(r, m, xt, B, N):
    qkv = LinearFn.apply(r, xt, m.qkv.weight, m.qkv.bias, ACT_NONE, False)
    o = SelfAttentionFn.apply(r, qkv, B, N, m.num_heads)
    return LinearFn.apply(r, o, m.proj.weight, m.proj.bias, ACT_NONE, True)


def cross_attention_autograd(r, m, xt, yt, B, N, S):
    q = LinearFn.apply(r, xt, m.wq.weight, m.wq.bias, ACT_NONE, False)
    k = LinearFn.apply(r, yt, m.wk.weight, m.wk.bias, ACT_NONE, False)
    v = LinearFn.apply(r, yt, m.wv.weight, m.wv.bias, ACT_NONE, False)
    o = CrossAttentionFn.apply(r, q, k, v, B, N, S, m.num_heads)
    return LinearFn.apply(r, o, m.proj.weight, m.proj.bias, ACT_NONE, True)


def layernorm_autograd(r, x_f32, norm):
    return LayerNormFn.apply(r, x_f32, norm.weight, norm.bias, float(norm.eps))
