"""GPU parity of the generic MFMA GEMM (countr_gemm) against torch fp64 matmul on the same inputs.

Covers every operand-mode combination, both dtypes, ragged M/N/K, batching, split-K and the
epilogue (bias, GELU, residual with row modulo, second output).
"""
import ctypes as C

import pytest
import torch

from countr_amd import _lib

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _mk(shape, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1)
    return x.to("cuda").to(dtype)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ma,mb", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (576, 768, 768), (200, 144, 96), (64, 2304, 32), (1152, 576, 64)])
def test_gemm_modes(hip, dt, ma, mb, M, N, K):
    # logical A[M,K], B[N,K]; stored transposed for COL mode
    A = _mk((M, K), dt, 1)
    B = _mk((N, K), dt, 2)
    As = A if ma == 0 else A.t().contiguous()
    Bs = B if mb == 0 else B.t().contiguous()
    out = torch.empty((M, N), device="cuda", dtype=torch.float32)
    a = _lib.GemmArgs()
    a.A, a.B, a.C = As.data_ptr(), Bs.data_ptr(), out.data_ptr()
    a.lda = K if ma == 0 else M
    a.ldb = K if mb == 0 else N
    a.ldc = N
    a.M, a.N, a.K = M, N, K
    a.alpha = 1.0
    a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    _lib.check(hip.countr_gemm(C.byref(a), 0 if dt == torch.float32 else 1, ma, mb, _stream()), "gemm")
    torch.cuda.synchronize()
    ref = (A.double() @ B.double().t())
    # operands are exactly representable in both dtypes -> only fp32 accumulation-order error
    err = (out.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 1e-4 * scale + 1e-5, (err, scale)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_epilogue(hip, dt):
    M, N, K = 1152, 256, 128
    A = _mk((M, K), dt, 3)
    B = _mk((N, K), dt, 4)
    bias = _mk((N,), torch.float32, 5)
    resid = _mk((576, N), torch.float32, 6)
    for out_bf16 in (0, 1):
        odt = torch.bfloat16 if out_bf16 else torch.float32
        out = torch.empty((M, N), device="cuda", dtype=odt)
        pre = torch.empty((M, N), device="cuda", dtype=odt)
        a = _lib.GemmArgs()
        a.A, a.B, a.C, a.C2 = A.data_ptr(), B.data_ptr(), out.data_ptr(), pre.data_ptr()
        a.bias, a.resid = bias.data_ptr(), resid.data_ptr()
        a.lda, a.ldb, a.ldc, a.ldres = K, K, N, N
        a.M, a.N, a.K = M, N, K
        a.res_mod = 576
        a.act = 1
        a.out_bf16 = out_bf16
        a.alpha = 0.5
        a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        _lib.check(hip.countr_gemm(C.byref(a), 0 if dt == torch.float32 else 1, 0, 0, _stream()), "gemm")
        torch.cuda.synchronize()
        z = 0.5 * (A.double() @ B.double().t()) + bias.double()
        ref = torch.nn.functional.gelu(z) + resid.double().repeat(2, 1)
        tol = 2e-2 if out_bf16 else 1e-4
        assert (pre.double() - z).abs().max().item() <= tol * z.abs().max().item()
        assert (out.double() - ref).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_gelu_backward_epilogue(hip, dt):
    """fc2 dgrad fused with GELU': out = (dy @ W) * gelu'(pre), pre = the saved pre-activation read through C2 (ROW x COL)."""
    M, N, K = 1000, 512, 128          # dy [M, K] (K = fc2 out features), W [K, N] (N = hidden), out/pre [M, N]
    dy = _mk((M, K), dt, 13)
    W = _mk((K, N), dt, 14)
    pre = (_mk((M, N), torch.float32, 15) * 3).to(dt)
    out = torch.empty((M, N), device="cuda", dtype=dt)
    a = _lib.GemmArgs()
    a.A, a.B, a.C, a.C2 = dy.data_ptr(), W.data_ptr(), out.data_ptr(), pre.data_ptr()
    a.lda, a.ldb, a.ldc = K, N, N
    a.M, a.N, a.K = M, N, K
    a.act = _lib.ACT_GELU_BWD
    a.out_bf16 = int(dt == torch.bfloat16)
    a.alpha = 1.0
    a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    keep = pre.clone()
    _lib.check(hip.countr_gemm(C.byref(a), 0 if dt == torch.float32 else 1, 0, 1, _stream()), "gemm")
    torch.cuda.synchronize()
    assert torch.equal(pre, keep)                      # C2 is an input in this mode
    x = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(dy.double() @ W.double())
    tol = 2e-2 if dt == torch.bfloat16 else 1e-4
    assert (out.double() - x.grad).abs().max().item() <= tol * x.grad.abs().max().item()
    a.C2 = None                                         # the mode requires the saved pre-activation
    assert hip.countr_gemm(C.byref(a), 0 if dt == torch.float32 else 1, 0, 1, _stream()) != 0
    a.act = 7
    assert hip.countr_gemm(C.byref(a), 0 if dt == torch.float32 else 1, 0, 1, _stream()) != 0


@pytest.mark.parametrize("shape", [(800, 3072, 768), (3152, 2048, 512), (4608, 2048, 512), (1000, 1024, 256)])
def test_gelu_backward_epilogue_on_the_lean_kernel(hip, shape, monkeypatch):
    """The fc2 input gradient with the transposed weight shadow ((ROW, ROW), bf16): dx = (dy @ Wt^T) * GELU'(pre) in the epilogue of
    linear.hip's three tile forms (one round of 128 x 128, 192 x 256, 256 x 128) == the generic kernel's fused form bit for bit (same
    accumulation order, same derivative), and within bf16 rounding of fp64 autograd (reference: Mlp.forward, models_crossvit.py:61-67)."""
    M, N, K = shape                   # dy [M, K], Wt [N, K] (= fc2.weight^T), out / pre [M, N]
    dy = _mk((M, K), torch.bfloat16, 21)
    Wt = _mk((N, K), torch.bfloat16, 22)
    pre = (_mk((M, N), torch.float32, 23) * 3).to(torch.bfloat16)
    keep = pre.clone()

    def run(lean):
        monkeypatch.setenv("COUNTR_LEAN", "1" if lean else "0")
        monkeypatch.setenv("COUNTR_G256", "1" if lean else "0")
        out = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        a = _lib.GemmArgs()
        a.A, a.B, a.C, a.C2 = dy.data_ptr(), Wt.data_ptr(), out.data_ptr(), pre.data_ptr()
        a.lda, a.ldb, a.ldc = K, K, N
        a.M, a.N, a.K = M, N, K
        a.act = _lib.ACT_GELU_BWD
        a.out_bf16 = 1
        a.alpha = 1.0
        a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "gemm")
        torch.cuda.synchronize()
        return out
    lean, generic = run(True), run(False)
    assert torch.equal(pre, keep)
    x = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(dy.double() @ Wt.double().t())
    assert (lean.double() - x.grad).abs().max().item() <= 2e-2 * x.grad.abs().max().item()
    assert (lean.double() - generic.double()).abs().max().item() <= 1e-2 * x.grad.abs().max().item()
    # ... and against the two launches it replaces (GEMM -> bf16, then countr_gelu_bwd): one bf16 rounding apart
    monkeypatch.setenv("COUNTR_LEAN", "1")
    two = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    a = _lib.GemmArgs()
    a.A, a.B, a.C = dy.data_ptr(), Wt.data_ptr(), two.data_ptr()
    a.lda, a.ldb, a.ldc = K, K, N
    a.M, a.N, a.K = M, N, K
    a.out_bf16 = 1
    a.alpha = 1.0
    a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "gemm")
    _lib.check(hip.countr_gelu_bwd(two.data_ptr(), pre.data_ptr(), two.data_ptr(), M * N, 1, _stream()), "gelu_bwd")
    torch.cuda.synchronize()
    assert (lean.double() - two.double()).abs().max().item() <= 1.2e-2 * x.grad.abs().max().item()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_batched_and_splitk(hip, dt):
    # batched: q k^T per (b, h) read straight out of a packed qkv [B, N, 3, H, dh]
    Bsz, Ntok, H, dh = 2, 576, 3, 64
    qkv = _mk((Bsz, Ntok, 3, H, dh), dt, 7)
    scores = torch.empty((Bsz, H, Ntok, Ntok), device="cuda", dtype=torch.float32)
    a = _lib.GemmArgs()
    a.A = qkv.data_ptr()
    a.B = qkv.data_ptr() + H * dh * qkv.element_size()
    a.C = scores.data_ptr()
    a.lda = a.ldb = 3 * H * dh
    a.ldc = Ntok
    a.M = a.N = Ntok
    a.K = dh
    a.nbatch = Bsz * H; a.nb1 = H; a.splitk = 1
    a.sA0 = a.sB0 = Ntok * 3 * H * dh
    a.sA1 = a.sB1 = dh
    a.sC0 = H * Ntok * Ntok
    a.sC1 = Ntok * Ntok
    a.alpha = dh ** -0.5
    _lib.check(hip.countr_gemm(C.byref(a), 0 if dt == torch.float32 else 1, 0, 0, _stream()), "gemm")
    q = qkv[:, :, 0].permute(0, 2, 1, 3).double()
    k = qkv[:, :, 1].permute(0, 2, 1, 3).double()
    ref = (q @ k.transpose(-1, -2)) * dh ** -0.5
    torch.cuda.synchronize()
    assert (scores.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-5

    # split-K wgrad-style: dW[N,K] = dY[M,N]^T X[M,K]
    M, N, K = 4608, 256, 384
    dY = _mk((M, N), dt, 8)
    X = _mk((M, K), dt, 9)
    splitk = 4
    part = torch.empty((splitk, N, K), device="cuda", dtype=torch.float32)
    dW = torch.full((N, K), 1.0, device="cuda", dtype=torch.float32)
    a = _lib.GemmArgs()
    a.A, a.B, a.partial = dY.data_ptr(), X.data_ptr(), part.data_ptr()
    a.lda, a.ldb, a.ldc = N, K, K
    a.M, a.N, a.K = N, K, M
    a.nbatch = 1; a.nb1 = 1; a.splitk = splitk
    a.alpha = 1.0
    _lib.check(hip.countr_gemm(C.byref(a), 0 if dt == torch.float32 else 1, 1, 1, _stream()), "gemm")
    _lib.check(hip.countr_splitk_reduce(part.data_ptr(), dW.data_ptr(), splitk, N, K, 0, 1, None, None, _stream()), "reduce")
    torch.cuda.synchronize()
    ref = dY.double().t() @ X.double() + 1.0
    assert (dW.double() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Bsz,H,W,Cin,Cout", [(2, 24, 24, 128, 256), (1, 12, 20, 64, 128), (3, 8, 8, 256, 512),
                                              # odd geometry for the im2col fast paths: widths below / above / not dividing the 64-pixel
                                              # k-step, a single row, more channels than one 64-chunk, > 256 workgroups (plain kernel)
                                              (1, 5, 7, 64, 64), (2, 33, 17, 192, 72), (1, 1, 40, 128, 136), (1, 70, 66, 64, 256),
                                              (2, 96, 96, 64, 128),
                                              # maps smaller than one 64-pixel k-step with several k-steps (row counter wraps
                                              # more than once per step)
                                              (8, 4, 4, 64, 64), (8, 3, 8, 128, 64), (16, 2, 6, 64, 72)])
def test_conv3x3_implicit_gemm(hip, dt, Bsz, H, W, Cin, Cout):
    x = _mk((Bsz, Cin, H, W), dt, 10)            # logical NCHW
    w = _mk((Cout, Cin, 3, 3), dt, 11) * 0.1
    w = w.to(dt)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    w_ohwi = w.permute(0, 2, 3, 1).contiguous()   # [Cout][tap][Cin]
    bias = _mk((Cout,), torch.float32, 12)
    M = Bsz * H * W
    out = torch.empty((M, Cout), device="cuda", dtype=torch.float32)
    a = _lib.GemmArgs()
    a.A, a.B, a.C, a.bias = x_nhwc.data_ptr(), w_ohwi.data_ptr(), out.data_ptr(), bias.data_ptr()
    a.ldb, a.ldc = 9 * Cin, Cout
    a.M, a.N, a.K = M, Cout, 9 * Cin
    a.H, a.W, a.Cin = H, W, Cin
    a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.alpha = 1.0
    code = 0 if dt == torch.float32 else 1
    _lib.check(hip.countr_gemm(C.byref(a), code, 2, 0, _stream()), "conv fwd")
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1)
    ref_nhwc = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    assert (out.double() - ref_nhwc).abs().max().item() <= 2e-4 * ref_nhwc.abs().max().item()

    # wgrad: dW[co][tap][ci] = sum_p dY[p][co] * X[p+off][ci]   (split-K, then permute to OIHW)
    dy = _mk((M, Cout), dt, 13)
    splitk = 3
    part = torch.empty((splitk, Cout, 9 * Cin), device="cuda", dtype=torch.float32)
    dw = torch.zeros((Cout, Cin, 3, 3), device="cuda", dtype=torch.float32)
    a = _lib.GemmArgs()
    a.A, a.B, a.partial = dy.data_ptr(), x_nhwc.data_ptr(), part.data_ptr()
    a.lda, a.ldc = Cout, 9 * Cin
    a.M, a.N, a.K = Cout, 9 * Cin, M
    a.H, a.W, a.Cin = H, W, Cin
    a.nbatch = 1; a.nb1 = 1; a.splitk = splitk
    a.alpha = 1.0
    _lib.check(hip.countr_gemm(C.byref(a), code, 1, 3, _stream()), "conv wgrad")
    _lib.check(hip.countr_splitk_reduce(part.data_ptr(), dw.data_ptr(), splitk, Cout, 9 * Cin, 9, 0, None, None, _stream()), "reduce")
    torch.cuda.synchronize()
    xd = x.double().requires_grad_(False)
    wd = w.double().clone().requires_grad_(True)
    y = torch.nn.functional.conv2d(xd, wd, None, padding=1)
    gy = dy.double().reshape(Bsz, H, W, Cout).permute(0, 3, 1, 2)
    (gw,) = torch.autograd.grad(y, wd, gy)
    assert (dw.double() - gw).abs().max().item() <= 2e-4 * gw.abs().max().item()


def test_wgrad_with_fused_bias_gradient(hip):
    """bf16 split-K wgrad with rowsum_partial: db[n] = sum_m dY[m, n] comes out of the same GEMM (ones-fragment MFMA)."""
    M, N, K = 4608, 384, 256
    dY = _mk((M, N), torch.bfloat16, 21)
    X = _mk((M, K), torch.bfloat16, 22)
    for splitk in (1, 5):
        part = torch.empty((splitk, N, K), device="cuda", dtype=torch.float32)
        rs = torch.full((splitk, N), float("nan"), device="cuda", dtype=torch.float32)
        dW = torch.zeros((N, K), device="cuda"); db = torch.zeros(N, device="cuda")
        a = _lib.GemmArgs()
        a.A, a.B, a.partial, a.rowsum_partial = dY.data_ptr(), X.data_ptr(), part.data_ptr(), rs.data_ptr()
        a.lda, a.ldb, a.ldc = N, K, K
        a.M, a.N, a.K = N, K, M
        a.nbatch = 1; a.nb1 = 1; a.splitk = splitk; a.alpha = 1.0
        _lib.check(hip.countr_gemm(C.byref(a), 1, 1, 1, _stream()), "gemm")
        _lib.check(hip.countr_splitk_reduce(part.data_ptr(), dW.data_ptr(), splitk, N, K, 0, 0, rs.data_ptr(), db.data_ptr(), _stream()), "reduce")
        torch.cuda.synchronize()
        assert (dW.double() - dY.double().t() @ X.double()).abs().max().item() <= 2e-4 * (dY.double().t() @ X.double()).abs().max().item()
        ref_b = dY.double().sum(0)
        assert (db.double() - ref_b).abs().max().item() <= 1e-4 * ref_b.abs().max().item() + 1e-4


def test_gemm_randomised_shapes_strides_and_options(hip):
    """Seeded sweep over operand modes, ragged and padded shapes (leading dimensions larger than the row), both dtypes, bias /
    residual / GELU / output-type options, split-K and grid sizes on both sides of the kernel-selection thresholds (256 workgroups:
    wave-specialised vs plain; K a multiple of 64 or not: uniform-base vs per-lane addressing)."""
    import random
    rng = random.Random(1234)
    checked = 0
    for it in range(72):
        dt = torch.bfloat16 if it % 3 else torch.float32
        code = 1 if dt == torch.bfloat16 else 0
        chunk = 8 if code else 4
        ma, mb = rng.choice([(0, 0), (0, 1), (1, 1), (1, 0)])
        big = it % 9 == 0
        M = rng.choice([8, 24, 72, 200, 576, 1000]) if not big else rng.choice([2304, 4608])
        N = rng.choice([8, 64, 136, 256, 512, 768]) if not big else rng.choice([768, 1536])
        K = rng.choice([8, 64, 96, 128, 192, 320, 768])
        M, N = -(-M // chunk) * chunk, -(-N // chunk) * chunk
        pad = rng.choice([0, chunk, 5 * chunk])
        A, B = _mk((M, K), dt, 100 + it), _mk((N, K), dt, 200 + it)
        As = torch.zeros((M, K + pad) if ma == 0 else (K, M + pad), device="cuda", dtype=dt)
        Bs = torch.zeros((N, K + pad) if mb == 0 else (K, N + pad), device="cuda", dtype=dt)
        (As[:, :K] if ma == 0 else As[:, :M]).copy_(A if ma == 0 else A.t())
        (Bs[:, :K] if mb == 0 else Bs[:, :N]).copy_(B if mb == 0 else B.t())
        a = _lib.GemmArgs()
        a.alpha = rng.choice([1.0, 0.5]); a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        a.A, a.B = As.data_ptr(), Bs.data_ptr()
        a.lda, a.ldb = As.shape[1], Bs.shape[1]
        a.M, a.N, a.K = M, N, K
        ref = a.alpha * (A.double() @ B.double().t())
        if it % 4 == 3:   # split-K slabs + deterministic reduce
            sk = rng.choice([1, 2, 3, 5])
            part = torch.empty((sk, M, N), device="cuda")
            out = torch.zeros((M, N), device="cuda")
            a.partial, a.splitk, a.ldc, a.alpha = part.data_ptr(), sk, N, 1.0
            ref = A.double() @ B.double().t()
            _lib.check(hip.countr_gemm(C.byref(a), code, ma, mb, _stream()), "gemm")
            _lib.check(hip.countr_splitk_reduce(part.data_ptr(), out.data_ptr(), sk, M, N, 0, 0, None, None, _stream()), "reduce")
            tol = 2e-4
        else:
            obf = code == 1 and rng.random() < 0.5
            ldc = N + rng.choice([0, 8])
            out = torch.zeros((M, ldc), device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
            a.C, a.ldc, a.out_bf16 = out.data_ptr(), ldc, int(obf)
            opt = rng.choice(["plain", "bias", "bias+gelu", "bias+resid", "resid"])
            if "bias" in opt:
                bias = _mk((N,), torch.float32, 300 + it)
                a.bias = bias.data_ptr()
                ref = ref + bias.double()
            if "gelu" in opt:
                a.act = _lib.ACT_GELU
                ref = torch.nn.functional.gelu(ref)
            if "resid" in opt:
                rmod = rng.choice([0, max(chunk, M // 2)])
                res = _mk((rmod if rmod else M, N), torch.float32, 400 + it)
                a.resid, a.ldres, a.res_mod = res.data_ptr(), N, rmod
                ref = ref + (res.double() if not rmod else res.double()[torch.arange(M, device="cuda") % rmod])
            _lib.check(hip.countr_gemm(C.byref(a), code, ma, mb, _stream()), "gemm")
            out = out[:, :N]
            tol = 2e-2 if obf else (3e-3 if ("gelu" in opt and code) else 2e-4)   # bf16 GELU uses the 1.5e-7-accurate fast erf
        torch.cuda.synchronize()
        err = (out.double() - ref).abs().max().item()
        assert err <= tol * max(ref.abs().max().item(), 1e-3) + 1e-5, (it, dt, ma, mb, M, N, K, pad, err)
        checked += 1
    assert checked == 72


@pytest.mark.parametrize("M,N,K", [(200, 136, 96), (1000, 768, 128), (4608, 512, 64), (333 * 8, 2304, 64)])
def test_gemm_staged_and_direct_epilogues(hip, M, N, K):
    """bf16 kernels store finished values either as whole rows staged through LDS (16-byte aligned C rows) or directly from the
    accumulator layout (any other leading dimension / base alignment): both must give the same values, honour M / N tails and
    leave the padding of a wider C untouched; same for the fp32 residual read and the training pre-activation copy (C2)."""
    A, B = _mk((M, K), torch.bfloat16, 11), _mk((N, K), torch.bfloat16, 12)
    bias = _mk((N,), torch.float32, 13)
    res = _mk((M, N), torch.float32, 14)
    base = A.double() @ B.double().t()
    for opt in ("plain", "bias", "bias+gelu", "bias+gelu+c2", "bias+resid", "resid"):
        for obf in ((1,) if "gelu" in opt else (0,) if "resid" in opt else (0, 1)):
            odt = torch.bfloat16 if obf else torch.float32
            for ldc, off in ((N, 0), (N + 8, 0), (N + 4, 0), (N + 8, 4), (N + 4, 4)):
                store = torch.full((M * ldc + 8,), 7.0, device="cuda", dtype=odt)
                store2 = torch.full((M * ldc + 8,), 7.0, device="cuda", dtype=odt)
                a = _lib.GemmArgs()
                a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
                a.A, a.B, a.lda, a.ldb = A.data_ptr(), B.data_ptr(), K, K
                a.M, a.N, a.K = M, N, K
                a.C, a.ldc, a.out_bf16 = store.data_ptr() + off * store.element_size(), ldc, obf
                ref = base
                if "bias" in opt:
                    a.bias = bias.data_ptr(); ref = ref + bias.double()
                pre = ref
                if "c2" in opt:
                    a.C2 = store2.data_ptr() + off * store2.element_size()
                if "gelu" in opt:
                    a.act = _lib.ACT_GELU; ref = torch.nn.functional.gelu(ref)
                if "resid" in opt:
                    a.resid, a.ldres = res.data_ptr(), N; ref = ref + res.double()
                _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "gemm")
                torch.cuda.synchronize()
                tol = 2e-2 if obf else 2e-4
                for buf, want in ((store, ref), (store2, pre if "c2" in opt else None)):
                    view = buf[off:off + M * ldc].view(M, ldc)
                    if want is None:
                        assert (buf == 7.0).all()
                        continue
                    err = (view[:, :N].double() - want).abs().max().item()
                    assert err <= tol * want.abs().max().item() + 1e-5, (opt, obf, ldc, off, err)
                    assert (view[:, N:] == 7.0).all() and (buf[:off] == 7.0).all() and (buf[off + M * ldc:] == 7.0).all(), (opt, obf, ldc, off)


def test_reduce_table_matches_individual_reductions(hip):
    """countr_reduce_table (one launch for many deferred slab sums) against countr_splitk_reduce / plain torch sums: split-K slabs
    with and without the conv tap permutation, accumulate on and off, a strided many-slab ("wide") entry as the LayerNorm
    dgamma / dbeta partials use, odd counts."""
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *shape: torch.rand(shape, generator=g).cuda() - 0.5
    entries, keep, blk, checks = [], [], 0, []
    for (sk, M, N, taps, acc) in ((3, 40, 72, 0, 0), (7, 8, 9 * 64, 9, 1), (1, 130, 4, 0, 1), (5, 1, 1000, 0, 0)):
        part = rnd(sk, M, N)
        out0 = rnd(M, N)
        out = out0.clone()
        ref = out0.clone()
        _lib.check(hip.countr_splitk_reduce(part.data_ptr(), ref.data_ptr(), sk, M, N, taps, acc, None, None, _stream()), "reduce")
        entries.append([part.data_ptr(), out.data_ptr(), sk | (acc << 32), M * N, M * N, N, taps, blk])
        blk += -(-(M * N) // 256)
        keep += [part, out]
        checks.append((out, ref))
    for (nb, D, acc) in ((256, 72, 0), (100, 513, 1)):          # workspace rows {dgamma[D], dbeta[D]}: two strided wide entries
        ws = rnd(nb, 2 * D)
        for half in (0, 1):
            out0 = rnd(D)
            out = out0.clone()
            ref = ws[:, half * D:(half + 1) * D].double().sum(0).float() + (out0 if acc else 0)
            entries.append([ws.data_ptr() + 4 * half * D, out.data_ptr(), nb | (acc << 32) | (1 << 33), 2 * D, D, 0, 0, blk])
            blk += -(-D // 16)
            keep += [ws, out]
            checks.append((out, ref))
    firsts = [e[7] for e in entries] + [blk]
    owner = torch.cat([torch.full((firsts[i + 1] - firsts[i],), i, dtype=torch.int32) for i in range(len(entries))])
    if owner.numel() & 1:
        owner = torch.cat([owner, owner.new_zeros(1)])
    tab = torch.cat([torch.tensor(entries, dtype=torch.int64).reshape(-1), owner.view(torch.int64)]).cuda()
    _lib.check(hip.countr_reduce_table(tab.data_ptr(), len(entries), blk, _stream()), "reduce_table")
    torch.cuda.synchronize()
    for i, (out, ref) in enumerate(checks):
        assert (out - ref).abs().max().item() <= 1e-5 * max(ref.abs().max().item(), 1.0), i


@pytest.mark.parametrize("obf", [0, 1])
def test_splitk_finish_matches_unsplit_conv(hip, obf):
    """Forward-type split-K (few tiles, long K: the 24x24 / 8x8 convolutions of the bf16 engine): implicit-GEMM conv with
    splitk = 4 into fp32 partials + countr_splitk_finish (sum, bias, output dtype) against the same conv in one launch."""
    Bn, H, W, Cin, Cout = 3, 8, 8, 512, 256
    x = _mk((Bn, H, W, Cin), torch.bfloat16, 31)
    w = _mk((Cout, 9 * Cin), torch.bfloat16, 32) * 0.05
    bias = _mk((Cout,), torch.float32, 33)
    M, K = Bn * H * W, 9 * Cin
    odt = torch.bfloat16 if obf else torch.float32
    ref = torch.empty((M, Cout), device="cuda", dtype=odt)
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.C, a.bias = x.data_ptr(), w.data_ptr(), ref.data_ptr(), bias.data_ptr()
    a.ldb, a.ldc, a.M, a.N, a.K, a.H, a.W, a.Cin, a.out_bf16 = K, Cout, M, Cout, K, H, W, Cin, obf
    _lib.check(hip.countr_gemm(C.byref(a), 1, 2, 0, _stream()), "gemm")
    part = torch.full((4, M, Cout), float("nan"), device="cuda")
    out = torch.full((M, Cout), float("nan"), device="cuda", dtype=odt)
    b = _lib.GemmArgs(); b.alpha = 1.0; b.nbatch = 1; b.nb1 = 1; b.splitk = 4
    b.A, b.B, b.partial = x.data_ptr(), w.data_ptr(), part.data_ptr()
    b.ldb, b.ldc, b.M, b.N, b.K, b.H, b.W, b.Cin = K, Cout, M, Cout, K, H, W, Cin
    _lib.check(hip.countr_gemm(C.byref(b), 1, 2, 0, _stream()), "gemm split")
    _lib.check(hip.countr_splitk_finish(part.data_ptr(), out.data_ptr(), bias.data_ptr(), 4, M, Cout, obf, _stream()), "finish")
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    scale = ref.float().abs().max().item()
    assert (out.float() - ref.float()).abs().max().item() <= (8e-3 if obf else 2e-5) * scale


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (256, 384, 192), (4608, 512, 512), (1152, 768, 3072), (4608, 1536, 256), (4736, 1024, 128),
                                   (2304, 3072, 768), (576, 768, 768), (1728, 512, 512), (2880, 3072, 768), (40, 128, 128),
                                   (4608, 2304, 768), (4544, 2304, 128), (3520, 3072, 192)])   # (the last three: the 192x256 form, 96x64 wave tiles)
@pytest.mark.parametrize("epi", ["bf16", "gelu", "gelu_pre", "res", "resmod"])
def test_lean_linear_matches_fp64_and_generic(hip, monkeypatch, M, N, K, epi):
    """csrc/linear.hip (full-tile nn.Linear forward: 32x32x16 MFMA, staged epilogue, sigmoid-polynomial GELU) through countr_gemm:
    every epilogue; the wave-specialised 128x128 form (<= 256 tiles) and the 256x128 form (bigger grids); row counts that are not a
    multiple of the tile (B = 1, 3, 5 images: 576, 1728, 2880 rows; 40 rows: rows beyond M stage zeros and are never stored),
    against torch fp64 and against gemm_kernel on the
    same inputs (COUNTR_LEAN=0).  Operands are bf16-exact, so the only error is fp32 accumulation order + the output rounding
    (+ <= 2.6e-5 absolute of the GELU fit)."""
    A = _mk((M, K), torch.bfloat16, 21)
    W = (_mk((N, K), torch.float32, 22) * 0.25).to(torch.bfloat16)
    bias = _mk((N,), torch.float32, 23)
    obf = epi in ("bf16", "gelu", "gelu_pre")
    rmod = (8 if M == 40 else (64 if M % 128 else 128)) if epi == "resmod" else 0      # (a divisor of M)
    resid = None if obf else _mk((rmod if rmod else M, N), torch.float32, 24)
    z = A.double() @ W.double().t() + bias.double()
    ref = torch.nn.functional.gelu(z) if epi.startswith("gelu") else z
    if resid is not None:
        ref = ref + (resid.double().repeat(M // rmod, 1) if rmod else resid.double())
    outs = []
    for lean in ("1", "0"):
        monkeypatch.setenv("COUNTR_LEAN", lean)
        out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
        pre = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16) if epi == "gelu_pre" else None
        a = _lib.GemmArgs()
        a.A, a.B, a.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
        a.C2 = pre.data_ptr() if pre is not None else None
        a.bias = bias.data_ptr()
        a.resid = resid.data_ptr() if resid is not None else None
        a.lda, a.ldb, a.ldc, a.ldres = K, K, N, N
        a.M, a.N, a.K = M, N, K
        a.res_mod = rmod
        a.act = 1 if epi.startswith("gelu") else 0
        a.out_bf16 = int(obf)
        a.alpha = 1.0
        a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "gemm")
        torch.cuda.synchronize()
        tol = 4e-3 if obf else 2e-5                      # bf16 rounding (2^-9 relative to the element, <= that of the maximum) / fp32 order
        scale = ref.abs().max().item()
        assert torch.isfinite(out.float()).all()
        assert (out.double() - ref).abs().max().item() <= tol * scale + 3e-5, (lean, (out.double() - ref).abs().max().item(), scale)
        if pre is not None:
            assert (pre.double() - z).abs().max().item() <= 4e-3 * z.abs().max().item()
        outs.append(out)
    # the two kernels differ by accumulation order (and the GELU form): far below one bf16 ulp of the largest element
    assert (outs[0].double() - outs[1].double()).abs().max().item() <= (8e-3 if obf else 2e-5) * ref.abs().max().item()


def test_lean_linear_in_place_residual(hip):
    """proj / fc2 write the residual stream in place (C == resid): every element is read before it is written by the same lane."""
    M, N, K = 4608, 768, 768
    A = _mk((M, K), torch.bfloat16, 31)
    W = (_mk((N, K), torch.float32, 32) * 0.25).to(torch.bfloat16)
    bias = _mk((N,), torch.float32, 33)
    x = _mk((M, N), torch.float32, 34)
    ref = x.double() + A.double() @ W.double().t() + bias.double()
    a = _lib.GemmArgs()
    a.A, a.B, a.C, a.resid, a.bias = A.data_ptr(), W.data_ptr(), x.data_ptr(), x.data_ptr(), bias.data_ptr()
    a.lda, a.ldb, a.ldc, a.ldres = K, K, N, N
    a.M, a.N, a.K = M, N, K
    a.alpha = 1.0
    a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "gemm")
    torch.cuda.synchronize()
    assert (x.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("Bsz,H,W,Cin,Cout,use_bias", [(2, 96, 96, 256, 256, True), (8, 48, 48, 512, 256, True), (4, 48, 96, 256, 256, False),
                                                   (32, 24, 24, 64, 256, True), (2, 96, 96, 256, 512, False), (3, 100, 100, 64, 256, True)])
def test_lean_conv3x3_matches_fp64_and_generic(hip, monkeypatch, Bsz, H, W, Cin, Cout, use_bias):
    """The density-head / exemplar 3x3 convolutions on the big maps (forward, and dgrad through the dgrad-form weights) run the lean
    kernel of linear.hip with im2row LDS-DMA addressing (256x128 tiles, 8 compute + 4 loader waves): against torch conv2d in fp64 and
    against gemm_kernel (COUNTR_LEAN_CONV=0) on the same bf16 inputs -- zero padding at every image border, tiles that span image
    boundaries (H x W not a multiple of 256), rows narrower than a 32-pixel staging pass (W = 24), Cin = 64 (one k-tile per tap) and
    512, with and without bias (dgrad)."""
    x = _mk((Bsz, H, W, Cin), torch.bfloat16, 51)                 # NHWC
    w = (_mk((Cout, 3, 3, Cin), torch.float32, 52) * 0.1).to(torch.bfloat16)   # OHWI = [Cout][tap][Cin]
    bias = _mk((Cout,), torch.float32, 53) if use_bias else None
    M, K = Bsz * H * W, 9 * Cin
    assert -(-M // 128) * (Cout // 128) > 256
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double() if use_bias else None, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    outs = []
    monkeypatch.setenv("COUNTR_G256", "0")            # (the 256 x 256 kernel has its own test below)
    for lean in ("1", "0"):
        monkeypatch.setenv("COUNTR_LEAN_CONV", lean)
        out = torch.full((M, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
        a = _lib.GemmArgs()
        a.A, a.B, a.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
        a.bias = bias.data_ptr() if use_bias else None
        a.ldb, a.ldc = K, Cout
        a.M, a.N, a.K = M, Cout, K
        a.H, a.W, a.Cin = H, W, Cin
        a.out_bf16 = 1
        a.alpha = 1.0
        a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        _lib.check(hip.countr_gemm(C.byref(a), 1, 2, 0, _stream()), "conv")
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        assert (out.double() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item(), lean
        outs.append(out)
    assert (outs[0].double() - outs[1].double()).abs().max().item() <= 8e-3 * ref.abs().max().item()


@pytest.mark.parametrize("form", ["1", "2", "3"])
@pytest.mark.parametrize("Bsz,H,W,Cin,Cout,sk,with_bias", [(2, 96, 96, 256, 256, 7, True), (4, 48, 48, 256, 256, 5, True), (8, 24, 24, 512, 256, 3, True),
                                                          (3, 32, 48, 128, 128, 2, False), (1, 64, 64, 256, 384, 40, True), (2, 8, 16, 128, 128, 1, True),
                                                          (2, 64, 128, 128, 256, 5, True), (3, 5, 64, 256, 128, 4, True), (1, 192, 192, 256, 256, 21, True),
                                                          (2, 3, 192, 128, 128, 1, False), (1, 4, 96, 128, 128, 3, True), (2, 2, 96, 256, 128, 1, True)])
def test_lean_conv_wgrad_matches_fp64_and_generic(hip, monkeypatch, form, Bsz, H, W, Cin, Cout, sk, with_bias):
    """Weight (+ bias) gradient of the 3x3 convolutions as the (COL, IM2COL) split-K GEMM: the lean kernel of conv_wgrad.hip (both maps
    staged K-major by LDS-DMA, transposing fragment reads, bias-gradient MFMAs dealt over the waves) against torch fp64 autograd on the
    same bf16 maps and against gemm_kernel (COUNTR_LEAN_WGRAD=0).  Rows narrower than a 64-pixel k-tile (W = 48, 24, 16), k-tiles
    spanning image boundaries, 128x128 and 128x256 tiles, more slabs than k-tiles per slab allow (sk = 40 of 64 k-tiles: empty slabs
    must come out as zeros), a single slab, Cin = 128 / 512.  Form 3 (the three taps of a kernel row per workgroup, one staged row
    segment read at three offsets; the loader waves take the bias gradient) applies where W % 64 == 0 -- one, two and three k-tiles per
    image row, images of 3 and 5 rows (every k-tile next to a padded row), the 192 x 192 layer itself -- and, with 96-pixel k-tiles (six
    k-steps, one image row per k-tile), where W % 96 == 0: the 96 x 96 layer, 2- and 4-row images; it falls back elsewhere."""
    dy = _mk((Bsz, H, W, Cout), torch.bfloat16, 71)
    x = _mk((Bsz, H, W, Cin), torch.bfloat16, 72)
    P, N = Bsz * H * W, 9 * Cin
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(False)
    wz = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    bz = torch.zeros(Cout, dtype=torch.float64, device="cuda", requires_grad=True)
    y = torch.nn.functional.conv2d(xd, wz, bz, padding=1)
    y.backward(dy.double().permute(0, 3, 1, 2))
    ref_w = wz.grad.permute(0, 2, 3, 1).reshape(Cout, N)          # OHWI
    ref_b = bz.grad
    res = []
    for lean in ("1", "0"):
        monkeypatch.setenv("COUNTR_LEAN_WGRAD", lean)
        monkeypatch.setenv("COUNTR_LEAN_WGRAD_FORM", form)
        a = _lib.GemmArgs()
        a.A, a.B = dy.data_ptr(), x.data_ptr()
        a.lda, a.ldc = Cout, N
        a.M, a.N, a.K = Cout, N, P
        a.H, a.W, a.Cin = H, W, Cin
        a.alpha = 1.0
        a.nbatch = 1; a.nb1 = 1; a.splitk = sk
        slabs = hip.countr_gemm_rowsum_slabs(C.byref(a), 1, 1, 3)
        per_slab = 3 * (Cin // 128) if (form == "3" and (W % 64 == 0 or W % 96 == 0)) else N // 128
        assert slabs == (sk * per_slab if lean == "1" else sk)
        part = torch.full((sk, Cout, N), float("nan"), device="cuda", dtype=torch.float32)
        rs = torch.full((slabs, Cout), float("nan"), device="cuda", dtype=torch.float32)
        a.partial = part.data_ptr()
        if with_bias:
            a.rowsum_partial, a.rowsum_slabs = rs.data_ptr(), slabs
        _lib.check(hip.countr_gemm(C.byref(a), 1, 1, 3, _stream()), "wgrad")
        torch.cuda.synchronize()
        assert torch.isfinite(part).all()
        gw = part.double().sum(0)
        assert (gw - ref_w).abs().max().item() <= 2e-5 * ref_w.abs().max().item() + 1e-9, lean
        if with_bias:
            assert torch.isfinite(rs).all()
            gb = rs.double().sum(0)
            assert (gb - ref_b).abs().max().item() <= 2e-5 * ref_b.abs().max().item() + 1e-9, lean
        res.append(gw)
    assert (res[0] - res[1]).abs().max().item() <= 4e-5 * ref_w.abs().max().item() + 1e-9
    # legacy layout request (rowsum_slabs = 0) must stay on the generic kernel and fill exactly [splitk][M]
    if with_bias:
        monkeypatch.setenv("COUNTR_LEAN_WGRAD", "1")
        rs = torch.full((sk + 1, Cout), float("nan"), device="cuda", dtype=torch.float32)
        a.rowsum_partial, a.rowsum_slabs = rs.data_ptr(), 0
        _lib.check(hip.countr_gemm(C.byref(a), 1, 1, 3, _stream()), "wgrad")
        torch.cuda.synchronize()
        assert torch.isfinite(rs[:sk]).all() and torch.isnan(rs[sk]).all()
        assert (rs[:sk].double().sum(0) - ref_b).abs().max().item() <= 2e-5 * ref_b.abs().max().item() + 1e-9


@pytest.mark.parametrize("R,N,K,lddy,ldx,sk,with_bias", [(4608, 512, 512, 512, 512, 16, True), (2304, 3072, 768, 3072, 768, 1, True), (2304, 768, 3072, 768, 3072, 2, True),
                                                         (4608, 512, 512, 1536, 512, 8, False), (1152, 256, 384, 256, 512, 3, True), (64, 128, 128, 128, 128, 1, True),
                                                         (4608, 2048, 512, 2048, 512, 4, True)])
def test_lean_linear_wgrad_matches_fp64_and_generic(hip, monkeypatch, R, N, K, lddy, ldx, sk, with_bias):
    """Weight (+ bias) gradient of an nn.Linear as the (COL, COL) split-K GEMM dW[N][K] = dy^T x on the lean kernel of conv_wgrad.hip
    (LIN form: both operands staged K-major as they lie in memory, transposing reads) against fp64 and gemm_kernel
    (COUNTR_LEAN_LWGRAD=0): operand pitches larger than the matrix (a q / k / v slice of a packed qkv gradient, an x with a wider
    row), 128x128 and 128x256 tiles, one slab, a single k-tile per slab."""
    dyf = _mk((R, lddy), torch.bfloat16, 81)
    xf = _mk((R, ldx), torch.bfloat16, 82)
    ref_w = dyf[:, :N].double().t() @ xf[:, :K].double()
    ref_b = dyf[:, :N].double().sum(0)
    res = []
    for lean in ("1", "0"):
        monkeypatch.setenv("COUNTR_LEAN_LWGRAD", lean)
        a = _lib.GemmArgs()
        a.A, a.B = dyf.data_ptr(), xf.data_ptr()
        a.lda, a.ldb, a.ldc = lddy, ldx, K
        a.M, a.N, a.K = N, K, R
        a.alpha = 1.0
        a.nbatch = 1; a.nb1 = 1; a.splitk = sk
        slabs = hip.countr_gemm_rowsum_slabs(C.byref(a), 1, 1, 1)
        assert slabs == (sk * (K // 128) if lean == "1" else sk)
        part = torch.full((sk, N, K), float("nan"), device="cuda", dtype=torch.float32)
        rs = torch.full((slabs, N), float("nan"), device="cuda", dtype=torch.float32)
        a.partial = part.data_ptr()
        if with_bias:
            a.rowsum_partial, a.rowsum_slabs = rs.data_ptr(), slabs
        _lib.check(hip.countr_gemm(C.byref(a), 1, 1, 1, _stream()), "wgrad")
        torch.cuda.synchronize()
        assert torch.isfinite(part).all()
        gw = part.double().sum(0)
        assert (gw - ref_w).abs().max().item() <= 2e-5 * ref_w.abs().max().item() + 1e-9, lean
        if with_bias:
            assert torch.isfinite(rs).all()
            assert (rs.double().sum(0) - ref_b).abs().max().item() <= 2e-5 * ref_b.abs().max().item() + 1e-9, lean
        res.append(gw)
    assert (res[0] - res[1]).abs().max().item() <= 4e-5 * ref_w.abs().max().item() + 1e-9


@pytest.mark.parametrize("R,shapes,sk", [
    (2304, [(768, 3072), (3072, 768), (768, 768), (2304, 768)], 1),        # MAE encoder block at 8 images: fc2, fc1, proj, qkv = 216 tiles of 128x256
    (4608, [(512, 2048), (2048, 512), (512, 512), (512, 512)], 2),         # finetune decoder block: fc2, fc1, proj, wq = 96 tiles x 2 slabs
    (1152, [(512, 2048), (1536, 512)], 3),                                 # two launches, three slabs
    (2304, [(768, 3072), (384, 384), (768, 768)], 1),                      # a 384-column problem: the whole group on 128x128 tiles
    (4608, [(512, 2048), (2048, 512), (512, 512), (512, 512), (512, 512), (1536, 512), (512, 768)], 2),   # a whole decoder block + decoder_embed: 7 launches
])
def test_grouped_linear_wgrads_equal_the_separate_launches(hip, R, shapes, sk):
    """countr_gemm_group: the weight (+ bias) gradients of a block's nn.Linear layers in ONE launch == the same launches through
    countr_gemm one by one, bit for bit (partials and row-sum partials), and == fp64.  Reference: autograd of Mlp / Attention /
    CrossAttention (models_crossvit.py:46-128)."""
    n = len(shapes)
    dys = [_mk((R, N), torch.bfloat16, 91 + i) for i, (N, K) in enumerate(shapes)]
    xs = [_mk((R, K), torch.bfloat16, 95 + i) for i, (N, K) in enumerate(shapes)]

    def make():
        arr = (_lib.GemmArgs * n)()
        keep = []
        for i, (N, K) in enumerate(shapes):
            a = arr[i]
            a.A, a.B = dys[i].data_ptr(), xs[i].data_ptr()
            a.lda, a.ldb, a.ldc = N, K, K
            a.M, a.N, a.K = N, K, R
            a.alpha = 1.0
            a.nbatch = 1; a.nb1 = 1; a.splitk = sk
            slabs = hip.countr_gemm_rowsum_slabs(C.byref(a), 1, 1, 1)
            part = torch.full((sk, N, K), float("nan"), device="cuda", dtype=torch.float32)
            rs = torch.full((slabs, N), float("nan"), device="cuda", dtype=torch.float32)
            a.partial, a.rowsum_partial, a.rowsum_slabs = part.data_ptr(), rs.data_ptr(), slabs
            keep.append((part, rs))
        return arr, keep
    arr, grouped = make()
    tiles = hip.countr_gemm_group_tiles(arr, n, 1, 1, 1)
    wide = all(K % 256 == 0 for _, K in shapes)
    assert tiles == sum((N // 128) * (K // (256 if wide else 128)) for N, K in shapes)
    _lib.check(hip.countr_gemm_group(arr, n, 1, 1, 1, _stream()), "group")
    arr2, separate = make()
    for i in range(n):
        _lib.check(hip.countr_gemm(C.byref(arr2[i]), 1, 1, 1, _stream()), "wgrad")
    torch.cuda.synchronize()
    for i, (N, K) in enumerate(shapes):
        ref_w = dys[i].double().t() @ xs[i].double()
        ref_b = dys[i].double().sum(0)
        assert torch.isfinite(grouped[i][0]).all() and torch.isfinite(grouped[i][1]).all()
        assert (grouped[i][0].double().sum(0) - ref_w).abs().max().item() <= 2e-5 * ref_w.abs().max().item() + 1e-9, i
        assert (grouped[i][1].double().sum(0) - ref_b).abs().max().item() <= 2e-5 * ref_b.abs().max().item() + 1e-9, i
        assert torch.equal(grouped[i][0], separate[i][0]), i
        assert torch.equal(grouped[i][1], separate[i][1]), i
    # what does not qualify runs one by one: fp32 launches, a single launch
    assert hip.countr_gemm_group_tiles(arr, n, 0, 1, 1) == 0 and hip.countr_gemm_group_tiles(arr, 1, 1, 1, 1) == 0
    assert hip.countr_gemm_group(arr, 11, 1, 1, 1, _stream()) != 0


@pytest.mark.parametrize("M,N2", [(4608, 1536), (576, 1536), (14976, 1536), (4608, 2304), (4400, 2304)])   # N2 = 2304 at B = 8: the 192x256 form
@pytest.mark.parametrize("act", [0, 1])
def test_lean_linear_layernorm_folding(hip, M, N2, act):
    """LayerNorm folded into the Linear layers around it (frozen encoder): a producer GEMM (bias + fp32 residual) also emits the bf16
    copy of its output and the {sum, sum of squares} of every 64-column block of each output row; a consumer GEMM on that copy with
    gamma folded into its weights applies rstd (acc - mean colsum) + (b + W beta).  Reference: torch fp64
    LayerNorm(x) W^T + b (+ GELU) with x = the producer's fp32 output.  What the fold changes numerically: the GEMM operand is the
    rounded RAW row instead of the rounded normalised row, so its rounding error relative to sigma grows by sqrt(1 + (mean / sigma)^2).
    Rows with mean ~ 0 (the transformer's case: per-token means of the residual stream are a fraction of sigma) must match the unfolded
    path's error; every 7th row here has |mean| = 12 sigma and is held to that factor (documented limit, COUNTR_LN_FOLD=0 is the way out)."""
    D = 768
    eps = 1e-6
    A0 = _mk((M, D), torch.bfloat16, 61)
    W0 = (_mk((D, D), torch.float32, 62) * 0.05).to(torch.bfloat16)
    b0 = _mk((D,), torch.float32, 63)
    x_in = _mk((M, D), torch.float32, 64) * 2
    x_in[::7] += 15.0                                             # rows with mean >> sigma
    x = x_in.clone()
    xb = torch.full((M, D), float("nan"), device="cuda", dtype=torch.bfloat16)
    st = torch.full((M, D // 64, 2), float("nan"), device="cuda", dtype=torch.float32)
    a = _lib.GemmArgs()
    a.A, a.B, a.C, a.resid, a.bias = A0.data_ptr(), W0.data_ptr(), x.data_ptr(), x.data_ptr(), b0.data_ptr()
    a.lda, a.ldb, a.ldc, a.ldres = D, D, D, D
    a.M, a.N, a.K = M, D, D
    a.alpha = 1.0
    a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.ln_xcopy, a.ln_stats_out = xb.data_ptr(), st.data_ptr()
    _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "producer")
    torch.cuda.synchronize()
    xr = x_in.double() + A0.double() @ W0.double().t() + b0.double()
    assert (x.double() - xr).abs().max().item() <= 2e-5 * xr.abs().max().item()
    assert torch.equal(xb, x.to(torch.bfloat16))                                            # the copy is the rounded fp32 output
    blocks = x.double().view(M, D // 64, 64)
    assert (st[..., 0].double() - blocks.sum(-1)).abs().max().item() <= 1e-4 * blocks.sum(-1).abs().max().item()
    assert (st[..., 1].double() - (blocks ** 2).sum(-1)).abs().max().item() <= 1e-5 * (blocks ** 2).sum(-1).abs().max().item()
    # consumer
    gamma = _mk((D,), torch.float32, 65) * 0.5 + 1.0
    beta = _mk((D,), torch.float32, 66) * 0.2
    W1 = _mk((N2, D), torch.float32, 67) * 0.05
    b1 = _mk((N2,), torch.float32, 68)
    Wf = (W1 * gamma[None, :]).to(torch.bfloat16)
    colsum = Wf.float().sum(1).contiguous()
    bf = (b1 + W1 @ beta).contiguous()
    out = torch.full((M, N2), float("nan"), device="cuda", dtype=torch.bfloat16)
    c = _lib.GemmArgs()
    c.A, c.B, c.C, c.bias = xb.data_ptr(), Wf.data_ptr(), out.data_ptr(), bf.data_ptr()
    c.lda, c.ldb, c.ldc = D, D, N2
    c.M, c.N, c.K = M, N2, D
    c.act = act
    c.out_bf16 = 1
    c.alpha = 1.0
    c.nbatch = 1; c.nb1 = 1; c.splitk = 1
    c.ln_stats, c.ln_colsum, c.ln_nblk, c.ln_eps = st.data_ptr(), colsum.data_ptr(), D // 64, eps
    _lib.check(hip.countr_gemm(C.byref(c), 1, 0, 0, _stream()), "consumer")
    torch.cuda.synchronize()
    ln = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), eps)
    ref = ln @ W1.double().t() + b1.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    err = (out.double() - ref).abs()
    # what the unfolded path costs: LN output rounded to bf16, weights rounded to bf16, bf16 output
    base = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), eps).to(torch.bfloat16).double() @ W1.to(torch.bfloat16).double().t() + b1.double()
    if act:
        base = torch.nn.functional.gelu(base)
    base_err = (base.to(torch.bfloat16).double() - ref).abs()
    assert torch.isfinite(out.float()).all()
    kappa = torch.sqrt(1 + (x.double().mean(1) / x.double().std(1, unbiased=False)) ** 2)          # per-row amplification of the operand rounding
    plain = kappa < 1.5
    assert plain.float().mean().item() > 0.8 and kappa.max().item() > 8
    bmax, brms = base_err.max().item(), base_err.pow(2).mean().sqrt().item()
    assert err[plain].max().item() <= 2.5 * bmax, (err[plain].max().item(), bmax)
    assert err[plain].pow(2).mean().sqrt().item() <= 1.5 * brms + 1e-4, (err[plain].pow(2).mean().sqrt().item(), brms)
    assert (err.max(1).values / kappa).max().item() <= 2.5 * bmax, ((err.max(1).values / kappa).max().item(), bmax)
    # the generic kernel refuses the fields loudly (no silent un-normalised result)
    c.N = 1536 + 4
    assert hip.countr_gemm(C.byref(c), 1, 0, 0, _stream()) != 0


@pytest.mark.parametrize("M,N,K", [(4608, 3072, 768), (2880, 3072, 768), (256, 256, 256), (1000, 512, 384), (4608, 2304, 768), (8192, 1024, 1280),
                                   (40, 256, 256), (4608, 2048, 512)])
@pytest.mark.parametrize("epi", ["bf16", "gelu", "gelu_pre"])
def test_g256_linear_matches_fp64_and_lean(hip, monkeypatch, M, N, K, epi):
    """csrc/gemm256.hip (256 x 256 workgroup tile, 8 waves that stage AND multiply on the two-k-tile phase schedule, counted vmcnt,
    M halves one barrier apart) through countr_gemm with COUNTR_G256=2 (wherever it qualifies): fc1's shape (the one it serves in the
    step), ragged last row tiles (B = 5 images; 1000 and 40 rows: rows beyond M stage zeros and are never stored), the shortest
    legal K (4 k-tiles: prologue and drain overlap), K = 20 k-tiles; every epilogue; against torch fp64 and against the 128-row
    forms of linear.hip (COUNTR_G256=0).
    The launch is repeated: a phase-schedule race (an LDS-DMA unit read before the barrier that orders it) shows as run-to-run
    differences long before it shows against fp64."""
    A = _mk((M, K), torch.bfloat16, 121)
    W = (_mk((N, K), torch.float32, 122) * 0.25).to(torch.bfloat16)
    bias = _mk((N,), torch.float32, 123)
    z = A.double() @ W.double().t() + bias.double()
    ref = torch.nn.functional.gelu(z) if epi.startswith("gelu") else z
    outs = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("COUNTR_G256", mode)
        runs = []
        for rep in range(3 if mode == "2" else 1):
            out = torch.full((M + 8, N), float("nan"), device="cuda", dtype=torch.bfloat16)     # 8 guard rows behind M
            pre = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16) if epi == "gelu_pre" else None
            a = _lib.GemmArgs()
            a.A, a.B, a.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
            a.C2 = pre.data_ptr() if pre is not None else None
            a.bias = bias.data_ptr()
            a.lda, a.ldb, a.ldc = K, K, N
            a.M, a.N, a.K = M, N, K
            a.act = 1 if epi.startswith("gelu") else 0
            a.out_bf16 = 1
            a.alpha = 1.0
            a.nbatch = 1; a.nb1 = 1; a.splitk = 1
            _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "gemm")
            torch.cuda.synchronize()
            assert torch.isnan(out[M:].float()).all()                    # nothing stored behind the last row
            out = out[:M]
            assert torch.isfinite(out.float()).all()
            scale = ref.abs().max().item()
            assert (out.double() - ref).abs().max().item() <= 4e-3 * scale + 3e-5, (mode, (out.double() - ref).abs().max().item(), scale)
            if pre is not None:
                assert (pre.double() - z).abs().max().item() <= 4e-3 * z.abs().max().item()
            runs.append(out)
        for r in runs[1:]:
            assert torch.equal(r, runs[0]), "run-to-run difference (phase-schedule race)"
        outs[mode] = runs[0]
    # same k order per element and the same epilogue arithmetic: which kernel a batch size selects does not change a result
    assert torch.equal(outs["2"], outs["0"]), (outs["2"].double() - outs["0"].double()).abs().max().item()


@pytest.mark.parametrize("M,N,K,res_mod", [(18432, 768, 768, 0), (18432, 768, 3072, 0), (1000, 512, 384, 0), (1152, 256, 256, 576), (40, 768, 256, 0)])
@pytest.mark.parametrize("producer", [True, False])
def test_g256_linear_residual_and_layernorm_producer(hip, monkeypatch, M, N, K, res_mod, producer):
    """The residual epilogue of the 256 x 256 kernel (attn.proj / mlp.fc2 of the encoder at 32 windows: fp32 out = acc + bias + residual,
    optionally with the LayerNorm producer's bf16 copy and per-64-column {sum, sum of squares} row partials): against fp64 and --
    output, copy and partials BIT FOR BIT -- against linear.hip's kernel (COUNTR_G256=0), ragged M and a row-modulo residual included.
    Reference: Block.forward x = x + attn(norm1(x)); x = x + mlp(norm2(x)) (timm 0.4.9, models_mae_cross.py:32-34)."""
    A = _mk((M, K), torch.bfloat16, 141)
    W = (_mk((N, K), torch.float32, 142) * 0.25).to(torch.bfloat16)
    bias = _mk((N,), torch.float32, 143)
    resid = _mk((res_mod if res_mod else M, N), torch.float32, 144) * 3.0
    rr = resid.double().repeat(M // res_mod, 1) if res_mod else resid.double()
    ref = A.double() @ W.double().t() + bias.double() + rr
    got = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("COUNTR_G256", mode)
        runs = []
        for rep in range(3 if mode == "2" else 1):
            out = torch.full((M + 8, N), float("nan"), device="cuda", dtype=torch.float32)
            xcopy = torch.full((M + 8, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            stats = torch.full((M + 8, N // 64, 2), float("nan"), device="cuda", dtype=torch.float32)
            a = _lib.GemmArgs()
            a.A, a.B, a.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
            a.bias, a.resid = bias.data_ptr(), resid.data_ptr()
            a.lda, a.ldb, a.ldc, a.ldres = K, K, N, N
            a.res_mod = res_mod
            a.M, a.N, a.K = M, N, K
            a.out_bf16 = 0
            a.alpha = 1.0
            a.nbatch = 1; a.nb1 = 1; a.splitk = 1
            if producer:
                a.ln_xcopy, a.ln_stats_out = xcopy.data_ptr(), stats.data_ptr()
            _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "gemm")
            torch.cuda.synchronize()
            assert torch.isnan(out[M:]).all() and torch.isnan(xcopy[M:].float()).all() and torch.isnan(stats[M:]).all()
            assert (out[:M].double() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item()
            if producer:
                o = out[:M].double()
                want = torch.stack([o.view(M, N // 64, 64).sum(-1), (o * o).view(M, N // 64, 64).sum(-1)], dim=-1)
                assert (stats[:M].double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()
                assert torch.equal(xcopy[:M], out[:M].to(torch.bfloat16))
            else:
                assert torch.isnan(xcopy.float()).all() and torch.isnan(stats).all()
            runs.append((out[:M].clone(), xcopy[:M].clone(), stats[:M].clone()))
        for r in runs[1:]:
            assert torch.equal(r[0], runs[0][0]), "run-to-run difference (phase-schedule race)"
        got[mode] = runs[0]
    assert torch.equal(got["2"][0], got["0"][0]), (got["2"][0].double() - got["0"][0].double()).abs().max().item()
    if producer:
        assert torch.equal(got["2"][1], got["0"][1])
        assert torch.equal(got["2"][2], got["0"][2]), (got["2"][2].double() - got["0"][2].double()).abs().max().item()


@pytest.mark.parametrize("M,N,K", [(4608, 3072, 768), (18432, 3072, 768), (1152, 1536, 512), (300, 256, 256)])
def test_g256_linear_layernorm_consumer(hip, monkeypatch, M, N, K):
    """The LayerNorm-fold consumer epilogue of the 256 x 256 kernel (fc1 in the frozen encoder: rstd (acc - mean colsum) + bias' from the
    producer's per-64-column {sum, sum of squares} row partials) against the same launch on linear.hip's kernel and against fp64."""
    x = _mk((M, K), torch.float32, 131) * 2.0 + _mk((M, 1), torch.float32, 132)
    xb = x.to(torch.bfloat16)
    W = (_mk((N, K), torch.float32, 133) * 0.25).to(torch.bfloat16)
    bias = _mk((N,), torch.float32, 134)
    xs = x.double()                                   # statistics of the fp32 row (what the producer's epilogue sums)
    stats = torch.stack([xs.view(M, K // 64, 64).sum(-1), (xs * xs).view(M, K // 64, 64).sum(-1)], dim=-1).float().contiguous()
    colsum = W.double().sum(1).float().contiguous()
    mean = xs.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(xs.var(1, unbiased=False, keepdim=True) + 1e-6)
    ref = torch.nn.functional.gelu(rstd * (xb.double() @ W.double().t() - mean * colsum.double()) + bias.double())
    outs = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("COUNTR_G256", mode)
        out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
        a = _lib.GemmArgs()
        a.A, a.B, a.C, a.bias = xb.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr()
        a.lda, a.ldb, a.ldc = K, K, N
        a.M, a.N, a.K = M, N, K
        a.act, a.out_bf16, a.alpha = 1, 1, 1.0
        a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        a.ln_stats, a.ln_colsum, a.ln_nblk, a.ln_eps = stats.data_ptr(), colsum.data_ptr(), K // 64, 1e-6
        _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "gemm")
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        assert (out.double() - ref).abs().max().item() <= 6e-3 * ref.abs().max().item() + 3e-5, mode
        outs[mode] = out
    # same k order per element, same epilogue arithmetic (explicit fma in both): the kernel a batch size selects must not change a result
    assert torch.equal(outs["2"], outs["0"]), (outs["2"].double() - outs["0"].double()).abs().max().item()


@pytest.mark.parametrize("Bsz,H,W,Cin,Cout,use_bias", [(2, 96, 96, 256, 256, True), (1, 192, 192, 256, 256, True), (8, 48, 48, 512, 256, False),
                                                   (3, 100, 100, 128, 512, True), (32, 24, 24, 256, 256, True), (1, 20, 12, 128, 256, True),
                                                   (23, 64, 64, 128, 256, True)])      # (368 tiles: one full round here + 112 tiles as a 128-row tail launch of 224 workgroups)
def test_g256_conv3x3_matches_fp64_and_lean(hip, monkeypatch, Bsz, H, W, Cin, Cout, use_bias):
    """3x3 convolution forward / dgrad on the 256 x 256 8-phase kernel (im2row LDS-DMA descriptors: tap shifts as scalar offsets, padding
    taps and ragged rows as lanes pushed outside the descriptor) with COUNTR_G256=2: against torch conv2d in fp64 and against the
    128x256 form of linear.hip on the same bf16 maps -- zero padding at every border, tiles that span image boundaries (100 x 100,
    24 x 24 = 2.25 images per tile), a map smaller than one tile (20 x 12: most staged rows are masked), Cin = 128 / 256 / 512
    (2 / 4 / 8 k-tiles per tap), with and without bias; a grid of 1.44 rounds of workgroups, which runs as one full round on this kernel plus
    a tail launch of the 128-row kernel on the last rows (split rounds); three runs must agree bit for bit (race screen)."""
    x = _mk((Bsz, H, W, Cin), torch.bfloat16, 151)
    w = (_mk((Cout, 3, 3, Cin), torch.float32, 152) * 0.1).to(torch.bfloat16)
    bias = _mk((Cout,), torch.float32, 153) if use_bias else None
    M, K = Bsz * H * W, 9 * Cin
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double() if use_bias else None, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    outs = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("COUNTR_G256", mode)
        runs = []
        for rep in range(3 if mode == "2" else 1):
            out = torch.full((M, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
            a = _lib.GemmArgs()
            a.A, a.B, a.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
            a.bias = bias.data_ptr() if use_bias else None
            a.ldb, a.ldc = K, Cout
            a.M, a.N, a.K = M, Cout, K
            a.H, a.W, a.Cin = H, W, Cin
            a.out_bf16, a.alpha = 1, 1.0
            a.nbatch = 1; a.nb1 = 1; a.splitk = 1
            _lib.check(hip.countr_gemm(C.byref(a), 1, 2, 0, _stream()), "conv")
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all()
            assert (out.double() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item(), mode
            runs.append(out)
        for r in runs[1:]:
            assert torch.equal(r, runs[0]), "run-to-run difference (phase-schedule race)"
        outs[mode] = runs[0]
    assert torch.equal(outs["2"], outs["0"]), (outs["2"].double() - outs["0"].double()).abs().max().item()


@pytest.mark.parametrize("Bsz,H,W,Cin,Cout", [(2, 96, 96, 256, 256), (1, 192, 192, 256, 256), (23, 64, 64, 128, 256), (3, 100, 100, 128, 512),
                                              (8, 48, 48, 256, 256)])
def test_conv3x3_groupnorm_row_partials(hip, monkeypatch, Bsz, H, W, Cin, Cout):
    """countr_gemm_args.gn_rows (ABI 9): the 16-bit convolution kernels leave {sum, sum of squares} of every 32-channel block of every
    ROUNDED output row -- what a GroupNorm statistics pass over the stored map would read.  Against fp64 sums of the stored map (the
    partials are fp32 trees over 32 values: 1e-6), unchanged outputs, and -- because a row's value must not depend on the kernel, the
    tile or the batch it was computed in -- bit for bit between the 256 x 256 kernel (incl. its split rounds: 23 images of 64 x 64),
    the 128-row kernel alone, and a launch over the first image only; rows behind M are not written; countr_gemm_gn_rows answers the
    selection question and a launch that falls to a kernel without the epilogue fails instead of leaving the buffer unwritten."""
    x = _mk((Bsz, H, W, Cin), torch.bfloat16, 161)
    w = (_mk((Cout, 3, 3, Cin), torch.float32, 162) * 0.1).to(torch.bfloat16)
    bias = _mk((Cout,), torch.float32, 163)
    M, K, NG = Bsz * H * W, 9 * Cin, Cout // 32

    def launch(mode, m_rows, with_rows=True):
        monkeypatch.setenv("COUNTR_G256", mode)
        out = torch.full((m_rows, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
        rows = torch.full((m_rows + 4, NG, 2), float("nan"), device="cuda", dtype=torch.float32)
        a = _lib.GemmArgs()
        a.A, a.B, a.C, a.bias = x.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr()
        a.ldb, a.ldc = K, Cout
        a.M, a.N, a.K = m_rows, Cout, K
        a.H, a.W, a.Cin = H, W, Cin
        a.out_bf16, a.alpha = 1, 1.0
        a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        if with_rows:
            a.gn_rows = rows.data_ptr()
            assert hip.countr_gemm_gn_rows(C.byref(a), 1, 2, 0) == 1
        _lib.check(hip.countr_gemm(C.byref(a), 1, 2, 0, _stream()), "conv")
        torch.cuda.synchronize()
        return out, rows

    out_ref, _ = launch("2", M, with_rows=False)
    res = {}
    for mode in ("2", "1", "0"):
        out, rows = launch(mode, M)
        assert torch.equal(out, out_ref), mode                       # the epilogue's extra work changes no output
        assert torch.isnan(rows[M:]).all() and torch.isfinite(rows[:M]).all(), mode
        v = out.double().view(M, NG, 32)
        s1, s2 = v.sum(-1), (v * v).sum(-1)
        assert (rows[:M, :, 0].double() - s1).abs().max().item() <= 2e-6 * s1.abs().max().item() + 1e-6, mode
        assert (rows[:M, :, 1].double() - s2).abs().max().item() <= 2e-6 * s2.abs().max().item() + 1e-6, mode
        res[mode] = rows[:M].clone()
    assert torch.equal(res["2"], res["0"]) and torch.equal(res["1"], res["0"])
    if Cout == 256:
        # the two routes to a GroupNorm's statistics -- the pass over the stored map and the pass over these row partials -- agree BIT FOR
        # BIT (one association tree: common.hpp::countr_gn_quad_sums, norm.hip::gn_stats_tree_kernel / gn_stats_rows_kernel): which
        # route a batch size selects must not change a result
        gam = _mk((256,), torch.float32, 165) * 0.2 + 1.0
        bet = _mk((256,), torch.float32, 166) * 0.1
        got = {}
        for route in ("map", "rows"):
            ws = torch.zeros(Bsz * 128 * 3 * 256 + 64 + 16 * Bsz + Bsz * 3 * 256, device="cuda")
            stats = torch.empty((Bsz, 8, 2), device="cuda")
            y = torch.empty_like(out_ref)
            if route == "map":
                _lib.check(hip.countr_groupnorm_relu_fwd(out_ref.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), None, None, None,
                                                         stats.data_ptr(), ws.data_ptr(), Bsz, H * W, 256, 8, 1e-5, 1, _stream()), "gn")
            else:
                _lib.check(hip.countr_groupnorm_relu_fwd_rows(out_ref.data_ptr(), res["0"].data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(),
                                                              None, None, None, stats.data_ptr(), ws.data_ptr(), Bsz, H * W, 256, 8, 1e-5, 1,
                                                              _stream()), "gn rows")
            torch.cuda.synchronize()
            got[route] = (stats, y)
        assert torch.equal(got["map"][0], got["rows"][0]) and torch.equal(got["map"][1], got["rows"][1])
    if Bsz > 1 and H * W * (Cout // 128) > 256 * 128:                # the first image alone still runs on the lean kernels
        _o, rows1 = launch("1", H * W)
        assert torch.equal(rows1[:H * W], res["0"][:H * W])
    # a map of fewer than 257 tiles runs on the generic kernel: the query says so, the launch refuses
    xs = _mk((1, 24, 24, Cin), torch.bfloat16, 164)
    outs = torch.empty((576, Cout), device="cuda", dtype=torch.bfloat16)
    rows = torch.zeros((576, NG, 2), device="cuda")
    a = _lib.GemmArgs()
    a.A, a.B, a.C, a.bias = xs.data_ptr(), w.data_ptr(), outs.data_ptr(), bias.data_ptr()
    a.ldb, a.ldc = K, Cout
    a.M, a.N, a.K = 576, Cout, K
    a.H, a.W, a.Cin = 24, 24, Cin
    a.out_bf16, a.alpha = 1, 1.0
    a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.gn_rows = rows.data_ptr()
    assert hip.countr_gemm_gn_rows(C.byref(a), 1, 2, 0) == 0
    assert hip.countr_gemm(C.byref(a), 1, 2, 0, _stream()) != 0 and b"gn_rows" in hip.countr_last_error()


@pytest.mark.parametrize("M,N,K,epi", [(4608, 768, 3072, "res"), (4608, 3072, 768, "gelu"), (4608, 2304, 768, "bf16"), (1152, 512, 512, "res"), (40, 128, 128, "bf16")])
def test_prefetch_hint_changes_nothing(hip, M, N, K, epi):
    """countr_gemm_args.prefetch (ABI 3): spare workgroups of a single-round launch read a range and leave -- the 128-row kernel, the
    192 x 256 form and the 256 x 256 kernel (the encoder's fc2 / fc1 / qkv shapes at B = 8: 216 tiles + 40 warm-up workgroups), a small
    grid (4 tiles) and a one-tile launch; the output is bit-identical to the launch without the hint, the hinted range and the guard
    rows are untouched, and an unaligned hint is ignored rather than refused (it is a hint)."""
    A = _mk((M, K), torch.bfloat16, 221)
    W = (_mk((N, K), torch.float32, 222) * 0.25).to(torch.bfloat16)
    bias = _mk((N,), torch.float32, 223)
    resid = _mk((M, N), torch.float32, 224) if epi == "res" else None
    nxt = _mk((3 * 1024 * 1024 + 8,), torch.float32, 225)
    nxt0 = nxt.clone()
    outs = []
    for hint in (None, (nxt.data_ptr(), 4 * nxt.numel()), (nxt.data_ptr() + 4, 4096)):
        out = torch.full((M + 8, N), float("nan"), device="cuda", dtype=torch.float32 if epi == "res" else torch.bfloat16)
        a = _lib.GemmArgs()
        a.A, a.B, a.C, a.bias = A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr()
        a.resid = resid.data_ptr() if resid is not None else None
        a.lda, a.ldb, a.ldc, a.ldres = K, K, N, N
        a.M, a.N, a.K = M, N, K
        a.act = 1 if epi == "gelu" else 0
        a.out_bf16 = int(epi != "res")
        a.alpha = 1.0
        a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        if hint is not None:
            a.prefetch, a.prefetch_bytes = hint
        _lib.check(hip.countr_gemm(C.byref(a), 1, 0, 0, _stream()), "gemm")
        torch.cuda.synchronize()
        assert torch.isnan(out[M:].float()).all() and torch.isfinite(out[:M].float()).all()
        outs.append(out[:M].clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(nxt, nxt0)
    ref = A.double() @ W.double().t() + bias.double()
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if resid is not None:
        ref = ref + resid.double()
    assert (outs[1].double() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item() + 3e-5


_FIRST_CALL_CHILD = r'''
import ctypes as C, sys, threading, torch
from countr_amd import _lib
L = _lib.lib()
M, N, K = 512, 256, 128
g = torch.Generator().manual_seed(5)
A = (torch.rand(M, K, generator=g) * 2 - 1).cuda().bfloat16()
W = (torch.rand(N, K, generator=g) * 2 - 1).cuda().bfloat16()
ref = A.double() @ W.double().t()
def args(out):
    a = _lib.GemmArgs()
    a.A, a.B, a.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
    a.lda, a.ldb, a.ldc = K, K, N
    a.M, a.N, a.K = M, N, K
    a.alpha = 1.0; a.out_bf16 = 1; a.nbatch = 1; a.nb1 = 1; a.splitk = 1      # no bias: the epilogue adds the per-device zero vector
    return a
mode = sys.argv[1]
if mode == "noinit":
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    rc = L.countr_gemm(C.byref(args(out)), 1, 0, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"countr_init" in L.countr_last_error(), (rc, L.countr_last_error())
elif mode == "threads":
    outs = [torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    bar = threading.Barrier(2)
    rcs = [None, None]
    def run(i):
        bar.wait()
        r0 = L.countr_init(0)                                              # both threads race through the one-time allocation
        rcs[i] = (r0, L.countr_gemm(C.byref(args(outs[i])), 1, 0, 0, C.c_void_p(streams[i].cuda_stream)))
    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    torch.cuda.synchronize()
    assert rcs == [(0, 0), (0, 0)], rcs
    for o in outs:
        assert (o.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert torch.equal(outs[0], outs[1])
elif mode == "capture":
    _lib.check(L.countr_init(0), "init")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    a = args(out)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):                                             # the process's FIRST bias-less launch happens under capture
        _lib.check(L.countr_gemm(C.byref(a), 1, 0, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
    torch.cuda.synchronize()
    assert out.abs().max().item() == 0                                     # captured, not run
    gr.replay(); torch.cuda.synchronize()
    assert (out.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
print("child ok")
'''


@pytest.mark.parametrize("mode", ["noinit", "threads", "capture"])
def test_bias_less_launch_first_call(mode):
    """ABI conventions: no allocation inside a launch, no mutable global state after countr_init, thread-safe.  Bias-less
    launches of linear.hip / gemm256.hip read a per-device vector of zeros that countr_init allocates (rounds 3-4 allocated it lazily
    inside the first launch).  Fresh process each: (noinit) a launch without countr_init fails with a message instead of allocating;
    (threads) two threads race countr_init + their first launch on two streams; (capture) the first launch of the process is recorded
    into a hipGraph."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FIRST_CALL_CHILD, mode], capture_output=True, text=True, timeout=300, cwd=root,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0 and "child ok" in r.stdout, (r.stdout[-400:], r.stderr[-1200:])
