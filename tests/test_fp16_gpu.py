"""GPU: the fp16 build of the library (libcountr_hip_f16.so = the same sources with IEEE fp16 as the 16-bit storage / matrix-operand
type, csrc/common.hpp; precision="fp16", the reference's autocast dtype: FSC_finetune_cross.py:273-275,286) at kernel level, through
the C ABI, against fp64 on the same fp16-representable inputs.  Bars are the fp16 rounding of the outputs (2^-11 relative) plus the
fp32 accumulation error -- 8x tighter than the bf16 bars of tests/test_kernels_gpu.py / test_gemm_gpu.py for the same kernels."""
import ctypes as C

import pytest
import torch

from countr_amd import _lib

pytestmark = pytest.mark.gpu
H16 = torch.float16


@pytest.fixture(scope="module")
def hip16():
    L = _lib.lib("f16")
    _lib.check(L.countr_init(0), "countr_init")
    return L


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def rnd(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale)


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("ma,mb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(4608, 768, 768), (576, 2304, 768), (1000, 256, 192)])
def test_gemm_fp16(hip16, ma, mb, M, N, K):
    """nn.Linear forward (ROW, ROW -- the lean / 256 x 256 kernels where the shape qualifies), input gradient (ROW, COL) and weight
    gradient (COL, COL) operand modes with fp16 operands and an fp16 result."""
    A = rnd((M, K), 1).to(H16)
    B = rnd((N, K), 2).to(H16)
    As = (A if ma == 0 else A.t().contiguous()).cuda()
    Bs = (B if mb == 0 else B.t().contiguous()).cuda()
    bias = rnd((N,), 3).cuda()
    out = torch.empty((M, N), device="cuda", dtype=H16)
    a = _lib.GemmArgs()
    a.A, a.B, a.C, a.bias = As.data_ptr(), Bs.data_ptr(), out.data_ptr(), bias.data_ptr()
    a.lda, a.ldb, a.ldc = (K if ma == 0 else M), (K if mb == 0 else N), N
    a.M, a.N, a.K, a.alpha, a.out_bf16 = M, N, K, 1.0, 1
    a.nbatch = a.nb1 = a.splitk = 1
    _lib.check(hip16.countr_gemm(C.byref(a), _lib.BF16, ma, mb, st()), "gemm")     # (dtype code BF16 = "the 16-bit type of this build")
    torch.cuda.synchronize()
    ref = A.double() @ B.double().t() + bias.double().cpu()
    assert relerr(out, ref) < 8e-4, relerr(out, ref)


def test_gelu_epilogue_and_residual_fp16(hip16):
    M, N, K = 4608, 3072, 768
    A, Wt = rnd((M, K), 4).to(H16), rnd((N, K), 5, 0.05).to(H16)
    bias = rnd((N,), 6).cuda()
    out = torch.empty((M, N), device="cuda", dtype=H16)
    a = _lib.GemmArgs()
    Ad, Wd = A.cuda(), Wt.cuda()
    a.A, a.B, a.C, a.bias = Ad.data_ptr(), Wd.data_ptr(), out.data_ptr(), bias.data_ptr()
    a.lda, a.ldb, a.ldc, a.M, a.N, a.K, a.alpha, a.out_bf16, a.act = K, K, N, M, N, K, 1.0, 1, _lib.ACT_GELU
    a.nbatch = a.nb1 = a.splitk = 1
    _lib.check(hip16.countr_gemm(C.byref(a), _lib.BF16, 0, 0, st()), "gemm")       # fc1 of the encoder at B = 8: the 256 x 256 kernel
    ref = torch.nn.functional.gelu(A.double() @ Wt.double().t() + bias.double().cpu())
    assert relerr(out, ref) < 1e-3      # (+ the 2.6e-5 of the sigmoid form of GELU)


@pytest.mark.parametrize("B,N,Hh,dh", [(2, 576, 12, 64), (2, 576, 16, 32), (1, 200, 3, 64)])
def test_flash_attention_fp16_fwd_bwd(hip16, B, N, Hh, dh):
    qkv = rnd((B, N, 3, Hh, dh), 50).to(H16)
    qkv[0, N // 2, 1, 0] = 6.0
    do = rnd((B, N, Hh * dh), 61).to(H16)
    qd, dod = qkv.cuda(), do.cuda()
    out = torch.empty((B, N, Hh * dh), device="cuda", dtype=H16)
    lse = torch.empty((B, Hh, N), device="cuda")
    delta = torch.empty((B, Hh, N), device="cuda")
    dqkv = torch.full((B, N, 3, Hh, dh), float("nan"), device="cuda", dtype=H16)
    scale = dh ** -0.5
    _lib.check(hip16.countr_attn_fwd(P(qd), P(out), P(lse), B, N, Hh, dh, scale, st()))
    _lib.check(hip16.countr_attn_bwd(P(qd), P(out), P(dod), P(lse), P(delta), P(dqkv), B, N, Hh, dh, scale, st()))
    x = qkv.double().requires_grad_(True)
    q = x[:, :, 0].permute(0, 2, 1, 3); k = x[:, :, 1].permute(0, 2, 1, 3); v = x[:, :, 2].permute(0, 2, 1, 3)
    sc = q @ k.transpose(-1, -2) * scale
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, Hh * dh)
    ref.backward(do.double())
    assert relerr(out, ref) < 2e-3                        # bf16 build: 1.5e-2
    assert relerr(lse, torch.logsumexp(sc.detach(), -1)) < 1e-4
    assert torch.isfinite(dqkv.float()).all()
    for slot, name in enumerate(("dq", "dk", "dv")):
        assert relerr(dqkv[:, :, slot], x.grad[:, :, slot]) < 4e-3, name      # bf16 build: 2.5e-2


@pytest.mark.parametrize("form", ["2", "3"])
@pytest.mark.parametrize("Bsz,H,W,Cin,Cout,sk", [(1, 64, 64, 256, 256, 4), (2, 6, 96, 128, 128, 2), (1, 48, 48, 128, 256, 3)])
def test_conv_wgrad_fp16(hip16, monkeypatch, form, Bsz, H, W, Cin, Cout, sk):
    """Weight + bias gradient of a 3x3 convolution in the fp16 build: the lean kernels of conv_wgrad.hip (128 x 256 tiles; the three
    taps of a kernel row per workgroup with 64- and 96-pixel k-tiles, and its fall-back on a 48-wide map) against fp64 autograd on the
    same fp16 maps -- the partial sums are fp32, so the bar is the accumulation's."""
    monkeypatch.setenv("COUNTR_LEAN_WGRAD_FORM", form)
    dy = rnd((Bsz, H, W, Cout), 5).to(H16).cuda()
    x = rnd((Bsz, H, W, Cin), 6).to(H16).cuda()
    Pn, N = Bsz * H * W, 9 * Cin
    wz = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    bz = torch.zeros(Cout, dtype=torch.float64, device="cuda", requires_grad=True)
    torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wz, bz, padding=1).backward(dy.double().permute(0, 3, 1, 2))
    ref_w, ref_b = wz.grad.permute(0, 2, 3, 1).reshape(Cout, N), bz.grad
    a = _lib.GemmArgs()
    a.A, a.B = dy.data_ptr(), x.data_ptr()
    a.lda, a.ldc = Cout, N
    a.M, a.N, a.K, a.H, a.W, a.Cin = Cout, N, Pn, H, W, Cin
    a.alpha, a.nbatch, a.nb1, a.splitk = 1.0, 1, 1, sk
    slabs = hip16.countr_gemm_rowsum_slabs(C.byref(a), _lib.BF16, 1, 3)
    part = torch.full((sk, Cout, N), float("nan"), device="cuda")
    rs = torch.full((slabs, Cout), float("nan"), device="cuda")
    a.partial, a.rowsum_partial, a.rowsum_slabs = part.data_ptr(), rs.data_ptr(), slabs
    _lib.check(hip16.countr_gemm(C.byref(a), _lib.BF16, 1, 3, st()), "wgrad")
    torch.cuda.synchronize()
    assert (part.double().sum(0) - ref_w).abs().max().item() <= 2e-5 * ref_w.abs().max().item()
    assert (rs.double().sum(0) - ref_b).abs().max().item() <= 2e-5 * ref_b.abs().max().item()


def test_transpose16_fp16(hip16):
    """countr_transpose16 moves 16-bit words: the fp16 build's W^T shadows are the bits of the fp16 shadow, transposed."""
    sh = [(768, 2304), (512, 512), (128, 64)]
    ws = [rnd(s_, 7 + i).to(H16).cuda() for i, s_ in enumerate(sh)]
    wt = [torch.zeros(c, r, device="cuda", dtype=H16) for r, c in sh]
    n = len(sh)
    vp, ip = C.c_void_p * n, C.c_int * n
    _lib.check(hip16.countr_transpose16(n, vp(*[w.data_ptr() for w in ws]), vp(*[w.data_ptr() for w in wt]), ip(*[s_[0] for s_ in sh]),
                                        ip(*[s_[1] for s_ in sh]), st()), "transpose16")
    torch.cuda.synchronize()
    for w, t in zip(ws, wt):
        assert torch.equal(t, w.t().contiguous())


def test_layernorm_and_cast_fp16(hip16):
    rows, D = 1157, 768
    x = (rnd((rows, D), 1, 2.0) + 0.3).cuda()
    g, b = (1 + 0.1 * rnd((D,), 2)).cuda(), (0.1 * rnd((D,), 3)).cuda()
    y = torch.empty((rows, D), device="cuda", dtype=H16)
    _lib.check(hip16.countr_layernorm_fwd(P(x), P(g), P(b), P(y), None, None, rows, D, 1e-6, 1, st()))
    ref = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-6)
    assert relerr(y, ref) < 6e-4
    w = torch.empty((rows, D), device="cuda", dtype=H16)
    _lib.check(hip16.countr_cast_permute(P(x), P(w), rows * D, 0, 0, 0, 0, _lib.BF16, st()))
    assert torch.equal(w, x.to(H16))                      # round-to-nearest-even, as torch's cast
