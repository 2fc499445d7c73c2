"""GPU parity of the individual HIP kernels (through the C ABI) against the CPU oracle's functions
(oracle/countr_ref.py) on the same seeded inputs.  fp32 mode: <= 1e-4 rel; bf16 storage mode: tolerance
set by bf16 rounding of inputs/outputs (2^-8), stated per test."""
import ctypes as C

import pytest
import torch

from countr_amd import _lib
from oracle import countr_ref as R

pytestmark = pytest.mark.gpu

DT = [(torch.float32, 0, 2e-4), (torch.bfloat16, 1, 2e-2)]


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def relerr(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("D", [768, 512, 1024])
@pytest.mark.parametrize("tdt,code,tol", DT)
def test_layernorm_fwd_bwd(hip, D, tdt, code, tol):
    rows = 1157
    x = rnd((rows, D), 1, 2.0) + 0.3
    g = 1 + 0.1 * rnd((D,), 2)
    b = 0.1 * rnd((D,), 3)
    dy = rnd((rows, D), 4).to(tdt)
    xd, gd, bd = x.cuda(), g.cuda(), b.cuda()
    y = torch.empty((rows, D), device="cuda", dtype=tdt)
    mean = torch.empty(rows, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    _lib.check(hip.countr_layernorm_fwd(P(xd), P(gd), P(bd), P(y), P(mean), P(rstd), rows, D, 1e-6, code, st()))
    xr = x.double().requires_grad_(True)
    gr = g.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = R.layer_norm(xr, gr, br)
    assert relerr(y, yr) < tol
    nb = hip.countr_layernorm_bwd_nblocks()
    ws = torch.empty((nb, 2, D), device="cuda")
    dx = torch.full((rows, D), 0.5, device="cuda")
    dgb = torch.zeros((2, D), device="cuda")
    dyd = dy.cuda()
    dxb = torch.empty((rows, D), device="cuda", dtype=torch.bfloat16)
    _lib.check(hip.countr_layernorm_bwd(P(dyd), P(xd), P(gd), P(mean), P(rstd), P(dx), P(dgb[0]), P(dgb[1]), P(ws), rows, D,
                                        code, 1, 0, P(dxb), st()))   # adjacent dgamma/dbeta: single finisher launch
    yr.backward(dy.double())
    assert relerr(dx - 0.5, xr.grad) < 1e-4
    assert relerr(dgb[0], gr.grad) < 1e-4
    assert relerr(dgb[1], br.grad) < 1e-4
    assert torch.equal(dxb, dx.bfloat16())          # the optional bf16 copy is the RNE rounding of the updated dx
    # separate (non-adjacent) outputs, accumulate into existing values, no bf16 copy
    dg2, db2 = torch.ones(D, device="cuda"), torch.ones(D, device="cuda")
    _lib.check(hip.countr_layernorm_bwd(P(dyd), P(xd), P(gd), P(mean), P(rstd), P(dx), P(dg2), P(db2), P(ws), rows, D,
                                        code, 0, 1, None, st()))
    assert relerr(dg2 - 1, gr.grad) < 1e-4 and relerr(db2 - 1, br.grad) < 1e-4
    assert relerr(dx, xr.grad) < 1e-4


@pytest.mark.parametrize("HW", [24 * 24, 96 * 96, 35 * 35, 100])   # incl. pixel counts that do not divide into the splits
@pytest.mark.parametrize("tdt,code,tol", DT)
def test_groupnorm_relu_fwd_bwd(hip, HW, tdt, code, tol):
    B, Cc, G = 2, 256, 8
    x = (rnd((B, HW, Cc), 5, 1.5) + 0.2).to(tdt)
    g = 1 + 0.1 * rnd((Cc,), 6)
    b = 0.1 * rnd((Cc,), 7)
    dy = rnd((B, HW, Cc), 8).to(tdt)
    w1 = 0.1 * rnd((Cc,), 9)
    b1 = torch.tensor([0.05])
    d1 = rnd((B, HW), 10)
    ns = hip.countr_groupnorm_nsplit(HW)
    ws = torch.empty(B * ns * 3 * Cc + 64 + 16 * B + B * 3 * Cc, device="cuda")
    stats = torch.empty((B, G, 2), device="cuda")
    xd, gd, bd = x.cuda(), g.cuda(), b.cuda()
    y = torch.empty((B, HW, Cc), device="cuda", dtype=tdt)
    _lib.check(hip.countr_groupnorm_relu_fwd(P(xd), P(gd), P(bd), P(y), None, None, None, P(stats), P(ws), B, HW, Cc, G, 1e-5,
                                             code, st()))
    side = int(HW ** 0.5)
    xr = x.double().reshape(B, side, side, Cc).permute(0, 3, 1, 2).requires_grad_(True)
    gr = g.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = R.group_norm_relu(xr, gr, br)
    y_ref = yr.permute(0, 2, 3, 1).reshape(B, HW, Cc)
    assert relerr(y, y_ref) < tol
    # backward (plain)
    dx = torch.empty_like(xd)
    dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    dyd = dy.cuda()
    _lib.check(hip.countr_groupnorm_relu_bwd(P(xd), P(dyd), None, None, P(stats), P(gd), P(bd), P(dx), P(dg), P(db), None, None,
                                             P(ws), B, HW, Cc, G, code, 0, st()))
    yr.backward(dy.double().reshape(B, side, side, Cc).permute(0, 3, 1, 2))
    dx_ref = xr.grad.permute(0, 2, 3, 1).reshape(B, HW, Cc)
    assert relerr(dx, dx_ref) < (tol if code else 5e-4)
    assert relerr(dg, gr.grad) < 2e-3 and relerr(db, br.grad) < 2e-3
    # fused 1x1 head forward + backward
    out1 = torch.empty((B, HW), device="cuda")
    w1d, b1d, d1d = w1.cuda(), b1.cuda(), d1.cuda()
    _lib.check(hip.countr_groupnorm_relu_fwd(P(xd), P(gd), P(bd), None, P(w1d), P(b1d), P(out1), P(stats), P(ws), B, HW, Cc, G,
                                             1e-5, code, st()))
    xr2 = x.double().reshape(B, side, side, Cc).permute(0, 3, 1, 2).requires_grad_(True)
    gr2 = g.double().requires_grad_(True); br2 = b.double().requires_grad_(True)
    w1r = w1.double().requires_grad_(True); b1r = b1.double().requires_grad_(True)
    o_ref = (R.group_norm_relu(xr2, gr2, br2) * w1r.view(1, Cc, 1, 1)).sum(1) + b1r
    assert relerr(out1, o_ref.reshape(B, HW)) < 1e-3
    dw1 = torch.zeros(Cc, device="cuda"); db1 = torch.zeros(1, device="cuda")
    _lib.check(hip.countr_groupnorm_relu_bwd(P(xd), None, P(d1d), P(w1d), P(stats), P(gd), P(bd), P(dx), P(dg), P(db), P(dw1),
                                             P(db1), P(ws), B, HW, Cc, G, code, 0, st()))
    o_ref.backward(d1.double().reshape(B, side, side))
    assert relerr(dx, xr2.grad.permute(0, 2, 3, 1).reshape(B, HW, Cc)) < (tol if code else 5e-4)
    assert relerr(dg, gr2.grad) < 2e-3 and relerr(db, br2.grad) < 2e-3
    assert relerr(dw1, w1r.grad) < 2e-3 and relerr(db1, b1r.grad) < 1e-4


@pytest.mark.parametrize("HW", [96 * 96, 48 * 48, 35 * 35, 100])
def test_groupnorm_statistics_from_row_partials(hip, HW):
    """countr_groupnorm_relu_fwd_rows (ABI 9): the statistics pass reads {sum, sum of squares} per pixel and 32-channel group (what a
    convolution's epilogue leaves through countr_gemm_args.gn_rows) instead of the map -- same split partials, same finalize, same
    apply pass: mean / rstd within 2e-6 of the map-reading pass and of fp64, outputs equal to 1 bf16 ulp."""
    B, Cc, G = 3, 256, 8
    x = (rnd((B, HW, Cc), 41, 1.5) + 0.3).to(torch.bfloat16).cuda()
    g = (1 + 0.1 * rnd((Cc,), 42)).cuda(); b = (0.1 * rnd((Cc,), 43)).cuda()
    v = x.float().view(B * HW, G, 32)
    rows = torch.stack([v.sum(-1), (v * v).sum(-1)], dim=-1).contiguous()
    ns = hip.countr_groupnorm_nsplit(HW)
    outs = {}
    for name in ("map", "rows"):
        ws = torch.zeros(B * 128 * 3 * Cc + 64 + 16 * B + B * 3 * Cc, device="cuda")
        stats = torch.empty((B, G, 2), device="cuda")
        y = torch.empty_like(x)
        if name == "map":
            _lib.check(hip.countr_groupnorm_relu_fwd(P(x), P(g), P(b), P(y), None, None, None, P(stats), P(ws), B, HW, Cc, G, 1e-5, 1, st()))
        else:
            _lib.check(hip.countr_groupnorm_relu_fwd_rows(P(x), P(rows), P(g), P(b), P(y), None, None, None, P(stats), P(ws), B, HW, Cc, G, 1e-5, 1, st()))
        torch.cuda.synchronize()
        outs[name] = (stats.clone(), y.clone())
    xd = x.double().view(B, HW, G, 32)
    mean = xd.mean(dim=(1, 3)); var = xd.var(dim=(1, 3), unbiased=False)
    ref = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-5)], dim=-1)
    for name in ("map", "rows"):
        assert (outs[name][0].double().cpu() - ref.cpu()).abs().max().item() <= 3e-6 * ref.abs().max().item(), name
    assert relerr(outs["rows"][1], outs["map"][1].double()) <= 2 ** -7
    assert (outs["rows"][1].float() - outs["map"][1].float()).abs().mean().item() <= 1e-4


@pytest.mark.parametrize("use_ws", [False, True])   # True: pixel-band two-kernel path where the shape qualifies
@pytest.mark.parametrize("H,Cc,avg", [(64, 64, 0), (32, 128, 0), (16, 256, 0), (8, 512, 1), (40, 64, 0), (12, 128, 0)])   # 40, 12: odd band counts
@pytest.mark.parametrize("tdt,code,tol", DT)
def test_instnorm_relu_pool(hip, H, Cc, avg, tdt, code, tol, use_ws):
    S = 3
    x = (rnd((S, H, H, Cc), 11, 1.3) + 0.1).to(tdt)
    xd = x.cuda()
    oshape = (S, Cc) if avg else (S, H // 2, H // 2, Cc)
    y = torch.empty(oshape, device="cuda", dtype=tdt)
    stats = torch.empty((S, Cc, 2), device="cuda")
    ws = torch.empty(hip.countr_instnorm_workspace_floats(S, Cc), device="cuda") if use_ws else None
    _lib.check(hip.countr_instnorm_relu_pool_fwd(P(xd), P(y), P(stats), S, H, H, Cc, avg, 1e-5, code, P(ws) if use_ws else None, None, 0, st()))
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    a = R.instance_norm_relu(xr)
    yr = a.mean((2, 3)) if avg else R.max_pool2(a).permute(0, 2, 3, 1)
    assert relerr(y, yr) < tol
    dyp = rnd(oshape, 12).to(tdt)
    dx = torch.empty_like(xd)
    dypd = dyp.cuda()
    _lib.check(hip.countr_instnorm_relu_pool_bwd(P(xd), P(dypd), P(stats), P(dx), S, H, H, Cc, avg, code, P(ws) if use_ws else None, 0, st()))
    yr.backward(dyp.double())
    assert relerr(dx, xr.grad.permute(0, 2, 3, 1)) < (tol if code else 5e-4)
    # x-hat stored by the forward (xhat_out), read by the backward (x_is_xhat).  fp32: in place, same tolerance.  bf16: the map comes in
    # as fp32 (x_f32, what the engine does) and x-hat is its only rounding; the reference runs on the unrounded map, so a max-pool
    # window whose two largest values round to the same bf16 routes its gradient differently (first max wins, as in torch): compared in
    # cosine / rms, not in max norm
    if code == 0:
        xh = xd.clone()
        y2 = torch.empty_like(y)
        _lib.check(hip.countr_instnorm_relu_pool_fwd(P(xh), P(y2), P(stats), S, H, H, Cc, avg, 1e-5, code, P(ws) if use_ws else None, P(xh), 0, st()))
        assert relerr(y2, yr) < tol
        dx2 = torch.empty_like(xd)
        _lib.check(hip.countr_instnorm_relu_pool_bwd(P(xh), P(dypd), P(stats), P(dx2), S, H, H, Cc, avg, code, P(ws) if use_ws else None, 1, st()))
        assert relerr(dx2, xr.grad.permute(0, 2, 3, 1)) < 5e-4
    else:
        x32 = x.float().cuda()                       # (the bf16-exact values as an fp32 map: same reference)
        xh = torch.empty_like(xd)
        y2 = torch.empty_like(y)
        _lib.check(hip.countr_instnorm_relu_pool_fwd(P(x32), P(y2), P(stats), S, H, H, Cc, avg, 1e-5, code, P(ws) if use_ws else None, P(xh), 1, st()))
        assert relerr(y2, yr) < tol
        dx2 = torch.empty_like(xd)
        _lib.check(hip.countr_instnorm_relu_pool_bwd(P(xh), P(dypd), P(stats), P(dx2), S, H, H, Cc, avg, code, P(ws) if use_ws else None, 1, st()))
        torch.cuda.synchronize()
        d, r = dx2.double().cpu().reshape(-1), xr.grad.permute(0, 2, 3, 1).reshape(-1)
        assert (d @ r / (d.norm() * r.norm())).item() > 0.995 and abs(d.norm().item() / r.norm().item() - 1) < 0.02
        assert hip.countr_instnorm_relu_pool_fwd(P(x32), P(y2), P(stats), S, H, H, Cc, avg, 1e-5, code, None, P(x32), 1, st()) != 0   # bf16 x-hat in place of an fp32 map: refused


@pytest.mark.parametrize("avg,H", [(1, 8), (0, 16)])
def test_instnorm_on_fp32_maps_survives_large_channel_means(hip, avg, H):
    """Maps whose channel means are many sigma (what a conv bias in front of an InstanceNorm produces on an 8x8 map).  Rounded to bf16
    BEFORE the InstanceNorm, a value moves by 2^-9 of itself = (|mean| / sigma) * 2^-9 sigma: pixels cross the ReLU boundary of the
    normalised map and the input gradient loses several per cent in cosine against the fp32 truth.  Given to the kernel as fp32 (x_f32)
    with the bf16 NORMALISED map stored for the backward (xhat_out / x_is_xhat), the only rounding is 2^-9 of x-hat."""
    S, Cc = 6, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(S, H, H, Cc, generator=g) + 12.0 * torch.randn(1, 1, 1, Cc, generator=g)
    oshape = (S, Cc) if avg else (S, H // 2, H // 2, Cc)
    dyp = torch.randn(oshape, generator=g).to(torch.bfloat16).cuda()
    stats = torch.empty((S, Cc, 2), device="cuda")
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    a = R.instance_norm_relu(xr)
    yr = a.mean((2, 3)) if avg else R.max_pool2(a).permute(0, 2, 3, 1)
    yr.backward(dyp.double().cpu())
    ref = xr.grad.permute(0, 2, 3, 1).reshape(-1)
    errs = []
    for f32map in (0, 1):
        y = torch.empty(oshape, device="cuda", dtype=torch.bfloat16)
        dx = torch.empty((S, H, H, Cc), device="cuda", dtype=torch.bfloat16)
        if f32map:
            xin, xh = x.cuda(), torch.empty((S, H, H, Cc), device="cuda", dtype=torch.bfloat16)
            _lib.check(hip.countr_instnorm_relu_pool_fwd(P(xin), P(y), P(stats), S, H, H, Cc, avg, 1e-5, 1, None, P(xh), 1, st()))
            _lib.check(hip.countr_instnorm_relu_pool_bwd(P(xh), P(dyp), P(stats), P(dx), S, H, H, Cc, avg, 1, None, 1, st()))
        else:
            xin = x.to(torch.bfloat16).cuda()
            _lib.check(hip.countr_instnorm_relu_pool_fwd(P(xin), P(y), P(stats), S, H, H, Cc, avg, 1e-5, 1, None, None, 0, st()))
            _lib.check(hip.countr_instnorm_relu_pool_bwd(P(xin), P(dyp), P(stats), P(dx), S, H, H, Cc, avg, 1, None, 0, st()))
        torch.cuda.synchronize()
        d = dx.double().cpu().reshape(-1)
        errs.append(1.0 - (d @ ref / (d.norm() * ref.norm())).item())
    print("1 - cos of dx vs fp64 (bf16 map, fp32 map + stored x-hat):", errs)
    assert errs[1] < 3e-3, errs
    assert errs[0] > 4 * errs[1], errs


@pytest.mark.parametrize("tdt,code,tol", DT)
def test_softmax_fwd_bwd(hip, tdt, code, tol):
    rows, n = 2 * 16 * 576, 576
    s = rnd((rows, n), 13, 3.0)
    sd = s.cuda()
    p = torch.empty((rows, n), device="cuda", dtype=tdt)
    _lib.check(hip.countr_softmax_fwd(P(sd), P(p), rows, n, code, st()))
    sr = s.double().requires_grad_(True)
    pr = torch.softmax(sr, -1)
    assert relerr(p, pr) < tol
    dp = rnd((rows, n), 14)
    ds = torch.empty_like(p)
    dpd = dp.cuda()
    _lib.check(hip.countr_softmax_bwd(P(p), P(dpd), P(ds), rows, n, 0.25, code, st()))
    pp = p.double().cpu()
    ref = pp * (dp.double() - (pp * dp.double()).sum(-1, keepdim=True)) * 0.25
    assert relerr(ds, ref) < tol


@pytest.mark.parametrize("S", [1, 3, 8, 9, 14, 37])
@pytest.mark.parametrize("tdt,code,tol", DT)
def test_cross_attention_fwd_bwd(hip, S, tdt, code, tol):
    """S <= 8: keys in registers; more (models_crossvit.py:111-128 has no limit, FSC_test_cross(few-shot).py --box_bound -1 passes every
    annotated box): online softmax over the key rows, dk / dv through per-wave LDS regions in key chunks."""
    B, N, D, Hh = 2, 576, 512, 16
    q = rnd((B, N, D), 15).to(tdt)
    k = rnd((B, S, D), 16).to(tdt)
    v = rnd((B, S, D), 17).to(tdt)
    do = rnd((B, N, D), 18).to(tdt)
    qd, kd, vd, dod = q.cuda(), k.cuda(), v.cuda(), do.cuda()
    o = torch.empty_like(qd)
    scale = 32 ** -0.5
    _lib.check(hip.countr_xattn_fwd(P(qd), P(kd), P(vd), P(o), B, N, S, D, Hh, D, scale, code, st()))
    qr = q.double().requires_grad_(True); kr = k.double().requires_grad_(True); vr = v.double().requires_grad_(True)
    qh = qr.reshape(B, N, Hh, 32).transpose(1, 2)
    kh = kr.reshape(B, S, Hh, 32).transpose(1, 2)
    vh = vr.reshape(B, S, Hh, 32).transpose(1, 2)
    a = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    oref = (a @ vh).transpose(1, 2).reshape(B, N, D)
    assert relerr(o, oref) < tol
    dq = torch.empty_like(qd)
    dk = torch.empty((B, S, D), device="cuda"); dv = torch.empty((B, S, D), device="cuda")
    ws = torch.empty(hip.countr_xattn_bwd_workspace_floats(B, N, S, D), device="cuda")
    dkb = torch.empty((B, S, D), device="cuda", dtype=torch.bfloat16); dvb = torch.empty_like(dkb)
    _lib.check(hip.countr_xattn_bwd(P(qd), P(kd), P(vd), P(dod), P(dq), P(dk), P(dv), P(ws), B, N, S, D, Hh, D, scale, code, P(dkb), P(dvb), st()))
    torch.cuda.synchronize()
    assert torch.equal(dkb, dk.to(torch.bfloat16)) and torch.equal(dvb, dv.to(torch.bfloat16))     # the optional bf16 copies
    oref.backward(do.double())
    assert relerr(dq, qr.grad) < tol
    assert relerr(dk, kr.grad) < 1e-3 and relerr(dv, vr.grad) < 1e-3


@pytest.mark.parametrize("tdt,code,tol", DT)
def test_im2patch_and_c3conv(hip, tdt, code, tol):
    B = 2
    img = torch.rand((B, 3, 384, 384), generator=torch.Generator().manual_seed(19))
    out = torch.empty((B * 576, 768), device="cuda", dtype=tdt)
    imgd = img.cuda()
    _lib.check(hip.countr_im2patch(P(imgd), P(out), B, 384, 384, 16, code, st()))
    ref = img.reshape(B, 3, 24, 16, 24, 16).permute(0, 2, 4, 1, 3, 5).reshape(B * 576, 768)
    assert relerr(out, ref) < (1e-7 if code == 0 else 4e-3)
    # first exemplar conv
    S = 4
    bx = torch.rand((S, 3, 64, 64), generator=torch.Generator().manual_seed(20))
    w = rnd((64, 3, 3, 3), 21, 0.2); bias = rnd((64,), 22, 0.1)
    y = torch.empty((S, 64, 64, 64), device="cuda", dtype=tdt)
    bxd, wd, bd = bx.cuda(), w.cuda(), bias.cuda()
    _lib.check(hip.countr_conv3x3_c3_fwd(P(bxd), P(wd), P(bd), P(y), S, 64, 64, code, st()))
    wr = w.double().requires_grad_(True); br = bias.double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(bx.double(), wr, br, padding=1)
    assert relerr(y, yr.permute(0, 2, 3, 1)) < tol
    dy = rnd((S, 64, 64, 64), 23).to(tdt)
    dw = torch.zeros_like(wd); db = torch.zeros_like(bd)
    ws = torch.empty(hip.countr_conv3x3_c3_wgrad_nblocks() * 64 * 28, device="cuda")
    dyd = dy.cuda()
    _lib.check(hip.countr_conv3x3_c3_wgrad(P(bxd), P(dyd), P(dw), P(db), P(ws), S, 64, 64, code, 0, st()))
    yr.backward(dy.double().permute(0, 3, 1, 2))
    assert relerr(dw, wr.grad) < 1e-4 and relerr(db, br.grad) < 1e-4


@pytest.mark.parametrize("S,H,W", [(24, 64, 64), (3, 20, 12), (2, 5, 8), (1, 64, 4)])
@pytest.mark.parametrize("tdt,code,tol", DT)
def test_first_exemplar_conv_quad_kernel_equals_the_per_pixel_kernel(hip, monkeypatch, tdt, code, tol, S, H, W):
    """conv3x3_c3_fwd4_kernel (four pixels of a row per thread: a tap's weights read from LDS once for four pixels, the 3 x 6 input
    window loaded once) against the per-pixel kernel (COUNTR_C3_QUAD=0): same taps in the same order per output -- IDENTICAL values --
    and against conv2d in fp64; image borders at every quad position, rows of one quad."""
    x = torch.rand((S, 3, H, W), generator=torch.Generator().manual_seed(61)).cuda()
    w = rnd((64, 3, 3, 3), 62, 0.2).cuda(); bias = rnd((64,), 63, 0.1).cuda()
    outs = {}
    for q in ("1", "0"):
        monkeypatch.setenv("COUNTR_C3_QUAD", q)
        y = torch.full((S, H, W, 64), float("nan"), device="cuda", dtype=tdt)
        _lib.check(hip.countr_conv3x3_c3_fwd(P(x), P(w), P(bias), P(y), S, H, W, code, st()))
        torch.cuda.synchronize()
        outs[q] = y
    assert torch.isfinite(outs["1"].float()).all() and torch.equal(outs["1"], outs["0"])
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    assert relerr(outs["1"], ref) < tol


@pytest.mark.parametrize("H,W", [(24, 20), (5, 7), (2, 2), (4, 2), (1, 6)])   # even sizes: 2x2-block kernels; odd: per-pixel kernels
@pytest.mark.parametrize("Cc", [256, 1])
@pytest.mark.parametrize("tdt,code,tol", DT)
def test_upsample2x_fwd_bwd(hip, Cc, tdt, code, tol, H, W):
    B = 2
    x = rnd((B, H, W, Cc), 24).to(tdt)
    xd = x.cuda()
    y = torch.empty((B, 2 * H, 2 * W, Cc), device="cuda", dtype=tdt)
    _lib.check(hip.countr_upsample2x_fwd(P(xd), P(y), B, H, W, Cc, code, st()))
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = R.upsample2x(xr)
    ref_t = torch.nn.functional.interpolate(x.double().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    assert relerr(yr, ref_t) < 1e-12  # oracle restatement == torch semantics
    assert relerr(y, yr.permute(0, 2, 3, 1)) < tol
    dy = rnd((B, 2 * H, 2 * W, Cc), 25).to(tdt)
    dx = torch.empty_like(xd)
    dyd = dy.cuda()
    _lib.check(hip.countr_upsample2x_bwd(P(dyd), P(dx), B, H, W, Cc, code, st()))
    yr.backward(dy.double().permute(0, 3, 1, 2))
    assert relerr(dx, xr.grad.permute(0, 2, 3, 1)) < tol


@pytest.mark.parametrize("B,H,W", [(8, 24, 24), (2, 96, 96), (5, 32, 40), (3, 64, 8)])
@pytest.mark.parametrize("tdt,code,tol", DT)
def test_upsample2x_chunk_kernel_equals_the_per_pixel_kernels(hip, monkeypatch, tdt, code, tol, B, H, W):
    """The density head's maps (256 channels, rows of whole 8-pixel chunks) run the bilinear adjoint with one workgroup per 8-pixel chunk
    of a coarse row, indices from wave-uniform arithmetic (upsample2x_bwd_chunk_kernel: the per-pixel kernel spends two thirds of its
    instructions on per-lane indices), and both directions in an XCD-contiguous block order; same loads and arithmetic in the same order
    -- the outputs must be IDENTICAL to the per-pixel kernels' (COUNTR_UP2_ROWS=0, with and without the XCD order) and meet the
    oracle's tolerance."""
    Cc = 256
    x = rnd((B, H, W, Cc), 31).to(tdt).cuda()
    dy = rnd((B, 2 * H, 2 * W, Cc), 32).to(tdt).cuda()
    outs = {}
    for name, env in (("rows", {}), ("pixel_xcd", {"COUNTR_UP2_ROWS": "0"}), ("pixel", {"COUNTR_UP2_ROWS": "0", "COUNTR_UP2_XCD": "0"})):
        for k in ("COUNTR_UP2_ROWS", "COUNTR_UP2_XCD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        y = torch.full((B, 2 * H, 2 * W, Cc), float("nan"), device="cuda", dtype=tdt)
        dx = torch.full((B, H, W, Cc), float("nan"), device="cuda", dtype=tdt)
        _lib.check(hip.countr_upsample2x_fwd(P(x), P(y), B, H, W, Cc, code, st()))
        _lib.check(hip.countr_upsample2x_bwd(P(dy), P(dx), B, H, W, Cc, code, st()))
        torch.cuda.synchronize()
        assert torch.isfinite(y.float()).all() and torch.isfinite(dx.float()).all(), name
        outs[name] = (y, dx)
    for name in ("pixel_xcd", "pixel"):
        assert torch.equal(outs["rows"][0], outs[name][0]), name
        assert torch.equal(outs["rows"][1], outs[name][1]), name
    xr = x.cpu().double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = R.upsample2x(xr)
    assert relerr(outs["rows"][0].cpu(), yr.permute(0, 2, 3, 1)) < tol
    yr.backward(dy.cpu().double().permute(0, 3, 1, 2))
    assert relerr(outs["rows"][1].cpu(), xr.grad.permute(0, 2, 3, 1)) < tol


@pytest.mark.parametrize("tdt,code,tol", DT)
def test_gelu_bwd_colsum(hip, tdt, code, tol):
    M, N = 1152, 2048
    pre = rnd((M, N), 26, 1.5).to(tdt)
    dh = rnd((M, N), 27).to(tdt)
    pred, dhd = pre.cuda(), dh.cuda()
    out = torch.empty_like(pred)
    _lib.check(hip.countr_gelu_bwd(P(dhd), P(pred), P(out), M * N, code, st()))
    pr = pre.double().requires_grad_(True)
    R.gelu(pr).backward(dh.double())
    assert relerr(out, pr.grad) < tol
    cs = torch.ones(N, device="cuda")
    ws = torch.empty(hip.countr_colsum_nparts() * N, device="cuda")
    _lib.check(hip.countr_colsum(P(dhd), P(cs), P(ws), M, N, code, 1, st()))
    assert relerr(cs, dh.double().sum(0) + 1) < 1e-4


@pytest.mark.parametrize("tdt,code,tol", DT)
def test_conv_dgrad_via_permuted_weights(hip, tdt, code, tol):
    # cast_permute mode 1 (OHWI) feeds the forward conv; mode 2 (dgrad form) turns dgrad into a forward conv
    B, H, W, Ci, Co = 2, 12, 12, 128, 256
    w = rnd((Co, Ci, 3, 3), 28, 0.05)
    wd = w.cuda()
    w_f = torch.empty((Co, 9, Ci), device="cuda", dtype=tdt)
    w_d = torch.empty((Ci, 9, Co), device="cuda", dtype=tdt)
    _lib.check(hip.countr_cast_permute(P(wd), P(w_f), w.numel(), 1, Co, Ci, 9, code, st()))
    _lib.check(hip.countr_cast_permute(P(wd), P(w_d), w.numel(), 2, Co, Ci, 9, code, st()))
    assert relerr(w_f, w.permute(0, 2, 3, 1).reshape(Co, 9, Ci)) < 5e-3
    dy = rnd((B, H, W, Co), 29).to(tdt)
    dyd = dy.cuda()
    dx = torch.empty((B * H * W, Ci), device="cuda", dtype=torch.float32)
    a = _lib.GemmArgs()
    a.A, a.B, a.C = dyd.data_ptr(), w_d.data_ptr(), dx.data_ptr()
    a.ldb, a.ldc = 9 * Co, Ci
    a.M, a.N, a.K = B * H * W, Ci, 9 * Co
    a.H, a.W, a.Cin = H, W, Co
    a.nbatch = 1; a.nb1 = 1; a.splitk = 1; a.alpha = 1.0
    _lib.check(hip.countr_gemm(C.byref(a), code, 2, 0, st()))
    wq = w_f.double().cpu().reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2)  # weights as the kernel sees them
    xr = torch.zeros((B, Ci, H, W), dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xr, wq, None, padding=1).backward(dy.double().permute(0, 3, 1, 2))
    assert relerr(dx, xr.grad.permute(0, 2, 3, 1).reshape(B * H * W, Ci)) < 1e-4


def test_masked_mse_and_adamw(hip):
    B, HW = 3, 384 * 384
    g = torch.Generator().manual_seed(30)
    pred = torch.rand((B, HW), generator=g); gt = torch.rand((B, HW), generator=g)
    mask = (torch.rand(HW, generator=g) < 0.8).float()
    pd, gd, md = pred.cuda(), gt.cuda(), mask.cuda()
    dp = torch.empty_like(pd)
    sums = torch.empty(1 + 2 * B, device="cuda")
    ws = torch.empty(hip.countr_masked_mse_workspace_floats(B), device="cuda")
    _lib.check(hip.countr_masked_mse(P(pd), P(gd), P(md), P(dp), P(sums), P(ws), B, HW, 1.0, st()))
    pr = pred.double().requires_grad_(True)
    loss = R.masked_mse_loss(pr.reshape(B, 384, 384), gt.double().reshape(B, 384, 384), mask.double().reshape(384, 384))
    loss.backward()
    assert abs(sums[0].item() - loss.item()) < 1e-5 * loss.item()
    assert relerr(dp, pr.grad) < 1e-5
    assert relerr(sums[1:1 + B], R.counts(pred.double())) < 1e-5
    assert relerr(sums[1 + B:], R.counts(gt.double())) < 1e-5
    # AdamW vs the real torch.optim.AdamW (FSC_finetune_cross.py:235): two weight-decay groups; the second range joins at step 2
    # (its gradient was None before -> skipped, own step counter) and gets a zero gradient at step 4 (torch 1.13 zero_grad()
    # semantics: zero tensor, still stepped); scalars and device-hyper forms; gradient norm (util/misc.py:289-301)
    n, n0 = 100003, 60000
    p0 = rnd((n,), 31); p = p0.cuda().clone()
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    shadow = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    gn = torch.zeros(hip.countr_adamw_gnorm_floats(), device="cuda")
    ta, tb = torch.nn.Parameter(p0[:n0].double().clone()), torch.nn.Parameter(p0[n0:].double().clone())
    opt = torch.optim.AdamW([{"params": [ta], "weight_decay": 0.05}, {"params": [tb], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
    starts = (C.c_int64 * 2)(0, n0); ends = (C.c_int64 * 2)(n0, n); wds = (C.c_float * 2)(0.05, 0.0)
    groups = (C.c_int * 2)(0, 1)
    tsteps = [0, 0]
    for step in range(1, 6):
        gsteps = rnd((n,), 40 + step)
        gdev = gsteps.cuda()
        lr = 1e-3 * step
        for grp in opt.param_groups:
            grp["lr"] = lr
        b_has = step >= 2 and step != 4          # range b: no gradient at step 1 (None), zero gradient at step 4
        ta.grad = gsteps[:n0].double().clone()
        if b_has:
            tb.grad = gsteps[n0:].double().clone()
        elif step == 4:
            tb.grad = torch.zeros_like(tb)       # what optimizer.zero_grad(set_to_none=False) leaves behind
        else:
            tb.grad = None
        opt.step()
        nr = 2 if step >= 2 else 1
        zeros = (C.c_int * 2)(0, 1 if step == 4 else 0)
        tsteps[0] += 1
        tsteps[1] += 1 if step >= 2 else 0
        bc = lambda t: (1 - 0.9 ** max(t, 1), 1 - 0.95 ** max(t, 1))
        hyper = torch.tensor([lr, *bc(tsteps[0]), 1.0, *bc(tsteps[1]), 1.0, 1.0], device="cuda")
        _lib.check(hip.countr_adamw_step(P(p), P(gdev), P(m), P(v), P(shadow), nr, starts, ends, wds, groups, zeros, 0.0, 0.9, 0.95, 1e-8,
                                         0, 0.0, P(hyper), P(gn), st()))
        pr_ = torch.cat([ta.detach(), tb.detach()])
        assert relerr(p, pr_) < 1e-5, step
        want = gsteps[:n0].double().pow(2).sum() + (gsteps[n0:].double().pow(2).sum() if b_has else 0.0)
        assert abs(gn[0].item() - want.sqrt().item()) < 1e-5 * want.sqrt().item(), step
    assert relerr(shadow, pr_) < 5e-3
    # scalar form (no device hyper): every range uses `step`
    p2 = p0.cuda().clone(); m2 = torch.zeros(n, device="cuda"); v2 = torch.zeros(n, device="cuda")
    g1 = rnd((n,), 77)
    _lib.check(hip.countr_adamw_step(P(p2), P(g1.cuda()), P(m2), P(v2), None, 2, starts, ends, wds, None, None, 1e-3, 0.9, 0.95, 1e-8,
                                     1, 1.0, None, None, st()))
    a, _, _ = R.adamw_step(p0.double()[:n0], g1.double()[:n0], torch.zeros(n0, dtype=torch.float64), torch.zeros(n0, dtype=torch.float64), 1, 1e-3, wd=0.05)
    b, _, _ = R.adamw_step(p0.double()[n0:], g1.double()[n0:], torch.zeros(n - n0, dtype=torch.float64), torch.zeros(n - n0, dtype=torch.float64), 1, 1e-3, wd=0.0)
    assert relerr(p2, torch.cat([a, b])) < 1e-5


@pytest.mark.parametrize("B,N,H,dh", [(2, 576, 12, 64), (2, 576, 16, 32), (1, 200, 3, 64), (1, 64, 2, 32), (2, 729, 16, 32)])
def test_flash_attention_fwd(hip, B, N, H, dh):
    """Fused bf16 attention vs fp64 softmax(q k^T * scale) v on the same (bf16-representable) inputs, including a
    spiked key that forces the online-softmax rescale branch and ragged N (not a multiple of the 64-key tile)."""
    qkv = rnd((B, N, 3, H, dh), 50, 1.0).to(torch.bfloat16)
    qkv[0, N // 2, 1, 0] = 6.0  # one key with large logits against every query -> running max jumps mid-sequence
    qd = qkv.cuda()
    out = torch.empty((B, N, H * dh), device="cuda", dtype=torch.bfloat16)
    lse = torch.empty((B, H, N), device="cuda")
    scale = dh ** -0.5
    _lib.check(hip.countr_attn_fwd(P(qd), P(out), P(lse), B, N, H, dh, scale, st()))
    q = qkv[:, :, 0].double().permute(0, 2, 1, 3); k = qkv[:, :, 1].double().permute(0, 2, 1, 3)
    v = qkv[:, :, 2].double().permute(0, 2, 1, 3)
    sc = (q @ k.transpose(-1, -2)) * scale
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * dh)
    assert relerr(out, ref) < 1.5e-2   # bf16 P and bf16 output rounding
    assert relerr(lse, torch.logsumexp(sc, -1)) < 1e-4


@pytest.mark.parametrize("B,N,H", [(2, 576, 12), (1, 64, 3), (1, 1088, 2)])
def test_flash_attention_fwd_prescaled_q(hip, B, N, H):
    """scale <= 0: q arrives multiplied by dh^-0.5 * log2(e) (rounded once to bf16, as the engine packs the frozen encoder's q
    projection); the kernel works in the exp2 domain with accumulators started at -m_ref.  Reference in fp64 on the SAME bf16
    inputs: softmax over 2^(q' k).  A spiked key forces the rescale branch (which must also shift the pending next-tile scores)."""
    dh = 64
    c = dh ** -0.5 * 1.4426950408889634
    qkv = rnd((B, N, 3, H, dh), 52, 1.0)
    qkv[0, N // 2, 1, 0] = 6.0
    qkv[0, N - 3, 1, H - 1] = -5.0
    qkv[:, :, 0] *= c
    qkv = qkv.to(torch.bfloat16)
    qd = qkv.cuda()
    out = torch.full((B, N, H * dh), float("nan"), device="cuda", dtype=torch.bfloat16)
    lse = torch.empty((B, H, N), device="cuda")
    _lib.check(hip.countr_attn_fwd(P(qd), P(out), P(lse), B, N, H, dh, 0.0, st()))
    q = qkv[:, :, 0].double().permute(0, 2, 1, 3); k = qkv[:, :, 1].double().permute(0, 2, 1, 3)
    v = qkv[:, :, 2].double().permute(0, 2, 1, 3)
    sc = (q @ k.transpose(-1, -2)) * 0.6931471805599453
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * dh)
    assert relerr(out, ref) < 1.5e-2
    assert relerr(lse, torch.logsumexp(sc, -1)) < 1e-4
    # unsupported shapes are refused, not silently computed with another scale
    assert hip.countr_attn_fwd(P(qd), P(out), P(lse), B, N, H * 2, 32, 0.0, st()) != 0


@pytest.mark.parametrize("N,H,dh,pre,spike", [(576, 12, 64, False, 40.0), (576, 12, 64, True, 40.0), (576, 12, 64, True, 250.0), (576, 16, 32, False, 60.0),
                                              (200, 3, 64, False, 60.0), (288, 12, 64, False, 250.0)])
def test_flash_attention_fwd_scores_far_above_the_first_key_tile(hip, N, H, dh, pre, spike):
    """Round 6: the bf16 kernel's steady state keeps the FIRST key tile's row max as the softmax reference (no running max: a later score
    d above it just makes P 2^d, exact while nothing overflows); every lane checks l < 2^64 and O finite at the end and a workgroup with
    a miss runs its strip again on the exact running-max loop.  Here keys behind the first tile score tens (spike 40 / 60) to hundreds
    (250: exp2 overflows to inf on the fast path) of log2 units above it for some query rows of ONE (batch, head) -- above AND below, the
    scores are signed -- while the other heads stay ordinary: both paths inside one launch, each against fp64."""
    B = 2
    c = dh ** -0.5 * 1.4426950408889634
    qkv = rnd((B, N, 3, H, dh), 57, 1.0)
    g = torch.Generator().manual_seed(58)
    sign = (torch.randint(0, 2, (dh,), generator=g) * 2 - 1).float()
    for j in (70, N // 2 + 5, N - 2):                       # all behind key tile 0
        qkv[0, j, 1, 1] = spike * sign
    qkv[1, N - 1, 1, H - 1] = spike * sign                  # ... and the very last key of another (batch, head)
    if pre:
        qkv[:, :, 0] *= c
    qkv = qkv.to(torch.bfloat16)
    qd = qkv.cuda()
    out = torch.full((B, N, H * dh), float("nan"), device="cuda", dtype=torch.bfloat16)
    lse = torch.empty((B, H, N), device="cuda")
    _lib.check(hip.countr_attn_fwd(P(qd), P(out), P(lse), B, N, H, dh, 0.0 if pre else dh ** -0.5, st()))
    q = qkv[:, :, 0].double().permute(0, 2, 1, 3); k = qkv[:, :, 1].double().permute(0, 2, 1, 3)
    v = qkv[:, :, 2].double().permute(0, 2, 1, 3)
    sc = (q @ k.transpose(-1, -2)) * (0.6931471805599453 if pre else dh ** -0.5)
    top = (sc[0, 1].max(-1).values - sc[0, 1, :, :64].max(-1).values) * 1.4426950408889634
    assert (top > 64).sum() > 8 and (top < 1).sum() > 8        # rows that need the exact path, rows that do not
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * dh)
    assert torch.isfinite(out.float()).all()
    assert relerr(out, ref) < 1.5e-2
    assert relerr(lse, torch.logsumexp(sc, -1)) < 1e-4


@pytest.mark.parametrize("B,N,H,dh", [(2, 576, 16, 32), (1, 576, 12, 64), (1, 200, 3, 32), (2, 64, 2, 64)])
def test_flash_attention_bwd(hip, B, N, H, dh):
    """Fused attention backward (dq, dk, dv in one packed tensor) vs fp64 autograd of softmax(q k^T * scale) v."""
    qkv = (rnd((B, N, 3, H, dh), 60, 1.0)).to(torch.bfloat16)
    qkv[0, N // 3, 1, 0] = 5.0
    do = rnd((B, N, H * dh), 61, 1.0).to(torch.bfloat16)
    qd, dod = qkv.cuda(), do.cuda()
    out = torch.empty((B, N, H * dh), device="cuda", dtype=torch.bfloat16)
    lse = torch.empty((B, H, N), device="cuda")
    delta = torch.empty((B, H, N), device="cuda")
    dqkv = torch.full((B, N, 3, H, dh), float("nan"), device="cuda", dtype=torch.bfloat16)
    scale = dh ** -0.5
    _lib.check(hip.countr_attn_fwd(P(qd), P(out), P(lse), B, N, H, dh, scale, st()))
    _lib.check(hip.countr_attn_bwd(P(qd), P(out), P(dod), P(lse), P(delta), P(dqkv), B, N, H, dh, scale, st()))
    x = qkv.double().requires_grad_(True)
    q = x[:, :, 0].permute(0, 2, 1, 3); k = x[:, :, 1].permute(0, 2, 1, 3); v = x[:, :, 2].permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * dh)
    ref.backward(do.double())
    assert torch.isfinite(dqkv.float()).all()
    for slot, name in enumerate(("dq", "dk", "dv")):
        assert relerr(dqkv[:, :, slot], x.grad[:, :, slot]) < 2.5e-2, name   # bf16 P/dS operands and bf16 outputs


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_conv_shadows_permutations(hip, dt):
    """countr_conv_shadows: torch OIHW conv weights -> the forward operand [Co][tap][Ci] (OHWI) and the dgrad operand
    [Ci][8 - tap][Co] (transposed, taps reversed: the flipped kernel of the transposed convolution), several convolutions per
    launch, ragged channel counts included (tiles are 32 x 32)."""
    import ctypes as C
    shapes = [(256, 512), (128, 64), (40, 24), (33, 65)]
    ws = [torch.randn(co, ci, 3, 3, device="cuda") for co, ci in shapes]
    wf = [torch.full((co, 9, ci), float("nan"), device="cuda", dtype=dt) for co, ci in shapes]
    wd = [torch.full((ci, 9, co), float("nan"), device="cuda", dtype=dt) for co, ci in shapes]
    n = len(shapes)
    vp, ip = C.c_void_p * n, C.c_int * n
    _lib.check(hip.countr_conv_shadows(n, vp(*[w.data_ptr() for w in ws]), vp(*[w.data_ptr() for w in wf]), vp(*[w.data_ptr() for w in wd]),
                                       ip(*[s[0] for s in shapes]), ip(*[s[1] for s in shapes]), ip(*[9] * n), 1 if dt == torch.bfloat16 else 0,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "conv_shadows")
    torch.cuda.synchronize()
    for w, f, d in zip(ws, wf, wd):
        co, ci = w.shape[:2]
        ref_f = w.reshape(co, ci, 9).permute(0, 2, 1).to(dt)
        ref_d = w.reshape(co, ci, 9).flip(2).permute(1, 2, 0).to(dt)
        assert torch.equal(f, ref_f) and torch.equal(d, ref_d)


def test_conv_shadows_transposes_linear_weights(hip):
    """taps = 1, wf = NULL: the entry is an nn.Linear weight [N][K] and wd receives its bf16 transpose [K][N] (the B operand of the
    (ROW, ROW) input-gradient GEMM); 20 weights in one launch beside a 3x3 convolution (the table holds up to 32)."""
    import ctypes as C
    lin = [(512, 512), (1536, 512), (2048, 512), (512, 2048), (40, 24)] * 4
    ws = [torch.randn(n, k, device="cuda") for n, k in lin]
    wt = [torch.full((k, n), float("nan"), device="cuda", dtype=torch.bfloat16) for n, k in lin]
    cw = torch.randn(64, 32, 3, 3, device="cuda")
    cf = torch.full((64, 9, 32), float("nan"), device="cuda", dtype=torch.bfloat16)
    cd = torch.full((32, 9, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
    n = len(lin) + 1
    vp, ip = C.c_void_p * n, C.c_int * n
    _lib.check(hip.countr_conv_shadows(n, vp(*([w.data_ptr() for w in ws] + [cw.data_ptr()])), vp(*([None] * len(lin) + [cf.data_ptr()])),
                                       vp(*([w.data_ptr() for w in wt] + [cd.data_ptr()])), ip(*([s[0] for s in lin] + [64])),
                                       ip(*([s[1] for s in lin] + [32])), ip(*([1] * len(lin) + [9])), 1,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "conv_shadows")
    torch.cuda.synchronize()
    for w, t in zip(ws, wt):
        assert torch.equal(t, w.t().contiguous().to(torch.bfloat16))
    assert torch.equal(cf, cw.reshape(64, 32, 9).permute(0, 2, 1).to(torch.bfloat16))
    assert torch.equal(cd, cw.reshape(64, 32, 9).flip(2).permute(1, 2, 0).to(torch.bfloat16))


def test_transpose16_many_matrices_one_launch(hip):
    """countr_transpose16 (ABI 8): the W^T shadows from the 16-bit shadow W -- 82 matrices of the MAE model's shapes (and 96, the table's
    size) in one launch, bit-identical to torch's transpose of the same 16-bit values and to what countr_conv_shadows (taps = 1) makes
    of the fp32 master; shapes that are not multiples of 64 and more than 96 matrices are refused."""
    import ctypes as C
    shapes = ([(2304, 768), (768, 768), (3072, 768), (768, 3072)] * 12 + [(1536, 512), (512, 512), (2048, 512), (512, 2048)] * 8 + [(512, 768), (768, 512)])[:82]
    shapes += [(64, 64), (128, 192)] * 7
    assert len(shapes) == 96
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (82, 96, 1):
        sh = shapes[:n]
        ws = [torch.randn(r, c, device="cuda") for r, c in sh]
        w16 = [w.to(torch.bfloat16) for w in ws]
        wt = [torch.full((c, r), float("nan"), device="cuda", dtype=torch.bfloat16) for r, c in sh]
        vp, ip = C.c_void_p * n, C.c_int * n
        _lib.check(hip.countr_transpose16(n, vp(*[w.data_ptr() for w in w16]), vp(*[w.data_ptr() for w in wt]), ip(*[s_[0] for s_ in sh]),
                                          ip(*[s_[1] for s_ in sh]), st), "transpose16")
        torch.cuda.synchronize()
        for w, t in zip(w16, wt):
            assert torch.equal(t, w.t().contiguous())
        if n == 82:      # against the fp32 route, 32 at a time
            m = 32
            old = [torch.full((c, r), float("nan"), device="cuda", dtype=torch.bfloat16) for r, c in sh[:m]]
            vpm, ipm = C.c_void_p * m, C.c_int * m
            _lib.check(hip.countr_conv_shadows(m, vpm(*[w.data_ptr() for w in ws[:m]]), vpm(*([None] * m)), vpm(*[w.data_ptr() for w in old]),
                                               ipm(*[s_[0] for s_ in sh[:m]]), ipm(*[s_[1] for s_ in sh[:m]]), ipm(*([1] * m)), 1, st), "conv_shadows")
            torch.cuda.synchronize()
            for a, b in zip(old, wt[:m]):
                assert torch.equal(a, b)
    a = torch.zeros(40, 64, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(64, 40, device="cuda", dtype=torch.bfloat16)
    one_p, one_i = C.c_void_p * 1, C.c_int * 1
    assert hip.countr_transpose16(1, one_p(a.data_ptr()), one_p(b.data_ptr()), one_i(40), one_i(64), st) != 0
    assert b"multiples of 64" in hip.countr_last_error()
    assert hip.countr_transpose16(97, None, None, None, None, st) != 0


def test_copy_multi(hip):
    """countr_copy_multi: several dense device-to-device copies in one launch (the staging of a batch), sizes from 16 bytes to 14 MB,
    nothing outside the destinations touched."""
    import ctypes as C
    sizes = [4, 3 * 384 * 384 * 8, 8 * 384 * 384, 384 * 384, 8 * 3 * 3 * 64 * 64, 1028, 12]
    src = [torch.randn(n, device="cuda") for n in sizes]
    dst = [torch.full((n + 8,), float("nan"), device="cuda") for n in sizes]
    n = len(sizes)
    vp = C.c_void_p * n
    _lib.check(hip.countr_copy_multi(n, vp(*[s.data_ptr() for s in src]), vp(*[d.data_ptr() + 16 for d in dst]),
                                     (C.c_int64 * n)(*[4 * k for k in sizes]), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "copy_multi")
    torch.cuda.synchronize()
    for s, d in zip(src, dst):
        assert torch.equal(d[4:4 + s.numel()], s) and torch.isnan(d[:4]).all() and torch.isnan(d[4 + s.numel():]).all()
    assert hip.countr_copy_multi(1, vp(*[src[0].data_ptr()] * n), vp(*[dst[0].data_ptr() + 4] * n), (C.c_int64 * n)(*[16] * n), None) != 0   # misaligned


def test_copy_multi_zero_fill(hip):
    """countr_copy_multi with a NULL source zero-fills its destination (the gradient range of a conditional parameter set this rank
    did not use while another rank did: per-rank shot_num), next to ordinary copies in the same launch."""
    import ctypes as C
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = torch.randn(5000, device="cuda")
    b = torch.full((1 << 20,), float("nan"), device="cuda")
    keep = torch.randn(1000, device="cuda")
    out = torch.empty(1000, device="cuda")
    vp, i64 = C.c_void_p * 3, C.c_int64 * 3
    _lib.check(hip.countr_copy_multi(3, vp(None, keep.data_ptr(), None), vp(a.data_ptr() + 16, out.data_ptr(), b.data_ptr()),
                                     i64(4 * 4000, 4 * 1000, 4 * (1 << 20)), st), "copy_multi")
    torch.cuda.synchronize()
    assert (a[4:4004] == 0).all() and (a[:4] != 0).any() and (a[4004:] != 0).any() and (b == 0).all() and torch.equal(out, keep)


@pytest.mark.parametrize("widths", [[672, 672, 512], [386, 1000], [384]])
def test_window_gather_and_blend_equal_the_torch_slicing(hip, widths):
    """countr_window_gather / countr_window_blend (the sliding-window test path, FSC_test_cross(few-shot).py:326-349) against the tensor
    slicing and the sequential blend of countr_amd.inference (which follow the reference's loop): bit for bit, for widths that allow
    16-byte row segments (672, 512) and widths that do not (386: second window at column 2; 1000: last window snapped to 616)."""
    import ctypes as Ct
    from countr_amd import inference
    H = 384
    torch.manual_seed(5)
    imgs = [torch.randn(1, 3, H, w, device="cuda") for w in widths]
    plan = [(i, s0) for i, im in enumerate(imgs) for s0 in inference.window_starts(im.shape[-1])]
    nw = len(plan)
    wins = torch.full((nw + 1, 3, H, 384), float("nan"), device="cuda")
    frames = (Ct.c_void_p * nw)(*[imgs[i].data_ptr() for i, _ in plan])
    ws = (Ct.c_int * nw)(*[imgs[i].shape[-1] for i, _ in plan])
    sarr = (Ct.c_int * nw)(*[s0 for _, s0 in plan])
    _lib.check(hip.countr_window_gather(frames, ws, sarr, nw, H, wins.data_ptr(), st()), "gather")
    torch.cuda.synchronize()
    assert torch.isnan(wins[nw]).all()
    for j, (i, s0) in enumerate(plan):
        assert torch.equal(wins[j], imgs[i][0, :, :, s0:s0 + 384]), j
    bad = (Ct.c_int * nw)(*[w for w in ws])           # a window that leaves its image is refused
    assert hip.countr_window_gather(frames, ws, bad, nw, H, wins.data_ptr(), st()) != 0
    # blend: n images of one width
    for w in sorted(set(widths)):
        starts = inference.window_starts(w)
        n = 3
        outs = torch.randn(n * len(starts), H, 384, device="cuda")
        dm = torch.full((n, H, w), float("nan"), device="cuda")
        sums = torch.full((n,), float("nan"), device="cuda")
        wsp = torch.empty(n * hip.countr_window_blend_blocks(H, w), device="cuda")
        arr = (Ct.c_int * len(starts))(*starts)
        _lib.check(hip.countr_window_blend(outs.data_ptr(), n, len(starts), arr, H, w, dm.data_ptr(), sums.data_ptr(), wsp.data_ptr(), st()), "blend")
        torch.cuda.synchronize()
        for k in range(n):
            ref = inference.blend_windows(outs[k * len(starts):(k + 1) * len(starts)], starts, w, H)
            assert torch.equal(dm[k], ref), (w, k)
            assert abs(sums[k].item() - ref.double().sum().item()) <= 1e-5 * ref.abs().double().sum().item()



def test_step_prologue_ring_copies_mask_and_counter(hip):
    """countr_step_prologue (the first node of a captured step): execution k reads record k % slots of the host ring -- copies (incl.
    a zero fill and an odd size), AdamW scalars, the Philox loss mask bit for bit against oracle/philox.py -- counts its own executions
    on the device, and does so under hipGraph replay too (arguments frozen, records rewritten between replays)."""
    import numpy as np
    from oracle.philox import loss_mask
    L = hip
    RB, NBLK, SLOTS = L.countr_step_prologue_record_bytes(), L.countr_step_prologue_copy_blocks(), 3
    assert RB == 256
    ring = torch.zeros(SLOTS * RB, dtype=torch.uint8).pin_memory()
    view = ring.numpy()
    counter = torch.zeros(2 + 32, dtype=torch.int64, device="cuda")
    hyper = torch.zeros(8, device="cuda")
    mask = torch.full((384 * 384,), -1.0, device="cuda")
    g = torch.Generator().manual_seed(3)
    srcs = [torch.rand(n, generator=g).cuda() for n in (8 * 3 * 384 * 384, 8 * 384 * 384, 1236)]
    dsts = [torch.zeros_like(t) for t in srcs] + [torch.ones(4096, device="cuda")]
    seed = 0x1234_5678_9abc

    def fill(k, t, draw, scale):
        rec = view[k * RB:(k + 1) * RB]
        u64, i32, u32 = rec[0:144].view(np.uint64), rec[144:176].view(np.int32), rec[176:192].view(np.uint32)
        pairs = [(s_.data_ptr(), d_.data_ptr(), s_.numel() * 4) for s_, d_ in zip(srcs, dsts)] + [(0, dsts[3].data_ptr(), 4096 * 4)]
        first = [0, 700, 1000, 1100]
        for i in range(6):
            s_, d_, b_ = pairs[i] if i < 4 else (0, 0, 0)
            u64[i], u64[6 + i], u64[12 + i] = s_, d_, b_ // 16
            i32[i] = first[i] if i < 4 else NBLK
        i32[6], i32[7] = 4, int(draw)
        u32[0], u32[1], u32[2], u32[3] = seed & 0xFFFFFFFF, seed >> 32, t, 0
        rec[192:224].view(np.float32)[:] = np.arange(8, dtype=np.float32) * scale
        rec[224:228].view(np.uint32)[0] = int(0.8 * 4294967296.0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (ring.data_ptr(), SLOTS, counter.data_ptr(), hyper.data_ptr(), mask.data_ptr(), mask.numel())
    for k in range(4):                     # eager executions 0..3 (slot 0 is reused by execution 3)
        for s_ in srcs:
            s_.mul_(0.5).add_(k)
        for d_ in dsts[:3]:
            d_.zero_()
        dsts[3].fill_(1.0)
        fill(k % SLOTS, k, draw=(k != 1), scale=k + 1.0)
        torch.cuda.synchronize()
        _lib.check(L.countr_step_prologue(*args, st), "prologue")
        torch.cuda.synchronize()
        assert counter[:2].tolist() == [k + 1, 0]
        for s_, d_ in zip(srcs, dsts):
            assert torch.equal(s_, d_)
        assert float(dsts[3].abs().max()) == 0.0
        assert torch.equal(hyper.cpu(), torch.arange(8.0) * (k + 1.0))
        want = loss_mask(seed, 0 if k <= 1 else k)          # k = 1 draws nothing: the mask of execution 0 stays
        assert np.array_equal(mask.cpu().numpy(), want), k
    # the same launch captured once and replayed: executions 4, 5, 6
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            _lib.check(L.countr_step_prologue(*args, C.c_void_p(side.cuda_stream)), "prologue(capture)")
    torch.cuda.synchronize()
    assert counter[:2].tolist() == [4, 0]                        # capture executes nothing
    for k in range(4, 7):
        srcs[2].add_(1.0)
        fill(k % SLOTS, k, draw=True, scale=0.25 * k)
        torch.cuda.synchronize()
        gr.replay()
        torch.cuda.synchronize()
        assert counter[:2].tolist() == [k + 1, 0]
        assert torch.equal(srcs[2], dsts[2]) and torch.equal(hyper.cpu(), torch.arange(8.0) * 0.25 * k)
        assert np.array_equal(mask.cpu().numpy(), loss_mask(seed, k)), k
    assert L.countr_step_prologue(None, SLOTS, counter.data_ptr(), hyper.data_ptr(), None, 0, st) != 0
