"""Conv weight-gradient GEMM (COL, IM2COL) timing: lean kernel (conv_wgrad.hip) forms 1 / 2 / 3 against gemm_kernel, on the density-head shapes
of the B = 8 finetune step.  python tools/bench_wgrad.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib

L = _lib.lib()
L.countr_init(0)
st = torch.cuda.current_stream().cuda_stream
shapes = [(8, 192, 192, 256, 256), (8, 96, 96, 256, 256), (8, 48, 48, 256, 256), (8, 24, 24, 512, 256)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
tag = os.path.basename(os.environ.get("COUNTR_LIB", "default"))
for (B, H, W, Cin, Cout) in shapes:
    dy = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16)
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    P, N = B * H * W, 9 * Cin
    for name, env, bias in [("generic", dict(COUNTR_LEAN_WGRAD="0"), True), ("lean", dict(COUNTR_LEAN_WGRAD="1"), True),
                            ("lean-form2", dict(COUNTR_LEAN_WGRAD="1", COUNTR_LEAN_WGRAD_FORM="2"), True),
                            ("lean-form3", dict(COUNTR_LEAN_WGRAD="1", COUNTR_LEAN_WGRAD_FORM="3"), True),
                            ("lean-nobias", dict(COUNTR_LEAN_WGRAD="1"), False)]:
        os.environ.pop("COUNTR_LEAN_WGRAD_FORM", None)
        os.environ.update(env)
        a = _lib.GemmArgs()
        a.A, a.B = dy.data_ptr(), x.data_ptr()
        a.lda, a.ldc = Cout, N
        a.M, a.N, a.K = Cout, N, P
        a.H, a.W, a.Cin = H, W, Cin
        a.alpha = 1.0
        a.nbatch = 1; a.nb1 = 1
        tiles = L.countr_gemm_tiles(C.byref(a), 1, 1, 3)
        a.splitk = sk_ = max(1, min(64, 256 // tiles, P // 64))
        slabs = L.countr_gemm_rowsum_slabs(C.byref(a), 1, 1, 3)
        part = torch.empty(sk_, Cout, N, device="cuda")
        rs = torch.empty(slabs, Cout, device="cuda")
        a.partial = part.data_ptr()
        if bias:
            a.rowsum_partial, a.rowsum_slabs = rs.data_ptr(), slabs
        for _ in range(3):
            _lib.check(L.countr_gemm(C.byref(a), 1, 1, 3, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            L.countr_gemm(C.byref(a), 1, 1, 3, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print("[%s] %dx%d Cin %d: %-12s tiles %3d sk %3d  %7.1f us  %.2f PF" % (tag, H, W, Cin, name, tiles, sk_, us, 2.0 * Cout * N * P / us / 1e9), flush=True)
