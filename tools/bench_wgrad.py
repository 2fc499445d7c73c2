"""Conv weight-gradient GEMM (COL, IM2COL) timing: lean kernel (conv_wgrad.hip) forms 1 / 2 against gemm_kernel, on the density-head shapes
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
for (B, H, W, Cin, Cout) in shapes:
    dy = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16)
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    P, N = B * H * W, 9 * Cin
    for name, env, sk in [("generic", dict(COUNTR_LEAN_WGRAD="0"), None), ("lean128", dict(COUNTR_LEAN_WGRAD="1", COUNTR_LEAN_WGRAD_FORM="1"), None),
                          ("lean256", dict(COUNTR_LEAN_WGRAD="1", COUNTR_LEAN_WGRAD_FORM="2"), None),
                          ("lean256x2", dict(COUNTR_LEAN_WGRAD="1", COUNTR_LEAN_WGRAD_FORM="2"), "x2")]:
        os.environ.update(env)
        tiles = (Cout // 128) * (N // 128)
        sk_ = max(1, min(256 // tiles, P // 64 // 4))
        if sk == "x2":
            sk_ = max(1, min(512 // tiles, P // 64 // 4))
        a = _lib.GemmArgs()
        a.A, a.B = dy.data_ptr(), x.data_ptr()
        a.lda, a.ldc = Cout, N
        a.M, a.N, a.K = Cout, N, P
        a.H, a.W, a.Cin = H, W, Cin
        a.alpha = 1.0
        a.nbatch = 1; a.nb1 = 1; a.splitk = sk_
        slabs = L.countr_gemm_rowsum_slabs(C.byref(a), 1, 1, 3)
        part = torch.empty(sk_, Cout, N, device="cuda")
        rs = torch.empty(slabs, Cout, device="cuda")
        a.partial, a.rowsum_partial, a.rowsum_slabs = part.data_ptr(), rs.data_ptr(), slabs
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
        print("%dx%d Cin %d: %-10s sk %3d  %7.1f us  %.2f PF" % (H, W, Cin, name, sk_, us, 2.0 * Cout * N * P / us / 1e9), flush=True)
