"""Split-K factor sweep for the decoder wgrad GEMMs (dW = dy^T x, K = 4608 tokens): confirms Engine._splitk (<= 256 workgroups)."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def mk(*shape): return (torch.rand(shape, device="cuda") - 0.5).to(torch.bfloat16)
M = 4608
for Nl, Kl in ((512, 512), (1536, 512), (2048, 512), (512, 2048)):
    dy, x = mk(M, Nl), mk(M, Kl)
    tiles = (Nl // 128) * (Kl // 128)
    for sk in sorted({1, 2, 4, 8, 16, max(1, 256 // tiles)}):
        if tiles * sk > 512: continue
        part = torch.empty((sk, Nl, Kl), device="cuda"); rs = torch.empty((sk, Nl), device="cuda")
        a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1
        a.A, a.B, a.partial = dy.data_ptr(), x.data_ptr(), part.data_ptr(); a.lda, a.ldb, a.ldc = Nl, Kl, Kl
        a.M, a.N, a.K = Nl, Kl, M; a.splitk = sk; a.rowsum_partial = rs.data_ptr()
        for _ in range(3): L.countr_gemm(C.byref(a), 1, 1, 1, st())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): L.countr_gemm(C.byref(a), 1, 1, 1, st())
        e1.record(); torch.cuda.synchronize()
        print("wgrad %dx%d K=4608 sk%-2d (%3d wgs): %5.1f us%s" % (Nl, Kl, sk, tiles * sk, e0.elapsed_time(e1) * 1e3 / 40, "   <- policy" if sk == max(1, min(64, 256 // tiles)) else ""), flush=True)
