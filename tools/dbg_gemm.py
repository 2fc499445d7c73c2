import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
for (M, N, K) in ((576, 768, 768), (128, 128, 768), (128, 128, 128), (128, 128, 192), (128, 128, 256), (4608, 2304, 768)):
    A = (torch.rand(M, K) * 2 - 1).to(torch.bfloat16).cuda(); B = (torch.rand(N, K) * 2 - 1).to(torch.bfloat16).cuda()
    for rep in range(3):
        out = torch.zeros((M, N), device="cuda")
        a = _lib.GemmArgs(); a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), out.data_ptr(); a.lda = a.ldb = K; a.ldc = N
        a.M, a.N, a.K = M, N, K; a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        _lib.check(L.countr_gemm(C.byref(a), 1, 0, 0, st())); torch.cuda.synchronize()
        ref = A.double() @ B.double().t()
        d = (out.double() - ref).abs()
        bad = (d > 1e-3 * ref.abs().max()).nonzero()
        print(M, N, K, "rep", rep, "maxerr %.3e" % d.max().item(), "bad elems", bad.shape[0], "first", bad[:3].tolist(), flush=True)
