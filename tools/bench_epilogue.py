"""Cost of the GEMM epilogue options (bias / GELU / residual / fp32 output) on the encoder shapes, 50 back-to-back launches each."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def mk(*shape): return (torch.rand(shape, device="cuda") - 0.5).to(torch.bfloat16)
M = 4608
for name, N, K in (("qkv", 2304, 768), ("fc1", 3072, 768), ("proj", 768, 768), ("fc2", 768, 3072)):
    A_, B_ = mk(M, K), mk(N, K); bvec = torch.rand(N, device="cuda"); res = torch.rand(M, N, device="cuda")
    for cfg in ("plain bf16", "bias bf16", "bias+gelu bf16", "plain f32out", "bias+resid f32out"):
        odt = torch.float32 if "f32out" in cfg else torch.bfloat16
        Cc = torch.empty((M, N), device="cuda", dtype=odt)
        a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
        a.A, a.B, a.C = A_.data_ptr(), B_.data_ptr(), Cc.data_ptr(); a.lda = a.ldb = K; a.ldc = N; a.ldres = N; a.M, a.N, a.K = M, N, K
        a.out_bf16 = int(odt == torch.bfloat16)
        a.bias = bvec.data_ptr() if "bias" in cfg else None
        a.act = 1 if "gelu" in cfg else 0
        a.resid = res.data_ptr() if "resid" in cfg else None
        for _ in range(3): L.countr_gemm(C.byref(a), 1, 0, 0, st())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): L.countr_gemm(C.byref(a), 1, 0, 0, st())
        e1.record(); torch.cuda.synchronize()
        print("%-5s %-20s %6.1f us" % (name, cfg, e0.elapsed_time(e1) * 1e3 / 50), flush=True)
