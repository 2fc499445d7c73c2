"""fc1 / qkv GEMMs with warm vs evicted caches (a 1 GiB fill between launches), per-launch HIP-event timing."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def mk(*shape): return (torch.rand(shape, device="cuda") - 0.5).to(torch.bfloat16)
trash = torch.empty(1 << 28, device="cuda")   # 1 GiB
M = 4608
for name, N, K, act, bias in (("fc1", 3072, 768, 1, True), ("qkv", 2304, 768, 0, True), ("fc1 plain", 3072, 768, 0, False)):
    A_, B_ = mk(M, K), mk(N, K); Cc = torch.empty((M, N), device="cuda", dtype=torch.bfloat16); bvec = torch.rand(N, device="cuda")
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.C = A_.data_ptr(), B_.data_ptr(), Cc.data_ptr(); a.lda = a.ldb = K; a.ldc = N; a.M, a.N, a.K = M, N, K; a.out_bf16 = 1
    a.act = act; a.bias = bvec.data_ptr() if bias else None
    for mode in ("warm", "cold"):
        ts = []
        for it in range(12):
            if mode == "cold": trash.fill_(float(it))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); L.countr_gemm(C.byref(a), 1, 0, 0, st()); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        print("%-10s %s: median %.1f us  min %.1f" % (name, mode, ts[len(ts) // 2], ts[0]), flush=True)
