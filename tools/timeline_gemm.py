"""Kernel-level timeline of the encoder GEMMs (library built with -DCOUNTR_GEMM_STAMP): per workgroup absolute s_memtime at kernel
entry, main-loop start, main-loop end and exit.  Shows where the time outside the loop goes: dispatch ramp, prologue, epilogue, and
the epilogue's halves.  (s_memtime counters of different XCDs are not aligned: only per-workgroup differences are used.)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def mk(*shape): return (torch.rand(shape, device="cuda") - 0.5).to(torch.bfloat16)
M = 4608
for name, N, K, opts in (("qkv", 2304, 768, "bias"), ("fc1", 3072, 768, "bias+gelu"), ("proj", 768, 768, "bias+resid"), ("fc2", 768, 3072, "bias+resid")):
    A_, B_ = mk(M, K), mk(N, K); bias = torch.rand(N, device="cuda"); res = torch.rand(M, N, device="cuda")
    obf = 0 if "resid" in opts else 1
    Cc = torch.empty((M, N), device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.C = A_.data_ptr(), B_.data_ptr(), Cc.data_ptr(); a.lda = a.ldb = K; a.ldc = N; a.ldres = N; a.M, a.N, a.K = M, N, K; a.out_bf16 = obf
    a.bias = bias.data_ptr(); a.act = 1 if "gelu" in opts else 0; a.resid = res.data_ptr() if "resid" in opts else None
    dbg = torch.zeros(500000 + 4096 * 16, device="cuda")
    a.sC1 = dbg.data_ptr()
    for _ in range(5): _lib.check(L.countr_gemm(C.byref(a), 1, 0, 0, st()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(L.countr_gemm(C.byref(a), 1, 0, 0, st())); e1.record(); torch.cuda.synchronize()
    x = dbg[500000:].cpu().numpy().view(np.uint64).reshape(-1, 8)
    t = dbg[400000:500000].cpu().numpy().view(np.uint64).reshape(-1, 4)
    n = min(len(x), len(t))
    x, t = x[:n], t[:n]
    x = x[t[:, 3] > 0].astype(np.int64)
    t = t[t[:, 3] > 0].astype(np.int64)
    t0 = t[:, 0].min()
    ent, l0, l1, ex = (t[:, k] - t0 for k in range(4))
    nwg = len(t)
    q = lambda v: "%6.0f %6.0f %6.0f" % tuple(np.percentile(v, [10, 50, 90]))
    print("%s (%s): %d workgroups, %.1f us by events (stamped build); cycles p10 / p50 / p90 per workgroup" % (name, opts, nwg, e0.elapsed_time(e1) * 1e3))
    print("   prologue (entry -> loop) : %s" % q(l0 - ent))
    print("   main loop                : %s" % q(l1 - l0))
    print("   epilogue (loop end->exit): %s" % q(ex - l1))
    # finer stamps: 0 before the residual prefetch, 1 / 2 around the loader init, 3 bias loaded, 4 first half staged in LDS, 5 first half
    # stored, 6 epilogue done, 7 before the final store-acknowledgement wait
    xs = x - t[:, :1]                      # all relative to the workgroup's own entry (the counters of different XCDs are not aligned)
    r0, r1, rx = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0], t[:, 3] - t[:, 0]
    m = lambda v: "%6.0f" % np.median(v)
    print("   setup: entry->resid prefetch %s | prefetch issue %s | loader init %s | ->loop %s" % (m(xs[:, 0]), m(xs[:, 1] - xs[:, 0]), m(xs[:, 2] - xs[:, 1]), m(r0 - xs[:, 2])))
    print("   epilogue: loop end->bias loaded %s | first half to LDS %s | first half stored %s | second half %s | ack wait %s"
          % (m(xs[:, 3] - r1), m(xs[:, 4] - xs[:, 3]), m(xs[:, 5] - xs[:, 4]), m(xs[:, 6] - xs[:, 5]), m(rx - xs[:, 7])))
