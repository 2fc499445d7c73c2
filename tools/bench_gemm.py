"""Micro-benchmark of countr_gemm on the hot shapes of the finetune step (bf16).  Prints TF/s per shape."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
FILTER = sys.argv[1] if len(sys.argv) > 1 else ""
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
def run(name, a, code, ma, mb, flops, iters=None):
    iters = iters or ITERS
    if FILTER and FILTER not in name: return
    for _ in range(3): _lib.check(L.countr_gemm(C.byref(a), code, ma, mb, st()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.countr_gemm(C.byref(a), code, ma, mb, st())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%-34s %8.1f us  %7.1f TF/s" % (name, us, flops / us / 1e6), flush=True)
def mk(*shape): return (torch.rand(shape, device="cuda") - 0.5).to(torch.bfloat16)
def base():
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1; return a
B = 8; M = B * 576
for name, N, K in (("qkv 4608x2304x768", 2304, 768), ("proj 4608x768x768", 768, 768), ("fc1 4608x3072x768", 3072, 768),
                   ("fc2 4608x768x3072", 768, 3072), ("dec fc1 4608x2048x512", 2048, 512), ("dec proj 4608x512x512", 512, 512)):
    A_, B_ = mk(M, K), mk(N, K); Cc = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    a = base(); a.A, a.B, a.C = A_.data_ptr(), B_.data_ptr(), Cc.data_ptr(); a.lda = a.ldb = K; a.ldc = N
    a.M, a.N, a.K = M, N, K; a.out_bf16 = 1
    run(name, a, 1, 0, 0, 2.0 * M * N * K)
# MAE pretrain encoder shapes (288 kept tokens per image): small grids
Mp = B * 288
for name, N, K in (("pre qkv 2304x2304x768", 2304, 768), ("pre proj 2304x768x768", 768, 768), ("pre fc1 2304x3072x768", 3072, 768),
                   ("pre fc2 2304x768x3072", 768, 3072)):
    A_, B_ = mk(Mp, K), mk(N, K); Cc = torch.empty((Mp, N), device="cuda", dtype=torch.bfloat16)
    a = base(); a.A, a.B, a.C = A_.data_ptr(), B_.data_ptr(), Cc.data_ptr(); a.lda = a.ldb = K; a.ldc = N
    a.M, a.N, a.K = Mp, N, K; a.out_bf16 = 1
    run(name, a, 1, 0, 0, 2.0 * Mp * N * K)
for name, N, K in (("pre dgrad fc1 2304x768x3072", 3072, 768), ("pre dgrad fc2 2304x3072x768", 768, 3072), ("pre dgrad qkv 2304x768x2304", 2304, 768)):
    dy, w = mk(Mp, N), mk(N, K); dx = torch.empty((Mp, K), device="cuda", dtype=torch.bfloat16)
    a = base(); a.A, a.B, a.C = dy.data_ptr(), w.data_ptr(), dx.data_ptr(); a.lda, a.ldb, a.ldc = N, K, K; a.M, a.N, a.K = Mp, K, N; a.out_bf16 = 1
    run(name + " (row,col)", a, 1, 0, 1, 2.0 * Mp * N * K)
# dgrad-style ROW x COL and wgrad-style COL x COL
N, K = 2048, 512
dy, w = mk(M, N), mk(N, K); dx = torch.empty((M, K), device="cuda", dtype=torch.bfloat16)
a = base(); a.A, a.B, a.C = dy.data_ptr(), w.data_ptr(), dx.data_ptr(); a.lda, a.ldb, a.ldc = N, K, K; a.M, a.N, a.K = M, K, N; a.out_bf16 = 1
run("dgrad 4608x512x2048 (row,col)", a, 1, 0, 1, 2.0 * M * N * K)
x = mk(M, K); part = torch.empty((8, N, K), device="cuda")
a = base(); a.A, a.B, a.partial = dy.data_ptr(), x.data_ptr(), part.data_ptr(); a.lda, a.ldb, a.ldc = N, K, K; a.M, a.N, a.K = N, K, M; a.splitk = 8
run("wgrad 2048x512x4608 sk8 (col,col)", a, 1, 1, 1, 2.0 * M * N * K)
# MAE pretrain wgrad shapes (K = 2304 kept tokens): dW[N_lin, K_lin] = dy^T x, 128x128 tiles x split-K
for name, Nl, Kl in (("pre wgrad fc1 3072x768", 3072, 768), ("pre wgrad qkv 2304x768", 2304, 768), ("pre wgrad proj 768x768", 768, 768)):
    dyp, xp = mk(Mp, Nl), mk(Mp, Kl)
    tiles = (Nl // 128) * (Kl // 128)
    for sk in sorted({1, 2, max(1, 256 // tiles), max(1, round(512 / tiles))}):
        part = torch.empty((sk, Nl, Kl), device="cuda")
        a = base(); a.A, a.B, a.partial = dyp.data_ptr(), xp.data_ptr(), part.data_ptr(); a.lda, a.ldb, a.ldc = Nl, Kl, Kl
        a.M, a.N, a.K = Nl, Kl, Mp; a.splitk = sk
        run("%s x2304 sk%d (%d wgs)" % (name, sk, tiles * sk), a, 1, 1, 1, 2.0 * Mp * Nl * Kl)
for Hs, Cin in ((192, 256), (96, 256), (24, 512)):
    xx = mk(B, Hs, Hs, Cin); w = mk(256, 9 * Cin); y = torch.empty((B * Hs * Hs, 256), device="cuda", dtype=torch.bfloat16)
    a = base(); a.A, a.B, a.C = xx.data_ptr(), w.data_ptr(), y.data_ptr(); a.ldb, a.ldc = 9 * Cin, 256
    a.M, a.N, a.K = B * Hs * Hs, 256, 9 * Cin; a.H = a.W = Hs; a.Cin = Cin; a.out_bf16 = 1
    run("conv fwd %dx%d Cin%d" % (Hs, Hs, Cin), a, 1, 2, 0, 2.0 * B * Hs * Hs * 256 * 9 * Cin)
    dyc = mk(B * Hs * Hs, 256); sk = max(1, 256 // (2 * (9 * Cin // 128)))   # engine policy: <= 256 workgroups
    part = torch.empty((sk, 256, 9 * Cin), device="cuda")
    a = base(); a.A, a.B, a.partial = dyc.data_ptr(), xx.data_ptr(), part.data_ptr(); a.lda, a.ldc = 256, 9 * Cin
    a.M, a.N, a.K = 256, 9 * Cin, B * Hs * Hs; a.H = a.W = Hs; a.Cin = Cin; a.splitk = sk
    run("conv wgrad %dx%d Cin%d sk%d" % (Hs, Hs, Cin, sk), a, 1, 1, 3, 2.0 * B * Hs * Hs * 256 * 9 * Cin)
