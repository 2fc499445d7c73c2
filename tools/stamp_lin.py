"""s_memtime anatomy of the lean linear kernel's wave-specialised loop (build: bash tools/exp_file.sh linear stamp -DLIN_STAMP; run with
COUNTR_LIB=tools/_abl/libcountr_stamp.so).  Per k-tile means over all workgroups: loaders {load wait, barrier, DMA issue}, compute waves
{barrier, everything else}, plus the shader clock during the loop (s_memtime / s_memrealtime)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def mk(*shape, dt=torch.bfloat16, s=1.0): return ((torch.rand(shape, device="cuda") - 0.5) * s).to(dt)
M = 4608
for name, N, K, res in (("proj", 768, 768, True), ("fc2", 768, 3072, True), ("dec fc2", 512, 2048, True), ("dec wq", 512, 512, False),
                        ("qkv", 2304, 768, False), ("fc1 (no gelu)", 3072, 768, False)):
    A_, W_ = mk(M, K, s=2.0), mk(N, K, s=0.2); bias = mk(N, dt=torch.float32)
    out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if res else torch.bfloat16)
    resid = mk(M, N, dt=torch.float32)
    dbg = torch.zeros(1 << 20, device="cuda")
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.C, a.C2 = A_.data_ptr(), W_.data_ptr(), out.data_ptr(), dbg.data_ptr()
    a.lda = a.ldb = K; a.ldc = N; a.ldres = N; a.M, a.N, a.K = M, N, K; a.bias = bias.data_ptr(); a.out_bf16 = int(not res)
    if res: a.resid = resid.data_ptr()
    for _ in range(5): _lib.check(L.countr_gemm(C.byref(a), 1, 0, 0, st()))
    torch.cuda.synchronize()
    d = dbg.view(-1, 8).cpu()
    d = d[d[:, 6] > 0]
    nt = K // 64
    ld, cp = d[d[:, 6] == 1], d[d[:, 6] == 2]
    f = lambda t, i: t[:, i].mean().item() / nt
    mhz = (d[:, 0] / (d[:, 7] * 1e-2)).mean().item()
    print("%-8s %dx%dx%d k-tiles %d waves %d+%d | loaders: loop %.0f/k-tile = load wait %.0f + barrier %.0f + DMA issue %.0f | compute: loop %.0f/k-tile, barrier %.0f | shader clock %.0f MHz"
          % (name, M, N, K, nt, len(cp), len(ld), f(ld, 0), f(ld, 1), f(ld, 2), f(ld, 3), f(cp, 0), f(cp, 2), mhz), flush=True)

# 3x3 convolution forward (im2row A operand), 128x256 tiles
for Hs, Cin in ((192, 256), (96, 256)):
    B = 8
    xx = mk(B, Hs, Hs, Cin); w = mk(256, 9 * Cin, s=0.2); y = torch.empty((B * Hs * Hs, 256), device="cuda", dtype=torch.bfloat16)
    bias = mk(256, dt=torch.float32)
    dbg = torch.zeros(1 << 22, device="cuda")
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.C, a.C2 = xx.data_ptr(), w.data_ptr(), y.data_ptr(), dbg.data_ptr(); a.ldb, a.ldc = 9 * Cin, 256; a.bias = bias.data_ptr()
    a.M, a.N, a.K = B * Hs * Hs, 256, 9 * Cin; a.H = a.W = Hs; a.Cin = Cin; a.out_bf16 = 1
    for _ in range(3): _lib.check(L.countr_gemm(C.byref(a), 1, 2, 0, st()))
    torch.cuda.synchronize()
    d = dbg.view(-1, 8).cpu(); d = d[d[:, 6] > 0]
    nt = 9 * Cin // 64
    ld, cp = d[d[:, 6] == 1], d[d[:, 6] == 2]
    f = lambda t, i: t[:, i].mean().item() / nt
    print("conv %dx%d Cin%d k-tiles %d waves %d+%d | loaders: loop %.0f/k-tile = load wait %.0f + barrier %.0f + DMA issue %.0f | compute: loop %.0f/k-tile, barrier %.0f | shader clock %.0f MHz"
          % (Hs, Hs, Cin, nt, len(cp), len(ld), f(ld, 0), f(ld, 1), f(ld, 2), f(ld, 3), f(cp, 0), f(cp, 2), (d[:, 0] / (d[:, 7] * 1e-2)).mean().item()), flush=True)
