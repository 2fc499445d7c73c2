"""s_memtime anatomy of the 2-stage GEMM loop (library built with -DCOUNTR_GEMM_STAMP -DCOUNTR_GEMM_EXP): mean cycles per k-tile
and wave spent in DMA issue / fragment reads + MFMAs / waiting for the next tile's loads / the barrier."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def mk(*shape): return (torch.rand(shape, device="cuda") - 0.5).to(torch.bfloat16)
M = 4608
for name, N, K in (("qkv", 2304, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    A_, B_ = mk(M, K), mk(N, K); Cc = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.C = A_.data_ptr(), B_.data_ptr(), Cc.data_ptr(); a.lda = a.ldb = K; a.ldc = N; a.M, a.N, a.K = M, N, K; a.out_bf16 = 1
    dbg = torch.zeros(4096 * 8 * 8, device="cuda")
    a.sC1 = dbg.data_ptr()
    for _ in range(5): _lib.check(L.countr_gemm(C.byref(a), 1, 0, 0, st()))
    torch.cuda.synchronize()
    d = dbg.view(-1, 8).cpu(); d = d[d[:, 0] > 0]
    nt = d[:, 5].mean().item()
    if (d[:, 6] > 0).any():   # wave-specialised kernel: loaders (6 == 1) and compute waves (6 == 2) report different columns
        ld, cp = d[d[:, 6] == 1], d[d[:, 6] == 2]
        print("%s [wave-specialised]: k-tiles %.0f | loaders (%d): loop %.0f, per k-tile: load wait %.0f  barrier %.0f  dma issue %.0f | compute (%d): loop %.0f, per k-tile: barrier %.0f  frags+mfma %.0f"
              % (name, nt, ld.shape[0], ld[:, 0].mean(), ld[:, 1].mean() / nt, ld[:, 2].mean() / nt, ld[:, 3].mean() / nt,
                 cp.shape[0], cp[:, 0].mean(), cp[:, 2].mean() / nt, cp[:, 4].mean() / nt))
        continue
    print("%s: %d waves, k-tiles %.0f; mean cycles per wave: loop total %.0f | per k-tile: dma issue %.0f  frags+mfma %.0f  load wait %.0f  barrier %.0f"
          % (name, d.shape[0], nt, d[:, 0].mean(), d[:, 1].mean() / nt, d[:, 2].mean() / nt, d[:, 3].mean() / nt, d[:, 4].mean() / nt))
    w = d.view(-1, 8, 8) if d.shape[0] % 8 == 0 else None

# 3x3 convolution, 192x192, Cin = Cout = 256, B = 8: forward (IM2ROW x ROW, 128x256 tile, 8 + 4 waves) and wgrad (COL x IM2COL,
# 128x128 wave-specialised, split-K 7)
def report(name, dbg):
    d = dbg.view(-1, 8).cpu(); d = d[d[:, 0] > 0]
    nt = d[:, 5].mean().item()
    ld, cp = d[d[:, 6] == 1], d[d[:, 6] == 2]
    print("%s [wave-specialised]: k-tiles %.0f | loaders (%d): loop %.0f, per k-tile: load wait %.0f  barrier %.0f  dma issue %.0f | compute (%d): loop %.0f, per k-tile: barrier %.0f  frags+mfma %.0f"
          % (name, nt, ld.shape[0], ld[:, 0].mean(), ld[:, 1].mean() / nt, ld[:, 2].mean() / nt, ld[:, 3].mean() / nt,
             cp.shape[0], cp[:, 0].mean(), cp[:, 2].mean() / nt, cp[:, 4].mean() / nt))
B, Hs, Cin = 8, 192, 256
xx = mk(B, Hs, Hs, Cin); w = mk(256, 9 * Cin); y = torch.empty((B * Hs * Hs, 256), device="cuda", dtype=torch.bfloat16)
a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
a.A, a.B, a.C = xx.data_ptr(), w.data_ptr(), y.data_ptr(); a.ldb, a.ldc = 9 * Cin, 256
a.M, a.N, a.K = B * Hs * Hs, 256, 9 * Cin; a.H = a.W = Hs; a.Cin = Cin; a.out_bf16 = 1
dbg = torch.zeros(4096 * 12 * 8, device="cuda"); a.sC1 = dbg.data_ptr()
for _ in range(3): _lib.check(L.countr_gemm(C.byref(a), 1, 2, 0, st()))
torch.cuda.synchronize(); report("conv fwd 192", dbg)
dyc = mk(B * Hs * Hs, 256); sk = 7
part = torch.empty((sk, 256, 9 * Cin), device="cuda")
a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1
a.A, a.B, a.partial = dyc.data_ptr(), xx.data_ptr(), part.data_ptr(); a.lda, a.ldc = 256, 9 * Cin
a.M, a.N, a.K = 256, 9 * Cin, B * Hs * Hs; a.H = a.W = Hs; a.Cin = Cin; a.splitk = sk
dbg = torch.zeros(4096 * 12 * 8, device="cuda"); a.sC1 = dbg.data_ptr()
for _ in range(3): _lib.check(L.countr_gemm(C.byref(a), 1, 1, 3, st()))
torch.cuda.synchronize(); report("conv wgrad 192 sk7", dbg)
