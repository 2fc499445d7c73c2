"""s_memtime anatomy of the 256 x 256 8-phase kernel (build: bash tools/exp_file.sh gemm256 g256stamp -DG256_STAMP; run with
COUNTR_LIB=tools/_abl/libcountr_g256stamp.so COUNTR_G256=2).  Four stamps per phase of two steady-state k-tiles (kept in SGPRs, recorded a
phase later: the instrumentation adds four s_memtime issues per phase and nothing else), means over all workgroups per phase kind and
M half: M part (fragment reads + DMA issue + vmcnt(8)), wait at the mid barrier, C part (lgkmcnt(0) + 8 MFMAs = 256 matrix cycles), wait
at the end barrier; whole-loop cycles per k-tile and the shader clock (s_memtime / s_memrealtime)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def mk(*shape, dt=torch.bfloat16, s=1.0): return ((torch.rand(shape, device="cuda") - 0.5) * s).to(dt)
def show(name, dbg, nwg):
    d = dbg.view(torch.int32)[: nwg * 8 * 64].view(nwg * 8, 64).cpu().to(torch.int64) & 0xffffffff
    nt = d[:, 34].double()
    ok = nt > 0
    print("%s: %d workgroups, loop %.0f cycles per k-tile (2048 matrix cycles), clock %.0f MHz" % (
        name, nwg, (d[ok, 32].double() / nt[ok]).mean().item(), (d[ok, 32].double() / (d[ok, 33].double() * 1e-2)).mean().item()), flush=True)
    NPH = int(os.environ.get("G256_PH", "2"))       # phases per k-tile of the build
    s = d[:, :32].view(-1, 8, 4)[:, : 2 * NPH]
    w = torch.arange(d.shape[0]) % 8
    for half, sel in (("first M half ", (w < 4) & ok), ("second M half", (w >= 4) & ok)):
        t = s[sel]
        dd = lambda a, b: ((a - b) & 0xffffffff).double().mean().item()
        tot = [0.0] * 4
        for ph in range(NPH):
            vals = [0.0] * 4
            for buf in range(2):
                P = buf * NPH + ph
                vals[0] += dd(t[:, P, 1], t[:, P, 0]) / 2; vals[1] += dd(t[:, P, 2], t[:, P, 1]) / 2; vals[2] += dd(t[:, P, 3], t[:, P, 2]) / 2
                if P < 2 * NPH - 1: vals[3] += dd(t[:, P + 1, 0], t[:, P, 3]) / (1 if ph == NPH - 1 else 2)
            for i in range(4): tot[i] += vals[i]
            print("   %s phase %d: M %4.0f | mid barrier %4.0f | C %4.0f | end barrier %4.0f  = %4.0f" % (half, ph + 1, vals[0], vals[1], vals[2], vals[3], sum(vals)), flush=True)
        print("   %s k-tile : M %4.0f | mid barrier %4.0f | C %4.0f | end barrier %4.0f  = %4.0f" % (half, tot[0], tot[1], tot[2], tot[3], sum(tot)), flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else ""
for Hs, Cin, B in ((192, 256, 8), (96, 256, 8)):
    if which and which not in "conv%d" % Hs: continue
    xx = mk(B, Hs, Hs, Cin); w = mk(256, 9 * Cin, s=0.2); y = torch.empty((B * Hs * Hs, 256), device="cuda", dtype=torch.bfloat16)
    bias = mk(256, dt=torch.float32)
    dbg = torch.zeros(1 << 22, device="cuda")
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.C, a.C2 = xx.data_ptr(), w.data_ptr(), y.data_ptr(), dbg.data_ptr(); a.ldb, a.ldc = 9 * Cin, 256; a.bias = bias.data_ptr()
    a.M, a.N, a.K = B * Hs * Hs, 256, 9 * Cin; a.H = a.W = Hs; a.Cin = Cin; a.out_bf16 = 1
    for _ in range(3): _lib.check(L.countr_gemm(C.byref(a), 1, 2, 0, st()))
    torch.cuda.synchronize()
    show("conv %dx%d Cin %d" % (Hs, Hs, Cin), dbg, B * Hs * Hs // 256)
for name, M, N, K in (("fc1", 4608, 3072, 768), ("gemm 8192x4096x4096", 8192, 4096, 4096)):
    if which and which not in name: continue
    A_, W_ = mk(M, K, s=2.0), mk(N, K, s=0.2); bias = mk(N, dt=torch.float32)
    out = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
    dbg = torch.zeros(1 << 22, device="cuda")
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.C, a.C2 = A_.data_ptr(), W_.data_ptr(), out.data_ptr(), dbg.data_ptr()
    a.lda = a.ldb = K; a.ldc = N; a.M, a.N, a.K = M, N, K; a.bias = bias.data_ptr(); a.out_bf16 = 1
    for _ in range(3): _lib.check(L.countr_gemm(C.byref(a), 1, 0, 0, st()))
    torch.cuda.synchronize()
    show(name, dbg, (M // 256) * (N // 256))
