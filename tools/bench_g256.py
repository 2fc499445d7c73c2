"""256 x 256 8-phase kernel (csrc/gemm256.hip) against the 128-row forms of linear.hip on the shapes it is meant for: the 3x3 convolutions
of the density head on the big maps and the encoder's fc1.  Same inputs, outputs checked against fp64 and against each other, interleaved
timing rounds inside one process (COUNTR_G256 is read per call).  usage: bench_g256.py [iters] [filter] [rounds]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
FILTER = sys.argv[2] if len(sys.argv) > 2 else ""
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
def mk(*shape, dt=torch.bfloat16, s=1.0): return ((torch.rand(shape, device="cuda") - 0.5) * s).to(dt)
def timeit(a, ma, mb):
    for _ in range(2): _lib.check(L.countr_gemm(C.byref(a), 1, ma, mb, st()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): L.countr_gemm(C.byref(a), 1, ma, mb, st())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / ITERS
def report(name, fl, res, ref, outs):
    d = (outs["2"].double() - outs["0"].double()).abs().max().item() / ref.abs().max().item()
    errs = {k: (outs[k].double() - ref).abs().max().item() / ref.abs().max().item() for k in outs}
    t2, t0 = sorted(res["2"]), sorted(res["0"])
    print("%-22s g256 %7.1f us (min %7.1f) %6.0f TF/s err %.1e | 128-row %7.1f us (min %7.1f) %6.0f TF/s err %.1e | diff %.1e" % (
        name, t2[len(t2) // 2], t2[0], fl / t2[len(t2) // 2] / 1e6, errs["2"], t0[len(t0) // 2], t0[0], fl / t0[len(t0) // 2] / 1e6, errs["0"], d), flush=True)

B = int(os.environ.get("B", "8"))
# ---- convolutions
for name, Bsz, H, W, Cin, Cout in [("conv 192x192 c256", B, 192, 192, 256, 256), ("conv 96x96 c256", B, 96, 96, 256, 256), ("conv 48x48 c256", 4 * B, 48, 48, 256, 256)]:
    if FILTER and FILTER not in name: continue
    x = mk(Bsz, H, W, Cin, s=2.0); w = mk(Cout, 9 * Cin, s=0.1); bias = mk(Cout, dt=torch.float32)
    M, K = Bsz * H * W, 9 * Cin
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), bias.double(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout).contiguous()
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.bias = x.data_ptr(), w.data_ptr(), bias.data_ptr(); a.ldb, a.ldc = K, Cout; a.M, a.N, a.K = M, Cout, K
    a.H, a.W, a.Cin = H, W, Cin; a.out_bf16 = 1
    outs, res = {}, {"2": [], "0": []}
    for r in range(ROUNDS):
        for mode in ("2", "0"):
            os.environ["COUNTR_G256"] = mode
            out = torch.full((M, Cout), float("nan"), device="cuda", dtype=torch.bfloat16); a.C = out.data_ptr()
            res[mode].append(timeit(a, 2, 0)); outs[mode] = out
    report(name, 2.0 * M * Cout * K, res, ref, outs)
    del ref, x
# ---- nn.Linear forward shapes (M = B * 576)
M = B * 576
for name, N, K, epi in [("fc1 gelu", 3072, 768, "gelu"), ("fc1 gelu+pre", 3072, 768, "gelu2"), ("qkv", 2304, 768, "bf16"), ("dec fc1", 2048, 512, "gelu2"), ("big 8192x4096x4096", 4096, 4096, "bf16")]:
    if FILTER and FILTER not in name: continue
    Mx = 8192 if name.startswith("big") else M
    A_, W_ = mk(Mx, K, s=2.0), mk(N, K, s=0.2); bias = mk(N, dt=torch.float32)
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B = A_.data_ptr(), W_.data_ptr(); a.lda = a.ldb = K; a.ldc = N; a.M, a.N, a.K = Mx, N, K
    a.bias = bias.data_ptr(); a.out_bf16 = 1; a.act = 1 if epi.startswith("gelu") else 0
    ref = A_.double() @ W_.double().t() + bias.double()
    pre_ref = ref
    if a.act: ref = torch.nn.functional.gelu(ref)
    outs, res = {}, {"2": [], "0": []}
    for r in range(ROUNDS):
        for mode in ("2", "0"):
            os.environ["COUNTR_G256"] = mode
            out = torch.full((Mx, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            pre = torch.full((Mx, N), float("nan"), device="cuda", dtype=torch.bfloat16) if epi == "gelu2" else None
            a.C = out.data_ptr(); a.C2 = pre.data_ptr() if pre is not None else None
            res[mode].append(timeit(a, 0, 0)); outs[mode] = out
            if pre is not None:
                assert (pre.double() - pre_ref).abs().max().item() <= 5e-3 * pre_ref.abs().max().item(), "pre-activation copy"
    report(name, 2.0 * Mx * N * K, res, ref, outs)
