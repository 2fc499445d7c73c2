"""Lean nn.Linear kernel (csrc/linear.hip) against the generic gemm_kernel on the encoder / decoder forward shapes at B = 8:
same inputs, outputs compared with each other and with an fp64 reference, back-to-back timing of both (COUNTR_LEAN read per call)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
FILTER = sys.argv[2] if len(sys.argv) > 2 else ""
def mk(*shape, dt=torch.bfloat16, s=1.0): return ((torch.rand(shape, device="cuda") - 0.5) * s).to(dt)
def timeit(a):
    for _ in range(3): _lib.check(L.countr_gemm(C.byref(a), 1, 0, 0, st()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): L.countr_gemm(C.byref(a), 1, 0, 0, st())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / ITERS
B = int(os.environ.get("B", "8")); M = B * 576
cases = [("qkv", 2304, 768, "bf16"), ("proj", 768, 768, "res"), ("fc1", 3072, 768, "gelu"), ("fc2", 768, 3072, "res"),
         ("dec_embed", 512, 768, "resmod"), ("dec qkv", 1536, 512, "bf16"), ("dec proj", 512, 512, "res"), ("dec wq", 512, 512, "bf16"),
         ("dec fc1", 2048, 512, "gelu2"), ("dec fc2", 512, 2048, "res")]
for name, N, K, epi in cases:
    if FILTER and FILTER not in name: continue
    A_, W_ = mk(M, K, s=2.0), mk(N, K, s=0.2); bias = mk(N, dt=torch.float32)
    obf = epi in ("bf16", "gelu", "gelu2")
    outs = {}
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B = A_.data_ptr(), W_.data_ptr(); a.lda = a.ldb = K; a.ldc = N; a.ldres = N; a.M, a.N, a.K = M, N, K
    a.bias = bias.data_ptr(); a.out_bf16 = int(obf); a.act = 1 if epi.startswith("gelu") else 0
    resid = mk(576 if epi == "resmod" else M, N, dt=torch.float32) if not obf else None
    if resid is not None: a.resid = resid.data_ptr(); a.res_mod = 576 if epi == "resmod" else 0
    ref = A_.double() @ W_.double().t() + bias.double()
    pre_ref = ref
    if a.act: ref = torch.nn.functional.gelu(ref)
    if resid is not None: ref = ref + (resid.double().repeat(M // 576, 1) if epi == "resmod" else resid.double())
    res = {}
    for lean in ("1", "0"):
        os.environ["COUNTR_LEAN"] = lean
        out = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
        pre = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16) if epi == "gelu2" else None
        a.C = out.data_ptr(); a.C2 = pre.data_ptr() if pre is not None else None
        us = timeit(a)
        torch.cuda.synchronize()
        err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
        perr = (pre.double() - pre_ref).abs().max().item() / pre_ref.abs().max().item() if pre is not None else 0.0
        res[lean] = (us, err, perr, out)
    d = (res["1"][3].double() - res["0"][3].double()).abs().max().item() / ref.abs().max().item()
    fl = 2.0 * M * N * K
    print("%-10s %5dx%4dx%4d %-6s lean %6.1f us %6.0f TF/s (err %.1e pre %.1e) | generic %6.1f us %6.0f TF/s (err %.1e) | lean-generic %.1e" % (
        name, M, N, K, epi, res["1"][0], fl / res["1"][0] / 1e6, res["1"][1], res["1"][2], res["0"][0], fl / res["0"][0] / 1e6, res["0"][1], d), flush=True)
