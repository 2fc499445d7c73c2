"""Vendor-library yardstick (measurement tool, not product code): the step's GEMM / attention / convolution shapes on
hipBLASLt / rocBLAS (torch.matmul, F.linear), torch's fused SDPA (AOTriton / CK on ROCm) and MIOpen (F.conv2d, channels-last) beside
this library's own launches through the C ABI -- same box, same process, same rotating-buffer loop, HIP events on the launch stream.

Answers one question the roofline fractions cannot: how far is each kernel from what the vendor's tuned libraries reach on the SAME
shape on the SAME chip (power-limited clock, 216-tile launches on 256 CUs and all)?

python tools/yardstick_vendor.py [iters]   ->  one line per shape
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from countr_amd import _lib

L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
NBUF = 4


def timed(fn, iters=ITERS, reps=5):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = []
    for _ in range(reps):
        e0.record()
        for i in range(iters): fn(i)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(res)[len(res) // 2]


def mk(*shape, dt=torch.bfloat16, s=1.0): return ((torch.rand(shape, device="cuda") - 0.5) * s).to(dt)


def gemms(B):
    M = B * 576
    cases = [("enc qkv", 2304, 768, "bf16"), ("enc proj", 768, 768, "res"), ("enc fc1", 3072, 768, "gelu"), ("enc fc2", 768, 3072, "res"),
             ("dec qkv", 1536, 512, "bf16"), ("dec proj", 512, 512, "res"), ("dec fc1", 2048, 512, "gelu"), ("dec fc2", 512, 2048, "res")]
    for name, N, K, epi in cases:
        As = [mk(M, K, s=2.0) for _ in range(NBUF)]; W = mk(N, K, s=0.2); bias = mk(N, dt=torch.float32); bias16 = bias.to(torch.bfloat16)
        obf = epi in ("bf16", "gelu")
        out = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16 if obf else torch.float32)
        resid = mk(M, N, dt=torch.float32) if not obf else None
        args = []
        for A_ in As:
            a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
            a.A, a.B = A_.data_ptr(), W.data_ptr(); a.lda = a.ldb = K; a.ldc = N; a.ldres = N; a.M, a.N, a.K = M, N, K
            a.bias = bias.data_ptr(); a.out_bf16 = int(obf); a.act = 1 if epi == "gelu" else 0; a.C = out.data_ptr()
            if resid is not None: a.resid = resid.data_ptr()
            args.append(a)
        _lib.check(L.countr_gemm(C.byref(args[0]), 1, 0, 0, st()))
        ours = timed(lambda i: L.countr_gemm(C.byref(args[i % NBUF]), 1, 0, 0, st()))
        Wt = W.t().contiguous()
        o2 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        plain = timed(lambda i: torch.matmul(As[i % NBUF], W.t(), out=o2))
        plain_nn = timed(lambda i: torch.matmul(As[i % NBUF], Wt, out=o2))
        lin = timed(lambda i: F.linear(As[i % NBUF], W, bias16))
        if epi == "gelu": full = timed(lambda i: F.gelu(F.linear(As[i % NBUF], W, bias16)))
        elif epi == "res": full = timed(lambda i: torch.add(resid, F.linear(As[i % NBUF], W, bias16)))
        else: full = lin
        fl = 2.0 * M * N * K
        print("B%-2d %-9s %5dx%4dx%4d %-5s ours(fused epilogue) %6.1f us %5.0f TF/s | hipBLASLt matmul NT %6.1f us %5.0f TF/s, NN %6.1f us, +bias %6.1f us, "
              "+bias+%s (torch ops) %6.1f us" % (B, name, M, N, K, epi, ours, fl / ours / 1e6, plain, fl / plain / 1e6, plain_nn, lin, epi, full), flush=True)


def attention():
    for (B, N, H, dh) in ((8, 576, 12, 64), (32, 576, 12, 64), (8, 576, 16, 32)):
        qkvs = [torch.randn(B, N, 3, H, dh, device="cuda").to(torch.bfloat16) for _ in range(NBUF)]
        out = torch.empty(B, N, H * dh, device="cuda", dtype=torch.bfloat16)
        ours = timed(lambda i: L.countr_attn_fwd(qkvs[i % NBUF].data_ptr(), out.data_ptr(), None, B, N, H, dh, dh ** -0.5, st()))
        # SDPA wants [B, H, N, dh]; give it contiguous tensors in its preferred layout (no transposes in the timed region)
        qs = [[x[:, :, j].permute(0, 2, 1, 3).contiguous() for j in range(3)] for x in qkvs]
        res = {}
        from torch.nn.attention import SDPBackend, sdpa_kernel
        for nm, be in (("flash", SDPBackend.FLASH_ATTENTION), ("mem_eff", SDPBackend.EFFICIENT_ATTENTION)):
            try:
                with sdpa_kernel(be):
                    res[nm] = timed(lambda i: F.scaled_dot_product_attention(*qs[i % NBUF]))
            except Exception as e:
                res[nm] = float("nan"); print("  sdpa %s: %s" % (nm, str(e)[:120]))
        fl = 4.0 * N * N * dh * H * B
        print("attention B%-2d N%d H%d dh%d: ours %6.1f us %5.0f TF/s | torch SDPA flash %6.1f us %5.0f TF/s, mem_eff %6.1f us" % (
            B, N, H, dh, ours, fl / ours / 1e6, res["flash"], fl / res["flash"] / 1e6, res["mem_eff"]), flush=True)


def convs():
    B = 8
    for (HW, Cin, Cout) in ((192, 256, 256), (96, 256, 256), (48, 256, 256), (24, 512, 256)):
        xs = [mk(B, Cin, HW, HW).contiguous(memory_format=torch.channels_last) for _ in range(2)]
        w = mk(Cout, Cin, 3, 3, s=0.1).contiguous(memory_format=torch.channels_last)
        b = mk(Cout)
        fl = 2.0 * B * HW * HW * Cin * Cout * 9
        try:
            torch.backends.cudnn.benchmark = True
            fwd = timed(lambda i: F.conv2d(xs[i % 2], w, b, padding=1), iters=10, reps=3)
            x = xs[0].clone().requires_grad_(True); wp = w.clone().requires_grad_(True)
            y = F.conv2d(x, wp, None, padding=1); dy = torch.randn_like(y)
            dg = timed(lambda i: torch.autograd.grad(y, x, dy, retain_graph=True), iters=10, reps=3)
            wg = timed(lambda i: torch.autograd.grad(y, wp, dy, retain_graph=True), iters=10, reps=3)
            print("conv3x3 B8 %3dx%-3d %d->%d (%5.1f GF): MIOpen fwd %7.1f us %5.0f TF/s | dgrad %7.1f us %5.0f TF/s | wgrad %7.1f us %5.0f TF/s" % (
                HW, HW, Cin, Cout, fl / 1e9, fwd, fl / fwd / 1e6, dg, fl / dg / 1e6, wg, fl / wg / 1e6), flush=True)
        except Exception as e:
            print("conv %d: %s" % (HW, str(e)[:200]), flush=True)


if __name__ == "__main__":
    which = os.environ.get("YARD", "gemm,attn,conv").split(",")
    print("torch %s, hip %s, device %s" % (torch.__version__, torch.version.hip, torch.cuda.get_device_name(0)), flush=True)
    if "gemm" in which:
        gemms(8); gemms(32)
    if "attn" in which: attention()
    if "conv" in which: convs()
