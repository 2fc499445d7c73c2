"""Times countr_groupnorm_relu_bwd / _fwd and the InstanceNorm+pool pair at the density-head / exemplar shapes (bf16)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: t.data_ptr() if t is not None else None
def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
B, Cc = 8, 256
for HW in (24 * 24, 48 * 48, 96 * 96, 192 * 192):
    x = torch.randn(B, HW, Cc, device="cuda").bfloat16(); dy = torch.randn(B, HW, Cc, device="cuda").bfloat16()
    g = torch.ones(Cc, device="cuda"); b = torch.zeros(Cc, device="cuda")
    y = torch.empty_like(x); dx = torch.empty_like(x); stats = torch.empty(B, 8, 2, device="cuda")
    ns = L.countr_groupnorm_nsplit(HW); ws = torch.empty(B * ns * 3 * Cc + 64 + 16 * B, device="cuda")
    dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    f = lambda: L.countr_groupnorm_relu_fwd(P(x), P(g), P(b), P(y), None, None, None, P(stats), P(ws), B, HW, Cc, 8, 1e-5, 1, st())
    bw = lambda: L.countr_groupnorm_relu_bwd(P(x), P(dy), None, None, P(stats), P(g), P(b), P(dx), P(dg), P(db), None, None, P(ws), B, HW, Cc, 8, 1, 0, st())
    print("GN %5d px: fwd %6.1f us   bwd %6.1f us   (%.0f MB tensor)" % (HW, timeit(f), timeit(bw), x.numel() * 2 / 1e6), flush=True)
S = 24
for H, Cc, avg in ((64, 64, 0), (32, 128, 0), (16, 256, 0), (8, 512, 1)):
    x = torch.randn(S, H, H, Cc, device="cuda").bfloat16()
    y = torch.empty(S, H // 2, H // 2, Cc, device="cuda", dtype=torch.bfloat16) if not avg else torch.empty(S, Cc, device="cuda", dtype=torch.bfloat16)
    dyp = torch.randn_like(y); dx = torch.empty_like(x); stats = torch.empty(S, Cc, 2, device="cuda")
    iws = torch.empty(L.countr_instnorm_workspace_floats(S, Cc), device="cuda")
    f = lambda: L.countr_instnorm_relu_pool_fwd(P(x), P(y), P(stats), S, H, H, Cc, avg, 1e-5, 1, P(iws), None, 0, st())
    bw = lambda: L.countr_instnorm_relu_pool_bwd(P(x), P(dyp), P(stats), P(dx), S, H, H, Cc, avg, 1, P(iws), 0, st())
    print("IN %2dx%2d C%3d: fwd %6.1f us   bwd %6.1f us" % (H, H, Cc, timeit(f), timeit(bw)), flush=True)
# first exemplar conv (3 -> 64, direct) forward / wgrad
S, H = 24, 64
x = torch.rand(S, 3, H, H, device="cuda"); w = torch.randn(64, 3, 3, 3, device="cuda") * 0.1; b = torch.zeros(64, device="cuda")
y = torch.empty(S, H, H, 64, device="cuda", dtype=torch.bfloat16); dy = torch.randn(S, H, H, 64, device="cuda").bfloat16()
dw = torch.empty(64, 27, device="cuda"); dbias = torch.empty(64, device="cuda")
ws = torch.empty(L.countr_conv3x3_c3_wgrad_nblocks() * 64 * 28, device="cuda")
f = lambda: L.countr_conv3x3_c3_fwd(P(x), P(w), P(b), P(y), S, H, H, 1, st())
bw = lambda: L.countr_conv3x3_c3_wgrad(P(x), P(dy), P(dw), P(dbias), P(ws), S, H, H, 1, 0, st())
print("conv 3->64 64x64 x24: fwd %6.1f us   wgrad %6.1f us" % (timeit(f), timeit(bw)), flush=True)
