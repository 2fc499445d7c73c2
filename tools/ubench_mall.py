"""Where does a consumer's input come from?  The bilinear adjoint (a read-dominated streaming kernel: 4 bytes read per byte written) timed
on a map that (a) it has just read itself (memory-side cache warm), (b) another kernel has just WRITTEN (the in-step situation: the
convolution's dgrad writes the 151-MB map the adjoint then reads), (c) lies behind 600 MB of unrelated traffic (cold), at the step's
size (8 images, 151 MB) and at half of it (4 images, 75 MB) -- does a producer -> consumer pair of half-batch launches stay inside the
256-MB memory-side cache?   python tools/ubench_mall.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
H, Cc = 96, 256
big = torch.empty(300 << 20, device="cuda", dtype=torch.uint8); big2 = torch.empty_like(big)
def run(B, prep, n=12):
    dy = torch.randn(B, 2 * H, 2 * H, Cc, device="cuda").bfloat16(); src = torch.randn_like(dy); dx = torch.empty(B, H, H, Cc, device="cuda", dtype=torch.bfloat16)
    ts = []
    for _ in range(n):
        prep(dy, src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); L.countr_upsample2x_bwd(dy.data_ptr(), dx.data_ptr(), B, H, H, Cc, 1, st()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
def warm(dy, src): L.countr_upsample2x_bwd(dy.data_ptr(), torch.empty(dy.shape[0], H, H, Cc, device="cuda", dtype=torch.bfloat16).data_ptr(), dy.shape[0], H, H, Cc, 1, st())
def written(dy, src): dy.copy_(src)
def written_after_flush(dy, src): big2.copy_(big); dy.copy_(src)
def cold(dy, src): big2.copy_(big); big.copy_(big2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize(); print("empty event pair: %.1f us" % (e0.elapsed_time(e1) * 1e3))
for B in (8, 4, 2):
    print("B = %d (%3d MB read): just read %6.1f us | just written by a copy %6.1f us | written behind 600 MB of other traffic %6.1f us | cold %6.1f us" % (
        B, B * 4 * H * H * Cc * 2 >> 20, run(B, warm), run(B, written), run(B, written_after_flush), run(B, cold)), flush=True)
