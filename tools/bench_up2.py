"""Times countr_upsample2x_fwd / _bwd at the density-head shapes (bf16, 256 channels) from captured graphs of 10 launches."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def graph_time(call, n=10, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): _lib.check(call())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): call()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)
B, Cc = 8, 256
for H in (24, 48, 96):
    x = torch.randn(B, H, H, Cc, device="cuda").bfloat16(); y = torch.empty(B, 2 * H, 2 * H, Cc, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    f = lambda: L.countr_upsample2x_fwd(x.data_ptr(), y.data_ptr(), B, H, H, Cc, 1, st())
    b = lambda: L.countr_upsample2x_bwd(dy.data_ptr(), dx.data_ptr(), B, H, H, Cc, 1, st())
    print("up2 %3d -> %3d: fwd %6.1f us   bwd %6.1f us   (%.0f MB out)" % (H, 2 * H, graph_time(f), graph_time(b), y.numel() * 2 / 1e6), flush=True)
