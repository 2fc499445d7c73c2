"""LayerNorm forward / backward timings at the step's shapes (rows x D): 4608 x 768 (encoder), 4608 x 512 (decoder), 2304 x 768 (MAE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
nb = L.countr_layernorm_bwd_nblocks()
for rows, D in ((4608, 768), (4608, 512), (2304, 768)):
    x = torch.randn(rows, D, device="cuda"); g = torch.rand(D, device="cuda"); b = torch.rand(D, device="cuda")
    y = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16); mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    dy = torch.randn(rows, D, device="cuda").to(torch.bfloat16); dx = torch.zeros(rows, D, device="cuda"); dxb = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(nb * 2 * D, device="cuda")
    def fwd(): L.countr_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D, 1e-6, 1, st())
    def bwd(): L.countr_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), None, None, ws.data_ptr(), rows, D, 1, 1, 0, dxb.data_ptr(), st())
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print("layernorm %s %d x %d (blocks %d): %.1f us" % (name, rows, D, nb, e0.elapsed_time(e1) * 1e3 / 50), flush=True)
