"""s_memtime anatomy of the pipelined attention forward: per-workgroup cycle counters of wave 0.  Needs the stamp build:
bash tools/exp_file.sh flash_attn_fwd abl7 -DCOUNTR_FA_ABL_BUILD=7; run with COUNTR_LIB=tools/_abl/libcountr_abl7.so."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for B in (8, 32, 1):
    N, H, dh = 576, 12, 64
    qkvs = [torch.randn(B, N, 3, H, dh, device="cuda").to(torch.bfloat16) for _ in range(4)]
    out = torch.empty(B, N, H * dh, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B, H, N, device="cuda")
    for i in range(8): _lib.check(L.countr_attn_fwd(qkvs[i % 4].data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, dh, dh ** -0.5, st()))
    torch.cuda.synchronize()
    nwg = B * H * 5
    d = lse.flatten()[: nwg * 8].view(nwg, 8).cpu()
    full = d[d[:, 0] > 0]
    names = ["total", "prologue", "compute", "loadwait+ldsstore", "barrier", "epilogue", "t0", "loop"]
    print("B=%d: %d workgroups; mean cycles (wave 0): " % (B, nwg) + "  ".join("%s %.0f" % (n, full[:, i].mean()) for i, n in enumerate(names) if n != "t0"))
    t0 = d[:, 6]
    print("     start-time spread (cycles, 24-bit): min %.0f max %.0f ; total: min %.0f max %.0f" % (t0.min(), t0.max(), d[:, 0].min(), d[:, 0].max()))
