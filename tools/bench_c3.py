"""Times countr_conv3x3_c3_fwd (first exemplar conv, 3 -> 64 channels, 24 boxes of 64x64) from a captured graph of 20 launches,
so the Python launch rate does not bound the measurement; COUNTR_C3_BLOCKS sweeps the grid cap."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
S=24
x=torch.rand(S,3,64,64,device="cuda"); w=torch.rand(64,3,3,3,device="cuda")-0.5; b=torch.rand(64,device="cuda")
out=torch.empty(S,64,64,64,device="cuda",dtype=torch.bfloat16)
call=lambda: L.countr_conv3x3_c3_fwd(x.data_ptr(),w.data_ptr(),b.data_ptr(),out.data_ptr(),S,64,64,1,st())
for _ in range(5): _lib.check(call())
g=torch.cuda.CUDAGraph()
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): call()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20): call()
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
print(os.environ.get("COUNTR_C3_BLOCKS"), "%.1f us"%(e0.elapsed_time(e1)*1e3/200))
