"""Micro-benchmark of countr_attn_fwd at the encoder / decoder shapes (bf16).  Prints us and TF/s; checks vs fp64."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, N, H, dh) in ((8, 576, 12, 64), (8, 576, 16, 32), (32, 576, 12, 64)):
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3, H, dh, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, N, H * dh, device="cuda", dtype=torch.bfloat16)
    call = lambda: L.countr_attn_fwd(qkv.data_ptr(), out.data_ptr(), None, B, N, H, dh, dh ** -0.5, st())
    for _ in range(5): _lib.check(call())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = 4.0 * N * N * dh * H * B
    q = qkv[:2, :, 0].double().permute(0, 2, 1, 3); k = qkv[:2, :, 1].double().permute(0, 2, 1, 3); v = qkv[:2, :, 2].double().permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, -1) @ v).permute(0, 2, 1, 3).reshape(2, N, H * dh)
    err = ((out[:2].double() - ref).abs().max() / ref.abs().max()).item()
    print("B%d H%d dh%d: %7.1f us  %7.1f TF/s   max rel err %.2e" % (B, H, dh, us, fl / us / 1e6, err), flush=True)
