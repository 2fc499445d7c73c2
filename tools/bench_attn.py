"""Micro-benchmark of countr_attn_fwd at the encoder / decoder shapes (bf16).  Prints us and TF/s; checks vs fp64.

python tools/bench_attn.py            # (COUNTR_LIB=<variant .so> times an experimental build: bash tools/exp_file.sh flash_attn_fwd <tag> ...)
The timed loop rotates over several qkv buffers (each launch reads a buffer that was not touched by the previous launch) so the
figure is closer to what the kernel sees inside a step than 50 launches on one cache-hot tensor.
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    from countr_amd import _lib
    L = _lib.lib(); _lib.check(L.countr_init(0))
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = ((8, 576, 12, 64), (8, 576, 16, 32), (32, 576, 12, 64), (8, 288, 12, 64), (1, 576, 12, 64))
    if os.environ.get("BENCH_ATTN_SHAPES"):   # e.g. "8,576,12,64;32,576,12,64"
        shapes = tuple(tuple(int(x) for x in sh.split(",")) for sh in os.environ["BENCH_ATTN_SHAPES"].split(";"))
    for (B, N, H, dh) in shapes:
        torch.manual_seed(0)
        nbuf = 6
        qkvs = [torch.randn(B, N, 3, H, dh, device="cuda").to(torch.bfloat16) for _ in range(nbuf)]
        out = torch.empty(B, N, H * dh, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, N, device="cuda")
        call = lambda i: L.countr_attn_fwd(qkvs[i % nbuf].data_ptr(), out.data_ptr(), None, B, N, H, dh, dh ** -0.5, st())
        for i in range(6): _lib.check(call(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res = []
        for rep in range(5):
            e0.record()
            for i in range(60): call(i)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 60)
        us = sorted(res)[len(res) // 2]
        fl = 4.0 * N * N * dh * H * B
        qkv = qkvs[0]
        _lib.check(L.countr_attn_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, H, dh, dh ** -0.5, st()))
        nb = min(B, 2)
        q = qkv[:nb, :, 0].double().permute(0, 2, 1, 3); k = qkv[:nb, :, 1].double().permute(0, 2, 1, 3); v = qkv[:nb, :, 2].double().permute(0, 2, 1, 3)
        sc = q @ k.transpose(-1, -2) * dh ** -0.5
        ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(nb, N, H * dh)
        err = ((out[:nb].double() - ref).abs().max() / ref.abs().max()).item()
        lerr = (lse[:nb].double() - torch.logsumexp(sc, -1)).abs().max().item()
        print("B%-2d N%d H%d dh%d: %7.1f us (min %.1f)  %7.1f TF/s   max rel err %.2e  lse err %.1e" % (B, N, H, dh, us, min(res), fl / us / 1e6, err, lerr), flush=True)


if __name__ == "__main__":
    one()
