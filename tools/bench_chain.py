"""fc1 -> fc2 chains over 12 'layers': the same buffers every layer (warm) vs distinct buffers per layer (as in the engine) vs
distinct buffers carved from ONE allocation.  Separates cache/TLB state effects from kernel cost."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, D, NL = 4608, 768, 12


def carve(arena, off, shape, dtype):
    n = 1
    for s in shape: n *= s
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    t = arena[off:off + nbytes].view(dtype).view(*shape)
    return t, (off + nbytes + 4095) // 4096 * 4096


PAD = int(os.environ.get("PAD", "64"))      # "padded" mode: pitch of the fc1 output / fc2 operands = 4 D + PAD elements


def make(mode):
    sets = []
    arena, off = (torch.empty(3 << 30, dtype=torch.uint8, device="cuda"), 0) if mode == "arena" else (None, 0)
    for l in range(NL if mode != "shared" else 1):
        d = {}
        KP = 4 * D + (PAD if mode == "padded" else 0)
        for k, shape, dt in (("n2", (M, D), torch.bfloat16), ("hact", (M, KP), torch.bfloat16), ("x2", (M, D), torch.float32),
                             ("x3", (M, D), torch.float32), ("w1", (4 * D, D), torch.bfloat16), ("w2", (D, KP), torch.bfloat16),
                             ("b1", (4 * D,), torch.float32), ("b2", (D,), torch.float32)):
            if arena is not None:
                d[k], off = carve(arena, off, shape, dt)
            else:
                d[k] = torch.empty(shape, dtype=dt, device="cuda")
            d[k].copy_(torch.rand(shape, device="cuda") - 0.5)
        sets.append(d)
    if mode in ("wdistinct", "adistinct"):      # only the weights (wdistinct) / only the activations (adistinct) differ per layer
        keep = ("w1", "w2", "b1", "b2") if mode == "wdistinct" else ("n2", "hact", "x2", "x3")
        sets = [{k: (d[k] if k in keep else sets[0][k]) for k in d} for d in sets]
    return sets * (NL if mode == "shared" else 1), arena


def gemm(A, B, Cc, N, K, bias, resid, act, obf):
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr(); a.lda, a.ldb = A.shape[1], B.shape[1]; a.ldc = Cc.shape[1]; a.ldres = N; a.M, a.N, a.K = M, N, K
    a.out_bf16 = obf; a.bias = bias.data_ptr(); a.act = act; a.resid = resid.data_ptr() if resid is not None else None
    return a


for mode in (sys.argv[1:] or ("shared", "distinct", "arena", "padded")):
    sets, keep = make(mode)
    calls = []
    for d in sets:
        calls.append(gemm(d["n2"], d["w1"], d["hact"], 4 * D, D, d["b1"], None, 1, 1))
        calls.append(gemm(d["hact"], d["w2"], d["x3"], D, 4 * D, d["b2"], d["x2"], 0, 0))
    PF = os.environ.get("PF", "0") == "1"      # touch the NEXT launch's weights (a torch reduction: a serial stand-in for a prefetch kernel) first
    wts = [d[k] for d in sets for k in ("w1", "w2")]
    def run(sel, idx):
        for a, i in zip(sel, idx):
            if PF: wts[(i + 1) % len(wts)].view(torch.int32).sum()
            L.countr_gemm(C.byref(a), 1, 0, 0, st())
    allidx = list(range(len(calls)))
    for which, sel, idx in (("fc1", calls[0::2], allidx[0::2]), ("fc2", calls[1::2], allidx[1::2]), ("fc1+fc2", calls, allidx)):
        for _ in range(2): run(sel, idx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 5
        e0.record()
        for _ in range(R): run(sel, idx)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / R / NL
        if PF:        # cost of the stand-in itself
            e0.record()
            for _ in range(R):
                for i in idx: wts[(i + 1) % len(wts)].view(torch.int32).sum()
            e1.record(); torch.cuda.synchronize()
            print("%-9s %-8s %6.1f us per layer incl. %.1f us of weight touching (alone: warm)" % (mode, which, t, e0.elapsed_time(e1) * 1e3 / R / NL), flush=True)
        else:
            print("%-9s %-8s %6.1f us per layer" % (mode, which, t), flush=True)
    del sets, keep, calls
    torch.cuda.empty_cache()
