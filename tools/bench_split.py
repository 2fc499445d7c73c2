"""Encoder-layer GEMM chain (qkv -> proj -> fc1 -> fc2, 12 layers) at B = 8: one stream with M = 4608 per launch vs two streams with
M = 2304 each (independent halves of the batch), to see whether kernels of the two halves fill each other's tails."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from countr_amd import _lib
L = _lib.lib(); _lib.check(L.countr_init(0))
M, D, NL = 4608, 768, 12
def mk(*shape, dt=torch.bfloat16): return (torch.rand(shape, device="cuda") - 0.5).to(dt)
def gemm(A, B, Cc, Mr, N, K, bias, resid, act, obf, row0=0):
    a = _lib.GemmArgs(); a.alpha = 1.0; a.nbatch = 1; a.nb1 = 1; a.splitk = 1
    es = 2 if obf else 4
    a.A, a.B, a.C = A.data_ptr() + row0 * K * 2, B.data_ptr(), Cc.data_ptr() + row0 * N * es
    a.lda = a.ldb = K; a.ldc = N; a.ldres = N; a.M, a.N, a.K = Mr, N, K
    a.out_bf16 = obf; a.bias = bias.data_ptr(); a.act = act
    a.resid = (resid.data_ptr() + row0 * N * 4) if resid is not None else None
    return a
layers = []
for l in range(NL):
    layers.append(dict(n1=mk(M, D), qkv=mk(M, 3 * D), att=mk(M, D), x1=mk(M, D, dt=torch.float32), x2=mk(M, D, dt=torch.float32), n2=mk(M, D),
                       h=mk(M, 4 * D), x3=mk(M, D, dt=torch.float32), wq=mk(3 * D, D), wp=mk(D, D), w1=mk(4 * D, D), w2=mk(D, 4 * D),
                       bq=mk(3 * D, dt=torch.float32), bp=mk(D, dt=torch.float32), b1=mk(4 * D, dt=torch.float32), b2=mk(D, dt=torch.float32)))
def chain(row0, rows):
    ops = []
    for d in layers:
        ops.append(gemm(d["n1"], d["wq"], d["qkv"], rows, 3 * D, D, d["bq"], None, 0, 1, row0))
        ops.append(gemm(d["att"], d["wp"], d["x2"], rows, D, D, d["bp"], d["x1"], 0, 0, row0))
        ops.append(gemm(d["n2"], d["w1"], d["h"], rows, 4 * D, D, d["b1"], None, 1, 1, row0))
        ops.append(gemm(d["h"], d["w2"], d["x3"], rows, D, 4 * D, d["b2"], d["x2"], 0, 0, row0))
    return ops
full = chain(0, M)
halves = [chain(0, M // 2), chain(M // 2, M // 2)]
s0 = torch.cuda.current_stream(); s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(ops, st):
    sp = C.c_void_p(st.cuda_stream)
    for a in ops: L.countr_gemm(C.byref(a), 1, 0, 0, sp)
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s0)
    for _ in range(reps): fn()
    e1.record(s0); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / NL
def one(): run(full, s0)
def two():
    ev = torch.cuda.Event(); ev.record(s0); s1.wait_event(ev); s2.wait_event(ev)
    run(halves[0], s1); run(halves[1], s2)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event(); e1.record(s1); e2.record(s2); s0.wait_event(e1); s0.wait_event(e2)
def seq(): run(halves[0], s0); run(halves[1], s0)
for r in range(2):
    print("one stream, M=4608 per launch   : %6.1f us per layer (4 GEMMs)" % timed(one))
    print("two streams, M=2304 per launch  : %6.1f us per layer" % timed(two))
    print("one stream, the two halves in turn: %6.1f us per layer" % timed(seq), flush=True)
# same under hipGraph capture (no host launch gaps)
def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s0):
        fn()
    return g
for name, fn in (("one", one), ("two", two)):
    try:
        g = capture(fn)
        print("graph %s: %6.1f us per layer" % (name, timed(g.replay)), flush=True)
    except Exception as e:  # noqa: BLE001
        print("graph %s failed: %s" % (name, str(e)[:200]))
