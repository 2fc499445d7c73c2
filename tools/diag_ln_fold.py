"""How large is |mean| / sigma of the rows that the folded LayerNorms see?  (The fold rounds the RAW residual-stream row to bf16, so the
operand's rounding error relative to sigma grows by sqrt(1 + (mean / sigma)^2) -- tests/test_gemm_gpu.py::test_lean_linear_layernorm_folding.)
CPU, oracle forward of the encoder on the deterministic test weights: per block, the distribution of |mean| / sigma over tokens at the
inputs of norm1 and norm2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import countr_ref as R, weights as W
torch.set_num_threads(min(os.cpu_count(), 16))
name = "mae_vit_base_patch16"
sd = W.make_state_dict(name, seed=0)
imgs, _, _, _ = W.make_inputs(batch=2, shots=3, seed=5)
p = R.Params(sd)
cfg = W.CONFIGS[name]
x = R.patch_embed(torch.from_numpy(imgs), p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg[0]) + p["pos_embed"]
def ratio(t):
    m, s = t.mean(-1), t.std(-1, unbiased=False)
    r = (m.abs() / s).flatten()
    return "median %.3f  p99 %.3f  max %.3f" % (r.median().item(), r.quantile(0.99).item(), r.max().item())
for i in range(cfg[2]):
    pre = "blocks.%d" % i
    print("block %2d norm1 input |mean|/sigma: %s" % (i, ratio(x)))
    x = x + R.self_attention(R.layer_norm(x, p[pre + ".norm1.weight"], p[pre + ".norm1.bias"]), p, pre + ".attn", cfg[3])
    print("block %2d norm2 input |mean|/sigma: %s" % (i, ratio(x)))
    x = x + R.mlp(R.layer_norm(x, p[pre + ".norm2.weight"], p[pre + ".norm2.bias"]), p, pre + ".mlp")
