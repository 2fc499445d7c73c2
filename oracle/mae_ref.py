"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the MAE pretraining model of the reference
(/root/reference/models_mae_noct.py:84-204: patchify, random_masking with a given permutation, forward_encoder,
forward_decoder, forward_loss).  Blocks reuse oracle.countr_ref (timm Block == x + attn(LN(x)); x + mlp(LN(x))).
Pinned against the reference by tools/oracle/make_golden_mae.py (tests/golden/mae_*.npz)."""
import torch

from . import countr_ref as R
from .weights import MAE_CONFIGS


def patchify(imgs, p):
    """models_mae_noct.py:84-96: [N,3,H,W] -> [N, L, p*p*3], feature order (py, px, c)."""
    n, _, hh, ww = imgs.shape
    h = w = hh // p
    x = imgs.reshape(n, 3, h, p, w, p)
    x = torch.einsum("nchpwq->nhwpqc", x)
    return x.reshape(n, h * w, p * p * 3)


def block(x, P, b, heads):
    x = x + R.self_attention(R.layer_norm(x, P[b + ".norm1.weight"], P[b + ".norm1.bias"]), P, b + ".attn", heads)
    return x + R.mlp(R.layer_norm(x, P[b + ".norm2.weight"], P[b + ".norm2.bias"]), P, b + ".mlp")


def forward(P, imgs, ids_shuffle, ids_restore, len_keep, model="mae_vit_base_patch16", norm_pix_loss=False):
    """-> (loss, pred [N, L, p*p*3], mask [N, L]) as MaskedAutoencoderViTNoCT.forward (models_mae_noct.py:200-204),
    with the random permutation supplied instead of drawn."""
    p, D, depth, H, Dd, ddepth, Hd = MAE_CONFIGS[model]
    imgs = torch.as_tensor(imgs)
    ids_shuffle = torch.as_tensor(ids_shuffle)
    ids_restore = torch.as_tensor(ids_restore)
    x = R.patch_embed(imgs, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], p) + P["pos_embed"]
    N, L, _ = x.shape
    ids_keep = ids_shuffle[:, :len_keep]
    x = torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    mask = torch.ones(N, L, dtype=x.dtype)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    for i in range(depth):
        x = block(x, P, "blocks.%d" % i, H)
    x = R.layer_norm(x, P["norm.weight"], P["norm.bias"])
    x = R.linear(x, P["decoder_embed.weight"], P["decoder_embed.bias"])
    mask_tokens = P["mask_token"].expand(N, L - len_keep, -1)
    x_ = torch.cat([x, mask_tokens], 1)
    x = torch.gather(x_, 1, ids_restore.unsqueeze(-1).expand(-1, -1, Dd)) + P["decoder_pos_embed"]
    for i in range(ddepth):
        x = block(x, P, "decoder_blocks.%d" % i, Hd)
    x = R.layer_norm(x, P["decoder_norm.weight"], P["decoder_norm.bias"])
    pred = R.linear(x, P["decoder_pred.weight"], P["decoder_pred.bias"])
    target = patchify(imgs, p)
    if norm_pix_loss:
        mean = target.mean(-1, keepdim=True)
        var = target.var(-1, keepdim=True)   # unbiased, as torch.var in the reference (:189)
        target = (target - mean) / (var + 1e-6) ** 0.5
    loss = ((pred - target) ** 2).mean(-1)
    loss = loss.sum() / (N * L)               # "mean loss on all patches" (:192-195)
    return loss, pred, mask


def loss_and_grads(sd, imgs, ids_shuffle, ids_restore, len_keep, model="mae_vit_base_patch16", dtype=torch.float32,
                   norm_pix_loss=False):
    P = {k: torch.as_tensor(v).to(dtype).clone() for k, v in sd.items()}
    for k, v in P.items():
        if k not in ("pos_embed", "decoder_pos_embed"):
            v.requires_grad_(True)
    loss, pred, mask = forward(P, torch.as_tensor(imgs).to(dtype), ids_shuffle, ids_restore, len_keep, model, norm_pix_loss)
    names = [k for k, v in P.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [P[k] for k in names])
    return loss.detach(), pred.detach(), mask, dict(zip(names, grads))
