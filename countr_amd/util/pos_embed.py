"""2-D sin-cos position embedding tables (init-time only).

Same table as the reference's util/pos_embed.py:20-67: float64 numpy, first half of the channels encodes
the column (w) index, second half the row; each half is [sin(p*w_k), cos(p*w_k)], w_k = 10000^(-k/(D/4)).
"""
import numpy as np


def _sincos_1d(dim, pos):
    omega = np.arange(dim // 2, dtype=np.float64) / (dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", np.asarray(pos, dtype=np.float64).reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    assert embed_dim % 4 == 0
    cols, rows = np.meshgrid(np.arange(grid_size, dtype=np.float32), np.arange(grid_size, dtype=np.float32))
    emb = np.concatenate([_sincos_1d(embed_dim // 2, cols), _sincos_1d(embed_dim // 2, rows)], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb
