"""CPU: host logic of the MAE pretraining path -- module schema vs the reference, no CPU fallback, flat-buffer layout
and gradient buckets, patchify/unpatchify index shuffles, data-parallel gradient property (oracle), 2-rank gloo bucket sync."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mae_ref as M
from oracle import weights as W

G = os.path.join(os.path.dirname(__file__), "golden")


def test_mae_module_schema_matches_reference_and_has_no_cpu_path():
    import models_mae_noct
    meta = json.load(open(os.path.join(G, "mae_meta.json")))
    m = models_mae_noct.mae_vit_base_patch16()
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(a, b) for a, b in meta["schema"]]
    frozen = sorted(k for k, p in m.named_parameters() if not p.requires_grad)
    assert frozen == ["decoder_pos_embed", "pos_embed"]                       # models_mae_noct.py:25,39
    assert set(models_mae_noct.__dict__) >= {"mae_vit_base_patch16", "mae_vit_large_patch16", "mae_vit_huge_patch14",
                                             "mae_vit_base_patch16_dec512d8b", "MaskedAutoencoderViTNoCT"}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 384, 384), mask_ratio=0.5)
    # reference initialisation: sin-cos tables, zero biases, unit LayerNorm
    sd = W.make_state_dict_mae()
    assert np.allclose(m.pos_embed.numpy(), sd["pos_embed"], atol=1e-6) and np.allclose(m.decoder_pos_embed.numpy(), sd["decoder_pos_embed"], atol=1e-6)
    assert float(m.decoder_pred.bias.detach().abs().max()) == 0.0 and float((m.norm.weight.detach() - 1).abs().max()) == 0.0
    assert m.len_keep(0.5) == 288 and m.len_keep(0.75) == 144


def test_patchify_roundtrip_matches_oracle():
    import models_mae_noct
    m = models_mae_noct.mae_vit_base_patch16()
    x = torch.rand(2, 3, 384, 384, generator=torch.Generator().manual_seed(0))
    p = m.patchify(x)
    assert torch.equal(p, M.patchify(x, 16))
    assert torch.equal(m.unpatchify(p), x)


def test_mae_layout_buckets_cover_all_trainable_parameters():
    from countr_amd.engine import ParamLayout, no_weight_decay
    from countr_amd.mae_engine import mae_bucket_fn, mae_trainable
    shapes = [(n, s) for n, s, _ in W.schema_mae()]
    from countr_amd.mae_engine import mae_enc_parts
    assert mae_enc_parts(12) == 6 and mae_enc_parts(2) == 2
    old = mae_bucket_fn(12, parts=3)                       # round 2's thirds (the function still takes a part count)
    assert [old("blocks.%d.attn.qkv.weight" % i) for i in range(12)] == [3] * 4 + [2] * 4 + [1] * 4 and old("patch_embed.proj.weight") == 3
    mae_bucket = mae_bucket_fn(12)
    NB = 7                                                  # decoder side + six encoder groups of two blocks
    lay = ParamLayout(shapes, trainable=mae_trainable, bucket=mae_bucket)
    assert lay.frozen_names == ["pos_embed", "decoder_pos_embed"]
    rng = [lay.bucket_range(b) for b in range(NB)]
    assert rng[0][0] == 0 and rng[-1][1] == lay.n_train and all(rng[b][1] == rng[b + 1][0] for b in range(NB - 1))   # contiguous RCCL buckets
    assert mae_bucket("decoder_pred.weight") == 0 and mae_bucket("mask_token") == 0 and mae_bucket("norm.weight") == 1
    assert [mae_bucket("blocks.%d.attn.qkv.weight" % i) for i in range(12)] == [6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1]   # backward order
    assert mae_bucket("patch_embed.proj.weight") == 6
    assert max(4 * (e - s) for s, e in rng[1:]) < 60e6     # the exposed (last) encoder bucket: 57 MB of fp32 gradient, was 115
    for n in lay.train_names:
        o = lay.off[n] - lay.train_start
        lo, hi = rng[mae_bucket(n)]
        assert lo <= o and o + int(np.prod(lay.shapes[n])) <= hi, n
        assert lay.off[n] % 64 == 0
    # segments alternate (bucket, no-decay first); every trainable element is in exactly one AdamW range
    keys = [tuple(k) for k, _, _ in lay.segments]
    assert keys == [(b, nd) for b in range(NB) for nd in (True, False)]
    covered = sum(e - s for _, s, e in lay.segments)
    assert covered == lay.n_train
    for (bk, nodecay), s, e in lay.segments:
        for n in lay.train_names:
            o = lay.off[n] - lay.train_start
            if s <= o < e:
                assert mae_bucket(n) == bk and no_weight_decay(n, lay.shapes[n]) == nodecay, n
    # LayerNorm weight/bias are adjacent in the flat gradient (single dgamma/dbeta finisher launch)
    assert lay.off["blocks.0.norm1.bias"] == lay.off["blocks.0.norm1.weight"] + 768


def test_mae_data_parallel_gradient_equals_large_batch_gradient():
    """The loss is a mean over the local batch's patches (models_mae_noct.py:192-195), so averaging the per-rank gradients of a
    sharded batch equals the whole-batch gradient -- what PretrainStep's sum all-reduce + 1/world in AdamW relies on."""
    name = "tiny_test"
    sd = W.make_state_dict_mae(name, seed=1)
    imgs, ids_shuffle, ids_restore, K = W.make_mae_inputs(batch=2, seed=3)
    _, _, _, g_all = M.loss_and_grads(sd, imgs, ids_shuffle, ids_restore, K, name, dtype=torch.float64)
    parts = [M.loss_and_grads(sd, imgs[i:i + 1], ids_shuffle[i:i + 1], ids_restore[i:i + 1], K, name, dtype=torch.float64)[3] for i in range(2)]
    for k, g in g_all.items():
        avg = (parts[0][k] + parts[1][k]) / 2
        assert (avg - g).abs().max() <= 1e-9 * max(g.abs().max().item(), 1e-12) + 1e-15, k


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from countr_amd.engine import ParamLayout
        from countr_amd.mae_engine import mae_bucket_fn, mae_trainable
        from countr_amd.parallel import GradSync
        lay = ParamLayout([(n, s) for n, s, _ in W.schema_mae("tiny_test")], trainable=mae_trainable, bucket=mae_bucket_fn(2))
        n = lay.n_train
        mine = torch.randn(n, generator=torch.Generator().manual_seed(50 + rank), dtype=torch.float64)
        other = torch.randn(n, generator=torch.Generator().manual_seed(50 + (1 - rank)), dtype=torch.float64)
        flat = mine.clone()
        sync = GradSync(flat, None, None, buckets=[lay.bucket_range(b) for b in range(4)])
        for b in range(3):                         # PretrainStep: bucket b starts while the next backward phase runs
            sync.start(b)
        assert sync.grad_scale == 0.5
        sync.finish()
        assert torch.allclose(flat, mine + other, atol=1e-12)
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        out.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


def test_two_rank_bucket_sync_on_the_mae_layout():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
