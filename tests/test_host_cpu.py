"""CPU-only tests of the host logic around the HIP path: C-ABI library exports, parameter layout / gradient buckets /
AdamW ranges, drop-in module surface (constructor, factories, state_dict schema), LR schedule, pos-embed tables,
and the loud failure when no GPU / no library is available (the product path has no CPU fallback)."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(G, "meta.json")))


def test_library_loads_and_exports_every_declared_symbol():
    from countr_amd import _lib
    L = _lib.lib()
    syms = _lib.exported_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), "include/countr_hip.h declares %s but libcountr_hip.so does not export it" % s
    assert L.countr_version() == _lib.ABI_VERSION == 9
    L16 = _lib.lib("f16")                        # the fp16 build of the same sources exports the same ABI
    assert all(hasattr(L16, s) for s in syms) and L16.countr_version() == _lib.ABI_VERSION and L16 is not L


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU error path")
def test_init_without_gpu_fails_loudly_not_silently():
    from countr_amd import _lib
    L = _lib.lib()
    rc = L.countr_init(0)
    assert rc < 0
    assert b"countr_init" in L.countr_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU error path")
def test_model_forward_on_cpu_raises_no_fallback():
    import models_mae_cross
    m = models_mae_cross.mae_vit_base_patch16(norm_pix_loss=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 384, 384), torch.zeros(1, 3, 3, 64, 64), 3)


def test_module_schema_matches_reference(meta):
    import models_mae_cross
    m = models_mae_cross.__dict__["mae_vit_base_patch16"](norm_pix_loss=False)
    keys = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    assert keys == [(a, b) for a, b in meta["schema"]]
    assert sum(p.numel() for p in m.parameters()) == meta["n_params"]
    # frozen pos-embeds, trainable rest (models_mae_cross.py:30,42)
    assert not m.pos_embed.requires_grad and not m.decoder_pos_embed.requires_grad
    # factories of models_mae_cross.py:210-253 exist; demo_zero.py:102 passes a string for norm_pix_loss
    for name in ("mae_vit_base4_patch16", "mae_vit_base6_patch16", "mae_vit_large_patch16", "mae_vit_huge_patch14",
                 "mae_vit_base_patch16_dec512d8b", "mae_vit_base_patch16_fim4", "mae_vit_base_patch16_fim6"):
        assert callable(models_mae_cross.__dict__[name])
    m4 = models_mae_cross.mae_vit_base4_patch16(norm_pix_loss="store_true")
    assert len(m4.decoder_blocks) == 4
    # init follows models_mae_cross.py:108-134: LayerNorm 1/0, Linear bias 0, sincos tables
    assert torch.all(m.norm.weight == 1) and torch.all(m.norm.bias == 0)
    assert torch.all(m.decoder_embed.bias == 0)
    g = np.load(os.path.join(G, "pos_embed_rows.npz"))
    assert np.abs(m.pos_embed[0, g["rows"]].numpy() - g["pe768"].astype(np.float32)).max() < 1e-6
    assert np.abs(m.decoder_pos_embed[0, g["rows"]].numpy() - g["pe512"].astype(np.float32)).max() < 1e-6


def test_load_state_dict_roundtrip_and_strict_false():
    import models_mae_cross
    from oracle import weights as W
    m = models_mae_cross.mae_vit_base_patch16()
    sd = {k: torch.from_numpy(v) for k, v in W.make_state_dict("mae_vit_base_patch16", 0).items()}
    m.load_state_dict(sd, strict=True)
    assert torch.equal(m.state_dict()["decode_head3.3.weight"], sd["decode_head3.3.weight"])
    del sd["pos_embed"]  # util/misc.py:400-421 drops pos_embed on shape mismatch and loads with strict=False
    r = m.load_state_dict(sd, strict=False)
    assert r.missing_keys == ["pos_embed"]


def test_param_layout_buckets_and_adam_ranges(meta):
    from countr_amd.engine import ParamLayout, is_trainable, no_weight_decay, ALIGN
    shapes = [(a, tuple(b)) for a, b in meta["schema"]]
    lay = ParamLayout(shapes)
    # every tensor aligned, no overlap, frozen region first
    spans = sorted((lay.off[n], lay.off[n] + int(np.prod(s)), n) for n, s in shapes)
    for (a0, a1, _), (b0, _, _) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert all(o % ALIGN == 0 for o in lay.off.values())
    assert all(lay.off[n] < lay.train_start for n, _ in shapes if not is_trainable(n))
    assert all(lay.off[n] >= lay.train_start for n, _ in shapes if is_trainable(n))
    n_train = sum(int(np.prod(s)) for n, s in shapes if is_trainable(n))
    assert n_train == 13306241 + 512  # 13 306 241 decoder-side elements + shot_token
    # buckets are contiguous and ordered by backward completion: head+decoder_norm, blocks+embed, exemplar CNN, shot_token
    b = [lay.bucket_range(i) for i in range(4)]
    assert b[0][0] == 0 and b[0][1] == b[1][0] and b[1][1] == b[2][0] and b[2][1] == b[3][0] and b[3][1] == lay.n_train
    for n, s in shapes:
        if n.startswith(("decode_head", "decoder_norm")):
            assert b[0][0] <= lay.off[n] - lay.train_start < b[0][1]
    # AdamW groups (timm add_weight_decay): 1-D tensors and biases get no decay
    assert no_weight_decay("decoder_norm.weight", (512,)) and no_weight_decay("decode_head0.0.bias", (256,))
    assert not no_weight_decay("decode_head0.0.weight", (256, 512, 3, 3))
    r3 = lay.adam_ranges(3, 0.05)
    r0 = lay.adam_ranges(0, 0.05)
    cover3 = sum(e - s for s, e, _ in r3)
    cover0 = sum(e - s for s, e, _ in r0)
    cnn = b[2][1] - b[2][0]
    tok = b[3][1] - b[3][0]
    assert cover3 == lay.n_train - tok and cover0 == lay.n_train - cnn   # grad-less parameters are skipped
    assert len(r3) <= 8 and len(r0) <= 8
    for s, e, wd in r3:
        assert wd in (0.0, 0.05)
    # LayerNorm weight/bias gradients are adjacent in the flat buffer
    assert lay.off["decoder_norm.bias"] == lay.off["decoder_norm.weight"] + 512


def test_lr_schedule_matches_reference_table(meta):
    from countr_amd.util.lr_sched import adjust_learning_rate

    class A:
        lr, min_lr, warmup_epochs, epochs = 1e-5, 0.0, 10, 1000

    class Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}]
    for e, lr in meta["lr_table"]:
        assert abs(adjust_learning_rate(Opt, e, A) - lr) < 1e-18
        assert Opt.param_groups[0]["lr"] == pytest.approx(lr) and Opt.param_groups[1]["lr"] == pytest.approx(0.5 * lr)


def test_gemm_arg_validation_needs_no_gpu():
    from countr_amd import _lib
    L = _lib.lib()
    a = _lib.GemmArgs()
    assert L.countr_gemm(C.byref(a), 1, 0, 0, None) < 0  # null pointers are rejected before any launch
    assert b"countr_gemm" in L.countr_last_error()


def test_shared_shot_num_and_sharding():
    from countr_amd.parallel import shared_shot_num, shard_batch
    draws = [shared_shot_num(s, seed=3) for s in range(200)]
    assert draws == [shared_shot_num(s, seed=3) for s in range(200)] and set(draws) == {0, 1, 2, 3}
    assert set(shared_shot_num(s, 3, allow_zero=False) for s in range(200)) == {1, 2, 3}
    parts = [shard_batch(35, r, 8) for r in range(8)]
    assert parts[0][0] == 0 and parts[-1][1] == 35 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert max(e - s for s, e in parts) - min(e - s for s, e in parts) <= 1


def test_philox_oracle_matches_random123_known_answers():
    """oracle/philox.py (the checker of the step prologue's loss-mask draw) against the three philox4x32-10 known-answer vectors the
    Random123 distribution ships (kat_vectors: zero, all-ones, pi-digit counters / keys), + the Bernoulli(0.8) mask built on it."""
    import numpy as np
    from oracle.philox import philox4x32_10, loss_mask
    kat = [([0, 0, 0, 0], (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
           ([0xffffffff] * 4, (0xffffffff, 0xffffffff), "408f276d 41c83b0e a20bc7c6 6d5451fd"),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0), "d16cfe09 94fdcceb 5001e420 24126ea1")]
    for ctr, key, want in kat:
        got = " ".join("%08x" % v for v in philox4x32_10(np.array(ctr, dtype=np.uint32), key))
        assert got == want, (ctr, got)
    m0, m1 = loss_mask(7, 0), loss_mask(7, 1)
    assert m0.shape == (384 * 384,) and set(np.unique(m0)) == {0.0, 1.0}
    assert abs(m0.mean() - 0.8) < 5e-3 and abs(m1.mean() - 0.8) < 5e-3          # sigma of the mean = 1.04e-3
    assert 0.15 < (m0 != m1).mean() < 0.5                                        # independent draws differ on 2 p (1 - p) = 32 %
    assert (loss_mask(8, 0) != m0).any()


def test_init_distributed_mode_reads_the_three_launcher_conventions(monkeypatch):
    """util/misc.py:225-257: --dist_on_itp (OpenMPI variables, tcp:// URL, torchrun's variables exported), RANK / WORLD_SIZE / LOCAL_RANK,
    SLURM_PROCID (world size from --world_size, gpu = rank modulo the devices), none of them -> single process."""
    import types
    import torch.distributed as dist
    from countr_amd.util import misc
    calls = []
    monkeypatch.setattr(dist, "init_process_group", lambda **kw: calls.append(kw))
    monkeypatch.setattr(dist, "barrier", lambda: None)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.append(("device", d)))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SLURM_PROCID", "OMPI_COMM_WORLD_RANK", "MASTER_ADDR", "MASTER_PORT", "COUNTR_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    a = types.SimpleNamespace(dist_on_itp=False, world_size=1, dist_url="env://")
    misc.init_distributed_mode(a)
    assert not a.distributed and (a.rank, a.world_size, a.gpu) == (0, 1, 0) and not calls
    for k, v in (("OMPI_COMM_WORLD_RANK", "5"), ("OMPI_COMM_WORLD_SIZE", "8"), ("OMPI_COMM_WORLD_LOCAL_RANK", "5"), ("MASTER_ADDR", "10.0.0.1"), ("MASTER_PORT", "2345")):
        monkeypatch.setenv(k, v)
    a = types.SimpleNamespace(dist_on_itp=True, world_size=1, dist_url="env://")
    misc.init_distributed_mode(a)
    assert a.distributed and (a.rank, a.world_size, a.gpu, a.dist_url) == (5, 8, 5, "tcp://10.0.0.1:2345")
    assert os.environ["RANK"] == "5" and os.environ["WORLD_SIZE"] == "8" and os.environ["LOCAL_RANK"] == "5"
    assert calls[-1] == dict(backend="nccl", init_method="tcp://10.0.0.1:2345", world_size=8, rank=5) and ("device", 5) in calls
    calls.clear()
    monkeypatch.setenv("RANK", "3"); monkeypatch.setenv("WORLD_SIZE", "4"); monkeypatch.setenv("LOCAL_RANK", "3")
    a = types.SimpleNamespace(dist_on_itp=False, world_size=1, dist_url="env://")
    misc.init_distributed_mode(a)
    assert a.distributed and (a.rank, a.world_size, a.gpu) == (3, 4, 3) and calls[-1]["init_method"] == "env://"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k)
    monkeypatch.setenv("SLURM_PROCID", "11")
    a = types.SimpleNamespace(dist_on_itp=False, world_size=16, dist_url="tcp://head:1")
    misc.init_distributed_mode(a)
    assert a.distributed and (a.rank, a.world_size, a.gpu) == (11, 16, 3) and calls[-1]["rank"] == 11 and calls[-1]["world_size"] == 16


def test_checkpoint_scaler_entry_is_a_gradscaler_state_or_absent(tmp_path):
    """util/misc.py:312-318 saves loss_scaler.state_dict() under 'scaler' and :418-419 reloads it `if 'scaler' in checkpoint`; torch's
    GradScaler.load_state_dict raises on an EMPTY dict.  So: fp16 steps hand save_model the device-side scaler's state (the GradScaler
    keys), bf16 / fp32 steps hand it None and the key is omitted -- either way the reference's resume accepts the file."""
    import types
    from countr_amd.util import misc
    args = types.SimpleNamespace(output_dir=str(tmp_path), lr=1e-5)
    model = torch.nn.Linear(2, 2)
    p = misc.save_model(args, 3, model, {"state": {}, "param_groups": []}, suffix="a")
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert "scaler" not in ck and ck["epoch"] == 3 and set(ck) == {"model", "optimizer", "epoch", "args"}
    st = {"scale": 4096.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 17}
    p = misc.save_model(args, 4, model, {"state": {}, "param_groups": []}, suffix="b", scaler_state=st)
    ck = torch.load(p, map_location="cpu", weights_only=False)
    assert ck["scaler"] == st
    # the file's entry is what torch's own GradScaler writes (same key set)
    ref = torch.amp.GradScaler("cpu", enabled=True)
    assert set(ref.state_dict()) == set(st)


def test_density_maps_stream_looks_one_group_ahead(monkeypatch):
    """Host logic of inference.density_maps_stream (no GPU): group k is handed group k + 1's images as `ahead`, the ownership token a
    call returns for the encoder forward it ran ahead is handed to the next call as `have` (and only to the next), results come in
    order, a group the native path refuses goes to density_maps, and an iterator is read exactly one group ahead."""
    from countr_amd import inference
    seen, pulled = [], []

    def fake_native(model, images, boxes, shot_num, max_batch, want_sums, have=None, ahead=None, flags=None):
        seen.append((images, have, ahead))
        if images == "torchpath":
            return None
        flags["ahead"] = ("tok", images) if ahead not in (None, "torchpath") else None
        return "res:" + images

    monkeypatch.setattr(inference, "_native_maps", fake_native)
    monkeypatch.setattr(inference, "density_maps", lambda model, images, boxes, S, mb, rs=False: "torch:" + images)

    def gen():
        for name in ("a", "b", "torchpath", "c", "d"):
            pulled.append(name)
            yield name, None
    out = []
    for r in inference.density_maps_stream(None, gen(), 0):
        out.append((r, list(pulled)))
    assert [r for r, _p in out] == ["res:a", "res:b", "torch:torchpath", "res:c", "res:d"]
    assert out[0][1] == ["a", "b"] and out[1][1] == ["a", "b", "torchpath"]          # one group of look-ahead, no more
    assert seen == [("a", None, "b"), ("b", ("tok", "a"), "torchpath"), ("torchpath", None, "c"), ("c", None, "d"), ("d", ("tok", "c"), None)]
