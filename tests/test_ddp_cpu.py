"""CPU / gloo, world_size 2: the N > 1 path of the finetune step -- two-bucket gradient all-reduce over the flat buffer,
1/world folded into AdamW, identical shot_num on every rank, identical parameters after the step, and equality with the
single-process large-batch gradient (DDP semantics, FSC_finetune_cross.py:230)."""
import json
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

G = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from countr_amd.engine import ParamLayout
        from countr_amd.parallel import GradSync, shared_shot_num
        from oracle import countr_ref as R
        meta = json.load(open(os.path.join(G, "meta.json")))
        lay = ParamLayout([(a, tuple(b)) for a, b in meta["schema"]])
        n = lay.n_train
        gen = torch.Generator().manual_seed(100 + rank)
        g_local = torch.randn(n, generator=gen, dtype=torch.float64)
        S = shared_shot_num(step=7, seed=0)
        # grad-less parameters for this shot_num are zero-filled so the bucket is well defined on every rank
        s0, e0 = lay.bucket_range(2 if S == 0 else 3)
        g_local[s0:e0] = 0
        flat = g_local.clone()
        sync = GradSync(flat, lay.bucket_range(0), (lay.bucket_range(0)[1], n))
        sync.start_bucket0()
        sync.finish()
        # reference: explicit sum of both ranks' gradients
        other = torch.randn(n, generator=torch.Generator().manual_seed(100 + (1 - rank)), dtype=torch.float64)
        other[s0:e0] = 0
        assert torch.allclose(flat, g_local + other, atol=1e-12)
        # the finetune step's schedule: four buckets in backward-completion order, the first two started early, the bucket
        # without gradients for this shot_num skipped (it keeps its local, unreduced content and AdamW never reads it)
        flat4 = g_local.clone()
        sync4 = GradSync(flat4, None, None, buckets=[lay.bucket_range(b) for b in range(4)])
        sync4.start(0); sync4.start(1)
        skip = 2 if S == 0 else 3
        sync4.finish(skip=(skip,))
        for b in range(4):
            lo, hi = lay.bucket_range(b)
            want = g_local[lo:hi] if b == skip else (g_local + other)[lo:hi]
            assert torch.allclose(flat4[lo:hi], want, atol=1e-12), b
        assert [lay.bucket_range(b)[1] for b in range(3)] == [lay.bucket_range(b + 1)[0] for b in range(3)]
        # AdamW with grad_scale = 1/world == AdamW on the averaged gradient, only on the ranges with gradients
        p = torch.zeros(n, dtype=torch.float64) + 0.1
        m = torch.zeros(n, dtype=torch.float64); v = torch.zeros(n, dtype=torch.float64)
        p_ref = p.clone()
        for s, e, wd in lay.adam_ranges(S, 0.05):
            p[s:e], m[s:e], v[s:e] = R.adamw_step(p[s:e], flat[s:e] * sync.grad_scale, m[s:e], v[s:e], 1, 1e-3, wd=wd)
            p_ref[s:e], _, _ = R.adamw_step(p_ref[s:e], (g_local[s:e] + other[s:e]) / world, torch.zeros(e - s, dtype=torch.float64),
                                             torch.zeros(e - s, dtype=torch.float64), 1, 1e-3, wd=wd)
        assert torch.equal(p, p_ref)
        assert torch.all(p[s0:e0] == 0.1)  # skipped (no gradient), like torch AdamW with grad None
        # every rank holds identical parameters and the same shot_num
        gather = [torch.zeros_like(p) for _ in range(world)]
        dist.all_gather(gather, p)
        assert torch.equal(gather[0], gather[1])
        shots = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(shots, torch.tensor([S]))
        assert int(shots[0]) == int(shots[1])
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        out.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


def _worker_prs(rank, world, port, out):
    """Per-rank shot_num (reference semantics): rank 0 runs shot_num 3, rank 1 shot_num 0 on its half-batch; the flat gradient with the
    locally unused conditional bucket ZERO-FILLED, all four buckets reduced, must equal the sum of the oracle's two half-batch
    gradients (None -> 0), i.e. world x the mean gradient DDP(find_unused_parameters=True) hands to AdamW."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import math
        from countr_amd.engine import ParamLayout
        from countr_amd.parallel import GradSync, rank_shot_nums
        from oracle import countr_ref as R, weights as W
        torch.set_num_threads(4)
        name = "tiny_test"
        sd = W.make_state_dict(name, seed=3)
        lay = ParamLayout([(k, tuple(v.shape)) for k, v in sd.items()])
        shots = [3, 0]
        imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=4)

        def flat_of(r):
            _, _, g = R.loss_and_grads(sd, imgs[r:r + 1], boxes[r:r + 1], gt[r:r + 1], mask, shots[r], name, dtype=torch.float64)
            f = torch.full((lay.n_train,), float("nan"), dtype=torch.float64)      # NaN = "never written by this rank's backward"
            for k in lay.train_names:
                o = lay.off[k] - lay.train_start
                n = math.prod(lay.shapes[k])
                pad = -(-n // 4) * 4
                f[o:o + pad] = 0
                if g.get(k) is not None:
                    f[o:o + n] = g[k].reshape(-1)
            return f, g
        mine, g_mine = flat_of(rank)
        touched_local = {3 if shots[rank] == 0 else 2}
        touched_any = {3 if s == 0 else 2 for s in shots}
        assert touched_any == {2, 3}
        # the locally untouched conditional bucket holds whatever the last step left there: poison it, then zero-fill (what the step does)
        for b in touched_any - touched_local:
            lo, hi = lay.bucket_range(b)
            assert all(g_mine.get(k) is None for k in lay.train_names if lo <= lay.off[k] - lay.train_start < hi)
            mine[lo:hi] = float("nan")
            mine[lo:hi] = 0
        mine = torch.nan_to_num(mine, nan=0.0)        # alignment padding between tensors
        sync = GradSync(mine, None, None, buckets=[lay.bucket_range(b) for b in range(4)])
        sync.start(0); sync.start(1)
        sync.finish(skip=tuple(b for b in (2, 3) if b not in touched_any))
        other, _ = flat_of(1 - rank)
        for b in touched_any - {3 if shots[1 - rank] == 0 else 2}:
            lo, hi = lay.bucket_range(b)
            other[lo:hi] = 0
        other = torch.nan_to_num(other, nan=0.0)
        mine_again = torch.nan_to_num(flat_of(rank)[0], nan=0.0)
        for b in touched_any - touched_local:
            lo, hi = lay.bucket_range(b)
            mine_again[lo:hi] = 0
        assert torch.allclose(mine, mine_again + other, atol=1e-12)
        # both conditional sets now carry a non-zero reduced gradient on BOTH ranks
        for b in (2, 3):
            lo, hi = lay.bucket_range(b)
            assert mine[lo:hi].abs().max() > 0, b
        # every rank can evaluate every rank's draw: same list everywhere, own entry = own draw, the ban is per rank
        a = rank_shot_nums(11, world, seed=5)
        allv = [None] * world
        dist.all_gather_object(allv, a)
        assert allv[0] == allv[1] and all(0 <= s <= 3 for s in a)
        assert all(rank_shot_nums(t, world, seed=5, allow_zero=[False, True])[0] >= 1 for t in range(64))
        assert {tuple(rank_shot_nums(t, world, seed=5)) for t in range(64)} != {(s, s) for s in range(4)}   # the ranks' draws differ
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        out.put((rank, "FAIL: %r %s" % (e, traceback.format_exc()[-800:])))
    finally:
        dist.destroy_process_group()


def _worker_w8(rank, world, port, steps, out):
    """BASELINE config 3's world (8 ranks, FSC_finetune_cross.py:178-183,230) on the real model's bucket layout: per-rank shot draws,
    the trainer's own window / skip / AdamW set logic (the unbound methods on a stand-in object: the logic is host-only), zero-fill of
    the conditional bucket a rank has no gradient for, four-bucket all-reduce with the first two started early -- exact integer-valued
    gradients, so the reduced buffer must EQUAL the sum over the eight ranks whatever order gloo adds in."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types
        from countr_amd.engine import ParamLayout
        from countr_amd.parallel import GradSync, rank_shot_nums, shard_batch
        from countr_amd.trainer import FinetuneStep, _GraphStep
        torch.set_num_threads(1)
        meta = json.load(open(os.path.join(G, "meta.json")))
        lay = ParamLayout([(a, tuple(b)) for a, b in meta["schema"]])
        n = lay.n_train
        buckets = [lay.bucket_range(b) for b in range(4)]
        assert buckets[0][0] == 0 and buckets[3][1] == n and all(buckets[i][1] == buckets[i + 1][0] for i in range(3))
        assert [e - s for s, e in buckets][3] == 512                        # shot_token alone
        eng = types.SimpleNamespace(opt_seen=set())
        me = types.SimpleNamespace(per_rank=True, _touched=set(), _touched_any=set(), eng=eng)

        def grad_of(r, step, shots):
            g = torch.randint(-8, 9, (n,), generator=torch.Generator().manual_seed(1000 * step + r), dtype=torch.int32).float()
            unused = 2 if shots[r] == 0 else 3                             # the conditional bucket rank r's backward does not write
            g[buckets[unused][0]:buckets[unused][1]] = float("nan")         # ... holds stale content
            return g
        flat = torch.zeros(n)
        sync = GradSync(flat, None, None, buckets=buckets)
        assert sync.world == world and sync.comm and sync.grad_scale == 1.0 / world
        log = []
        for step in steps:
            shots = rank_shot_nums(step, world, seed=3)
            flat.copy_(grad_of(rank, step, shots))
            me._touched = {3 if shots[rank] == 0 else 2}                    # FinetuneStep._phases
            me._touched_any = {3 if s_ == 0 else 2 for s_ in shots}         # FinetuneStep.step(shots_all=...)
            touched, zfill = _GraphStep._window_sets(me)
            cskip = FinetuneStep._comm_skip(me, touched)
            skip, zero = FinetuneStep._adam_sets(me, touched)
            for b in zfill:                                                 # _zero_buckets
                flat[buckets[b][0]:buckets[b][1]] = 0
            sync.start(0); sync.start(1)
            sync.finish(skip=cskip)
            want = torch.zeros(n)
            for r in range(world):
                g = grad_of(r, step, shots)
                unused = 2 if shots[r] == 0 else 3
                if unused in touched:
                    g[buckets[unused][0]:buckets[unused][1]] = 0
                want += torch.nan_to_num(g, nan=0.0) if unused in touched else g
            for b in range(4):
                lo, hi = buckets[b]
                if b in cskip:                                              # nobody has a gradient: not reduced, never read
                    assert torch.isnan(flat[lo:hi]).all(), (step, b)
                    assert b in skip or b in zero
                else:
                    assert torch.equal(flat[lo:hi], want[lo:hi]), (step, b)
            log.append((step, tuple(shots), tuple(sorted(touched)), tuple(zfill), tuple(cskip), tuple(skip), tuple(zero), sorted(eng.opt_seen)))
        every = [None] * world
        dist.all_gather_object(every, [(a[0], a[1], a[2], a[4], a[5], a[6], a[7]) for a in log])     # all but the rank's own zero-fill set
        assert all(e_ == every[0] for e_ in every)
        parts = [shard_batch(64, r, world) for r in range(world)]           # config 3: global batch 64 -> 8 per rank, contiguous, complete
        assert parts == [(8 * r, 8 * r + 8) for r in range(world)]
        out.put((rank, "ok", log if rank == 0 else None))
    except Exception as e:  # noqa: BLE001
        import traceback
        out.put((rank, "FAIL: %r %s" % (e, traceback.format_exc()[-800:]), None))
    finally:
        dist.destroy_process_group()


def test_eight_rank_buckets_per_rank_shots_and_zero_fill():
    from countr_amd.parallel import rank_shot_nums
    world = 8
    draws = {t: rank_shot_nums(t, world, seed=3) for t in range(400)}
    no_zero = [t for t, d in draws.items() if all(s_ > 0 for s_ in d)]      # nobody uses shot_token: bucket 3 skipped by all
    mixed = [t for t, d in draws.items() if 0 in d and any(s_ > 0 for s_ in d)]
    assert no_zero and mixed
    # first a step where no rank draws 0 (shot_token: no gradient anywhere, never seen -> AdamW skips it), then mixed steps (both
    # conditional buckets reduced, zero-filled on the ranks without), then the no-zero step again (shot_token now stepped with g = 0)
    steps = [no_zero[0], mixed[0], mixed[1], no_zero[0]]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_w8, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted((r, st) for r, st, _ in res) == [(r, "ok") for r in range(world)], res
    log = [l for _r, _s, l in res if l is not None][0]
    assert log[0][4] == (3,) and log[0][5] == (3,) and log[0][6] == ()      # step 1: bucket 3 neither reduced nor stepped
    assert log[1][2] == (2, 3) and log[1][4] == () and log[1][5] == ()      # mixed: everything reduced and stepped
    assert log[3][4] == (3,) and log[3][5] == () and log[3][6] == (3,)      # later no-zero step: not reduced, stepped with g = 0 (torch 1.13 zero_grad)


def test_per_rank_shot_num_zero_filled_buckets_sum_to_the_oracle_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_prs, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_two_rank_gradient_sync_and_adamw():
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


def test_data_parallel_gradient_equals_large_batch_gradient():
    """Sharding a batch over 2 ranks and averaging gradients == the single-process gradient of the whole batch
    (the loss divides by the local batch, FSC_finetune_cross.py:295), checked with the CPU oracle on the reduced model."""
    from oracle import countr_ref as R, weights as W
    name = "tiny_test"
    sd = W.make_state_dict(name, seed=3)
    imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=4)
    _, _, g_all = R.loss_and_grads(sd, imgs, boxes, gt, mask, 3, name, dtype=torch.float64)
    parts = [R.loss_and_grads(sd, imgs[i:i + 1], boxes[i:i + 1], gt[i:i + 1], mask, 3, name, dtype=torch.float64)[2] for i in range(2)]
    for k, g in g_all.items():
        if g is None:
            continue
        avg = (parts[0][k] + parts[1][k]) / 2
        assert (avg - g).abs().max() <= 1e-9 * max(g.abs().max().item(), 1e-12) + 1e-15, k
