#!/usr/bin/env python3
"""FSC147 finetune-step throughput of the CounTR HIP engine (BASELINE.json metric).

python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)
One step = forward (frozen ViT-B/16 encoder + decoder) + masked-MSE + decoder-side backward + gradient all-reduce +
AdamW on a synthetic batch of 8 images 384x384 with 3 exemplars per GPU (configs[1]); weak scaling.
Prints ONE JSON line with the whole-job images/sec, the roofline of the attention core kernel (live HIP-event
timing) and the CPU baseline (the oracle's finetune step on the host cores, bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (the pool's driver supports dmabuf IPC only: RCCL across processes needs this)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GF_STEP_PER_IMG = 321.16e9   # encoder fwd + 3x decoder side
ATT_FLOP_PER_IMG_LAYER = 4 * 576 * 576 * 64 * 12  # 4 N^2 dh H = 1.0192 GF
MFMA_BF16_PEAK = 2.5e15


def _sha(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def cpu_baseline(batch=4, steps=4, threads=None):
    """The oracle (CPU restatement of the reference, validated against it) timed on this host's cores.
    Threads are capped at 32: torch-CPU on all 256 hardware threads of the GPU box is ~100x slower (oversubscription)."""
    from oracle import countr_ref as R, weights as W
    threads = threads or min(os.cpu_count(), 32)
    torch.set_num_threads(threads)
    sd = W.make_state_dict("mae_vit_base_patch16", seed=0)
    imgs, boxes, gt, mask = W.make_inputs(batch=batch, shots=3, seed=0)
    R.loss_and_grads(sd, imgs[:1], boxes[:1], gt[:1], mask, 3)  # warm-up
    t0 = time.time()
    for _ in range(steps):
        out, loss, grads = R.loss_and_grads(sd, imgs, boxes, gt, mask, 3)
        for k, g in grads.items():
            if g is not None:
                R.adamw_step(torch.from_numpy(sd[k]), g, torch.zeros_like(g), torch.zeros_like(g), 1, 1e-5)
    dt = time.time() - t0
    return {"value": batch * steps / dt, "unit": "images/sec", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "%d finetune step(s) of batch %d (fwd + decoder bwd + AdamW), fp32, torch-CPU oracle" % (steps, batch)}


def parity_check(model, step, world, rank, B, NB, dev, kick=lambda what: None):
    """The timed object -- the same FinetuneStep, graph replay, batch k = 0 of every rank -- against the oracle, OUTSIDE every timed region
    (FSC_finetune_cross.py:286-316).  The oracle is the CHECKER here, never the thing measured.  Every rank steps once on its batch 0 with
    that batch's seeded loss mask; rank 0 evaluates the oracle at the parameters the engine held in front of the step for EVERY rank's
    batch and compares: its own loss (1e-2) and counts (5 % of the density map's mass), and per trainable tensor the gradient left in the step's flat buffer --
    after the all-reduce that is the SUM over ranks, so at N > 1 this also checks what RCCL carried -- by direction (cos >= 0.998;
    exemplar CNN 0.95) and norm (3 % / 5 %).  These are GUARD bars: the check must hold behind any number of warm-up steps the caller
    asks for (measured over --warmup 2..12: min cos 0.99963-0.99988, norm error 0.4-1.4 %, exemplar CNN 0.976-0.979, counts 0.4-1.9 % of
    the map's mass); the tight bars on a fixed schedule are tests/test_trainer_gpu.py::test_finetune_step_at_the_real_config_matches_oracle's.
    At N > 1 every rank evaluates the oracle for ITS OWN batch (in parallel: the check takes one oracle pass, not N) and hands the gradients
    to rank 0 through files in a per-job directory under the node's /tmp (bench.py is a one-node program by contract) -- not through RCCL,
    which is the thing being checked; rank 0 polls for them and kicks the hang watchdog while it waits.
    Raises on a miss; returns the dict reported as `parity` in the JSON line."""
    from countr_amd.synthetic import make_batch
    step.flush()                                    # (defer_optimizer: the parameters the checked step starts from)
    cur = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    imgs, boxes, gt, mask = make_batch(B, shots=3, seed=rank * NB, device=dev)
    with step.on_stream():
        step.load(imgs, boxes, gt, mask, 3)
        sums = step.step(3).clone()
    torch.cuda.synchronize()
    from oracle import countr_ref as R
    import numpy as np
    t0 = time.time()
    torch.set_num_threads(max(1, min((os.cpu_count() or 1) // max(world, 1), 32)))
    b = [t.cpu().numpy() for t in (imgs, boxes, gt, mask)]
    out, rloss, rg = R.loss_and_grads(cur, b[0], b[1], b[2], b[3], 3)
    total = {k: g.double() for k, g in rg.items() if g is not None}
    if world > 1:
        xdir = os.path.join("/tmp", "countr_bench_parity_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
        os.makedirs(xdir, exist_ok=True)
        if rank != 0:
            tmp = os.path.join(xdir, "r%d.tmp.npz" % rank)
            np.savez(tmp, **{k: g.float().numpy() for k, g in total.items()})
            os.replace(tmp, os.path.join(xdir, "r%d.npz" % rank))          # (atomic: rank 0 never reads a partial file)
            return None
        limit = float(os.environ.get("COUNTR_BENCH_PARITY_WAIT_S", "900"))
        for r in range(1, world):
            path = os.path.join(xdir, "r%d.npz" % r)
            while not os.path.exists(path):
                if time.time() - t0 > limit:
                    raise SystemExit("bench.py: parity check: rank %d's oracle gradients did not arrive within %.0f s" % (r, limit))
                kick("parity check: waiting for the oracle pass of rank %d" % r)
                time.sleep(0.5)
            with np.load(path) as z:
                for k in z.files:
                    g = torch.from_numpy(z[k]).double()
                    total[k] = g if k not in total else total[k] + g
            os.remove(path)
            kick("parity check")
        try:
            os.rmdir(xdir)
        except OSError:
            pass
    loss = sums[0].item()
    rel_loss = abs(loss - rloss.item()) / abs(rloss.item())
    rc = R.counts(out).numpy()
    # counts: a torch-initialised model's density map is signed and its count (sum / 60) swings through zero as the first steps move the
    # last bias (-460, -30, +350, +530 after 2, 3, 4, 5 steps), so the error is taken against the map's MASS (sum |map| / 60: equal to the
    # count for a trained, non-negative map) -- relative to the count itself the same ~2-count bf16 error read 0.4 % after five
    # warm-up steps and 9.7 % after three
    mass = float((abs(out.numpy()).reshape(B, -1).sum(1) / 60).max())
    rel_cnt = float(abs(sums[1:1 + B].cpu().numpy() - rc).max() / mass)
    if os.environ.get("COUNTR_BENCH_PARITY_DEBUG") == "1":
        import numpy as _np
        _np.set_printoptions(precision=2, linewidth=400, suppress=True)
        sys.stderr.write("parity debug: engine %s\nparity debug: oracle %s\nparity debug: mass %.2f\n" % (sums[1:1 + B].cpu().numpy(), rc, mass))
    worst_cos, worst_norm, worst_cnn, checked = 1.0, 0.0, 1.0, 0
    loss_scale = step.loss_scale                  # (fp16 mode: the flat buffer holds loss_scale x gradient)
    for k, ref in total.items():
        if ref.norm() < 1e-3:
            continue
        got = step.eng.gview(k).detach().cpu().double() / loss_scale
        cos = ((got * ref).sum() / (got.norm() * ref.norm())).item()
        ratio = (got.norm() / ref.norm()).item()
        if k.startswith("decoder_proj"):
            ok = cos > 0.95 and abs(ratio - 1) < 0.05
            worst_cnn = min(worst_cnn, cos)
        else:
            ok = cos > 0.998 and abs(ratio - 1) < 0.03
            worst_cos, worst_norm = min(worst_cos, cos), max(worst_norm, abs(ratio - 1))
        if not ok:
            raise SystemExit("bench.py: parity check FAILED on %s: cos %.5f, norm ratio %.4f" % (k, cos, ratio))
        checked += 1
    # (counts: tests hold 1 % on the pinned golden weights; this model is torch-initialised and the check must hold after ANY number of
    # warm-up steps -- measured 0.3-2 % of the map's mass over 2..12 steps -- so the bar here is 5 %; tensors: 54-58 of the 74 trainable
    # ones carry a gradient norm above 1e-3 depending on the state, the others are zero by construction (biases in front of a
    # normalisation) or too small to judge a direction in bf16)
    if rel_loss > 1e-2 or rel_cnt > 5e-2 or checked < 40:
        raise SystemExit("bench.py: parity check FAILED: loss off by %.2e, counts by %.2e, %d gradient tensors" % (rel_loss, rel_cnt, checked))
    return {"checked": True, "against": "oracle/countr_ref.py (fp32 torch-CPU restatement pinned to the reference's goldens) at the engine's own parameters",
            "object": "the timed FinetuneStep (graph replay), batch 0 of every rank, shot_num 3", "loss_rel_err": rel_loss, "count_rel_err": rel_cnt,
            "gradient_tensors": checked, "gradients": "flat buffer after the all-reduce vs the sum of the oracle's per-rank gradients" if world > 1
            else "flat buffer vs the oracle's gradients", "min_cos": worst_cos, "max_norm_err": worst_norm, "min_cos_exemplar_cnn": worst_cnn,
            "seconds": time.time() - t0}


def attention_roofline(model, batch, iters=20, instep_passes=6, shot=3, train=True):
    """Roofline of the dominant fused kernel: the attention core of one encoder layer (B x 12 heads, N=576, dh=64).
    `us_per_launch` is measured IN-STEP with HIP events on the launch stream: the forward launch list of the plan is replayed
    eagerly, so each attention launch sees the cache state the preceding qkv GEMM leaves (12 launches per pass).  An event pair
    around a single ~15 us kernel reads 2-3 us more than the kernel (an EMPTY pair reads `event_pair_overhead_us`), so the figure is
    differential: [attention + following proj GEMM] minus [proj GEMM alone]; the plain single-launch bracket is reported beside it
    (`us_per_launch_direct_bracket`), as is the back-to-back figure (`iters` launches in a row on one cache-hot qkv).  The rocprofv3
    kernel-trace average of the same command is committed under profiles/ (r2_bench_kernel_stats.csv)."""
    eng = model._engine()
    p = eng.plan(batch, shot, train)
    st = torch.cuda.current_stream()
    ops = p.fwd_par
    att = [i for i, (fn, args, _k) in enumerate(ops) if fn is eng.L.countr_attn_fwd and args[5] == eng.H and args[6] == eng.D // eng.H]
    # differential bracket: pass A times [attention + the launch behind it (proj GEMM)], pass B only [that launch] with the
    # attention in front of the first event -- the event-pair cost (a few us, `event_pair_overhead_us`) is in both and cancels
    with_att, without, direct, empty = [], [], [], []
    for k in range(3 * instep_passes):
        evs, prev, mode = [], 0, k % 3            # 0: [attn, next]  1: attn [next]  2: [attn] (direct bracket, reported beside)
        for i in att:
            eng.run(ops[prev:i])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if mode == 1:
                eng.run(ops[i:i + 1])
                e0.record(st)
                eng.run(ops[i + 1:i + 2])
            elif mode == 0:
                e0.record(st)
                eng.run(ops[i:i + 2])
            else:
                e0.record(st)
                eng.run(ops[i:i + 1])
            e1.record(st)
            if mode == 2:
                eng.run(ops[i + 1:i + 2])
            evs.append((e0, e1))
            prev = i + 2
        eng.run(ops[prev:])
        z0, z1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        z0.record(st); z1.record(st)
        torch.cuda.synchronize()
        (with_att, without, direct)[mode].extend(a.elapsed_time(b) * 1e3 for a, b in evs)
        empty.append(z0.elapsed_time(z1) * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    us = med(with_att) - med(without)
    if not us > 0.25 * med(direct):      # (never seen) a disturbed run: fall back to the plain bracket rather than report nonsense
        us = med(direct)
    one = []
    eng._attention_fwd(one, p, p.buf["qkv"], p.buf["att"], batch, eng.H, eng.D, prescaled=eng.prescale_q)
    for _ in range(3):
        eng.run(one)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        eng.run(one)
    e1.record(st)
    torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) * 1e3 / iters
    achieved = ATT_FLOP_PER_IMG_LAYER * batch / (us * 1e-6)
    traffic, traffic_source = None, None
    for rnd in ("r6", "r5", "r4", "r3", "r2"):   # HBM/fabric bytes per launch: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/README.md), NOT this run
        try:
            path = os.path.join(ROOT, "profiles", "%s_attention_traffic.json" % rnd)
            t = json.load(open(path))
            if batch == 8:
                traffic = t["bytes_per_launch"]
                traffic_source = "static: profiles/%s_attention_traffic.json (sha256 %s; rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate " \
                                 "passes over the eager step), not measured in this run" % (rnd, _sha(path))
            break
        except Exception:  # noqa: BLE001
            continue
    return {"bound": "mfma", "achieved": achieved / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
            "frac": achieved / MFMA_BF16_PEAK, "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": 28311552 * batch // 8,
            "north_star_target_frac": 0.9,
            "measured_ceiling_frac": 0.67,
            "ceiling_note": "0.9 of 2.5 PF is not reachable for N = 576 on this chip whatever the kernel does: (1) with all 256 CUs issuing "
                            "MFMAs the clock settles at 1.92 GHz = 2.02 PF sustained = 0.81 (tools/ubench_clock.hip, profiles/r3_clock_microbench.txt); "
                            "(2) 576 query rows are 18 waves of 32 rows per (batch, head), 1728 waves on 2048 wave slots: 0.84 of that = 0.67.  This "
                            "kernel's distance from 0.67: at B = 8 every workgroup is resident at once, so prologue (4.5 k cycles) + tail / epilogue "
                            "(~5 k) of a 24 k-cycle workgroup are not amortised; in the loop the two waves of a SIMD need 1617 cycles per step for "
                            "2 x 16 MFMAs = 63 % of the matrix pipe, and that is the cost of the instruction stream itself: a replica of the step's "
                            "compute without any memory traffic (tools/ubench_fa_step.hip, profiles/r3_fa_step_microbench.txt) gives the younger "
                            "wave 1611 cycles -- 1024 matrix + ~590 cycles of exp / row-sum / pack / max VALU that two waves cannot hide under "
                            "their MFMAs on this SIMD (v_exp_f32 alone: 210).  Round 5 (counters in profiles/r5_attention_pmc.txt: matrix pipes "
                            "26.9 % busy, 33.7 % of wave cycles waiting, traffic 1.005 x algorithmic): a persistent form has nothing to amortise "
                            "at B = 8 (1728 strips for 2048 wave slots: one strip per wave), and at B = 32 the launch already is 3.75 rounds of "
                            "workgroups with two per CU in different phases -- the overlap a persistent loop would schedule -- and reaches "
                            "roofline_b32.frac = 0.25 against the loop's own 0.33 (DESIGN.md section 5).  Round 6: the running row max left the loop (the "
                            "first key tile's max stays the softmax reference; overflow is detected at the end and such a workgroup reruns on the "
                            "exact loop) -- the microbenchmark's 1611 -> 1398 cycles per step; same-box 17.2 -> 16.4 us stand-alone, 14.7 -> 14.3 in "
                            "the step (profiles/r6_attention_pmc.txt: 27.5 % matrix-busy); the 40 % of a workgroup's life outside the loop stays.",
            "kernel": "fa_fwd_pipe_kernel<64>: encoder attention core (QK^T, softmax, PV), one launch per layer",
            "us_per_launch": us,
            "timing": "in-step, HIP events on the launch stream, %d encoder launches x %d eager forward passes: median of [attention + next launch] "
                      "minus median of [next launch] (the event-pair cost cancels)" % (len(att), instep_passes),
            "us_per_launch_direct_bracket": med(direct), "event_pair_overhead_us": med(empty), "us_per_launch_back_to_back": b2b}


def family_breakdown(model, step, batch, passes=5):
    """Where the step's GPU time goes, measured in this run: the step's launch lists (forward, loss, backward, AdamW) are replayed
    eagerly in their real order; every maximal run of consecutive launches of one family is bracketed by a HIP event pair on the launch
    stream and the cost of an empty pair is subtracted per run.  Families: linear (nn.Linear fwd / dgrad / wgrad GEMMs), conv (3x3
    implicit-GEMM convolutions fwd / dgrad / wgrad + the 3->64 direct conv), attention (fused self-attention fwd / bwd), other
    (norms, pooling, upsampling, cross-attention, reductions, loss, AdamW).  Algorithmic work per image: encoder linear
    layers 97.85 GF + decoder linear layers fwd 9.7 GF (x3 with backward), convolutions 58.45 + 1.4 GF fwd (x3), attention 1.0192 GF
    per encoder layer."""
    eng = model._engine()
    L = eng.L
    p = eng.plan(batch, 3, True)
    st = torch.cuda.current_stream()
    conv_fns = (L.countr_conv3x3_c3_fwd, L.countr_conv3x3_c3_wgrad)      # (ctypes function objects are not hashable)
    attn_fns = (L.countr_attn_fwd, L.countr_attn_bwd)

    def fam(fn, args):
        if fn is L.countr_gemm:
            return "conv" if (args[2] == 2 or args[3] == 3) else "linear"      # OP_IM2ROW A operand / OP_IM2COL B operand
        if fn is L.countr_gemm_group:                                          # a block's nn.Linear weight gradients in one launch
            return "linear"
        if any(fn is f for f in conv_fns):
            return "conv"
        if any(fn is f for f in attn_fns):
            return "attention"
        return "other"
    lists = [p.fwd_par, p.bwd_head, p.bwd_rest, p.bwd_tok]
    seq = [(fn, args, keep) for ops in lists for (fn, args, keep) in ops if fn is not None]
    runs = []
    for op in seq:
        f = fam(op[0], op[1])
        if runs and runs[-1][0] == f:
            runs[-1][1].append(op)
        else:
            runs.append([f, [op]])
    tot = {"linear": [], "conv": [], "attention": [], "other": []}
    empties = []
    for _ in range(passes):
        evs = []
        for f, ops in runs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            eng.run(ops)
            e1.record(st)
            evs.append((f, e0, e1))
        z0, z1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        z0.record(st); z1.record(st)
        torch.cuda.synchronize()
        empty = z0.elapsed_time(z1) * 1e3
        empties.append(empty)
        acc = {k: 0.0 for k in tot}
        for f, e0, e1 in evs:
            acc[f] += max(e0.elapsed_time(e1) * 1e3 - empty, 0.0)
        for k in tot:
            tot[k].append(acc[k])
    med = lambda v: sorted(v)[len(v) // 2]
    us = {k: med(v) for k, v in tot.items()}
    gf = {"linear": (97.85 + 0.453 + 3 * 2 * 4.86) * 1e9 * batch,                 # encoder fwd + decoder_embed + 3 x two decoder blocks' linears
          "conv": 3 * (58.454 + 3 * 0.467) * 1e9 * batch,
          "attention": (12 * 1.0192 + 3 * 2 * 0.68) * 1e9 * batch}
    out = {"method": "eager replay of the step's launch lists, HIP-event pair around every run of same-family launches (%d runs), empty-pair cost "
                     "(%.1f us) subtracted per run, median of %d passes; exemplar side lanes run inline here" % (len(runs), med(empties), passes),
           "us_per_step": {k: round(v, 1) for k, v in us.items()}}
    for k, g in gf.items():
        out[k] = {"achieved": g / (us[k] * 1e-6) / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s", "frac": g / (us[k] * 1e-6) / MFMA_BF16_PEAK}
    # algorithmic bytes of the GEMM families: every launch's operands once + its result once (split-K slabs, re-reads of an im2row
    # operand per column tile, fp32 partials are NOT algorithmic: they show up as traffic above this figure)
    alg = {"linear": 0, "conv": 0}
    nl = {"linear": 0, "conv": 0}
    flat = []
    for fn, args, keep in seq:
        if fn is L.countr_gemm_group and keep is not None:     # (items, n, dtype, modeA, modeB): n launches of one kind
            flat += [(L.countr_gemm, (None, args[2], args[3], args[4]), keep[i]) for i in range(args[1])]
            nl["linear"] -= args[1] - 1                        # ... counted as ONE launch
        else:
            flat.append((fn, args, keep))
    for fn, args, keep in flat:
        if fn is not L.countr_gemm or keep is None:
            continue
        f = fam(fn, args)
        es = 2 if args[1] == 1 else 4
        a = keep
        nb = max(a.nbatch, 1)
        if args[2] == 2:       # IM2ROW: the NHWC map once
            bytes_a = (a.M * a.Cin) * es
        elif args[3] == 3:     # (COL, IM2COL) wgrad: dy [P, Cout] and the map [P, Cin]
            bytes_a = a.K * a.M * es
        else:
            bytes_a = a.M * a.K * es * nb
        bytes_b = (a.K * a.Cin * es) if args[3] == 3 else a.N * a.K * es * nb
        bytes_c = a.M * a.N * ((2 if a.out_bf16 else 4) if not a.partial else 4) * nb
        extra = (a.M * a.N * 4 if a.resid else 0) + (a.M * a.N * 2 if (a.C2 and a.act == 1) else 0)
        alg[f] += bytes_a + bytes_b + bytes_c + extra
        nl[f] += 1
    traffic, tpath = None, None
    for rnd in ("r6", "r5", "r4"):
        try:
            tpath = os.path.join(ROOT, "profiles", "%s_family_traffic.json" % rnd)
            traffic = json.load(open(tpath))
            break
        except Exception:  # noqa: BLE001
            continue
    for k in ("linear", "conv"):
        out[k]["algorithmic_bytes"] = alg[k]
        out[k]["gemm_launches"] = nl[k]
        t = (traffic or {}).get(k) if batch == 8 else None
        out[k]["traffic"] = t["traffic_bytes_per_step"] if t else None
        out[k]["traffic_over_algorithmic"] = (t["traffic_bytes_per_step"] / alg[k]) if t else None
        out[k]["traffic_source"] = ("static: profiles/%s (sha256 %s, collected on HEAD %s%s; rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate "
                                    "passes over the eager step, tools/pmc_step.sh: per step, all launches of the family), not measured in this run"
                                    % (os.path.basename(tpath), _sha(tpath), traffic.get("collected_on_head") or "unrecorded (round 4 file)",
                                       " + uncommitted changes" if traffic.get("tree_dirty") else "")) if t else None
    return out


PRETRAIN_GF_PER_IMG = 3 * (288 * 12 * 2 * (12 * 768 * 768) + 12 * 4 * 288 * 288 * 768      # encoder on 288 kept tokens
                           + 576 * 8 * 2 * (12 * 512 * 512) + 8 * 4 * 576 * 576 * 512      # decoder on 576 tokens
                           + 2 * (288 * 768 * 768 + 288 * 768 * 512 + 576 * 512 * 768))    # embeds + pixel head; x3 = fwd+bwd


def time_pretrain(args, world, rank, dev, batch, steps, warmup):
    """MAE pretraining step (FSC_pretrain.py:254-301; mask_ratio 0.5): (seconds for `steps` steps, max over ranks; last loss)."""
    import models_mae_noct
    from countr_amd.trainer import PretrainStep
    torch.manual_seed(rank)
    model = models_mae_noct.__dict__["mae_vit_base_patch16"](norm_pix_loss=False, precision=args.precision)
    model.to(dev).train()
    step = PretrainStep(model, batch=batch, mask_ratio=0.5, lr=5e-6, weight_decay=0.05, use_graph=not args.no_graph)
    imgs = torch.rand(batch, 3, 384, 384, device=dev)

    def one():
        step.load(imgs)          # draws a fresh masking permutation (torch.rand + argsort) every step, as the reference
        return step.step()
    with step.on_stream():         # the loop runs on the step's stream (see the finetune loop below)
        for _ in range(max(warmup, 2)):
            one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with step.on_stream():
        for _ in range(steps):
            loss = one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt, loss.item()


def bench_pretrain(args, world, rank, dev):
    """MAE pretraining step (batch 8 per GPU by default): not the BASELINE.json metric -- selected explicitly with --workload pretrain."""
    B = args.batch
    dt, lv = time_pretrain(args, world, rank, dev, B, args.steps, args.warmup)
    if rank == 0:
        ips = world * B * args.steps / dt
        print(json.dumps({
            "metric": "images/sec (384x384, mask_ratio 0.5) MAE pretrain step", "value": ips, "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "MAE pretrain ViT-B/16 + 8x512-d decoder (models_mae_noct.mae_vit_base_patch16), batch=%d per GPU, "
                                   "random masking + fwd + all-patch MSE + full bwd + AdamW" % B,
                       "global_batch": world * B, "parallelism": "dp%d" % world, "hipgraph": not args.no_graph},
            "final_loss": lv, "step_tflops": PRETRAIN_GF_PER_IMG * ips / 1e12}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def time_infer(args, world, rank, dev, steps, warmup):
    """BASELINE.json configs[4]: zero-shot sliding-window inference (demo_zero.py:41-74) on 1920x1080 frames -- resized to
    384 x 672 they give 4 windows each, so 8 frames = one forward batch of 32 windows (shot_num = 0, shot_token path), then the
    per-column blend.  -> (seconds for `steps` passes over 8 frames, max over ranks; mean count)."""
    import models_mae_cross
    from countr_amd import inference
    from countr_amd.synthetic import wide_frames
    torch.manual_seed(0)
    model = models_mae_cross.__dict__["mae_vit_base_patch16"](norm_pix_loss=False, precision=args.precision)
    model.to(dev).eval()
    frames = wide_frames(8, 672, dev, seed=rank)
    empty = [torch.zeros(1, 0, device=dev)] * len(frames)

    def run(n):
        """n passes over the 8 frames, as a video stream would bring them: inference.density_maps_stream runs pass k + 1's frozen-encoder
        forward beside pass k's decoder / density head (bit-identical maps).  The first pass of a run computes its own encoder forward and
        the last one computes nothing ahead: n passes = n encoder forwards + n decoder / head passes + n blends, all inside the run."""
        cnt = None
        if args.no_pipeline:
            for _ in range(n):
                _dms, sums = inference.density_maps(model, frames, empty, 0, max_batch=32, return_sums=True)
                cnt = torch.stack(sums) / 60
            return cnt
        for _dms, sums in inference.density_maps_stream(model, ((frames, empty) for _ in range(n)), 0, max_batch=32, return_sums=True):
            cnt = torch.stack(sums) / 60
        return cnt
    run(max(warmup, 3))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cnt = run(steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    return dt, float(cnt.mean().item())


def bench_infer(args, world, rank, dev):
    """Replicas only: every rank counts its own frames, no collective.  Selected with --workload infer."""
    dt, mean_count = time_infer(args, world, rank, dev, args.steps, args.warmup)
    if rank == 0:
        ips = world * 8 * args.steps / dt
        print(json.dumps({
            "metric": "frames/sec (1920x1080 -> 384x672, zero-shot sliding window, 4 windows per frame)", "value": ips, "unit": "frames/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "zero-shot inference ViT-B/16 (mae_vit_base_patch16), 8 frames x 4 windows = batch 32 per GPU, "
                                   "forward + sliding-window blend + counts", "global_batch": world * 32, "parallelism": "replicas%d" % world,
                       "encoder_pipelining": "off" if args.no_pipeline else "on: pass k + 1's frozen-encoder forward beside pass k's decoder / density head (inference.density_maps_stream; bit-identical maps); every timed run computes all of its own encoder forwards"},
            "windows_per_sec": 4 * ips, "mean_count": mean_count, "fwd_tflops": 180.89e9 * 4 * ips / 1e12}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def other_workloads(args, dev, steps=10, warmup=3):
    """BASELINE configs[3] and [4] at their single-GPU shapes, timed by the SAME `python bench.py` invocation so that the driver's one run
    clocks all three single-GPU workloads: MAE pretraining step at 16 images per GPU and zero-shot inference on 8
    frames = 32 windows (C5); `steps` timed steps each behind `warmup` untimed ones.  Each runs in a fresh child process of this script
    (--workload ...) once the headline measurement has released the GPU: inside the parent's process the pretraining step measured
    anywhere between 10.8 and 16.1 ms depending on what the caching allocator handed it after the finetune plans (10.6 ms on its own)."""
    import subprocess
    out = {}

    def child(*extra):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--precision", args.precision] + list(extra)
        # a plain single-process run: nothing of the parent's launcher may leak in.  (torch.distributed.run also exports
        # TORCHELASTIC_USE_AGENT_STORE: with it a child that initialises a process group waits for the AGENT's store on a port nobody
        # serves -- the one-rank RCCL launch of tests/test_ddp_gpu.py hung here until its timeout.)
        drop = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE",
                "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME", "COUNTR_BENCH_INIT_PG", "COUNTR_FORCE_COMM", "COUNTR_GRAPH_COMM")
        env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_", "NCCL_ASYNC"))}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        except subprocess.TimeoutExpired:
            return {"error": "child process timed out after 300 s"}
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": (r.stderr or r.stdout)[-400:]}
        return json.loads(lines[-1])
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    d = child("--workload", "pretrain", "--batch", "16")
    out["pretrain"] = d if "error" in d else {
        "ms_per_step": d["ms_per_step"], "images_per_sec": d["value"], "batch": 16, "steps": steps, "final_loss": d["final_loss"],
        "step_tflops": d["step_tflops"],
        "workload": "MAE pretrain ViT-B/16 + 8x512-d decoder, mask_ratio 0.5: masking + fwd + all-patch MSE + full bwd + AdamW (child process)"}
    if args.precision == "bf16":
        # the headline step in the reference-numerics mode (precision="fp16": the reference's own autocast dtype, FSC_finetune_cross.py:273-275,
        # 286; same kernels built with fp16 operands) beside the bf16 headline
        d = child("--precision", "fp16", "--plain", "--reps", "3", "--steps", "30")
        out["finetune_fp16"] = d if "error" in d else {
            "ms_per_step": d["ms_per_step"], "images_per_sec": d["value"], "steps": 30,
            "workload": "the headline finetune step with precision='fp16' (libcountr_hip_f16.so; dynamic loss scale on the device -- GradScaler semantics, initial 2^16), child process"}
    d = child("--workload", "infer")
    out["infer"] = d if "error" in d else {
        "ms_per_32_windows": d["ms_per_step"], "frames_per_sec": d["value"], "windows_per_sec": d["windows_per_sec"], "steps": steps,
        "mean_count": d["mean_count"], "fwd_tflops": d["fwd_tflops"],
        "workload": "zero-shot sliding-window inference, 8 frames 1920x1080 -> 384x672 = one forward of 32 windows + blend + counts (child process)"}
    return out


class _Watchdog:
    """Multi-GPU runs replay RCCL all-reduces as nodes of the step's hipGraph -- a path no multi-GPU box has run here yet.  An exception
    there is handled by the trainer (every rank falls back to host-issued collectives); a HANG cannot be.  So the communicating bench
    first measures the step in the host-issued mode (plain RCCL calls between per-phase graphs), keeps that line, and runs everything
    behind it under this watchdog: no progress for `idle` seconds -> rank 0 prints the kept line (marked as the fallback), every rank
    leaves with os._exit(0).  kick() at every milestone; cancel() when the real line is about to be printed."""

    def __init__(self, idle, rank, line_fn):
        import threading
        self.idle, self.rank, self.line_fn = idle, rank, line_fn
        self.last, self.what = time.monotonic(), "start"
        self.done = threading.Event()
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def kick(self, what):
        self.last, self.what = time.monotonic(), what

    def cancel(self):
        self.done.set()

    def _run(self):
        while not self.done.wait(1.0):
            if time.monotonic() - self.last > self.idle:
                if self.rank == 0:
                    sys.stdout.write(json.dumps(self.line_fn(self.what)) + "\n")
                    sys.stdout.flush()
                sys.stderr.write("bench.py: rank %d made no progress for %d s in '%s' (captured RCCL collectives); leaving with the "
                                 "host-issued measurement\n" % (self.rank, self.idle, self.what))
                sys.stderr.flush()
                os._exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5, help="the timed block of --steps steps is repeated this often; `value` is the MEDIAN block")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="no cross-iteration pipelining of the frozen encoder (FinetuneStep(pipeline_encoder=False): "
                                                                "every step computes its own encoder forward before its decoder side)")
    ap.add_argument("--no-defer", action="store_true", help="optimizer update at the tail of its own step instead of beside the next step's "
                                                            "frozen-encoder forward (FinetuneStep(defer_optimizer=False))")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the timed step (it runs outside the timed regions)")
    ap.add_argument("--no-b32", action="store_true", help="skip roofline_b32 (profiling runs: keeps the attention kernel's launches of the "
                                                           "kernel-stats CSV at the one problem size of the step)")
    ap.add_argument("--plain", action="store_true", help="counter / trace runs: warm-up + the timed blocks only -- no shot-mix loop, no roofline "
                                                          "passes, no other workloads, no CPU baseline (every launch of the run belongs to a headline step)")
    ap.add_argument("--no-families", action="store_true", help="skip roofline_families (counter runs: no eager event-bracketed replay)")
    ap.add_argument("--no-other", action="store_true", help="skip other_workloads (pretrain / inference timed beside the headline)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="finetune only: batches start in pinned HOST memory (the DataLoader's hand-over) -> the PCIe-inclusive rate "
                         "quoted in DESIGN.md; never the headline value (inputs are HBM-resident there)")
    ap.add_argument("--workload", default="finetune", choices=["finetune", "pretrain", "infer"],
                    help="finetune = BASELINE.json metric (default); pretrain = MAE pretraining step (SURVEY 8f rank 3, config 4); "
                         "infer = zero-shot sliding-window inference, batch 32 windows (config 5)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only for single-GPU dry runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched like the N = 1 run (plain `python bench.py --gpus N`): re-exec under torch.distributed.run, one rank per GPU
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node == --gpus)" % (args.gpus, world))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1 or os.environ.get("COUNTR_BENCH_INIT_PG") == "1":   # (the last: one-rank RCCL dry run, tests/test_ddp_gpu.py)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)   # dry runs may oversubscribe one GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.workload == "pretrain":
        return bench_pretrain(args, world, rank, dev)
    if args.workload == "infer":
        return bench_infer(args, world, rank, dev)
    import models_mae_cross
    from countr_amd.trainer import FinetuneStep
    from countr_amd.synthetic import make_batch
    torch.manual_seed(0)
    model = models_mae_cross.__dict__["mae_vit_base_patch16"](norm_pix_loss=False, precision=args.precision)
    model.to(dev).train()
    B = args.batch
    step = FinetuneStep(model, batch=B, lr=1e-5, weight_decay=0.05, use_graph=not args.no_graph, process_group=None, mask_seed=1234 + rank,
                        defer_optimizer=not args.no_defer, pipeline_encoder=not args.no_pipeline)
    # device-resident synthetic batches (inputs are in HBM when the timed region starts); every timed step stages a batch into the
    # plan's input buffers (device-to-device) and draws a fresh Bernoulli(0.8) loss mask, as the reference loop does per iteration --
    # both inside the step's graph (its prologue kernel: trainer._Prologue)
    NB = 4
    batches = [make_batch(B, shots=3, seed=rank * NB + k, device=dev) for k in range(NB)]
    if args.host_inputs:
        batches = [tuple(t.cpu().pin_memory() if torch.is_tensor(t) else t for t in b) for b in batches]

    def one(k, S, more=True):
        imgs, boxes, gt, _ = batches[k % NB]
        # the loop runs on the step's stream (callers enter step.on_stream() around their loop), as everything does on ONE stream in
        # the reference's loop: from another stream every step pays two cross-queue hand-overs (inputs ready -> step, step done -> caller).
        # mask=None: a fresh Bernoulli(0.8) mask per step (FSC_finetune_cross.py:290-292), drawn by the step itself.
        # next_imgs: a training loop knows its next batch (the DataLoader has it ready): pipeline_encoder runs that batch's frozen-encoder
        # forward beside this batch's decoder side.  `more` is False for the LAST step of a timed block: nothing is computed ahead across
        # the end of a timed region (K timed steps = K encoder forwards + K decoder-side passes + K updates, all inside it).
        step.load(imgs, boxes, gt, None, S, next_imgs=batches[(k + 1) % NB][0] if more else None)
        return step.step(S)

    host_dt = [0.0]

    def timed(shots):
        step.drop_lookahead()      # pipeline_encoder: no encoder forward computed OUTSIDE this timed region is used inside it
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nh = min(len(shots), 16)
        with step.on_stream():
            for k, S in enumerate(shots):
                sums = one(k, S, more=k + 1 < len(shots))
                if k + 1 == nh:
                    # the host's own time per step, from the block's first 16 enqueues: further in, a host that runs ahead blocks on the
                    # full hardware queue (~40 graphs of ~205 nodes) and its loop time converges to the GPU's
                    host_dt[0] = (time.perf_counter() - t0) * len(shots) / nh
            step.flush()        # defer_optimizer: the last step's update is applied INSIDE the timed region (K steps = K optimizer updates)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, sums

    wd = None
    if step.sync.comm and getattr(step.sync, "capturable", False) and not args.no_graph:
        step.sync.capturable = False
        with step.on_stream():
            for k in range(max(args.warmup, 2)):
                one(k, 3)
        dt_safe, sums_safe = timed([3] * args.steps)
        loss_safe = sums_safe[0].item()
        step.sync.capturable = True

        def safe_line(where, dt_safe=dt_safe, loss_safe=loss_safe):
            return {"metric": "images/sec (384x384, 3 exemplars) FSC147 finetune step", "value": world * B * args.steps / dt_safe,
                    "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_safe / args.steps,
                    "timed_blocks": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
                    "data": "synthetic",
                    "config": {"workload": "FSC147 finetune ViT-B/16 (mae_vit_base_patch16), batch=%d per GPU, 384x384, shot_num=3, "
                                           "frozen encoder fwd + decoder fwd/bwd + masked-MSE + AdamW" % B,
                               "global_batch": world * B, "parallelism": "dp%d" % world, "hipgraph": True,
                               "optimizer_update": "at the tail of its step"},
                    "final_loss": loss_safe, "parity_checked": False, "inputs": "resident in HBM",
                    "step_tflops": GF_STEP_PER_IMG * world * B * args.steps / dt_safe / 1e12,
                    "multi_gpu": {"collectives": "issued by the host between per-phase graph replays",
                                  "fallback": "the captured-collective form (RCCL all-reduces as nodes of the step's hipGraph) made no "
                                              "progress for %d s in '%s'; this line is the host-issued measurement taken before it" % (wd.idle, where)}}
        wd = _Watchdog(int(os.environ.get("COUNTR_BENCH_WATCHDOG_S", "240")), rank, safe_line)
    kick = (lambda what: wd.kick(what)) if wd is not None else (lambda what: None)
    kick("warm-up: capture of the communicating step")
    with step.on_stream():
        nw = max(args.warmup, 2)               # the first steps build the plan and capture the graphs (pipeline_encoder: the forms of a step
        for k in range(nw):                    # -- first of a loop, middle (from --warmup 3 on), last -- so that no timed block captures)
            one(k, 3, more=k + 1 < nw)
    torch.cuda.synchronize()
    if os.environ.get("COUNTR_BENCH_FAKE_HANG") == "1" and wd is not None:      # (tests/test_ddp_gpu.py: the watchdog's exit path)
        time.sleep(3600)
    kick("parity check")
    # (after the warm-up: the checked step REPLAYS the captured graph the timed blocks replay)
    parity = None if (args.plain or args.no_parity) else parity_check(model, step, world, rank, B, NB, dev, kick=kick)
    step.sync.profile = world > 1 or step.sync.comm
    # box-to-box and run-to-run spread (5.3-5.55 ms over the boxes of round 2) is larger than most single optimisations: the block of
    # --steps steps is timed --reps times (each bracketed by barrier + synchronize, max over ranks) and the MEDIAN block is the value
    kick("timed blocks")
    blocks = []
    for _ in range(max(args.reps, 1)):
        blocks.append(timed([3] * args.steps))
        kick("timed blocks")
    host_ms = 1e3 * host_dt[0] / args.steps
    dts = sorted(b[0] for b in blocks)
    dt, sums = dts[len(dts) // 2], blocks[-1][1]
    exposed = step.sync.exposed_us() if step.sync.profile else None
    comm_mode, host_issued = None, None
    if step.sync.comm:
        comm_mode = "captured" if getattr(step.sync, "capturable", False) else "host-issued"
        if comm_mode == "captured":
            # the headline blocks replayed ONE graph per step with the RCCL all-reduces as nodes of it: nothing to bracket.  The exposed
            # communication is measured on a short extra run in the per-phase-graph mode (host-issued collectives between the phases),
            # whose step time is reported beside it
            kick("host-issued mode: exposed communication")
            step.sync.capturable = False
            with step.on_stream():
                for k in range(3):
                    one(k, 3)
            torch.cuda.synchronize()
            step.sync.exposed_us()          # (drops the warm-up steps' event pairs)
            dt_h, _ = timed([3] * min(args.steps, 20))
            exposed = step.sync.exposed_us()
            host_issued = 1e3 * dt_h / min(args.steps, 20)
            step.sync.capturable = True
    # the reference draws shot_num uniformly from 0..3 per iteration (FSC_finetune_cross.py:276-284): same loop on that mix,
    # reported beside the headline (shot_num = 3) number
    if args.plain:
        if rank == 0:
            print(json.dumps({"metric": "images/sec (384x384, 3 exemplars) FSC147 finetune step", "value": world * B * args.steps / dt, "unit": "images/sec",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "plain": True, "host_enqueue_ms_per_step": host_ms,
                              "hipgraph": not args.no_graph, "dtype": args.precision, "data": "synthetic"}))
        if wd is not None:
            wd.cancel()
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    from countr_amd.parallel import shared_shot_num
    kick("shot mix: capture of the other shot counts")
    mix = [shared_shot_num(i, seed=0) for i in range(args.steps)]
    with step.on_stream():
        for S in sorted(set(mix) | {0, 1, 2}):
            # build / capture the plans of the other shot counts outside the timed region (pipeline_encoder: the three forms of a step --
            # own encoder forward + look-ahead, look-ahead only, no look-ahead -- are three graphs per shot count)
            one(0, S); one(1, S); one(2, S, more=False)
    dt_mix, _ = timed(mix)
    loss = sums[0].item()
    ranks_seen = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "device": local, "name": props.name, "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        ranks_seen = gathered
    if rank != 0 and wd is not None:
        wd.cancel()          # (this rank's GPU work is over: what is left is rank 0's own roofline passes, then the closing barrier)
    kick("roofline passes (rank 0)")
    if rank == 0:
        ips = world * B * args.steps / dt
        line = {
            "metric": "images/sec (384x384, 3 exemplars) FSC147 finetune step", "value": ips, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "timed_blocks": len(dts), "ms_per_step_min": 1e3 * dts[0] / args.steps, "ms_per_step_max": 1e3 * dts[-1] / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "FSC147 finetune ViT-B/16 (mae_vit_base_patch16), batch=%d per GPU, 384x384, shot_num=3, "
                                   "frozen encoder fwd + decoder fwd/bwd + masked-MSE + AdamW" % B,
                       "global_batch": world * B, "parallelism": "dp%d" % world, "hipgraph": not args.no_graph,
                       "optimizer_update": ("deferred: AdamW of step k runs beside the frozen-encoder forward of step k + 1 (bit-identical "
                                            "parameters; the last update of a timed block is flushed inside it)")
                                           if (step.defer and step.use_graph and (not step.sync.comm or step.sync.capturable)) else "at the tail of its step",
                       "encoder_pipelining": ("on: the frozen-encoder forward of batch k + 1 runs on its own lane of step k's graph beside batch k's "
                                              "decoder forward / loss / backward / AdamW (bit-identical results); the first step of every timed block "
                                              "computes its own encoder forward first and the last one computes nothing ahead, so a block of K steps "
                                              "contains exactly K encoder forwards, K decoder-side passes and K updates")
                                             if step._pipe_ok() else "off"},
            "final_loss": loss, "host_enqueue_ms_per_step": host_ms,
            "parity_checked": bool(parity and parity["checked"]), "parity": parity,
            "inputs": "pinned host memory, copied over PCIe every step" if args.host_inputs else "resident in HBM",
            "step_tflops": GF_STEP_PER_IMG * ips / 1e12,
            "timed_region": ("per step (ONE hipGraph replay at N = 1): device-to-device staging of a batch + fresh loss mask + fwd + loss + decoder bwd + "
                             "grad all-reduce + AdamW" + ("; encoder_pipelining: the graph of step k holds batch k + 1's frozen-encoder forward (own lane) "
                             "beside batch k's decoder side -- a timed block of K steps holds exactly K encoder forwards" if step._pipe_ok() else "")),
            "shot_mix": {"images_per_sec": world * B * args.steps / dt_mix, "ms_per_step": 1e3 * dt_mix / args.steps,
                         "shot_nums": "uniform 0..3 per step (%s)" % "".join(str(x) for x in mix[:32])},
        }
        line["roofline"] = attention_roofline(model, B)
        line["roofline_step"] = {"bound": "mfma", "achieved": GF_STEP_PER_IMG * ips / world / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                                 "frac": GF_STEP_PER_IMG * ips / world / MFMA_BF16_PEAK,
                                 "note": "whole step per GPU: 321.16 GF algorithmic per image (SURVEY 8d) x images/s; the sustained full-chip matrix peak "
                                         "is 2.0 PF (1.92 GHz under load), and the GEMM family is bound by the LDS-DMA path (~42 B/clk per CU), DESIGN.md"}
        if world == 1:
            if not args.no_families:
                line["roofline_families"] = family_breakdown(model, step, B)
            if not args.no_b32:
                # BASELINE configs[4]: the same kernel at 32 windows (zero-shot inference plan), amortising the per-launch fixed cost
                model.eval()
                with torch.no_grad():
                    model(torch.rand(32, 3, 384, 384, device=dev), torch.zeros(32, 0, device=dev), 0)
                r32 = attention_roofline(model, 32, shot=0, train=False)
                line["roofline_b32"] = {k: r32[k] for k in ("bound", "achieved", "peak", "unit", "frac", "us_per_launch", "us_per_launch_back_to_back")}
                model.train()
        else:
            line["multi_gpu"] = {"ranks_seen": ranks_seen, "bucket_bytes": step.sync.bucket_bytes(),
                                 "collectives": ("RCCL all-reduces captured as nodes of the step's hipGraph (side stream between the backward phases): "
                                                 "one graph replay per step" if comm_mode == "captured" else
                                                 "issued by the host between per-phase graph replays"),
                                 "ms_per_step_host_issued": host_issued,
                                 "exposed_comm_us": exposed,
                                 "note": "buckets in backward-completion order (head | decoder blocks | exemplar CNN | shot_token); every bucket but "
                                         "the last is all-reduced on a side stream under the next backward phase; exposed_comm_us = median time the "
                                         "step stream waited in GradSync.finish() (measured in the host-issued mode: event pairs cannot be captured)"}
        if world == 1 and not args.no_other and not args.no_graph:
            del step
            line["other_workloads"] = other_workloads(args, dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        if wd is not None:
            wd.cancel()
            line["multi_gpu_safe_mode"] = {"ms_per_step_host_issued_first": 1e3 * dt_safe / args.steps,
                                           "note": "measured before the captured-collective form was tried (bench.py::_Watchdog)"}
        print(json.dumps(line))
        sys.stdout.flush()
    if wd is not None:
        wd.cancel()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
