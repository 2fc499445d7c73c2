"""Finetune step for the HIP engine: forward, masked-MSE loss, decoder-side backward, gradient all-reduce over
RCCL (one flat fp32 buffer in two buckets, the first launched while the rest of backward runs) and fused AdamW.

Mirrors the hot loop of the reference (FSC_finetune_cross.py:265-319; util/misc.py:266-280) minus its three
per-step host syncs; bf16 needs no GradScaler.  With use_graph=True each phase is captured once per shot_num into a
hipGraph and replayed (torch.cuda.CUDAGraph is only the capture/replay handle; every node is one of our kernels).
"""

import ctypes as C
import os

import torch

from . import _lib
from .parallel import GradSync


class _Prologue:
    """Host side of countr_step_prologue (include/countr_hip.h): the per-iteration hand-over of an optimisation step -- staging copies
    of the batch, the Bernoulli loss mask (FSC_finetune_cross.py:290-292) and the AdamW scalars -- as the FIRST NODE of the step's
    captured graph, so that nothing is launched between two graph replays (round 4: a torch mask draw + a staging launch sat in
    ~160 us of idle GPU there).  A captured node's arguments are frozen; what changes per step goes through a ring of 256-byte records
    in pinned host memory.  Execution k (eager run or replay) reads record k % SLOTS -- the kernel counts its own executions on the
    device, `seq` mirrors that count here -- so fill() writes record seq % SLOTS once the execution that last read it has finished
    (an event per slot, SLOTS steps old: never a wait in practice)."""
    SLOTS = 16

    def __init__(self, eng, seed=0):
        import numpy as np
        self.eng = eng
        L = eng.L
        self.rb = int(L.countr_step_prologue_record_bytes())
        self.nblk = int(L.countr_step_prologue_copy_blocks())
        assert self.rb == 256
        self.ring = torch.zeros(self.SLOTS * self.rb, dtype=torch.uint8).pin_memory()
        self.view = self.ring.numpy()
        self.counter = torch.zeros(2 + 32, dtype=torch.int64, device=eng.device)      # {executions, reserved, record copy}
        self.seq = 0
        self.events = [None] * self.SLOTS
        self.hyper = np.zeros(8, np.float32)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.draws = 0                      # masks drawn so far = the Philox counter's high words (checkpointable)
        self.keep = ()

    def fill(self, pairs, draw_mask, p_keep=0.8, keep=()):
        """pairs: [(src ptr or None, dst ptr, bytes)], at most 6, 16-byte aligned.  Writes the record the NEXT execution reads."""
        import numpy as np
        k = self.seq % self.SLOTS
        ev = self.events[k]
        if ev is not None:
            ev.synchronize()
        n = len(pairs)
        assert n <= 6
        rec = self.view[k * self.rb:(k + 1) * self.rb]
        u64 = rec[0:144].view(np.uint64)
        i32 = rec[144:176].view(np.int32)
        u32 = rec[176:192].view(np.uint32)
        tot = sum(b for _s, _d, b in pairs) or 1
        first, used = [], 0
        for i, (_s, _d, b) in enumerate(pairs):          # copy blocks in proportion to the bytes, at least one each
            first.append(used)
            used += max(1, min(self.nblk - used - (n - 1 - i), int(round(self.nblk * b / tot))))
        for i in range(6):
            s_, d_, b_ = pairs[i] if i < n else (0, 0, 0)
            u64[i], u64[6 + i], u64[12 + i] = (s_ or 0), d_, b_ // 16
            i32[i] = first[i] if i < n else self.nblk
        i32[6], i32[7] = n, int(bool(draw_mask))
        u32[0], u32[1] = self.seed & 0xFFFFFFFF, self.seed >> 32
        u32[2], u32[3] = self.draws & 0xFFFFFFFF, (self.draws >> 32) & 0xFFFFFFFF
        rec[192:224].view(np.float32)[:] = self.hyper
        rec[224:228].view(np.uint32)[0] = min(int(p_keep * 4294967296.0), 0xFFFFFFFF)
        if draw_mask:
            self.draws += 1
        self.keep = keep

    def launch(self, mask):
        """Enqueue (or capture) the prologue on the engine's current stream.  mask: the fp32 mask tensor the record may ask to draw."""
        eng = self.eng
        _lib.check(eng.L.countr_step_prologue(self.ring.data_ptr(), self.SLOTS, self.counter.data_ptr(), eng.hyper.data_ptr(),
                                              mask.data_ptr() if mask is not None else None, mask.numel() if mask is not None else 0,
                                              eng._stream()), "countr_step_prologue")

    def resync(self):
        """After an exception between fill() and executed(): the prologue may or may not have run on the device, so `seq` is re-read from
        the device's own execution count -- otherwise every later step would fetch a stale ring record (old source pointers, lr and bias
        corrections) without any error.  Best effort (the device may be unusable after a kernel fault); host sync."""
        try:
            torch.cuda.synchronize(self.eng.device)
            self.seq = int(self.counter[0].item())
            self.events = [None] * self.SLOTS
        except Exception:  # noqa: BLE001 -- the original exception is the one to report
            pass
        self.keep = ()

    def executed(self, stream):
        """One execution (eager or replayed) of a phase that holds the prologue has been enqueued on `stream`."""
        k = self.seq % self.SLOTS
        if self.events[k] is None:
            self.events[k] = torch.cuda.Event()
        self.events[k].record(stream)
        self.seq += 1
        for t in self.keep:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)
        self.keep = ()


class _GraphStep:
    """Shared machinery of the optimisation steps: dedicated stream (`self.stream`: load() makes it wait for the caller's current
    stream and step() makes the caller's stream wait for it -- a caller that runs its loop under `torch.cuda.stream(step.stream)`,
    as bench.py and the CLIs do, turns both into same-queue no-ops and saves ~50 us of idle GPU per step), per-phase hipGraph capture/replay, bucketed gradient
    all-reduce behind the backward phases, host-batch staging on a copy stream, device-side AdamW scalars."""

    def __init__(self, model, batch, lr, weight_decay, betas, eps, use_graph, process_group, accum_iter=1, mask_seed=0):
        self.model = model
        self.accum = int(accum_iter)   # micro-steps per optimisation step (reference --accum_iter: loss / accum_iter, update every
        self._micro = 0                # accum_iter-th iteration -- FSC_finetune_cross.py:300-305); gradients accumulate in eng.G
        self._touched = set()          # conditional buckets that received a gradient in the current accumulation window
        self.per_rank = False          # FinetuneStep(per_rank_shot=True): every rank draws its own shot_num (reference semantics)
        self._touched_any = set()      # ... then: conditional buckets that received a gradient on ANY rank in this window
        self._pending = None           # staging copies of load() not launched yet: (src ptrs, dst ptrs, sizes, tensors kept alive)
        self.eng = model._engine()
        self.B = batch
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.use_graph = use_graph
        self.pg = process_group
        self.graphs = {}
        # all work of the step (input copies, graph replays, collectives' producers) runs on ONE dedicated non-default
        # stream; sources are record_stream()'ed so the caching allocator cannot recycle them under a pending copy.
        # (Captured regions contain kernels only: a hipMemsetAsync node + float atomics gave wrong sums under graph
        # replay on ROCm 7.2 (round 1) -- so the loss reduction is a deterministic two-stage kernel pair.)
        self.stream = torch.cuda.Stream(device=self.eng.device)
        self._gen = self.eng.generation
        self.sums = {}
        lay = self.eng.layout
        self.bucket0 = lay.bucket_range(0)
        self.bucket_rest = (self.bucket0[1], lay.n_train)
        self.sync = self._make_sync(process_group)
        self.world = self.sync.world
        self.pro = _Prologue(self.eng, mask_seed)
        self._draw_mask = False
        self.defer = False             # FinetuneStep(defer_optimizer=True): AdamW of step k runs at the head of step k + 1's graph
        self._pc = None                # ... the (skip, zero) key of the update that is still pending
        self._pc_hyper = None          # ... and its scalars {lr, bias corrections, grad_scale}
        self._cur_pc = None
        self.pipe = False              # FinetuneStep(pipeline_encoder=True): the next batch's frozen-encoder forward beside this batch's decoder side
        self._pipe_mode = "plain"      # ... what the step that load() prepared will run: plain | coldnext | steady | last
        self._cur_pipe = "plain"       # ... and what the step being executed / captured runs (part of the graph key)
        self._pipe_ready = None        # ... the images tensor whose latent is waiting in the shared buffer (identity, not contents)
        self._pipe_next = None
        self._pipe_gen = -1
        self._pipe_tok = None
        self.grad_scale = 1.0 / self.accum
        # fp16 mode: torch.cuda.amp.GradScaler as the reference uses it (util/misc.py:260-286; GradScaler() defaults: scale 65536, growth
        # x 2 every 2000 clean steps, x 0.5 and NO optimizer step on a non-finite gradient), kept on the device so that the captured step
        # needs no host decision: amp = {scale, clean steps, found_inf, skipped steps, growth interval}.  The loss kernel multiplies dL/dout
        # by amp[0] (it is ~1e-9 per pixel: below fp16's normal range), the fused AdamW checks the (all-reduced) gradient, skips or
        # unscales, and updates the scale (countr_*_amp).  One deviation: the host-side bias-correction counters also advance on a
        # skipped step (torch's do not) -- they differ by the number of overflows so far (`skipped_steps()`).  bf16 / fp32: no scaling.
        self.amp = (torch.tensor([65536.0, 0.0, 0.0, 0.0, 2000.0, 0.0, 0.0, 0.0], device=self.eng.device, dtype=torch.float32)
                    if self.eng.precision == "fp16" else None)

    def _make_sync(self, process_group):
        return GradSync(self.eng.G, self.bucket0, self.bucket_rest, process_group)

    def on_stream(self):
        """Context manager for a training loop that wants all of its device work on the step's stream (no cross-queue hand-over per
        step): on entry the step's stream waits for everything the caller's stream holds (model build, weight casts, LayerNorm-fold
        repack, accumulators allocated outside), inside `torch.cuda.current_stream()` IS the step's stream, on exit the caller's stream
        waits for the step's.  (A bare `with torch.cuda.stream(step.stream)` lacks the first edge: load()'s wait is then a self-wait.)"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            outer = torch.cuda.current_stream(self.eng.device)
            self.stream.wait_stream(outer)
            with torch.cuda.stream(self.stream):
                yield self.stream
            outer.wait_stream(self.stream)
        return cm()

    def _to_device(self, tensors):
        """Host batches (the DataLoader's pinned tensors) go host -> device on a dedicated copy stream into staging buffers, so the
        PCIe transfer of batch t+1 overlaps the compute of step t; the step stream then only does device-to-device copies into
        the plan's input buffers.  Device tensors pass through.  Returns device tensors valid on the step's stream."""
        if all(t.is_cuda for t in tensors):
            return tensors
        if not hasattr(self, "_copy_stream"):
            self._copy_stream = torch.cuda.Stream(device=self.eng.device)
            self._staging = {}
            self._staged_ev = torch.cuda.Event()
            self._stage_slot = 0
            self._consumed_evs = [None] * self.STAGING_SLOTS
        self._stage_slot = slot = (self._stage_slot + 1) % self.STAGING_SLOTS
        out = []
        if self._consumed_evs[slot] is not None:
            self._consumed_evs[slot].synchronize()    # the batch that last used this slot has been copied out of it (see STAGING_SLOTS)
        with torch.cuda.stream(self._copy_stream):
            for i, t in enumerate(tensors):
                if t.is_cuda:
                    out.append(t)
                    continue
                key = (slot, i, tuple(t.shape), t.dtype)
                if key not in self._staging:
                    self._staging[key] = torch.empty(t.shape, dtype=t.dtype, device=self.eng.device)
                self._staging[key].copy_(t, non_blocking=True)
                out.append(self._staging[key])
            self._staged_ev.record(self._copy_stream)
        self.stream.wait_event(self._staged_ev)
        return out

    # Staging slots are used round-robin and the HOST waits for the slot's last consumer (an event STAGING_SLOTS steps old: complete
    # unless the host runs that far ahead).  The copy stream itself waits for nothing: with a stream-side wait on the step stream's
    # "consumed" event (one slot, the natural formulation) the H2D copy of batch t+1 no longer overlapped step t on ROCm 7.2 --
    # +0.4 ms of exposed PCIe time per step (tools/host_async.py host: 5.8 ms against 5.4 with either wait removed).
    STAGING_SLOTS = 3

    def _staging_consumed(self):
        if hasattr(self, "_copy_stream"):
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._consumed_evs[self._stage_slot] = ev

    def _phases(self, key):     # [(name, launcher, graph key)]: forward + loss + first backward part, then the remaining backward parts
        raise NotImplementedError

    def _run_phases_merged(self, phases):
        """All phases of a (micro-)step back to back (one rank: nothing to exchange between them)."""
        for _name, fn, gkey in phases:
            fn(gkey)

    def _comm_skip(self, touched):
        """Gradient buckets that are not all-reduced in this step (nobody has a gradient for them)."""
        return ()

    def _pipe_ok(self):
        """The pipelined encoder needs the whole step as ONE graph (one rank, or RCCL with captured collectives): the lane is joined
        inside the graph that forked it."""
        return self.pipe and self.use_graph and (not self.sync.comm or self.sync.capturable) and self.eng.code != 0

    def _adam_sets(self, touched):
        """(skip, zero): buckets the optimizer skips (never had a gradient) / steps with a zero gradient."""
        return (), ()

    def _window_sets(self):
        """(touched, zero_fill) at the end of an accumulation window.  touched: conditional buckets with a gradient somewhere -- they
        are all-reduced and stepped.  zero_fill: the ones among them THIS rank has no gradient for (per-rank shot_num only): DDP with
        find_unused_parameters=True (FSC_finetune_cross.py:230) lets such a rank contribute zeros, so its range of the flat gradient is
        zero-filled in front of the all-reduce."""
        if not self.per_rank:
            return frozenset(self._touched), ()
        touched = frozenset(self._touched_any | self._touched)
        return touched, tuple(sorted(touched - self._touched))

    def _zero_buckets(self, buckets):
        """Zero-fill the gradient ranges of `buckets` (one countr_copy_multi launch with NULL sources: a kernel node when captured)."""
        rng = [self.sync.buckets[b] for b in buckets if self.sync.buckets[b][1] > self.sync.buckets[b][0]]
        if not rng:
            return
        n = len(rng)
        vp = C.c_void_p * n
        base = self.eng.G.data_ptr()
        _lib.check(self.eng.L.countr_copy_multi(n, vp(*[None] * n), vp(*[base + 4 * s0 for s0, _e in rng]),
                                                (C.c_int64 * n)(*[4 * (e0 - s0) for s0, e0 in rng]), self.eng._stream()), "copy_multi(zero)")

    def _lists(self, plan, acc):
        """The plan's backward launch lists that overwrite (first micro-step) or accumulate into (later ones) eng.G."""
        return plan.acc if acc else plan

    def _phase_c(self, key, stream=None):
        skip, zero = key
        self.eng.adamw_launch(1, self.wd, self.betas, self.eps, hyper_dev=self.eng.hyper, skip=skip, zero=zero, gnorm=True, stream=stream, amp=self.amp)

    @property
    def loss_scale(self):
        """The factor the flat gradient buffer currently carries (fp16 mode: the dynamic loss scale, read back from the device -- a host
        sync; 1 otherwise)."""
        return float(self.amp[0].item()) if self.amp is not None else 1.0

    def skipped_steps(self):
        """fp16 mode: optimizer steps skipped so far because a gradient was not finite (GradScaler semantics); host sync."""
        return int(self.amp[3].item()) if self.amp is not None else 0

    def scaler_state(self):
        """fp16 mode: the checkpoint's 'scaler' entry -- a torch.cuda.amp.GradScaler.state_dict() of the device-side scaler (the reference
        saves loss_scaler.state_dict() and reloads it under --do_resume: util/misc.py:316, :419), so a resumed run continues at the scale
        and growth count it stopped at and the file loads into the reference's GradScaler.  None in bf16 / fp32 (no scaler: the key is
        omitted -- GradScaler.load_state_dict({}) raises).  Host sync."""
        if self.amp is None:
            return None
        self.flush()
        a = self.amp.cpu().tolist()
        return {"scale": float(a[0]), "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": int(a[4]), "_growth_tracker": int(a[1])}

    def load_scaler_state(self, sd):
        """Restores a 'scaler' entry written by scaler_state() or by the reference's GradScaler (same keys).  Only the factors the device
        kernel implements (x 2 growth, x 0.5 backoff) are accepted.  A no-op outside fp16 mode or for an empty / missing entry."""
        if self.amp is None or not sd:
            return False
        if float(sd.get("growth_factor", 2.0)) != 2.0 or float(sd.get("backoff_factor", 0.5)) != 0.5:
            raise ValueError("GradScaler state with growth_factor %r / backoff_factor %r: the device-side scaler implements 2.0 / 0.5"
                             % (sd.get("growth_factor"), sd.get("backoff_factor")))
        self.flush()
        with torch.cuda.stream(self.stream):
            self.amp[0] = float(sd["scale"])
            self.amp[1] = float(int(sd.get("_growth_tracker", 0)))
            self.amp[2] = 0.0
            self.amp[4] = float(int(sd.get("growth_interval", 2000)))
        torch.cuda.current_stream(self.eng.device).wait_stream(self.stream)
        return True

    def drop_lookahead(self):
        """pipeline_encoder: forget the encoder output computed ahead for the next batch (the next step then computes its own).  For
        callers that must not carry work across a boundary -- a benchmark's timed region, a change of the input pipeline."""
        self._pipe_ready = None

    def flush(self):
        """defer_optimizer mode: apply the optimizer update that is still pending (the last step's), so that the parameters, the AdamW
        state and grad_norm() are those of a step that has completed.  A no-op otherwise."""
        if self._pc is None:
            return
        with torch.cuda.stream(self.stream):
            keep = self.pro.hyper.copy()
            self.pro.hyper[:] = self._pc_hyper
            self.pro.fill([], False)
            self.pro.hyper[:] = keep
            try:
                self._prologue_launch()            # (eager: the scalars reach eng.hyper)
                self._phase_c(self._pc)
            except BaseException:
                self.pro.resync()
                raise
            self.pro.executed(self.stream)
        torch.cuda.current_stream(self.eng.device).wait_stream(self.stream)
        self._pc = self._pc_hyper = None

    def _run_phase(self, name, fn, S):
        """S: hashable argument of the phase; (name, S) identifies its captured graph."""
        if not self.use_graph:
            fn(S)
            return
        if self._gen != self.eng.generation:   # the engine re-planned (bigger batch elsewhere): old graphs are stale
            self.graphs.clear()
            self._gen = self.eng.generation
        key = (name, S)
        g = self.graphs.get(key)
        if g is None:
            fn(S)  # first use: run eagerly (does the real work, triggers lazy kernel attributes and plan builds) ...
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):  # ... then capture the same launches for every later step
                fn(S)
            self.graphs[key] = g
            return
        g.replay()

    def _run_captured_comm(self, gk, whole):
        """One communicating step as ONE graph (RCCL collectives captured with the phases).  A new key runs EAGERLY first -- outside
        any handler: a kernel-launch, memory or RCCL error of the step itself propagates -- and is then captured.  Only a failure of
        that CAPTURE (an RCCL build whose collectives cannot become graph nodes) selects the host-issued form, and only at the first
        capture this step object ever attempts, where every rank is in the same place: the ranks agree on the outcome with a MIN
        all-reduce of a flag before any of them flips GradSync.capturable, so no rank replays captured collectives against another
        rank's host-issued ones.  Later capture failures are real errors and raise."""
        if self._gen != self.eng.generation:
            self.graphs.clear()
            self._gen = self.eng.generation
        g = self.graphs.get(gk)
        if g is not None:
            g.replay()
            return
        whole(gk[1])                      # the step itself (eager); errors propagate
        torch.cuda.synchronize()
        first = not getattr(self, "_capture_agreed", False)
        g, err = None, None
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                whole(gk[1])
        except Exception as e:  # noqa: BLE001
            if not first:
                raise
            g, err = None, e
        if first:
            import torch.distributed as dist
            ok = torch.tensor([0 if g is None else 1], dtype=torch.int32, device=self.eng.device)
            torch.cuda.synchronize()
            if self.sync.comm:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.pg)
            self._capture_agreed = True
            if int(ok.item()) == 0:
                import warnings
                warnings.warn("countr_amd: capturing the RCCL all-reduces into the step graph failed on some rank (%r here); every rank "
                              "falls back to host-issued collectives between per-phase graphs (COUNTR_GRAPH_COMM=0)" % (err,))
                self.sync.capturable = False
                self.sync._started = set()
                return
        self.graphs[gk] = g

    def _upload_hyper(self, skip=()):
        """Step-dependent AdamW scalars travel in the prologue's record (-> eng.hyper on the device) so that graph replay sees new
        values.  Bias corrections are per counter group (torch.optim.AdamW counts steps per parameter): group 0 steps always, group 1
        (exemplar CNN, bucket 2) and group 2 (shot_token, bucket 3) only once they have had a gradient."""
        eng = self.eng
        eng.step_count += 1
        eng.group_steps[0] += 1
        for bucket, grp in ((2, 1), (3, 2)):
            if bucket not in skip:
                eng.group_steps[grp] += 1
        h = self.pro.hyper
        h[0] = self.lr
        h[3] = self.grad_scale / self.world          # (fp16 mode: the AdamW kernel divides by the device-side loss scale itself)
        for grp, (i1, i2) in enumerate(((1, 2), (4, 5), (6, 7))):
            t = max(eng.group_steps[grp], 1)
            h[i1] = 1.0 - self.betas[0] ** t
            h[i2] = 1.0 - self.betas[1] ** t

    def _prologue_mask(self):
        """The mask buffer the prologue may draw into (None: this step has no loss mask)."""
        return None

    def _prologue_launch(self):
        self.pro.launch(self._prologue_mask())

    # ------------------------------------------------------------------ optimizer state (checkpoint 'optimizer' entry)
    # The checkpoint's 'optimizer' entry is a torch.optim.AdamW state_dict in the REFERENCE's parameter order (util/misc.py:312-318
    # saves optimizer.state_dict(); :400-421 loads it under --do_resume), so checkpoints move between the two code bases in both
    # directions: timm's add_weight_decay (FSC_finetune_cross.py:234, FSC_pretrain.py:226) puts every requires_grad parameter that is
    # 1-D or a bias into group 0 and the rest into group 1, each in named_parameters() order -- INCLUDING the encoder, whose
    # parameters sit in the groups but never get a gradient (forward_encoder runs under no_grad) and therefore never get state.
    FROZEN = ("pos_embed", "decoder_pos_embed")          # requires_grad=False in the reference (models_mae_cross.py:30,42)

    def _torch_param_order(self):
        from .engine import no_weight_decay
        shapes = self.eng.layout.shapes
        names = [n for n, _ in self.model.named_parameters() if n not in self.FROZEN]
        nd = [n for n in names if no_weight_decay(n, shapes[n])]
        dc = [n for n in names if not no_weight_decay(n, shapes[n])]
        return nd, dc

    def _bucket_of(self, name):
        lay = self.eng.layout
        o = lay.off[name] - lay.train_start
        for (bucket, _nodecay), s0, e0 in lay.segments:
            if s0 <= o < e0:
                return bucket
        raise KeyError(name)

    def _group_of_bucket(self, bucket):
        """AdamW step-counter group of a gradient bucket (engine.ADAM_GROUP_OF_BUCKET for finetuning; one group otherwise)."""
        return 0

    def _conditional_buckets(self):
        return ()

    def optimizer_state(self):
        """torch.optim.AdamW.state_dict() of the equivalent reference optimizer: per-parameter {'step', 'exp_avg', 'exp_avg_sq'} for
        every parameter that has had a gradient, two param_groups in timm's add_weight_decay order."""
        self.flush()
        eng = self.eng
        nd, dc = self._torch_param_order()
        index = {n: i for i, n in enumerate(nd + dc)}
        state = {}
        if eng.M is not None and eng.step_count > 0:
            M, V = eng.M.cpu(), eng.V.cpu()
            lay = eng.layout
            import math
            for n in lay.train_names:
                b = self._bucket_of(n)
                if b in self._conditional_buckets() and b not in eng.opt_seen:
                    continue                       # never had a gradient: torch has no state for it
                o = lay.off[n] - lay.train_start
                k = math.prod(lay.shapes[n])
                shp = lay.shapes[n]
                t = eng.group_steps[self._group_of_bucket(b)]
                state[index[n]] = {"step": torch.tensor(float(t)), "exp_avg": M[o:o + k].view(shp).clone(),
                                   "exp_avg_sq": V[o:o + k].view(shp).clone()}
        common = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "amsgrad": False, "maximize": False, "foreach": None,
                  "capturable": False}
        groups = [dict(common, weight_decay=0.0, params=list(range(len(nd)))),
                  dict(common, weight_decay=self.wd, params=list(range(len(nd), len(nd) + len(dc))))]
        return {"state": state, "param_groups": groups,
                "countr_amd": {"format": "torch_adamw.v3", "step": eng.step_count, "group_steps": list(eng.group_steps),
                               "seen_buckets": sorted(eng.opt_seen), "mask_draws": int(self.pro.draws)}}

    def _moments(self):
        eng = self.eng
        if eng.M is None:
            eng.M = torch.zeros_like(eng.G)
            eng.V = torch.zeros_like(eng.G)
        return eng.M, eng.V

    def load_optimizer_state(self, opt):
        """Restores a checkpoint's 'optimizer' entry: a torch.optim.AdamW state_dict (ours or the reference's -- same layout) or the
        flat v1 / v2 form older countr_amd checkpoints hold.  The moments are copied INTO the existing buffers (captured graphs keep
        pointing at them).  Raises on a state that does not fit this model; returns True."""
        self.flush()
        eng = self.eng
        M, V = self._moments()
        if isinstance(opt, dict) and opt.get("exp_avg") is not None:            # flat v1 / v2
            if opt["exp_avg"].numel() != eng.G.numel() or opt["exp_avg_sq"].numel() != eng.G.numel():
                raise ValueError("checkpoint optimizer state has %d elements, this model trains %d" % (opt["exp_avg"].numel(), eng.G.numel()))
            M.copy_(opt["exp_avg"].to(torch.float32))
            V.copy_(opt["exp_avg_sq"].to(torch.float32))
            eng.step_count = int(opt["step"])
            gs = opt.get("group_steps")
            eng.group_steps = [int(x) for x in gs] if gs is not None else [eng.step_count] * 3   # v1 files: one global counter
            eng.opt_seen = set(int(b) for b in opt.get("seen_buckets", (2, 3) if eng.step_count else ()))
            return True
        if not isinstance(opt, dict) or "state" not in opt or "param_groups" not in opt:
            raise ValueError("checkpoint 'optimizer' entry is neither a torch.optim.AdamW state_dict nor a countr_amd flat AdamW state")
        nd, dc = self._torch_param_order()
        pg = opt["param_groups"]
        if len(pg) != 2 or len(pg[0]["params"]) != len(nd) or len(pg[1]["params"]) != len(dc):
            raise ValueError("optimizer state_dict has groups of %s parameters, this model's add_weight_decay groups hold %s"
                             % ([len(g_["params"]) for g_ in pg], [len(nd), len(dc)]))
        ids = list(pg[0]["params"]) + list(pg[1]["params"])
        name_of = dict(zip(ids, nd + dc))
        lay = eng.layout
        trainable = set(lay.train_names)
        import math
        M.zero_()
        V.zero_()
        steps = {}            # counter group -> step
        seen = set()
        for pid, st_ in opt["state"].items():
            n = name_of.get(pid, name_of.get(int(pid)) if not isinstance(pid, int) else None)
            if n is None:
                raise ValueError("optimizer state for unknown parameter id %r" % (pid,))
            if n not in trainable:
                raise ValueError("optimizer state for %s, which this engine does not train" % n)
            if tuple(st_["exp_avg"].shape) != tuple(lay.shapes[n]):
                raise ValueError("optimizer state of %s has shape %s, expected %s" % (n, tuple(st_["exp_avg"].shape), tuple(lay.shapes[n])))
            o = lay.off[n] - lay.train_start
            k = math.prod(lay.shapes[n])
            M[o:o + k].copy_(st_["exp_avg"].reshape(-1).to(torch.float32))
            V[o:o + k].copy_(st_["exp_avg_sq"].reshape(-1).to(torch.float32))
            b = self._bucket_of(n)
            t = int(float(st_["step"]))
            g_ = self._group_of_bucket(b)
            if steps.setdefault(g_, t) != t:
                raise ValueError("parameters of one AdamW counter group carry different step counts (%d vs %d at %s): "
                                 "not a state this fused optimizer can continue" % (steps[g_], t, n))
            if b in self._conditional_buckets():
                seen.add(b)
        eng.group_steps = [int(steps.get(g_, 0)) for g_ in range(3)]
        eng.step_count = max(eng.group_steps)
        eng.opt_seen = seen
        extra = opt.get("countr_amd")
        if isinstance(extra, dict):                   # our own files also carry the global counter (pinned-slot rotation only)
            eng.step_count = int(extra.get("step", eng.step_count))
            self.pro.draws = int(extra.get("mask_draws", self.pro.draws))   # ... and the loss-mask stream's position (Philox counter)
        return True

    def _step(self, key):
        """backward phases in bucket order: after phase i the gradients of bucket i are final and its all-reduce starts on the
        side stream, overlapping phase i+1; the last phase's bucket(s) are reduced by finish(); then the fused AdamW.  The first
        phase starts with the prologue (staging copies of load(), loss mask, AdamW scalars: _Prologue)."""
        eng = self.eng
        last = self._micro + 1 == self.accum      # only the last micro-step of a window reduces and applies the gradients
        with torch.cuda.stream(self.stream):
            if eng.M is None:
                eng.M = torch.zeros_like(eng.G)
                eng.V = torch.zeros_like(eng.G)
            phases = self._phases(key)
            # this step's forward overwrites the plan's activation buffers (also when it is a graph replay): a pending autograd
            # backward of an earlier module forward with the same (batch, shot_num) must refuse (models_mae_cross._DecoderFn)
            for _name, _fn, gkey in phases[:1]:
                pl = eng.plans.get((self.B, gkey[0], True))
                if pl is not None:
                    pl.fwd_gen = getattr(pl, "fwd_gen", 0) + 1
            # what the window's end reduces / steps is known before anything runs (the sets follow from the shot_nums drawn so far), so
            # the AdamW scalars of an applying step travel with the prologue at the head of its first phase
            ckey, cskip, zfill = None, (), ()
            if last:
                touched, zfill = self._window_sets()
                cskip = tuple(self._comm_skip(touched))
                skip, zero = self._adam_sets(touched)
                self._upload_hyper(skip)
                ckey = (tuple(skip), tuple(zero))
            # defer_optimizer: this execution applies the PREVIOUS step's update (beside its frozen-encoder forward) and leaves its own
            # pending; the record it reads therefore carries the previous update's scalars
            defer = self.defer and self.use_graph and (not self.sync.comm or self.sync.capturable)
            # pipelined encoder: decided by load() (it chose the copies); only in the whole-step forms -- the lane must be joined inside
            # the graph that forked it
            pmode = self._pipe_mode
            assert pmode == "plain" or (self._pipe_ok() and not defer)
            pc = self._pc if defer else None
            if defer:
                hyper_now = self.pro.hyper.copy() if last else None
                if pc is not None:
                    self.pro.hyper[:] = self._pc_hyper
            elif self._pc is not None:
                self.flush()
            self._cur_pc = pc
            pend, self._pending = self._pending, None
            self.pro.fill(list(zip(*pend[:3])) if pend is not None else [], self._draw_mask, keep=pend[3] if pend is not None else ())
            # Between fill() and executed() the host's record count and the device's execution count must move together: an exception in
            # here (a kernel or RCCL error, a failed late capture, KeyboardInterrupt) re-reads the device's count before it propagates.
            try:
                if self.use_graph and not self.sync.comm:
                    # no collective between the phases (one rank): the whole (micro-)step is ONE graph replay -- every graph boundary
                    # costs ~20 us of idle GPU (4 launches per step before) -- and nothing else is launched between two replays
                    def whole(k, phases=phases):
                        self._cur_pc, self._cur_pipe = k[2], k[3]
                        self._run_phases_merged(phases)
                        if k[1] is not None:
                            self._phase_c(k[1])
                        if k[3] in ("steady", "coldnext"):
                            self.eng.pipe_join()
                    self._run_phase("all", whole, (tuple((name, gkey) for name, _fn, gkey in phases), None if defer else ckey, pc, pmode))
                elif self.use_graph and self.sync.capturable:
                    # RCCL: the bucket all-reduces are captured WITH the phases (graph nodes on the side stream between them), so a
                    # communicating step is one graph replay as well -- no host-issued collective, no graph boundary per phase
                    def whole(k, phases=phases, cskip=cskip, zfill=zfill, apply_now=not defer):
                        self._cur_pc, self._cur_pipe = k[4], k[5]
                        for i, (_name, fn, gkey) in enumerate(phases):
                            fn(gkey)
                            if k[1] is not None and i + 1 < len(phases):
                                self.sync.start(i)
                        if k[1] is not None:
                            self._zero_buckets(zfill)
                            self.sync.finish(skip=cskip)
                            if apply_now:
                                self._phase_c(k[1])
                        if k[5] in ("steady", "coldnext"):
                            self.eng.pipe_join()
                    gk = ("allc", (tuple((name, gkey) for name, _fn, gkey in phases), ckey, cskip if last else (), zfill if last else (), pc, pmode))
                    self._run_captured_comm(gk, whole)
                else:
                    self._cur_pipe = "plain"
                    for i, (name, fn, gkey) in enumerate(phases):
                        self._run_phase(name, fn, gkey)
                        if last and i + 1 < len(phases):
                            self.sync.start(i)
                    if last:
                        if zfill:
                            self._run_phase("z", self._zero_buckets, zfill)
                        self.sync.finish(skip=cskip)
                        self._run_phase("c", self._phase_c, ckey)
            except BaseException:
                self.pro.resync()
                self._cur_pc, self._cur_pipe, self._pipe_ready, self._pipe_mode = None, "plain", None, "plain"
                raise
            self.pro.executed(self.stream)
            self._cur_pc, self._cur_pipe = None, "plain"
            # the latent the encoder lane of this step left behind belongs to the images load() was told come next
            self._pipe_ready = self._pipe_next if pmode in ("steady", "coldnext") else None
            self._pipe_tok = eng.pipe_claim() if self._pipe_ready is not None else None
            self._pipe_gen = eng.generation
            self._pipe_next, self._pipe_mode = None, "plain"
            if defer:
                self._pc, self._pc_hyper = (ckey, hyper_now) if last else (None, None)
            if pend is not None:
                self._staging_consumed()
        torch.cuda.current_stream(eng.device).wait_stream(self.stream)   # results are visible to the caller's stream
        if last:
            self.model.mark_weights_synced()
            self._micro = 0
            self._touched = set()
            self._touched_any = set()
        else:
            self._micro += 1
        return last


class FinetuneStep(_GraphStep):
    def __init__(self, model, batch, lr=1e-5, weight_decay=0.05, betas=(0.9, 0.95), eps=1e-8, use_graph=True,
                 process_group=None, accum_iter=1, per_rank_shot=False, mask_seed=0, defer_optimizer=False, pipeline_encoder=False):
        """pipeline_encoder (round 6): software pipelining of the FROZEN ENCODER across iterations.  forward_encoder runs under no_grad on
        frozen weights (models_mae_cross.py:203-205), so batch k + 1's encoder forward depends on nothing step k computes: told the next
        batch's images (load(..., next_imgs=)), step k's graph runs it on a lane of its own beside batch k's decoder forward, loss,
        backward, all-reduce and AdamW, and leaves the latent in a buffer the next step's prologue copies into place.  The same launches on
        the same data: losses, counts, gradients and parameters are bit-identical to the unpipelined step (tests/test_trainer_gpu.py);
        what changes is that the decoder side's many small launches (normalisations, finishers, gaps between dependent kernels) are
        filled with the encoder's GEMMs instead of leaving the chip idle: 4.30 -> 3.95 ms per step at B = 8.  A step whose images were NOT
        announced by the previous load() (the first of a loop, any step behind a gap) computes its own encoder forward first; the last
        step of a loop passes next_imgs=None.  Whole-step graph modes and the 16-bit precisions only (else the plain step runs); takes
        the place of defer_optimizer (the update then runs at the tail of its own step, beside the encoder lane).
        defer_optimizer: software pipelining across iterations.  The encoder is frozen (models_mae_cross.py:204-205), so the first
        ~1.3 ms of a step do not depend on the previous step's optimizer update: AdamW + the shadow refresh of step k (~0.1 ms, bandwidth-
        bound) run at the HEAD of step k + 1's graph on the side lane in front of the exemplar CNN, beside the encoder's GEMMs, instead
        of alone at the tail of step k.  Same launches on the same data in the same order per buffer: parameters are bit-identical to the
        eager order (tests/test_trainer_gpu.py).  The price: after step() the last update is still PENDING -- flush() applies it;
        optimizer_state(), load_optimizer_state() and grad_norm() flush themselves; call flush() before reading parameters (validation,
        checkpoints) or letting anything else use the model.  Whole-step graph modes only (one rank, or RCCL with captured collectives).
        mask_seed: load(..., mask=None) lets the step draw the iteration's Bernoulli(0.8) loss mask itself (FSC_finetune_cross.py:
        290-292 draws np.random.binomial per iteration; the reference seeds numpy with seed + rank, :168-170 -- pass the same here):
        mask number t of this step object = Philox4x32-10(key = mask_seed, counter = (element / 4, 0, t)) < 0.8 * 2^32, drawn by the
        step's prologue kernel inside the captured graph.
        per_rank_shot: the reference's semantics (FSC_finetune_cross.py:276-284 draws shot_num per RANK; DDP's
        find_unused_parameters=True at :230 makes differing parameter subsets legal): step(S, shots_all=...) takes this rank's
        shot_num and every rank's; a conditional bucket (exemplar CNN / shot_token) some rank has a gradient for is all-reduced by
        EVERY rank -- zero-filled on the ranks without one -- and stepped by every rank, so parameters stay identical."""
        super().__init__(model, batch, lr, weight_decay, betas, eps, use_graph, process_group, accum_iter, mask_seed)
        self.per_rank = bool(per_rank_shot)
        self.pipe = bool(pipeline_encoder) and os.environ.get("COUNTR_PIPELINE_ENCODER", "1") != "0"
        sp = os.environ.get("COUNTR_PIPE_SPLIT", "")          # experiments: "nf,nh" launches of the encoder lane at the step's head / in front of
        self.pipe_split = tuple(int(x) for x in sp.split(",")) if sp else (10 ** 6, 0)      # the head's backward (rest: in front of the blocks' backward)
        self.defer = bool(defer_optimizer) and not self.pipe
        self.mse_ws = torch.zeros(self.eng.L.countr_masked_mse_workspace_floats(batch), device=self.eng.device)
        self.gt = torch.zeros((batch, self.eng.img, self.eng.img), device=self.eng.device)
        self.mask = torch.ones((self.eng.img, self.eng.img), device=self.eng.device)

    # ------------------------------------------------------------------ phases
    def _phase_a(self, key):
        """forward + loss (+ dL/dout) + backward until the head/decoder_norm gradients (bucket 0) are final."""
        S, acc = key
        eng = self.eng
        p = eng.plan(self.B, S, True)
        self._prologue_launch()
        eng.run(self._fwd_list(p, self._cur_pc))
        if S not in self.sums:
            self.sums[S] = torch.zeros(1 + 2 * self.B, device=eng.device)
        sums = self.sums[S]
        HW = eng.img * eng.img
        _lib.check(eng.L.countr_masked_mse_amp(p.buf["out"].data_ptr(), self.gt.data_ptr(), self.mask.data_ptr(), p.buf["dout"].data_ptr(),
                                               sums.data_ptr(), self.mse_ws.data_ptr(), self.B, HW, 1.0,
                                               self.amp.data_ptr() if self.amp is not None else None, eng._stream()), "masked_mse")
        if self._cur_pipe in ("steady", "coldnext"):
            eng.run(self._pipe_part(p, 1))
        eng.run(self._lists(p, acc).bwd_head)

    def _prologue_mask(self):
        return self.mask

    def _pipe_part(self, p, k):
        """Part k of the encoder lane's launches: issued (0) at the head of the step, (1) in front of the head's backward, (2) in front of
        the decoder blocks' backward -- one ordered lane, forked from the main lane at each of the three points.  `pipe_split` = how many
        launches the first two parts get (the third takes the rest)."""
        n = len(p.enc_pipe)
        nf = min(self.pipe_split[0], n)
        nh = min(self.pipe_split[1], n - nf)
        lo, hi = ((0, nf), (nf, nf + nh), (nf + nh, n))[k]
        if hi <= lo:
            return []
        mark = lambda *a: (None, a, None)
        return [mark("pfork")] + p.enc_pipe[lo:hi] + [mark("pmain")]

    def _fwd_list(self, p, pc):
        """The forward launch list of plan p; with a pending optimizer update pc (defer_optimizer) the list that applies it first:
        [fork | lane 1: AdamW + shadow refresh, exemplar CNN | lane 0: the frozen encoder | join | decoder_embed ... head].  Everything
        that reads a trainable parameter sits behind the AdamW on lane 1 or behind the join."""
        mode = self._cur_pipe
        if mode != "plain":
            # pipelined encoder: this batch's latent is in place (steady / last: copied from the shared buffer by the prologue; coldnext:
            # computed here first) and -- steady / coldnext -- the NEXT batch's frozen-encoder forward runs on its own lane until the end
            # of the step ("pfork" ... eng.pipe_join() behind the optimizer update)
            assert pc is None
            cache = p.__dict__.setdefault("_pipe_lists", {})
            if mode not in cache:
                mark = lambda *a: (None, a, None)
                dec = self.eng.decoder_ops_with_exemplar_lane(p)
                lane = self._pipe_part(p, 0)
                cache[mode] = {"steady": lane + dec, "last": dec, "coldnext": p.fwd[:p.enc_ops] + lane + dec}[mode]
            return cache[mode]
        if pc is None:
            return p.fwd_par
        cache = p.__dict__.setdefault("_defer_lists", {})
        if pc not in cache:
            mark = lambda *a: (None, a, None)
            adam = (lambda st, pc=pc: (self._phase_c(pc, stream=st), 0)[1], (), None)
            ex = getattr(p, "ex_range", None)
            enc = p.fwd[:p.enc_ops]
            if ex is None:
                lane1, rest = [adam], p.fwd[p.enc_ops:]
            else:
                lane1, rest = [adam] + p.fwd[ex[0]:ex[1]], p.fwd[p.enc_ops:ex[0]] + p.fwd[ex[1]:]
            cache[pc] = [mark("xfork"), mark("xlane", 1)] + lane1 + [mark("xlane", 0)] + enc + [mark("xjoin")] + rest
        return cache[pc]

    def _make_sync(self, process_group):
        lay = self.eng.layout   # buckets in backward-completion order: head | decoder blocks + embed | exemplar CNN | shot_token
        return GradSync(self.eng.G, None, None, process_group, buckets=[lay.bucket_range(b) for b in range(4)])

    def _group_of_bucket(self, bucket):
        from .engine import ADAM_GROUP_OF_BUCKET
        return ADAM_GROUP_OF_BUCKET.get(bucket, 0)

    def _conditional_buckets(self):
        return (2, 3)            # exemplar CNN (gradient only when shot_num > 0), shot_token (only when shot_num == 0)

    def _comm_skip(self, touched):
        """Conditional buckets without a gradient in this window (exemplar CNN when every micro-step had shot_num 0, shot_token
        when none had) are not all-reduced: their gradient is None / zero on every rank (`touched` is the union over ranks)."""
        return tuple(b for b in (2, 3) if b not in touched)

    def _adam_sets(self, touched):
        """Optimizer semantics of the reference environment (torch 1.13.1, FSC_finetune_cross.py:313-316): a parameter whose
        .grad is None is skipped; optimizer.zero_grad() leaves ZERO tensors behind, so once a conditional parameter set has had a
        gradient it is stepped in every later iteration -- with g = 0 (moment decay + bias-corrected update) when the drawn
        shot_num does not use it -- and its step counter runs from its first gradient."""
        seen = self.eng.opt_seen
        seen |= set(touched)
        skip = tuple(b for b in (2, 3) if b not in seen)
        zero = tuple(b for b in (2, 3) if b in seen and b not in touched)
        return skip, zero

    def _phase_b(self, key):
        S, acc = key
        p = self.eng.plan(self.B, S, True)
        if self._cur_pipe in ("steady", "coldnext"):
            self.eng.run(self._pipe_part(p, 2))
        self.eng.run(self._lists(p, acc).bwd_rest)

    def _phase_b2(self, key):
        S, acc = key
        self.eng.run(self._lists(self.eng.plan(self.B, S, True), acc).bwd_tok)

    def _run_phases_merged(self, phases):
        """a, then b and b2 together: the exemplar-token backward runs beside the tail of the decoder-block backward."""
        (_a, fa, ka), (_b, _fb, (S, acc)), (_b2, _fb2, (_S2, acc_tok)) = phases
        fa(ka)
        p = self.eng.plan(self.B, S, True)
        if self._cur_pipe in ("steady", "coldnext"):
            self.eng.run(self._pipe_part(p, 2))
        rest, tok = self._lists(p, acc), self._lists(p, acc_tok)
        if rest is tok:
            self.eng.run_backward_rest_and_tok(rest)
        else:                                        # (gradient accumulation with a changing shot_num: the two lists differ in their accumulate flags)
            self.eng.run(rest.bwd_rest)
            self.eng.run(tok.bwd_tok)

    def _phases(self, S):
        acc = int(self._micro > 0)
        tok = 3 if S == 0 else 2                     # the conditional bucket this shot_num writes: accumulate only if the window
        acc_tok = int(tok in self._touched)          # already wrote it (shot_num may change between micro-steps)
        self._touched.add(tok)
        return [("a", self._phase_a, (S, acc)), ("b", self._phase_b, (S, acc)), ("b2", self._phase_b2, (S, acc_tok))]

    # ------------------------------------------------------------------ public
    def load(self, imgs, boxes, gt, mask, S, next_imgs=None):
        """Stage one batch (device or host tensors) into the plan's input buffers on the step's stream.  mask: the iteration's loss
        mask [384, 384], or None -- the step then draws Bernoulli(0.8) itself (mask_seed of the constructor).  Dense fp32 device
        tensors of the buffers' shapes are copied by the prologue kernel at the head of step()'s graph: the sources are kept alive
        until then and MUST NOT BE MODIFIED between load() and the return of the following step() (a persistent input buffer refilled in
        place in between would change the batch that trains); a load() that is superseded by another load() drops its references.
        Anything else is copied here, tensor by tensor.
        next_imgs (pipeline_encoder): the images of the batch the NEXT load() will bring -- that very tensor object, unmodified until then
        -- or None (last iteration / unknown).  The next load() recognises it by identity; any other tensor simply gets its encoder
        forward computed in its own step."""
        cur = torch.cuda.current_stream(self.eng.device)
        self.stream.wait_stream(cur)           # producers of the inputs ran on the caller's stream
        pipe = self._pipe_ok()
        # this batch's latent is already waiting (and nothing has re-planned the engine or re-loaded its weights since it was computed)
        have = (pipe and self._pipe_ready is not None and imgs is self._pipe_ready and self.eng._ln_checked
                and self._pipe_gen == self.eng.generation and self.eng.pipe_owner(self._pipe_tok))
        if not pipe or next_imgs is None or not torch.is_tensor(next_imgs) or tuple(next_imgs.shape) != (self.B, 3, self.eng.img, self.eng.img):
            next_imgs = None
        mode = ("steady" if next_imgs is not None else "last") if have else ("coldnext" if next_imgs is not None else "plain")
        src = tuple(t for t in (imgs, boxes, gt, mask, next_imgs) if t is not None)
        self._draw_mask = mask is None
        raw_imgs = imgs
        staged = self._to_device(tuple(t for t in ((None if have else imgs), boxes, gt, mask, next_imgs) if t is not None))
        it_ = iter(staged)
        imgs = None if have else next(it_)
        boxes, gt = next(it_), next(it_)
        mask = next(it_) if mask is not None else None
        nxt = next(it_) if next_imgs is not None else None
        with torch.cuda.stream(self.stream):
            if not have:
                self.eng.check_ln_fold(imgs)         # (first batch behind a weight load only: the LayerNorm-fold guard, engine.py)
            p = self.eng.plan(self.B, S, True)
            if mode != "plain" and p.enc_pipe is None:
                mode, nxt, next_imgs = "plain", None, None
                assert not have
            self._pending = None
            self._pipe_mode, self._pipe_next = mode, next_imgs
            if not self._load_fused(p, imgs, boxes, gt, mask, S, keep=src, nxt=nxt, have=have):
                if not have:
                    self.eng._load_inputs(p, imgs, boxes, S)
                elif S > 0:
                    self.eng._load_boxes(p, boxes, S)
                self.gt.copy_(gt, non_blocking=True)
                if mask is not None:
                    self.mask.copy_(mask, non_blocking=True)
                if nxt is not None:
                    p.pipe_img[:nxt.numel()].view(nxt.shape).copy_(nxt, non_blocking=True)
                if have:
                    n = p.buf["latent"].numel()
                    p.buf["latent"].view(-1).copy_(p.pipe_latent[:n], non_blocking=True)
                self._staging_consumed()
                for t in src:                          # their memory must not be recycled before our copies have run
                    if t.is_cuda:
                        t.record_stream(self.stream)
        del raw_imgs

    def _load_fused(self, p, imgs, boxes, gt, mask, S, keep=(), nxt=None, have=False):
        """All staging copies of a batch by the step's prologue kernel when every source is a dense fp32 device tensor of the
        destination's shape; otherwise False (the caller copies tensor by tensor: dtype conversion, strided exemplar slices).
        pipeline_encoder: `have` -- the batch's latent comes out of the shared buffer instead of its images going in; `nxt` -- the next
        batch's images go to the encoder lane's input buffer (at most six copies per record: gt, mask, boxes, + two of these)."""
        pairs = ([] if have else [(imgs, p.buf["img"])]) + [(gt, self.gt)] + ([(mask, self.mask)] if mask is not None else [])
        if S > 0:
            if boxes.dim() != 5 or boxes.shape[1] != S:
                return False
            pairs.append((boxes, p.buf["boxes"]))
        for src, dst in pairs + ([(nxt, p.buf["img"])] if nxt is not None else []):      # (the lane's input buffer has the shape of `img`)
            if (not torch.is_tensor(src) or not src.is_cuda or src.dtype != dst.dtype or not src.is_contiguous() or src.numel() != dst.numel()
                    or (src.numel() * src.element_size()) % 16 or src.data_ptr() % 16 or dst.data_ptr() % 16):
                return False
        srcs = [s_.data_ptr() for s_, _ in pairs]
        dsts = [d.data_ptr() for _, d in pairs]
        nbytes = [s_.numel() * s_.element_size() for s_, _ in pairs]
        if nxt is not None:
            srcs.append(nxt.data_ptr()); dsts.append(p.pipe_img.data_ptr()); nbytes.append(nxt.numel() * nxt.element_size())
        if have:
            srcs.append(p.pipe_latent.data_ptr()); dsts.append(p.buf["latent"].data_ptr()); nbytes.append(p.pipe_latent_bytes)
        if len(srcs) > 6 or any(b % 16 for b in nbytes) or any(a % 16 for a in srcs + dsts):
            return False
        self._pending = (srcs, dsts, nbytes, [s_ for s_, _ in pairs] + [t for t in keep if torch.is_tensor(t)] + ([nxt] if nxt is not None else []))
        return True

    def step(self, S, lr=None, shots_all=None):
        """One (micro-)step on the inputs last given to load(): with accum_iter == k, every k-th call reduces the accumulated
        gradients and applies AdamW (self.applied tells which).  Returns the device tensor
        [loss, pred counts (B), gt counts (B)] of this batch without synchronising the host.
        per_rank_shot mode: S is THIS rank's shot_num and shots_all the shot_num of every rank of the group in this micro-step (e.g.
        parallel.rank_shot_nums: seeded per-rank draws every rank can evaluate -- no communication); without shots_all the ranks
        all-gather their S (a host-synchronous collective per step)."""
        if lr is not None:
            self.lr = lr
        if self.per_rank:
            if shots_all is None:
                shots_all = self._gather_shots(S)
            self._touched_any |= {3 if int(s_) == 0 else 2 for s_ in shots_all}
        self.applied = self._step(S)
        return self.sums[S]

    def _gather_shots(self, S):
        import torch.distributed as dist
        if self.world <= 1:
            return [S]
        dev = self.eng.device if dist.get_backend(self.pg) == "nccl" else "cpu"
        mine = torch.tensor([int(S)], dtype=torch.int64, device=dev)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine, group=self.pg)
        return [int(t.item()) for t in out]

    def grad_norm(self):
        """Device scalar: L2 norm of the (averaged) gradients the last optimizer step consumed -- get_grad_norm_ of
        util/misc.py:289-301 as returned by NativeScalerWithGradNormCount.__call__ (:266-280).  No host sync.  (defer_optimizer: applies
        the pending update first -- the norm is computed by the AdamW launch.)"""
        self.flush()
        return self.eng.gnorm[0] if self.eng.gnorm is not None else None


class PretrainStep(_GraphStep):
    """MAE pretraining step (reference FSC_pretrain.py:254-301 minus the per-step host syncs): random masking, forward,
    all-patch pixel MSE, full backward (decoder-side bucket all-reduced while the encoder backward runs), fused AdamW.
    The masking permutation is drawn with torch.rand + argsort exactly as models_mae_noct.py:119-121 (outside the
    captured region, so every replay sees fresh indices through the plan's index buffers)."""

    def __init__(self, model, batch, mask_ratio=0.5, lr=5e-6, weight_decay=0.05, betas=(0.9, 0.95), eps=1e-8, use_graph=True,
                 process_group=None, accum_iter=1):
        super().__init__(model, batch, lr, weight_decay, betas, eps, use_graph, process_group, accum_iter)
        self.K = model.len_keep(mask_ratio)

    def _phase_a(self, key):
        K, acc = key
        eng = self.eng
        p = eng.plan(self.B, K, True)
        self._prologue_launch()
        eng.run(p.fwd)
        eng.loss_launch(p, self.B, self.model.norm_pix_loss, amp=self.amp)
        eng.run(self._lists(p, acc).bwd_dec)

    def _make_sync(self, process_group):
        from .mae_engine import mae_enc_parts
        lay = self.eng.layout   # decoder side | encoder block groups from the top (mae_engine.mae_bucket_fn)
        self.parts = mae_enc_parts(self.eng.depth)
        return GradSync(self.eng.G, None, None, process_group, buckets=[lay.bucket_range(b) for b in range(self.parts + 1)])

    def _phases(self, K):
        acc = int(self._micro > 0)
        enc = lambda j: (lambda key: self.eng.run(self._lists(self.eng.plan(self.B, key[0], True), key[1]).bwd_enc[j]))
        return [("a", self._phase_a, (K, acc))] + [("b%d" % j, enc(j), (K, acc)) for j in range(self.parts)]

    def load(self, imgs, ids_shuffle=None):
        cur = torch.cuda.current_stream(self.eng.device)
        self.stream.wait_stream(cur)
        src = imgs
        (imgs,) = self._to_device((imgs,))
        with torch.cuda.stream(self.stream):
            p = self.eng.plan(self.B, self.K, True)
            dst = p.buf["img"]
            self._pending = None
            if (imgs.is_cuda and imgs.dtype == dst.dtype and imgs.is_contiguous() and imgs.numel() == dst.numel()
                    and not (imgs.numel() * imgs.element_size()) % 16 and not imgs.data_ptr() % 16):
                self._pending = ([imgs.data_ptr()], [dst.data_ptr()], [imgs.numel() * imgs.element_size()], [imgs, src])   # copied by the prologue
            else:
                dst.copy_(imgs, non_blocking=True)
                self._staging_consumed()
            if ids_shuffle is None:
                ids_shuffle = self.model.draw_masking(self.B, self.eng.device)
            self.eng.set_masking(p, ids_shuffle)
        if src.is_cuda:
            src.record_stream(self.stream)

    def step(self, lr=None):
        """One optimisation step on the batch last given to load(); returns the device loss tensor [1] (no host sync).
        pred / mask of the step stay in eng.plan(B, K, True).buf["pred" / "mask"]."""
        if lr is not None:
            self.lr = lr
        self.applied = self._step(self.K)
        return self.eng.plan(self.B, self.K, True).buf["loss"]
