"""Data-parallel plumbing of the finetune step (reference: DDP wrap at FSC_finetune_cross.py:230 and the loss
all-reduce at util/misc.py:424-432).

The path shards by images (pure data parallel; IN/GN/LN statistics are per sample) and has ONE exchange step per
iteration: a sum all-reduce of the flat fp32 decoder-gradient buffer, split in buckets ordered by backward completion
(density head + decoder_norm, then decoder blocks + decoder_embed, then the exemplar CNN / shot_token) so that every
bucket but the last overlaps the rest of backward on a side stream.  Averaging (1/world) is folded into the fused AdamW (grad_scale).  The frozen encoder is never
communicated (the reference's DDP buckets all 98.9 M parameters).  Device-agnostic: the same code runs over
RCCL on GPUs and over gloo in the CPU tests.
"""
import os
import random

import torch
import torch.distributed as dist


def world_size(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def shared_shot_num(step, seed=0, allow_zero=True):
    """shot_num for an iteration, identical on every rank.  The reference draws random.randint(0, 3) per rank from an
    unseeded generator (FSC_finetune_cross.py:278-284); a shared draw keeps the set of parameters that receive
    gradients identical across ranks, so AdamW's 'skip parameters without gradient' rule stays well defined."""
    rng = random.Random(1_000_003 * seed + step)
    return rng.randint(0 if allow_zero else 1, 3)


def rank_shot_nums(step, world, seed=0, allow_zero=True):
    """The reference's semantics: every rank draws its OWN shot_num per iteration (random.randint(0, 3) per process,
    FSC_finetune_cross.py:276-284).  Here rank r's draw comes from a generator seeded by (seed, step, r), so that every rank can
    evaluate ALL ranks' draws locally: which conditional parameter sets (exemplar CNN / shot_token) have a gradient somewhere is
    then known without a collective (trainer.FinetuneStep(per_rank_shot=True).step(S, shots_all=...)).  `allow_zero` may be a
    sequence with one entry per rank (the mosaic ban of :276-277 is per rank in the reference)."""
    az = list(allow_zero) if isinstance(allow_zero, (list, tuple)) else [allow_zero] * world
    return [random.Random(1_000_003 * seed + 7919 * (r + 1) + 104_729 * step).randint(0 if az[r] else 1, 3) for r in range(world)]


class GradSync:
    """Bucketed sum all-reduce of a flat gradient buffer.  Buckets are contiguous [start, end) slices ordered by the time their
    gradients become final in the backward pass; start(i) launches bucket i on a side stream as soon as the caller's stream
    has produced it (the collective then overlaps the rest of backward), finish() reduces what was not started and joins."""

    def __init__(self, flat_grad, bucket0, bucket_rest=None, group=None, buckets=None):
        self.g = flat_grad
        self.buckets = list(buckets) if buckets is not None else [tuple(bucket0), tuple(bucket_rest)]
        self.group = group
        self.world = world_size(group)
        # COUNTR_FORCE_COMM=1: issue the collectives even in a one-rank group (sum over one rank = identity).  A single-GPU box can
        # then run the REAL RCCL path -- communicator set-up, side stream, ordering against the graph replays (tests/test_ddp_gpu.py)
        self.comm = self.world > 1 or (os.environ.get("COUNTR_FORCE_COMM", "0") == "1" and dist.is_available() and dist.is_initialized())
        self.stream = None
        # RCCL collectives can be CAPTURED into a hipGraph (stream capture records them as graph nodes; gloo's host-side collectives
        # cannot): the trainer then replays a whole communicating step as one graph (COUNTR_GRAPH_COMM=0: one graph per phase with the
        # collectives issued by the host in between, as in rounds 1-2)
        self.capturable = bool(self.comm and flat_grad.is_cuda and dist.is_available() and dist.is_initialized()
                               and dist.get_backend(group) == "nccl" and os.environ.get("COUNTR_GRAPH_COMM", "1") != "0")
        self._started = set()
        self.profile = False        # bench.py: event pair around the join of every finish() -> exposed_us()
        self._exposed = []
        if self.comm and flat_grad.is_cuda:
            self.stream = torch.cuda.Stream(device=flat_grad.device)

    def start(self, i):
        """Call when the gradients of bucket i are final; returns immediately on GPU (side stream)."""
        self._started.add(i)
        if not self.comm:
            return
        s, e = self.buckets[i]
        if e <= s:
            return
        view = self.g[s:e]
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.g.device))
            with torch.cuda.stream(self.stream):
                dist.all_reduce(view, group=self.group)
        else:
            dist.all_reduce(view, group=self.group)

    def start_bucket0(self):
        self.start(0)

    def finish(self, skip=()):
        """Call after the last backward kernel: reduces the buckets not started yet (except `skip`: buckets whose parameters have
        no gradient this step) in one collective per contiguous run, then joins the side stream."""
        pending = [i for i in range(len(self.buckets)) if i not in self._started and i not in skip]
        self._started = set()
        if not self.comm:
            return
        runs = []
        for i in pending:
            s, e = self.buckets[i]
            if e <= s:
                continue
            if runs and runs[-1][1] == s:
                runs[-1][1] = e
            else:
                runs.append([s, e])
        main = torch.cuda.current_stream(self.g.device) if self.g.is_cuda else None
        prof = self.profile and main is not None and not torch.cuda.is_current_stream_capturing()   # (timing events cannot be captured)
        if prof:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(main)
        for s, e in runs:
            dist.all_reduce(self.g[s:e], group=self.group)
        if self.stream is not None:
            main.wait_stream(self.stream)
        if prof:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(main)
            self._exposed.append((e0, e1))

    def exposed_us(self):
        """Median time the step's stream spent between reaching finish() and having every bucket reduced (the collectives issued there
        + the wait for the ones still running on the side stream): the communication that did NOT hide under backward.  Call after a
        synchronize; needs profile = True."""
        if not self._exposed:
            return None
        v = sorted(a.elapsed_time(b) * 1e3 for a, b in self._exposed)
        self._exposed = []
        return v[len(v) // 2]

    def bucket_bytes(self):
        return [4 * max(0, e - s) for s, e in self.buckets]

    @property
    def grad_scale(self):
        return 1.0 / self.world


def shard_batch(n_items, rank, world):
    """Contiguous, near-equal partition of n_items (images / windows) over ranks: inference shards, no collective."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
