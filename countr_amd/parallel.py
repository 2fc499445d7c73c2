"""Data-parallel plumbing of the finetune step (reference: DDP wrap at FSC_finetune_cross.py:230 and the loss
all-reduce at util/misc.py:424-432).

The path shards by images (pure data parallel; IN/GN/LN statistics are per sample) and has ONE exchange step per
iteration: a sum all-reduce of the flat fp32 decoder-gradient buffer, split in two buckets so the first
(density head + decoder_norm, final when ~3/4 of backward is done) overlaps the rest of backward on a side
stream.  Averaging (1/world) is folded into the fused AdamW (grad_scale).  The frozen encoder is never
communicated (the reference's DDP buckets all 98.9 M parameters).  Device-agnostic: the same code runs over
RCCL on GPUs and over gloo in the CPU tests.
"""
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


class GradSync:
    """Two-bucket all-reduce of a flat gradient buffer."""

    def __init__(self, flat_grad, bucket0, bucket_rest, group=None):
        self.g = flat_grad
        self.b0, self.b1 = bucket0, bucket_rest
        self.group = group
        self.world = world_size(group)
        self.stream = None
        if self.world > 1 and flat_grad.is_cuda:
            self.stream = torch.cuda.Stream(device=flat_grad.device)

    def start_bucket0(self):
        """Call when the gradients in bucket 0 are final; returns immediately on GPU (side stream)."""
        if self.world == 1:
            return
        view = self.g[self.b0[0]:self.b0[1]]
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.g.device))
            with torch.cuda.stream(self.stream):
                dist.all_reduce(view, group=self.group)
        else:
            dist.all_reduce(view, group=self.group)

    def finish(self):
        """Call after the rest of backward: reduces bucket 1 and joins the side stream."""
        if self.world == 1:
            return
        dist.all_reduce(self.g[self.b1[0]:self.b1[1]], group=self.group)
        if self.stream is not None:
            torch.cuda.current_stream(self.g.device).wait_stream(self.stream)

    @property
    def grad_scale(self):
        return 1.0 / self.world


def shard_batch(n_items, rank, world):
    """Contiguous, near-equal partition of n_items (images / windows) over ranks: inference shards, no collective."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
