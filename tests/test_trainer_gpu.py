"""GPU: the fused finetune step (forward, masked-MSE, decoder backward, AdamW; eager and hipGraph replay)
against the CPU oracle's loss/gradients + AdamW restatement, fp32 parity mode, reduced-depth model."""
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import countr_ref as R
from oracle import weights as W

pytestmark = pytest.mark.gpu
NAME = "tiny_test"


def make(precision):
    from countr_amd.models_mae_cross import SupervisedMAE
    p, D, depth, H, Dd, ddepth, Hd = W.CONFIGS[NAME]
    sd = W.make_state_dict(NAME, seed=3)
    m = SupervisedMAE(patch_size=p, embed_dim=D, depth=depth, num_heads=H, decoder_embed_dim=Dd, decoder_depth=ddepth,
                      decoder_num_heads=Hd, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to("cuda"), sd


@pytest.mark.parametrize("use_graph", [False, True])
def test_finetune_steps_match_oracle(use_graph):
    from countr_amd.trainer import FinetuneStep
    from countr_amd.engine import no_weight_decay
    m, sd = make("fp32")
    # eps=1e-4 keeps AdamW's g/(|g|+eps) well conditioned for the (near-)zero gradients; with 1e-8 their sign is noise
    step = FinetuneStep(m, batch=2, lr=1e-3, weight_decay=0.05, eps=1e-4, use_graph=use_graph)
    ref = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in ref.items()}
    t = 0
    # shot schedule exercises both parameter subsets twice (second use of each replays the captured graphs)
    for it, S in enumerate([3, 0, 3, 0]):
        imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=10 + it)
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), S)
        sums = step.step(S).clone()
        torch.cuda.synchronize()
        cur = {k: v.float().numpy() for k, v in ref.items()}
        out, rloss, rg = R.loss_and_grads(cur, imgs, boxes, gt, mask, S, NAME)
        assert abs(sums[0].item() - rloss.item()) <= 2e-3 * abs(rloss.item()), (it, S)
        assert np.abs(sums[1:3].cpu().numpy() - R.counts(out).numpy()).max() < 0.5
        t += 1
        for k, g in rg.items():
            if g is None:
                continue  # AdamW skips parameters without a gradient (exemplar CNN at S=0, shot_token at S>0)
            wd = 0.0 if no_weight_decay(k, ref[k].shape) else 0.05
            ref[k], m1, m2 = R.adamw_step(ref[k], g.double(), mom[k][0], mom[k][1], t, 1e-3, eps=1e-4, wd=wd)
            mom[k] = (m1, m2)
        for k, p in m.named_parameters():
            got = p.detach().cpu().double()
            # AdamW normalises the step to ~lr per element, so compare in units of lr: rms error everywhere, max error
            # outside the exemplar CNN (one fp32 ReLU-boundary flip, |xhat| ~ 1e-7, moves single elements there by
            # a fraction of lr: tools/diag_exemplar.py).  A wrong or missing step would be >= 1 lr off.
            d = (got - ref[k]).abs()
            assert d.pow(2).mean().sqrt().item() <= 0.03 * 1e-3 * (it + 1), (it, S, k)
            if not k.startswith("decoder_proj"):
                assert d.max().item() <= 0.25 * 1e-3 * (it + 1), (it, S, k, d.max().item())
    # frozen encoder untouched
    for k, p in m.named_parameters():
        if not k.startswith(("decoder", "decode_head", "shot_token")):
            assert torch.equal(p.detach().cpu(), torch.from_numpy(sd[k])), k


def test_bf16_step_runs_and_reduces_loss():
    from countr_amd.trainer import FinetuneStep
    m, sd = make("bf16")
    step = FinetuneStep(m, batch=2, lr=2e-4, use_graph=True)
    imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=21)
    step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), 3)
    losses = []
    for _ in range(6):
        losses.append(step.step(3)[0].item())
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0], losses


def test_all_shot_counts_interleaved_with_graphs():
    """Every shot_num in {0,1,2,3} interleaved (new plans are built while older graphs already exist): the loss of each
    step must equal an eager single-use engine run on the same weights/inputs."""
    from countr_amd.trainer import FinetuneStep
    m, sd = make("fp32")
    step = FinetuneStep(m, batch=2, lr=1e-4, use_graph=True)
    losses = {}
    order = [3, 0, 1, 2, 3, 1, 0, 2]
    for it, S in enumerate(order):
        imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=30 + it)
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), S)
        cur = {k: p.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
        loss = step.step(S)[0].item()
        _, rloss, _ = R.loss_and_grads(cur, imgs, boxes, gt, mask, S, NAME)
        assert abs(loss - rloss.item()) <= 2e-3 * abs(rloss.item()), (it, S, loss, rloss.item())


def test_host_batches_are_staged_and_match_device_batches():
    """load() accepts the DataLoader's host tensors (staged over a copy stream behind the previous step's compute): the step
    results are bit-identical to feeding the same batches as device tensors, also when the staging buffers are reused."""
    from countr_amd.trainer import FinetuneStep
    res = {}
    for mode in ("device", "host"):
        m, _ = make("bf16")
        step = FinetuneStep(m, batch=2, lr=1e-4, use_graph=True)
        out = []
        for it in range(4):
            arrs = W.make_inputs(batch=2, shots=3, seed=30 + it)
            ts = [torch.from_numpy(a) for a in arrs]
            ts = [t.pin_memory() for t in ts] if mode == "host" else [t.cuda() for t in ts]
            step.load(*ts, 3)
            out.append(step.step(3).clone())
        torch.cuda.synchronize()
        res[mode] = torch.stack(out)
    assert torch.equal(res["device"], res["host"])
    assert torch.isfinite(res["host"]).all()


@pytest.mark.parametrize("use_graph", [False, True])
def test_gradient_accumulation_matches_oracle(use_graph):
    """--accum_iter 2 (FSC_finetune_cross.py:300-305: loss / accum_iter, optimizer step every accum_iter-th iteration): the
    accumulated gradient is the mean over the window's micro-batches; shot_num changes inside a window, so a conditional
    parameter set (exemplar CNN / shot_token) steps iff some micro-step of the window gave it a gradient."""
    from countr_amd.trainer import FinetuneStep
    from countr_amd.engine import no_weight_decay
    m, sd = make("fp32")
    step = FinetuneStep(m, batch=2, lr=1e-3, weight_decay=0.05, eps=1e-4, use_graph=use_graph, accum_iter=2)
    ref = {k: torch.from_numpy(v).double() for k, v in sd.items()}
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in ref.items()}
    seed = 50
    for w, shots in enumerate([(3, 0), (0, 0), (2, 3)]):
        cur = {k: v.float().numpy() for k, v in ref.items()}
        acc = {}
        for j, S in enumerate(shots):
            imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=seed)
            seed += 1
            step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), S)
            sums = step.step(S).clone()
            assert step.applied == (j == 1)
            _, rloss, rg = R.loss_and_grads(cur, imgs, boxes, gt, mask, S, NAME)
            assert abs(sums[0].item() - rloss.item()) <= 2e-3 * abs(rloss.item()), (w, j)   # the logged loss is not divided
            for k, g in rg.items():
                if g is not None:
                    acc[k] = acc.get(k, 0) + g.double() / 2
        torch.cuda.synchronize()
        for k, g in acc.items():
            wd = 0.0 if no_weight_decay(k, ref[k].shape) else 0.05
            # the fused AdamW keeps ONE step counter (bias correction) for the whole flat buffer
            ref[k], m1, m2 = R.adamw_step(ref[k], g, mom[k][0], mom[k][1], w + 1, 1e-3, eps=1e-4, wd=wd)
            mom[k] = (m1, m2)
        for k, p in m.named_parameters():
            d = (p.detach().cpu().double() - ref[k]).abs()
            assert d.pow(2).mean().sqrt().item() <= 0.03 * 1e-3 * (w + 1), (w, k)
            if not k.startswith("decoder_proj"):
                assert d.max().item() <= 0.25 * 1e-3 * (w + 1), (w, k, d.max().item())
