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


class TorchAdamW:
    """The reference's optimizer on CPU in fp64: torch.optim.AdamW(betas=(0.9, 0.95)) over timm's two add_weight_decay groups
    (FSC_finetune_cross.py:234-235) with the zero_grad() of its pinned torch 1.13.1 (set_to_none=False: a parameter that had a
    gradient once keeps a ZERO gradient and is stepped in every later iteration; one that never had any is skipped)."""

    def __init__(self, sd, lr, eps, wd=0.05):
        from countr_amd.engine import is_trainable, no_weight_decay
        self.p = {k: torch.nn.Parameter(torch.from_numpy(v).double()) for k, v in sd.items() if is_trainable(k)}
        decay = [v for k, v in self.p.items() if not no_weight_decay(k, v.shape)]
        no_decay = [v for k, v in self.p.items() if no_weight_decay(k, v.shape)]
        self.opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": wd}], lr=lr,
                                     betas=(0.9, 0.95), eps=eps)

    def state_dict_f32(self, sd):
        cur = {k: v.copy() for k, v in sd.items()}
        for k, v in self.p.items():
            cur[k] = v.detach().float().numpy()
        return cur

    def accumulate(self, grads, scale=1.0):
        for k, g in grads.items():
            if g is None:
                continue
            g = g.double() * scale
            self.p[k].grad = g if self.p[k].grad is None else self.p[k].grad + g

    def grad_norm(self):
        return torch.sqrt(sum((v.grad.double() ** 2).sum() for v in self.p.values() if v.grad is not None))

    def step(self):
        self.opt.step()
        self.opt.zero_grad(set_to_none=False)


def check_params(m, ref, lr, it, tag):
    for k, p in m.named_parameters():
        if k not in ref.p:
            continue
        # AdamW normalises the step to ~lr per element, so compare in units of lr: rms error everywhere, max error
        # outside the exemplar CNN (one fp32 ReLU-boundary flip, |xhat| ~ 1e-7, moves single elements there by
        # a fraction of lr: tools/diag_exemplar.py).  A wrong or missing step (or a wrong bias correction: the first
        # update of a late-starting parameter is 1 lr per element) would be >= 0.3 lr off.
        d = (p.detach().cpu().double() - ref.p[k].detach()).abs()
        assert d.pow(2).mean().sqrt().item() <= 0.03 * lr * (it + 1), (tag, it, k, d.pow(2).mean().sqrt().item())
        if not k.startswith("decoder_proj"):
            assert d.max().item() <= 0.25 * lr * (it + 1), (tag, it, k, d.max().item())


@pytest.mark.parametrize("use_graph", [False, True])
def test_finetune_steps_match_torch_adamw(use_graph):
    """FinetuneStep (loss, gradients, optimizer) against the oracle's loss / gradients fed to the REAL torch.optim.AdamW over the
    shot schedule [3, 0, 3, 0, 1]: shot_token takes its first step at iteration 2 (own bias correction: that update is a full
    lr per element), the exemplar CNN is stepped with a zero gradient when shot_num == 0, shot_token likewise afterwards."""
    from countr_amd.trainer import FinetuneStep
    m, sd = make("fp32")
    lr = 1e-3
    # eps=1e-4 keeps AdamW's g/(|g|+eps) well conditioned for the (near-)zero gradients; with 1e-8 their sign is noise
    step = FinetuneStep(m, batch=2, lr=lr, weight_decay=0.05, eps=1e-4, use_graph=use_graph)
    ref = TorchAdamW(sd, lr, 1e-4)
    tok0 = torch.from_numpy(sd["shot_token"]).double()
    for it, S in enumerate([3, 0, 3, 0, 1]):
        imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=10 + it)
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), S)
        sums = step.step(S).clone()
        gn = step.grad_norm().item()
        torch.cuda.synchronize()
        out, rloss, rg = R.loss_and_grads(ref.state_dict_f32(sd), imgs, boxes, gt, mask, S, NAME)
        assert abs(sums[0].item() - rloss.item()) <= 2e-3 * abs(rloss.item()), (it, S)
        assert np.abs(sums[1:3].cpu().numpy() - R.counts(out).numpy()).max() < 0.5
        ref.accumulate(rg)
        assert abs(gn - ref.grad_norm().item()) <= 2e-3 * ref.grad_norm().item(), (it, gn, ref.grad_norm().item())
        ref.step()
        check_params(m, ref, lr, it, "steps")
        tok = dict(m.named_parameters())["shot_token"].detach().cpu().double()
        if it == 0:
            assert torch.equal(tok, tok0)                       # no gradient yet: skipped
        if it == 1:                                             # first step of shot_token: |update| = lr * g/(|g| + eps) ~ lr
            assert 0.5 * lr < (tok - tok0).abs().median().item() <= 1.01 * lr
    assert step.eng.group_steps == [5, 5, 4] and step.eng.opt_seen == {2, 3}
    # frozen encoder untouched
    for k, p in m.named_parameters():
        if not k.startswith(("decoder", "decode_head", "shot_token")):
            assert torch.equal(p.detach().cpu(), torch.from_numpy(sd[k])), k


@pytest.mark.parametrize("B,shots", [(3, [3, 0]), (5, [0, 2]), (26, [3])])
def test_finetune_step_at_odd_batch_sizes_matches_oracle(B, shots):
    """Batch sizes a user actually passes (the reference's and this CLI's default is --batch_size 26; 3 and 5 give row counts that are
    not multiples of any GEMM tile): M = 576 B takes the ragged-tile paths of gemm_kernel, other split-K factors, GroupNorm splits and
    the attention kernel's non-multiple-of-8 (batch x head) mapping.  Loss, counts, gradient norm and the stepped parameters against the
    oracle + the real torch.optim.AdamW, fp32 parity mode, eager and graph replay of the same step."""
    from countr_amd.trainer import FinetuneStep
    lr = 1e-3
    for use_graph in (False, True):
        m, sd = make("fp32")
        step = FinetuneStep(m, batch=B, lr=lr, weight_decay=0.05, eps=1e-4, use_graph=use_graph)
        ref = TorchAdamW(sd, lr, 1e-4)
        for it, S in enumerate(shots):
            imgs, boxes, gt, mask = W.make_inputs(batch=B, shots=3, seed=90 + it)
            step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), S)
            sums = step.step(S).clone()
            gn = step.grad_norm().item()
            torch.cuda.synchronize()
            if use_graph:            # the graph-replayed run is compared with the eager one (bit-identical), not with the oracle again
                continue
            torch.set_num_threads(min(__import__("os").cpu_count(), 32))
            out, rloss, rg = R.loss_and_grads(ref.state_dict_f32(sd), imgs, boxes, gt, mask, S, NAME)
            assert abs(sums[0].item() - rloss.item()) <= 2e-3 * abs(rloss.item()), (B, it, S)
            assert np.abs(sums[1:1 + B].cpu().numpy() - R.counts(out).numpy()).max() < 0.5
            ref.accumulate(rg)
            assert abs(gn - ref.grad_norm().item()) <= 2e-3 * ref.grad_norm().item(), (B, it, gn, ref.grad_norm().item())
            ref.step()
            check_params(m, ref, lr, it, "B=%d" % B)
        params = {k: p.detach().cpu().clone() for k, p in m.named_parameters()}
        if use_graph:
            for k in params:
                assert torch.equal(params[k], eager[k]), (B, k)
        eager = params


def test_optimizer_state_roundtrip():
    """optimizer_state() -> load_optimizer_state() on a fresh step continues bit-identically (moments, global and per-group
    step counters, the set of conditional buckets that already had a gradient); foreign dicts are refused."""
    from countr_amd.trainer import FinetuneStep
    res = []
    for resume in (False, True):
        m, sd = make("fp32")
        step = FinetuneStep(m, batch=2, lr=1e-3, eps=1e-4, use_graph=False)
        for it, S in enumerate([0, 0, 3, 0]):
            if resume and it == 2:
                state = step.optimizer_state()
                weights = {k: v.detach().clone() for k, v in m.state_dict().items()}
                m, _ = make("fp32")
                m.load_state_dict(weights)
                step = FinetuneStep(m, batch=2, lr=1e-3, eps=1e-4, use_graph=False)
                with pytest.raises(ValueError):
                    step.load_optimizer_state({"state": {}, "param_groups": []})        # not this model's parameter groups
                with pytest.raises(ValueError):
                    step.load_optimizer_state({"foo": 1})
                assert set(state) >= {"state", "param_groups"} and len(state["param_groups"]) == 2   # torch.optim.AdamW layout
                assert step.load_optimizer_state(state)
                assert step.eng.group_steps == [2, 0, 2] and step.eng.opt_seen == {3}
            imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=70 + it)
            step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), S)
            step.step(S)
        torch.cuda.synchronize()
        res.append({k: p.detach().cpu().clone() for k, p in m.named_parameters()})
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


def test_optimizer_state_is_the_reference_adamw_state_dict(tmp_path):
    """Checkpoint interchange with the reference (util/misc.py:312-318 saves optimizer.state_dict(), :400-421 loads it): the REAL
    torch.optim.AdamW over timm's add_weight_decay groups of ALL requires_grad parameters in named_parameters() order (encoder included,
    it just never gets a gradient) is stepped with the oracle's gradients for [3, 0, 3]; its state_dict() -- through torch.save /
    torch.load -- is loaded into a fresh FinetuneStep, both continue with [0, 1] and stay together.  And the other way round: the
    state FinetuneStep exports loads into a fresh torch.optim.AdamW (load_state_dict) and equals the reference optimizer's state."""
    from countr_amd.trainer import FinetuneStep
    from countr_amd.engine import no_weight_decay
    m, sd = make("fp32")
    lr, eps = 1e-3, 1e-4
    frozen = ("pos_embed", "decoder_pos_embed")

    def reference_optimizer(params):
        names = [k for k, _ in m.named_parameters() if k not in frozen]
        nd = [params[k] for k in names if no_weight_decay(k, params[k].shape)]
        dc = [params[k] for k in names if not no_weight_decay(k, params[k].shape)]
        return torch.optim.AdamW([{"params": nd, "weight_decay": 0.0}, {"params": dc, "weight_decay": 0.05}], lr=lr, betas=(0.9, 0.95), eps=eps)

    P = {k: torch.nn.Parameter(torch.from_numpy(v).double()) for k, v in sd.items() if k not in frozen}
    opt = reference_optimizer(P)
    cur = lambda: {k: (P[k].detach().float().numpy() if k in P else v) for k, v in sd.items()}
    sched = [3, 0, 3, 0, 1]
    for it, S in enumerate(sched[:3]):
        imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=40 + it)
        _, _, rg = R.loss_and_grads(cur(), imgs, boxes, gt, mask, S, NAME)
        for k, g_ in rg.items():
            if g_ is not None:
                P[k].grad = g_.double() if P[k].grad is None else P[k].grad + g_.double()
        opt.step()
        opt.zero_grad(set_to_none=False)
    path = tmp_path / "ref_opt.pth"
    torch.save({"optimizer": opt.state_dict()}, path)
    ref_state = torch.load(path, map_location="cpu", weights_only=False)["optimizer"]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in cur().items()})
    step = FinetuneStep(m, batch=2, lr=lr, weight_decay=0.05, eps=eps, use_graph=False)
    assert step.load_optimizer_state(ref_state)
    assert step.eng.group_steps == [3, 3, 2] and step.eng.opt_seen == {2, 3}
    for it, S in enumerate(sched[3:], start=3):
        imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=40 + it)
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), S)
        step.step(S)
        torch.cuda.synchronize()
        _, _, rg = R.loss_and_grads(cur(), imgs, boxes, gt, mask, S, NAME)
        for k in P:
            if P[k].grad is not None:
                P[k].grad.zero_()
        for k, g_ in rg.items():
            if g_ is not None:
                P[k].grad = g_.double() if P[k].grad is None else P[k].grad + g_.double()
        opt.step()
        for k, p_ in m.named_parameters():
            if k in frozen or not k.startswith(("decoder", "decode_head", "shot_token")):
                continue
            d = (p_.detach().cpu().double() - P[k].detach()).abs()
            assert d.pow(2).mean().sqrt().item() <= 0.03 * lr * (it - 2), (it, k, d.pow(2).mean().sqrt().item())
    # export -> a fresh torch optimizer accepts it, and it carries the same moments / counters as the reference optimizer's
    mine = step.optimizer_state()
    P2 = {k: torch.nn.Parameter(v.detach().clone()) for k, v in P.items()}
    opt2 = reference_optimizer(P2)
    opt2.load_state_dict({"state": mine["state"], "param_groups": mine["param_groups"]})
    theirs = opt.state_dict()
    assert set(mine["state"]) == set(theirs["state"])
    for pid, st_ in theirs["state"].items():
        assert float(mine["state"][pid]["step"]) == float(st_["step"]), pid
        for key in ("exp_avg", "exp_avg_sq"):
            a_, b_ = mine["state"][pid][key].double(), st_[key].double()
            assert (a_ - b_).abs().max().item() <= 2e-2 * b_.abs().max().item() + 1e-12, (pid, key)


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
    results are bit-identical to feeding the same batches as device tensors, also when the (three round-robin) staging slots are
    reused -- seven steps walk the ring twice."""
    from countr_amd.trainer import FinetuneStep
    res = {}
    for mode in ("device", "host"):
        m, _ = make("bf16")
        step = FinetuneStep(m, batch=2, lr=1e-4, use_graph=True)
        out = []
        for it in range(7):
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
def test_gradient_accumulation_matches_torch_adamw(use_graph):
    """--accum_iter 2 (FSC_finetune_cross.py:300-305: loss / accum_iter, optimizer step every accum_iter-th iteration): the
    accumulated gradient is the mean over the window's micro-batches; shot_num changes inside a window, so a conditional
    parameter set (exemplar CNN / shot_token) gets a gradient iff some micro-step of the window gave it one -- and, once it had
    one, is stepped with zeros otherwise (window 2 below: the exemplar CNN)."""
    from countr_amd.trainer import FinetuneStep
    m, sd = make("fp32")
    lr = 1e-3
    step = FinetuneStep(m, batch=2, lr=lr, weight_decay=0.05, eps=1e-4, use_graph=use_graph, accum_iter=2)
    ref = TorchAdamW(sd, lr, 1e-4)
    seed = 50
    for w, shots in enumerate([(3, 0), (0, 0), (2, 3)]):
        cur = ref.state_dict_f32(sd)
        for j, S in enumerate(shots):
            imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=seed)
            seed += 1
            step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), S)
            sums = step.step(S).clone()
            assert step.applied == (j == 1)
            _, rloss, rg = R.loss_and_grads(cur, imgs, boxes, gt, mask, S, NAME)
            assert abs(sums[0].item() - rloss.item()) <= 2e-3 * abs(rloss.item()), (w, j)   # the logged loss is not divided
            ref.accumulate(rg, 0.5)
        torch.cuda.synchronize()
        ref.step()
        check_params(m, ref, lr, w, "accum")
    assert step.eng.group_steps == [3, 3, 3]


def check_step_against_oracle(step, m, cur, batch, S, sums, model_name="mae_vit_base_patch16", count_scale=None):
    """One FinetuneStep result (loss / counts in `sums`, gradients in the step's flat buffer) against the oracle evaluated at `cur` --
    the parameters the engine held BEFORE the step.  bf16 bars = tests/test_model_gpu.py::test_bf16_gradients_close_to_oracle (measured
    values in profiles/r3_bf16_gradient_quality.txt).  count_scale: the count bar is relative to max(|oracle count|, count_scale) -- the
    map is a sum of cancelling terms whose magnitude stays that of the initial model while AdamW drives the count itself down (784 ->
    228 in three steps at lr 1e-5: every weight moves by lr in a coherent direction, far below one bf16 ulp of the weights, so the bf16
    shadows follow the fp32 master only statistically); the absolute error shrinks (7 -> 4 counts), the ratio to the shrunken count
    does not.  Returns (worst (1 - cos, |ratio - 1|) seen outside the exemplar CNN, oracle counts)."""
    imgs, boxes, gt, mask = batch
    B = imgs.shape[0]
    out, rloss, rg = R.loss_and_grads(cur, imgs, boxes, gt, mask, S, model_name)
    loss = sums[0].item()
    assert abs(loss - rloss.item()) <= 1e-2 * abs(rloss.item()), (S, loss, rloss.item())
    rc = R.counts(out).numpy()
    cnt = sums[1:1 + B].cpu().numpy()
    den = np.maximum(np.abs(rc), count_scale if count_scale is not None else 0.0)
    # (first step = the pinned initial weights: the bars of tests/test_model_gpu.py; later steps: AdamW has moved every fp32 master weight
    # by a few 1e-5, 1e-3 of a bf16 ulp, so the bf16 shadows realise the coherent shift only through the few weights that crossed a
    # rounding boundary -- measured 1.2 % of the initial count after three steps; bar 2.5 %)
    bar = (6e-2 if S == 0 else 1e-2) if count_scale is None else (6e-2 if S == 0 else 2.5e-2)
    assert (np.abs(cnt - rc) / den).max() < bar, (S, cnt, rc)
    gtc = sums[1 + B:1 + 2 * B].cpu().numpy()
    assert np.abs(gtc - gt.reshape(B, -1).sum(1) / 60).max() < 1e-2
    checked, worst = 0, (0.0, 0.0)
    for k, ref in rg.items():
        if ref is None:
            continue
        ref = ref.double()
        if ref.norm() < 1e-3:
            continue
        got = step.eng.gview(k).detach().cpu().double()
        cos = ((got * ref).sum() / (got.norm() * ref.norm())).item()
        ratio = (got.norm() / ref.norm()).item()
        if S == 0:
            assert cos > 0.985 and 0.78 < ratio < 1.02, (S, k, cos, ratio)
        elif k.startswith("decoder_proj"):
            assert cos > 0.97 and abs(ratio - 1) < (0.02 if count_scale is None else 0.05), (S, k, cos, ratio)
        else:
            # (norms follow the magnitude of the density map -- dL/dout ~ out where gt = 0 -- so behind the first step they inherit the
            # count bar above: measured 3.0 % low where the counts are 1.9 % low; the direction bar does not move)
            assert cos > 0.999 and abs(ratio - 1) < (0.015 if count_scale is None else 0.045), (S, k, cos, ratio)
            worst = (max(worst[0], 1 - cos), max(worst[1], abs(ratio - 1)))
        checked += 1
    assert checked >= (50 if S == 0 else 55), checked
    return worst, rc


def test_finetune_step_at_the_real_config_matches_oracle():
    """The BENCHMARKED object at its own size (BASELINE config 2; FSC_finetune_cross.py:286-316): FinetuneStep on mae_vit_base_patch16,
    bf16, B = 8, use_graph=True (side lanes, fused loss, grouped weight gradients, fused AdamW) over the shot schedule [3, 0, 1, 3, 3, 0].
    A graph key = (shot_num, AdamW skip / zero sets): steps 1-4 each meet a new key (run eagerly, then captured), steps 5 and 6 REPLAY
    the graphs steps 4 and 2 captured, on parameters the fused AdamW has moved since.  Every step is compared with
    the oracle evaluated at the engine's own parameters in front of it: loss to 1e-2, counts to 1 % (6 % at shot_num 0: module
    docstring of test_model_gpu.py), every trainable tensor's gradient (read from the step's flat buffer) by direction and norm."""
    import models_mae_cross as mm
    from countr_amd.trainer import FinetuneStep
    name = "mae_vit_base_patch16"
    m = mm.__dict__[name](norm_pix_loss=False, precision="bf16")
    sd = W.make_state_dict(name, seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.to("cuda").train()
    B = 8
    step = FinetuneStep(m, batch=B, lr=1e-5, weight_decay=0.05, use_graph=True)
    torch.set_num_threads(min(__import__("os").cpu_count(), 32))
    before, scale = None, None
    for it, S in enumerate([3, 0, 1, 3, 3, 0]):
        batch = W.make_inputs(batch=B, shots=3, seed=80 + it)
        ngraphs = len(step.graphs)
        cur = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
        before = before or cur
        with step.on_stream():                               # as bench.py and the CLI drive it
            step.load(*(torch.from_numpy(a).cuda() for a in batch), S)
            sums = step.step(S).clone()
        torch.cuda.synchronize()
        assert len(step.graphs) == (ngraphs + 1 if it < 4 else 4), (it, len(step.graphs))      # steps 5, 6: pure replays
        worst, rc = check_step_against_oracle(step, m, cur, batch, S, sums, name, count_scale=scale)
        scale = scale if scale is not None else float(np.abs(rc).mean())       # the initial model's count magnitude
        print("step", it, "shot_num", S, "worst 1-cos %.2e, |norm ratio - 1| %.2e" % worst, "counts", rc[:3])
    after = m.state_dict()
    moved = sum(float((after[k].detach().float().cpu() - torch.from_numpy(before[k])).abs().max()) > 0 for k in before
                if k.startswith(("decoder", "decode_head", "shot_token")) and "pos_embed" not in k)
    assert moved >= 70


@pytest.mark.parametrize("use_graph", [False, True])
def test_step_draws_its_own_loss_mask(use_graph):
    """load(mask=None): the step draws the iteration's Bernoulli(0.8) loss mask in its prologue (FSC_finetune_cross.py:290-292 draws one
    per iteration) -- mask t of the step object is oracle/philox.loss_mask(mask_seed, t); the loss of every step equals the oracle's
    loss under exactly that mask (a mask that was not redrawn, or drawn for another t, moves the loss by ~1e-2), eager and replayed."""
    from oracle.philox import loss_mask
    from countr_amd.trainer import FinetuneStep
    m, sd = make("fp32")
    step = FinetuneStep(m, batch=2, lr=1e-4, use_graph=use_graph, mask_seed=99)
    masks = []
    for it, S in enumerate([3, 3, 0, 3]):
        imgs, boxes, gt, _mask = W.make_inputs(batch=2, shots=3, seed=60 + it)
        cur = {k: p.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt)), None, S)
        loss = step.step(S)[0].item()
        mk = loss_mask(99, it).reshape(384, 384)
        assert np.array_equal(step.mask.cpu().numpy(), mk), it
        masks.append(mk)
        _, rloss, _ = R.loss_and_grads(cur, imgs, boxes, gt, mk, S, NAME)
        assert abs(loss - rloss.item()) <= 2e-3 * abs(rloss.item()), (it, S, loss, rloss.item())
    assert all((masks[0] != mk).mean() > 0.2 for mk in masks[1:])
    # a caller-supplied mask still wins
    imgs, boxes, gt, mask = W.make_inputs(batch=2, shots=3, seed=70)
    step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt, mask)), 3)
    step.step(3)
    assert np.array_equal(step.mask.cpu().numpy(), mask)


@pytest.mark.parametrize("precision,accum", [("bf16", 1), ("fp32", 1), ("bf16", 2)])
def test_deferred_optimizer_is_bit_identical(precision, accum):
    """FinetuneStep(defer_optimizer=True): AdamW + shadow refresh of step k run at the head of step k + 1's graph, on the side lane
    beside the frozen encoder's forward (models_mae_cross.py:204-205: nothing in it depends on the update).  Same launches, same data:
    every step's loss / counts and, after flush(), every parameter, AdamW moment and the gradient norm are BIT-identical to the eager
    order -- over a shot schedule that changes the AdamW key (first shot_token step, zero-gradient steps), with accumulation windows,
    and with a parameter read (flush) in the middle."""
    from countr_amd.trainer import FinetuneStep
    sched = [3, 0, 3, 1, 0, 3, 3, 2]
    res = {}
    for defer in (False, True):
        m, sd = make(precision)
        step = FinetuneStep(m, batch=2, lr=1e-3, eps=1e-4, use_graph=True, accum_iter=accum, defer_optimizer=defer, mask_seed=5)
        sums, mids = [], None
        for it, S in enumerate(sched):
            imgs, boxes, gt, _mask = W.make_inputs(batch=2, shots=3, seed=200 + it)
            step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt)), None, S)
            sums.append(step.step(S, lr=1e-3 * (1 + 0.1 * it)).clone())
            if defer:
                assert (step._pc is not None) == step.applied
            if it == 4:
                step.flush()                        # a reader in the middle of training (validation, checkpoint)
                mids = {k: p.detach().clone() for k, p in m.named_parameters()}
        gn = step.grad_norm().clone()               # (flushes)
        assert step._pc is None
        torch.cuda.synchronize()
        res[defer] = (torch.stack(sums), {k: p.detach().clone() for k, p in m.named_parameters()}, mids, step.eng.M.clone(), step.eng.V.clone(), gn,
                      step.optimizer_state())
    a, b = res[False], res[True]
    assert torch.equal(a[0], b[0])
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k
        assert torch.equal(a[2][k], b[2][k]), k
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]) and torch.equal(a[5], b[5])
    assert a[6]["countr_amd"] == b[6]["countr_amd"]
    moved = sum(float((a[1][k].cpu() - torch.from_numpy(sd[k])).abs().max()) > 0 for k in a[1] if k.startswith(("decoder", "decode_head", "shot_token")) and "pos_embed" not in k)
    assert moved >= 40


@pytest.mark.parametrize("defer", [False, True])
def test_fp16_dynamic_loss_scale_follows_gradscaler(defer):
    """fp16 mode = the reference's AMP loop (FSC_finetune_cross.py:286,313; util/misc.py:260-286: GradScaler().scale(loss).backward(),
    unscale_, step skipped on a non-finite gradient, update()) with the scaler's state on the device.  Started from an absurd scale (2^40:
    every 16-bit gradient overflows) with growth interval 3: while the gradient is not finite the parameters and moments do not move
    (bit for bit), the logged norm is inf and the scale halves; from the first finite gradient on the parameters move, and the scale
    follows GradScaler's rule step by step (x 2 after 3 clean steps, x 0.5 on an overflow)."""
    from countr_amd.trainer import FinetuneStep
    m, sd = make("fp16")
    step = FinetuneStep(m, batch=2, lr=1e-3, eps=1e-4, use_graph=True, defer_optimizer=defer, mask_seed=3)
    step.amp[0], step.amp[4] = 2.0 ** 40, 3.0
    p0 = {k: p.detach().clone() for k, p in m.named_parameters()}
    scale, good, skipped, first_clean = 2.0 ** 40, 0, 0, None
    for it in range(60):
        imgs, boxes, gt, _mask = W.make_inputs(batch=2, shots=3, seed=300 + it % 4)
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt)), None, 3)
        loss = step.step(3)[0].item()
        gn = step.grad_norm().item()                   # (flushes a deferred update)
        assert np.isfinite(loss)
        now, nskip = step.loss_scale, step.skipped_steps()
        if nskip > skipped:                            # this step's gradient was not finite
            assert gn == float("inf") and now == scale * 0.5
            scale, good, skipped = scale * 0.5, 0, nskip
            if first_clean is None:
                for k, p in m.named_parameters():
                    assert torch.equal(p.detach(), p0[k]), k
                assert float(step.eng.M.abs().max()) == 0.0 and float(step.eng.V.abs().max()) == 0.0
        else:
            assert np.isfinite(gn) and gn > 0
            good += 1
            if good == 3:
                scale, good = scale * 2.0, 0
            assert now == scale, (it, now, scale)
            if first_clean is None:
                first_clean = it
                moved = sum(float((p.detach() - p0[k]).abs().max()) > 0 for k, p in m.named_parameters() if k.startswith(("decoder", "decode_head")) and "pos_embed" not in k)
                assert moved >= 40
                # the unscaled gradient of that step against the oracle at the untouched initial parameters
                _, _, rg = R.loss_and_grads(sd, imgs, boxes, gt, step.mask.cpu().numpy(), 3, NAME)
                k = "decode_head0.0.weight"
                got = step.eng.gview(k).detach().cpu().double() / (now if good else now / 2.0)
                ref = rg[k].double()
                assert ((got * ref).sum() / (got.norm() * ref.norm())).item() > 0.999
    assert first_clean is not None and 10 <= first_clean <= 45, first_clean       # 2^40 -> the first scale whose backward stays finite
    assert skipped >= first_clean and all(torch.isfinite(p).all() for p in m.parameters())


def test_fp16_scaler_state_roundtrips_through_the_checkpoint_and_gradscaler():
    """the fp16 step's GradScaler lives on the device; its state must travel with the checkpoint like the reference's
    (util/misc.py:316 saves loss_scaler.state_dict(), :419 reloads it).  The exported entry loads into a real torch GradScaler and back
    into a fresh step (scale, growth tracker, interval), a scaler state written by torch loads too, the loss-mask stream's position rides
    in the optimizer entry's extras, and a bf16 step exports None (key omitted)."""
    from countr_amd.trainer import FinetuneStep
    m, _sd = make("fp16")
    step = FinetuneStep(m, batch=2, lr=1e-3, eps=1e-4, use_graph=True, defer_optimizer=True, mask_seed=3)
    step.amp[0], step.amp[4] = 2.0 ** 12, 5.0
    for it in range(3):
        imgs, boxes, gt, _mask = W.make_inputs(batch=2, shots=3, seed=310 + it)
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt)), None, 3)
        step.step(3)
    st = step.scaler_state()
    assert set(st) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    assert st["growth_interval"] == 5 and st["scale"] == step.loss_scale and 0 <= st["_growth_tracker"] < 5
    assert st["_growth_tracker"] + 5 * 0 == int(step.amp[1].item())
    gs = torch.amp.GradScaler("cuda")
    gs.load_state_dict(dict(st))                       # the reference's resume path (raises on a malformed / empty state)
    assert gs.state_dict()["scale"] == st["scale"] and gs.state_dict()["_growth_tracker"] == st["_growth_tracker"]
    opt = step.optimizer_state()
    assert opt["countr_amd"]["mask_draws"] == 3
    m2, _ = make("fp16")
    step2 = FinetuneStep(m2, batch=2, lr=1e-3, eps=1e-4, use_graph=True, defer_optimizer=True, mask_seed=3)
    assert step2.load_scaler_state(st) and step2.load_optimizer_state(opt)
    assert step2.scaler_state() == st and step2.pro.draws == 3
    assert step2.load_scaler_state(gs.state_dict())    # a state torch wrote
    assert not step2.load_scaler_state({})
    with pytest.raises(ValueError):
        step2.load_scaler_state(dict(st, growth_factor=3.0))
    mb, _ = make("bf16")
    assert FinetuneStep(mb, batch=2, use_graph=False).scaler_state() is None


@pytest.mark.parametrize("precision,accum", [("bf16", 1), ("fp16", 1), ("bf16", 2)])
def test_pipelined_encoder_is_bit_identical(precision, accum):
    """FinetuneStep(pipeline_encoder=True): the frozen-encoder forward of batch k + 1 (models_mae_cross.py:203-205: no_grad, frozen
    weights -- nothing in it depends on step k) runs on its own lane of step k's graph beside batch k's decoder forward, loss, backward
    and AdamW; the next step's prologue copies the latent into place.  Same launches, same data: every step's loss / counts, every
    parameter, both AdamW moments and the gradient norm are BIT-identical to the plain step -- over a shot schedule that changes the plan
    (and with it the encoder's scratch buffers) from step to step, with accumulation windows, with a step whose images were NOT announced
    (it computes its own encoder forward: 'coldnext'), with an announced batch that never comes (the waiting latent is dropped), and with
    the last step announcing nothing.  The modes the steps ran in are checked too, so the test cannot pass by never pipelining."""
    from countr_amd.trainer import FinetuneStep
    sched = [3, 0, 3, 1, 0, 3, 3, 2, 1]
    batches = []
    for it in range(len(sched)):
        imgs, boxes, gt, _mask = W.make_inputs(batch=2, shots=3, seed=400 + it)
        batches.append(tuple(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt)))
    decoy = batches[0][0].clone()
    res = {}
    for pipe in (False, True):
        m, sd = make(precision)
        step = FinetuneStep(m, batch=2, lr=1e-3, eps=1e-4, use_graph=True, accum_iter=accum, pipeline_encoder=pipe, mask_seed=5)
        sums, modes = [], []
        for it, S in enumerate(sched):
            imgs, boxes, gt = batches[it]
            if it == 4:
                nxt = decoy                           # announces a batch that never comes: step 5 must NOT use that latent
            elif it == 6 or it + 1 == len(sched):
                nxt = None                            # a gap: step 7 computes its own encoder forward; the last step announces nothing
            else:
                nxt = batches[it + 1][0]
            step.load(imgs, boxes, gt, None, S, next_imgs=nxt)
            modes.append(step._pipe_mode)
            sums.append(step.step(S, lr=1e-3 * (1 + 0.1 * it)).clone())
        gn = step.grad_norm().clone()
        torch.cuda.synchronize()
        res[pipe] = (torch.stack(sums), {k: p.detach().clone() for k, p in m.named_parameters()}, step.eng.M.clone(), step.eng.V.clone(), gn, modes)
    a, b = res[False], res[True]
    assert a[5] == ["plain"] * len(sched)
    assert b[5] == ["coldnext", "steady", "steady", "steady", "steady", "coldnext", "last", "coldnext", "last"], b[5]
    assert torch.equal(a[0], b[0]), (a[0] - b[0]).abs().max()
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    moved = sum(float((a[1][k].cpu() - torch.from_numpy(sd[k])).abs().max()) > 0 for k in a[1] if k.startswith(("decoder", "decode_head", "shot_token")) and "pos_embed" not in k)
    assert moved >= 40


def test_pipelined_encoder_falls_back_where_it_cannot_run():
    """fp32 parity mode (its unfused attention shares scratch with the decoder's) and eager steps run the plain step whatever is
    announced; drop_lookahead() makes the next step compute its own encoder forward."""
    from countr_amd.trainer import FinetuneStep
    imgs, boxes, gt, _mask = W.make_inputs(batch=2, shots=3, seed=410)
    t = tuple(torch.from_numpy(a).cuda() for a in (imgs, boxes, gt))
    for precision, graph in (("fp32", True), ("bf16", False)):
        m, _sd = make(precision)
        step = FinetuneStep(m, batch=2, use_graph=graph, pipeline_encoder=True)
        step.load(*t, None, 3, next_imgs=t[0])
        assert step._pipe_mode == "plain"
        step.step(3)
    m, _sd = make("bf16")
    step = FinetuneStep(m, batch=2, use_graph=True, pipeline_encoder=True)
    step.load(*t, None, 3, next_imgs=t[0])
    assert step._pipe_mode == "coldnext"
    a = step.step(3).clone()
    step.drop_lookahead()
    step.load(*t, None, 3, next_imgs=t[0])
    assert step._pipe_mode == "coldnext"
    step.step(3)
    step.load(*t, None, 3)
    assert step._pipe_mode == "last"
    step.step(3)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
