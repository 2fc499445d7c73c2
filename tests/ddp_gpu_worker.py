"""Worker of tests/test_ddp_gpu.py: one rank of a world-2 job sharing the single GPU (gloo backend: RCCL refuses two ranks on one
device).  Runs FinetuneStep (or PretrainStep) under hipGraph replay on its half of a global batch and saves its parameters.
Launched by `python -m torch.distributed.run --nproc-per-node 2 tests/ddp_gpu_worker.py <finetune|pretrain> <outdir>`."""
import os
import sys
from functools import partial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import weights as W  # noqa: E402

SHOTS = [3, 0, 3, 1]
# per-rank shot_num (reference semantics, FSC_finetune_cross.py:276-284): PRS[it][rank]
PRS = [[3, 0], [0, 0], [1, 3], [0, 2]]


def finetune_model(precision="fp32"):
    from countr_amd.models_mae_cross import SupervisedMAE
    p, D, depth, H, Dd, ddepth, Hd = W.CONFIGS["tiny_test"]
    sd = W.make_state_dict("tiny_test", seed=3)
    m = SupervisedMAE(patch_size=p, embed_dim=D, depth=depth, num_heads=H, decoder_embed_dim=Dd, decoder_depth=ddepth,
                      decoder_num_heads=Hd, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to("cuda").train()


def pretrain_model(precision="fp32"):
    from countr_amd.models_mae_noct import MaskedAutoencoderViTNoCT
    name = "tiny_test"
    p, D, depth, H, Dd, ddepth, Hd = W.MAE_CONFIGS[name]
    sd = W.make_state_dict_mae(name, seed=4)
    m = MaskedAutoencoderViTNoCT(patch_size=p, embed_dim=D, depth=depth, num_heads=H, decoder_embed_dim=Dd, decoder_depth=ddepth,
                                 decoder_num_heads=Hd, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to("cuda").train()


def run_finetune(rank, world, per_rank, use_graph=True):
    from countr_amd.trainer import FinetuneStep
    m = finetune_model()
    step = FinetuneStep(m, batch=per_rank, lr=1e-3, weight_decay=0.05, eps=1e-4, use_graph=use_graph)
    losses = []
    for it, S in enumerate(SHOTS):
        imgs, boxes, gt, mask = W.make_inputs(batch=per_rank * world, shots=3, seed=40 + it)
        sl = slice(rank * per_rank, (rank + 1) * per_rank)
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs[sl], boxes[sl], gt[sl], mask)), S)
        losses.append(step.step(S)[0].item())
    torch.cuda.synchronize()
    m._graph_kinds = sorted({k[0] for k in step.graphs})
    return m, losses


def run_finetune_per_rank_shot(rank, world, per_rank, use_graph=True):
    """Every rank steps ITS half of the batch with ITS OWN shot_num; the last iteration lets the ranks all-gather the counts."""
    from countr_amd.trainer import FinetuneStep
    m = finetune_model()
    step = FinetuneStep(m, batch=per_rank, lr=1e-3, weight_decay=0.05, eps=1e-4, use_graph=use_graph, per_rank_shot=True)
    losses = []
    for it, shots in enumerate(PRS):
        S = shots[rank]
        imgs, boxes, gt, mask = W.make_inputs(batch=per_rank * world, shots=3, seed=140 + it)
        sl = slice(rank * per_rank, (rank + 1) * per_rank)
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs[sl], boxes[sl], gt[sl], mask)), S)
        losses.append(step.step(S, shots_all=shots if it + 1 < len(PRS) else None)[0].item())
    torch.cuda.synchronize()
    m._graph_kinds = sorted({k[0] for k in step.graphs})
    m._group_steps = list(step.eng.group_steps)
    return m, losses


def run_finetune_real(rank, world, per_rank=8, use_graph=True):
    """BASELINE config 3's per-GPU work: ViT-B/16, bf16, 8 images per rank, hipGraph replay, shot schedule with every bucket case."""
    import models_mae_cross as mm
    from countr_amd.trainer import FinetuneStep
    m = mm.__dict__["mae_vit_base_patch16"](precision="bf16")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_state_dict("mae_vit_base_patch16", seed=0).items()}, strict=True)
    m.to("cuda").train()
    step = FinetuneStep(m, batch=per_rank, lr=1e-4, weight_decay=0.05, use_graph=use_graph)
    losses = []
    for it, S in enumerate(SHOTS + [3, 3]):
        imgs, boxes, gt, mask = W.make_inputs(batch=per_rank * world, shots=3, seed=60 + it)
        sl = slice(rank * per_rank, (rank + 1) * per_rank)
        step.load(*(torch.from_numpy(a).cuda() for a in (imgs[sl], boxes[sl], gt[sl], mask)), S)
        losses.append(step.step(S)[0].item())
    torch.cuda.synchronize()
    return m, losses


def run_pretrain(rank, world, per_rank, use_graph=True):
    from countr_amd.trainer import PretrainStep
    m = pretrain_model()
    step = PretrainStep(m, batch=per_rank, mask_ratio=0.5, lr=1e-3, weight_decay=0.05, eps=1e-4, use_graph=use_graph)
    losses = []
    for it in range(3):
        rs = np.random.RandomState(90 + it)
        imgs = rs.uniform(0, 1, size=(per_rank * world, 3, 384, 384)).astype(np.float32)
        ids = np.stack([rs.permutation(m.patch_embed.num_patches) for _ in range(per_rank * world)])
        sl = slice(rank * per_rank, (rank + 1) * per_rank)
        step.load(torch.from_numpy(imgs[sl]).cuda(), ids_shuffle=torch.from_numpy(ids[sl]).cuda())
        losses.append(step.step().item())
    torch.cuda.synchronize()
    m._graph_kinds = sorted({k[0] for k in step.graphs})
    return m, losses


if __name__ == "__main__":
    what, outdir = sys.argv[1], sys.argv[2]
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    if what == "finetune_real":
        m, losses = run_finetune_real(rank, world)
        keep = lambda k: k.startswith(("decoder", "decode_head", "shot_token"))
    elif what == "finetune_prs":
        m, losses = run_finetune_per_rank_shot(rank, world, 2)
        keep = lambda k: True
    else:
        m, losses = (run_finetune if what == "finetune" else run_pretrain)(rank, world, 2 if world > 1 else 4)
        keep = lambda k: True
    torch.save({"params": {k: p.detach().cpu() for k, p in m.named_parameters() if keep(k)}, "losses": losses,
                "graph_kinds": getattr(m, "_graph_kinds", None), "group_steps": getattr(m, "_group_steps", None)},
               os.path.join(outdir, "%s_rank%d.pt" % (what, rank)))
    dist.barrier()
    dist.destroy_process_group()
