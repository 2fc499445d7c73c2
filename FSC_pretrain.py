#!/usr/bin/env python3
"""MAE pretraining CLI with the reference's flags (FSC_pretrain.py:33-110), running the MI355X engine.

The hot loop (:254-310) is the fused PretrainStep: random masking + forward + all-patch pixel MSE + full backward + RCCL
gradient all-reduce + AdamW, graph-captured, bf16 (no GradScaler), LR schedule per iteration as util/lr_sched.py.
Launch one process per GPU: `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 FSC_pretrain.py ...`.
Data: with the FSC147 files present images come from countr_amd/data/fsc147.py (PIL restatement of ResizePreTrainImage +
RandomResizedCrop/flip, util/FSC147.py:58-83,369-374); `--synthetic_steps K` trains on synthetic 384x384 images instead (K
iterations per epoch) and is the automatic fallback when the dataset is absent.  TensorBoard / W&B logging is
replaced by JSON lines on stdout and log.txt."""
import argparse
import json
import os
import time

import numpy as np
import torch

import models_mae_noct
from countr_amd.trainer import PretrainStep
from countr_amd.util import lr_sched, misc


def get_args_parser():
    p = argparse.ArgumentParser("MAE pre-training (MI355X engine)", add_help=True)
    p.add_argument("--batch_size", default=8, type=int, help="batch size per GPU")
    p.add_argument("--epochs", default=200, type=int)
    p.add_argument("--accum_iter", default=1, type=int)
    p.add_argument("--model", default="mae_vit_base_patch16", type=str)
    p.add_argument("--mask_ratio", default=0.5, type=float)
    p.add_argument("--norm_pix_loss", action="store_true")
    p.add_argument("--weight_decay", type=float, default=0.05)
    p.add_argument("--lr", type=float, default=None)
    p.add_argument("--blr", type=float, default=1e-3)
    p.add_argument("--min_lr", type=float, default=0.0)
    p.add_argument("--warmup_epochs", type=int, default=10)
    p.add_argument("--data_path", default="./data/FSC147/", type=str)
    p.add_argument("--anno_file", default="annotation_FSC147_384.json", type=str)
    p.add_argument("--data_split_file", default="Train_Test_Val_FSC_147.json", type=str)
    p.add_argument("--im_dir", default="images_384_VarV2", type=str)
    p.add_argument("--gt_dir", default="gt_density_map_adaptive_384_VarV2", type=str)
    p.add_argument("--output_dir", default="./data/out/pre_4_dir")
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--resume", default="./weights/mae_pretrain_vit_base_full.pth")
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--num_workers", default=10, type=int)
    p.add_argument("--pin_mem", action="store_true")
    p.add_argument("--no_pin_mem", action="store_false", dest="pin_mem")
    p.set_defaults(pin_mem=True)
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local_rank", default=-1, type=int)
    p.add_argument("--dist_on_itp", action="store_true")
    p.add_argument("--dist_url", default="env://")
    p.add_argument("--log_dir", default="./logs/pre_4_dir")
    p.add_argument("--title", default="CounTR_pretraining", type=str)
    p.add_argument("--wandb", default=None, type=str)
    p.add_argument("--team", default=None, type=str)
    p.add_argument("--wandb_id", default=None, type=str)
    # additions
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    p.add_argument("--synthetic_steps", default=0, type=int,
                   help="K > 0: K iterations per epoch on synthetic images; 0: FSC147 from --data_path (synthetic, 50 it/epoch, if absent)")
    p.add_argument("--log_every", default=20, type=int, help="iterations between loss reports (each report is a host sync)")
    return p


def main(args):
    misc.init_distributed_mode(args)
    device = torch.device("cuda", args.gpu)
    torch.cuda.set_device(device)
    seed = args.seed + misc.get_rank()          # FSC_pretrain.py:150-152
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = models_mae_noct.__dict__[args.model](norm_pix_loss=args.norm_pix_loss, precision=args.precision)
    ckpt = misc.load_model(args, model)
    model.to(device).train()
    eff_batch = args.batch_size * args.accum_iter * misc.get_world_size()
    if args.lr is None:
        args.lr = args.blr * eff_batch / 256     # :211-212
    print("actual lr: %.2e, effective batch size: %d" % (args.lr, eff_batch))
    step = PretrainStep(model, batch=args.batch_size, mask_ratio=args.mask_ratio, lr=args.lr, weight_decay=args.weight_decay,
                        betas=(0.9, 0.95), accum_iter=args.accum_iter)
    if ckpt is not None and "optimizer" in ckpt and "epoch" in ckpt:      # util/misc.py:352-361
        step.load_optimizer_state(ckpt["optimizer"])      # torch.optim.AdamW state_dict (ours or the reference's) or the older flat form; raises otherwise
        args.start_epoch = ckpt["epoch"] + 1
        if "scaler" in ckpt:                      # util/misc.py:359-360
            step.load_scaler_state(ckpt["scaler"])
        print("With optim & sched!")
    from countr_amd.data import fsc147
    loader = None
    if args.synthetic_steps <= 0 and fsc147.available(args):
        ds = fsc147.PretrainData(args)
        sampler = torch.utils.data.DistributedSampler(ds, num_replicas=misc.get_world_size(), rank=misc.get_rank(), shuffle=True)
        loader = torch.utils.data.DataLoader(ds, sampler=sampler, batch_size=args.batch_size, num_workers=args.num_workers,
                                             pin_memory=args.pin_mem, drop_last=True)   # drop_last: the fused step has a static batch
        n_iter = len(loader)
    else:
        if args.synthetic_steps <= 0:
            print("FSC147 not found under %s: training on synthetic images" % args.data_path)
        n_iter = args.synthetic_steps if args.synthetic_steps > 0 else 50
    g = torch.Generator(device=device).manual_seed(seed)
    t_start = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        losses = []
        lr = args.lr
        if loader is not None:
            loader.sampler.set_epoch(epoch)                                             # :236-237
        it_data = iter(loader) if loader is not None else None
        with step.on_stream():      # one stream for the loop's device work and the step (no cross-queue hand-over per step)
            for it in range(n_iter):
                if it % args.accum_iter == 0:                                               # :258-259 (per accumulation window)
                    lr = lr_sched.adjust_learning_rate(None, it / n_iter + epoch, args)
                if it_data is not None:
                    imgs = next(it_data)     # host tensor: load() stages it over PCIe on a copy stream while the previous step computes
                else:
                    imgs = torch.rand(args.batch_size, 3, 384, 384, device=device, generator=g)
                step.load(imgs)
                loss = step.step(lr=lr)
                if (it + 1) % args.log_every == 0 or it + 1 == n_iter:
                    lv = misc.all_reduce_mean(float(loss.item()))                           # the only host sync (:265, :310)
                    if not np.isfinite(lv):
                        raise SystemExit("Loss is %s, stopping training" % lv)              # :292-294
                    losses.append(lv)
                    if misc.is_main_process():
                        print(json.dumps({"epoch": epoch, "it": it + 1, "loss": lv, "lr": lr}))
        opt_state = step.optimizer_state()
        if args.output_dir and (epoch % 100 == 0 or epoch + 1 == args.epochs):         # :327-329
            misc.save_model(args, epoch, model, opt_state, suffix="pretraining_%d" % epoch, scaler_state=step.scaler_state())
        if args.output_dir and misc.is_main_process():
            os.makedirs(args.output_dir, exist_ok=True)
            with open(os.path.join(args.output_dir, "log.txt"), "a", encoding="utf-8") as f:
                f.write(json.dumps({"train_loss": float(np.mean(losses)), "train_lr": lr, "epoch": epoch}) + "\n")
    print("Training time %.1fs" % (time.time() - t_start))


if __name__ == "__main__":
    main(get_args_parser().parse_args())
