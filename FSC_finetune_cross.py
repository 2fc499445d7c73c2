#!/usr/bin/env python3
"""Finetuning CLI with the reference's flags (FSC_finetune_cross.py:30-107), running the MI355X engine.

The hot loop (:265-319) is the fused FinetuneStep: forward + masked-MSE + decoder backward + RCCL gradient all-reduce +
AdamW, graph-captured, bf16 (no GradScaler), LR schedule per iteration as util/lr_sched.py.  Launch one process per GPU
with `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 FSC_finetune_cross.py ...`.
Data: with the FSC147 files present (--data_path/--anno_file/--data_split_file/--im_dir) batches come from
countr_amd/data/fsc147.py (PIL/scipy restatement of the reference's non-augmented train transform; the imgaug/cv2 augmentation
pipeline of util/FSC147.py is out of scope); `--synthetic_steps K` trains on synthetic FSC147-shaped batches instead (K
iterations per epoch) and is the automatic fallback when the dataset is absent."""
import argparse
import json
import time

import numpy as np
import torch

import models_mae_cross
from countr_amd.parallel import shared_shot_num
from countr_amd.synthetic import make_batch
from countr_amd.trainer import FinetuneStep
from countr_amd.util import lr_sched, misc


def get_args_parser():
    p = argparse.ArgumentParser("CounTR finetuning (MI355X engine)", add_help=True)
    p.add_argument("--batch_size", default=26, type=int, help="batch size per GPU")
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
    p.add_argument("--class_file", default="ImageClasses_FSC147.txt", type=str)
    p.add_argument("--im_dir", default="images_384_VarV2", type=str)
    p.add_argument("--output_dir", default="./data/out/fim6_dir")
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--resume", default="./data/out/pre_4_dir/checkpoint-300.pth")
    p.add_argument("--do_resume", action="store_true")
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--num_workers", default=10, type=int)
    p.add_argument("--pin_mem", action="store_true")
    p.add_argument("--no_pin_mem", action="store_false", dest="pin_mem")
    p.set_defaults(pin_mem=True)
    p.add_argument("--do_aug", action="store_true")
    p.add_argument("--no_do_aug", action="store_false", dest="do_aug")
    p.set_defaults(do_aug=True)
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local_rank", default=-1, type=int)
    p.add_argument("--dist_on_itp", action="store_true")
    p.add_argument("--dist_url", default="env://")
    p.add_argument("--title", default="CounTR_finetuning", type=str)
    p.add_argument("--wandb", default=None, type=str)
    p.add_argument("--team", default=None, type=str)
    p.add_argument("--wandb_id", default=None, type=str)
    # additions
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--synthetic_steps", default=0, type=int,
                   help="K > 0: K iterations per epoch on synthetic batches; 0: FSC147 from --data_path (synthetic, 50 it/epoch, if absent)")
    p.add_argument("--log_every", default=50, type=int, help="iterations between loss reports (each report is a host sync)")
    return p


def main(args):
    misc.init_distributed_mode(args)
    device = torch.device("cuda", args.gpu)
    torch.cuda.set_device(device)
    seed = args.seed + misc.get_rank()          # FSC_finetune_cross.py:168-170
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = models_mae_cross.__dict__[args.model](norm_pix_loss=args.norm_pix_loss, precision=args.precision)
    ckpt = misc.load_model_FSC(args, model)
    model.to(device).train()
    eff_batch = args.batch_size * args.accum_iter * misc.get_world_size()
    if args.lr is None:
        args.lr = args.blr * eff_batch / 256     # :218-221
    print("actual lr: %.2e, effective batch size: %d" % (args.lr, eff_batch))
    step = FinetuneStep(model, batch=args.batch_size, lr=args.lr, weight_decay=args.weight_decay, betas=(0.9, 0.95),
                        accum_iter=args.accum_iter)
    if ckpt is not None and args.do_resume and "epoch" in ckpt:      # util/misc.py:400-421: optimizer / epoch only with --do_resume
        args.start_epoch = ckpt["epoch"] + 1
        opt = ckpt.get("optimizer")
        if isinstance(opt, dict) and opt.get("exp_avg") is not None and opt["exp_avg"].numel() == step.eng.G.numel():
            step.eng.M = opt["exp_avg"].to(device)               # our flat AdamW state (a reference optimizer dict is per-tensor
            step.eng.V = opt["exp_avg_sq"].to(device)            # and in a different order: not convertible without its param groups)
            step.eng.step_count = int(opt["step"])
            print("With optim & sched!")
    from countr_amd.data import fsc147
    loader = None
    if args.synthetic_steps <= 0 and fsc147.available(args):
        ds = fsc147.TrainData(args, split="train", do_aug=args.do_aug)
        sampler = torch.utils.data.DistributedSampler(ds, num_replicas=misc.get_world_size(), rank=misc.get_rank(), shuffle=True)
        loader = torch.utils.data.DataLoader(ds, sampler=sampler, batch_size=args.batch_size, num_workers=args.num_workers,
                                             pin_memory=args.pin_mem, drop_last=True)   # drop_last: the fused step has a static batch
        n_iter = len(loader)
    else:
        if args.synthetic_steps <= 0:
            print("FSC147 not found under %s: training on synthetic batches" % args.data_path)
        n_iter = args.synthetic_steps if args.synthetic_steps > 0 else 50
    loss_mask_gen = torch.Generator(device=device).manual_seed(seed)
    t_start = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        mae = rmse = 0.0
        if loader is not None:
            loader.sampler.set_epoch(epoch)                                             # :260-261
        it_data = iter(loader) if loader is not None else None
        for it in range(n_iter):
            if it % args.accum_iter == 0:                                               # :270-271 (per accumulation window)
                lr = lr_sched.adjust_learning_rate(None, it / n_iter + epoch, args)
            S = shared_shot_num(epoch * n_iter + it, seed=args.seed)                   # :278-284 (shared across ranks)
            if it_data is not None:
                imgs, gt, _n, boxes, _pos, _m, _ids = next(it_data)
                # loss mask: Bernoulli(0.8) per pixel, one mask per batch (FSC_finetune_cross.py:290-292)
                mask = (torch.rand(384, 384, device=device, generator=loss_mask_gen) < 0.8).float()
                # host tensors go straight to load(): it stages them over PCIe on a copy stream while the previous step computes
            else:
                imgs, boxes, gt, mask = make_batch(args.batch_size, shots=3, seed=seed * 100003 + epoch * n_iter + it, device=device)
            step.load(imgs, boxes, gt, mask, S)
            sums = step.step(S, lr=lr)
            if (it + 1) % args.log_every == 0 or it + 1 == n_iter:
                s = sums.float().cpu().numpy()                                          # the only host sync
                B = args.batch_size
                err = np.abs(s[1:1 + B] - s[1 + B:1 + 2 * B])
                mae += err.mean(); rmse += (err ** 2).mean()
                loss = misc.all_reduce_mean(float(s[0]))                                # :319
                if not np.isfinite(loss):
                    raise SystemExit("Loss is %s, stopping training" % loss)            # :308-310
                if misc.is_main_process():
                    print(json.dumps({"epoch": epoch, "it": it + 1, "loss": loss, "lr": lr, "shot_num": S,
                                      "batch_MAE": float(err.mean())}))
        opt_state = {"step": step.eng.step_count, "exp_avg": step.eng.M.cpu() if step.eng.M is not None else None,
                     "exp_avg_sq": step.eng.V.cpu() if step.eng.V is not None else None}
        misc.save_model(args, epoch, model, opt_state, suffix="finetuning_last")
        if args.output_dir and (epoch % 50 == 0 or epoch + 1 == args.epochs) and epoch != 0:
            misc.save_model(args, epoch, model, opt_state, suffix="finetuning_%d" % epoch)
    print("Training time %.1fs" % (time.time() - t_start))


if __name__ == "__main__":
    main(get_args_parser().parse_args())
