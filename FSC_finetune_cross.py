#!/usr/bin/env python3
"""Finetuning CLI with the reference's flags (FSC_finetune_cross.py:30-107), running the MI355X engine.

The hot loop (:265-319) is the fused FinetuneStep: forward + masked-MSE + decoder backward + RCCL gradient all-reduce +
AdamW, graph-captured, bf16 (no GradScaler), LR schedule per iteration as util/lr_sched.py.  Launch one process per GPU
with `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 FSC_finetune_cross.py ...`.
Data: with the FSC147 files present (--data_path/--anno_file/--data_split_file/--im_dir) batches come from
countr_amd/data/fsc147.py (PIL/scipy/torch restatement of util/FSC147.py's train transform, with the noise / colour jitter / blur /
affine / flip / mosaic augmentation when --do_aug is on, the reference's default; --class_file is needed then);
`--synthetic_steps K` trains on synthetic FSC147-shaped batches instead (K iterations per epoch) and is the automatic fallback when
the dataset is absent."""
import argparse
import json
import random
import time

import numpy as np
import torch

import models_mae_cross
from countr_amd.parallel import rank_shot_nums, shared_shot_num
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
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    p.add_argument("--synthetic_steps", default=0, type=int,
                   help="K > 0: K iterations per epoch on synthetic batches; 0: FSC147 from --data_path (synthetic, 50 it/epoch, if absent)")
    p.add_argument("--per_rank_shot", action="store_true",
                   help="reference semantics for N > 1: every rank draws its own shot_num per iteration (FSC_finetune_cross.py:276-284; "
                        "default: one draw shared by all ranks)")
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
                        accum_iter=args.accum_iter, per_rank_shot=args.per_rank_shot, mask_seed=seed,   # seed = args.seed + rank (:168)
                        pipeline_encoder=True,      # the next batch's frozen-encoder forward beside this batch's decoder side (bit-identical)
                        defer_optimizer=True)       # (what the step does where the pipelined form cannot run: fp32, host-issued collectives)
    if ckpt is not None and args.do_resume and "optimizer" in ckpt and "epoch" in ckpt:      # util/misc.py:415: all three, else skipped
        # (raises when the entry EXISTS and fits neither this model's torch.optim.AdamW layout nor the older flat form: continuing late
        # in the LR schedule with zeroed moments would be a silent restart of the bias correction)
        step.load_optimizer_state(ckpt["optimizer"])
        args.start_epoch = ckpt["epoch"] + 1
        if "scaler" in ckpt:                      # util/misc.py:418-419 (fp16 mode: the GradScaler's scale / growth count; else ignored)
            step.load_scaler_state(ckpt["scaler"])
        print("With optim & sched!")
    from countr_amd.data import fsc147
    loader = val_loader = None
    if args.synthetic_steps <= 0 and fsc147.available(args):
        ds = fsc147.TrainData(args, split="train", do_aug=args.do_aug)
        sampler = torch.utils.data.DistributedSampler(ds, num_replicas=misc.get_world_size(), rank=misc.get_rank(), shuffle=True)
        loader = torch.utils.data.DataLoader(ds, sampler=sampler, batch_size=args.batch_size, num_workers=args.num_workers,
                                             pin_memory=args.pin_mem, drop_last=True)   # drop_last: the fused step has a static batch
        n_iter = len(loader)
        dsv = fsc147.TrainData(args, split="val", do_aug=False)                         # :146-155, :185-191
        vsampler = torch.utils.data.DistributedSampler(dsv, num_replicas=misc.get_world_size(), rank=misc.get_rank(), shuffle=True)
        val_loader = torch.utils.data.DataLoader(dsv, sampler=vsampler, batch_size=args.batch_size, num_workers=args.num_workers,
                                                 pin_memory=args.pin_mem, drop_last=False)
        n_val = len(val_loader)
    else:
        if args.synthetic_steps <= 0:
            print("FSC147 not found under %s: training on synthetic batches" % args.data_path)
        n_iter = args.synthetic_steps if args.synthetic_steps > 0 else 50
        n_val = max(1, n_iter // 4)
    # mosaics only come out of the augmented loader: without it no rank ever bans shot_num 0 and there is nothing to agree on
    flag_group = (torch.distributed.new_group(backend="gloo") if misc.get_world_size() > 1 and loader is not None and args.do_aug
                  else None)
    val_rng = random.Random(seed + 7919)      # the reference draws the validation shot_num per rank from `random` (:338)
    min_MAE = 99999.0                         # :253
    B = args.batch_size
    t_start = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        # running MAE / RMSE of the epoch (:256-257, :303-304): accumulated on the device, read once per epoch
        train_acc = torch.zeros(2, dtype=torch.float64, device=device)
        if loader is not None:
            loader.sampler.set_epoch(epoch)                                             # :260-261
        it_data = iter(loader) if loader is not None else None
        # the loop's own device work (mask draw, error sums) runs on the step's stream: from another stream every step pays two
        # cross-queue hand-overs (inputs ready -> step, step done -> caller), ~50 us of idle GPU per step on MI355X
        # one batch of look-ahead: the step is told the NEXT batch's images (load(..., next_imgs=)) and runs their frozen-encoder forward
        # beside this batch's decoder side; the DataLoader's workers have that batch ready anyway

        def fetch(it):
            if it >= n_iter:
                return None
            if it_data is not None:
                return next(it_data)
            return make_batch(B, shots=3, seed=seed * 100003 + epoch * n_iter + it, device=device)
        ahead = fetch(0)
        with step.on_stream():
            for it in range(n_iter):
                if it % args.accum_iter == 0:                                               # :270-271 (per accumulation window)
                    lr = lr_sched.adjust_learning_rate(None, it / n_iter + epoch, args)
                item, ahead = ahead, fetch(it + 1)
                next_imgs = ahead[0] if ahead is not None else None
                if it_data is not None:
                    imgs, gt, _n, boxes, _pos, m_flag, _ids = item
                    # loss mask: Bernoulli(0.8) per pixel, one mask per batch (FSC_finetune_cross.py:290-292)
                    mask = None      # drawn by the step itself (FinetuneStep(mask_seed=seed): Philox stream keyed by seed + rank)
                    # host tensors go straight to load(): it stages them over PCIe on a copy stream while the previous step computes
                    mosaic = int(torch.as_tensor(m_flag).sum().item()) != 0
                else:
                    imgs, boxes, gt, mask = item
                    mosaic = False
                world = misc.get_world_size()
                mosaics = [mosaic] * world
                if flag_group is not None:               # Host data, host collective (gloo): reading a device flag back would drain the
                    flags = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]                         # GPU queue every step
                    torch.distributed.all_gather(flags, torch.tensor([int(mosaic)]), group=flag_group)
                    mosaics = [bool(f.item()) for f in flags]
                # :276-284: "If there is at least one image in the batch using Type 2 Mosaic, 0-shot is banned."
                if args.per_rank_shot:                   # the reference: every rank draws for ITS batch, the ban is ITS batch's
                    shots_all = rank_shot_nums(epoch * n_iter + it, world, seed=args.seed, allow_zero=[not m_ for m_ in mosaics])
                    S = shots_all[misc.get_rank()]
                else:                                    # one draw for all ranks, so the ban is shared too: any rank with a Type-2 mosaic
                    shots_all = None
                    S = shared_shot_num(epoch * n_iter + it, seed=args.seed, allow_zero=not any(mosaics))
                step.load(imgs, boxes, gt, mask, S, next_imgs=next_imgs)
                sums = step.step(S, lr=lr, shots_all=shots_all)
                err = (sums[1:1 + B] - sums[1 + B:1 + 2 * B]).abs().double()                # :296-304, no host sync
                train_acc[0] += err.mean()
                train_acc[1] += (err ** 2).mean()
                if (it + 1) % args.log_every == 0 or it + 1 == n_iter:
                    s = sums.float().cpu().numpy()                                          # host sync (logging only)
                    loss = misc.all_reduce_mean(float(s[0]))                                # :319
                    if not np.isfinite(loss):
                        raise SystemExit("Loss is %s, stopping training" % loss)            # :308-310
                    # grad_norm() applies a deferred AdamW update (flush): EVERY rank does it at the log step, so that all ranks keep
                    # replaying the same graph keys in step -- a flush on rank 0 alone would send only rank 0 through an eager update and
                    # (the first time) a late capture while its peers replay captured RCCL nodes
                    gn = step.grad_norm()
                    if misc.is_main_process():
                        print(json.dumps({"epoch": epoch, "it": it + 1, "loss": loss, "lr": lr, "shot_num": S,
                                          "batch_MAE": float(np.abs(s[1:1 + B] - s[1 + B:1 + 2 * B]).mean()),
                                          "grad_norm": float(gn.item()) if gn is not None else None}))
        # ---- evaluation on the validation split (:329-350): no_grad forward, shot_num drawn per batch, MAE / RMSE / NAE of the counts
        step.flush()       # defer_optimizer: the epoch's last update is applied before anything reads the parameters (validation, checkpoint)
        val = evaluate(model, val_loader, n_val, B, device, val_rng, seed, epoch)
        train_mae, train_mse = (train_acc / n_iter).tolist()
        opt_state = step.optimizer_state()
        sc_state = step.scaler_state()            # fp16: GradScaler.state_dict() of the device-side scaler; None otherwise (key omitted)
        if args.output_dir and (epoch % 50 == 0 or epoch + 1 == args.epochs) and epoch != 0:      # :408-412
            misc.save_model(args, epoch, model, opt_state, suffix="finetuning_%d" % epoch, scaler_state=sc_state)
        misc.save_model(args, epoch, model, opt_state, suffix="finetuning_last", scaler_state=sc_state)   # :413-415
        if args.output_dir and val["MAE"] < min_MAE:                                              # :416-420
            min_MAE = val["MAE"]
            misc.save_model(args, epoch, model, opt_state, suffix="finetuning_minMAE", scaler_state=sc_state)
        print("[Train Epoch #%d] - MAE: %5.2f, RMSE: %5.2f" % (epoch, train_mae, train_mse ** 0.5), flush=True)             # :422
        print("[Val Epoch #%d] - MAE: %5.2f, RMSE: %5.2f, NAE: %5.2f" % (epoch, val["MAE"], val["RMSE"], val["NAE"]), flush=True)  # :423
    print("Training time %.1fs" % (time.time() - t_start))


def evaluate(model, val_loader, n_val, B, device, rng, seed, epoch):
    """Validation pass of FSC_finetune_cross.py:329-350 (per rank, like the reference: its val metrics are not all-reduced)."""
    was_training = model.training
    model.eval()
    acc = torch.zeros(3, dtype=torch.float64, device=device)
    it_val = iter(val_loader) if val_loader is not None else None
    with torch.no_grad():
        for j in range(n_val):
            if it_val is not None:
                imgs, gt, _n, boxes, _pos, _m, _ids = next(it_val)
                imgs, gt, boxes = imgs.to(device, non_blocking=True), gt.to(device, non_blocking=True), boxes.to(device, non_blocking=True)
            else:
                imgs, boxes, gt, _mask = make_batch(B, shots=3, seed=seed * 100003 + 7_000_000 + epoch * n_val + j, device=device)
            S = rng.randint(0, 3)                                                       # :338
            out = model(imgs, boxes, S)
            pred = out.reshape(len(imgs), -1).sum(1) / 60
            gtc = gt.reshape(len(imgs), -1).sum(1) / 60
            err = (pred - gtc).abs().float()
            nae = err / gtc
            nae[nae == float("inf")] = 0                                                # :348-349
            acc[0] += err.double().mean()
            acc[1] += (err ** 2).double().mean()
            acc[2] += nae.double().mean()
    if was_training:
        model.train()
    a = (acc / n_val).tolist()
    return {"MAE": a[0], "RMSE": a[1] ** 0.5, "NAE": a[2]}


if __name__ == "__main__":
    main(get_args_parser().parse_args())
