#!/usr/bin/env python3
"""Few-/zero-shot evaluation CLI with the reference's flags (FSC_test_cross(few-shot).py:26-78), running the MI355X engine.

`--box_bound 0` = zero-shot.  With the FSC147 files present (--data_path/--anno_file/--data_split_file/--im_dir) images are
loaded as TestData does (countr_amd/data/fsc147.py::test_item, pinned against the reference class in tests/golden/data_test.npz;
`--external`: the split's own exemplar crops, cut to --box_bound, serve every image -- :96-129).
Without a dataset (none is available offline) `--synthetic N` evaluates N synthetic wide images through the same
sliding-window / stitching / normalisation code (countr_amd/inference.py)."""
import argparse
import json
import os
import time

import numpy as np
import torch

import models_mae_cross
from countr_amd import inference
from countr_amd.util import misc


def get_args_parser():
    p = argparse.ArgumentParser("CounTR testing (MI355X engine)", add_help=True)
    p.add_argument("--model", default="mae_vit_base_patch16", type=str)
    p.add_argument("--mask_ratio", default=0.5, type=float)
    p.add_argument("--norm_pix_loss", action="store_true")
    p.add_argument("--data_path", default="./data/FSC147/", type=str)
    p.add_argument("--anno_file", default="annotation_FSC147_384.json", type=str)
    p.add_argument("--data_split_file", default="Train_Test_Val_FSC_147.json", type=str)
    p.add_argument("--im_dir", default="images_384_VarV2", type=str)
    p.add_argument("--output_dir", default="./Image")
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--resume", default="./output_fim6_dir/checkpoint-0.pth")
    p.add_argument("--external", action="store_true")
    p.add_argument("--box_bound", default=-1, type=int)
    p.add_argument("--split", default="test", type=str)
    p.add_argument("--max_s_cnt", default=1, type=int)
    p.add_argument("--num_workers", default=0, type=int)
    p.add_argument("--pin_mem", action="store_true")
    p.add_argument("--no_pin_mem", action="store_false", dest="pin_mem")
    p.set_defaults(pin_mem=True)
    p.add_argument("--normalization", default=True)
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local_rank", default=-1, type=int)
    p.add_argument("--dist_on_itp", action="store_true")
    p.add_argument("--dist_url", default="env://")
    # additions
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp16"], help="the reference tests in fp32")
    p.add_argument("--synthetic", default=0, type=int, help="evaluate N synthetic images instead of FSC147")
    p.add_argument("--group_images", default=8, type=int, help="images whose sliding windows share forward batches (up to 32 windows each)")
    return p


def main(args):
    misc.init_distributed_mode(args)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    device = torch.device(args.device)
    model = models_mae_cross.__dict__[args.model](norm_pix_loss=args.norm_pix_loss, precision=args.precision)
    misc.load_model_FSC(args, model)
    model.to(device).eval()
    items = []
    if args.synthetic:
        rs = np.random.RandomState(args.seed)
        for i in range(args.synthetic):
            w = 16 * int(rs.randint(24, 60))
            img = torch.from_numpy(rs.uniform(0, 1, size=(3, 384, w)).astype(np.float32))
            k = 3 if args.box_bound < 0 else min(args.box_bound, 3)
            boxes = torch.from_numpy(rs.uniform(0, 1, size=(k, 3, 64, 64)).astype(np.float32)) if k else torch.zeros(0)
            pos = [(10 * j, 10 * j, 10 * j + 40, 10 * j + 40) for j in range(k)]
            items.append(("synthetic_%d" % i, img, boxes, pos, int(rs.randint(5, 200))))
    else:
        from countr_amd.data import fsc147
        annotations = json.load(open(os.path.join(args.data_path, args.anno_file)))
        split = json.load(open(os.path.join(args.data_path, args.data_split_file)))[args.split]
        im_dir = os.path.join(args.data_path, args.im_dir)
        ext = None
        if args.external:     # FSC_test_cross(few-shot).py:96-129: the split's own exemplar crops, cut to --box_bound, for every image
            ext = fsc147.external_exemplars(annotations, split, im_dir, args.box_bound)
        for im_id in split:
            img, dots, boxes, pos, _gt_map = fsc147.test_item(annotations, im_dir, im_id, args.box_bound, ext)
            items.append((im_id, img, boxes, [tuple(r) for r in pos], dots.shape[0]))
    from countr_amd.parallel import shard_batch
    lo, hi = shard_batch(len(items), misc.get_rank(), misc.get_world_size())   # replicas only: images sharded, no collective
    mae = rmse = nae = 0.0
    # windows are batched ACROSS images (every 384-px window is an independent forward): groups of --group_images images go
    # through inference.count_images together
    mine = items[lo:hi]
    t0 = time.time()
    preds = []
    for g0 in range(0, len(mine), args.group_images):
        grp = mine[g0:g0 + args.group_images]
        its = [(img.unsqueeze(0).to(device), boxes.unsqueeze(0).to(device), pos) for _name, img, boxes, pos, _gt in grp]
        preds += [p for p, _dm in inference.count_images(model, its, normalization=bool(args.normalization), max_s_cnt=args.max_s_cnt)]
    torch.cuda.synchronize()
    t_inf = time.time() - t0
    for (name, _img, _boxes, _pos, gt_cnt), pred in zip(mine, preds):
        err = abs(pred - gt_cnt)
        mae += err; rmse += err ** 2; nae += err / gt_cnt if gt_cnt > 0 else 0
        print("%s: pred_cnt: %5.3f, gt_cnt: %5.3f, error: %5.3f" % (name, pred, gt_cnt, err))
    n = max(hi - lo, 1)
    print(json.dumps({"MAE": mae / n, "RMSE": (rmse / n) ** 0.5, "NAE": nae / n, "images": hi - lo, "mean_infer_time_s": t_inf / n}))


if __name__ == "__main__":
    main(get_args_parser().parse_args())
