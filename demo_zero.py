#!/usr/bin/env python3
"""Zero-shot counting demo with the reference's flags (demo_zero.py:23-126), running the MI355X engine.

    python demo_zero.py --input_path <image or directory> [--output_path results] [--model_path weights/FSC147.pth]

Every image is resized to height 384 (width a multiple of 16, demo_zero.py:23-38), covered by 384-px windows with stride 128 and
counted with shot_num = 0 and an empty exemplar tensor (:41-74).  The reference runs one window per forward; here the windows of
--group_images images (default 8: eight 1920x1080 frames = 32 windows = BASELINE config 5) share ONE forward
(countr_amd.inference.count_images).  The visualisation (:77-90: image / 2 + density in the red channel / 2 + the count as text,
resized back to the input size) is written with PIL, since torchvision is not part of this build.
`--model_path ""` runs the randomly initialised model (dry runs / tests)."""
import time
from argparse import ArgumentParser
from itertools import chain
from pathlib import Path

import numpy as np
import torch
from PIL import Image, ImageDraw

import models_mae_cross
from countr_amd import inference

shot_num = 0


def load_image(img_path):
    """demo_zero.py:23-38 -> (image tensor [3, 384, new_W] in [0, 1], empty boxes, original W, H)."""
    image = Image.open(img_path).convert("RGB")
    image.load()
    W, H = image.size
    new_H = 384
    new_W = 16 * int((W / H * 384) / 16)
    image = image.resize((new_W, new_H), Image.BILINEAR)       # transforms.Resize on a PIL image
    t = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
    return t, torch.Tensor([]), W, H


def save_visualisation(sample, density_map, pred_cnt, path, old_w, old_h):
    """demo_zero.py:77-90."""
    _, h, w = sample.shape
    pred_fig = torch.stack((density_map, torch.zeros_like(density_map), torch.zeros_like(density_map)))
    count_im = Image.new(mode="RGB", size=(w, h), color=(0, 0, 0))
    ImageDraw.Draw(count_im).text((w - 70, h - 50), "%.3f" % pred_cnt, (255, 255, 255))
    count_im = torch.from_numpy(np.array(count_im).transpose((2, 0, 1)).copy()).to(sample.device)   # 0 / 255, as in the reference
    fig = torch.clamp(sample / 2 + pred_fig / 2 + count_im, 0, 1)
    arr = (fig.permute(1, 2, 0).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
    Image.fromarray(arr).resize((old_w, old_h), Image.BILINEAR).save(path)


def main():
    p = ArgumentParser()
    p.add_argument("--input_path", type=Path, required=True)
    p.add_argument("--output_path", type=Path, default="results")
    p.add_argument("--model_path", type=str, default="weights/FSC147.pth")
    p.add_argument("--group_images", type=int, default=8, help="images whose windows share one forward (8 x 1920x1080 = 32 windows)")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    p.add_argument("--no_viz", action="store_true", help="counts only, no viz_*.jpg")
    args = p.parse_args()
    args.output_path.mkdir(exist_ok=True, parents=True)
    device = torch.device("cuda")

    if not args.model_path:
        torch.manual_seed(0)          # dry runs without a checkpoint: the same random model every time
    model = models_mae_cross.__dict__["mae_vit_base_patch16"](norm_pix_loss="store_true", precision=args.precision)
    if args.model_path:
        checkpoint = torch.load(args.model_path, map_location="cpu", weights_only=False)   # raises if the file is missing, as upstream
        model.load_state_dict(checkpoint["model"], strict=False)
        print("Resume checkpoint %s" % args.model_path)
    model.to(device).eval()

    if args.input_path.is_dir():
        inputs = sorted(chain(args.input_path.glob("*.jpg"), args.input_path.glob("*.png")))
    else:
        inputs = [args.input_path]
    done = 0
    for g0 in range(0, len(inputs), max(args.group_images, 1)):
        paths = inputs[g0:g0 + max(args.group_images, 1)]
        loaded = [load_image(pth) for pth in paths]
        t0 = time.perf_counter()
        items = [(s.unsqueeze(0).to(device, non_blocking=True), b.unsqueeze(0).to(device), None) for s, b, _w, _h in loaded]
        results = inference.count_images(model, items, normalization=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(paths)
        for pth, (sample, _b, old_w, old_h), (pred_cnt, dm) in zip(paths, loaded, results):
            done += 1
            if not args.no_viz:
                save_visualisation(sample.to(device), dm.float(), pred_cnt, args.output_path / ("viz_%s.jpg" % pth.stem), old_w, old_h)
            if len(inputs) > 1:
                print("[%3d/%d] %s:\tcount = %5.2f  -  time = %5.2f" % (done, len(inputs), pth.name, pred_cnt, dt))
            else:
                print("Count:", pred_cnt, "- Time:", dt)


if __name__ == "__main__":
    main()
