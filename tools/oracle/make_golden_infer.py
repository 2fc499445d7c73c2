#!/usr/bin/env python3
"""Golden vectors of the reference's TEST-TIME path (runs only in the build container).

The per-image code of the reference lives inline in its scripts, so it cannot be imported.  It is not copied either: this
generator READS the line ranges below from /root/reference at run time, dedents them and executes them unchanged against the
real reference model (imported through the shims of make_golden.py):
  * FSC_test_cross(few-shot).py:261-359  box count, the s_cnt rule, the 3x3 crop-and-upscale path, the sliding-window loop, the
                                         test-time normalisation by the exemplar-box count
  * demo_zero.py:42-74                    run_one_image's zero-shot sliding-window loop (boxes = torch.Tensor([]), shot_num = 0)
The only stand-ins are for torchvision, which is not installed: TF.crop on a tensor == slicing, transforms.Resize((h, w)) on a
tensor == F.interpolate(mode="bilinear", align_corners=False) (torchvision 0.14.1: no antialias for tensors).
Output: tests/golden/infer.npz (inputs are regenerated from seeds by the tests)."""
import os
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, ROOT, REF, OUT  # noqa: E402


def ref_lines(fname, lo, hi):
    with open(os.path.join(REF, fname)) as f:
        lines = f.readlines()[lo - 1:hi]
    return textwrap.dedent("".join(lines))


class _TF:
    @staticmethod
    def crop(img, top, left, height, width):
        return img[..., top:top + height, left:left + width]


class _Resize:
    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        return F.interpolate(img.unsqueeze(0), size=self.size, mode="bilinear", align_corners=False)[0]


def inputs(seed, width, shots):
    """Also used by tests/test_inference_gpu.py (kept in sync through oracle/weights.make_wide_inputs)."""
    from oracle import weights as W
    return W.make_wide_inputs(seed, width, shots)


def main():
    from oracle import weights as W
    mm, _ = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = mm.__dict__["mae_vit_base_patch16"](norm_pix_loss=False)
    sd = W.make_state_dict("mae_vit_base_patch16", seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    device = torch.device("cpu")
    few = ref_lines("FSC_test_cross(few-shot).py", 261, 359)
    zero = ref_lines("demo_zero.py", 42, 74)
    # the zero-shot loop sits inside `with measure_time() as et:` -- give it a no-op context manager
    import contextlib
    out = {}

    # ---- zero-shot, 1920x1080 -> 384 x 672 (config 5): 2 images
    for k in range(2):
        img, _bx, _pos = inputs(100 + k, 672, 0)
        ns = {"torch": torch, "nn": nn, "model": model, "device": device, "samples": torch.from_numpy(img),
              "boxes": torch.Tensor([]).unsqueeze(0),        # demo_zero.py:38,118: load_image's torch.Tensor([]) then unsqueeze(0)
              "shot_num": 0, "measure_time": contextlib.nullcontext}
        exec(zero, ns)
        dm = ns["density_map"]
        out["zero%d_count" % k] = np.float64(ns["pred_cnt"])
        out["zero%d_colsum" % k] = dm.sum(0).numpy()
        out["zero%d_rowsum" % k] = dm.sum(1).numpy()

    # ---- few-shot per-image path: (a) plain windows + normalisation, (b) tiny exemplars -> 3x3 split path
    args = types.SimpleNamespace(max_s_cnt=1, normalization=True)
    for tag, seed, width, tiny in (("plain", 200, 512, False), ("split", 201, 400, True)):
        img, bx, pos = inputs(seed, width, 3)
        if tiny:
            pos = [(20, 30, 27, 38), (100, 200, 150, 260), (300, 90, 306, 95)]     # two rectangles below 10 px -> s_cnt = 2 >= max_s_cnt
        ns = {"torch": torch, "nn": nn, "model": model, "device": device, "samples": torch.from_numpy(img), "boxes": torch.from_numpy(bx),
              "pos": pos, "args": args, "TF": _TF, "transforms": types.SimpleNamespace(Resize=_Resize)}
        exec(few, ns)
        out[tag + "_count"] = np.float64(ns["pred_cnt"])
        out[tag + "_s_cnt"] = np.int64(ns["s_cnt"])
        out[tag + "_colsum"] = ns["density_map"].sum(0).numpy()          # split path: the density of the LAST crop, as in the reference
        out[tag + "_pos"] = np.array(pos)
        if tiny:
            out[tag + "_crop_counts"] = np.array([float(d.sum() / 60) for d in ns["r_densities"]])
    np.savez_compressed(os.path.join(OUT, "infer.npz"), **out)
    for k, v in out.items():
        print(k, np.asarray(v).shape, np.asarray(v).reshape(-1)[:3])


if __name__ == "__main__":
    main()
