"""Where does the bf16 engine's shot_num = 0 count error come from, and would fp32 activations in the LAST head stage remove it?
(round-3 verdict, item 5).  CPU study on the oracle's restatement (test infrastructure: oracle/countr_ref.py) of the b1_s0 golden
case (tests/test_model_gpu.py::case_inputs): bf16 roundings are injected at chosen places of an otherwise fp32 forward --

  enc     every nn.Linear of the encoder multiplies bf16-rounded operands (inputs and weights), attention probabilities / outputs rounded
  dec     the same for decoder_embed and the two decoder blocks
  head    the four head convolutions multiply bf16 inputs and weights and store bf16 maps (what the engine does)
  head<i> ... only head stage i (0..3)
  hc3     ONLY the output map of the last 3x3 convolution is rounded (the one rounding fp32 last-stage activations would remove)

-- and the count / relative error of the density map is reported for every set, and for "all" and "all minus hc3".
usage: python tools/study_bf16_head.py [case ...]   (cases: b1_s0 b1_zero_empty b1_s1; default b1_s0 b1_zero_empty)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import countr_ref as R
from oracle import weights as W

MODEL = "mae_vit_base_patch16"
q = lambda t: t.to(torch.bfloat16).to(torch.float32)
ACTIVE = set()
REGION = [None]          # "enc" | "dec" | "head"
HEAD_IDX = [0]

_linear, _conv2d, _softmax_attn = R.linear, F.conv2d, R.self_attention


def linear(x, w, b):
    if REGION[0] in ACTIVE:
        return _linear(q(x), q(w), b)
    return _linear(x, w, b)


def conv2d(x, w, b=None, padding=0, **kw):
    if REGION[0] != "head":
        return _conv2d(x, w, b, padding=padding, **kw)
    i = HEAD_IDX[0]
    three = w.shape[-1] == 3
    on = ("head" in ACTIVE or ("head%d" % i) in ACTIVE) and three
    y = _conv2d(q(x), q(w), b, padding=padding, **kw) if on else _conv2d(x, w, b, padding=padding, **kw)
    if three:
        if on or (i == 3 and "hc3" in ACTIVE):
            y = q(y)
        HEAD_IDX[0] += 1
    return y


def forward(sd, imgs, boxes, S):
    cfg = R.CONFIGS[MODEL]
    p = R.Params(sd, torch.float32)
    R.linear, F.conv2d = linear, conv2d
    try:
        with torch.no_grad():
            REGION[0] = "enc"
            lat = R.forward_encoder(p, torch.as_tensor(imgs), cfg)
            if "enc" in ACTIVE:
                lat = q(lat)
            REGION[0] = "dec"
            # forward_decoder runs decoder blocks then the head: switch the region when the first convolution arrives
            HEAD_IDX[0] = 0
            orig = F.conv2d

            def conv_switch(x, w, *a, **k):
                if w.shape[1] >= 256 and x.shape[-1] >= 24:     # (not the exemplar CNN's 64 .. 8-pixel maps)
                    REGION[0] = "head"
                return orig(x, w, *a, **k)
            F.conv2d = conv_switch
            out = R.forward_decoder(p, lat, torch.as_tensor(boxes), S, cfg)
    finally:
        R.linear, F.conv2d = _linear, _conv2d
    return out.numpy()


def main():
    torch.set_num_threads(min(os.cpu_count(), 32))
    cases = sys.argv[1:] or ["b1_s0", "b1_zero_empty"]
    sd = W.make_state_dict(MODEL, seed=0)
    imgs, boxes, _gt, _mask = W.make_inputs(batch=2, shots=3, seed=0)
    inputs = {"b1_s0": (imgs[:1], boxes[:1], 0), "b1_zero_empty": (imgs[1:2], np.zeros((1, 0), np.float32), 0), "b1_s1": (imgs[1:2], boxes[1:2], 1)}
    sets = [(), ("hc3",), ("head3",), ("head",), ("dec",), ("enc",), ("enc", "dec"), ("enc", "dec", "head"), ("enc", "dec", "head0", "head1", "head2")]
    for c in cases:
        im, bx, S = inputs[c]
        ref = None
        print("case %s" % c)
        for s in sets:
            ACTIVE.clear(); ACTIVE.update(s)
            out = forward(sd, im, bx, S)
            if ref is None:
                ref = out
            cnt, rc = out.sum() / 60, ref.sum() / 60
            err = np.abs(out - ref)
            print("  bf16 at %-34s count %9.3f (%+6.2f %%)   map max-rel %.2e  rms-rel %.2e" % (
                "+".join(s) if s else "(nowhere: fp32)", cnt, 100 * (cnt - rc) / rc, err.max() / np.abs(ref).max(),
                np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean())), flush=True)


if __name__ == "__main__":
    main()
