"""Static-plan execution engine for the CounTR SupervisedMAE hot path on MI355X.

PyTorch is used only for device memory and streams.  For every (batch, shot_num, train) configuration
the engine builds ONE launch list over pre-allocated buffers (no allocation, no host sync inside), so a
whole forward / finetune step can be replayed from a hipGraph.  All math runs in libcountr_hip.so.

Mirrors (file:line of the reference):
  forward_encoder  models_mae_cross.py:136-148      forward_decoder  models_mae_cross.py:150-199
  Block            timm 0.4.9 / models_crossvit.py:69-94   CrossAttentionBlock models_crossvit.py:130-156
  loss / step      FSC_finetune_cross.py:286-316, util/misc.py:266-280 (no GradScaler: bf16 needs none)
Layouts: tokens [B*N, C]; feature maps NHWC; parameters live in one flat fp32 buffer (frozen region, then
the trainable decoder-side region ordered by backward completion so that gradient buckets are contiguous).
"""
import ctypes as C
import math
import os
import re

import torch

from . import _lib
from ._lib import F32, BF16, OP_ROW, OP_COL, OP_IM2ROW, OP_IM2COL, ACT_NONE, ACT_GELU, ACT_GELU_BWD, GemmArgs

HALF_DTYPES = (torch.bfloat16, torch.float16)      # the 16-bit storage types (dtype code BF16 on the C side: one per library build)

ALIGN = 64  # elements; every tensor starts on a 256-byte boundary of the flat buffers


def is_trainable(name):
    """Encoder runs under no_grad (models_mae_cross.py:204-205); pos-embeds are frozen (:30,42)."""
    if name in ("pos_embed", "decoder_pos_embed"):
        return False
    return name.startswith(("decoder_", "decode_head", "shot_token"))


def no_weight_decay(name, shape):
    """timm.optim.optim_factory.add_weight_decay (call site FSC_finetune_cross.py:234)."""
    return len(shape) == 1 or name.endswith(".bias")


def _bucket(name):
    if name.startswith(("decode_head", "decoder_norm")):
        return 0
    if name.startswith(("decoder_blocks", "decoder_embed")):
        return 1
    if name.startswith("decoder_proj"):
        return 2
    return 3  # shot_token


class ParamLayout:
    """Offsets of every state-dict tensor inside the flat buffers."""

    def __init__(self, named_shapes, trainable=None, bucket=None):
        trainable = is_trainable if trainable is None else trainable
        _bucket = globals()["_bucket"] if bucket is None else bucket
        self.shapes = dict(named_shapes)
        frozen = [n for n, _ in named_shapes if not trainable(n)]
        train = [n for n, _ in named_shapes if trainable(n)]
        train.sort(key=lambda n: (_bucket(n), 0 if no_weight_decay(n, self.shapes[n]) else 1))
        self.order = frozen + train
        self.off = {}
        o = 0
        for n in frozen:
            self.off[n] = o
            o += -(-math.prod(self.shapes[n]) // ALIGN) * ALIGN
        self.train_start = o
        self.segments = []  # (bucket, nodecay?, start, end) relative to train_start, contiguous
        for n in train:
            self.off[n] = o
            sz = -(-math.prod(self.shapes[n]) // ALIGN) * ALIGN
            key = (_bucket(n), no_weight_decay(n, self.shapes[n]))
            if self.segments and self.segments[-1][0] == key:
                self.segments[-1][2] = o + sz - self.train_start
            else:
                self.segments.append([key, o - self.train_start, o + sz - self.train_start])
            o += sz
        self.total = o
        self.n_train = o - self.train_start
        self.train_names = train
        self.frozen_names = frozen

    def bucket_range(self, bucket):
        segs = [s for s in self.segments if s[0][0] == bucket]
        return (segs[0][1], segs[-1][2]) if segs else (0, 0)

    def adam_ranges(self, S, weight_decay, skip=None):
        """(start, end, wd) ranges of the trainable region touched when shot_num == S: parameters whose gradient is
        None in the reference are skipped by torch AdamW (exemplar CNN for S == 0, shot_token otherwise).  `skip` (a set of
        buckets) overrides the shot_num rule: an accumulation window may have touched both."""
        if skip is None:
            skip = (2,) if S == 0 else (3,)
        out = []
        for (bucket, nodecay), s, e in self.segments:
            if bucket in skip:
                continue
            out.append((s, e, 0.0 if nodecay else weight_decay))
        return out

    def adam_plan(self, weight_decay, skip=(), zero=(), group_of_bucket=None):
        """(start, end, wd, counter group, zero-gradient flag) of every range the optimizer steps: buckets in `skip` never had a
        gradient (torch: grad is None -> skipped), buckets in `zero` had one earlier but none in this window (torch 1.13
        zero_grad() leaves a zero tensor -> stepped with g = 0)."""
        gob = ADAM_GROUP_OF_BUCKET if group_of_bucket is None else group_of_bucket
        return [(s, e, 0.0 if nodecay else weight_decay, gob.get(bucket, 0), int(bucket in zero))
                for (bucket, nodecay), s, e in self.segments if bucket not in skip]


def check_supported(cfg, img_size, precision="bf16"):
    """What the gfx950 kernels cover.  The CounTR shapes proper -- a patch grid that tiles the image, head_dim 32 / 64 (the fused
    attention kernels), a token count that is a multiple of 8 -- run in every precision, forward and backward.  mae_vit_huge_patch14
    (models_mae_cross.py:235-239: patch 14 does not divide 384 -> timm's PatchEmbed conv drops the last 6 pixels and leaves 27 x 27 = 729
    tokens; head_dim 1280 / 16 = 80; the density head turns 27 into a 432 x 432 map) runs FORWARD-ONLY, as the reference's own forward
    does (its training loss compares the map with a 384 x 384 ground truth and cannot run: FSC_finetune_cross.py:294), in every
    precision: head_dim 80 goes through the batched-GEMM attention (score / probability matrices padded to 736 columns,
    countr_softmax_fwd_ld), the patch matrix with K = 588 (rows of 1176 bytes in 16-bit storage: no 16-byte chunks) stays an fp32
    product in the 16-bit modes too (0.6 % of the forward), the 729-token decoder runs the fused dh = 32 kernel's ragged form; no
    LayerNorm folding / pre-scaled q (both are packed for the head_dim-64 encoder).  Returns True for such a forward-only configuration."""
    patch, D, _depth, H, Dd, _dd, Hd = cfg
    bad, special = [], False
    if img_size < patch:
        bad.append("patch size %d exceeds the %d-pixel input" % (patch, img_size))
    if img_size % patch:
        special = True
    if ((img_size // max(patch, 1)) ** 2) % 8:
        special = True
    for what, dim, heads in (("encoder", D, H), ("decoder", Dd, Hd)):
        if dim % heads or (dim // heads) % 4:
            bad.append("%s head_dim %s (embed_dim %d / %d heads) is not a multiple of 4" % (what, dim / heads, dim, heads))
        elif dim // heads not in (32, 64):
            special = True
    if bad:
        raise _lib.CountrError("this model configuration is not supported by the gfx950 kernels: " + "; ".join(bad))
    return special


# bias-correction counter group of an AdamW range by gradient bucket: the conditional parameter sets start stepping later
ADAM_GROUP_OF_BUCKET = {0: 0, 1: 0, 2: 1, 3: 2}


class _Fake:
    """Stand-in tensor used by the sizing pass of a plan build (no memory is touched)."""

    def __init__(self, dtype):
        self.dtype = dtype

    def data_ptr(self):
        return 0

    def element_size(self):
        return torch.empty((), dtype=self.dtype).element_size()


class _Cols:
    """Columns [off, ...) of a row-major buffer as a launch operand (data_ptr only; works on the sizing pass's stand-ins too)."""

    def __init__(self, t, off):
        self.t, self.off = t, off

    def data_ptr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()


class Plan:
    """Launch lists + buffers for one (B, S, train) configuration."""

    def __init__(self):
        self.fwd = []
        self.fwd_par = None  # the same launches with the exemplar CNN on a side lane beside the encoder (full forward only)
        self.bwd_head = []   # backward until bucket 0 (head + decoder_norm) gradients are final
        self.bwd_rest = []   # ... until bucket 1 (decoder blocks + decoder_embed) is final
        self.bwd_tok = []    # exemplar tokens: exemplar CNN (bucket 2) or shot_token (bucket 3)
        self.acc = None      # the same three lists with parameter gradients ACCUMULATED (micro-steps 2.. of gradient accumulation)
        self.buf = {}


class Engine:
    FROZEN_ENCODER = True    # SupervisedMAE: the ViT encoder runs under no_grad (models_mae_cross.py:204-205) and is never updated

    def __init__(self, cfg, named_shapes, device, precision="bf16", img_size=384, attention="auto", ln_eps=1e-6):
        """cfg = (patch, embed_dim, depth, heads, dec_dim, dec_depth, dec_heads); ln_eps: eps of the model's norm_layer (the
        reference factories pass partial(nn.LayerNorm, eps=1e-6), models_mae_cross.py:210-239)."""
        self.ln_eps = float(ln_eps)
        self.L = _lib.lib(_lib.variant_of(precision))
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.CountrError("the CounTR HIP engine needs a GPU device (no CPU fallback)")
        if self.device.index is None:      # plain "cuda": the calling thread's current device (rank k of a multi-GPU job), not 0
            self.device = torch.device("cuda", torch.cuda.current_device())
        _lib.check(self.L.countr_init(self.device.index), "countr_init")
        self.cfg = cfg
        self.patch, self.D, self.depth, self.H, self.Dd, self.ddepth, self.Hd = cfg
        self.forward_only = check_supported(cfg, img_size, precision)      # (mae_vit_huge_patch14: forward only, like the reference's)
        self.img = img_size
        self.grid = img_size // self.patch
        self.N = self.grid * self.grid
        self.Np = -(-self.N // 8) * 8        # row pitch of the unfused attention's score / probability matrices
        if precision not in ("bf16", "fp16", "fp32"):
            raise ValueError("precision must be 'bf16', 'fp16' or 'fp32'")
        self.precision = precision
        # 'bf16' (throughput mode) and 'fp16' (the reference's autocast dtype, FSC_finetune_cross.py:273-275,286: three more mantissa
        # bits, same MFMA rate) run the SAME 16-bit code paths -- the dtype code BF16 means "16-bit storage" -- from two builds of the
        # library that differ in the conversions and the matrix instruction's operand type (_lib.variant_of, csrc/common.hpp)
        self.half = precision in ("bf16", "fp16")
        self.code = BF16 if self.half else F32
        self.tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[precision]
        self.attention = attention
        self.layout = self._make_layout(named_shapes)
        lay = self.layout
        self.P = torch.zeros(lay.total, device=self.device, dtype=torch.float32)
        self.G = torch.zeros(lay.n_train, device=self.device, dtype=torch.float32)
        self.M = None  # AdamW state, allocated on first optimizer use
        self.V = None
        self.Wt = self.P if precision == "fp32" else torch.zeros(lay.total, device=self.device, dtype=self.tdt)
        # permuted conv-weight shadows: OHWI for the forward/wgrad implicit GEMM, "dgrad form" for dgrad
        self.conv_names = self._conv_names()
        self.Wf, self.Wd = {}, {}
        for n in self.conv_names:
            numel = math.prod(lay.shapes[n])
            self.Wf[n] = torch.zeros(numel, device=self.device, dtype=self.tdt)
            self.Wd[n] = torch.zeros(numel, device=self.device, dtype=self.tdt)
        # transposed bf16 shadows of the trainable decoder Linear weights: dx = dy W becomes a (ROW, ROW) GEMM on W^T and runs the lean
        # kernel (linear.hip) like the forward -- ~4 us per launch against the (ROW, COL) form on gemm_kernel; written by the shadow
        # launch that follows AdamW
        self.WtT = {}
        if self.half and self.FROZEN_ENCODER and os.environ.get("COUNTR_LEAN", "1") != "0":
            for n in lay.train_names:
                shp = lay.shapes[n]
                if n.startswith("decoder_blocks.") and n.endswith(".weight") and len(shp) == 2 and shp[0] % 128 == 0 and shp[1] % 128 == 0:
                    self.WtT[n] = torch.zeros(math.prod(shp), device=self.device, dtype=self.tdt)
        # MAE pretraining (every Linear trains and has an input gradient): all 82 transposes = 222 MB, refreshed in three shadow launches
        # behind AdamW.  Round 3 measured this as a loss at 8 images (8.40 against 8.12 ms: the transposes cost 376 us, more than the
        # (ROW, ROW) launches saved at M = 2304); with round 4's 64x64-tile transposes and the warm-up hints -- which only the (ROW, ROW)
        # kernels honour -- it is neutral at 8 images (7.82 vs 7.80 ms) and a gain at config 4's 16: 10.79 -> 10.55 ms (two A/B pairs)
        elif self.half and not self.FROZEN_ENCODER and os.environ.get("COUNTR_LEAN", "1") != "0":
            for n in lay.train_names:
                shp = lay.shapes[n]
                if n.endswith(".weight") and len(shp) == 2 and shp[0] % 128 == 0 and shp[1] % 128 == 0:
                    self.WtT[n] = torch.zeros(math.prod(shp), device=self.device, dtype=self.tdt)
        self.plans = {}
        self.hyper = torch.zeros(8, device=self.device, dtype=torch.float32)   # {lr, bc1[0], bc2[0], grad_scale, bc1[1], bc2[1], bc1[2], bc2[2]}
        self.step_count = 0
        self.group_steps = [0, 0, 0]    # optimizer steps taken by counter group 0 (always), 1 (exemplar CNN), 2 (shot_token)
        self.opt_seen = set()           # conditional gradient buckets (2, 3) that have had a gradient at least once
        self.gnorm = None               # device fp32 [countr_adamw_gnorm_floats()]: [0] = gradient L2 norm of the last step
        # frozen-encoder q projection packed pre-scaled for the attention kernel (see _pack_prescaled_q)
        self.prescale_q = (self.FROZEN_ENCODER and self.half and attention != "unfused" and self.D // self.H == 64
                           and self.N % 64 == 0)
        self.qkv_bias_pre = None
        # frozen encoder, bf16: norm1 / norm2 are folded into the qkv / fc1 layers (gamma into the packed weights, beta into the bias;
        # the proj / fc2 / patch-embed epilogues emit the bf16 operand + 64-column row partials, the qkv / fc1 epilogues apply mean and
        # rstd -- _pack_prescaled_q, _build): 24 LayerNorm launches and their 0.5 GB of traffic per step gone.  COUNTR_LN_FOLD=0 disables.
        self.ln_fold = (self.FROZEN_ENCODER and self.half and self.D % 128 == 0 and os.environ.get("COUNTR_LN_FOLD", "1") != "0"
                        and os.environ.get("COUNTR_LEAN", "1") != "0" and not self.forward_only)
        self.fc1_bias_pre = self.ln_c_qkv = self.ln_c_fc1 = None
        self._ln_checked = False     # check_ln_fold() has looked at the activations this weight set produces
        self.ln_fold_ratio = None    # ... and this is the largest |mean| / sigma it saw at a folded LayerNorm's input
        self._ws = {}
        self._need = {}
        self._sizing = False
        self.reduce_vec4 = True      # 16-byte loads in the deferred-sum kernel
        self._defer = {}             # id(ops) -> (ops, [pending reduction entries]): flushed into countr_reduce_table launches
        self._tables = []
        self.warm_weights = os.environ.get("COUNTR_WARM", "1") != "0"     # spare workgroups of a GEMM read the next GEMM's (cold) weight panel
        self.defer_reduce = True     # split-K slabs / bias row sums / LayerNorm block partials are summed by table-driven launches
        self._acc = 0                # accumulate flag baked into the parameter-gradient launches being built (gradient accumulation)
        self.generation = 0
        self._sides = None
        # (fork / join of the independent backward branches -- wgrad | dgrad | bias grad -- measured a wash, 9.62 vs 9.56 ms, and removed:
        # the fork / lane / join markers inside the launch lists are ignored by run(); only the exemplar branch's "x" markers fork)
        self.parallel_lanes = False
        # forward: the exemplar CNN (~20 launches of fewer than 200 workgroups, ~0.25 ms serial) runs on a side lane beside the encoder.
        # Round 1 measured this as a loss (6.45 vs 6.20 ms, four graph launches per step); with the whole step in ONE graph it
        # gains 50-70 us per step at B = 8 (5.63 -> 5.57 ms, three A/B pairs on one box).
        self.overlap_exemplar = True
        self.group_wgrads = os.environ.get("COUNTR_GROUP_WGRADS", "1") != "0"   # a block's Linear weight gradients in one launch (A/B switch)
        self.act_splitk = True       # split-K + finisher for few-tile, long-K forward GEMMs

    def _make_layout(self, named_shapes):
        return ParamLayout(named_shapes)

    def _conv_names(self):
        return ["decoder_proj%d.0.weight" % i for i in (2, 3, 4)] + ["decode_head%d.0.weight" % i for i in range(4)]

    # ------------------------------------------------------------------ parameter views
    def pview(self, name):
        o = self.layout.off[name]
        shp = self.layout.shapes[name]
        return self.P[o:o + math.prod(shp)].view(shp)

    def gview(self, name):
        o = self.layout.off[name] - self.layout.train_start
        shp = self.layout.shapes[name]
        return self.G[o:o + math.prod(shp)].view(shp)

    def _pp(self, name):  # fp32 master pointer
        return self.P.data_ptr() + 4 * self.layout.off[name]

    def _wp(self, name):  # GEMM-operand (T) pointer of a linear weight
        return self.Wt.data_ptr() + self.Wt.element_size() * self.layout.off[name]

    def _gp(self, name):
        return self.G.data_ptr() + 4 * (self.layout.off[name] - self.layout.train_start)

    def sync_weights(self, trainable_only=False, stream=None):
        """Refresh the low-precision / permuted shadows from the fp32 master buffer."""
        st = self._stream() if stream is None else stream
        L, lay = self.L, self.layout
        if self.half:
            lo = lay.train_start if trainable_only else 0
            _lib.check(L.countr_cast_permute(self.P.data_ptr() + 4 * lo, self.Wt.data_ptr() + 2 * lo, lay.total - lo, 0, 0, 0, 0,
                                             BF16, st), "cast")
            if (self.prescale_q or self.ln_fold) and not trainable_only and stream is None:
                self._pack_prescaled_q()
        self._refresh_conv_shadows(st)      # (the W^T shadows are taken from the 16-bit shadow the cast above wrote on this stream)

    def _pack_prescaled_q(self):
        """Frozen encoder, bf16 mode, weight-packing time only (load_state_dict / .to()): a few torch ops on the engine's stream.
        (1) pre-scaled q: the q rows of every blocks.i.attn.qkv shadow (and a copy of its bias) carry the factor dh^-0.5 * log2(e),
        rounded ONCE from the fp32 master -- q = x W_q^T + b_q is linear, so the attention kernel receives scores in the exp2 domain
        (countr_attn_fwd with scale <= 0) and does no per-score scale / subtract.
        (2) LayerNorm folding (ln_fold): LN(x) W^T + b = rstd (x (gamma o W)^T - mean c) + (b + W beta), c_n = sum_k (gamma o W)_nk: the
        qkv / fc1 shadows carry gamma of norm1 / norm2, the bias copies carry W beta, and c is summed from the ROUNDED bf16 shadow (the
        values the GEMM multiplies), so that a constant row still maps to exactly the bias."""
        D, lay, P = self.D, self.layout, self.P
        self._ln_checked = False            # new frozen weights: the next forward looks at the activations again (check_ln_fold)
        c = (D // self.H) ** -0.5 * 1.4426950408889634 if self.prescale_q else 1.0
        if self.qkv_bias_pre is None:
            self.qkv_bias_pre = torch.zeros((self.depth, 3 * D), device=self.device, dtype=torch.float32)
            if self.ln_fold:
                self.fc1_bias_pre = torch.zeros((self.depth, 4 * D), device=self.device, dtype=torch.float32)
                self.ln_c_qkv = torch.zeros((self.depth, 3 * D), device=self.device, dtype=torch.float32)
                self.ln_c_fc1 = torch.zeros((self.depth, 4 * D), device=self.device, dtype=torch.float32)
        view = lambda name, *shape: P[lay.off[name]:lay.off[name] + math.prod(shape)].view(*shape)
        for i in range(self.depth):
            b = "blocks.%d." % i
            W, bias = view(b + "attn.qkv.weight", 3 * D, D), view(b + "attn.qkv.bias", 3 * D)
            ow = lay.off[b + "attn.qkv.weight"]
            if self.ln_fold:
                g1, be1 = view(b + "norm1.weight", D), view(b + "norm1.bias", D)
                Wf, bf = W * g1[None, :], bias + W @ be1
            else:
                Wf, bf = W.clone(), bias.clone()
            Wf[:D] *= c
            bf[:D] *= c
            self.Wt[ow:ow + 3 * D * D].copy_(Wf.reshape(-1).to(self.tdt))
            self.qkv_bias_pre[i].copy_(bf)
            if self.ln_fold:
                self.ln_c_qkv[i].copy_(self.Wt[ow:ow + 3 * D * D].view(3 * D, D).float().sum(1))
                W1, b1 = view(b + "mlp.fc1.weight", 4 * D, D), view(b + "mlp.fc1.bias", 4 * D)
                g2, be2 = view(b + "norm2.weight", D), view(b + "norm2.bias", D)
                o1 = lay.off[b + "mlp.fc1.weight"]
                self.Wt[o1:o1 + 4 * D * D].copy_((W1 * g2[None, :]).reshape(-1).to(self.tdt))
                self.fc1_bias_pre[i].copy_(b1 + W1 @ be2)
                self.ln_c_fc1[i].copy_(self.Wt[o1:o1 + 4 * D * D].view(4 * D, D).float().sum(1))

    # LayerNorm folding rounds the RAW residual row to the 16-bit operand type (the folded GEMM then subtracts mean * colsum): relative to
    # the row's sigma the operand's rounding error is 2^-9 (bf16) / 2^-12 (fp16) times sqrt(1 + (mean / sigma)^2).  The deterministic test
    # weights stay below 0.8 (profiles/r3_ln_fold_mean_over_sigma.txt); a real checkpoint is not under our control, so the FIRST forward
    # behind every (re)load of the frozen weights measures the ratio on its own inputs and falls back to the LayerNorm kernels beyond
    # these limits (bf16: the operand error would exceed 4x the unfolded one).
    LN_FOLD_LIMIT = {"bf16": 4.0, "fp16": 32.0}

    def check_ln_fold(self, imgs):
        """Guard of the LayerNorm fold (once per weight set; a few ms, one host sync): runs the encoder of up to two of `imgs` launch by
        launch and reads, behind every producer of the residual stream, the row partials {sum, sum of squares} it leaves for the folded
        consumer -> max over rows and layers of |mean| / sigma.  Above LN_FOLD_LIMIT the fold is switched off (plans rebuilt with the
        LayerNorm launches, shadows re-packed without gamma, captured graphs dropped through `generation`) with a warning.  Returns the
        ratio, or None when nothing was checked."""
        if not self.ln_fold or self._ln_checked or torch.cuda.is_current_stream_capturing():
            return None
        self._ln_checked = True
        Bc = min(int(imgs.shape[0]), 2)
        p = self.plan(Bc, 0, False)
        p.buf["img"].copy_(imgs[:Bc].to(torch.float32), non_blocking=True)
        st = p.buf["lnstats"]
        worst = torch.zeros((), device=self.device, dtype=torch.float32)
        D = float(self.D)
        for op in p.fwd[:p.enc_ops]:
            self.run([op])
            a = op[2]
            if op[0] is self.L.countr_gemm and a is not None and a.ln_stats_out:
                mean = st[..., 0].sum(1) / D
                var = (st[..., 1].sum(1) / D - mean * mean).clamp_min(1e-30)
                worst = torch.maximum(worst, (mean.abs() / var.sqrt()).max())
        self.ln_fold_ratio = r = float(worst.item())
        if not (r <= self.LN_FOLD_LIMIT[self.precision]):       # (also trips on NaN)
            import warnings
            warnings.warn("countr_amd: |mean| / sigma = %.1f at the input of a folded LayerNorm (limit %.0f in %s mode): these weights put "
                          "large common offsets on the residual stream; LayerNorm folding is switched off for this model (separate "
                          "LayerNorm launches, ~2 %% slower)" % (r, self.LN_FOLD_LIMIT[self.precision], self.precision))
            self.ln_fold = False
            self.plans.clear()
            self.generation += 1
            self.sync_weights()
            self._ln_checked = True
        return r

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ plan helpers
    def _alloc(self, plan, key, shape, dtype):
        if self._sizing:
            return _Fake(dtype)
        if key in plan.buf:          # the backward lists are built twice (overwrite / accumulate) over the same buffers
            return plan.buf[key]
        t = torch.empty(shape, device=self.device, dtype=dtype)
        plan.buf[key] = t
        return t

    def _shared(self, key, numel, dtype=torch.float32):
        """Scratch shared by all plans.  A sizing pass records the maximum request; the real pass then finds
        the buffer already large enough, so pointers baked into launch lists stay valid."""
        if self._sizing:
            old = self._need.get(key, (0, dtype))
            self._need[key] = (max(old[0], int(numel)), dtype)
            return _Fake(dtype)
        cur = self._ws[key]
        assert cur.numel() >= numel and cur.dtype == dtype, key
        return cur

    def _reserve(self):
        grew = False
        for key, (numel, dtype) in self._need.items():
            cur = self._ws.get(key)
            if cur is None or cur.numel() < numel or cur.dtype != dtype:
                self._ws[key] = torch.empty(max(numel, 1), device=self.device, dtype=dtype)
                grew = True
        if grew and self.plans:
            self.plans.clear()  # launch lists of older plans hold pointers into the replaced scratch
            self.generation += 1  # captured graphs built on those plans must be dropped too (see FinetuneStep)

    def _gemm(self, ops, dtype_code, ma, mb, **kw):
        a = GemmArgs()
        a.alpha = 1.0
        a.nbatch = 1
        a.nb1 = 1
        a.splitk = 1
        for k, v in kw.items():
            setattr(a, k, v)
        ops.append((self.L.countr_gemm, (C.byref(a), dtype_code, ma, mb), a))

    def _op(self, ops, fn, *args):
        ops.append((fn, args, None))

    # ---- parallel branches: ops between FORK and JOIN are distributed over up to 3 stream "lanes" (lane 0 = the caller's
    # stream).  Under graph capture the event fork/join becomes parallel branches of the hipGraph, so independent small
    # kernels (bias-grad, wgrad, dgrad of one layer) share the GPU instead of running back to back.
    FORK, JOIN = ("fork",), ("join",)

    def _fork(self, ops):
        ops.append((None, self.FORK, None))

    def _lane(self, ops, k):
        ops.append((None, ("lane", k), None))

    def _join(self, ops):
        ops.append((None, self.JOIN, None))

    def _side_streams(self):
        if self._sides is None:
            self._sides = [torch.cuda.Stream(device=self.device) for _ in range(2)]
            self._fork_ev = torch.cuda.Event()
            self._join_ev = [torch.cuda.Event() for _ in range(2)]
        return self._sides

    def _pipe_stream(self):
        if getattr(self, "_pipe", None) is None:
            self._pipe = torch.cuda.Stream(device=self.device)
            self._pipe_ev = [torch.cuda.Event(), torch.cuda.Event()]
        return self._pipe

    def pipe_claim(self):
        """Whoever launches the encoder lane owns what it leaves in the shared pipe_latent buffer until the next launch of the lane (by a
        training step or an inference stream on this engine): the returned token is compared by identity before the latent is used."""
        self._pipe_token = object()
        return self._pipe_token

    def pipe_owner(self, token):
        return token is not None and getattr(self, "_pipe_token", None) is token

    def pipe_join(self):
        """The caller's stream waits for the pipelined-encoder lane (the closing edge of run()'s "pfork")."""
        ps = self._pipe_stream()
        self._pipe_ev[1].record(ps)
        torch.cuda.current_stream(self.device).wait_event(self._pipe_ev[1])

    def run(self, ops, stream=None):
        main = torch.cuda.current_stream(self.device)
        st_main = C.c_void_p(main.cuda_stream) if stream is None else stream
        st = st_main
        used = set()
        for fn, args, _keep in ops:
            if fn is None:
                kind = args[0]
                if kind == "tokready":
                    continue
                if kind == "pfork":          # pipelined encoder (trainer.FinetuneStep(pipeline_encoder=True)): the next batch's frozen-encoder
                    ps = self._pipe_stream()         # forward on its own lane beside this batch's decoder side; joined by pipe_join()
                    self._pipe_ev[0].record(main)
                    ps.wait_event(self._pipe_ev[0])
                    st = C.c_void_p(ps.cuda_stream)
                    continue
                if kind == "pmain":
                    st = st_main
                    continue
                if kind[0] == "x":           # forward overlap of the exemplar CNN with the encoder (overlap_exemplar)
                    if not self.overlap_exemplar:
                        continue
                    kind = kind[1:]
                elif not self.parallel_lanes:
                    continue
                sides = self._side_streams()
                if kind == "fork":
                    self._fork_ev.record(main)
                    for sd in sides:
                        sd.wait_event(self._fork_ev)
                    used = set()
                elif kind == "join":
                    for k in sorted(used):
                        self._join_ev[k].record(sides[k])
                        main.wait_event(self._join_ev[k])
                    st = st_main
                else:
                    k = args[1]
                    if k == 0:
                        st = st_main
                    else:
                        st = C.c_void_p(sides[k - 1].cuda_stream)
                        used.add(k - 1)
                continue
            rc = fn(*args, st)
            if rc != 0:
                _lib.check(rc, getattr(fn, "__name__", "countr op"))

    def run_backward_rest_and_tok(self, lists):
        """bwd_rest followed by bwd_tok -- or, in bf16 mode with lanes enabled, the exemplar-token backward (~20 launches of fewer than
        200 workgroups, serial in their own dependency chain, headed by the K / V input-gradient GEMMs that build dy_tok) on a side lane
        beside what is left of the decoder-block backward once block 0's cross-attention backward has produced the last dK / dV.  The two branches share no scratch buffer in bf16 mode
        (fp32 mode's unfused bias gradients use one column-sum workspace: it stays serial).  Only for a step without a collective
        between the two lists (one rank)."""
        m = next((k for k, op in enumerate(lists.bwd_rest) if op[0] is None and op[1][0] == "tokready"), None)
        # (serial as well when the deferred reductions are off -- _conv_wgrad in bwd_tok and _linear_wgrad behind the marker would then
        # share the 'splitk' / 'rowsum' scratch -- and with COUNTR_PARALLEL_LANES=1, whose nested fork / join inside bwd_tok would take
        # the rest of that list off the side lane)
        if m is None or not self.overlap_exemplar or self.code != BF16 or not self.defer_reduce or self.parallel_lanes:
            self.run(lists.bwd_rest)
            self.run(lists.bwd_tok)
            return
        comb = getattr(lists, "_bwd_comb", None)
        if comb is None:
            mark = lambda *a: (None, a, None)
            comb = lists._bwd_comb = (lists.bwd_rest[:m] + [mark("xfork"), mark("xlane", 1)] + lists.bwd_tok + [mark("xlane", 0)]
                                      + lists.bwd_rest[m + 1:] + [mark("xjoin")])
        self.run(comb)

    # ---- deferred reductions: the split-K slabs of a wgrad, the fused bias-gradient row sums and the LayerNorm dgamma / dbeta block
    # partials are not finished by one ~5-9 us launch each but collected per launch list and summed by ONE table-driven launch
    # (countr_reduce_table).  Partial buffers are shared by name with the layer index stripped, so the table is flushed when a
    # buffer is about to be reused -- once per transformer block: the partials are still cache-resident (deferring a whole
    # backward pass measured slower for the MAE step: 1.1 GB of partials went out to HBM and back) -- and at the end of a list.
    @staticmethod
    def _role(name):
        return re.sub(r"\d+", "#", name)

    def _reduce_later(self, ops, owner_ptr, partial_ptr, out_ptr, nslabs, stride, count, N=0, taps=0):
        self._defer.setdefault(id(ops), (ops, []))[1].append((partial_ptr, out_ptr, int(nslabs), int(stride), int(count), int(N),
                                                               int(taps), int(self._acc), owner_ptr))

    def _claim(self, owner_ptr):
        """A producer is about to overwrite the partial buffer at owner_ptr: finish the pending sums that still read it."""
        if self._sizing:
            return
        for key, (_ops, entries) in list(self._defer.items()):
            if any(e[8] == owner_ptr for e in entries):
                self._flush_list(key)

    def _flush_list(self, key):
        ops, entries = self._defer.pop(key)
        if self._sizing or not entries:
            return
        rows, blk = [], 0
        for (pp, op, nslabs, stride, count, N, taps, acc, _owner) in entries:
            wide = int(nslabs > 16 and taps == 0)          # LayerNorm block partials: 16 columns x 16 slab groups per block
            cin = N // taps if taps else 4
            vec = int(not wide and self.reduce_vec4 and count % 4 == 0 and stride % 4 == 0 and pp % 16 == 0 and op % 16 == 0 and cin % 4 == 0)
            rows.append([pp, op, nslabs | (acc << 32) | (wide << 33) | (vec << 34), stride, count, N, taps, blk])
            blk += -(-count // (16 if wide else (1024 if vec else 256)))
        tab = torch.tensor(rows, dtype=torch.int64).reshape(-1)
        owner = torch.repeat_interleave(torch.arange(len(rows), dtype=torch.int32),
                                        torch.tensor([(rows[i + 1][7] if i + 1 < len(rows) else blk) - rows[i][7] for i in range(len(rows))]))
        if owner.numel() & 1:
            owner = torch.cat([owner, owner.new_zeros(1)])
        tab = torch.cat([tab, owner.view(torch.int64)]).to(self.device)   # rows, then the int32 block -> entry map
        self._tables.append(tab)   # read by the launch at every replay
        self._op(ops, self.L.countr_reduce_table, tab.data_ptr(), len(rows), blk)

    def _flush_reductions(self, plan=None):
        for key in list(self._defer):
            self._flush_list(key)

    # linear forward: out = act(x W^T + b) (+ resid)
    def _linear(self, ops, x, wname, out, M, N, K, act=ACT_NONE, resid=None, res_mod=0, out_bf16=None, pre=None, bias=True, bias_ptr=None, **ln):
        """**ln: the LayerNorm-folding fields of countr_gemm_args (ln_xcopy / ln_stats_out for a producer, ln_stats / ln_colsum /
        ln_nblk / ln_eps for a consumer)."""
        out_bf16 = (out.dtype in HALF_DTYPES) if out_bf16 is None else out_bf16
        self._gemm(ops, self.code, OP_ROW, OP_ROW, A=x.data_ptr(), B=self._wp(wname), C=out.data_ptr(), **ln,
                   C2=(pre.data_ptr() if pre is not None else None),
                   bias=(bias_ptr if bias_ptr is not None else (self._pp(wname[:-6] + "bias") if bias else None)),
                   resid=(resid if isinstance(resid, int) else (resid.data_ptr() if resid is not None else None)),
                   lda=K, ldb=K, ldc=N, ldres=N, M=M, N=N, K=K, res_mod=res_mod, act=act, out_bf16=int(out_bf16))

    def _weight_buffers(self):
        """[lo, hi) address ranges of the GEMM-operand copies of the parameters (the buffers whose contents are COLD at every use: a
        step streams ~5 GB through the 256-MB memory-side cache between two uses of a layer's weights)."""
        ts = [self.Wt] + list(self.Wf.values()) + list(self.Wd.values()) + list(self.WtT.values())
        return [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in ts]

    def _auto_warm(self, ops):
        """Cache warm-up hints (countr_gemm_args.prefetch) for a finished launch list: every GEMM whose B operand is a weight panel gets
        that panel read -- once, at full width, by a few extra workgroups -- by the nearest launch IN FRONT of it that honours the hint
        (the bf16 (ROW, ROW) kernels of linear.hip / gemm256.hip).  Why: a single-round GEMM that walks a cold [N][K] panel k-slab by
        k-slab, 128-byte pieces two k-tiles ahead, pays the HBM latency on every k-tile -- the encoder's fc2 took 38.7 us in the step
        against 28.7 us with the panel resident (tools/bench_chain.py); with the hints fc2 runs at 29.4, proj 16.5 -> 15, qkv 23 ->
        20.7 us, step 4.83 -> 4.70 ms (A/B inside one gpurun call).  A side-lane warm-up kernel gave the same kernel gains and lost them
        again to the fork / join gaps and to the GEMM it ran beside (4.80 vs 4.68 ms).  Results never change.
        `ops` must be in EXECUTION order of one lane; a hint never crosses a fork / lane / join marker (the launch in front of one may
        run beside or after the consumer: wasted bandwidth)."""
        if not self.warm_weights or self.code != BF16 or self._sizing:
            return
        bufs = self._weight_buffers()
        nxt = None
        for fn, args, a in reversed(ops):
            if fn is None and args and (args[0] in ("xfork", "xlane", "xjoin") or (self.parallel_lanes and args[0] in ("fork", "lane", "join"))):
                nxt = None       # (the plain fork / lane / join markers are inert unless parallel_lanes: run() executes them in list order)
                continue
            if fn is not self.L.countr_gemm or a is None:
                continue
            dtype, ma, mb = args[1], args[2], args[3]
            if dtype == BF16 and ma == OP_ROW and mb == OP_ROW and not a.partial and a.nbatch <= 1 and nxt is not None and not a.prefetch:
                a.prefetch, a.prefetch_bytes = nxt
                nxt = None
            b = a.B
            if b and a.nbatch <= 1 and any(lo <= b < hi for lo, hi in bufs):
                n_el = ((a.N - 1) * a.ldb + a.K) if mb == OP_ROW else (((a.K - 1) * a.ldb + a.N) if mb == OP_COL else 0)
                hi = next(h for lo, h in bufs if lo <= b < h)
                lo16, end = b - b % 16, min(b + 2 * n_el, hi)
                if end - lo16 >= 4096:
                    nxt = (lo16, (end - lo16) // 16 * 16)

    def _splitk(self, tiles, ktiles):
        """Split-K factor of a wgrad GEMM: as many slabs as keep the launch within 256 workgroups, i.e. one wave-specialised
        workgroup per CU (measured against the former 512-workgroup target: finetune step -2 %, pretrain step -6 %, and fewer
        fp32 slabs to reduce)."""
        # (round 1 gave 129..256-tile wgrads two slabs on the plain kernel -- fc1 wgrad 3072x768, 144 tiles: 26.5 vs 29.1 us; since the
        # wave-specialised loop hides its first-fragment latency one slab wins: 24.5 vs 28.1 us, MAE pretrain step 8.78 -> 8.66 ms)
        # (capping the slabs at 8 / 4 to save partial traffic: step 4.82 -> 4.96 / 5.42 ms -- the split is what fills the chip)
        return max(1, min(64, ktiles, 256 // max(tiles, 1)))

    # linear backward pieces.  dy [M,N] (T), x [M,K] (T)
    def _linear_wgrad(self, ops, dy, x, wname, M, N, K, lddy=None, ldx=None, bias_name=None):
        """dW[N,K] = dy^T x (split-K slabs + deterministic reduce).  bias_name: also produce db = column sums of dy,
        fused into the GEMM as one extra MFMA per tile against a ones fragment (bf16 mode)."""
        lddy = N if lddy is None else lddy
        ldx = K if ldx is None else ldx
        bk = 64 if self.code == BF16 else 32
        kw = dict(A=dy.data_ptr() if not isinstance(dy, int) else dy, B=x.data_ptr() if not isinstance(x, int) else x,
                  lda=lddy, ldb=ldx, ldc=K, M=N, N=K, K=M)
        q = GemmArgs()
        for k_, v_ in kw.items():
            setattr(q, k_, v_)
        q.alpha, q.nbatch, q.nb1 = 1.0, 1, 1
        tiles = int(self.L.countr_gemm_tiles(C.byref(q), self.code, OP_COL, OP_COL))     # 128x128, or 128x256 on the lean kernel (conv_wgrad.hip)
        sk = self._splitk(tiles, -(-M // bk))
        part = self._shared("skp." + self._role(wname), sk * N * K)
        fuse_bias = bias_name is not None and self.code == BF16
        rs = self._shared("rsp." + self._role(wname), 64 * 4096) if fuse_bias else None
        self._claim(part.data_ptr())
        kw.update(partial=part.data_ptr(), splitk=sk)
        rslabs = sk
        if fuse_bias:
            q.splitk = sk
            n = int(self.L.countr_gemm_rowsum_slabs(C.byref(q), self.code, OP_COL, OP_COL))
            if n * N <= 64 * 4096:
                rslabs = n
            kw.update(rowsum_partial=rs.data_ptr(), rowsum_slabs=rslabs)
        self._gemm(ops, self.code, OP_COL, OP_COL, **kw)
        # slab sums are deferred to the list's table-driven launches (countr_reduce_table)
        self._reduce_later(ops, part.data_ptr(), part.data_ptr(), self._gp(wname), sk, N * K, N * K)
        if fuse_bias:
            self._reduce_later(ops, part.data_ptr(), rs.data_ptr(), self._gp(bias_name), rslabs, N, N)
        if bias_name is not None and not fuse_bias:
            self._bias_grad(ops, dy, bias_name, M, N)

    def _linear_wgrad_group(self, ops, items):
        """The weight (+ bias) gradients of several nn.Linear layers -- items: (dy, x, wname, M, N, K, bias_name), all final by now -- in
        ONE launch (countr_gemm_group) with a COMMON split-K factor chosen for the group's tile count: the four weight gradients of a
        transformer block are 16-72 tiles each and needed 3-16 slabs apiece to fill the chip; together one or two.  With one slab and
        no accumulation window the kernel writes the gradient itself (no slab sum).  fp32 mode, or a group the library would run as
        separate launches anyway: the per-layer path."""
        n = len(items)
        arr = (GemmArgs * n)() if 2 <= n <= 10 and self.code == BF16 and self.group_wgrads else None
        tiles = 0
        if arr is not None:
            for q, (dy, x, wname, M, N, K, _b) in zip(arr, items):
                q.A, q.B, q.lda, q.ldb, q.ldc, q.M, q.N, q.K = dy.data_ptr(), x.data_ptr(), N, K, K, N, K, M
                q.alpha, q.nbatch, q.nb1, q.splitk = 1.0, 1, 1, 1
            tiles = int(self.L.countr_gemm_group_tiles(arr, n, self.code, OP_COL, OP_COL))
        if tiles <= 0:
            for (dy, x, wname, M, N, K, bias_name) in items:
                self._linear_wgrad(ops, dy, x, wname, M, N, K, bias_name=bias_name)
            return
        sk = self._splitk(tiles, -(-max(it[3] for it in items) // 64))
        if any(b_ is not None and sk * (K // 128) * N > 64 * 4096 for (_dy, _x, _w, _M, N, K, b_) in items):   # bias-gradient slabs beyond their workspace
            for (dy, x, wname, M, N, K, bias_name) in items:
                self._linear_wgrad(ops, dy, x, wname, M, N, K, bias_name=bias_name)
            return
        direct = sk == 1 and not self._acc
        later = []
        # (workspaces per position in the group: _role() maps mlp.fc1 and mlp.fc2 to ONE name, fine for launches that follow each other)
        for i, (q, (dy, x, wname, M, N, K, bias_name)) in enumerate(zip(arr, items)):     # every claim before the launch, every deferred sum behind it
            q.splitk = sk
            if direct:
                q.partial = self._gp(wname)
            else:
                part = self._shared("skg%d." % i + self._role(wname), sk * N * K)
                self._claim(part.data_ptr())
                q.partial = part.data_ptr()
                later.append((part.data_ptr(), part.data_ptr(), self._gp(wname), sk, N * K, N * K))
            if bias_name is not None:
                # (1 MB per layer, not per role: with the weights' slabs gone -- `direct` -- nothing else would force the pending sums out per block)
                rs = self._shared("rsg." + wname, 64 * 4096)
                rslabs = int(self.L.countr_gemm_rowsum_slabs(C.byref(q), self.code, OP_COL, OP_COL))
                assert rslabs * N <= 64 * 4096, (wname, rslabs, N)
                self._claim(rs.data_ptr())
                q.rowsum_partial, q.rowsum_slabs = rs.data_ptr(), rslabs
                later.append((rs.data_ptr(), rs.data_ptr(), self._gp(bias_name), rslabs, N, N))
        ops.append((self.L.countr_gemm_group, (arr, n, self.code, OP_COL, OP_COL), arr))
        for e in later:
            self._reduce_later(ops, *e)

    def _bias_grad(self, ops, dy, bname, M, N):
        ws = self._shared("colsum", 256 * 4096)
        self._op(ops, self.L.countr_colsum, dy.data_ptr(), self._gp(bname), ws.data_ptr(), M, N, self.code, self._acc)

    def _linear_dgrad(self, ops, dy, wname, dx, M, N, K, resid=None, out_bf16=None, gelu_pre=None):
        """dx[M,K] = dy[M,N] W[N,K] (+ resid).  gelu_pre (the saved pre-activation h of x = GELU(h), layout of dx): dx *= GELU'(h) --
        in the GEMM's epilogue where the lean kernel serves the launch (bf16, transposed shadow), otherwise as a separate pass.
        (Measured, profiles/r4_gelu_bwd_fuse_ab.txt: the derivative costs the GEMM's epilogue 6.6-10.9 us -- 2 transcendentals + 15 VALU per
        element with the matrix pipes idle -- against 8-12 us for the separate bandwidth-bound pass: 20 launches fewer, ~1 us each saved.)"""
        out_bf16 = (dx.dtype in HALF_DTYPES) if out_bf16 is None else out_bf16
        if wname in self.WtT:      # W^T [K][N]: (ROW, ROW), the lean kernel
            fuse = gelu_pre is not None and self.code == BF16 and out_bf16 and resid is None
            self._gemm(ops, self.code, OP_ROW, OP_ROW, A=dy.data_ptr(), B=self.WtT[wname].data_ptr(), C=dx.data_ptr(),
                       resid=(resid.data_ptr() if resid is not None else None), lda=N, ldb=N, ldc=K, ldres=K, M=M, N=K, K=N,
                       out_bf16=int(out_bf16), **({"act": ACT_GELU_BWD, "C2": gelu_pre.data_ptr()} if fuse else {}))
            if gelu_pre is not None and not fuse:
                self._op(ops, self.L.countr_gelu_bwd, dx.data_ptr(), gelu_pre.data_ptr(), dx.data_ptr(), M * K, self.code)
            return
        self._gemm(ops, self.code, OP_ROW, OP_COL, A=dy.data_ptr(), B=self._wp(wname), C=dx.data_ptr(),
                   resid=(resid.data_ptr() if resid is not None else None), lda=N, ldb=K, ldc=K, ldres=K, M=M, N=K, K=N,
                   out_bf16=int(out_bf16))
        if gelu_pre is not None:
            self._op(ops, self.L.countr_gelu_bwd, dx.data_ptr(), gelu_pre.data_ptr(), dx.data_ptr(), M * K, self.code)

    def _linear_bwd(self, ops, dy, x, wname, M, N, K, dx=None, resid=None, dx_bf16=None, gelu_pre=None, group=None):
        """bias grad | weight grad | input grad of one nn.Linear: three independent branches.  group (a list): the weight / bias
        gradient is not launched here but handed to the caller's _linear_wgrad_group (dy and x must stay untouched until then)."""
        self._fork(ops)
        self._lane(ops, 2)
        if group is not None:
            group.append((dy, x, wname, M, N, K, wname[:-6] + "bias"))
        else:
            self._linear_wgrad(ops, dy, x, wname, M, N, K, bias_name=wname[:-6] + "bias")
        if dx is not None:
            self._lane(ops, 0)
            self._linear_dgrad(ops, dy, wname, dx, M, N, K, resid=resid, out_bf16=dx_bf16, gelu_pre=gelu_pre)
        self._join(ops)

    def _cast(self, ops, src_f32, dst_t, n):
        if self.code == F32:
            return src_f32
        self._op(ops, self.L.countr_cast_permute, src_f32.data_ptr(), dst_t.data_ptr(), n, 0, 0, 0, 0, BF16)
        return dst_t

    def _layernorm(self, ops, x, name, y, rows, D, mean=None, rstd=None):
        self._op(ops, self.L.countr_layernorm_fwd, x.data_ptr(), self._pp(name + ".weight"), self._pp(name + ".bias"), y.data_ptr(),
                 mean.data_ptr() if mean is not None else None, rstd.data_ptr() if rstd is not None else None, rows, D, self.ln_eps,
                 int(y.dtype in HALF_DTYPES))

    def _layernorm_bwd(self, ops, dy, x, name, mean, rstd, dx, rows, D, accumulate, dx_t=None):
        """dx_t (bf16 mode): also emit the updated residual gradient as the bf16 operand of the next backward GEMM."""
        # per-block {dgamma, dbeta} partials stay in this LayerNorm's own workspace until the list's table launch
        nb = self.L.countr_layernorm_bwd_nblocks()
        # (one workspace per LayerNorm, ~1 MB: with a name shared by role every LayerNorm backward forced the pending sums out -- seven
        # table launches per step where the list's end and the grouped weight gradients' buffer reuse need two)
        ws = self._shared("lnw." + name, nb * 2 * D)
        self._claim(ws.data_ptr())
        self._op(ops, self.L.countr_layernorm_bwd, dy.data_ptr(), x.data_ptr(), self._pp(name + ".weight"), mean.data_ptr(),
                 rstd.data_ptr(), dx.data_ptr(), None, None, ws.data_ptr(), rows, D,
                 int(dy.dtype in HALF_DTYPES), int(accumulate), 0, dx_t.data_ptr() if dx_t is not None else None)
        self._reduce_later(ops, ws.data_ptr(), ws.data_ptr(), self._gp(name + ".weight"), nb, 2 * D, D)
        self._reduce_later(ops, ws.data_ptr(), ws.data_ptr() + 4 * D, self._gp(name + ".bias"), nb, 2 * D, D)
        return dx if self.code == F32 else dx_t

    # unfused self-attention forward on a packed qkv [rows, 3*Dm]
    def _fused_attention(self, dh):
        return self.code == BF16 and dh in (32, 64) and self.attention != "unfused"

    def _attention_fwd(self, ops, plan, qkv, out, B, heads, Dm, probs=None, lse=None, N=None, prescaled=False):
        N, dh = (self.N if N is None else N), Dm // heads
        scale = dh ** -0.5
        if self._fused_attention(dh) and probs is None:
            self._op(ops, self.L.countr_attn_fwd, qkv.data_ptr(), out.data_ptr(), lse.data_ptr() if lse is not None else None, B, N,
                     heads, dh, 0.0 if prescaled else scale)   # scale <= 0: q carries dh^-0.5 * log2(e) already
            return
        assert not prescaled
        # Np > N (a token count that is not a multiple of the GEMM's 16-byte chunk; forward-only configurations): the score and
        # probability matrices get Np columns.  The extra score columns are products with the rows BEHIND this image's keys (the next
        # image's first tokens, or the zeroed pad rows of the qkv buffer), the softmax writes zeros there, and the PV product multiplies
        # those zeros with the (finite) rows behind this image's values.
        Np = N if N % 8 == 0 else -(-N // 8) * 8
        assert Np == N or probs is None
        scores = self._shared("scores", B * heads * N * Np)
        if probs is None:
            probs = self._shared("probs", B * heads * N * Np, self.tdt)
        es = qkv.element_size()
        self._gemm(ops, self.code, OP_ROW, OP_ROW, A=qkv.data_ptr(), B=qkv.data_ptr() + Dm * es, C=scores.data_ptr(),
                   lda=3 * Dm, ldb=3 * Dm, ldc=Np, M=N, N=Np, K=dh, nbatch=B * heads, nb1=heads, sA0=N * 3 * Dm, sA1=dh,
                   sB0=N * 3 * Dm, sB1=dh, sC0=heads * N * Np, sC1=N * Np, alpha=scale, out_bf16=0)
        self._op(ops, self.L.countr_softmax_fwd_ld, scores.data_ptr(), probs.data_ptr(), B * heads * N, N, Np, int(self.code == BF16))
        self._gemm(ops, self.code, OP_ROW, OP_COL, A=probs.data_ptr(), B=qkv.data_ptr() + 2 * Dm * es, C=out.data_ptr(),
                   lda=Np, ldb=3 * Dm, ldc=Dm, M=N, N=dh, K=Np, nbatch=B * heads, nb1=heads, sA0=heads * N * Np, sA1=N * Np,
                   sB0=N * 3 * Dm, sB1=dh, sC0=N * Dm, sC1=dh, out_bf16=int(self.code == BF16))

    def _attention_bwd(self, ops, qkv, probs, dout, dqkv, B, heads, Dm, N=None):
        """dout [rows, Dm] (T) -> dqkv [rows, 3*Dm] (T); probs [B,h,N,N] (T) saved by the forward."""
        N, dh = (self.N if N is None else N), Dm // heads
        scale = dh ** -0.5
        es = qkv.element_size()
        dP = self._shared("scores", B * heads * N * N)
        dS = self._shared("probs", B * heads * N * N, self.tdt)
        ob = int(self.code == BF16)
        nb = dict(nbatch=B * heads, nb1=heads)
        self._fork(ops)
        self._lane(ops, 1)
        # dV[j,d] = sum_i P[i,j] dO[i,d]
        self._gemm(ops, self.code, OP_COL, OP_COL, A=probs.data_ptr(), B=dout.data_ptr(), C=dqkv.data_ptr() + 2 * Dm * es,
                   lda=N, ldb=Dm, ldc=3 * Dm, M=N, N=dh, K=N, sA0=heads * N * N, sA1=N * N, sB0=N * Dm, sB1=dh,
                   sC0=N * 3 * Dm, sC1=dh, out_bf16=ob, **nb)
        self._lane(ops, 0)
        # dP[i,j] = sum_d dO[i,d] V[j,d]
        self._gemm(ops, self.code, OP_ROW, OP_ROW, A=dout.data_ptr(), B=qkv.data_ptr() + 2 * Dm * es, C=dP.data_ptr(),
                   lda=Dm, ldb=3 * Dm, ldc=N, M=N, N=N, K=dh, sA0=N * Dm, sA1=dh, sB0=N * 3 * Dm, sB1=dh,
                   sC0=heads * N * N, sC1=N * N, out_bf16=0, **nb)
        self._op(ops, self.L.countr_softmax_bwd, probs.data_ptr(), dP.data_ptr(), dS.data_ptr(), B * heads * N, N, scale, self.code)
        self._join(ops)
        self._fork(ops)
        self._lane(ops, 1)
        # dQ[i,d] = sum_j dS[i,j] K[j,d]
        self._gemm(ops, self.code, OP_ROW, OP_COL, A=dS.data_ptr(), B=qkv.data_ptr() + Dm * es, C=dqkv.data_ptr(),
                   lda=N, ldb=3 * Dm, ldc=3 * Dm, M=N, N=dh, K=N, sA0=heads * N * N, sA1=N * N, sB0=N * 3 * Dm, sB1=dh,
                   sC0=N * 3 * Dm, sC1=dh, out_bf16=ob, **nb)
        self._lane(ops, 0)
        # dK[j,d] = sum_i dS[i,j] Q[i,d]
        self._gemm(ops, self.code, OP_COL, OP_COL, A=dS.data_ptr(), B=qkv.data_ptr(), C=dqkv.data_ptr() + Dm * es,
                   lda=N, ldb=3 * Dm, ldc=3 * Dm, M=N, N=dh, K=N, sA0=heads * N * N, sA1=N * N, sB0=N * 3 * Dm, sB1=dh,
                   sC0=N * 3 * Dm, sC1=dh, out_bf16=ob, **nb)
        self._join(ops)

    def _act_splitk(self, HW, K):
        """Split-K factor for a forward-type convolution (bf16 mode).  On the small maps (<= 24x24 per image) the implicit GEMM has
        few 128x128 tiles and a long K -- decode_head0 forward: 72 tiles x 72 k-tiles = 44 us on a quarter of the CUs; exemplar conv4
        dgrad: 24 x 72 = 43 us -- so K is cut over the idle CUs and a finisher pass (countr_splitk_finish) adds the slabs.  The factor
        depends on the LAYER only (map size, K), never on the batch: the fp32 summation tree of a sample is the same in every batch
        (tests/test_properties_gpu.py: sample i alone == sample i in a batch, bit for bit).  1 = no split."""
        if self.code != BF16 or not self.act_splitk or HW > 576:
            return 1
        ktiles = K // 64
        return 8 if ktiles >= 64 else (4 if ktiles >= 32 else 1)

    # 3x3 conv (NHWC, pad 1) as implicit GEMM (forward, and dgrad with the dgrad-form weights)
    def _conv_fwd(self, ops, x, w_ohwi, bias_ptr, out, Bn, H, W, Cin, Cout, gn_rows=None):
        """gn_rows (a callable -> fp32 buffer [Bn * H * W, Cout / 32, 2], or None): ask for the GroupNorm row partials of the rounded
        output from the convolution's epilogue (countr_gemm_args.gn_rows).  Returns the buffer when the launch runs on a kernel with
        that epilogue (countr_gemm_gn_rows: the lean 16-bit kernels on maps of more than 256 tiles), else None -- the caller then keeps
        the statistics pass over the map."""
        M, K = Bn * H * W, 9 * Cin
        sk = self._act_splitk(H * W, K)
        if sk == 1 and gn_rows is not None and self.code == BF16 and os.environ.get("COUNTR_GN_ROWS", "1") != "0":
            q = GemmArgs()
            kw = dict(A=x.data_ptr() or 16, B=w_ohwi.data_ptr() or 16, C=out.data_ptr() or 16, bias=bias_ptr, ldb=K, ldc=Cout, M=M, N=Cout, K=K,
                      H=H, W=W, Cin=Cin, out_bf16=int(out.dtype in HALF_DTYPES), alpha=1.0, nbatch=1, nb1=1, splitk=1)
            for k_, v_ in kw.items():
                setattr(q, k_, v_)
            if int(self.L.countr_gemm_gn_rows(C.byref(q), self.code, OP_IM2ROW, OP_ROW)) == 1:
                rows = gn_rows()
                self._gemm(ops, self.code, OP_IM2ROW, OP_ROW, A=x.data_ptr(), B=w_ohwi.data_ptr(), C=out.data_ptr(), bias=bias_ptr,
                           ldb=K, ldc=Cout, M=M, N=Cout, K=K, H=H, W=W, Cin=Cin, out_bf16=int(out.dtype in HALF_DTYPES), gn_rows=rows.data_ptr())
                return rows
        if sk > 1:
            part = self._shared("actsk", sk * M * Cout)
            self._gemm(ops, self.code, OP_IM2ROW, OP_ROW, A=x.data_ptr(), B=w_ohwi.data_ptr(), partial=part.data_ptr(), ldb=K, ldc=Cout,
                       M=M, N=Cout, K=K, H=H, W=W, Cin=Cin, splitk=sk)
            self._op(ops, self.L.countr_splitk_finish, part.data_ptr(), out.data_ptr(), bias_ptr, sk, M, Cout, int(out.dtype in HALF_DTYPES))
            return None
        self._gemm(ops, self.code, OP_IM2ROW, OP_ROW, A=x.data_ptr(), B=w_ohwi.data_ptr(), C=out.data_ptr(), bias=bias_ptr,
                   ldb=K, ldc=Cout, M=M, N=Cout, K=K, H=H, W=W, Cin=Cin, out_bf16=int(out.dtype in HALF_DTYPES))
        return None

    def _conv_wgrad(self, ops, dy, x, wname, Bn, H, W, Cin, Cout, bias_name=None):
        bk = 64 if self.code == BF16 else 32
        Kp = Bn * H * W
        q = GemmArgs()
        q.M, q.N, q.K, q.H, q.W, q.Cin, q.lda, q.ldc, q.alpha, q.nbatch, q.nb1 = Cout, 9 * Cin, Kp, H, W, Cin, Cout, 9 * Cin, 1.0, 1, 1
        tiles = int(self.L.countr_gemm_tiles(C.byref(q), self.code, OP_COL, OP_IM2COL))   # 128x128, or 128x256 on the lean kernel
        sk = self._splitk(tiles, -(-Kp // bk))
        defer = self.defer_reduce
        cw_key = wname     # one partial buffer per convolution (<= 130 MB each): with a buffer per ROLE every weight gradient forced the pending sums out
        part = self._shared(("skp." + cw_key) if defer else "splitk", sk * Cout * 9 * Cin)
        fuse_bias = bias_name is not None and self.code == BF16
        rs = self._shared(("rsp." + cw_key) if defer else "rowsum", 64 * 4096) if fuse_bias else None
        if defer:
            self._claim(part.data_ptr())
        kw = dict(A=dy.data_ptr(), B=x.data_ptr(), partial=part.data_ptr(), lda=Cout, ldc=9 * Cin, M=Cout, N=9 * Cin, K=Kp, H=H, W=W,
                  Cin=Cin, splitk=sk)
        rslabs = sk
        if fuse_bias:
            # the lean weight-gradient kernel (conv_wgrad.hip) deals the bias-gradient work over more waves: more, thinner slabs.  Only
            # the table-driven finisher sums an arbitrary slab count; countr_splitk_reduce keeps the [splitk][M] layout (rowsum_slabs 0)
            q = GemmArgs()
            for k_, v_ in kw.items():
                setattr(q, k_, v_)
            q.alpha, q.nbatch, q.nb1 = 1.0, 1, 1
            n = int(self.L.countr_gemm_rowsum_slabs(C.byref(q), self.code, OP_COL, OP_IM2COL)) if defer else sk
            if n * Cout <= 64 * 4096:
                rslabs = n
            kw.update(rowsum_partial=rs.data_ptr(), rowsum_slabs=(rslabs if defer else 0))
        self._gemm(ops, self.code, OP_COL, OP_IM2COL, **kw)
        if defer:
            self._reduce_later(ops, part.data_ptr(), part.data_ptr(), self._gp(wname), sk, Cout * 9 * Cin, Cout * 9 * Cin, N=9 * Cin, taps=9)
            if fuse_bias:
                self._reduce_later(ops, part.data_ptr(), rs.data_ptr(), self._gp(bias_name), rslabs, Cout, Cout)
        else:
            self._op(ops, self.L.countr_splitk_reduce, part.data_ptr(), self._gp(wname), sk, Cout, 9 * Cin, 9, self._acc,
                     rs.data_ptr() if fuse_bias else None, self._gp(bias_name) if fuse_bias else None)
        if bias_name is not None and not fuse_bias:
            self._bias_grad(ops, dy, bias_name, Kp, Cout)

    # ------------------------------------------------------------------ plan construction
    def plan(self, B, S, train):
        key = (B, S, bool(train))
        if train and self.forward_only:
            raise _lib.CountrError("this configuration (patch %d on %d pixels, head_dim %d) runs forward-only: its density map is %d x %d, "
                                   "which the reference's training loss (FSC_finetune_cross.py:294, against a %d x %d ground truth) cannot "
                                   "consume either" % (self.patch, self.img, self.D // self.H, 16 * self.grid, 16 * self.grid, self.img, self.img))
        if key not in self.plans:
            # size the shared scratch for EVERY shot count and both modes of this batch size at once, so that building
            # another plan later never moves scratch that launch lists (and captured graphs) already point to
            self._sizing = True
            try:
                for s_ in sorted({0, 1, 2, 3, S}):
                    for tr in ((False,) if self.forward_only else (False, True)):
                        self._build(B, s_, tr)
            finally:
                self._sizing = False
            self._reserve()
            self.plans[key] = self._build(B, S, bool(train))
        return self.plans[key]

    def _build(self, B, S, train):
        L = self.L
        p = Plan()
        T, f32 = self.tdt, torch.float32
        N, D, Dd, H, Hd = self.N, self.D, self.Dd, self.H, self.Hd
        rows = B * N
        code = self.code
        ops = p.fwd
        A = lambda k, shape, dt: self._alloc(p, k, shape, dt)

        # ---------------- encoder (no grad): models_mae_cross.py:136-148
        img = A("img", (B, 3, self.img, self.img), f32)
        # (a patch row that is not a whole number of 16-byte chunks in 16-bit storage -- patch 14: K = 588 -- keeps the patch matrix and
        # its product in fp32: mae_vit_huge_patch14's forward-only configuration)
        pe_code = code if (3 * self.patch * self.patch) % 8 == 0 else F32
        patches = A("patches", (rows, 3 * self.patch * self.patch), T if pe_code == code else f32)
        x = A("x", (rows, D), f32)
        xn = A("xn", (rows, D), T)
        pad = self.Np - N       # rows behind the last image's tokens that the padded attention products read (zeros, never written)
        qkv = A("qkv", (rows + pad, 3 * D), T)
        if pad and not self._sizing:
            qkv[rows:].zero_()
        att = A("att", (rows, D), T)
        hid = A("hid", (rows, 4 * D), T)
        latent = A("latent", (rows, D), T)
        self._op(ops, L.countr_im2patch, img.data_ptr(), patches.data_ptr(), B, self.img, self.img, self.patch, pe_code)
        Kp = 3 * self.patch * self.patch
        pre_q, fold = self.prescale_q, self.ln_fold
        if (pre_q or fold) and self.qkv_bias_pre is None:
            self._pack_prescaled_q()
        # LayerNorm folding: every producer of the residual stream x (patch embed, proj, fc2) also leaves its bf16 copy in xn and the
        # row partials in lnst; the consumers (qkv, fc1) read xn and normalise in their epilogue
        lnst = A("lnstats", (rows, D // 64, 2), f32) if fold else None
        prod = dict(ln_xcopy=xn.data_ptr(), ln_stats_out=lnst.data_ptr()) if fold else {}
        cons = (lambda cvec: dict(ln_stats=lnst.data_ptr(), ln_colsum=cvec.data_ptr(), ln_nblk=D // 64, ln_eps=self.ln_eps)) if fold else (lambda cvec: {})
        self._gemm(ops, pe_code, OP_ROW, OP_ROW, A=patches.data_ptr(),
                   B=self._wp("patch_embed.proj.weight") if pe_code == code else self._pp("patch_embed.proj.weight"), C=x.data_ptr(),
                   bias=self._pp("patch_embed.proj.bias"), resid=self._pp("pos_embed"), lda=Kp, ldb=Kp, ldc=D, ldres=D,
                   M=rows, N=D, K=Kp, res_mod=N, out_bf16=0, **prod)
        for i in range(self.depth):
            b = "blocks.%d" % i
            if not fold:
                self._layernorm(ops, x, b + ".norm1", xn, rows, D)
            self._linear(ops, xn, b + ".attn.qkv.weight", qkv, rows, 3 * D, D,
                         bias_ptr=self.qkv_bias_pre[i].data_ptr() if (pre_q or fold) else None, **cons(self.ln_c_qkv[i] if fold else None))
            self._attention_fwd(ops, p, qkv, att, B, H, D, prescaled=pre_q)
            self._linear(ops, att, b + ".attn.proj.weight", x, rows, D, D, resid=x, **prod)
            if not fold:
                self._layernorm(ops, x, b + ".norm2", xn, rows, D)
            self._linear(ops, xn, b + ".mlp.fc1.weight", hid, rows, 4 * D, D, act=ACT_GELU,
                         bias_ptr=self.fc1_bias_pre[i].data_ptr() if fold else None, **cons(self.ln_c_fc1[i] if fold else None))
            self._linear(ops, hid, b + ".mlp.fc2.weight", x, rows, D, 4 * D, resid=x, **({} if i + 1 == self.depth else prod))
        self._layernorm(ops, x, "norm", latent, rows, D)
        p.enc_ops = len(ops)
        # Pipelined-encoder variant (trainer.FinetuneStep(pipeline_encoder=True); inference.density_maps_stream): the SAME launches reading the next batch's images from a
        # buffer shared by all plans and leaving its latent in another one, so that they can run on their own lane beside this batch's
        # decoder side (which keeps reading this plan's `latent`, down to decoder_embed's weight gradient at the end of the backward).
        # 16-bit modes only: the fp32 parity mode's unfused attention goes through scratch it shares with the decoder's.
        p.enc_pipe = None
        if code != F32 and self._fused_attention(D // H):
            pimg = self._shared("pipe_img", B * 3 * self.img * self.img)
            plat = self._shared("pipe_latent", rows * D, T)
            head, tail = [], []
            self._op(head, L.countr_im2patch, pimg.data_ptr(), patches.data_ptr(), B, self.img, self.img, self.patch, code)
            self._layernorm(tail, x, "norm", plat, rows, D)
            assert ops[0][0] is L.countr_im2patch and ops[-1][0] is L.countr_layernorm_fwd
            p.enc_pipe = head + ops[1:-1] + tail
            p.pipe_img, p.pipe_latent, p.pipe_latent_bytes = pimg, plat, rows * D * (2 if T in HALF_DTYPES else 4)

        # ---------------- decoder: models_mae_cross.py:150-199
        Sy = max(S, 1)
        xs = [A("dx0", (rows, Dd), f32)]
        self._linear(ops, latent, "decoder_embed.weight", xs[0], rows, Dd, D, resid=self._pp("decoder_pos_embed"), res_mod=N)
        # (rows padded to a multiple of 64 with zeros, never written: as a weight gradient's operand the token matrix is then whole k-tiles,
        # so attn.wk / attn.wv join the block's grouped weight-gradient launch instead of four 16-workgroup launches of their own)
        tok_rows = -(-(B * Sy) // 64) * 64
        ytok = A("ytok", (tok_rows, Dd), T)
        if not self._sizing:
            ytok.zero_()
        if S == 0:
            # y = shot_token broadcast over the batch (models_mae_cross.py:176): ONE row gather with an all-zero index (it was one cast
            # launch per batch element: 32 tiny launches in front of every B = 32 inference forward)
            zidx = A("ytok_idx", (B,), torch.int32)
            if not self._sizing:
                zidx.zero_()
            self._op(ops, L.countr_gather_rows, self._pp("shot_token"), zidx.data_ptr(), ytok.data_ptr(), None, None, 0, B, Dd, F32, code)
        else:
            ex0 = len(ops)
            BS = B * S
            boxes = A("boxes", (BS, 3, 64, 64), f32)
            chans = [64, 128, 256, Dd]
            sizes = [64, 32, 16, 8]
            # bf16 engine: the exemplar CNN's convolutions write fp32 (c*) and the InstanceNorm stage turns them into bf16 pooled
            # activations and -- training plans -- the bf16 NORMALISED maps (ch*) its backward reads.  A bf16 conv output is rounded at
            # 2^-9 of its value; on channels whose mean is several sigma that moves pixels across the ReLU boundary of the normalised
            # map, and the InstanceNorm backward turned it into cos 0.975 per layer / 0.96 for the CNN's weight gradients against fp32
            # (tools/diag_exemplar_bf16.py)
            xh = bool(code == BF16)
            p.in_xhat = xh
            c = [A("c%d" % (i + 1), (BS, sizes[i], sizes[i], chans[i]), f32 if xh else T) for i in range(4)]
            ch = [A("ch%d" % (i + 1), (BS, sizes[i], sizes[i], chans[i]), T) for i in range(4)] if (xh and train) else None
            p.in_maps = ch if ch is not None else c
            pl = [A("p%d" % (i + 1), (BS, sizes[i] // 2, sizes[i] // 2, chans[i]), T) for i in range(3)]
            stats = [A("instats%d" % (i + 1), (BS, chans[i], 2), f32) for i in range(4)]
            self._op(ops, L.countr_conv3x3_c3_fwd, boxes.data_ptr(), self._pp("decoder_proj1.0.weight"), self._pp("decoder_proj1.0.bias"),
                     c[0].data_ptr(), BS, 64, 64, F32 if xh else code)
            in_ws = self._shared("in_ws", L.countr_instnorm_workspace_floats(BS, Dd))
            self._op(ops, L.countr_instnorm_relu_pool_fwd, c[0].data_ptr(), pl[0].data_ptr(), stats[0].data_ptr(), BS, 64, 64, 64, 0, 1e-5,
                     code, in_ws.data_ptr(), ch[0].data_ptr() if ch is not None else None, int(xh))
            for i in (1, 2, 3):
                wn = "decoder_proj%d.0.weight" % (i + 1)
                self._conv_fwd(ops, pl[i - 1], self.Wf[wn], self._pp(wn[:-6] + "bias"), c[i], BS, sizes[i], sizes[i], chans[i - 1], chans[i])
                last = i == 3
                self._op(ops, L.countr_instnorm_relu_pool_fwd, c[i].data_ptr(), (ytok if last else pl[i]).data_ptr(), stats[i].data_ptr(),
                         BS, sizes[i], sizes[i], chans[i], int(last), 1e-5, code, in_ws.data_ptr(), ch[i].data_ptr() if ch is not None else None, int(xh))
        # the cross-attention keys / values depend on the exemplar tokens only: all blocks' wk / wv projections (tiny GEMMs, 4 tiles each)
        # follow the tokens directly -- with exemplars that is inside the side lane that runs beside the encoder
        # wk and wv of a block lie side by side in the parameter buffer (weights and biases alike: ParamLayout): ONE product per block
        # writes k | v as the two column halves of a [B * S, 2 Dd] buffer (countr_xattn_*'s ldkv) -- two launches less on the exemplar lane,
        # which heads the decoder's critical path in the pipelined forms
        kv, lay = [], self.layout
        for i in range(self.ddepth):
            b = "decoder_blocks.%d" % i
            wk, wv = b + ".attn.wk.weight", b + ".attn.wv.weight"
            joint = (lay.off[wv] == lay.off[wk] + Dd * Dd and lay.off[wv[:-6] + "bias"] == lay.off[wk[:-6] + "bias"] + Dd
                     and os.environ.get("COUNTR_JOINT_KV", "1") != "0")
            if joint:
                kvb = A(b + ".kv", (B * Sy, 2 * Dd), T)
                self._linear(ops, ytok, wk, kvb, B * Sy, 2 * Dd, Dd)
                kv.append((_Cols(kvb, 0), _Cols(kvb, Dd), 2 * Dd))
            else:
                k_, v_ = A(b + ".k", (B * Sy, Dd), T), A(b + ".v", (B * Sy, Dd), T)
                self._linear(ops, ytok, wk, k_, B * Sy, Dd, Dd)
                self._linear(ops, ytok, wv, v_, B * Sy, Dd, Dd)
                kv.append((k_, v_, Dd))
        if S > 0:
            p.ex_range = (ex0, len(ops))
        blk = []
        for i in range(self.ddepth):
            b = "decoder_blocks.%d" % i
            d = {}
            xin = xs[-1]
            d["n0"] = A(b + ".n0", (rows, Dd), T)
            d["m0"], d["r0"] = A(b + ".m0", (rows,), f32), A(b + ".r0", (rows,), f32)
            d["qkv"] = A(b + ".qkv", (rows + pad, 3 * Dd), T)
            if pad and not self._sizing:
                d["qkv"][rows:].zero_()
            fused = self._fused_attention(Dd // Hd)
            d["probs"] = A(b + ".probs", (B * Hd * N * N,), T) if (train and not fused) else None
            d["lse"] = A(b + ".lse", (B * Hd * N,), f32) if (train and fused) else None
            d["att"] = A(b + ".att", (rows, Dd), T)
            self._layernorm(ops, xin, b + ".norm0", d["n0"], rows, Dd, d["m0"], d["r0"])
            self._linear(ops, d["n0"], b + ".selfattn.qkv.weight", d["qkv"], rows, 3 * Dd, Dd)
            self._attention_fwd(ops, p, d["qkv"], d["att"], B, Hd, Dd, probs=d["probs"], lse=d["lse"])
            x1 = A(b + ".x1", (rows, Dd), f32)
            self._linear(ops, d["att"], b + ".selfattn.proj.weight", x1, rows, Dd, Dd, resid=xin)
            d["n1"] = A(b + ".n1", (rows, Dd), T)
            d["m1"], d["r1"] = A(b + ".m1", (rows,), f32), A(b + ".r1", (rows,), f32)
            d["q"] = A(b + ".q", (rows, Dd), T)
            d["k"], d["v"], d["ldkv"] = kv[i]
            d["xo"] = A(b + ".xo", (rows, Dd), T)
            self._layernorm(ops, x1, b + ".norm1", d["n1"], rows, Dd, d["m1"], d["r1"])
            self._linear(ops, d["n1"], b + ".attn.wq.weight", d["q"], rows, Dd, Dd)
            if i == 0:
                p.first_xattn = len(ops)      # the first launch that reads the exemplar tokens' keys / values (decoder_ops_with_exemplar_lane)
            self._op(ops, L.countr_xattn_fwd, d["q"].data_ptr(), d["k"].data_ptr(), d["v"].data_ptr(), d["xo"].data_ptr(), B, N, Sy, Dd,
                     Hd, d["ldkv"], (Dd // Hd) ** -0.5, code)
            x2 = A(b + ".x2", (rows, Dd), f32)
            self._linear(ops, d["xo"], b + ".attn.proj.weight", x2, rows, Dd, Dd, resid=x1)
            d["n2"] = A(b + ".n2", (rows, Dd), T)
            d["m2"], d["r2"] = A(b + ".m2", (rows,), f32), A(b + ".r2", (rows,), f32)
            d["hpre"] = A(b + ".hpre", (rows, 4 * Dd), T)
            d["hact"] = A(b + ".hact", (rows, 4 * Dd), T)
            self._layernorm(ops, x2, b + ".norm2", d["n2"], rows, Dd, d["m2"], d["r2"])
            self._linear(ops, d["n2"], b + ".mlp.fc1.weight", d["hact"], rows, 4 * Dd, Dd, act=ACT_GELU, pre=d["hpre"])
            x3 = A(b + ".x3", (rows, Dd), f32)
            self._linear(ops, d["hact"], b + ".mlp.fc2.weight", x3, rows, Dd, 4 * Dd, resid=x2)
            d["xin"], d["x1"], d["x2"] = xin, x1, x2
            xs.append(x3)
            blk.append(d)
        dn = A("dn", (rows, Dd), T)
        mN, rN = A("mN", (rows,), f32), A("rN", (rows,), f32)
        self._layernorm(ops, xs[-1], "decoder_norm", dn, rows, Dd, mN, rN)
        # density head on NHWC maps; tokens [B, N, Dd] already are [B, grid, grid, Dd]
        g = self.grid
        hs = [g, 2 * g, 4 * g, 8 * g]
        cin = [Dd, 256, 256, 256]
        hin = [dn]
        hc, hstats = [], []
        # GroupNorm workspace: the C side places the per-image sums behind the per-split partials (their count follows the map size and
        # COUNTR_GN_SPLIT_CAP), so size it from the library's own offset, the largest over the four head maps
        gn_ws = self._shared("gn", max(int(L.countr_groupnorm_bwd_image_sums_offset(B, h * h)) for h in hs) + B * 3 * 256)
        o1 = A("o1", (B, hs[3] * hs[3]), f32)
        out = A("out", (B, 2 * hs[3], 2 * hs[3]), f32)
        # (GroupNorm-apply + ReLU + bilinear x2 as ONE kernel was built in round 4: 75 us for the three stages against 78.5 for the two-
        # kernel chain -- the nine transformed loads per coarse pixel cost what the saved round trip of the activated map gains -- removed)
        hact_tmp = self._shared("hact_tmp", B * hs[2] * hs[2] * 256, T)
        for i in range(4):
            hn = "decode_head%d" % i
            ci = A("hc%d" % i, (B, hs[i], hs[i], 256), T)
            si = A("hstats%d" % i, (B, 8, 2), f32)
            # GroupNorm statistics: from the row partials the convolution's epilogue leaves where the launch runs on the lean kernels
            # (64 bytes per pixel instead of a pass over the 512-byte pixels: 30 -> 9 us on the 192 x 192 map), else from the map
            gnr = self._conv_fwd(ops, hin[i], self.Wf[hn + ".0.weight"], self._pp(hn + ".0.bias"), ci, B, hs[i], hs[i], cin[i], 256,
                                 gn_rows=lambda n=B * hs[i] * hs[i]: self._shared("gn_rows", n * 16))
            gn_fwd = (L.countr_groupnorm_relu_fwd, (ci.data_ptr(),)) if gnr is None else (L.countr_groupnorm_relu_fwd_rows, (ci.data_ptr(), gnr.data_ptr()))
            if i < 3:
                self._op(ops, gn_fwd[0], *gn_fwd[1], self._pp(hn + ".1.weight"), self._pp(hn + ".1.bias"),
                         hact_tmp.data_ptr(), None, None, None, si.data_ptr(), gn_ws.data_ptr(), B, hs[i] * hs[i], 256, 8, 1e-5, code)
                up = A("hu%d" % i, (B, hs[i + 1], hs[i + 1], 256), T)
                self._op(ops, L.countr_upsample2x_fwd, hact_tmp.data_ptr(), up.data_ptr(), B, hs[i], hs[i], 256, code)
                hin.append(up)
            else:
                self._op(ops, gn_fwd[0], *gn_fwd[1], self._pp(hn + ".1.weight"), self._pp(hn + ".1.bias"), None,
                         self._pp(hn + ".3.weight"), self._pp(hn + ".3.bias"), o1.data_ptr(), si.data_ptr(), gn_ws.data_ptr(), B,
                         hs[i] * hs[i], 256, 8, 1e-5, code)
                self._op(ops, L.countr_upsample2x_fwd, o1.data_ptr(), out.data_ptr(), B, hs[3], hs[3], 1, F32)
            hc.append(ci)
            hstats.append(si)
        ex = getattr(p, "ex_range", None)
        if ex is None:
            p.fwd_par = p.fwd
        else:     # [fork | lane 1: exemplar CNN | lane 0: encoder + decoder_embed | join | decoder blocks + head]
            mark = lambda *a: (None, a, None)
            p.fwd_par = ([mark("xfork"), mark("xlane", 1)] + p.fwd[ex[0]:ex[1]] + [mark("xlane", 0)] + p.fwd[:ex[0]] + [mark("xjoin")]
                         + p.fwd[ex[1]:])
        if ex is None:
            self._auto_warm(p.fwd)
        else:     # per lane, in the order fwd_par executes: the encoder's last launch warms the first decoder-block panel (not the exemplar
            self._auto_warm(p.fwd[:ex[0]] + p.fwd[ex[1]:])      # CNN's, which ran beside it), the exemplar lane keeps its hints to itself
            self._auto_warm(p.fwd[ex[0]:ex[1]])
        if not train:
            return p

        p.acc = Plan()
        for acc, lists in ((0, p), (1, p.acc)):
            self._acc = acc
            # =========================== backward (decoder side only) ===========================
            ops = lists.bwd_head
            dout = A("dout", (B, 2 * hs[3], 2 * hs[3]), f32)
            d1 = A("d1", (B, hs[3] * hs[3]), f32)
            self._op(ops, L.countr_upsample2x_bwd, dout.data_ptr(), d1.data_ptr(), B, hs[3], hs[3], 1, F32)
            # gradient scratch for maps: dpre (grad of conv output), dup (grad of conv input)
            dpre = self._shared("dpre", B * hs[3] * hs[3] * 256, T)
            dup = self._shared("dup", B * hs[3] * hs[3] * 256, T)
            dact = self._shared("dact", B * hs[2] * hs[2] * 256, T)
            ddn = A("ddn", (rows, Dd), T)
            for i in (3, 2, 1, 0):
                hn = "decode_head%d" % i
                HW = hs[i] * hs[i]
                # GroupNorm parameter gradients: with deferred reductions the per-block partials {dbeta, dgamma, dw1}[256] stay in this
                # layer's own workspace and are summed by the table launch that finishes the layer's conv wgrad anyway (they used to
                # be 2-3 colsum launches of 16 workgroups each: ~80 us per step of latency-bound finishers)
                defer = self.defer_reduce
                gws = self._shared("gnbw%d" % i, B * 64 * 3 * 256 + 64 + 16 * B + B * 3 * 256) if defer else gn_ws
                if defer:
                    self._claim(gws.data_ptr())
                gpar = lambda n: None if defer else self._gp(n)
                if i == 3:
                    self._op(ops, L.countr_groupnorm_relu_bwd, hc[i].data_ptr(), None, d1.data_ptr(), self._pp(hn + ".3.weight"),
                             hstats[i].data_ptr(), self._pp(hn + ".1.weight"), self._pp(hn + ".1.bias"), dpre.data_ptr(),
                             gpar(hn + ".1.weight"), gpar(hn + ".1.bias"), gpar(hn + ".3.weight"), self._gp(hn + ".3.bias"),
                             gws.data_ptr(), B, HW, 256, 8, code, self._acc)
                else:
                    self._op(ops, L.countr_upsample2x_bwd, dup.data_ptr(), dact.data_ptr(), B, hs[i], hs[i], 256, code)
                    self._op(ops, L.countr_groupnorm_relu_bwd, hc[i].data_ptr(), dact.data_ptr(), None, None, hstats[i].data_ptr(),
                             self._pp(hn + ".1.weight"), self._pp(hn + ".1.bias"), dpre.data_ptr(), gpar(hn + ".1.weight"),
                             gpar(hn + ".1.bias"), None, None, gws.data_ptr(), B, HW, 256, 8, code, self._acc)
                if defer:
                    # the backward's finalize pass leaves per-IMAGE sums behind the split partials: B rows to add, not B * nsplit
                    img = gws.data_ptr() + 4 * L.countr_groupnorm_bwd_image_sums_offset(B, HW)
                    planes = [(0, hn + ".1.bias"), (1, hn + ".1.weight")] + ([(2, hn + ".3.weight")] if i == 3 else [])
                    for plane, pname in planes:
                        self._reduce_later(ops, gws.data_ptr(), img + plane * 256 * 4, self._gp(pname), B, 3 * 256, 256)
                big = hs[i] >= 96   # each of these kernels fills the GPU on its own: forking only adds contention
                if not big:
                    self._fork(ops)
                    self._lane(ops, 1)
                self._conv_wgrad(ops, dpre, hin[i], hn + ".0.weight", B, hs[i], hs[i], cin[i], 256, bias_name=hn + ".0.bias")
                # dgrad == forward conv of dpre with the dgrad-form weights (Cin_gemm = 256 output channels)
                if not big:
                    self._lane(ops, 0)
                tgt = dup if i > 0 else ddn
                self._gemm(ops, code, OP_IM2ROW, OP_ROW, A=dpre.data_ptr(), B=self.Wd[hn + ".0.weight"].data_ptr(), C=tgt.data_ptr(),
                           ldb=9 * 256, ldc=cin[i], M=B * HW, N=cin[i], K=9 * 256, H=hs[i], W=hs[i], Cin=256,
                           out_bf16=int(code == BF16))
                if not big:
                    self._join(ops)
            gx = A("gx", (rows, Dd), f32)
            gxT = A("gxT", (rows, Dd), T) if code == BF16 else None
            # grouped weight gradients (bf16): the six Linear weight gradients of a block (and decoder_embed's with the last block's) run
            # as ONE launch behind the block's last LayerNorm backward -- so the operand view of the residual gradient cycles through
            # four buffers (the three versions the deferred launch still reads stay intact while the next one is written)
            grouped = code == BF16 and self.group_wgrads
            gxT_alt = [gxT] + [A("gxT%d" % k, (rows, Dd), T) for k in (2, 3, 4)] if grouped else [gxT] * 4
            gsel = [0]

            def next_gxT():
                gsel[0] = (gsel[0] + 1) % 4
                return gxT_alt[gsel[0]]
            g_t = self._layernorm_bwd(ops, ddn, xs[-1], "decoder_norm", mN, rN, gx, rows, Dd, accumulate=False, dx_t=gxT)

            ops = lists.bwd_rest
            dh = A("dh", (rows, 4 * Dd), T)
            dn_t = A("dn_t", (rows, Dd), T)      # grad wrt a LayerNorm output
            dproj_in = A("dproj_in", (rows, Dd), T)
            dqkv = A("dqkv", (rows, 3 * Dd), T)
            dq = A("dq", (rows, Dd), T)
            # one dK / dV per block: their input gradients (-> dy_tok, read only by the exemplar-token backward) are emitted at
            # the head of bwd_tok, off the decoder's own dependency chain
            dk_b = [A("dk%d" % i, (B * Sy, Dd), f32) for i in range(self.ddepth)]
            dv_b = [A("dv%d" % i, (B * Sy, Dd), f32) for i in range(self.ddepth)]
            dkT_b = [A("dkT%d" % i, (tok_rows, Dd), T) if code == BF16 else None for i in range(self.ddepth)]     # (rows padded with zeros, as ytok)
            dvT_b = [A("dvT%d" % i, (tok_rows, Dd), T) if code == BF16 else None for i in range(self.ddepth)]
            if not self._sizing and code == BF16:
                for t_ in dkT_b + dvT_b:
                    t_.zero_()
            dy_tok = A("dy_tok", (B * Sy, Dd), f32)
            xws = self._shared("xattn", L.countr_xattn_bwd_workspace_floats(B, N, Sy, Dd))
            tok_dgrads = []
            for i in reversed(range(self.ddepth)):
                b = "decoder_blocks.%d" % i
                d = blk[i]
                # ---- mlp: x3 = x2 + fc2(gelu(fc1(LN2(x2))))  (g_t = bf16/fp32 operand view of gx, emitted by the LN backward)
                grp = [] if grouped else None
                self._linear_bwd(ops, g_t, d["hact"], b + ".mlp.fc2.weight", rows, Dd, 4 * Dd, dx=dh, gelu_pre=d["hpre"], group=grp)
                self._linear_bwd(ops, dh, d["n2"], b + ".mlp.fc1.weight", rows, 4 * Dd, Dd, dx=dn_t, group=grp)
                g_t = self._layernorm_bwd(ops, dn_t, d["x2"], b + ".norm2", d["m2"], d["r2"], gx, rows, Dd, accumulate=True, dx_t=next_gxT())
                # ---- cross attention: x2 = x1 + proj(xattn(wq(LN1(x1)), wk(y), wv(y)))
                self._linear_bwd(ops, g_t, d["xo"], b + ".attn.proj.weight", rows, Dd, Dd, dx=dproj_in, group=grp)
                dk, dv, dkT, dvT = dk_b[i], dv_b[i], dkT_b[i], dvT_b[i]
                self._op(ops, L.countr_xattn_bwd, d["q"].data_ptr(), d["k"].data_ptr(), d["v"].data_ptr(), dproj_in.data_ptr(), dq.data_ptr(),
                         dk.data_ptr(), dv.data_ptr(), xws.data_ptr(), B, N, Sy, Dd, Hd, d["ldkv"], (Dd // Hd) ** -0.5, code,
                         dkT.data_ptr() if dkT is not None else None, dvT.data_ptr() if dvT is not None else None)
                if i == 0:
                    ops.append((None, ("tokready",), None))   # every block's dK / dV is final: the exemplar-token backward may start (run_backward_rest_and_tok)
                self._linear_bwd(ops, dq, d["n1"], b + ".attn.wq.weight", rows, Dd, Dd, dx=dn_t, group=grp)
                g_t = self._layernorm_bwd(ops, dn_t, d["x1"], b + ".norm1", d["m1"], d["r1"], gx, rows, Dd, accumulate=True, dx_t=next_gxT())
                dk_t, dv_t = (dk, dv) if dkT is None else (dkT, dvT)    # bf16 copies come out of the cross-attention backward
                for nm, g_kv in (("wk", dk_t), ("wv", dv_t)):
                    if grouped:     # over the zero-padded rows: whole k-tiles, same sums
                        grp.append((g_kv, ytok, b + ".attn.%s.weight" % nm, tok_rows, Dd, Dd, b + ".attn.%s.bias" % nm))
                    else:
                        self._linear_wgrad(ops, g_kv, ytok, b + ".attn.%s.weight" % nm, B * Sy, Dd, Dd, bias_name=b + ".attn.%s.bias" % nm)
                    tok_dgrads.append((g_kv, b + ".attn.%s.weight" % nm))
                # ---- self attention: x1 = xin + proj(attn(qkv(LN0(xin))))
                self._linear_bwd(ops, g_t, d["att"], b + ".selfattn.proj.weight", rows, Dd, Dd, dx=dproj_in, group=grp)
                if d["lse"] is not None:
                    dlt = self._shared("attn_delta", B * Hd * N)
                    self._op(ops, L.countr_attn_bwd, d["qkv"].data_ptr(), d["att"].data_ptr(), dproj_in.data_ptr(), d["lse"].data_ptr(),
                             dlt.data_ptr(), dqkv.data_ptr(), B, N, Hd, Dd // Hd, (Dd // Hd) ** -0.5)
                else:
                    self._attention_bwd(ops, d["qkv"], d["probs"], dproj_in, dqkv, B, Hd, Dd)
                self._linear_bwd(ops, dqkv, d["n0"], b + ".selfattn.qkv.weight", rows, 3 * Dd, Dd, dx=dn_t, group=grp)
                g_t = self._layernorm_bwd(ops, dn_t, d["xin"], b + ".norm0", d["m0"], d["r0"], gx, rows, Dd, accumulate=True, dx_t=next_gxT())
                if i == 0:   # ---- decoder_embed (no dgrad: the encoder is frozen)
                    self._linear_bwd(ops, g_t, latent, "decoder_embed.weight", rows, Dd, D, group=grp)
                if grouped:      # fc2, fc1, attn.proj, attn.wq, selfattn.proj, selfattn.qkv (, decoder_embed)
                    self._linear_wgrad_group(ops, grp)
            # ---- exemplar tokens: dy_tok = sum over blocks of dK Wk + dV Wv
            ops = lists.bwd_tok
            for j, (g_kv, wn) in enumerate(tok_dgrads):
                self._linear_dgrad(ops, g_kv, wn, dy_tok, B * Sy, Dd, Dd, resid=(None if j == 0 else dy_tok), out_bf16=False)
            if S == 0:
                ws = self._shared("colsum", 256 * 4096)
                self._op(ops, L.countr_colsum, dy_tok.data_ptr(), self._gp("shot_token"), ws.data_ptr(), B, Dd, F32, self._acc)
            else:
                BS = B * S
                dyt = A("dyt", (BS, Dd), T) if code == BF16 else None
                g_y = self._cast(ops, dy_tok, dyt, BS * Dd)
                dc = [A("dc%d" % (i + 1), (BS, sizes[i], sizes[i], chans[i]), T) for i in range(4)]
                dpl = [A("dp%d" % (i + 1), (BS, sizes[i] // 2, sizes[i] // 2, chans[i]), T) for i in range(3)]
                for i in (3, 2, 1, 0):
                    self._op(ops, L.countr_instnorm_relu_pool_bwd, p.in_maps[i].data_ptr(), (g_y if i == 3 else dpl[i]).data_ptr(), stats[i].data_ptr(),
                             dc[i].data_ptr(), BS, sizes[i], sizes[i], chans[i], int(i == 3), code, in_ws.data_ptr(), int(p.in_xhat))
                    wn = "decoder_proj%d.0.weight" % (i + 1)
                    if i == 0:
                        ws = self._shared("c3wgrad", L.countr_conv3x3_c3_wgrad_nblocks() * 64 * 28)
                        self._op(ops, L.countr_conv3x3_c3_wgrad, boxes.data_ptr(), dc[0].data_ptr(), self._gp(wn), self._gp(wn[:-6] + "bias"),
                                 ws.data_ptr(), BS, 64, 64, code, self._acc)
                    else:
                        self._conv_wgrad(ops, dc[i], pl[i - 1], wn, BS, sizes[i], sizes[i], chans[i - 1], chans[i], bias_name=wn[:-6] + "bias")
                        self._conv_fwd(ops, dc[i], self.Wd[wn], None, dpl[i - 1], BS, sizes[i], sizes[i], chans[i], chans[i - 1])
            self._flush_reductions(p)
            for ops_ in (lists.bwd_head, lists.bwd_rest, lists.bwd_tok):
                self._auto_warm(ops_)
        self._acc = 0
        return p

    # ------------------------------------------------------------------ execution API
    def _load_inputs(self, p, imgs, boxes, S):
        p.buf["img"].copy_(imgs, non_blocking=True)
        self._load_boxes(p, boxes, S)

    def _load_boxes(self, p, boxes, S):
        if S > 0:
            B = boxes.shape[0]
            p.buf["boxes"].view(B, S, 3, 64, 64).copy_(boxes[:, :S], non_blocking=True)

    def forward(self, imgs, boxes, shot_num, train=False):
        """SupervisedMAE.forward (models_mae_cross.py:201-207): returns the engine's output buffer [B, H, W]
        (overwritten by the next call on the same plan)."""
        B = imgs.shape[0]
        self.check_ln_fold(imgs)
        p = self.plan(B, int(shot_num), train)
        self._load_inputs(p, imgs, boxes, int(shot_num))
        self.run(p.fwd_par)
        if train:
            p.fwd_gen = getattr(p, "fwd_gen", 0) + 1   # the activations a backward of this plan will read belong to THIS forward
        return p.buf["out"]

    def forward_loaded(self, B, shot_num):
        """The inference forward of plan (B, shot_num) on inputs that are already IN the plan's buffers (p.buf["img"], p.buf["boxes"]:
        countr_amd.inference writes the sliding windows there directly, after its own check_ln_fold call).  Returns the output buffer
        [B, H, W]."""
        p = self.plan(B, int(shot_num), False)
        self.run(p.fwd_par)
        return p.buf["out"]

    def decoder_ops_with_exemplar_lane(self, p):
        """The decoder-side forward launches of plan p (p.fwd behind the encoder) for the pipelined forms, where no encoder runs on the
        main lane in front of them: the exemplar CNN and the blocks' wk / wv projections (a chain of ~15 launches of a few workgroups
        each, ~150 us at 24 exemplars) go to side lane 1 and the main lane runs what does not read the exemplar tokens -- decoder_embed and
        the first block's self-attention half, up to the query projection -- beside them; joined in front of the first cross-attention.
        In the plain forward the chain hides beside the encoder (p.fwd_par); inline it was the head of the decoder's critical path."""
        dec = p.fwd[p.enc_ops:]
        ex, xi = getattr(p, "ex_range", None), getattr(p, "first_xattn", None)
        if (ex is None or xi is None or not self.overlap_exemplar or self.code != BF16 or os.environ.get("COUNTR_PIPE_EXEMPLAR_LANE", "1") == "0"
                or not (p.enc_ops <= ex[0] <= ex[1] <= xi)):
            return dec
        cached = getattr(p, "_dec_lane", None)
        if cached is None:
            mark = lambda *a: (None, a, None)
            cached = p._dec_lane = ([mark("xfork"), mark("xlane", 1)] + p.fwd[ex[0]:ex[1]] + [mark("xlane", 0)] + p.fwd[p.enc_ops:ex[0]]
                                    + p.fwd[ex[1]:xi] + [mark("xjoin")] + p.fwd[xi:])
        return cached

    def forward_loaded_pipelined(self, B, shot_num, have, ahead):
        """forward_loaded with the frozen encoder pipelined across calls (inference has no trainable side at all: every forward's encoder
        is independent of every other forward).  `ahead`: the NEXT batch's windows are already in the plan's p.pipe_img -- their encoder
        forward runs on a lane of its own beside this batch's decoder / density head and leaves its latent in p.pipe_latent.  `have`:
        this batch's latent is waiting there (the previous call ran with ahead=True for it) -- it is copied into place and the encoder
        is not run again.  Same launches on the same data as forward_loaded: bit-identical maps (tests/test_inference_gpu.py)."""
        p = self.plan(B, int(shot_num), False)
        if p.enc_pipe is None:
            raise _lib.CountrError("the pipelined forward needs a 16-bit precision and the fused attention kernel")
        if have:
            n = p.buf["latent"].numel()
            p.buf["latent"].view(-1).copy_(p.pipe_latent[:n], non_blocking=True)
        key = (bool(have), bool(ahead))
        cache = p.__dict__.setdefault("_pipe_fwd", {})
        if key not in cache:
            mark = lambda *a: (None, a, None)
            lane = ([mark("pfork")] + p.enc_pipe + [mark("pmain")]) if ahead else []
            cache[key] = ([] if have else p.fwd[:p.enc_ops]) + lane + self.decoder_ops_with_exemplar_lane(p)
        self.run(cache[key])
        if ahead:
            self.pipe_join()
        return p.buf["out"]

    def backward(self, B, shot_num, dout):
        """Decoder-side backward for the last train-mode forward of plan (B, shot_num); fills self.G."""
        p = self.plan(B, int(shot_num), True)
        p.buf["dout"].copy_(dout, non_blocking=True)
        self.run(p.bwd_head)
        self.run(p.bwd_rest)
        self.run(p.bwd_tok)

    def adam_ranges(self, S, weight_decay, skip=None):
        return self.layout.adam_ranges(S, weight_decay, skip)

    def adam_plan(self, weight_decay, skip=(), zero=()):
        return self.layout.adam_plan(weight_decay, skip, zero)

    def adamw_launch(self, S, weight_decay=0.05, betas=(0.9, 0.95), eps=1e-8, lr=0.0, step=0, grad_scale=1.0, hyper_dev=None, skip=None,
                     zero=(), gnorm=False, stream=None, amp=None):
        """Enqueue the fused AdamW (+ shadow refresh).  hyper_dev: device fp32[8] {lr, bc1[0], bc2[0], grad_scale, bc1[1], bc2[1],
        bc1[2], bc2[2]} read by the kernel at run time, so a captured launch can be replayed with new values.  skip=None: the
        shot_num rule of adam_ranges (parameters without a gradient for S are skipped)."""
        if self.M is None:
            self.M = torch.zeros_like(self.G)
            self.V = torch.zeros_like(self.G)
        if skip is None:
            rng = [(s, e, wd, 0, 0) for s, e, wd in self.adam_ranges(S, weight_decay, None)]
        else:
            rng = self.adam_plan(weight_decay, tuple(skip), tuple(zero))
        n = len(rng)
        starts = (C.c_int64 * n)(*[r[0] for r in rng])
        ends = (C.c_int64 * n)(*[r[1] for r in rng])
        wds = (C.c_float * n)(*[r[2] for r in rng])
        groups = (C.c_int * n)(*[r[3] for r in rng])
        zeros = (C.c_int * n)(*[r[4] for r in rng])
        if gnorm and self.gnorm is None:
            self.gnorm = torch.zeros(self.L.countr_adamw_gnorm_floats(), device=self.device, dtype=torch.float32)
        lay = self.layout
        shadow = (self.Wt.data_ptr() + 2 * lay.train_start) if self.half else None
        # amp (fp16 mode): device fp32[8] of the dynamic loss scale -- non-finite gradients skip the update (countr_adamw_step_amp)
        _lib.check(self.L.countr_adamw_step_amp(self.P.data_ptr() + 4 * lay.train_start, self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                                shadow, n, starts, ends, wds, groups, zeros, lr, betas[0], betas[1], eps, step, grad_scale,
                                                hyper_dev.data_ptr() if hyper_dev is not None else None,
                                                self.gnorm.data_ptr() if gnorm else None, amp.data_ptr() if amp is not None else None,
                                                stream if stream is not None else self._stream()), "adamw")
        self._refresh_conv_shadows(stream)

    def adamw_step(self, S, lr, weight_decay=0.05, betas=(0.9, 0.95), eps=1e-8, grad_scale=1.0):
        """torch.optim.AdamW(betas=(0.9,0.95)) semantics (FSC_finetune_cross.py:235) fused over the flat buffers."""
        self.step_count += 1
        self.adamw_launch(S, weight_decay, betas, eps, lr=lr, step=self.step_count, grad_scale=grad_scale)

    def _refresh_conv_shadows(self, stream=None):
        """OHWI + dgrad-form shadows of every conv weight in one launch (fp32 master -> both permuted forms), and the W^T shadows of the
        Linear weights in another (16-bit modes: a transpose of the 16-bit shadow W that AdamW / sync_weights has just written -- the same
        bits as a cast of the master, half the bytes read).  The tables of pointers are built once."""
        if not self.conv_names and not self.WtT:
            return
        if getattr(self, "_shadow_tab", None) is None:
            lin = list(self.WtT)                      # Linear weights [N][K]: only the transposed form
            from16 = self.half and os.environ.get("COUNTR_TRANSPOSE16", "1") != "0" and all(self.layout.shapes[c][0] % 64 == 0 and self.layout.shapes[c][1] % 64 == 0 for c in lin)
            self._lin_tab = None
            if lin and from16:
                m = len(lin)
                shp = [self.layout.shapes[c] for c in lin]
                self._lin_tab = (m, (C.c_void_p * m)(*[self.Wt.data_ptr() + 2 * self.layout.off[c] for c in lin]),
                                 (C.c_void_p * m)(*[self.WtT[c].data_ptr() for c in lin]),
                                 (C.c_int * m)(*[s_[0] for s_ in shp]), (C.c_int * m)(*[s_[1] for s_ in shp]))
                lin = []
            names = self.conv_names + lin             # (fp32 mode keeps no W^T shadows; a 16-bit shape outside the tiles goes the fp32 way)
            n = len(names)
            shp = [self.layout.shapes[c] for c in names]
            self._shadow_tab = (n, (C.c_void_p * n)(*[self._pp(c) for c in names]),
                                (C.c_void_p * n)(*([self.Wf[c].data_ptr() for c in self.conv_names] + [None] * len(lin))),
                                (C.c_void_p * n)(*([self.Wd[c].data_ptr() for c in self.conv_names] + [self.WtT[c].data_ptr() for c in lin])),
                                (C.c_int * n)(*[s_[0] for s_ in shp]), (C.c_int * n)(*[s_[1] for s_ in shp]),
                                (C.c_int * n)(*[(s_[2] * s_[3] if len(s_) == 4 else 1) for s_ in shp]))
        st = stream if stream is not None else self._stream()
        n, src, wf, wd, co, ci, taps = self._shadow_tab
        vp, ip = C.c_void_p, C.c_int
        for i0 in range(0, n, 32):           # the launch takes 32 entries
            m = min(32, n - i0)
            off = lambda arr, ty: C.cast(C.byref(arr, i0 * C.sizeof(ty)), C.POINTER(ty))
            _lib.check(self.L.countr_conv_shadows(m, off(src, vp), off(wf, vp), off(wd, vp), off(co, ip), off(ci, ip), off(taps, ip),
                                                  self.code, st), "conv_shadows")
        if self._lin_tab is not None:
            m, s16, d16, rows, cols = self._lin_tab
            for i0 in range(0, m, 96):       # the launch takes 96 matrices
                k = min(96, m - i0)
                off = lambda arr, ty: C.cast(C.byref(arr, i0 * C.sizeof(ty)), C.POINTER(ty))
                _lib.check(self.L.countr_transpose16(k, off(s16, vp), off(d16, vp), off(rows, ip), off(cols, ip), st), "transpose16")
