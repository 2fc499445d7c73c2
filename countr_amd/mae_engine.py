"""Static-plan engine for the MAE pretraining model (reference models_mae_noct.py:11-204) on MI355X.

Same machinery as countr_amd.engine.Engine (flat fp32 parameters / gradients, bf16 shadows, launch lists over
pre-allocated buffers, hipGraph-replayable), different graph: the whole model trains.

  forward_encoder  models_mae_noct.py:137-157   patch-embed only the KEPT patches (a row gather of the patch matrix before
                                                the GEMM: per-row linear map, so identical to embed-then-gather :110-135)
  forward_decoder  models_mae_noct.py:159-179   unshuffle = one row gather with mask_token as default row (+ pos embed)
  forward_loss     models_mae_noct.py:181-198   countr_patch_mse (all-patch MSE, optional norm_pix)
  backward         autograd of the above        encoder + decoder ViT blocks (flash attention backward in bf16 mode)
Gradient buckets: 0 = decoder side (final first), 1..6 = encoder block groups from the top (mae_bucket_fn, mae_enc_parts).
"""
import torch

from . import _lib
from ._lib import F32, BF16, ACT_GELU
from .engine import Engine, ParamLayout, Plan


def mae_trainable(name):
    """requires_grad=False only for the fixed sin-cos embeddings (models_mae_noct.py:25,39)."""
    return name not in ("pos_embed", "decoder_pos_embed")


def mae_enc_parts(depth):
    """Encoder gradient buckets.  Every all-reduce overlaps the backward phase behind it, so only the LAST bucket is exposed: six parts
    (two ViT-B blocks = 57 MB of fp32 gradient each, + the 2.4-MB patch embedding in the last) instead of round 2's thirds (115 MB)."""
    return max(1, min(depth, 6))


def mae_bucket_fn(depth, parts=None):
    """Gradient buckets in backward-completion order: 0 = decoder side + mask_token, then the encoder in `parts` groups of blocks from
    the top: 1 = norm + the uppermost blocks, ..., parts = the lowest blocks + patch_embed (each all-reduce overlaps the next phase)."""
    parts = mae_enc_parts(depth) if parts is None else parts

    def bucket(name):
        if name.startswith(("decoder_", "mask_token")):
            return 0
        if name.startswith("blocks."):
            i = int(name.split(".")[1])
            return 1 + (depth - 1 - i) * parts // depth
        return 1 if name.startswith("norm.") else parts   # patch_embed

    return bucket


class MaePlan(Plan):
    def __init__(self, parts=3):
        super().__init__()
        self.bwd_dec = self.bwd_head
        self.bwd_enc = [[] for _ in range(parts)]   # encoder backward, one launch list per gradient bucket 1 .. parts


class MaeEngine(Engine):
    FROZEN_ENCODER = False   # the whole model trains: no pre-scaled q packing (the backward reads the same qkv)

    def _make_layout(self, named_shapes):
        return ParamLayout(named_shapes, trainable=mae_trainable, bucket=mae_bucket_fn(self.depth))

    def _conv_names(self):
        return []

    def adam_ranges(self, S, weight_decay, skip=None):
        return [(s, e, 0.0 if nodecay else weight_decay) for (_b, nodecay), s, e in self.layout.segments]

    def adam_plan(self, weight_decay, skip=(), zero=()):   # every parameter has a gradient in every step: one counter group
        return [(s, e, wd, 0, 0) for s, e, wd in self.adam_ranges(0, weight_decay)]

    # ------------------------------------------------------------------ timm Block (x += attn(norm1 x); x += mlp(norm2 x))
    def _block_fwd(self, ops, p, b, xin, B, N, Dm, heads, train):
        T, f32 = self.tdt, torch.float32
        rows = B * N
        A = lambda k, shape, dt: self._alloc(p, b + k, shape, dt)
        d = {"xin": xin}
        d["n1"], d["m1"], d["r1"] = A(".n1", (rows, Dm), T), A(".m1", (rows,), f32), A(".r1", (rows,), f32)
        d["qkv"] = A(".qkv", (rows, 3 * Dm), T)
        fused = self._fused_attention(Dm // heads)
        d["probs"] = A(".probs", (B * heads * N * N,), T) if (train and not fused) else None
        d["lse"] = A(".lse", (B * heads * N,), f32) if (train and fused) else None
        d["att"] = A(".att", (rows, Dm), T)
        self._layernorm(ops, xin, b + ".norm1", d["n1"], rows, Dm, d["m1"], d["r1"])
        self._linear(ops, d["n1"], b + ".attn.qkv.weight", d["qkv"], rows, 3 * Dm, Dm)
        self._attention_fwd(ops, p, d["qkv"], d["att"], B, heads, Dm, probs=d["probs"], lse=d["lse"], N=N)
        d["x1"] = A(".x1", (rows, Dm), f32)
        self._linear(ops, d["att"], b + ".attn.proj.weight", d["x1"], rows, Dm, Dm, resid=xin)
        d["n2"], d["m2"], d["r2"] = A(".n2", (rows, Dm), T), A(".m2", (rows,), f32), A(".r2", (rows,), f32)
        d["hpre"], d["hact"] = A(".hpre", (rows, 4 * Dm), T), A(".hact", (rows, 4 * Dm), T)
        self._layernorm(ops, d["x1"], b + ".norm2", d["n2"], rows, Dm, d["m2"], d["r2"])
        self._linear(ops, d["n2"], b + ".mlp.fc1.weight", d["hact"], rows, 4 * Dm, Dm, act=ACT_GELU, pre=d["hpre"])
        d["x2"] = A(".x2", (rows, Dm), f32)
        self._linear(ops, d["hact"], b + ".mlp.fc2.weight", d["x2"], rows, Dm, 4 * Dm, resid=d["x1"])
        return d

    def _block_bwd(self, ops, b, d, s, B, N, Dm, heads, g_t):
        """s: scratch dict {gx (fp32 grad of the residual stream, in/out), gxT, dh, dn_t, dproj_in, dqkv}; g_t = the GEMM-operand
        view of gx on entry (emitted by the previous LayerNorm backward).  Returns the operand view of the updated gx."""
        L, code = self.L, self.code
        rows = B * N
        gx = s["gx"]
        # bf16: the four weight gradients of the block run as ONE launch (Engine._linear_wgrad_group) just before the last LayerNorm
        # backward; until then the operand view of the residual gradient they read (g_t on entry, in gxT) must survive the first
        # LayerNorm backward, which therefore writes its view into gxT2
        grp = [] if s.get("gxT2") is not None else None
        self._linear_bwd(ops, g_t, d["hact"], b + ".mlp.fc2.weight", rows, Dm, 4 * Dm, dx=s["dh"], gelu_pre=d["hpre"], group=grp)
        self._linear_bwd(ops, s["dh"], d["n2"], b + ".mlp.fc1.weight", rows, 4 * Dm, Dm, dx=s["dn_t"], group=grp)
        g_t = self._layernorm_bwd(ops, s["dn_t"], d["x1"], b + ".norm2", d["m2"], d["r2"], gx, rows, Dm, accumulate=True,
                                  dx_t=s["gxT2"] if grp is not None else s["gxT"])
        self._linear_bwd(ops, g_t, d["att"], b + ".attn.proj.weight", rows, Dm, Dm, dx=s["dproj_in"], group=grp)
        if d["lse"] is not None:
            dlt = self._shared("attn_delta", B * heads * N)
            self._op(ops, L.countr_attn_bwd, d["qkv"].data_ptr(), d["att"].data_ptr(), s["dproj_in"].data_ptr(), d["lse"].data_ptr(),
                     dlt.data_ptr(), s["dqkv"].data_ptr(), B, N, heads, Dm // heads, (Dm // heads) ** -0.5)
        else:
            self._attention_bwd(ops, d["qkv"], d["probs"], s["dproj_in"], s["dqkv"], B, heads, Dm, N=N)
        self._linear_bwd(ops, s["dqkv"], d["n1"], b + ".attn.qkv.weight", rows, 3 * Dm, Dm, dx=s["dn_t"], group=grp)
        if grp is not None:
            self._linear_wgrad_group(ops, grp)
        return self._layernorm_bwd(ops, s["dn_t"], d["xin"], b + ".norm1", d["m1"], d["r1"], gx, rows, Dm, accumulate=True, dx_t=s["gxT"])

    def _bwd_scratch(self, p, tag, rows, Dm):
        T, f32 = self.tdt, torch.float32
        A = lambda k, shape, dt: self._alloc(p, tag + k, shape, dt)
        return {"gx": A(".gx", (rows, Dm), f32), "gxT": A(".gxT", (rows, Dm), T) if self.code == BF16 else None,
                "gxT2": A(".gxT2", (rows, Dm), T) if (self.code == BF16 and self.group_wgrads) else None,
                "dh": A(".dh", (rows, 4 * Dm), T), "dn_t": A(".dn_t", (rows, Dm), T), "dproj_in": A(".dproj_in", (rows, Dm), T),
                "dqkv": A(".dqkv", (rows, 3 * Dm), T)}

    # ------------------------------------------------------------------ plans
    def plan(self, B, K, train):
        key = (B, K, bool(train))
        if key not in self.plans:
            self._sizing = True
            try:
                for tr in (False, True):
                    self._build(B, K, tr)
            finally:
                self._sizing = False
            self._reserve()
            self.plans[key] = self._build(B, K, bool(train))
        return self.plans[key]

    def _build(self, B, K, train):
        """K = len_keep tokens per image seen by the encoder (models_mae_noct.py:117)."""
        L = self.L
        p = MaePlan(mae_enc_parts(self.depth))
        T, f32, i32 = self.tdt, torch.float32, torch.int32
        N, D, Dd, H, Hd = self.N, self.D, self.Dd, self.H, self.Hd
        code = self.code
        assert 1 <= K <= N
        rk, rn = B * K, B * N
        F = 3 * self.patch * self.patch
        A = lambda k, shape, dt: self._alloc(p, k, shape, dt)
        ops = p.fwd
        # masking indices (filled by set_masking): global source rows
        keep_src, keep_pos = A("keep_src", (rk,), i32), A("keep_pos", (rk,), i32)
        restore_src = A("restore_src", (rn,), i32)
        mask_src = A("mask_src", (max(rn - rk, 1),), i32)
        A("mask", (B, N), f32)
        # ---------------- encoder on the kept patches
        img = A("img", (B, 3, self.img, self.img), f32)
        patches = A("patches", (rn, F), T)
        pk = A("patches_keep", (rk, F), T)
        posk = A("pos_keep", (rk, D), f32)
        x0 = A("x0", (rk, D), f32)
        self._op(ops, L.countr_im2patch, img.data_ptr(), patches.data_ptr(), B, self.img, self.img, self.patch, code)
        self._op(ops, L.countr_gather_rows, patches.data_ptr(), keep_src.data_ptr(), pk.data_ptr(), None, None, 0, rk, F, code, code)
        self._op(ops, L.countr_gather_rows, self._pp("pos_embed"), keep_pos.data_ptr(), posk.data_ptr(), None, None, 0, rk, D, F32, F32)
        self._gemm(ops, code, _lib.OP_ROW, _lib.OP_ROW, A=pk.data_ptr(), B=self._wp("patch_embed.proj.weight"), C=x0.data_ptr(),
                   bias=self._pp("patch_embed.proj.bias"), resid=posk.data_ptr(), lda=F, ldb=F, ldc=D, ldres=D, M=rk, N=D, K=F,
                   res_mod=0, out_bf16=0)
        enc = []
        x = x0
        for i in range(self.depth):
            d = self._block_fwd(ops, p, "blocks.%d" % i, x, B, K, D, H, train)
            enc.append(d)
            x = d["x2"]
        latent = A("latent", (rk, D), T)
        mE, rE = A("mE", (rk,), f32), A("rE", (rk,), f32)
        self._layernorm(ops, x, "norm", latent, rk, D, mE, rE)
        p.enc_ops = len(ops)
        x_enc_out = x
        # ---------------- decoder on all tokens
        e = A("e", (rk, Dd), f32)
        self._linear(ops, latent, "decoder_embed.weight", e, rk, Dd, D)
        xd0 = A("xd0", (rn, Dd), f32)
        self._op(ops, L.countr_gather_rows, e.data_ptr(), restore_src.data_ptr(), xd0.data_ptr(), self._pp("mask_token"),
                 self._pp("decoder_pos_embed"), N, rn, Dd, F32, F32)
        dec = []
        x = xd0
        for i in range(self.ddepth):
            d = self._block_fwd(ops, p, "decoder_blocks.%d" % i, x, B, N, Dd, Hd, train)
            dec.append(d)
            x = d["x2"]
        dn = A("dn", (rn, Dd), T)
        mN, rN = A("mN", (rn,), f32), A("rN", (rn,), f32)
        self._layernorm(ops, x, "decoder_norm", dn, rn, Dd, mN, rN)
        pred = A("pred", (rn, F), f32)
        self._linear(ops, dn, "decoder_pred.weight", pred, rn, F, Dd)
        A("loss", (1,), f32)
        self._shared("mse", L.countr_patch_mse_workspace_floats(B, self.img, self.img, self.patch))
        self._auto_warm(p.fwd)
        if not train:
            return p

        p.acc = MaePlan(mae_enc_parts(self.depth))            # the same lists with parameter gradients accumulated (gradient accumulation, micro-steps 2..)
        for acc, lists in ((0, p), (1, p.acc)):
            self._acc = acc
            # =========================== backward ===========================
            ops = lists.bwd_dec
            dpred = A("dpred", (rn, F), T)
            ddn = A("ddn", (rn, Dd), T)
            sd = self._bwd_scratch(p, "dec", rn, Dd)
            self._linear_bwd(ops, dpred, dn, "decoder_pred.weight", rn, F, Dd, dx=ddn)
            g_t = self._layernorm_bwd(ops, ddn, x, "decoder_norm", mN, rN, sd["gx"], rn, Dd, accumulate=False, dx_t=sd["gxT"])
            for i in reversed(range(self.ddepth)):
                g_t = self._block_bwd(ops, "decoder_blocks.%d" % i, dec[i], sd, B, N, Dd, Hd, g_t)
            # mask_token: sum of the gradient rows at masked positions (models_mae_noct.py:166-167)
            if rn > rk:
                gm = A("gmask", (rn - rk, Dd), f32)
                ws = self._shared("colsum", 256 * 4096)
                self._op(ops, L.countr_gather_rows, sd["gx"].data_ptr(), mask_src.data_ptr(), gm.data_ptr(), None, None, 0, rn - rk, Dd, F32,
                         F32)
                self._op(ops, L.countr_colsum, gm.data_ptr(), self._gp("mask_token"), ws.data_ptr(), rn - rk, Dd, F32, self._acc)
            ge = A("ge", (rk, Dd), T)
            self._op(ops, L.countr_gather_rows, sd["gx"].data_ptr(), keep_src.data_ptr(), ge.data_ptr(), None, None, 0, rk, Dd, F32, code)
            dlat = A("dlat", (rk, D), T)
            self._linear_bwd(ops, ge, latent, "decoder_embed.weight", rk, Dd, D, dx=dlat)

            bucket = mae_bucket_fn(self.depth)
            ops = lists.bwd_enc[0]
            se = self._bwd_scratch(p, "enc", rk, D)
            g_t = self._layernorm_bwd(ops, dlat, x_enc_out, "norm", mE, rE, se["gx"], rk, D, accumulate=False, dx_t=se["gxT"])
            for i in reversed(range(self.depth)):
                ops = lists.bwd_enc[bucket("blocks.%d.norm1.weight" % i) - 1]
                g_t = self._block_bwd(ops, "blocks.%d" % i, enc[i], se, B, K, D, H, g_t)
            ops = lists.bwd_enc[-1]
            self._linear_wgrad(ops, g_t, pk, "patch_embed.proj.weight", rk, D, F, bias_name="patch_embed.proj.bias")
            self._flush_reductions(p)
            for ops_ in [lists.bwd_dec] + list(lists.bwd_enc):
                self._auto_warm(ops_)
        self._acc = 0
        return p

    # ------------------------------------------------------------------ execution API
    def set_masking(self, p, ids_shuffle):
        """ids_shuffle [B, N] (argsort of the per-sample noise, models_mae_noct.py:119-121) -> the plan's index buffers
        and the binary mask (:128-132): one launch (countr_mae_indices; it was ~14 torch int ops on tiny tensors per step)."""
        B, N = ids_shuffle.shape
        K = p.buf["keep_src"].numel() // B
        ids_shuffle = ids_shuffle.to(self.device, torch.int64).contiguous()
        ids_restore = torch.empty_like(ids_shuffle)
        _lib.check(self.L.countr_mae_indices(ids_shuffle.data_ptr(), ids_restore.data_ptr(), p.buf["keep_pos"].data_ptr(),
                                             p.buf["keep_src"].data_ptr(), p.buf["restore_src"].data_ptr(),
                                             p.buf["mask_src"].data_ptr() if K < N else None, p.buf["mask"].data_ptr(), B, N, K,
                                             self._stream()), "mae_indices")
        return ids_restore

    def loss_launch(self, p, B, norm_pix, grad_scale=1.0, with_grad=True, amp=None):
        ws = self._ws["mse"]
        dp = p.buf["dpred"].data_ptr() if with_grad else None
        _lib.check(self.L.countr_patch_mse_amp(p.buf["pred"].data_ptr(), p.buf["img"].data_ptr(), dp, p.buf["loss"].data_ptr(), ws.data_ptr(),
                                               B, self.img, self.img, self.patch, int(bool(norm_pix)), float(grad_scale), self.code,
                                               amp.data_ptr() if amp is not None else None, self._stream()), "patch_mse")

    def forward(self, imgs, ids_shuffle, len_keep, train=False, norm_pix=False):
        """-> (loss [1], pred [B, N, F], mask [B, N]) buffers of the plan (overwritten by the next call)."""
        B = imgs.shape[0]
        p = self.plan(B, int(len_keep), train)
        p.buf["img"].copy_(imgs, non_blocking=True)
        self.set_masking(p, ids_shuffle)
        self.run(p.fwd)
        self.loss_launch(p, B, norm_pix, with_grad=train)
        return p.buf["loss"], p.buf["pred"].view(B, self.N, -1), p.buf["mask"]

    def backward(self, B, len_keep):
        p = self.plan(B, int(len_keep), True)
        self.run(p.bwd_dec)
        for ops in p.bwd_enc:
            self.run(ops)
