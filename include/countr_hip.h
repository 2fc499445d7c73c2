/*
 * countr_hip.h -- C ABI of libcountr_hip.so: the MI355X (gfx950) kernels behind the CounTR
 * SupervisedMAE hot path (reference: /root/reference/models_mae_cross.py:136-207 and
 * models_crossvit.py:46-156).  The reference has no FFI of its own (it is pure PyTorch); each
 * entry point below replaces the torch.nn call sites cited next to it.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.  All pointers are DEVICE pointers to
 *     caller-owned, 16-byte aligned buffers unless noted.  `stream` is a hipStream_t passed as void*.
 *   - every function returns 0 on success, <0 on error; countr_last_error() returns the message
 *     (thread-local).  Nothing throws across the ABI.
 *   - no hipMalloc/hipFree/synchronisation inside any call: every call only enqueues kernels on
 *     `stream`, so whole steps are hipGraph-capturable.  Workspaces are caller-allocated.
 *   - dtype codes: COUNTR_F32 = 0 (fp32 storage, exact-f32 MFMA 16x16x4), COUNTR_BF16 = 1 (bf16
 *     storage, MFMA 16x16x32 bf16 with fp32 accumulate).  Statistics, softmax, loss, gradients of
 *     parameters and optimizer state are always fp32.
 *   - activations are token-major [rows, features] (== torch [B, N, C] flattened) and images /
 *     feature maps are NHWC.
 */
#ifndef COUNTR_HIP_H
#define COUNTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COUNTR_F32 0
#define COUNTR_BF16 1

/* operand addressing modes of the generic MFMA GEMM */
#define COUNTR_OP_ROW 0    /* elem(r,k) = p[r*ld + k]            (K contiguous)                    */
#define COUNTR_OP_COL 1    /* elem(r,k) = p[k*ld + r]            (K strided, rows contiguous)      */
#define COUNTR_OP_IM2ROW 2 /* r = pixel (b,y,x), k = tap*C + c : 3x3 pad-1 gather of NHWC p         */
#define COUNTR_OP_IM2COL 3 /* r = tap*C + c, k = pixel (b,y,x) : same gather, K = pixels (wgrad)    */

#define COUNTR_ACT_NONE 0
#define COUNTR_ACT_GELU 1 /* exact erf GELU, nn.GELU default (models_crossvit.py:49,63) */
#define COUNTR_ACT_GELU_BWD 2 /* out = acc * GELU'(C2): autograd of the line above fused into the fc2 dgrad GEMM; C2 is an
                                 INPUT here (the saved pre-activation, same layout and dtype as C) */

/* -------- library management -------- */
int countr_init(int device);            /* selects device, checks it is gfx950-class, allocates the library's ONE per-device
                                           constant (a vector of zeros that bias-less GEMM / convolution launches read); 0 = ok.
                                           REQUIRED once per device and PER LIBRARY before any launch: libcountr_hip.so (bf16) and
                                           libcountr_hip_f16.so (fp16) are separate images with separate state.  Idempotent and
                                           thread-safe.  A bias-less countr_gemm / countr_conv launch on a device without it fails
                                           with a negative code and a message naming countr_init (it does not allocate lazily:
                                           no allocation inside a launch) */
int countr_version(void);               /* ABI version, currently 9 (9: countr_gemm_args grew at its end (gn_rows) + countr_groupnorm_relu_fwd_rows -- rebuild callers; 8: countr_transpose16 added, no layout change; 7: countr_masked_mse_amp / countr_patch_mse_amp / countr_adamw_step_amp added, no layout change; 6: countr_softmax_fwd_ld added, no layout change; 5: countr_step_prologue added, no layout change; countr_gemm_args grew at its end -- round 3: ln_* fields, rowsum_slabs; round 4: prefetch hint (3) -- so a caller built against an older version must be rebuilt; 4: countr_gemm_group / countr_gemm_group_tiles added, no layout change) */
const char* countr_last_error(void);    /* thread-local message of the last failing call        */

/*
 * Generic tiled MFMA GEMM:  C[m,n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
 *   epilogue(v) = act(v + bias[n]) + resid[(m % res_mod) , n]      (each part optional)
 *   C2 (optional) receives v + bias[n] before the activation (saved for GELU backward).
 * A is the M-side operand, B the N-side operand, each in one of the COUNTR_OP_* modes.
 * Batched when nbatch > 1: z = b0*nb1 + b1 adds b0*s?0 + b1*s?1 (in elements) to A, B, C.
 * Split-K when partial != NULL (nbatch must be 1): block z of max(splitk,1) handles a K range and stores
 * raw fp32 sums to partial[z][M][N]; finish with countr_splitk_reduce.
 * Replaces: torch addmm/bmm behind nn.Linear (models_crossvit.py:62,65,84,92,115-127;
 * models_mae_cross.py:152), q@k^T / attn@v (models_crossvit.py:87,91,121,125), the PatchEmbed conv
 * (timm PatchEmbed, models_mae_cross.py:138) and the 3x3 convs (models_mae_cross.py:47-100) as
 * implicit GEMMs, plus their autograd backward formulas.
 */
typedef struct countr_gemm_args {
  const void* A;
  const void* B;
  void* C;
  void* C2;           /* optional second output (pre-activation), same type/ld as C           */
  const float* bias;  /* optional [N] fp32                                                    */
  const float* resid; /* optional fp32 [*, ldres]; may alias C when C is fp32                 */
  float* partial;     /* split-K workspace, fp32 [splitk][M][N]                               */
  int64_t lda, ldb, ldc, ldres;
  int64_t sA0, sA1, sB0, sB1, sC0, sC1;
  int32_t M, N, K;
  int32_t res_mod;  /* 0 = residual row is m                                                  */
  int32_t act;      /* COUNTR_ACT_*                                                           */
  int32_t out_bf16; /* 1: C/C2 stored as bf16, 0: fp32                                        */
  int32_t nbatch, nb1;
  int32_t splitk;
  int32_t H, W, Cin; /* conv geometry for IM2ROW / IM2COL                                     */
  float alpha;
  float* rowsum_partial; /* optional (bf16 split-K only): fp32 [max(splitk,1)][M] receives sum_k A(m,k) per
                            K split -- the bias gradient of a wgrad GEMM, fused as one extra MFMA per tile    */
  /* LayerNorm folded into the surrounding nn.Linear layers (frozen encoder, bf16, (ROW, ROW), N % 128 == 0, K % 64 == 0 -- K % 128 == 0
   * for a consumer, whose row partials are read 16 bytes at a time; timm Block:
   * x = x + proj(attn(norm1(x))); x = x + fc2(gelu(fc1(norm2(x)))) -- models_mae_cross.py:144-146).
   * Producer (fp32 output with residual): ln_xcopy receives a bf16 copy of C, ln_stats_out[m][N/64][2] the {sum, sum of squares} of
   * every 64-column block of the fp32 output row.  Consumer (bf16 output): A is that copy, B = gamma o W, bias = b + W beta,
   * ln_stats the producer's partials (ln_nblk = K / 64 blocks per row), ln_colsum[n] = sum_k B[n][k]; the epilogue applies
   * rstd_m (acc - mean_m ln_colsum[n]) + bias[n] with mean / rstd over the K features of row m (biased variance, ln_eps). */
  void* ln_xcopy;
  float* ln_stats_out;
  const float* ln_stats;
  const float* ln_colsum;
  int32_t ln_nblk;
  float ln_eps;
  int32_t rowsum_slabs; /* slabs the caller sized rowsum_partial for: 0 = max(splitk,1) (the layout above); otherwise it must be
                           countr_gemm_rowsum_slabs() of this launch (the lean convolution weight gradient deals the bias-gradient
                           work over more waves and writes [rowsum_slabs][M]; the sum over ALL slabs is the bias gradient)        */
  /* Cache warm-up hint (ABI 3): a read-only range -- normally the B operand of the NEXT launch on the stream -- that spare workgroups
   * of this launch read once and discard (16-byte aligned pointer and size).  Honoured by the bf16 (ROW, ROW) kernels: the CUs a grid of
   * fewer than 256 tiles leaves idle (up to 64 workgroups, a multiple of 8), or 32 extra workgroups in front of a bigger grid.  Never changes a result.  Why: the frozen encoder's weight panels
   * (blocks.i.attn / mlp, models_mae_cross.py:32-34,141-145) are cold in every step -- ~5 GB stream through the 256-MB memory-side cache
   * between two uses -- and a single-round GEMM that walks a cold [N][K] panel k-slab by k-slab pays the HBM latency on every k-tile
   * (fc2: 38.7 us in the step, 28.7 us with the panel resident: tools/bench_chain.py). */
  const void* prefetch;
  int64_t prefetch_bytes;
  /* GroupNorm statistics from the convolution's epilogue (ABI 9; bf16 (IM2ROW, ROW) launches on the lean kernels -- maps of more than
   * 256 output tiles --, N % 32 == 0): gn_rows[m][N / 32][2] receives {sum, sum of squares} of every 32-channel block of output row m,
   * taken from the ROUNDED 16-bit values the launch stores (what a separate statistics pass over C would read).  The density head's
   * Conv2d -> GroupNorm(8, 256) pairs (models_mae_cross.py:80-100): countr_groupnorm_relu_fwd_rows then reads 64 bytes per pixel
   * instead of the 512-byte pixel itself.  A launch that would run on a kernel without this epilogue fails instead of leaving the
   * buffer unwritten. */
  float* gn_rows;
} countr_gemm_args;

int countr_gemm(const countr_gemm_args* a, int dtype, int modeA, int modeB, void* stream);

/* 1 when countr_gemm on these arguments runs on a kernel whose epilogue writes a->gn_rows (ABI 9), else 0 -- shape, alignment and the
 * environment's kernel selectors decide; a caller asks before it replaces its statistics pass by countr_groupnorm_relu_fwd_rows. */
int countr_gemm_gn_rows(const countr_gemm_args* a, int dtype, int modeA, int modeB);

/* Number of [M]-slabs the launch described by a (rowsum_slabs ignored) writes to rowsum_partial.  Depends on the shape and on the
 * environment switches only; callers size rowsum_partial with it and pass the value back in a->rowsum_slabs. */
int countr_gemm_rowsum_slabs(const countr_gemm_args* a, int dtype, int modeA, int modeB);

/* Output tiles (= workgroups per split-K slab) the launch described by a would run with (pointers, splitk and partial may still be
 * unset): a split-K caller picks splitk ~ CUs / tiles.  128 x 128 tiles everywhere except the lean convolution weight gradient on
 * maps with Cin % 256 == 0 (128 x 256). */
int countr_gemm_tiles(const countr_gemm_args* a, int dtype, int modeA, int modeB);

/* n (1..10) independent launches of one (dtype, modeA, modeB) kind, with the results of n countr_gemm calls bit for bit, in ONE kernel
 * launch where a grouped form exists: bf16 (COL, COL) split-K launches, i.e. the weight gradients dW = dy^T x of a transformer block's
 * nn.Linear layers (autograd of Mlp / Attention / CrossAttention, models_crossvit.py:46-128; timm Block, models_mae_cross.py:32-34).
 * Why: those are 16-72 output tiles each -- alone each needs 3-16 split-K slabs to fill 256 CUs (fp32 partials written, then summed
 * again); together they fill the chip with one or two.  Otherwise the launches run one after the other.
 * countr_gemm_group_tiles: output tiles per split-K slab of that ONE launch (the caller picks a common splitk ~ CUs / tiles), or 0 when
 * the launches would run separately (then countr_gemm_tiles applies per launch). */
int countr_gemm_group(const countr_gemm_args* items, int n, int dtype, int modeA, int modeB, void* stream);
int countr_gemm_group_tiles(const countr_gemm_args* items, int n, int dtype, int modeA, int modeB);

/* out[M,N] (+)= sum_z partial[z][M][N]; optional permute for conv weights:
 * perm_taps > 0: partial is [Cout][taps][Cin] (OHWI) and out is torch OIHW [Cout][Cin][taps].
 * rowsum_partial/rowsum_out (optional): also reduces the fused bias-gradient partials [splitk][M] -> [M]. */
int countr_splitk_reduce(const float* partial, float* out, int splitk, int M, int N, int perm_taps,
                         int accumulate, const float* rowsum_partial, float* rowsum_out, void* stream);

/* Finisher of a split-K GEMM whose result is an ACTIVATION (forward / dgrad with few tiles and a long K, e.g. the 24x24 and 8x8
 * convolutions: Conv2d forward models_mae_cross.py:47-100 and its input gradient): out[M][N] (fp32 or bf16, dense) =
 * sum_z partial[z][M][N] (+ bias[n]).  N % 4 == 0. */
int countr_splitk_finish(const float* partial, void* out, const float* bias, int splitk, int M, int N, int out_bf16,
                         void* stream);

/* Many deferred slab reductions in ONE launch (the wgrad / bias-gradient / LayerNorm dgamma-dbeta finishers of a backward phase;
 * replaces the per-parameter accumulation of autograd's AccumulateGrad nodes, util/misc.py:266-280 loss_scaler -> backward()).
 * table: device int64 [n][8] rows {partial (const float*), out (float*), nslabs | accumulate << 32 | wide << 33, slab stride
 * (floats), count, N, perm_taps, first block}: out[perm(i)] (+)= sum_z partial[z*stride + i], i < count; entry e owns the
 * 256-thread blocks [first_block(e), first_block(e+1)), total_blocks in all: ceil(count / 256) blocks per entry, or
 * ceil(count / 16) for a "wide" entry (many slabs: 16 columns x 16 slab groups per block, no permutation).  The rows are
 * followed (at table + 8 n) by an int32 array [total_blocks] giving the entry of every block.
 * perm as in countr_splitk_reduce (taps > 0). */
int countr_reduce_table(const long long* table, int n, int total_blocks, void* stream);


/* -------- LayerNorm (nn.LayerNorm eps=1e-6: models_mae_cross.py:146,182; models_crossvit.py:153-155;
 * timm Block.norm1/norm2).  x is the fp32 residual stream [rows, D]; y is fp32 or bf16.
 * mean/rstd (fp32 [rows]) are optional outputs kept for the backward. */
int countr_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                         float* rstd, int rows, int D, float eps, int out_bf16, void* stream);
/* dx (+)= LN backward of dy; dgamma[D], dbeta[D] fp32 (may be NULL).  dx_bf16 (optional) receives a bf16 copy of the
 * updated dx (the operand of the next backward GEMM), saving a separate cast pass.
 * workspace: fp32 [countr_layernorm_bwd_nblocks()][2][D]. */
int countr_layernorm_bwd_nblocks(void);
int countr_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean,
                         const float* rstd, float* dx, float* dgamma, float* dbeta, float* workspace,
                         int rows, int D, int dy_bf16, int accumulate_dx, int accumulate_dgb, void* dx_bf16,
                         void* stream);
int countr_colsum_partials(const float* partial, float* out, int nparts, int C, int accumulate, void* stream);

/* -------- GroupNorm(8, 256) + ReLU on NHWC maps (decode_head*: models_mae_cross.py:80-100).
 * With w1 != NULL the 1x1 conv 256->1 of decode_head3 (:99) is fused: out1[b,p] = sum_c y*w1[c] + b1 and
 * y may be NULL.  stats: fp32 [B][G][2] (mean, rstd) output.  workspace: fp32 >= B*nsplit*3*256 + 64 + 16*B (forward),
 * + B*3*256 (backward: behind the split partials its finalize pass leaves the per-image sums [B][3][256] = {sum g, sum g*xhat,
 * sum d1*y} at float offset countr_groupnorm_bwd_image_sums_offset(B, HW), for callers that finish dbeta / dgamma / dw1 themselves). */
int countr_groupnorm_nsplit(int HW);
long long countr_groupnorm_bwd_image_sums_offset(int B, int HW);
int countr_groupnorm_relu_fwd(const void* x, const float* gamma, const float* beta, void* y, const float* w1,
                              const float* b1, float* out1, float* stats, float* workspace, int B, int HW,
                              int C, int G, float eps, int dtype, void* stream);
/* The same, with the statistics pass reading rows [B*HW][8][2] -- the {sum, sum of squares} per 32-channel group that the convolution
 * which produced x left through countr_gemm_args.gn_rows (ABI 9; 16-bit maps, G == 8) -- instead of the map: Conv2d -> GroupNorm of
 * decode_head1..3 on the big maps (models_mae_cross.py:86-100). */
int countr_groupnorm_relu_fwd_rows(const void* x, const float* rows, const float* gamma, const float* beta, void* y,
                                   const float* w1, const float* b1, float* out1, float* stats, float* workspace, int B,
                                   int HW, int C, int G, float eps, int dtype, void* stream);
/* dy (same dtype as x) is the gradient of y, or pass dy = NULL with d1 [B,HW] fp32 + w1 for the fused head.
 * dgamma/dbeta/dw1 [256], db1 [1] fp32 (each optional). */
int countr_groupnorm_relu_bwd(const void* x, const void* dy, const float* d1, const float* w1,
                              const float* stats, const float* gamma, const float* beta, void* dx,
                              float* dgamma, float* dbeta, float* dw1, float* db1, float* workspace, int B,
                              int HW, int C, int G, int dtype, int accumulate, void* stream);

/* -------- InstanceNorm2d(affine=False) + ReLU + MaxPool2d(2) | AdaptiveAvgPool2d(1) on NHWC
 * (decoder_proj1-4: models_mae_cross.py:47-71).  stats: fp32 [S][C][2]. */
/* workspace (optional, fp32 [countr_instnorm_workspace_floats(S, C)]): lets the wide first layers (few (sample, 64-channel)
 * blocks) run as two pixel-band kernels with Chan-combined statistics; NULL = single-kernel path. */
int countr_instnorm_workspace_floats(int S, int C);
/* xhat_out (optional; may be x itself = in place): the forward also stores the NORMALISED activation (x - mean) * rstd, rounded to the
 * dtype, and computes the pooled output from those rounded values; the backward is then given that buffer as x with x_is_xhat = 1.
 * What it is for: in bf16 an x-hat recomputed from a rounded x is off by (|mean| / sigma + |x-hat|) * 2^-9 relative, which the two
 * reductions of the InstanceNorm backward amplify (exemplar-CNN weight gradients: cosine 0.96 against fp32); a stored x-hat is good
 * to |x-hat| * 2^-9.
 * x_f32 = 1: the map x is fp32 although y / xhat_out have `dtype` (a convolution that writes its output -- bias included -- in fp32:
 * a bf16 map is rounded at 2^-9 of its VALUE, which for a channel whose mean is several sigma moves pixels across the ReLU boundary of
 * x-hat and costs the backward ~2 % in cosine per layer; the normalised activation does not have that problem). */
int countr_instnorm_relu_pool_fwd(const void* x, void* y, float* stats, int S, int H, int W, int C,
                                  int avgpool, float eps, int dtype, float* workspace, void* xhat_out, int x_f32, void* stream);
int countr_instnorm_relu_pool_bwd(const void* x, const void* dyp, const float* stats, void* dx, int S, int H,
                                  int W, int C, int avgpool, int dtype, float* workspace, int x_is_xhat, void* stream);

/* -------- fused self-attention forward (Attention.forward, models_crossvit.py:82-94 == timm Attention):
 * out = softmax(q k^T * scale) v on a packed bf16 qkv [B, N, 3, H, dh] (dh = 32 or 64) -> bf16 [B, N, H*dh].
 * lse: optional fp32 [B, H, N] log-sum-exp of the scaled scores.
 * scale <= 0 (dh = 64, N % 64 == 0 only): the caller has already multiplied q by dh^-0.5 * log2(e) -- e.g. by packing the
 * frozen encoder's q projection rows and bias pre-scaled, as engine.py does -- so the scores leave the matrix core in the exp2
 * domain and the kernel spends no VALU instruction on the scale / max subtraction. */
int countr_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int dh, float scale, void* stream);

/* Backward of countr_attn_fwd (autograd of models_crossvit.py:84-91), two fused passes, no P materialised, no atomics.
 * out/lse: the forward's outputs; dout bf16 [B, N, H*dh]; delta: fp32 [B, H, N] (must be non-NULL; not touched since both passes of the one launch compute it); dqkv bf16 [B, N, 3, H, dh]. */
int countr_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                    int B, int N, int H, int dh, float scale, void* stream);

/* -------- softmax rows for the unfused attention path (models_crossvit.py:87-88) */
int countr_softmax_fwd(const float* s, void* p, int64_t rows, int n, int out_bf16, void* stream);
/* The same with a row pitch ld >= n (ABI 6): columns n .. ld - 1 of p are written as zeros.  For token counts that are not a multiple of
 * the GEMM's 16-byte chunk (mae_vit_huge_patch14, models_mae_cross.py:235-239: 27 x 27 = 729 tokens): the score / probability
 * matrices are padded, nothing else is. */
int countr_softmax_fwd_ld(const float* s, void* p, int64_t rows, int n, int ld, int out_bf16, void* stream);
int countr_softmax_bwd(const void* p, const float* dp, void* ds, int64_t rows, int n, float scale, int dtype,
                       void* stream);

/* -------- cross attention against S <= 8 exemplar tokens (CrossAttention.forward,
 * models_crossvit.py:111-128; D = 512, 16 heads of 32).  q/out [B*N, D]; k, v [B*S, ldkv]. */
int countr_xattn_fwd(const void* q, const void* k, const void* v, void* out, int B, int N, int S, int D,
                     int heads, int ldkv, float scale, int dtype, void* stream);
int64_t countr_xattn_bwd_workspace_floats(int B, int N, int S, int D);
/* dk / dv: fp32 [B*S, D]; dk_bf16 / dv_bf16 (optional): bf16 copies of the same sums -- the operands of the wk / wv backward GEMMs */
int countr_xattn_bwd(const void* q, const void* k, const void* v, const void* dout, void* dq, float* dk,
                     float* dv, float* workspace, int B, int N, int S, int D, int heads, int ldkv, float scale,
                     int dtype, void* dk_bf16, void* dv_bf16, void* stream);

/* -------- data movement / elementwise */
/* timm PatchEmbed gather (models_mae_cross.py:138): img fp32 NCHW -> patches [B*gh*gw, 3*p*p], k=(c,py,px) */
int countr_im2patch(const float* img, void* out, int B, int H, int W, int patch, int dtype, void* stream);
/* decoder_proj1 conv 3->64 (models_mae_cross.py:48): in fp32 NCHW [S,3,H,W], w fp32 OIHW, out NHWC */
int countr_conv3x3_c3_fwd(const float* in, const float* w, const float* bias, void* out, int S, int H, int W,
                          int dtype, void* stream);
int countr_conv3x3_c3_wgrad_nblocks(void);
/* workspace: fp32 [nblocks][64*28] */
int countr_conv3x3_c3_wgrad(const float* in, const void* dy, float* dw, float* db, float* workspace, int S,
                            int H, int W, int dtype, int accumulate, void* stream);
/* F.interpolate(x2, bilinear, align_corners=False) on NHWC (models_mae_cross.py:189-196); C==1 or C%8==0 */
int countr_upsample2x_fwd(const void* in, void* out, int B, int H, int W, int C, int dtype, void* stream);
int countr_upsample2x_bwd(const void* dout, void* din, int B, int H, int W, int C, int dtype, void* stream);
int countr_gelu_bwd(const void* dh, const void* pre, void* dpre, int64_t n, int dtype, void* stream);
/* bias gradient: out[N] (+)= column sums of x [M,N]; workspace fp32 [countr_colsum_nparts()][N] */
int countr_colsum_nparts(void);
int countr_colsum(const void* x, float* out, float* workspace, int M, int N, int dtype, int accumulate,
                  void* stream);
/* weight shadows: mode 0 cast, 1 OIHW->OHWI, 2 conv-dgrad form Wd[ci][tap'][co] = W[co][ci][T-1-tap'] */
/* up to 8 device-to-device copies in one launch (src / dst / bytes are HOST arrays; 16-byte aligned pointers and sizes): the
 * per-iteration staging of a batch -- `samples.to(device)`, `boxes.to(device)`, `gt_density.to(device)` and the fresh loss mask of
 * FSC_finetune_cross.py:270-296 -- into the static input buffers of a plan.  src[i] == NULL zero-fills dst[i] (the gradients DDP's
 * find_unused_parameters=True, FSC_finetune_cross.py:230, contributes for parameters a rank did not use in an iteration). */
int countr_copy_multi(int n, const void* const* src, void* const* dst, const int64_t* bytes, void* stream);
/* Step prologue (ABI 5): what an optimisation step takes from the host per iteration -- the batch hand-over (samples / gt_density /
 * boxes .to(device): FSC_finetune_cross.py:273-275), the Bernoulli loss mask drawn per iteration (:290-292: np.random.binomial(1, 0.8,
 * [384, 384])) and the iteration's AdamW scalars (lr of lr_sched.adjust_learning_rate :271, bias corrections) -- as ONE launch with
 * FROZEN arguments, so that it can be the first node of the step's captured hipGraph and nothing is launched between two replays.
 * ring: `slots` records of countr_step_prologue_record_bytes() (= 256) bytes in memory the device can read and the host can write
 * (pinned host memory); counter: device int64[34] (16-byte aligned), zero-initialised by the caller, owned by the library afterwards:
 * counter[0] = executions so far, [2..33] = device copy of the current record.  Execution k (eager or replayed) reads record k % slots
 * -- a one-block kernel fetches it from the host and increments counter[0], a wide kernel does the work -- so the host counts
 * executions and fills record k before execution k, not before execution k - slots has finished.  Record layout (little endian): u64 src[6], u64 dst[6], i64 n16[6]
 * (16-byte units; src 0 = zero fill), i32 first[6] (first of the countr_step_prologue_copy_blocks() copy blocks dealt to entry i,
 * ascending, first[0] = 0), i32 n (0..6 copies), i32 draw_mask, u32 key[2], u32 ctr[2], f32 hyper[8], u32 mask_thr, 28 bytes pad.
 * hyper_dev: device fp32[8] <- hyper.  mask (NULL: none), mask_n % 4 == 0: with draw_mask, element 4g + j = (word j of
 * Philox4x32-10(counter = (g, 0, ctr[0], ctr[1]), key) < mask_thr) ? 1 : 0 -- Bernoulli(mask_thr / 2^32), reproducible from (key, ctr). */
int countr_step_prologue_record_bytes(void);
int countr_step_prologue_copy_blocks(void);
int countr_step_prologue(const void* ring, int slots, int64_t* counter, float* hyper_dev, float* mask, int mask_n, void* stream);
int countr_cast_permute(const float* src, void* dst, int64_t n, int mode, int Co, int Ci, int taps, int dtype,
                        void* stream);

/* -------- loss + optimizer (FSC_finetune_cross.py:290-303, :235) */
/* sums (fp32 [1+2B]) = {loss, pred counts[B], gt counts[B]}; dpred optional (= dloss/dpred * grad_scale);
 * workspace: fp32 [countr_masked_mse_workspace_floats(B)].  Deterministic two-stage reduction. */
/* both permuted shadows (OHWI and dgrad form, see countr_cast_permute modes 1 and 2) of up to 32 weights in ONE launch;
 * the arrays are HOST arrays of n device pointers / shapes (the optimiser step refreshes all shadows at once).  wf[i] may be NULL
 * (only wd is written); with taps = 1 a weight is an nn.Linear matrix [co][ci] and wd its TRANSPOSE [ci][co] -- the K-contiguous
 * B operand that turns the input gradient dx = dy W (autograd of models_crossvit.py:62-65, 84-92, 115-127) into a (ROW, ROW) GEMM. */
int countr_conv_shadows(int n, const float* const* src, void* const* wf, void* const* wd, const int* co, const int* ci,
                        const int* taps, int dtype, void* stream);
/* ABI 8: dst[i][c][r] = src[i][r][c] for up to 96 matrices of 16-bit elements (rows, cols multiples of 64; 16-byte aligned) in ONE launch:
 * the W^T shadows (see countr_conv_shadows, taps = 1) taken from the 16-bit shadow W that countr_adamw_step has just written instead of
 * from the fp32 master -- same bits, half the bytes read.  Host arrays of n device pointers / shapes. */
int countr_transpose16(int n, const void* const* src, void* const* dst, const int* rows, const int* cols, void* stream);
int countr_masked_mse_workspace_floats(int B);
int countr_masked_mse(const float* pred, const float* gt, const float* mask, float* dpred, float* sums,
                      float* workspace, int B, int HW, float grad_scale, void* stream);
/* fused AdamW over flat fp32 buffers (torch.optim.AdamW at FSC_finetune_cross.py:235); up to 16 [start,end) ranges each with
 * its weight decay, its bias-correction counter group (groups[i] in {0,1,2}, NULL = all 0: torch keeps a step counter per
 * parameter, util/misc.py:266-280 steps only parameters that have a gradient) and a zero-gradient flag (zero_grad[i] != 0: the
 * range is stepped with g = 0, which is what torch 1.13's zero_grad() -- zero tensors, not None -- makes of a parameter that is
 * unused in this iteration but had a gradient before).  hyper_dev (optional, device fp32[8] = {lr, bc1[0], bc2[0], grad_scale,
 * bc1[1], bc2[1], bc1[2], bc2[2]}, bc = 1 - beta^t of the group) overrides the scalars so a captured graph can be replayed with
 * new values; without it every group uses `step`.  shadow_bf16 (optional) receives bf16(p).  gnorm_ws (optional,
 * countr_adamw_gnorm_floats() floats): gnorm_ws[0] = L2 norm of the (scaled) gradients of the stepped ranges
 * (get_grad_norm_, util/misc.py:289-301). */
/* Dynamic loss scaling of the fp16 mode (ABI 7) = torch.cuda.amp.GradScaler as the reference uses it (util/misc.py:260-286: scale the
 * loss, unscale_, skip the optimizer step on a non-finite gradient, update()), on the device so that a captured step needs no host
 * decision.  amp: device fp32[8] owned by the caller = {scale (GradScaler's initial 65536), steps since the last change, found_inf,
 * skipped steps so far, growth interval (GradScaler's 2000), reserved x 3}.
 *   countr_masked_mse_amp / countr_patch_mse_amp: the loss gradient is multiplied by grad_scale * amp[0];
 *   countr_adamw_step_amp: amp[2] <- any non-finite gradient in the ranges it reads (after an all-reduce every rank sees the same);
 *     if set: parameters, moments and shadows are left untouched, gnorm_ws[0] = inf, scale *= 0.5, amp[3] += 1; otherwise the gradients
 *     are divided by amp[0] on the fly, and after `interval` updates without a skip scale *= 2.  amp == NULL: countr_adamw_step. */
int countr_masked_mse_amp(const float* pred, const float* gt, const float* mask, float* dpred, float* sums,
                          float* workspace, int B, int HW, float grad_scale, const float* amp, void* stream);
int countr_patch_mse_amp(const float* pred, const float* imgs, void* dpred, float* loss, float* workspace, int B, int H, int W,
                         int patch, int norm_pix, float grad_scale, int dpred_dtype, const float* amp, void* stream);
int countr_adamw_step_amp(float* p, const float* g, float* m, float* v, void* shadow_bf16, int nranges,
                          const int64_t* starts, const int64_t* ends, const float* wds, const int* groups,
                          const int* zero_grad, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                          const float* hyper_dev, float* gnorm_ws, float* amp, void* stream);
int countr_adamw_gnorm_floats(void);
int countr_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int nranges,
                      const int64_t* starts, const int64_t* ends, const float* wds, const int* groups, const int* zero_grad,
                      float lr, float beta1, float beta2, float eps, int step, float grad_scale, const float* hyper_dev,
                      float* gnorm_ws, void* stream);

/* ---- MAE pretraining (reference models_mae_noct.py) ----
 * row gather: dst[r,:] = (idx[r] >= 0 ? src[idx[r],:] : default_row[:]) + add[r % add_mod,:]; idx may be NULL (identity),
 * default_row/add may be NULL, add_mod <= 0 means add[r,:].  cols % 4 == 0; src/dst dtype codes independent.  Covers
 * random_masking's gather of kept tokens (:125-126), the unshuffle with mask tokens + decoder_pos_embed (:166-172) and
 * the backward of both. */
/* index buffers of random_masking (:119-132) from ids_shuffle [B][N] int64 (argsort of the noise): ids_restore [B][N] int64 (its
 * inverse permutation), keep_pos / keep_src [B*K] int32 (kept token = position s, row s + b N of the patch matrix), restore_src
 * [B*N] int32 (row of the kept-token matrix, or -1 = mask token), mask_src [B*(N-K)] int32 (rows of the masked tokens; may be NULL
 * when K == N), mask [B*N] fp32 (1 = masked). */
int countr_mae_indices(const long long* ids_shuffle, long long* ids_restore, int* keep_pos, int* keep_src, int* restore_src,
                       int* mask_src, float* mask, int B, int N, int K, void* stream);
int countr_gather_rows(const void* src, const int* idx, void* dst, const float* default_row, const float* add, int add_mod,
                       int rows, int cols, int src_dtype, int dst_dtype, void* stream);
/* forward_loss (:181-198): target = patchify(imgs) in (py,px,c) order (:84-96), optional norm_pix (unbiased var, eps 1e-6),
 * loss[0] = mean over ALL patches of the per-patch mean squared error; dpred (optional, dtype dpred_dtype) =
 * grad_scale * dloss/dpred.  pred fp32 [B*L, 3*patch^2]; imgs fp32 NCHW; workspace fp32[countr_patch_mse_workspace_floats].
 * Deterministic two-stage reduction (no atomics). */
int countr_patch_mse_workspace_floats(int B, int H, int W, int patch);
int countr_patch_mse(const float* pred, const float* imgs, void* dpred, float* loss, float* workspace, int B, int H, int W,
                     int patch, int norm_pix, float grad_scale, int dpred_dtype, void* stream);

/* ---- sliding-window test path (FSC_test_cross(few-shot).py:326-349, demo_zero.py:49-72): images of height H = 384 are covered by
 * 384-px windows (stride 128, the last snapped to w - 384), every window is one forward, densities are stitched column by column --
 * a column a previous window already covered becomes old / 2 + new / 2, in window order.
 * countr_window_gather: window j (0 <= j < nw <= 64) = columns [starts[j], starts[j] + 384) of the fp32 image frames[j] [3, H, widths[j]]
 *   (frames / widths / starts: HOST arrays, read at call time) -> wins fp32 [nw, 3, H, 384], normally the engine's input batch.
 * countr_window_blend: outs fp32 [n * nwin, H, 384] (image-major: the nwin windows of image 0, then of image 1, ...) of n images of ONE
 *   width W with the window starts starts[nwin] (HOST array, increasing, <= 16) -> dm fp32 [n, H, W]; sums (optional) fp32 [n] = sum
 *   of each stitched map (the predicted count x 60), deterministic two-pass sum through workspace fp32 [n * countr_window_blend_blocks(H, W)].
 * Both equal the reference's tensor slicing / blending bit for bit (x / 2 is exact in fp32). */
int countr_window_gather(const void* const* frames, const int* widths, const int* starts, int nw, int H, float* wins, void* stream);
int countr_window_blend(const float* outs, int n, int nwin, const int* starts, int H, int W, float* dm, float* sums, float* workspace, void* stream);
int countr_window_blend_blocks(int H, int W);

#ifdef __cplusplus
}
#endif
#endif /* COUNTR_HIP_H */
