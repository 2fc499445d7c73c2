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

/* -------- library management -------- */
int countr_init(int device);            /* selects device, checks it is gfx950-class; 0 = ok   */
int countr_version(void);               /* ABI version, currently 1                            */
const char* countr_last_error(void);    /* thread-local message of the last failing call        */

/*
 * Generic tiled MFMA GEMM:  C[m,n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
 *   epilogue(v) = act(v + bias[n]) + resid[(m % res_mod) , n]      (each part optional)
 *   C2 (optional) receives v + bias[n] before the activation (saved for GELU backward).
 * A is the M-side operand, B the N-side operand, each in one of the COUNTR_OP_* modes.
 * Batched when nbatch > 1: z = b0*nb1 + b1 adds b0*s?0 + b1*s?1 (in elements) to A, B, C.
 * Split-K when splitk > 1 (nbatch must be 1): block z handles a K range and stores raw fp32 sums
 * to partial[z][M][N]; finish with countr_splitk_reduce.
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
} countr_gemm_args;

int countr_gemm(const countr_gemm_args* a, int dtype, int modeA, int modeB, void* stream);

/* out[M,N] (+)= sum_z partial[z][M][N]; optional permute for conv weights:
 * perm_taps > 0: partial is [Cout][taps][Cin] (OHWI) and out is torch OIHW [Cout][Cin][taps]. */
int countr_splitk_reduce(const float* partial, float* out, int splitk, int M, int N, int perm_taps,
                         int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COUNTR_HIP_H */
