// Attention pieces (reference: models_crossvit.py:69-128).
//   softmax rows fwd/bwd  : the unfused self-attention path (scores via countr_gemm), fp32 statistics
//   cross attention       : q [B*N, D] against S exemplar tokens (<= 8: keys in registers; more: online softmax); one wave per query row
//   (the fused flash-style self-attention forward lives in flash_attn.hip)
#include "common.hpp"
#include "../../include/countr_hip.h"

namespace {

constexpr int SMAX_PER_LANE = 16;  // row length <= 1024

// P = softmax(S) row-wise; S fp32 (already scaled), P stored as TO.  One wave per row.
// ld >= n: row pitch of both matrices; columns n .. ld - 1 of P are written as zeros (a token count that is not a multiple of the GEMM's
// 16-byte chunk -- 27 x 27 = 729 tokens of mae_vit_huge_patch14 -- is padded in the score / probability matrices only).
template <typename TO>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s, TO* __restrict__ p, int64_t rows, int n, int ld) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + row * ld;
  float v[SMAX_PER_LANE];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SMAX_PER_LANE; ++i) {
    const int c = i * 64 + lane;
    v[i] = (c < n) ? sr[c] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < SMAX_PER_LANE; ++i) {
    v[i] = (i * 64 + lane < n) ? __expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  const float inv = 1.f / wave_sum(sum);
  TO* pr = p + row * ld;
#pragma unroll
  for (int i = 0; i < SMAX_PER_LANE; ++i) {
    const int c = i * 64 + lane;
    if (c < n) stf<TO>(pr + c, v[i] * inv);
    else if (c < ld) stf<TO>(pr + c, 0.f);
  }
}

// dS = P * (dP - sum(dP * P)) * scale ; P is T, dP fp32 (from the dO V^T GEMM), dS stored as T.
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ p, const float* __restrict__ dp, T* __restrict__ ds,
                                                          int64_t rows, int n, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float pv[SMAX_PER_LANE], dv[SMAX_PER_LANE];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < SMAX_PER_LANE; ++i) {
    const int c = i * 64 + lane;
    pv[i] = (c < n) ? ldf<T>(p + row * n + c) : 0.f;
    dv[i] = (c < n) ? dp[row * n + c] : 0.f;
    dot += pv[i] * dv[i];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < SMAX_PER_LANE; ++i) {
    const int c = i * 64 + lane;
    if (c < n) stf<T>(ds + row * n + c, pv[i] * (dv[i] - dot) * scale);
  }
}

// ------------------------------------------------------------------------------------------
// Cross attention (models_crossvit.py:111-128) for S <= 8 keys.  D = H * dh, dh == 32:
// one wave per query row: lane = head * 4 + quarter, each lane owns 8 channels of its head.
// ------------------------------------------------------------------------------------------
constexpr int XS = 8;

template <typename T>
__global__ __launch_bounds__(256) void xattn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        T* __restrict__ o, int B, int N, int S, int D, int ldkv, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * N) return;
  const int b = (int)(row / N);
  for (int c0 = lane * 8; c0 < D; c0 += 512) {  // D = 512 -> one pass
    float qv[8];
    ld8<T>(q + row * D + c0, qv);
    float sc[XS];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < XS; ++j) {
      sc[j] = -INFINITY;
      if (j < S) {
        float kv[8];
        ld8<T>(k + ((int64_t)b * S + j) * ldkv + c0, kv);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d += qv[e] * kv[e];
        d = quad_sum(d);
        sc[j] = d * scale;
        mx = fmaxf(mx, sc[j]);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < XS; ++j) { sc[j] = (j < S) ? __expf(sc[j] - mx) : 0.f; sum += sc[j]; }
    const float inv = 1.f / sum;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int j = 0; j < XS; ++j) {
      if (j < S) {
        float vv[8];
        ld8<T>(v + ((int64_t)b * S + j) * ldkv + c0, vv);
        const float pj = sc[j] * inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += pj * vv[e];
      }
    }
    st8<T>(o + row * D + c0, acc);
  }
}

// Backward: recomputes P; dq per row; dk/dv are reduced over the rows of a block in LDS and written as
// per-block partials  partial[blockIdx.x][2][S][D]  (blocks never straddle a batch element).
// KS = compile-time bound on the number of keys (1, 2, 3 or XS): the per-lane key / value / dk / dv register arrays scale with
// it (4 x KS x 8 floats), and shot_num <= 3 in the reference (FSC_finetune_cross.py:278-284), so XS-sized arrays were 2.7x the
// registers and arithmetic the common case needs.
template <typename T, int KS>
__global__ __launch_bounds__(256) void xattn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        const T* __restrict__ dout, T* __restrict__ dq, float* __restrict__ partial,
                                                        int N, int S, int D, int ldkv, float scale, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem_x[];
  float* sm = reinterpret_cast<float*>(smem_x);  // [4 waves][2][S][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blocks_per_b = (N + rows_per_block - 1) / rows_per_block;
  const int b = blockIdx.x / blocks_per_b;
  const int r0 = (blockIdx.x - b * blocks_per_b) * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  const int c0 = lane * 8;  // D == 512
  float kv[KS][8], vv[KS][8], dk[KS][8], dvv[KS][8];
#pragma unroll
  for (int j = 0; j < KS; ++j) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { dk[j][e] = 0.f; dvv[j][e] = 0.f; kv[j][e] = 0.f; vv[j][e] = 0.f; }
    if (j < S) {
      ld8<T>(k + ((int64_t)b * S + j) * ldkv + c0, kv[j]);
      ld8<T>(v + ((int64_t)b * S + j) * ldkv + c0, vv[j]);
    }
  }
  for (int r = r0 + wave; r < r1; r += 4) {
    const int64_t row = (int64_t)b * N + r;
    float qv[8], dov[8];
    ld8<T>(q + row * D + c0, qv);
    ld8<T>(dout + row * D + c0, dov);
    float sc[KS], dp[KS];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      float d = 0.f, g = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { d += qv[e] * kv[j][e]; g += dov[e] * vv[j][e]; }
      d = quad_sum(d);
      g = quad_sum(g);
      sc[j] = (j < S) ? d * scale : -INFINITY;
      dp[j] = g;
      mx = fmaxf(mx, sc[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j) { sc[j] = (j < S) ? __expf(sc[j] - mx) : 0.f; sum += sc[j]; }
    const float inv = 1.f / sum;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j) { sc[j] *= inv; dot += sc[j] * dp[j]; }
    float dqv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dqv[e] = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const float ds = sc[j] * (dp[j] - dot) * scale;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dqv[e] += ds * kv[j][e];
        dk[j][e] += ds * qv[e];
        dvv[j][e] += sc[j] * dov[e];
      }
    }
    st8<T>(dq + row * D + c0, dqv);
  }
  // reduce the 4 waves and emit this block's partial
  for (int j = 0; j < S; ++j) {
    st8<float>(sm + ((wave * 2 + 0) * S + j) * D + c0, dk[j]);
    st8<float>(sm + ((wave * 2 + 1) * S + j) * D + c0, dvv[j]);
  }
  __syncthreads();
  const int tot = 2 * S * D;
  for (int i = threadIdx.x; i < tot; i += 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += sm[w * tot + i];
    partial[(int64_t)blockIdx.x * tot + i] = s;
  }
}

// ---- any number of keys (models_crossvit.py:111-128 has no limit; FSC_test_cross(few-shot).py:56,138-139 defaults to --box_bound -1 =
// every annotated box as an exemplar).  Same lane layout; the keys are walked with an online softmax instead of sitting in registers.
template <typename T>
__global__ __launch_bounds__(256) void xattn_fwd_any_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                            T* __restrict__ o, int B, int N, int S, int D, int ldkv, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * N) return;
  const int b = (int)(row / N);
  const int c0 = lane * 8;                    // D == 512
  float qv[8], acc[8];
  ld8<T>(q + row * D + c0, qv);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  float mx = -INFINITY, sum = 0.f;
  for (int j = 0; j < S; ++j) {
    float kv[8], vv[8];
    ld8<T>(k + ((int64_t)b * S + j) * ldkv + c0, kv);
    ld8<T>(v + ((int64_t)b * S + j) * ldkv + c0, vv);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) d += qv[e] * kv[e];
    d = quad_sum(d) * scale;
    const float mn = fmaxf(mx, d);
    const float a = __expf(mx - mn), pj = __expf(d - mn);      // (first key: exp(-inf) = 0)
    sum = sum * a + pj;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] * a + pj * vv[e];
    mx = mn;
  }
  const float inv = 1.f / sum;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] *= inv;
  st8<T>(o + row * D + c0, acc);
}

// Backward for any S: per query row the softmax statistics (max, 1 / sum, sum_j p_j dp_j) are computed once, then the keys are walked in
// chunks of XCHUNK whose dk / dv contributions of the block's rows go to LDS -- one region per wave, summed in wave order: no atomics,
// bit-reproducible like every other reduction here -- and are written as per-block partials, the layout of xattn_bwd_kernel:
// [block][2][S][D].
constexpr int XCHUNK = 4, XROWS = 4;          // rows per wave (XATTN_ROWS_PER_BLOCK / 4 waves)
template <typename T>
__global__ __launch_bounds__(256) void xattn_bwd_any_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                            const T* __restrict__ dout, T* __restrict__ dq, float* __restrict__ partial,
                                                            int N, int S, int D, int ldkv, float scale, int rows_per_block) {
  __shared__ float sm[4 * 2 * XCHUNK * 512];   // [4 waves][2][XCHUNK][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blocks_per_b = (N + rows_per_block - 1) / rows_per_block;
  const int b = blockIdx.x / blocks_per_b;
  const int r0 = (blockIdx.x - b * blocks_per_b) * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  const int c0 = lane * 8;
  float qv[XROWS][8], dov[XROWS][8], dqv[XROWS][8], mx[XROWS], inv[XROWS], dot[XROWS];
#pragma unroll
  for (int u = 0; u < XROWS; ++u) {
    const int r = r0 + wave + 4 * u;
    mx[u] = -INFINITY; inv[u] = 0.f; dot[u] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { qv[u][e] = 0.f; dov[u][e] = 0.f; dqv[u][e] = 0.f; }
    if (r < r1) {
      const int64_t row = (int64_t)b * N + r;
      ld8<T>(q + row * D + c0, qv[u]);
      ld8<T>(dout + row * D + c0, dov[u]);
    }
  }
  // pass 1: max and sum; pass 2: sum_j p_j dp_j
  for (int j = 0; j < S; ++j) {
    float kv[8];
    ld8<T>(k + ((int64_t)b * S + j) * ldkv + c0, kv);
#pragma unroll
    for (int u = 0; u < XROWS; ++u) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d += qv[u][e] * kv[e];
      d = quad_sum(d) * scale;
      const float mn = fmaxf(mx[u], d);
      inv[u] = inv[u] * __expf(mx[u] - mn) + __expf(d - mn);
      mx[u] = mn;
    }
  }
#pragma unroll
  for (int u = 0; u < XROWS; ++u) inv[u] = 1.f / inv[u];
  for (int j = 0; j < S; ++j) {
    float kv[8], vv[8];
    ld8<T>(k + ((int64_t)b * S + j) * ldkv + c0, kv);
    ld8<T>(v + ((int64_t)b * S + j) * ldkv + c0, vv);
#pragma unroll
    for (int u = 0; u < XROWS; ++u) {
      float d = 0.f, g = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { d += qv[u][e] * kv[e]; g += dov[u][e] * vv[e]; }
      d = quad_sum(d) * scale;
      g = quad_sum(g);
      dot[u] += __expf(d - mx[u]) * inv[u] * g;
    }
  }
  const int tot = 2 * S * D;
  for (int j0 = 0; j0 < S; j0 += XCHUNK) {
    const int jn = min(XCHUNK, S - j0);
    for (int jj = 0; jj < jn; ++jj) {
      const int j = j0 + jj;
      float kv[8], vv[8];
      ld8<T>(k + ((int64_t)b * S + j) * ldkv + c0, kv);
      ld8<T>(v + ((int64_t)b * S + j) * ldkv + c0, vv);
      float dk[8], dvv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { dk[e] = 0.f; dvv[e] = 0.f; }
#pragma unroll
      for (int u = 0; u < XROWS; ++u) {
        float d = 0.f, g = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { d += qv[u][e] * kv[e]; g += dov[u][e] * vv[e]; }
        d = quad_sum(d) * scale;
        g = quad_sum(g);
        const bool ok = r0 + wave + 4 * u < r1;
        const float pj = ok ? __expf(d - mx[u]) * inv[u] : 0.f;
        const float ds = pj * (g - dot[u]) * scale;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dqv[u][e] += ds * kv[e];
          dk[e] += ds * qv[u][e];
          dvv[e] += pj * dov[u][e];
        }
      }
      st8<float>(sm + ((wave * 2 + 0) * XCHUNK + jj) * 512 + c0, dk);
      st8<float>(sm + ((wave * 2 + 1) * XCHUNK + jj) * 512 + c0, dvv);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * jn * 512; i += 256) {
      const int which = i / (jn * 512), rem = i - which * jn * 512;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += sm[(w * 2 + which) * XCHUNK * 512 + rem];
      partial[(int64_t)blockIdx.x * tot + (int64_t)which * S * D + (int64_t)j0 * D + rem] = t;
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < XROWS; ++u) {
    const int r = r0 + wave + 4 * u;
    if (r < r1) st8<T>(dq + ((int64_t)b * N + r) * D + c0, dqv[u]);
  }
}

// dkv[b][which][j][:] = sum over the blocks of batch b
__global__ void xattn_bwd_finish_kernel(const float* __restrict__ partial, float* __restrict__ dk, float* __restrict__ dv,
                                        bf16_t* __restrict__ dk_bf16, bf16_t* __restrict__ dv_bf16, int blocks_per_b, int S, int D) {
  const int b = blockIdx.y;
  const int tot = 2 * S * D;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tot) return;
  float s4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent chains: the walk over the block partials is load-latency bound
  int p = 0;
  for (; p + 4 <= blocks_per_b; p += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] += partial[((int64_t)b * blocks_per_b + p + u) * tot + i];
  }
  for (; p < blocks_per_b; ++p) s4[0] += partial[((int64_t)b * blocks_per_b + p) * tot + i];
  const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  const int which = i / (S * D), r = i - which * S * D;
  (which == 0 ? dk : dv)[(int64_t)b * S * D + r] = s;
  bf16_t* const t = which == 0 ? dk_bf16 : dv_bf16;    // optional bf16 copy: the operand of the wk / wv wgrad and dgrad GEMMs
  if (t) t[(int64_t)b * S * D + r] = f2bf(s);
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int countr_softmax_fwd_ld(const float* s, void* p, int64_t rows, int n, int ld, int out_bf16, void* stream) {
  if (!s || !p || n <= 0 || ld < n || ld > 64 * SMAX_PER_LANE) { countr_set_error("countr_softmax_fwd: need 0 < n <= ld <= 1024"); return -1; }
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (out_bf16) hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, grid, block, 0, STREAM(stream), s, (bf16_t*)p, rows, n, ld);
  else hipLaunchKernelGGL(softmax_fwd_kernel<float>, grid, block, 0, STREAM(stream), s, (float*)p, rows, n, ld);
  COUNTR_LAUNCH_CHECK("countr_softmax_fwd");
}
extern "C" int countr_softmax_fwd(const float* s, void* p, int64_t rows, int n, int out_bf16, void* stream) {
  return countr_softmax_fwd_ld(s, p, rows, n, n, out_bf16, stream);
}

extern "C" int countr_softmax_bwd(const void* p, const float* dp, void* ds, int64_t rows, int n, float scale, int dtype,
                                  void* stream) {
  if (!p || !dp || !ds || n > 64 * SMAX_PER_LANE) { countr_set_error("countr_softmax_bwd: bad args"); return -1; }
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, grid, block, 0, STREAM(stream), (const bf16_t*)p, dp, (bf16_t*)ds, rows, n, scale);
  else hipLaunchKernelGGL(softmax_bwd_kernel<float>, grid, block, 0, STREAM(stream), (const float*)p, dp, (float*)ds, rows, n, scale);
  COUNTR_LAUNCH_CHECK("countr_softmax_bwd");
}

extern "C" int countr_xattn_fwd(const void* q, const void* k, const void* v, void* out, int B, int N, int S, int D, int heads,
                                int ldkv, float scale, int dtype, void* stream) {
  if (!q || !k || !v || !out || S < 1 || D != 512 || heads * 32 != D) { countr_set_error("countr_xattn_fwd: need S >= 1, D == 512, head_dim == 32"); return -1; }
  dim3 grid((unsigned)(((int64_t)B * N + 3) / 4)), block(256);
  if (S > XS) {      // more keys than the register form holds: online softmax over the key rows
    if (dtype == COUNTR_BF16) hipLaunchKernelGGL(xattn_fwd_any_kernel<bf16_t>, grid, block, 0, STREAM(stream), (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, B, N, S, D, ldkv, scale);
    else hipLaunchKernelGGL(xattn_fwd_any_kernel<float>, grid, block, 0, STREAM(stream), (const float*)q, (const float*)k, (const float*)v, (float*)out, B, N, S, D, ldkv, scale);
    COUNTR_LAUNCH_CHECK("countr_xattn_fwd");
  }
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL(xattn_fwd_kernel<bf16_t>, grid, block, 0, STREAM(stream), (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, B, N, S, D, ldkv, scale);
  else hipLaunchKernelGGL(xattn_fwd_kernel<float>, grid, block, 0, STREAM(stream), (const float*)q, (const float*)k, (const float*)v, (float*)out, B, N, S, D, ldkv, scale);
  COUNTR_LAUNCH_CHECK("countr_xattn_fwd");
}

static constexpr int XATTN_ROWS_PER_BLOCK = 16;   // 8 x 36 = 288 blocks at B = 8: backward 11.8 us (32 rows per block: 14.8 us)
extern "C" int64_t countr_xattn_bwd_workspace_floats(int B, int N, int S, int D) {
  const int bpb = (N + XATTN_ROWS_PER_BLOCK - 1) / XATTN_ROWS_PER_BLOCK;
  return (int64_t)B * bpb * 2 * S * D;
}

// dk, dv: fp32 [B, S, D] (overwritten)
extern "C" int countr_xattn_bwd(const void* q, const void* k, const void* v, const void* dout, void* dq, float* dk, float* dv,
                                float* workspace, int B, int N, int S, int D, int heads, int ldkv, float scale, int dtype,
                                void* dk_bf16, void* dv_bf16, void* stream) {
  if (!q || !k || !v || !dout || !dq || !dk || !dv || !workspace || S < 1 || D != 512 || heads * 32 != D) { countr_set_error("countr_xattn_bwd: bad args (S >= 1, D == 512, head_dim == 32)"); return -1; }
  const int bpb = (N + XATTN_ROWS_PER_BLOCK - 1) / XATTN_ROWS_PER_BLOCK;
  if (S > XS) {
    static_assert(XATTN_ROWS_PER_BLOCK == 4 * XROWS, "xattn_bwd_any_kernel holds XROWS rows per wave");
    if (dtype == COUNTR_BF16) hipLaunchKernelGGL(xattn_bwd_any_kernel<bf16_t>, dim3(B * bpb), dim3(256), 0, STREAM(stream), (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout, (bf16_t*)dq, workspace, N, S, D, ldkv, scale, XATTN_ROWS_PER_BLOCK);
    else hipLaunchKernelGGL(xattn_bwd_any_kernel<float>, dim3(B * bpb), dim3(256), 0, STREAM(stream), (const float*)q, (const float*)k, (const float*)v, (const float*)dout, (float*)dq, workspace, N, S, D, ldkv, scale, XATTN_ROWS_PER_BLOCK);
    hipLaunchKernelGGL(xattn_bwd_finish_kernel, dim3((2 * S * D + 255) / 256, B), dim3(256), 0, STREAM(stream), workspace, dk, dv,
                       (bf16_t*)dk_bf16, (bf16_t*)dv_bf16, bpb, S, D);
    COUNTR_LAUNCH_CHECK("countr_xattn_bwd");
  }
  const size_t lds = (size_t)4 * 2 * S * D * sizeof(float);
#define COUNTR_XB_LAUNCH(TT, KSV)                                                                                             \
  do {                                                                                                                         \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_bwd_kernel<TT, KSV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              4 * 2 * XS * 512 * 4);                                                                           \
    hipLaunchKernelGGL((xattn_bwd_kernel<TT, KSV>), dim3(B * bpb), dim3(256), lds, STREAM(stream), (const TT*)q, (const TT*)k,   \
                       (const TT*)v, (const TT*)dout, (TT*)dq, workspace, N, S, D, ldkv, scale, XATTN_ROWS_PER_BLOCK);         \
  } while (0)
#define COUNTR_XB_DISPATCH(TT)                                                                                                 \
  do {                                                                                                                         \
    if (S == 1) COUNTR_XB_LAUNCH(TT, 1); else if (S == 2) COUNTR_XB_LAUNCH(TT, 2); else if (S == 3) COUNTR_XB_LAUNCH(TT, 3);  \
    else COUNTR_XB_LAUNCH(TT, XS);                                                                                             \
  } while (0)
  if (dtype == COUNTR_BF16) COUNTR_XB_DISPATCH(bf16_t); else COUNTR_XB_DISPATCH(float);
#undef COUNTR_XB_DISPATCH
#undef COUNTR_XB_LAUNCH
  hipLaunchKernelGGL(xattn_bwd_finish_kernel, dim3((2 * S * D + 255) / 256, B), dim3(256), 0, STREAM(stream), workspace, dk, dv,
                     (bf16_t*)dk_bf16, (bf16_t*)dv_bf16, bpb, S, D);
  COUNTR_LAUNCH_CHECK("countr_xattn_bwd");
}
