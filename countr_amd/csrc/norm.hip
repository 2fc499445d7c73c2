// Normalisation kernels (HBM-bound; fp32 statistics, 16-byte vector loads, wave-shuffle reductions).
//   LayerNorm fwd/bwd      : nn.LayerNorm(eps=1e-6)          models_mae_cross.py:146,182; models_crossvit.py:153-155
//   GroupNorm(8)+ReLU      : decode_head*                     models_mae_cross.py:80-100   (NHWC maps)
//   InstanceNorm+ReLU+pool : decoder_proj1-4                  models_mae_cross.py:47-71    (NHWC maps)
#include "common.hpp"
#include <stdlib.h>
#include "../../include/countr_hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row; x is the fp32 residual stream, y is the GEMM operand dtype.
// ------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, TO* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * D;
  constexpr int MAXV = 8;  // D <= 2048
  float v[MAXV][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < D) {
      ld4<float>(xr + c, v[i]);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
  if (lane == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
  TO* yr = y + (int64_t)row * D;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < D) {
      float g[4], b[4], o[4];
      ld4<float>(gamma + c, g);
      ld4<float>(beta + c, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      st4<TO>(yr + c, o);
    }
  }
}

// Backward: dx (+)= rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)); per-block partial dgamma / dbeta.
// Grid = NB blocks of LNB_WAVES waves (8: two resident waves per SIMD hide the row-to-row latency; 4 measured 15 vs ~10 us);
// wave w of block b walks rows (b*LNB_WAVES+w), +LNB_WAVES*NB, ... TWO rows per trip (round 4): the loads of both are in flight before
// the first wave reduction, so 4608 rows on 2048 waves are 1.25 memory round trips instead of 2.25 (the MAE step has 42 of these
// launches).  MAXV = 4-element column groups per lane (D <= 256 MAXV) is a template parameter: with the arrays sized for D = 2048 a
// wave held ~170 VGPRs whatever D was.
constexpr int LNB_WAVES = 8;
template <typename TI, int MAXV>
__global__ __launch_bounds__(64 * LNB_WAVES) void layernorm_bwd_kernel(const TI* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            bf16_t* __restrict__ dx_bf16, float* __restrict__ partial, int rows,
                                                            int D, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem_ln[];
  float* sm = reinterpret_cast<float*>(smem_ln);  // [LNB_WAVES][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[MAXV][4], db[MAXV][4];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; }
  const int rstride = gridDim.x * LNB_WAVES;
  for (int row0 = blockIdx.x * LNB_WAVES + wave; row0 < rows; row0 += 2 * rstride) {
    float xh[2][MAXV][4], gy[2][MAXV][4], od[2][MAXV][4];
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, rsv[2];
    bool have[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = row0 + r * rstride;
      have[r] = row < rows;            // wave-uniform
      if (!have[r]) continue;
      const float mu = mean[row];
      rsv[r] = rstd[row];
      const TI* dyr = dy + (int64_t)row * D;
      const float* xr = x + (int64_t)row * D;
      const float* dxr = dx + (int64_t)row * D;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
          float d[4], xv[4], g[4];
          ld4<TI>(dyr + c, d);
          ld4<float>(xr + c, xv);
          ld4<float>(gamma + c, g);
          // the residual gradient to accumulate into is requested with the row's other loads, not after the two wave reductions
          if (accumulate) ld4<float>(dxr + c, od[r][i]); else { od[r][i][0] = od[r][i][1] = od[r][i][2] = od[r][i][3] = 0.f; }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[r][i][e] = (xv[e] - mu) * rsv[r];
            gy[r][i][e] = d[e] * g[e];
            s1[r] += gy[r][i][e];
            s2[r] += gy[r][i][e] * xh[r][i][e];
            dg[i][e] += d[e] * xh[r][i][e];
            db[i][e] += d[e];
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (!have[r]) continue;
      const int row = row0 + r * rstride;
      const float m1 = wave_sum(s1[r]) / D, m2 = wave_sum(s2[r]) / D;
      float* dxr = dx + (int64_t)row * D;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = od[r][i][e] + rsv[r] * (gy[r][i][e] - m1 - xh[r][i][e] * m2);
          st4<float>(dxr + c, o);
          if (dx_bf16) st4<bf16_t>(dx_bf16 + (int64_t)row * D + c, o);   // GEMM-operand copy of the updated residual gradient
        }
      }
    }
  }
  // block reduce of the per-wave column sums, then one partial row per block: [block][2][D]
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < D) {
      st4<float>(sm + (wave * 2 + 0) * D + c, dg[i]);
      st4<float>(sm + (wave * 2 + 1) * D + c, db[i]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 64 * LNB_WAVES) {
    const int which = c / D, col = c - which * D;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_WAVES; ++w) s += sm[(w * 2 + which) * D + col];
    partial[((int64_t)blockIdx.x * 2 + which) * D + col] = s;
  }
}

// out[c] (+)= sum_p partial[p*stride + c]; block = 16 columns x 16 part lanes (short dependent chains: these
// finishers are latency-bound, not bandwidth-bound)
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int nparts,
                                                              int C, int64_t stride, int accumulate) {
  __shared__ float sm[256];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < C)
    for (int p = pl; p < nparts; p += 16) s += partial[(int64_t)p * stride + c];
  sm[threadIdx.x] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k * 16 + cl];
    out[c] = accumulate ? out[c] + s : s;
  }
}

// partial sums of n floats: part[blockIdx.x] = sum of a strided slice (finish with colsum_partials, C = 1)
__global__ __launch_bounds__(256) void sum_all_kernel(const float* __restrict__ in, int64_t n, float* __restrict__ part) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += in[i];
  s = block_sum<4>(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ------------------------------------------------------------------------------------------
// GroupNorm(G groups) on NHWC [B, HW, C], C = 256, 8 channels per thread (32 threads per pixel,
// 8 pixels per pass).  Statistics are accumulated per (image, pixel-split) block as {mean, M2} and combined once by
// gn_stats_finalize_kernel (exact two-pass combination, fixed order: deterministic).
// ------------------------------------------------------------------------------------------
constexpr int GN_C = 256;

// reduce `v` over the lanes/waves that share (threadIdx.x & 31): result valid in every thread.
__device__ __forceinline__ float reduce_same_chanvec(float v, float* sm /* [8][32] */) {
  v = xor32_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane < 32) sm[wave * 32 + lane] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) r += sm[w * 32 + (threadIdx.x & 31)];
  return r;
}

// the same for NV values at once: two barriers in total instead of two per value (a block's reduction tail was 32-48 barriers)
template <int NV, int NW = 4>
__device__ __forceinline__ void reduce_vec_same_chanvec(float (&v)[NV], float* sm /* [NW][32][NV] */) {
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = xor32_sum(v[i]);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane < 32) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sm[(wave * 32 + lane) * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += sm[(w * 32 + (threadIdx.x & 31)) * NV + i];
    v[i] = r;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ part /* [B][ns][G][2] */,
                                                       int HW, int G) {
  __shared__ float sm[128];
  const int b = blockIdx.y, split = blockIdx.x, ns = gridDim.x;
  const int per = (HW + ns - 1) / ns;
  const int p0 = split * per, p1 = min(HW, p0 + per);
  const int cv = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const T* xb = x + (int64_t)b * HW * GN_C;
  float s = 0.f, q = 0.f;
  constexpr int U = 4;
  for (int pb = p0 + slot; pb < p1; pb += 8 * U) {
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (pb + 8 * u < p1) ld8<T>(xb + (int64_t)(pb + 8 * u) * GN_C + cv * 8, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (pb + 8 * u >= p1) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s += v[u][e]; q += v[u][e] * v[u][e]; }
    }
  }
  s = reduce_same_chanvec(s, sm);
  q = reduce_same_chanvec(q, sm);
  // channel-vector cv belongs to group cv / (cpg/8); sum the cvs of a group
  const int cvpg = (GN_C / G) / 8;
  __syncthreads();
  if (threadIdx.x < 32) { sm[threadIdx.x] = s; sm[32 + threadIdx.x] = q; }
  __syncthreads();
  if (threadIdx.x < G) {
    float gs = 0.f, gq = 0.f;
    for (int i = 0; i < cvpg; ++i) { gs += sm[threadIdx.x * cvpg + i]; gq += sm[32 + threadIdx.x * cvpg + i]; }
    const float n = (float)(p1 - p0) * (GN_C / G);
    const float m = n > 0 ? gs / n : 0.f;
    float* o = part + (((int64_t)b * ns + split) * G + threadIdx.x) * 2;
    o[0] = m;
    o[1] = fmaxf(gq - gs * m, 0.f);  // M2 about the split mean
  }
}

// 16-bit maps, 8 groups of 32 channels: the statistics pass over the MAP with exactly the association of the route through a
// convolution's epilogue -- per pixel and group the quad tree of countr_gn_quad_sums, per split 32 pixel-slot accumulators walked in
// pixel order, the slots summed in order (gn_stats_rows_kernel below) -- so that a GroupNorm's statistics do not depend on which of the
// two routes a batch size selects (the lean convolution kernels serve maps of more than 256 tiles: 48 x 48 at B = 8, not at B = 1).
// thread = (channel vector cv, pixel slot ps of 8); a thread carries the four slots ps, ps + 8, ps + 16, ps + 24.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_tree_kernel(const T* __restrict__ x, float* __restrict__ part /* [B][ns][G][2] */, int HW) {
  constexpr int G = 8;
  static_assert(sizeof(T) == 2, "16-bit maps");
  __shared__ float sm[2][32][G];
  const int b = blockIdx.y, split = blockIdx.x, ns = gridDim.x;
  const int per = (HW + ns - 1) / ns;
  const int p0 = split * per, p1 = min(HW, p0 + per);
  const int cv = threadIdx.x & 31, ps = threadIdx.x >> 5;
  const T* xb = x + (int64_t)b * HW * GN_C + cv * 8;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  for (int pb = p0 + ps; pb < p1; pb += 32) {
    countr_u32x4_t v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (pb + 8 * r < p1) v[r] = *reinterpret_cast<const countr_u32x4_t*>(xb + (int64_t)(pb + 8 * r) * GN_C);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (pb + 8 * r >= p1) continue;      // (the same for the four lanes of a quad: they hold one pixel)
      float s1, s2;
      countr_gn_quad_sums(v[r], s1, s2);
      s[r] += s1;
      q[r] += s2;
    }
  }
  if ((cv & 3) == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { sm[0][ps + 8 * r][cv >> 2] = s[r]; sm[1][ps + 8 * r][cv >> 2] = q[r]; }
  }
  __syncthreads();
  if (threadIdx.x < G) {
    float gs = 0.f, gq = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { gs += sm[0][i][threadIdx.x]; gq += sm[1][i][threadIdx.x]; }
    const float n = (float)(p1 - p0) * (GN_C / G);
    const float m = n > 0 ? gs / n : 0.f;
    float* o = part + (((int64_t)b * ns + split) * G + threadIdx.x) * 2;
    o[0] = m;
    o[1] = fmaxf(gq - gs * m, 0.f);  // M2 about the split mean
  }
}

// The same split partials from the ROW partials a convolution's epilogue left (countr_gemm_args.gn_rows: [B * HW][G][2] = {sum, sum of
// squares} of each pixel's 32-channel groups, G = 8): 64 bytes per pixel instead of the 512-byte pixel.  thread = (group, pixel slot),
// 32 slots; a wave reads 8 consecutive pixels = 512 contiguous bytes per load.
__global__ __launch_bounds__(256) void gn_stats_rows_kernel(const float* __restrict__ rows, float* __restrict__ part /* [B][ns][G][2] */,
                                                            int HW) {
  constexpr int G = 8;
  __shared__ float sm[2][32][G];
  const int b = blockIdx.y, split = blockIdx.x, ns = gridDim.x;
  const int per = (HW + ns - 1) / ns;
  const int p0 = split * per, p1 = min(HW, p0 + per);
  const int g = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const float2* rb = reinterpret_cast<const float2*>(rows) + (int64_t)b * HW * G + g;
  float s = 0.f, q = 0.f;
  constexpr int U = 4;
  for (int pb = p0 + slot; pb < p1; pb += 32 * U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (pb + 32 * u < p1) ? rb[(int64_t)(pb + 32 * u) * G] : make_float2(0.f, 0.f);
#pragma unroll
    for (int u = 0; u < U; ++u) { s += v[u].x; q += v[u].y; }
  }
  sm[0][slot][g] = s;
  sm[1][slot][g] = q;
  __syncthreads();
  if (threadIdx.x < G) {
    float gs = 0.f, gq = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { gs += sm[0][i][threadIdx.x]; gq += sm[1][i][threadIdx.x]; }
    const float n = (float)(p1 - p0) * (GN_C / G);
    const float m = n > 0 ? gs / n : 0.f;
    float* o = part + (((int64_t)b * ns + split) * G + threadIdx.x) * 2;
    o[0] = m;
    o[1] = fmaxf(gq - gs * m, 0.f);  // M2 about the split mean
  }
}

// Per-(image, group) mean / rstd from the split partials {mean_s, M2_s}: one wave per group, lanes over the splits, the exact
// two-pass combination  mean = sum n_s mean_s / N,  M2 = sum (M2_s + n_s (mean_s - mean)^2)  (all loads in flight at once).
__global__ void gn_stats_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats /* [B][G][2] */, int HW, int G,
                                         int ns, float eps) {
  const int b = blockIdx.x, g = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per = (HW + ns - 1) / ns;
  const float cpg = (float)(GN_C / G);
  float nsum = 0.f, msum = 0.f;
  for (int s = lane; s < ns; s += 64) {
    const float n = (float)max(min(HW, (s + 1) * per) - s * per, 0) * cpg;
    if (n > 0.f) {   // an empty split's partial is never written
      nsum += n;
      msum += n * part[(((int64_t)b * ns + s) * G + g) * 2];
    }
  }
  const float ntot = wave_sum(nsum);
  const float mean = wave_sum(msum) / ntot;
  float m2 = 0.f;
  for (int s = lane; s < ns; s += 64) {
    const float n = (float)max(min(HW, (s + 1) * per) - s * per, 0) * cpg;
    if (n > 0.f) {
      const float* o = part + (((int64_t)b * ns + s) * G + g) * 2;
      const float d = o[0] - mean;
      m2 += o[1] + n * d * d;
    }
  }
  m2 = wave_sum(m2);
  if (lane == 0) {
    stats[((int64_t)b * G + g) * 2] = mean;
    stats[((int64_t)b * G + g) * 2 + 1] = rsqrtf(m2 / ntot + eps);
  }
}

// y = relu(gn(x)); with w1 != nullptr instead writes out1[b, p] = sum_c y[p, c] * w1[c] + b1 (the fused
// 1x1 conv of decode_head3, models_mae_cross.py:99) and y may be nullptr.
template <typename T>
__global__ __launch_bounds__(256) void gn_relu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ part,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          T* __restrict__ y, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, float* __restrict__ out1,
                                                          float* __restrict__ stats_out /* [B][G][2] */, int HW, int G,
                                                          int ns, float eps) {
  // statistics come finished from gn_stats_finalize_kernel: walking the ns split partials (Chan combine) in EVERY block of this
  // kernel was a chain of ns dependent L2 round trips, 16 us at ns = 64 -- half the kernel's time on the 192 x 192 map
  const int b = blockIdx.y;
  const int cv = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const int g = (cv * 8) / (GN_C / G);
  const float mu = stats_out[((int64_t)b * G + g) * 2], rs = stats_out[((int64_t)b * G + g) * 2 + 1];
  float ga[8], be[8], w[8];
  ld8<float>(gamma + cv * 8, ga);
  ld8<float>(beta + cv * 8, be);
  if (w1) ld8<float>(w1 + cv * 8, w);
  const float bias1 = w1 ? b1[0] : 0.f;
  const T* xb = x + (int64_t)b * HW * GN_C;
  constexpr int U = 4;
  const int stride = gridDim.x * 8;
  for (int p0 = blockIdx.x * 8 + slot; p0 < HW; p0 += stride * U) {
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (p0 + u * stride < HW) ld8<T>(xb + (int64_t)(p0 + u * stride) * GN_C + cv * 8, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * stride;
      float dot = 0.f;
      const bool ok = p < HW;   // keep the shuffles below wave-uniform
      if (ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[u][e] = fmaxf((v[u][e] - mu) * rs * ga[e] + be[e], 0.f);
          if (w1) dot += v[u][e] * w[e];
        }
        if (y) st8<T>(y + ((int64_t)b * HW + p) * GN_C + cv * 8, v[u]);
      }
      if (w1) {
        dot = half_wave_sum(dot);
        if (ok && cv == 0) out1[(int64_t)b * HW + p] = dot + bias1;
      }
    }
  }
}

// Backward pass 1: per-channel sums of g = dy * 1[y>0] and g * xhat over a pixel split.
// dy is either a tensor (T) or, for the fused 1x1 head, d1[b,p] * w1[c].
// partial layout: [B][ns][3][C] = {sum g, sum g*xhat, sum d1*y (dw1, head only)}
// HEAD (the fused 1x1 head, dy = d1[b,p] * w1[c]): with m = d1 * 1[y>0] every sum is a combination of TWO per-channel sums,
// S1 = sum m and S2 = sum m * xhat -- sum g = w1 S1, sum g * xhat = w1 S2, sum d1 * y = gamma S2 + beta S1 (y = xhat gamma + beta where
// the mask is set) -- 7 VALU operations per element instead of 11 on the 151-MB map of the 192 x 192 stage (the pass is as much
// VALU- as bandwidth-bound: 9.4 M (pixel, 8-channel) items of ~90 instructions); the mask is computed exactly as the apply pass does.
template <typename T, int NT = 256, bool HEAD = false>
__global__ __launch_bounds__(NT) void gn_relu_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                 const float* __restrict__ d1, const float* __restrict__ w1,
                                                                 const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ partial,
                                                                 int HW, int G) {
  constexpr int NW = NT / 64, SLOTS = NT / 32;
  __shared__ float smv[NW * 32 * 8];
  const int b = blockIdx.y, split = blockIdx.x, ns = gridDim.x;
  const int per = (HW + ns - 1) / ns;
  const int p0 = split * per, p1 = min(HW, p0 + per);
  const int cv = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const int g = (cv * 8) / (GN_C / G);
  const float mu = stats[((int64_t)b * G + g) * 2], rs = stats[((int64_t)b * G + g) * 2 + 1];
  float ga[8], be[8], w[8];
  ld8<float>(gamma + cv * 8, ga);
  ld8<float>(beta + cv * 8, be);
  if (w1) ld8<float>(w1 + cv * 8, w);
  float sg[8], sgx[8], sw[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sg[e] = 0.f; sgx[e] = 0.f; sw[e] = 0.f; }
  const int64_t base = (int64_t)b * HW * GN_C;
  constexpr int U = 4;
  for (int pb = p0 + slot; pb < p1; pb += SLOTS * U) {
    float v[U][8], d[U][8], dd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = pb + SLOTS * u;
      dd[u] = 0.f;
      if (p < p1) {
        ld8<T>(x + base + (int64_t)p * GN_C + cv * 8, v[u]);
        if (w1) dd[u] = d1[(int64_t)b * HW + p];
        else ld8<T>(dy + base + (int64_t)p * GN_C + cv * 8, d[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (pb + SLOTS * u >= p1) continue;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[u][e] - mu) * rs;
        const float yv = xh * ga[e] + be[e];
        if constexpr (HEAD) {
          const float md = (yv > 0.f) ? dd[u] : 0.f;
          sg[e] += md;             // S1
          sgx[e] += md * xh;       // S2
        } else {
          const float gg = (yv > 0.f) ? (w1 ? dd[u] * w[e] : d[u][e]) : 0.f;
          sg[e] += gg;
          sgx[e] += gg * xh;
          if (w1) sw[e] += dd[u] * fmaxf(yv, 0.f);
        }
      }
    }
  }
  float* o = partial + ((int64_t)b * ns + split) * 3 * GN_C;
  reduce_vec_same_chanvec<8, NW>(sg, smv);
  reduce_vec_same_chanvec<8, NW>(sgx, smv);
  if (!HEAD && w1) reduce_vec_same_chanvec<8, NW>(sw, smv);
  if (threadIdx.x < 32) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (HEAD) {
        o[cv * 8 + e] = w[e] * sg[e];
        o[GN_C + cv * 8 + e] = w[e] * sgx[e];
        o[2 * GN_C + cv * 8 + e] = ga[e] * sgx[e] + be[e] * sg[e];
      } else {
        o[cv * 8 + e] = sg[e];
        o[GN_C + cv * 8 + e] = sgx[e];
        o[2 * GN_C + cv * 8 + e] = w1 ? sw[e] : 0.f;
      }
    }
  }
}

// Backward pass 1b (one block per image): per-(image, group) means of g*gamma and g*gamma*xhat from the split partials.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                              float* __restrict__ gm /* [B][G][2] */,
                                                              float* __restrict__ per_image /* [B][3][C]: the image's split sums */,
                                                              int HW, int G, int ns) {
  const int b = blockIdx.x, c = threadIdx.x;  // 256 channels
  float a4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f}, w4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent chains: the loop is load-latency bound
  int s = 0;
  for (; s + 4 <= ns; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* o = partial + ((int64_t)b * ns + s + u) * 3 * GN_C;
      a4[u] += o[c];
      q4[u] += o[GN_C + c];
      w4[u] += o[2 * GN_C + c];
    }
  }
  for (; s < ns; ++s) {
    const float* o = partial + ((int64_t)b * ns + s) * 3 * GN_C;
    a4[0] += o[c];
    q4[0] += o[GN_C + c];
    w4[0] += o[2 * GN_C + c];
  }
  float a = (a4[0] + a4[1]) + (a4[2] + a4[3]), q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
  // the parameter gradients {dbeta, dgamma, dw1} are sums over ALL images and splits: hand the per-image sums on, so that whoever
  // finishes them adds B rows instead of walking B * ns partials (192x192: 512) in one serial chain per channel
  per_image[((int64_t)b * 3 + 0) * GN_C + c] = a;
  per_image[((int64_t)b * 3 + 1) * GN_C + c] = q;
  per_image[((int64_t)b * 3 + 2) * GN_C + c] = (w4[0] + w4[1]) + (w4[2] + w4[3]);
  const float ga = gamma[c];
  a *= ga; q *= ga;
  const int cpg = GN_C / G;  // 32 channels per group: reduce inside each 32-lane half wave
  if (cpg == 32) { a = half_wave_sum(a); q = half_wave_sum(q); }   // G = 8
  else for (int off = cpg >> 1; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); q += __shfl_xor(q, off, 64); }
  if ((c % cpg) == 0) {
    const float n = (float)HW * cpg;
    gm[((int64_t)b * G + c / cpg) * 2] = a / n;
    gm[((int64_t)b * G + c / cpg) * 2 + 1] = q / n;
  }
}

// Backward pass 2: dx = rstd * (g*gamma - m1 - xhat*m2), m1/m2 = group means of g*gamma, g*gamma*xhat.
template <typename T>
__global__ __launch_bounds__(256) void gn_relu_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                const float* __restrict__ d1, const float* __restrict__ w1,
                                                                const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ gm,
                                                                T* __restrict__ dx, int HW, int G, int ns) {
  const int b = blockIdx.y;
  const int cv = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const int g = (cv * 8) / (GN_C / G);
  const float mu = stats[((int64_t)b * G + g) * 2], rs = stats[((int64_t)b * G + g) * 2 + 1];
  const float m1 = gm[((int64_t)b * G + g) * 2], m2 = gm[((int64_t)b * G + g) * 2 + 1];  // from gn_bwd_finalize_kernel
  float ga[8], be[8], w[8];
  ld8<float>(gamma + cv * 8, ga);
  ld8<float>(beta + cv * 8, be);
  if (w1) ld8<float>(w1 + cv * 8, w);
  const int64_t base = (int64_t)b * HW * GN_C;
  constexpr int U = 4;  // pixels in flight per thread (memory-level parallelism; the kernel is HBM-bound)
  const int stride = gridDim.x * 8;
  for (int p0 = blockIdx.x * 8 + slot; p0 < HW; p0 += stride * U) {
    float v[U][8], d[U][8], dd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * stride;
      dd[u] = 0.f;
      if (p < HW) {
        ld8<T>(x + base + (int64_t)p * GN_C + cv * 8, v[u]);
        if (w1) dd[u] = d1[(int64_t)b * HW + p];
        else ld8<T>(dy + base + (int64_t)p * GN_C + cv * 8, d[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * stride;
      if (p >= HW) continue;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[u][e] - mu) * rs;
        const float yv = xh * ga[e] + be[e];
        const float gg = (yv > 0.f) ? (w1 ? dd[u] * w[e] : d[u][e]) : 0.f;
        o[e] = rs * (gg * ga[e] - m1 - xh * m2);
      }
      st8<T>(dx + base + (int64_t)p * GN_C + cv * 8, o);
    }
  }
}

// ------------------------------------------------------------------------------------------
// InstanceNorm2d (affine=False, eps, biased var) + ReLU + MaxPool2 (or global average pool) on NHWC
// [S, H, W, C].  One block per (sample, 64-channel chunk): thread = (8-channel vector, pixel slot).
// ------------------------------------------------------------------------------------------
// sums over the threads of a 256-thread block that share (threadIdx.x % NCV), for 8 values at once (two barriers in total);
// every thread gets its group's sums
template <int NCV>
__device__ __forceinline__ void reduce8_same_cv(float (&v)[8], float* sm /* [4][NCV][8] */) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = stride_sum<NCV>(v[i]);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane < NCV) {
#pragma unroll
    for (int i = 0; i < 8; ++i) sm[(wave * NCV + lane) * 8 + i] = v[i];
  }
  __syncthreads();
  const int c = threadIdx.x % NCV;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = sm[c * 8 + i] + sm[(NCV + c) * 8 + i] + sm[(2 * NCV + c) * 8 + i] + sm[(3 * NCV + c) * 8 + i];
}

// x-hat storage (xh_out / x_is_xhat): the forward can rewrite the map as the NORMALISED activation, rounded to T, and compute the pooled
// output from those rounded values; the backward then reads x-hat itself.  In bf16 that matters: x-hat recomputed from a rounded x
// carries a relative error of (|mean| / sigma + |x-hat|) * 2^-9 -- several per cent on the channels of a 64-pixel map whose mean is a
// few sigma -- which the two reductions of the InstanceNorm backward turn into a 0.975 cosine of dx against fp32 (the whole exemplar CNN's
// weight gradients: 0.96, tools/diag_exemplar_bf16.py); a stored x-hat is good to |x-hat| * 2^-9.
template <typename T> __device__ __forceinline__ float round_as(float v);
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_as<bf16_t>(float v) { return bf2f(f2bf(v)); }

// NCV = 8-channel vectors per block: 8 (64 channels, 32 pixel slots) or 1 (8 channels, 256 pixel slots: 8x the blocks for the
// 64- and 128-channel layers, whose (C/64, S) grid leaves most CUs idle)
template <typename T, int NCV, typename TX = T>   // TX: dtype of the map x (fp32 conv outputs in front of bf16 activations: see x-hat above)
__global__ __launch_bounds__(256) void in_relu_pool_fwd_kernel(const TX* x, T* __restrict__ y,
                                                               float* __restrict__ stats /* [S][C][2] */, int H, int W, int C,
                                                               int avgpool, float eps, T* xh_out /* may alias x */) {
  __shared__ float sm[4 * 8 * 8];
  const int s = blockIdx.y, c0 = blockIdx.x * (NCV * 8);
  constexpr int NSLOT = 256 / NCV;
  const int cv = threadIdx.x % NCV, slot = threadIdx.x / NCV;
  const int HW = H * W;
  const TX* xs = x + (int64_t)s * HW * C + c0 + cv * 8;
  T* xho = xh_out ? xh_out + (int64_t)s * HW * C + c0 + cv * 8 : nullptr;
  float sum[8], sq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sum[e] = 0.f; sq[e] = 0.f; }
  for (int p = slot; p < HW; p += NSLOT) {
    float v[8];
    ld8<TX>(xs + (int64_t)p * C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum[e] += v[e];
  }
  float mean[8], rstd[8];
  reduce8_same_cv<NCV>(sum, sm);
#pragma unroll
  for (int e = 0; e < 8; ++e) mean[e] = sum[e] / HW;
  for (int p = slot; p < HW; p += NSLOT) {
    float v[8];
    ld8<TX>(xs + (int64_t)p * C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean[e]; sq[e] += d * d; }
  }
  reduce8_same_cv<NCV>(sq, sm);
#pragma unroll
  for (int e = 0; e < 8; ++e) rstd[e] = rsqrtf(sq[e] / HW + eps);
  if (slot == 0 && stats) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      stats[((int64_t)s * C + c0 + cv * 8 + e) * 2] = mean[e];
      stats[((int64_t)s * C + c0 + cv * 8 + e) * 2 + 1] = rstd[e];
    }
  }
  if (!avgpool) {
    const int Ho = H / 2, Wo = W / 2;
    T* ys = y + (int64_t)s * Ho * Wo * C + c0 + cv * 8;
    for (int po = slot; po < Ho * Wo; po += NSLOT) {
      const int oy = po / Wo, ox = po - oy * Wo;
      float m[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = 0.f;  // relu floor
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[8];
        const int64_t off = (int64_t)((2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C;
        ld8<TX>(xs + off, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean[e]) * rstd[e];
        if (xho) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = round_as<T>(v[e]);
          st8<T>(xho + off, v);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
      }
      st8<T>(ys + (int64_t)po * C, m);
    }
  } else {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int p = slot; p < HW; p += NSLOT) {
      float v[8];
      ld8<TX>(xs + (int64_t)p * C, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean[e]) * rstd[e];
      if (xho) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = round_as<T>(v[e]);
        st8<T>(xho + (int64_t)p * C, v);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += fmaxf(v[e], 0.f);
    }
    float o[8];
    reduce8_same_cv<NCV>(acc, sm);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = acc[e] / HW;
    if (slot == 0) st8<T>(y + (int64_t)s * C + c0 + cv * 8, o);
  }
}

// Backward of the same block: dyp is the gradient of the pooled output ([S,H/2,W/2,C] or [S,C]).
// g = routed gradient (first max of the 2x2 window as in torch, or dy/HW for the average pool),
// gated by relu; dx = rstd * (g - mean(g) - xhat * mean(g * xhat)).
template <typename T, int NCV>
__global__ __launch_bounds__(256) void in_relu_pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dyp,
                                                               const float* __restrict__ stats, T* __restrict__ dx, int H,
                                                               int W, int C, int avgpool, int x_is_xhat) {
  __shared__ float sm[4 * 8 * 8];
  const int s = blockIdx.y, c0 = blockIdx.x * (NCV * 8);
  constexpr int NSLOT = 256 / NCV;
  const int cv = threadIdx.x % NCV, slot = threadIdx.x / NCV;
  const int HW = H * W, Ho = H / 2, Wo = W / 2;
  const T* xs = x + (int64_t)s * HW * C + c0 + cv * 8;
  T* dxs = dx + (int64_t)s * HW * C + c0 + cv * 8;
  float mean[8], rstd[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mean[e] = stats[((int64_t)s * C + c0 + cv * 8 + e) * 2];
    rstd[e] = stats[((int64_t)s * C + c0 + cv * 8 + e) * 2 + 1];
  }
  float xm[8], xr[8];      // x-hat = (v - xm) * xr: identity when the forward stored x-hat
#pragma unroll
  for (int e = 0; e < 8; ++e) { xm[e] = x_is_xhat ? 0.f : mean[e]; xr[e] = x_is_xhat ? 1.f : rstd[e]; }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  float davg[8];
  if (avgpool) {
    ld8<T>(dyp + (int64_t)s * C + c0 + cv * 8, davg);
#pragma unroll
    for (int e = 0; e < 8; ++e) davg[e] /= HW;
  }
  // pass 1: sums of g and g*xhat.  Work is organised per pooled window (4 input pixels).
  const int nwin = avgpool ? HW : Ho * Wo;
  for (int po = slot; po < nwin; po += NSLOT) {
    if (avgpool) {
      float v[8];
      ld8<T>(xs + (int64_t)po * C, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[e] - xm[e]) * xr[e];
        const float g = xh > 0.f ? davg[e] : 0.f;
        s1[e] += g; s2[e] += g * xh;
      }
    } else {
      const int oy = po / Wo, ox = po - oy * Wo;
      float d[8], best[8];
      ld8<T>(dyp + ((int64_t)s * Ho * Wo + po) * C + c0 + cv * 8, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[8];
        ld8<T>(xs + (int64_t)((2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], v[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (best[e] - xm[e]) * xr[e];
        const float g = xh > 0.f ? d[e] : 0.f;
        s1[e] += g; s2[e] += g * xh;
      }
    }
  }
  reduce8_same_cv<NCV>(s1, sm);
  reduce8_same_cv<NCV>(s2, sm);
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] /= HW; s2[e] /= HW; }
  // pass 2: write dx
  for (int po = slot; po < nwin; po += NSLOT) {
    if (avgpool) {
      float v[8], o[8];
      ld8<T>(xs + (int64_t)po * C, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[e] - xm[e]) * xr[e];
        const float g = xh > 0.f ? davg[e] : 0.f;
        o[e] = rstd[e] * (g - s1[e] - xh * s2[e]);
      }
      st8<T>(dxs + (int64_t)po * C, o);
    } else {
      const int oy = po / Wo, ox = po - oy * Wo;
      float d[8], v[4][8];
      int arg[8];
      float best[8];
      ld8<T>(dyp + ((int64_t)s * Ho * Wo + po) * C + c0 + cv * 8, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; arg[e] = 0; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ld8<T>(xs + (int64_t)((2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C, v[q]);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (v[q][e] > best[e]) { best[e] = v[q][e]; arg[e] = q; }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (v[q][e] - xm[e]) * xr[e];
          const float g = (arg[e] == q && xh > 0.f) ? d[e] : 0.f;
          o[e] = rstd[e] * (g - s1[e] - xh * s2[e]);
        }
        st8<T>(dxs + (int64_t)((2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C, o);
      }
    }
  }
}


// ---- pixel-band variants for the wide maps of the first exemplar layers ((C/64) x S blocks leave most CUs idle and one CU cannot
// stream a 0.5-MB sample fast enough): the map is cut into NS bands of pooled rows; band statistics are combined with Chan's
// formula (forward) or plain sums (backward) by every block of the second kernel.  partial layout: [S][C/64][NS][2][64] floats.
template <typename T>
__global__ __launch_bounds__(256) void in_stats_split_kernel(const T* __restrict__ x, float* __restrict__ partial, int H, int W, int C) {
  __shared__ float sm[4 * 8 * 8];
  const int s = blockIdx.y, cb = blockIdx.x, split = blockIdx.z, NS = gridDim.z;
  const int cv = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int rows = H / NS, p0 = split * rows * W, n = rows * W;
  const T* xs = x + ((int64_t)s * H * W + p0) * C + cb * 64 + cv * 8;
  float sum[8], sq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sum[e] = 0.f; sq[e] = 0.f; }
  for (int p = slot; p < n; p += 32) {
    float v[8];
    ld8<T>(xs + (int64_t)p * C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum[e] += v[e];
  }
  reduce8_same_cv<8>(sum, sm);
#pragma unroll
  for (int e = 0; e < 8; ++e) sum[e] /= n;   // band mean
  for (int p = slot; p < n; p += 32) {
    float v[8];
    ld8<T>(xs + (int64_t)p * C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - sum[e]; sq[e] += d * d; }
  }
  reduce8_same_cv<8>(sq, sm);
  if (slot == 0) {
    float* o = partial + ((((int64_t)s * gridDim.x + cb) * NS + split) * 2) * 64 + cv * 8;
    st8<float>(o, sum);
    st8<float>(o + 64, sq);
  }
}

template <typename T, typename TX = T>
__global__ __launch_bounds__(256) void in_apply_split_kernel(const TX* x, const float* __restrict__ partial, T* __restrict__ y,
                                                             float* __restrict__ stats, int H, int W, int C, float eps, T* xh_out /* may alias x */) {
  const int s = blockIdx.y, cb = blockIdx.x, split = blockIdx.z, NS = gridDim.z;
  const int cv = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int HW = H * W, n = HW / NS;
  const float* pp = partial + (((int64_t)s * gridDim.x + cb) * NS * 2) * 64 + cv * 8;
  float mean[8], rstd[8], m2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { mean[e] = 0.f; m2[e] = 0.f; }
  for (int q = 0; q < NS; ++q) {
    float a[8];
    ld8<float>(pp + q * 128, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) mean[e] += a[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) mean[e] /= NS;
  for (int q = 0; q < NS; ++q) {
    float a[8], b[8];
    ld8<float>(pp + q * 128, a);
    ld8<float>(pp + q * 128 + 64, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = a[e] - mean[e]; m2[e] += b[e] + n * d * d; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) rstd[e] = rsqrtf(m2[e] / HW + eps);
  const int c0 = cb * 64;
  if (split == 0 && slot == 0 && stats) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      stats[((int64_t)s * C + c0 + cv * 8 + e) * 2] = mean[e];
      stats[((int64_t)s * C + c0 + cv * 8 + e) * 2 + 1] = rstd[e];
    }
  }
  const int Ho = H / 2, Wo = W / 2, orow = Ho / NS;
  const TX* xs = x + (int64_t)s * HW * C + c0 + cv * 8;
  T* xho = xh_out ? xh_out + (int64_t)s * HW * C + c0 + cv * 8 : nullptr;
  T* ys = y + (int64_t)s * Ho * Wo * C + c0 + cv * 8;
  for (int po = split * orow * Wo + slot; po < (split + 1) * orow * Wo; po += 32) {
    const int oy = po / Wo, ox = po - oy * Wo;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = 0.f;  // relu floor
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[8];
      const int64_t off = (int64_t)((2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C;
      ld8<TX>(xs + off, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean[e]) * rstd[e];
      if (xho) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = round_as<T>(v[e]);
        st8<T>(xho + off, v);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
    }
    st8<T>(ys + (int64_t)po * C, m);
  }
}

// backward, max-pool case only: APPLY = false -> band sums of g and g * xhat; true -> dx of the band
template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void in_bwd_split_kernel(const T* __restrict__ x, const T* __restrict__ dyp, const float* __restrict__ stats,
                                                           float* __restrict__ partial, T* __restrict__ dx, int H, int W, int C, int x_is_xhat) {
  __shared__ float sm[4 * 8 * 8];
  const int s = blockIdx.y, cb = blockIdx.x, split = blockIdx.z, NS = gridDim.z;
  const int cv = threadIdx.x & 7, slot = threadIdx.x >> 3, c0 = cb * 64;
  const int HW = H * W, Ho = H / 2, Wo = W / 2, orow = Ho / NS;
  const T* xs = x + (int64_t)s * HW * C + c0 + cv * 8;
  T* dxs = dx + (int64_t)s * HW * C + c0 + cv * 8;
  float mean[8], rstd[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mean[e] = stats[((int64_t)s * C + c0 + cv * 8 + e) * 2];
    rstd[e] = stats[((int64_t)s * C + c0 + cv * 8 + e) * 2 + 1];
    s1[e] = 0.f; s2[e] = 0.f;
  }
  float xm[8], xr[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { xm[e] = x_is_xhat ? 0.f : mean[e]; xr[e] = x_is_xhat ? 1.f : rstd[e]; }
  float* pp = partial + (((int64_t)s * gridDim.x + cb) * NS * 2) * 64 + cv * 8;
  if (APPLY) {
    for (int q = 0; q < NS; ++q) {
      float a[8], b[8];
      ld8<float>(pp + q * 128, a);
      ld8<float>(pp + q * 128 + 64, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += a[e]; s2[e] += b[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] /= HW; s2[e] /= HW; }
  }
  for (int po = split * orow * Wo + slot; po < (split + 1) * orow * Wo; po += 32) {
    const int oy = po / Wo, ox = po - oy * Wo;
    float d[8], v[4][8], best[8];
    int arg[8];
    ld8<T>(dyp + ((int64_t)s * Ho * Wo + po) * C + c0 + cv * 8, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; arg[e] = 0; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ld8<T>(xs + (int64_t)((2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C, v[q]);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (v[q][e] > best[e]) { best[e] = v[q][e]; arg[e] = q; }
    }
    if (!APPLY) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (best[e] - xm[e]) * xr[e];
        const float g = xh > 0.f ? d[e] : 0.f;
        s1[e] += g; s2[e] += g * xh;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (v[q][e] - xm[e]) * xr[e];
          const float g = (arg[e] == q && xh > 0.f) ? d[e] : 0.f;
          o[e] = rstd[e] * (g - s1[e] - xh * s2[e]);
        }
        st8<T>(dxs + (int64_t)((2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C, o);
      }
    }
  }
  if (!APPLY) {
    reduce8_same_cv<8>(s1, sm);
    reduce8_same_cv<8>(s2, sm);
    if (slot == 0) {
      st8<float>(pp + split * 128, s1);
      st8<float>(pp + split * 128 + 64, s2);
    }
  }
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int countr_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean,
                                    float* rstd, int rows, int D, float eps, int out_bf16, void* stream) {
  if (!x || !gamma || !beta || !y || D % 4 || D > 2048 || rows <= 0) { countr_set_error("countr_layernorm_fwd: bad args (need D % 4 == 0, D <= 2048)"); return -1; }
  dim3 grid((rows + 3) / 4), block(256);
  if (out_bf16) hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, grid, block, 0, STREAM(stream), x, gamma, beta, (bf16_t*)y, mean, rstd, rows, D, eps);
  else hipLaunchKernelGGL(layernorm_fwd_kernel<float>, grid, block, 0, STREAM(stream), x, gamma, beta, (float*)y, mean, rstd, rows, D, eps);
  COUNTR_LAUNCH_CHECK("countr_layernorm_fwd");
}

extern "C" int countr_layernorm_bwd_nblocks(void) { return 256; }

extern "C" int countr_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                    float* dx, float* dgamma, float* dbeta, float* workspace, int rows, int D,
                                    int dy_bf16, int accumulate_dx, int accumulate_dgb, void* dx_bf16, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !workspace || D % 4 || D > 2048) { countr_set_error("countr_layernorm_bwd: bad args"); return -1; }
  const int nb = countr_layernorm_bwd_nblocks();
  const size_t lds = (size_t)LNB_WAVES * 2 * D * sizeof(float);
#define COUNTR_LNB_LAUNCH(TI, MV)                                                                                                          \
  {                                                                                                                                        \
    static bool attr_set = false;   /* D > 1024 needs more than the default 64 KiB of dynamic LDS */                                       \
    if (!attr_set) {                                                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&layernorm_bwd_kernel<TI, MV>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                LNB_WAVES * 2 * 256 * MV * 4);                                                                             \
      attr_set = true;                                                                                                                     \
    }                                                                                                                                      \
    hipLaunchKernelGGL((layernorm_bwd_kernel<TI, MV>), dim3(nb), dim3(64 * LNB_WAVES), lds, STREAM(stream), (const TI*)dy, x, gamma, mean,  \
                       rstd, dx, (bf16_t*)dx_bf16, workspace, rows, D, accumulate_dx);                                                     \
  }
#define COUNTR_LNB_BY_D(TI)                                                        \
  if (D <= 512) COUNTR_LNB_LAUNCH(TI, 2) else if (D <= 768) COUNTR_LNB_LAUNCH(TI, 3) \
  else if (D <= 1024) COUNTR_LNB_LAUNCH(TI, 4) else COUNTR_LNB_LAUNCH(TI, 8)
  if (dy_bf16) { COUNTR_LNB_BY_D(bf16_t) } else { COUNTR_LNB_BY_D(float) }
#undef COUNTR_LNB_BY_D
#undef COUNTR_LNB_LAUNCH
  // workspace rows are {dgamma[D], dbeta[D]} per block; adjacent outputs (the flat gradient buffer) finish in one launch
  if (dgamma && dbeta == dgamma + D) {
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, STREAM(stream), workspace, dgamma, nb, 2 * D, (int64_t)2 * D, accumulate_dgb);
    COUNTR_LAUNCH_CHECK("countr_layernorm_bwd");
  }
  if (dgamma) hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 15) / 16), dim3(256), 0, STREAM(stream), workspace, dgamma, nb, D, (int64_t)2 * D, accumulate_dgb);
  if (dbeta) hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 15) / 16), dim3(256), 0, STREAM(stream), workspace + D, dbeta, nb, D, (int64_t)2 * D, accumulate_dgb);
  COUNTR_LAUNCH_CHECK("countr_layernorm_bwd");
}

extern "C" int countr_colsum_partials(const float* partial, float* out, int nparts, int C, int accumulate, void* stream) {
  if (!partial || !out) { countr_set_error("countr_colsum_partials: null"); return -1; }
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((C + 15) / 16), dim3(256), 0, STREAM(stream), partial, out, nparts, C, (int64_t)C, accumulate);
  COUNTR_LAUNCH_CHECK("countr_colsum_partials");
}

// pixel splits of the GroupNorm statistics / backward reductions: enough blocks to hide the cold-miss latency of the small
// maps (24x24: 4 splits, 48x48: 16; bwd 38 -> 21 us, 56 -> 35 us) without multiplying the partials of the big ones
// (96x96: 32, 192x192: 64)
static int gn_splits(int HW) {
  constexpr int capa = 32;   // 96x96 backward: 16 splits 69.5 us, 32: 60.3, 64: 65.6
  const int a = HW / 144 < capa ? HW / 144 : capa, b = HW / 576 < 64 ? HW / 576 : 64;
  const int ns = a > b ? a : b;
  return ns < 1 ? 1 : ns;
}
extern "C" int countr_groupnorm_nsplit(int HW) { return gn_splits(HW); }
// float offset, inside countr_groupnorm_relu_bwd's workspace, of the per-image sums [B][3][C] = {sum g, sum g*xhat, sum d1*y} its
// finalize pass leaves behind (the workspace therefore needs B*ns*3*C + 64 + 16*B + B*3*C floats)
extern "C" long long countr_groupnorm_bwd_image_sums_offset(int B, int HW) { return (long long)B * gn_splits(HW) * 3 * GN_C + 64 + 16 * B; }
// forward statistics: their partials ({mean, M2} per group) are combined in parallel by gn_stats_finalize_kernel, so the big maps
// can be cut finer than the backward's (whose finishers walk the splits): 144 pixels per block, at most 128 blocks per image
static int gn_splits_fwd(int HW) {
  constexpr int cap = 128;   // 96x96: 37.9 -> 23.7 us
  const int ns = HW / 144 < cap ? HW / 144 : cap;
  const int lo = gn_splits(HW);
  return ns > lo ? ns : lo;
}

static int groupnorm_relu_fwd(const void* x, const float* rows, const float* gamma, const float* beta, void* y, const float* w1,
                              const float* b1, float* out1, float* stats, float* workspace, int B, int HW, int C, int G, float eps,
                              int dtype, void* stream, const char* who) {
  if (!x || !gamma || !beta || !stats || !workspace || C != GN_C || G > 16 || (GN_C / G) % 8 || (!y && !w1)) { countr_set_error("countr_groupnorm_relu_fwd: bad args (C must be 256, G <= 16)"); return -1; }
  if (rows && (G != 8 || dtype != COUNTR_BF16 || ((uintptr_t)rows & 7))) { countr_set_error("countr_groupnorm_relu_fwd_rows: row partials are those of 8 groups of 32 channels in a 16-bit map"); return -1; }
  const int ns = gn_splits_fwd(HW);
  const int nblk = min((HW + 7) / 8, 512);
  if (dtype == COUNTR_BF16) {
    if (rows) hipLaunchKernelGGL(gn_stats_rows_kernel, dim3(ns, B), dim3(256), 0, STREAM(stream), rows, workspace, HW);
    else if (G == 8) hipLaunchKernelGGL(gn_stats_tree_kernel<bf16_t>, dim3(ns, B), dim3(256), 0, STREAM(stream), (const bf16_t*)x, workspace, HW);
    else hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, dim3(ns, B), dim3(256), 0, STREAM(stream), (const bf16_t*)x, workspace, HW, G);
    hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(B), dim3(64 * G), 0, STREAM(stream), workspace, stats, HW, G, ns, eps);
    hipLaunchKernelGGL(gn_relu_fwd_kernel<bf16_t>, dim3(nblk, B), dim3(256), 0, STREAM(stream), (const bf16_t*)x, workspace, gamma, beta, (bf16_t*)y, w1, b1, out1, stats, HW, G, ns, eps);
  } else {
    hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(ns, B), dim3(256), 0, STREAM(stream), (const float*)x, workspace, HW, G);
    hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(B), dim3(64 * G), 0, STREAM(stream), workspace, stats, HW, G, ns, eps);
    hipLaunchKernelGGL(gn_relu_fwd_kernel<float>, dim3(nblk, B), dim3(256), 0, STREAM(stream), (const float*)x, workspace, gamma, beta, (float*)y, w1, b1, out1, stats, HW, G, ns, eps);
  }
  COUNTR_LAUNCH_CHECK(who);
}

extern "C" int countr_groupnorm_relu_fwd(const void* x, const float* gamma, const float* beta, void* y, const float* w1,
                                         const float* b1, float* out1, float* stats, float* workspace, int B, int HW, int C,
                                         int G, float eps, int dtype, void* stream) {
  return groupnorm_relu_fwd(x, nullptr, gamma, beta, y, w1, b1, out1, stats, workspace, B, HW, C, G, eps, dtype, stream, "countr_groupnorm_relu_fwd");
}

// ... with the statistics pass reading the ROW partials the producing convolution left (countr_gemm_args.gn_rows) instead of the map
extern "C" int countr_groupnorm_relu_fwd_rows(const void* x, const float* rows, const float* gamma, const float* beta, void* y,
                                              const float* w1, const float* b1, float* out1, float* stats, float* workspace, int B,
                                              int HW, int C, int G, float eps, int dtype, void* stream) {
  if (!rows) { countr_set_error("countr_groupnorm_relu_fwd_rows: null row partials"); return -1; }
  return groupnorm_relu_fwd(x, rows, gamma, beta, y, w1, b1, out1, stats, workspace, B, HW, C, G, eps, dtype, stream, "countr_groupnorm_relu_fwd_rows");
}

// workspace: fp32 [B][ns][3][C]; dgamma/dbeta(/dw1) accumulate flag applies to all.
extern "C" int countr_groupnorm_relu_bwd(const void* x, const void* dy, const float* d1, const float* w1, const float* stats,
                                         const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta,
                                         float* dw1, float* db1, float* workspace, int B, int HW, int C, int G, int dtype,
                                         int accumulate, void* stream) {
  if (!x || !stats || !gamma || !beta || !dx || !workspace || C != GN_C || (!dy && !(d1 && w1))) { countr_set_error("countr_groupnorm_relu_bwd: bad args"); return -1; }
  const int ns = gn_splits(HW);
  const int nblk = min((HW + 7) / 8, 512);
  float* gmean = workspace + (int64_t)B * ns * 3 * GN_C + 64;  // [B][G][2] behind the partials and the d1 sums
  float* per_image = gmean + 16 * B;                           // [B][3][C] behind that (countr_groupnorm_bwd_image_sums_offset)
  if (G != 8) { countr_set_error("countr_groupnorm_relu_bwd: G must be 8"); return -1; }
  if (dtype == COUNTR_BF16) {
    // threads per block of the reduction pass: 512 (16 pixel slots) keeps more loads in flight per CU than 256 (96x96: 31.6 -> 20.8 us,
    // 48x48 and 24x24: 17.4 -> 11.6 us); 1024 falls off a cliff (step +230 us: 128-VGPR budget)
    const int nt = w1 ? 256 : 512;   // (the fused-head form -- 192x192, x only -- is better off with 256: 41.4 vs 44.4 us)
    if (nt == 512) hipLaunchKernelGGL((gn_relu_bwd_reduce_kernel<bf16_t, 512>), dim3(ns, B), dim3(512), 0, STREAM(stream), (const bf16_t*)x, (const bf16_t*)dy, d1, w1, stats, gamma, beta, workspace, HW, G);
    else if (d1 && w1) hipLaunchKernelGGL((gn_relu_bwd_reduce_kernel<bf16_t, 256, true>), dim3(ns, B), dim3(256), 0, STREAM(stream), (const bf16_t*)x, (const bf16_t*)dy, d1, w1, stats, gamma, beta, workspace, HW, G);
    else hipLaunchKernelGGL((gn_relu_bwd_reduce_kernel<bf16_t, 256>), dim3(ns, B), dim3(256), 0, STREAM(stream), (const bf16_t*)x, (const bf16_t*)dy, d1, w1, stats, gamma, beta, workspace, HW, G);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), 0, STREAM(stream), workspace, gamma, gmean, per_image, HW, G, ns);
    hipLaunchKernelGGL(gn_relu_bwd_apply_kernel<bf16_t>, dim3(nblk, B), dim3(256), 0, STREAM(stream), (const bf16_t*)x, (const bf16_t*)dy, d1, w1, stats, gamma, beta, gmean, (bf16_t*)dx, HW, G, ns);
  } else {
    hipLaunchKernelGGL((gn_relu_bwd_reduce_kernel<float, 256>), dim3(ns, B), dim3(256), 0, STREAM(stream), (const float*)x, (const float*)dy, d1, w1, stats, gamma, beta, workspace, HW, G);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), 0, STREAM(stream), workspace, gamma, gmean, per_image, HW, G, ns);
    hipLaunchKernelGGL(gn_relu_bwd_apply_kernel<float>, dim3(nblk, B), dim3(256), 0, STREAM(stream), (const float*)x, (const float*)dy, d1, w1, stats, gamma, beta, gmean, (float*)dx, HW, G, ns);
  }
  // parameter gradients: partial[p] = {sum g (dbeta) [C], sum g*xhat (dgamma) [C], sum d1*y (dw1) [C]}
  float* outs[3] = {dbeta, dgamma, dw1};
  for (int i = 0; i < 3; ++i) {
    if (!outs[i]) continue;
    hipLaunchKernelGGL(colsum_partials_kernel, dim3(GN_C / 16), dim3(256), 0, STREAM(stream), per_image + i * GN_C, outs[i], B, GN_C,
                       (int64_t)3 * GN_C, accumulate);
  }
  if (db1 && d1) {  // the reduce/apply kernels are done with the first 64 floats of row 0's third plane only via dw1: use the tail
    float* part = workspace + (int64_t)B * ns * 3 * GN_C;  // 64 extra floats behind the partials
    hipLaunchKernelGGL(sum_all_kernel, dim3(64), dim3(256), 0, STREAM(stream), d1, (int64_t)B * HW, part);
    hipLaunchKernelGGL(colsum_partials_kernel, dim3(1), dim3(256), 0, STREAM(stream), part, db1, 64, 1, (int64_t)1, accumulate);
  }
  COUNTR_LAUNCH_CHECK("countr_groupnorm_relu_bwd");
}

// bands of pooled rows for the split path (0 = one-kernel path): only for the big maps whose (C/64) x S blocks cannot fill the chip
static int in_splits(int S, int H, int C, int avgpool) {
  constexpr int minh = 32;
  if (avgpool || (C / 64) * S >= 128 || H < minh) return 0;
  const int Ho = H / 2;
  int ns = Ho >= 32 ? 16 : 8;
  while (ns > 1 && (Ho % ns)) ns >>= 1;
  return ns > 1 ? ns : 0;
}
extern "C" int countr_instnorm_workspace_floats(int S, int C) { return S * C * 16 * 2; }

extern "C" int countr_instnorm_relu_pool_fwd(const void* x, void* y, float* stats, int S, int H, int W, int C, int avgpool,
                                             float eps, int dtype, float* workspace, void* xhat_out, int x_f32, void* stream) {
  if (!x || !y || C % 64 || (H & 1) || (W & 1)) { countr_set_error("countr_instnorm_relu_pool_fwd: bad args (C % 64, even H/W)"); return -1; }
  if (x_f32 && dtype == COUNTR_BF16 && xhat_out == x) { countr_set_error("countr_instnorm_relu_pool_fwd: a bf16 x-hat cannot be stored in place of an fp32 map"); return -1; }
  const int ns = workspace ? in_splits(S, H, C, avgpool) : 0;
  if (ns) {
    dim3 grid(C / 64, S, ns), block(256);
    if (dtype == COUNTR_BF16 && x_f32) {
      hipLaunchKernelGGL(in_stats_split_kernel<float>, grid, block, 0, STREAM(stream), (const float*)x, workspace, H, W, C);
      hipLaunchKernelGGL((in_apply_split_kernel<bf16_t, float>), grid, block, 0, STREAM(stream), (const float*)x, workspace, (bf16_t*)y, stats, H, W, C, eps, (bf16_t*)xhat_out);
    } else if (dtype == COUNTR_BF16) {
      hipLaunchKernelGGL(in_stats_split_kernel<bf16_t>, grid, block, 0, STREAM(stream), (const bf16_t*)x, workspace, H, W, C);
      hipLaunchKernelGGL(in_apply_split_kernel<bf16_t>, grid, block, 0, STREAM(stream), (const bf16_t*)x, workspace, (bf16_t*)y, stats, H, W, C, eps, (bf16_t*)xhat_out);
    } else {
      hipLaunchKernelGGL(in_stats_split_kernel<float>, grid, block, 0, STREAM(stream), (const float*)x, workspace, H, W, C);
      hipLaunchKernelGGL(in_apply_split_kernel<float>, grid, block, 0, STREAM(stream), (const float*)x, workspace, (float*)y, stats, H, W, C, eps, (float*)xhat_out);
    }
    COUNTR_LAUNCH_CHECK("countr_instnorm_relu_pool_fwd");
  }
  dim3 grid(C / 64, S), block(256);
  if (dtype == COUNTR_BF16 && x_f32) hipLaunchKernelGGL((in_relu_pool_fwd_kernel<bf16_t, 8, float>), grid, block, 0, STREAM(stream), (const float*)x, (bf16_t*)y, stats, H, W, C, avgpool, eps, (bf16_t*)xhat_out);
  else if (dtype == COUNTR_BF16) hipLaunchKernelGGL((in_relu_pool_fwd_kernel<bf16_t, 8>), grid, block, 0, STREAM(stream), (const bf16_t*)x, (bf16_t*)y, stats, H, W, C, avgpool, eps, (bf16_t*)xhat_out);
  else hipLaunchKernelGGL((in_relu_pool_fwd_kernel<float, 8>), grid, block, 0, STREAM(stream), (const float*)x, (float*)y, stats, H, W, C, avgpool, eps, (float*)xhat_out);
  COUNTR_LAUNCH_CHECK("countr_instnorm_relu_pool_fwd");
}

extern "C" int countr_instnorm_relu_pool_bwd(const void* x, const void* dyp, const float* stats, void* dx, int S, int H, int W,
                                             int C, int avgpool, int dtype, float* workspace, int x_is_xhat, void* stream) {
  if (!x || !dyp || !stats || !dx || C % 64) { countr_set_error("countr_instnorm_relu_pool_bwd: bad args"); return -1; }
  const int ns = workspace ? in_splits(S, H, C, avgpool) : 0;
  if (ns) {
    dim3 grid(C / 64, S, ns), block(256);
    if (dtype == COUNTR_BF16) {
      hipLaunchKernelGGL((in_bwd_split_kernel<bf16_t, false>), grid, block, 0, STREAM(stream), (const bf16_t*)x, (const bf16_t*)dyp, stats, workspace, (bf16_t*)dx, H, W, C, x_is_xhat);
      hipLaunchKernelGGL((in_bwd_split_kernel<bf16_t, true>), grid, block, 0, STREAM(stream), (const bf16_t*)x, (const bf16_t*)dyp, stats, workspace, (bf16_t*)dx, H, W, C, x_is_xhat);
    } else {
      hipLaunchKernelGGL((in_bwd_split_kernel<float, false>), grid, block, 0, STREAM(stream), (const float*)x, (const float*)dyp, stats, workspace, (float*)dx, H, W, C, x_is_xhat);
      hipLaunchKernelGGL((in_bwd_split_kernel<float, true>), grid, block, 0, STREAM(stream), (const float*)x, (const float*)dyp, stats, workspace, (float*)dx, H, W, C, x_is_xhat);
    }
    COUNTR_LAUNCH_CHECK("countr_instnorm_relu_pool_bwd");
  }
  dim3 grid(C / 64, S), block(256);
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL((in_relu_pool_bwd_kernel<bf16_t, 8>), grid, block, 0, STREAM(stream), (const bf16_t*)x, (const bf16_t*)dyp, stats, (bf16_t*)dx, H, W, C, avgpool, x_is_xhat);
  else hipLaunchKernelGGL((in_relu_pool_bwd_kernel<float, 8>), grid, block, 0, STREAM(stream), (const float*)x, (const float*)dyp, stats, (float*)dx, H, W, C, avgpool, x_is_xhat);
  COUNTR_LAUNCH_CHECK("countr_instnorm_relu_pool_bwd");
}
