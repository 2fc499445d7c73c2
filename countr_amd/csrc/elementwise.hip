// Data-movement / elementwise kernels of the CounTR hot path (HBM-bound, 16-byte vector accesses).
//   im2patch        : timm PatchEmbed conv k16 s16 gather          (models_mae_cross.py:138)
//   conv3x3_c3      : first exemplar conv 3->64 (direct, VALU)      (models_mae_cross.py:48)
//   upsample2x      : F.interpolate(bilinear, align_corners=False)  (models_mae_cross.py:189-196)
//   gelu_bwd, colsum: autograd pieces of Mlp / Linear bias          (models_crossvit.py:60-67)
//   cast / permute  : weight shadows (OIHW -> OHWI, dgrad form)
//   masked mse loss : FSC_finetune_cross.py:290-303
//   adamw           : torch.optim.AdamW(betas=(0.9,0.95))           (FSC_finetune_cross.py:235)
#include "common.hpp"
#include <stdlib.h>
#include "../../include/countr_hip.h"

extern "C" int countr_colsum_partials(const float* partial, float* out, int nparts, int C, int accumulate, void* stream);

namespace {

// ---------------- patch gather: img fp32 NCHW [B,3,H,W] -> patches [B*gh*gw, 3*p*p], k = (c, py, px)
template <typename T>
__global__ void im2patch_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int p, int gh,
                                int gw) {
  const int K = 3 * p * p;
  const int64_t total = (int64_t)B * gh * gw * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int64_t t = i / K;
    const int j = (int)(t % gw), ii = (int)((t / gw) % gh), b = (int)(t / ((int64_t)gw * gh));
    const int c = k / (p * p), r = k - c * p * p, py = r / p, px = r - py * p;
    stf<T>(out + i, img[(((int64_t)b * 3 + c) * H + ii * p + py) * W + j * p + px]);
  }
}

// patch sizes that are a multiple of 8 (ViT-B/16): 8 consecutive px per thread -- two 16-byte reads, one 16 / 32-byte store, 32-bit
// index arithmetic once per 8 elements (the scalar kernel pays five 64-bit divisions per element: 18 us for the 8 x 384 x 384 batch)
template <typename T>
__global__ __launch_bounds__(256) void im2patch_vec8_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W,
                                                            int p, int gh, int gw) {
  const int p8 = p >> 3, K8 = 3 * p * p8;                // 8-element groups per patch row / per token
  const int total = B * gh * gw * K8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k8 = i % K8, t = i / K8;
    const int j = t % gw, t2 = t / gw, ii = t2 % gh, b = t2 / gh;
    const int c = k8 / (p * p8), r = k8 - c * p * p8, py = r / p8, px = (r - py * p8) << 3;
    const float* src = img + (((int64_t)b * 3 + c) * H + ii * p + py) * W + j * p + px;
    float v[8];
    ld4<float>(src, *reinterpret_cast<float(*)[4]>(v));
    ld4<float>(src + 4, *reinterpret_cast<float(*)[4]>(v + 4));
    st8<T>(out + (int64_t)i * 8, v);
  }
}

// ---------------- first exemplar conv: in fp32 NCHW [S,3,H,W], w fp32 [64,3,3,3], out NHWC [S,H,W,64]
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_c3_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ out, int S,
                                                             int H, int W) {
  __shared__ float sw[27 * 64];  // [k][co]
  __shared__ float sb[64];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) { const int co = i / 27, k = i - co * 27; sw[k * 64 + co] = w[i]; }
  if (threadIdx.x < 64) sb[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int cv = threadIdx.x & 7;  // 8 output channels
  const int64_t npix = (int64_t)S * H * W;
  const int HW = H * W;
  // (the exemplar crops are 64 x 64: shifts instead of the two divisions per pixel where both sizes are powers of two)
  const bool pow2 = ((W & (W - 1)) | (H & (H - 1))) == 0;
  const int lw = 31 - __builtin_clz(W), lhw = 31 - __builtin_clz(HW);
  for (int64_t pix = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (int64_t)gridDim.x * 32) {
    int x, y, s;
    if (pow2) { const int p32 = (int)pix; s = p32 >> lhw; const int r = p32 & (HW - 1); y = r >> lw; x = r & (W - 1); }
    else { x = (int)(pix % W); y = (int)((pix / W) % H); s = (int)(pix / ((int64_t)W * H)); }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = sb[cv * 8 + e];
    // one 64-bit base per pixel, 32-bit offsets per tap (the per-tap 64-bit products were a third of the kernel's instructions);
    // same taps in the same order: identical values
    const float* img = in + (int64_t)s * 3 * HW;
    bool vy[3], vx[3];
    int ro[3], xo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      vy[k] = (unsigned)(y + k - 1) < (unsigned)H;
      vx[k] = (unsigned)(x + k - 1) < (unsigned)W;
      ro[k] = (y + k - 1) * W;
      xo[k] = x + k - 1;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float v = 0.f;
          if (vy[ky] && vx[kx]) v = img[c * HW + ro[ky] + xo[kx]];
          const float* wr = sw + ((c * 3 + ky) * 3 + kx) * 64 + cv * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v * wr[e];
        }
    st8<T>(out + pix * 64 + cv * 8, acc);
  }
}

// The same convolution, FOUR pixels of an image row per thread (W % 4 == 0): a tap's eight weights are read from LDS once for four
// pixels and the 3 x 6 input window of a channel is loaded once (54 loads + 54 LDS reads per four pixels instead of 108 + 216), the
// pixel decode is paid once -- the per-pixel kernel above is bound by exactly these instructions (560 VALU per pixel and channel
// vector for 108 packed FMAs).  Same taps in the same order per output: identical values.  The exemplar lane heads the decoder's
// critical path in the pipelined step, so this launch is on it.
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_c3_fwd4_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ bias, T* __restrict__ out, int S,
                                                              int H, int W) {
  __shared__ float sw[27 * 64];  // [k][co]
  __shared__ float sb[64];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) { const int co = i / 27, k = i - co * 27; sw[k * 64 + co] = w[i]; }
  if (threadIdx.x < 64) sb[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int cv = threadIdx.x & 7;  // 8 output channels
  const int W4 = W >> 2, HW = H * W;
  const int nq = S * H * W4;       // quads of four pixels
  for (int q = blockIdx.x * 32 + (threadIdx.x >> 3); q < nq; q += gridDim.x * 32) {
    const int xq = q % W4, t = q / W4, y = t % H, s = t / H;
    const int x0 = xq * 4;
    float acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[p][e] = sb[cv * 8 + e];
    const float* img = in + (int64_t)s * 3 * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        float v[6];      // columns x0 - 1 .. x0 + 4 of this input row (zero padding outside the image)
        const bool vy = (unsigned)yy < (unsigned)H;
        const float* row = img + c * HW + yy * W + x0;
        const float4 mid = vy ? *reinterpret_cast<const float4*>(row) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[0] = (vy && x0 > 0) ? row[-1] : 0.f;
        v[1] = mid.x; v[2] = mid.y; v[3] = mid.z; v[4] = mid.w;
        v[5] = (vy && x0 + 4 < W) ? row[4] : 0.f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* wr = sw + ((c * 3 + ky) * 3 + kx) * 64 + cv * 8;
          float wv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) wv[e] = wr[e];
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[p][e] += v[p + kx] * wv[e];
        }
      }
    T* o = out + ((int64_t)(s * H + y) * W + x0) * 64 + cv * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) st8<T>(o + p * 64, acc[p]);
  }
}

// wgrad of the same conv: partial[block][64*27] (+ bias grad partial[block][64] behind it).
// Each thread owns 7 of the 64 x 28 outputs (one channel, seven taps) and walks the block's pixel range through LDS tiles.
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_c3_wgrad_kernel(const float* __restrict__ in, const T* __restrict__ dy,
                                                               float* __restrict__ partial, int S, int H, int W) {
  __shared__ float s_dy[64][65];  // [pixel][co]
  __shared__ float s_in[64][28];  // [pixel][k], k = 27 is the constant 1 (bias grad)
  const int64_t npix = (int64_t)S * H * W;
  const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = blockIdx.x * per, p1 = min(npix, p0 + per);
  float acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) acc[i] = 0.f;
  for (int64_t base = p0; base < p1; base += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
      const int pp = i >> 3, c8 = i & 7;
      float v[8];
      if (base + pp < p1) ld8<T>(dy + (base + pp) * 64 + c8 * 8, v);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s_dy[pp][c8 * 8 + e] = v[e];
    }
    for (int i = threadIdx.x; i < 64 * 28; i += 256) {
      const int pp = i / 28, k = i - pp * 28;
      float v = 0.f;
      const int64_t pix = base + pp;
      if (pix < p1) {
        if (k == 27) v = 1.f;
        else {
          const int pi = (int)pix, x = pi % W, y = (pi / W) % H, s = pi / (W * H);   // S * H * W < 2^31 (checked on the host)
          const int c = k / 9, r = k - c * 9, ky = r / 3, kx = r - ky * 3;
          const int yy = y + ky - 1, xx = x + kx - 1;
          if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = in[(((int64_t)s * 3 + c) * H + yy) * W + xx];
        }
      }
      s_in[pp][k] = v;
    }
    __syncthreads();
    // thread = (output channel co, group kq of 7 taps): per pixel one conflict-free read of dy and seven wave-uniform
    // (broadcast) reads of the taps feed seven FMAs (the first version paired unrelated (co, k) per thread: 14 reads per 7 FMAs)
    const int co = threadIdx.x & 63, kq = threadIdx.x >> 6;
    for (int pp = 0; pp < 64; ++pp) {
      const float d = s_dy[pp][co];
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[i] = fmaf(d, s_in[pp][kq * 7 + i], acc[i]);
    }
  }
  {
    const int co = threadIdx.x & 63, kq = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 7; ++i) partial[(int64_t)blockIdx.x * (64 * 28) + co * 28 + kq * 7 + i] = acc[i];
  }
}

// partial [nb][64][28] -> dw [64][27] (torch OIHW order co, c, ky, kx == k) and db [64]; a block sums 16 outputs with 64 slab
// groups of nb / 64 blocks each and combines them in LDS (4 lanes per output walking nb / 4 slabs serially took 21.8 us; 16 groups:
// 10.5 us at 256 slabs, 15.3 at 512)
constexpr int C3F_GROUPS = 64;   // slab groups per block: 1024 threads = 16 outputs x 64 groups
__global__ __launch_bounds__(16 * C3F_GROUPS) void conv3x3_c3_wgrad_finish_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                                  float* __restrict__ db, int nb, int accumulate) {
  __shared__ float red[C3F_GROUPS][17];
  const int c = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + c;
  float s = 0.f;
  if (o < 64 * 28)
    for (int b = gq; b < nb; b += C3F_GROUPS) s += partial[(int64_t)b * (64 * 28) + o];
  red[gq][c] = s;
  __syncthreads();
  if (gq || o >= 64 * 28) return;
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < C3F_GROUPS; ++k) tot += red[k][c];
  const int co = o / 28, k = o - co * 28;
  float* dst = (k == 27) ? (db + co) : (dw + co * 27 + k);
  *dst = accumulate ? *dst + tot : tot;
}

// ---------------- bilinear x2 (align_corners=False) on NHWC, VEC channels per thread
// out[2m] = .25 in[m-1] + .75 in[m], out[2m+1] = .75 in[m] + .25 in[m+1], indices clamped.
template <typename T, int VEC>
__global__ void upsample2x_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C) {
  const int CV = C / VEC;
  const int64_t total = (int64_t)B * 2 * H * 2 * W * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    int64_t t = i / CV;
    const int ox = (int)(t % (2 * W)); t /= (2 * W);
    const int oy = (int)(t % (2 * H));
    const int b = (int)(t / (2 * H));
    const int my = oy >> 1, mx = ox >> 1;
    const int y0 = (oy & 1) ? my : max(my - 1, 0), y1 = (oy & 1) ? min(my + 1, H - 1) : my;
    const int x0 = (ox & 1) ? mx : max(mx - 1, 0), x1 = (ox & 1) ? min(mx + 1, W - 1) : mx;
    const float wy0 = (oy & 1) ? 0.75f : 0.25f, wx0 = (ox & 1) ? 0.75f : 0.25f;
    const T* base = in + (int64_t)b * H * W * C + cv * VEC;
    float a[VEC], bq[VEC], c[VEC], d[VEC], o[VEC];
    if constexpr (VEC == 8) {
      ld8<T>(base + ((int64_t)y0 * W + x0) * C, a); ld8<T>(base + ((int64_t)y0 * W + x1) * C, bq);
      ld8<T>(base + ((int64_t)y1 * W + x0) * C, c); ld8<T>(base + ((int64_t)y1 * W + x1) * C, d);
    } else {
      a[0] = ldf<T>(base + ((int64_t)y0 * W + x0) * C); bq[0] = ldf<T>(base + ((int64_t)y0 * W + x1) * C);
      c[0] = ldf<T>(base + ((int64_t)y1 * W + x0) * C); d[0] = ldf<T>(base + ((int64_t)y1 * W + x1) * C);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float top = wx0 * a[e] + (1.f - wx0) * bq[e];
      const float bot = wx0 * c[e] + (1.f - wx0) * d[e];
      o[e] = wy0 * top + (1.f - wy0) * bot;
    }
    T* dst = out + (((int64_t)b * 2 * H + oy) * 2 * W + ox) * C + cv * VEC;
    if constexpr (VEC == 8) st8<T>(dst, o); else stf<T>(dst, o[0]);
  }
}

// XCD-aware block order of the stencil kernels below (workgroup b runs on XCD b % 8, each XCD has its own L2): with xcd != 0 (the grid is
// then a multiple of 8 blocks that covers the map once) XCD x works on the x-th CONTIGUOUS eighth of the pixels, so that the rows two
// neighbouring output rows share are fetched into ONE L2 -- in launch order the blocks of an image row alternate over all eight, and
// every L2 fetches (nearly) the whole map: 3-4 x the map's bytes from the memory side for the adjoint's 4 x 4 taps.
__device__ __forceinline__ int up2_block(int xcd) {
  const int b = (int)blockIdx.x;
  return xcd ? (b & 7) * ((int)gridDim.x >> 3) + (b >> 3) : b;
}

// Same result, one thread per COARSE pixel and 8 channels: the 3x3 clamped neighbourhood is loaded once (9 loads for 4 outputs
// instead of 16) and interpolated separably in the same order as above (horizontal, then vertical), so the values are identical.
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_fwd_quad_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int xcd) {
  const int CV = C / 8;
  const int64_t total = (int64_t)B * H * W * CV;
  for (int64_t i = (int64_t)up2_block(xcd) * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    int64_t t = i / CV;
    const int mx = (int)(t % W); t /= W;
    const int my = (int)(t % H);
    const int b = (int)(t / H);
    const int xs[3] = {max(mx - 1, 0), mx, min(mx + 1, W - 1)};
    const int ys[3] = {max(my - 1, 0), my, min(my + 1, H - 1)};
    const T* base = in + (int64_t)b * H * W * C + cv * 8;
    float h0[3][8], h1[3][8];   // per input row: the even / odd output column
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float l[8], c[8], rr[8];
      ld8<T>(base + ((int64_t)ys[r] * W + xs[0]) * C, l);
      ld8<T>(base + ((int64_t)ys[r] * W + xs[1]) * C, c);
      ld8<T>(base + ((int64_t)ys[r] * W + xs[2]) * C, rr);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        h0[r][e] = 0.25f * l[e] + (1.f - 0.25f) * c[e];
        h1[r][e] = 0.75f * c[e] + (1.f - 0.75f) * rr[e];
      }
    }
    T* o00 = out + (((int64_t)b * 2 * H + 2 * my) * 2 * W + 2 * mx) * C + cv * 8;
    T* o10 = o00 + (int64_t)2 * W * C;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.25f * h0[0][e] + (1.f - 0.25f) * h0[1][e];
    st8<T>(o00, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.25f * h1[0][e] + (1.f - 0.25f) * h1[1][e];
    st8<T>(o00 + C, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.75f * h0[1][e] + (1.f - 0.75f) * h0[2][e];
    st8<T>(o10, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.75f * h1[1][e] + (1.f - 0.75f) * h1[2][e];
    st8<T>(o10 + C, v);
  }
}

// adjoint: din[m] = sum over the (up to) 4x4 fine pixels that read coarse pixel m.
// 1-D weights of fine index f on coarse m: f=2m-1 -> .25, 2m -> .75, 2m+1 -> .75, 2m+2 -> .25, with the
// clamped borders folding the out-of-range neighbour back (m=0: f=0 gets +.25; m=H-1: f=2H-1 gets +.25).
__device__ __forceinline__ void up2_adj_taps(int m, int n, int (&f)[4], float (&w)[4]) {
  f[0] = 2 * m - 1; w[0] = 0.25f;
  f[1] = 2 * m;     w[1] = 0.75f;
  f[2] = 2 * m + 1; w[2] = 0.75f;
  f[3] = 2 * m + 2; w[3] = 0.25f;
  if (m == 0) { w[0] = 0.f; f[0] = 0; w[1] = 1.0f; }
  if (m == n - 1) { w[3] = 0.f; f[3] = 2 * n - 1; w[2] = 1.0f; }
}

template <typename T, int VEC>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dout, T* __restrict__ din, int B, int H, int W, int C, int xcd) {
  const int CV = C / VEC;
  const int64_t total = (int64_t)B * H * W * CV;
  for (int64_t i = (int64_t)up2_block(xcd) * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    int64_t t = i / CV;
    const int mx = (int)(t % W); t /= W;
    const int my = (int)(t % H);
    const int b = (int)(t / H);
    int fy[4], fx[4];
    float wy[4], wx[4];
    up2_adj_taps(my, H, fy, wy);
    up2_adj_taps(mx, W, fx, wx);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    const T* base = dout + (int64_t)b * 4 * H * W * C + cv * VEC;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (wy[a] == 0.f) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (wx[c] == 0.f) continue;
        float v[VEC];
        if constexpr (VEC == 8) ld8<T>(base + ((int64_t)fy[a] * 2 * W + fx[c]) * C, v);
        else v[0] = ldf<T>(base + ((int64_t)fy[a] * 2 * W + fx[c]) * C);
        const float ww = wy[a] * wx[c];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += ww * v[e];
      }
    }
    T* dst = din + (((int64_t)b * H + my) * W + mx) * C + cv * VEC;
    if constexpr (VEC == 8) st8<T>(dst, acc); else stf<T>(dst, acc[0]);
  }
}

// The adjoint for the density head's maps (C = 256, W % 8 == 0): one workgroup per 8-pixel chunk of a coarse row (32 channel vectors x
// 8 pixels), its (image, row, chunk) taken from the block index with wave-uniform arithmetic -- the per-pixel kernel above spends ~490
// of its 705 VALU instructions per thread on per-lane indices (three integer divisions, 64-bit address products with quarter-rate
// multiplies); this form runs 339.  Same taps in the same order: identical values.  Workgroup order: XCD x takes the x-th contiguous
// eighth of the chunks (up2_block).  Back to back 50.7 -> 37.2 us at 96 -> 192 with both; in the step 54 -> 52 us, because there the
// input is the 151-MB map the convolution's dgrad has just written and the pass runs at the memory side's pace (tools/ubench_mall.py).
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_chunk_kernel(const T* __restrict__ dout, T* __restrict__ din, int B, int H, int W, int xcd) {
  constexpr int C = 256;
  const int nxc = W >> 3;
  const int lb = __builtin_amdgcn_readfirstlane(up2_block(xcd));
  const int row = lb / nxc, xc = lb - row * nxc;     // (uniform)
  const int b = row / H, my = row - b * H;
  if (b >= B) return;
  const int cv = threadIdx.x & 31, mx = xc * 8 + (threadIdx.x >> 5);
  int fy[4], fx[4];
  float wy[4], wx[4];
  up2_adj_taps(my, H, fy, wy);
  up2_adj_taps(mx, W, fx, wx);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const T* img = dout + (int64_t)b * 4 * H * W * C;                 // (uniform base; 32-bit offsets inside an image)
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (wy[a] == 0.f) continue;
    const int ro = fy[a] * 2 * W * C + cv * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (wx[c] == 0.f) continue;
      float v[8];
      ld8<T>(img + (ro + fx[c] * C), v);
      const float ww = wy[a] * wx[c];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += ww * v[e];
    }
  }
  st8<T>(din + (int64_t)b * H * W * C + ((my * W + mx) * C + cv * 8), acc);
}

// ---------------- dpre = dh * gelu'(pre)
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ pre, T* __restrict__ dpre, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float d[8], p[8], o[8];
    ld8<T>(dh + i * 8, d);
    ld8<T>(pre + i * 8, p);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = d[e] * gelu_grad_t<T>(p[e]);
    st8<T>(dpre + i * 8, o);
  }
}

// ---------------- column sums (bias gradients): partial[blockIdx.y][N] over row chunks.
// thread = (8-column vector, row lane): 16-byte loads, CV vectors per row pass, 256/CV row lanes per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ partial, int M, int N, int CV) {
  __shared__ float sm[256 * 8];
  const int cv = threadIdx.x % CV, rl = threadIdx.x / CV, RL = 256 / CV;
  const int col = (blockIdx.x * CV + cv) * 8;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(M, r0 + per);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (col < N) {
    for (int r = r0 + rl; r < r1; r += RL) {
      float v[8];
      ld8<T>(x + (int64_t)r * N + col, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sm[(rl * CV + cv) * 8 + e] = acc[e];
  __syncthreads();
  for (int i = threadIdx.x; i < CV * 8; i += 256) {
    float s = 0.f;
    for (int r = 0; r < RL; ++r) s += sm[r * CV * 8 + i];
    const int c = blockIdx.x * CV * 8 + i;
    if (c < N) partial[(int64_t)blockIdx.y * N + c] = s;
  }
}

// ---------------- weight shadows
// mode 0: plain cast [n]; mode 1: OIHW [Co][Ci][T] -> OHWI [Co][T][Ci];
// mode 2: dgrad form Wd[ci][tap'][co] = W[co][ci][T-1-tap'] (conv3x3 dgrad == fwd conv with Wd)
template <typename T>
__global__ void cast_permute_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n, int mode, int Co, int Ci,
                                    int taps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t s = i;
    if (mode == 1) {
      const int ci = (int)(i % Ci); const int tap = (int)((i / Ci) % taps); const int co = (int)(i / ((int64_t)Ci * taps));
      s = ((int64_t)co * Ci + ci) * taps + tap;
    } else if (mode == 2) {
      const int co = (int)(i % Co); const int tap = (int)((i / Co) % taps); const int ci = (int)(i / ((int64_t)Co * taps));
      s = ((int64_t)co * Ci + ci) * taps + (taps - 1 - tap);
    }
    stf<T>(dst + i, src[s]);
  }
}

// all conv-weight shadows of the model in ONE launch: blockIdx.y = convolution, both permuted forms per element
constexpr int CS_MAX = 32;
struct ConvShadowTable {
  int n;
  const float* src[CS_MAX];
  void* wf[CS_MAX];     // may be null: only the transposed form is wanted (a Linear weight, taps = 1: wd = W^T)
  void* wd[CS_MAX];
  int co[CS_MAX], ci[CS_MAX], taps[CS_MAX];
};
// One block = a 32 (co) x 8 (ci) tile of one convolution with all its taps: the torch OIHW source [co][ci][tap] is read as 32 runs of
// 8 x taps contiguous floats, the OHWI form [co][tap][ci] is written with ci fastest and the dgrad form [ci][taps-1-tap][co] -- the
// transpose -- with co fastest out of LDS (the element-per-thread version gathered the source with a 36-byte stride and paid six
// integer divisions per element).  Small tiles on purpose: the whole job is 4.5 M weights, so it needs many short blocks, not few long ones.
constexpr int CS_CI = 8;
template <typename T>
__global__ __launch_bounds__(256) void conv_shadows_kernel(const ConvShadowTable t) {
  __shared__ float tile[32][CS_CI * 9 + 1];
  const int e = blockIdx.y;
  const int Co = t.co[e], Ci = t.ci[e], taps = t.taps[e];
  const int tiles_ci = (Ci + CS_CI - 1) / CS_CI, ntiles = ((Co + 31) / 32) * tiles_ci;
  const float* __restrict__ src = t.src[e];
  T* __restrict__ wf = reinterpret_cast<T*>(t.wf[e]);
  T* __restrict__ wd = reinterpret_cast<T*>(t.wd[e]);
  if (taps == 1) {
    // a Linear weight [Co][Ci] -> its transpose [Ci][Co] (+ the plain cast when wf is given): 64 x 64 tiles, 256-byte source runs and
    // 128-byte destination runs (the 32 x 8 tiles below read 32-byte runs: these 8.4 M elements were 2/3 of the launch's time)
    __shared__ float tt[64][65];
    const int tci = (Ci + 63) / 64, nt = ((Co + 63) / 64) * tci;
    for (int tl = blockIdx.x; tl < nt; tl += gridDim.x) {
      const int co0 = (tl / tci) * 64, ci0 = (tl % tci) * 64;
      __syncthreads();
      for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63, co = co0 + r, ci = ci0 + c;
        const float v = (co < Co && ci < Ci) ? src[(int64_t)co * Ci + ci] : 0.f;
        tt[r][c] = v;
        if (wf && co < Co && ci < Ci) stf<T>(wf + (int64_t)co * Ci + ci, v);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63, ci = ci0 + r, co = co0 + c;
        if (co < Co && ci < Ci) stf<T>(wd + (int64_t)ci * Co + co, tt[c][r]);
      }
    }
    return;
  }
  const int run = CS_CI * taps;                   // floats per co row of the tile (taps <= 9)
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int co0 = (tl / tiles_ci) * 32, ci0 = (tl % tiles_ci) * CS_CI;
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
      const int r = i / run, c = i - r * run;     // c = ci_local * taps + tap
      const int co = co0 + r, ci = ci0 + c / taps;
      tile[r][c] = (co < Co && ci < Ci) ? src[((int64_t)co * Ci + ci0) * taps + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
      {  // OHWI [Co][tap][Ci]: lanes walk ci
        const int cl = i % CS_CI, rt = i / CS_CI, tap = rt % taps, r = rt / taps;
        const int co = co0 + r, ci = ci0 + cl;
        if (wf && co < Co && ci < Ci) stf<T>(wf + ((int64_t)co * taps + tap) * Ci + ci, tile[r][cl * taps + tap]);
      }
      {  // dgrad form [Ci][taps-1-tap][Co]: lanes walk co
        const int rl = i & 31, ct = i >> 5, tap = ct % taps, cl = ct / taps;
        const int co = co0 + rl, ci = ci0 + cl;
        if (co < Co && ci < Ci) stf<T>(wd + ((int64_t)ci * taps + (taps - 1 - tap)) * Co + co, tile[rl][cl * taps + tap]);
      }
    }
  }
}

// ---------------- masked MSE loss + its gradient + counts (FSC_finetune_cross.py:290-303)
// loss = sum((pred-gt)^2 * mask / HW) / B ; dpred = 2 (pred-gt) mask / (HW * B) * grad_scale
// sums[0] = loss, sums[1 + b] = sum(pred[b]) / 60, sums[1 + B + b] = sum(gt[b]) / 60
// Two deterministic stages (no atomics, no memset: safe under graph replay): per-block partials, then one finisher.
constexpr int MSE_BLOCKS = 64;
__global__ __launch_bounds__(256) void masked_mse_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         const float* __restrict__ mask, float* __restrict__ dpred,
                                                         float* __restrict__ partial /* [B][MSE_BLOCKS][3] */, int B, int HW,
                                                         float grad_scale, const float* __restrict__ amp) {
  __shared__ float sm[4];
  if (amp) grad_scale *= amp[0];      // dynamic loss scale (fp16 mode: countr_amp_* below)
  const int b = blockIdx.y;
  float l = 0.f, sp = 0.f, sg = 0.f;
  const float inv = 1.f / ((float)HW * B);
  const bool vec = (HW & 3) == 0 && ((((uintptr_t)pred | (uintptr_t)gt | (uintptr_t)mask | (uintptr_t)dpred) & 15) == 0);
  if (vec) {   // four pixels per thread and trip: 147456 pixels / 64 blocks were nine dependent scalar round trips per thread
    const float4* p4 = reinterpret_cast<const float4*>(pred + (int64_t)b * HW);
    const float4* g4 = reinterpret_cast<const float4*>(gt + (int64_t)b * HW);
    const float4* m4 = reinterpret_cast<const float4*>(mask);
    float4* d4 = dpred ? reinterpret_cast<float4*>(dpred + (int64_t)b * HW) : nullptr;
    const float k = 2.f * inv * grad_scale;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += gridDim.x * 256) {
      const float4 p = p4[i], g = g4[i], m = m4[i];
      const float dx = p.x - g.x, dy = p.y - g.y, dz = p.z - g.z, dw = p.w - g.w;
      l += dx * dx * m.x; l += dy * dy * m.y; l += dz * dz * m.z; l += dw * dw * m.w;
      sp += p.x; sp += p.y; sp += p.z; sp += p.w;
      sg += g.x; sg += g.y; sg += g.z; sg += g.w;
      if (d4) d4[i] = float4{dx * m.x * k, dy * m.y * k, dz * m.z * k, dw * m.w * k};
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
      const float p = pred[(int64_t)b * HW + i], g = gt[(int64_t)b * HW + i], m = mask[i];
      const float d = p - g;
      l += d * d * m;
      sp += p; sg += g;
      if (dpred) dpred[(int64_t)b * HW + i] = 2.f * d * m * inv * grad_scale;
    }
  }
  l = block_sum<4>(l, sm);
  sp = block_sum<4>(sp, sm);
  sg = block_sum<4>(sg, sm);
  if (threadIdx.x == 0) {
    float* o = partial + ((int64_t)b * gridDim.x + blockIdx.x) * 3;
    o[0] = l * inv; o[1] = sp / 60.f; o[2] = sg / 60.f;
  }
}
__global__ __launch_bounds__(64) void masked_mse_finish_kernel(const float* __restrict__ partial, float* __restrict__ sums, int B,
                                                               int nblk) {
  // one wave; lane = block partial index
  float loss = 0.f;
  for (int b = 0; b < B; ++b) {
    float l = 0.f, sp = 0.f, sg = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) {
      const float* o = partial + ((int64_t)b * nblk + i) * 3;
      l += o[0]; sp += o[1]; sg += o[2];
    }
    l = wave_sum(l); sp = wave_sum(sp); sg = wave_sum(sg);
    loss += l;
    if (threadIdx.x == 0) { sums[1 + b] = sp; sums[1 + B + b] = sg; }
  }
  if (threadIdx.x == 0) sums[0] = loss;
}

// ---------------- fused AdamW over flat fp32 buffers, with an optional low-precision shadow of the params
// torch.optim.AdamW keeps a step counter PER PARAMETER (bias corrections 1 - beta^t with t = number of steps that parameter took).
// The conditional parameter sets of the finetune step (exemplar CNN: no gradient when shot_num == 0; shot_token: only then) start
// later than the rest, so every range names one of three counter groups; with the device-side scalars (graph replay) the layout is
// hyper[8] = {lr, bc1[0], bc2[0], grad_scale, bc1[1], bc2[1], bc1[2], bc2[2]}.  zero[r] != 0: the range is stepped with a zero
// gradient (torch 1.13 optimizer.zero_grad() keeps zero tensors, so a parameter that had a gradient once is stepped ever after).
struct AdamRanges {
  int n;
  int64_t start[16], end[16];
  float wd[16];
  int grp[16], zero[16];
};
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             bf16_t* __restrict__ shadow, AdamRanges R, float lr, float b1, float b2, float eps, float bc1s,
                             float bc2s, float grad_scale, const float* __restrict__ hyper, float* __restrict__ gnorm_ws,
                             const float* __restrict__ amp) {
  __shared__ float red[4];
  if (hyper) { lr = hyper[0]; grad_scale = hyper[3]; }  // graph-replay safe
  if (amp) {                      // fp16 mode (GradScaler semantics, util/misc.py:260-286): a non-finite gradient skips the whole update
    if (amp[2] != 0.f) return;
    grad_scale /= amp[0];         // unscale
  }
  float sq = 0.f;
  for (int r = 0; r < R.n; ++r) {
    const float wd = R.wd[r];
    const int gr = R.grp[r];
    const float bc1 = hyper ? hyper[gr == 0 ? 1 : 2 + 2 * gr] : bc1s, bc2 = hyper ? hyper[gr == 0 ? 2 : 3 + 2 * gr] : bc2s;
    const bool zg = R.zero[r] != 0;
    for (int64_t i = R.start[r] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R.end[r];
         i += (int64_t)gridDim.x * blockDim.x) {
      const float gi = zg ? 0.f : g[i] * grad_scale;
      sq += gi * gi;
      float pi = p[i] * (1.f - lr * wd);
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      m[i] = mi; v[i] = vi;
      pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
      p[i] = pi;
      if (shadow) shadow[i] = f2bf(pi);
    }
  }
  if (gnorm_ws) {   // deterministic two-stage sum of squares of the (scaled) gradients: util/misc.py:289-301 get_grad_norm_
    sq = block_sum<4>(sq, red);
    if (threadIdx.x == 0) gnorm_ws[1 + blockIdx.x] = sq;
  }
}
// amp = {scale, good steps, found_inf (set by amp_check_kernel), skipped steps, growth interval}: torch.cuda.amp.GradScaler's update():
// found_inf -> scale *= 0.5, good = 0; else ++good == interval -> scale *= 2, good = 0 (util/misc.py:264,278: GradScaler() defaults)
__device__ __forceinline__ void amp_update(float* __restrict__ amp, float* __restrict__ norm_out) {
  if (amp[2] != 0.f) {
    amp[0] *= 0.5f; amp[1] = 0.f; amp[3] += 1.f;
    if (norm_out) *norm_out = INFINITY;            // (what get_grad_norm_ returns for a non-finite gradient)
  } else {
    amp[1] += 1.f;
    if (amp[1] >= amp[4]) { amp[0] *= 2.f; amp[1] = 0.f; }
  }
  amp[2] = 0.f;
}
__global__ __launch_bounds__(1024) void gnorm_finish_kernel(float* __restrict__ ws, int nb, float* __restrict__ amp) {   // nb <= 2048 partials: two independent loads per thread
  __shared__ float part[16];
  const int t = threadIdx.x;
  if (amp && amp[2] != 0.f) {     // skipped update: no partials were written
    __syncthreads();
    if (t == 0) amp_update(amp, ws);
    return;
  }
  const float a = t < nb ? ws[1 + t] : 0.f, b = t + 1024 < nb ? ws[1 + t + 1024] : 0.f;
  const float s = wave_sum(a + b);     // fixed summation tree: deterministic
  if ((t & 63) == 0) part[t >> 6] = s;
  __syncthreads();
  if (t == 0) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += part[k];
    ws[0] = sqrtf(tot);
    if (amp) amp_update(amp, nullptr);
  }
}
__global__ __launch_bounds__(64) void amp_update_kernel(float* __restrict__ amp) { if (threadIdx.x == 0) amp_update(amp, nullptr); }

// found_inf of GradScaler.unscale_: amp[2] = 1 if any gradient element of the ranges the update will read is non-finite.  (Plain stores of
// the same value from every block that finds one: no atomics.)  After a gradient all-reduce every rank sees the same flag: inf / nan
// survive the sum.
__global__ __launch_bounds__(256) void amp_check_kernel(const float* __restrict__ g, AdamRanges R, float* __restrict__ amp) {
  bool bad = false;
  for (int r = 0; r < R.n; ++r) {
    if (R.zero[r]) continue;
    for (int64_t i = R.start[r] + (int64_t)blockIdx.x * 256 + threadIdx.x; i < R.end[r]; i += (int64_t)gridDim.x * 256 * 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = i + (int64_t)u * gridDim.x * 256;
        if (k < R.end[r]) bad |= !isfinite(g[k]);
      }
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) amp[2] = 1.f;
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)
static inline int nblocks(int64_t work, int per_block = 256, int cap = 4096) {
  int64_t b = (work + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

extern "C" int countr_im2patch(const float* img, void* out, int B, int H, int W, int patch, int dtype, void* stream) {
  if (!img || !out || patch <= 0) { countr_set_error("countr_im2patch: bad args"); return -1; }
  const int gh = H / patch, gw = W / patch;
  const int64_t total = (int64_t)B * gh * gw * 3 * patch * patch;
  if ((patch & 7) == 0 && (W & 3) == 0 && ((uintptr_t)img & 15) == 0 && ((uintptr_t)out & 31) == 0 && total / 8 < (int64_t)0x7fffffff) {
    const int nb = nblocks(total / 8);
    if (dtype == COUNTR_BF16) hipLaunchKernelGGL(im2patch_vec8_kernel<bf16_t>, dim3(nb), dim3(256), 0, STREAM(stream), img, (bf16_t*)out, B, H, W, patch, gh, gw);
    else hipLaunchKernelGGL(im2patch_vec8_kernel<float>, dim3(nb), dim3(256), 0, STREAM(stream), img, (float*)out, B, H, W, patch, gh, gw);
    COUNTR_LAUNCH_CHECK("countr_im2patch");
  }
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL(im2patch_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), img, (bf16_t*)out, B, H, W, patch, gh, gw);
  else hipLaunchKernelGGL(im2patch_kernel<float>, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), img, (float*)out, B, H, W, patch, gh, gw);
  COUNTR_LAUNCH_CHECK("countr_im2patch");
}

extern "C" int countr_conv3x3_c3_fwd(const float* in, const float* w, const float* bias, void* out, int S, int H, int W,
                                     int dtype, void* stream) {
  if (!in || !w || !bias || !out) { countr_set_error("countr_conv3x3_c3_fwd: null"); return -1; }
  // every block first stages the 64x27 weights into LDS: few, long-lived blocks (COUNTR_C3_BLOCKS overrides the cap for tuning)
  constexpr int cap = 256;   // measured at 24 boxes: 2048 blocks 31.2 us, 512 23.6, 256 22.2, 128 38.2
  const int nb = nblocks((int64_t)S * H * W, 32, cap);
  // four pixels per thread where rows are whole quads and 16-byte aligned (the 64 x 64 exemplar crops): COUNTR_C3_QUAD=0 keeps the per-pixel kernel
  const char* eq = getenv("COUNTR_C3_QUAD");      // (read per call: tests compare the two kernels inside one process)
  const int quad = eq ? atoi(eq) : 1;
  if (quad && (W % 4) == 0 && ((uintptr_t)in & 15) == 0 && (int64_t)S * H * W < ((int64_t)1 << 30)) {
    const int nbq = nblocks((int64_t)S * H * (W / 4), 32, cap);
    if (dtype == COUNTR_BF16) hipLaunchKernelGGL(conv3x3_c3_fwd4_kernel<bf16_t>, dim3(nbq), dim3(256), 0, STREAM(stream), in, w, bias, (bf16_t*)out, S, H, W);
    else hipLaunchKernelGGL(conv3x3_c3_fwd4_kernel<float>, dim3(nbq), dim3(256), 0, STREAM(stream), in, w, bias, (float*)out, S, H, W);
    COUNTR_LAUNCH_CHECK("countr_conv3x3_c3_fwd");
  }
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL(conv3x3_c3_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, STREAM(stream), in, w, bias, (bf16_t*)out, S, H, W);
  else hipLaunchKernelGGL(conv3x3_c3_fwd_kernel<float>, dim3(nb), dim3(256), 0, STREAM(stream), in, w, bias, (float*)out, S, H, W);
  COUNTR_LAUNCH_CHECK("countr_conv3x3_c3_fwd");
}

extern "C" int countr_conv3x3_c3_wgrad_nblocks(void) {
  // 24 boxes of 64x64 (98304 pixels): 256 blocks 49.0 + 10.5 us (finish), 512: 32.3 + 15.3, 1024: 39.1 + 25.5 -- with the 16-group finish
  constexpr int nb = 512;
  return nb;
}
extern "C" int countr_conv3x3_c3_wgrad(const float* in, const void* dy, float* dw, float* db, float* workspace, int S, int H,
                                       int W, int dtype, int accumulate, void* stream) {
  if (!in || !dy || !dw || !db || !workspace) { countr_set_error("countr_conv3x3_c3_wgrad: null"); return -1; }
  if ((int64_t)S * H * W >= (int64_t)1 << 31) { countr_set_error("countr_conv3x3_c3_wgrad: more than 2^31 pixels"); return -1; }
  const int nb = countr_conv3x3_c3_wgrad_nblocks();
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL(conv3x3_c3_wgrad_kernel<bf16_t>, dim3(nb), dim3(256), 0, STREAM(stream), in, (const bf16_t*)dy, workspace, S, H, W);
  else hipLaunchKernelGGL(conv3x3_c3_wgrad_kernel<float>, dim3(nb), dim3(256), 0, STREAM(stream), in, (const float*)dy, workspace, S, H, W);
  hipLaunchKernelGGL(conv3x3_c3_wgrad_finish_kernel, dim3((64 * 28 + 15) / 16), dim3(16 * C3F_GROUPS), 0, STREAM(stream), workspace, dw, db, nb, accumulate);
  COUNTR_LAUNCH_CHECK("countr_conv3x3_c3_wgrad");
}

// Grid of the XCD-aware form (up2_block): one pass over the map, a multiple of 8 blocks, uncapped; COUNTR_UP2_XCD=0 keeps launch order.
static int up2_xcd_grid(int64_t work, int& nb) {
  const char* e = getenv("COUNTR_UP2_XCD");
  const int on = e ? atoi(e) : 1;
  const int64_t b = (work + 255) / 256;
  if (!on || b < 64 || b > (1 << 22)) return 0;
  nb = (int)((b + 7) / 8 * 8);
  return 1;
}

// the adjoint's chunk form (upsample2x_bwd_chunk_kernel): 256 channels, rows of whole 8-pixel chunks, 32-bit element offsets inside
// one image's fine map; COUNTR_UP2_ROWS=0 keeps the per-pixel kernel
static bool up2_rows_form(int B, int H, int W, int C) {
  const char* e = getenv("COUNTR_UP2_ROWS");      // (read per call: A/B and tests inside one process; not on a replay path)
  const int on = e ? atoi(e) : 1;
  return on && C == 256 && (W % 8) == 0 && W >= 8 && (long long)4 * H * W * C < (1ll << 30) && (long long)B * H * (W / 8) < (1ll << 30);
}

extern "C" int countr_upsample2x_fwd(const void* in, void* out, int B, int H, int W, int C, int dtype, void* stream) {
  if (!in || !out || (C != 1 && C % 8)) { countr_set_error("countr_upsample2x_fwd: C must be 1 or a multiple of 8"); return -1; }
  const int64_t total = (int64_t)B * 4 * H * W * (C == 1 ? 1 : C / 8);
  const int nb = nblocks(total, 256, 8192);
  int nbq = nblocks(total / 4, 256, 8192);     // multi-channel maps: one thread per 2x2 block of fine pixels (quad kernel)
  const int xcd = up2_xcd_grid(total / 4, nbq);
  if (dtype == COUNTR_BF16) {
    if (C == 1) hipLaunchKernelGGL((upsample2x_fwd_kernel<bf16_t, 1>), dim3(nb), dim3(256), 0, STREAM(stream), (const bf16_t*)in, (bf16_t*)out, B, H, W, C);
    else hipLaunchKernelGGL((upsample2x_fwd_quad_kernel<bf16_t>), dim3(nbq), dim3(256), 0, STREAM(stream), (const bf16_t*)in, (bf16_t*)out, B, H, W, C, xcd);
  } else {
    if (C == 1) hipLaunchKernelGGL((upsample2x_fwd_kernel<float, 1>), dim3(nb), dim3(256), 0, STREAM(stream), (const float*)in, (float*)out, B, H, W, C);
    else hipLaunchKernelGGL((upsample2x_fwd_quad_kernel<float>), dim3(nbq), dim3(256), 0, STREAM(stream), (const float*)in, (float*)out, B, H, W, C, xcd);
  }
  COUNTR_LAUNCH_CHECK("countr_upsample2x_fwd");
}

extern "C" int countr_upsample2x_bwd(const void* dout, void* din, int B, int H, int W, int C, int dtype, void* stream) {
  if (!dout || !din || (C != 1 && C % 8)) { countr_set_error("countr_upsample2x_bwd: C must be 1 or a multiple of 8"); return -1; }
  const int64_t total = (int64_t)B * H * W * (C == 1 ? 1 : C / 8);
  int nb = nblocks(total, 256, 8192);
  // (a 2x2-coarse-block variant like the forward's was measured: 50.7 vs 50.2 us at 96 -> 192 and slower on the small maps)
  const int xcd = C == 1 ? 0 : up2_xcd_grid(total, nb);
  if (up2_rows_form(B, H, W, C)) {     // the density head's maps: one workgroup per 8-pixel chunk of a coarse row
    const int nbr = (B * H * (W / 8) + 7) / 8 * 8;
    if (dtype == COUNTR_BF16) hipLaunchKernelGGL((upsample2x_bwd_chunk_kernel<bf16_t>), dim3(nbr), dim3(256), 0, STREAM(stream), (const bf16_t*)dout, (bf16_t*)din, B, H, W, 1);
    else hipLaunchKernelGGL((upsample2x_bwd_chunk_kernel<float>), dim3(nbr), dim3(256), 0, STREAM(stream), (const float*)dout, (float*)din, B, H, W, 1);
    COUNTR_LAUNCH_CHECK("countr_upsample2x_bwd");
  }
  if (dtype == COUNTR_BF16) {
    if (C == 1) hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t, 1>), dim3(nb), dim3(256), 0, STREAM(stream), (const bf16_t*)dout, (bf16_t*)din, B, H, W, C, 0);
    else hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t, 8>), dim3(nb), dim3(256), 0, STREAM(stream), (const bf16_t*)dout, (bf16_t*)din, B, H, W, C, xcd);
  } else {
    if (C == 1) hipLaunchKernelGGL((upsample2x_bwd_kernel<float, 1>), dim3(nb), dim3(256), 0, STREAM(stream), (const float*)dout, (float*)din, B, H, W, C, 0);
    else hipLaunchKernelGGL((upsample2x_bwd_kernel<float, 8>), dim3(nb), dim3(256), 0, STREAM(stream), (const float*)dout, (float*)din, B, H, W, C, xcd);
  }
  COUNTR_LAUNCH_CHECK("countr_upsample2x_bwd");
}

// Several device-to-device copies in ONE launch (the per-step staging of a batch: images, exemplar crops, ground-truth map, loss mask --
// four ~5-us copy launches in front of every step otherwise).  16-byte aligned pointers and sizes.  A NULL source zero-fills its
// destination (a gradient bucket this rank has no gradient for while another rank has: per-rank shot_num, trainer.py).
struct CopyTable { int n; const uint4* src[8]; uint4* dst[8]; long long n16[8]; int first[8]; };
__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyTable t) {
  int e = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) e += (i < t.n && (int)blockIdx.x >= t.first[i]) ? 1 : 0;
  const int nb = (e + 1 < t.n ? t.first[e + 1] : (int)gridDim.x) - t.first[e];
  const uint4* __restrict__ s = t.src[e];
  uint4* __restrict__ d = t.dst[e];
  const long long n = t.n16[e], stride = (long long)nb * 256 * 4;
  for (long long i = ((long long)blockIdx.x - t.first[e]) * 256 + threadIdx.x; i < n; i += stride) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + (long long)u * nb * 256 < n) v[u] = s ? s[i + (long long)u * nb * 256] : uint4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + (long long)u * nb * 256 < n) d[i + (long long)u * nb * 256] = v[u];
  }
}

extern "C" int countr_copy_multi(int n, const void* const* src, void* const* dst, const int64_t* bytes, void* stream) {
  if (n < 1 || n > 8 || !src || !dst || !bytes) { countr_set_error("countr_copy_multi: 1..8 copies"); return -1; }
  CopyTable t;
  t.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!dst[i] || bytes[i] <= 0 || (bytes[i] & 15) || (((uintptr_t)src[i] | (uintptr_t)dst[i]) & 15)) {      // src[i] == NULL: zero fill
      countr_set_error("countr_copy_multi: null, empty or not 16-byte aligned"); return -1;
    }
    t.src[i] = reinterpret_cast<const uint4*>(src[i]); t.dst[i] = reinterpret_cast<uint4*>(dst[i]); t.n16[i] = bytes[i] / 16;
    t.first[i] = blocks;
    blocks += nblocks(t.n16[i], 1024, 2048);     // 4 x 16 bytes per thread and trip
  }
  for (int i = n; i < 8; ++i) { t.src[i] = nullptr; t.dst[i] = nullptr; t.n16[i] = 0; t.first[i] = blocks; }
  hipLaunchKernelGGL(copy_multi_kernel, dim3(blocks), dim3(256), 0, STREAM(stream), t);
  COUNTR_LAUNCH_CHECK("countr_copy_multi");
}

// ---- step prologue: everything an optimisation step needs from the host, as ONE kernel node at the head of the step's hipGraph ----
// (FSC_finetune_cross.py:271-295: lr for the iteration, the batch hand-over, the Bernoulli(0.8) loss mask drawn per iteration.)
// Between two graph replays every separate launch / copy costs a queue hand-over (round 4: mask draw + staging launch = 3.6 + 10.6 us
// of work inside ~160 us of idle GPU).  The arguments of a captured node are frozen, so what changes per step travels through a RING
// of 256-byte records in pinned host memory that the device reads directly: record index = *counter % slots, where counter is a
// device int64 every execution increments -- eager launches and replays alike, so the host mirrors it by counting executions and
// fills record (executions % slots) before each one.
struct PrologueRec {               // 256 bytes, host-written (countr_amd/trainer.py::_Prologue)
  unsigned long long src[6], dst[6];
  long long n16[6];
  int first[6];                    // first copy block of entry i (blocks [0, COPY_BLOCKS) are dealt by the host in proportion to the bytes)
  int n, draw_mask;
  unsigned int key[2], ctr[2];     // Philox4x32-10 key / high counter words (seed, step)
  float hyper[8];                  // AdamW scalars {lr, bc1[0], bc2[0], grad_scale, bc1[1], bc2[1], bc1[2], bc2[2]}
  unsigned int mask_thr;           // mask = 1 where the 32-bit draw < mask_thr (= floor(p 2^32))
  unsigned int pad[7];
};
static_assert(sizeof(PrologueRec) == 256, "PrologueRec layout is part of the ABI");
constexpr int PRO_COPY_BLOCKS = 1152, PRO_MASK_BLOCKS = 96;

__device__ __forceinline__ void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3, unsigned int k0, unsigned int k1,
                                              unsigned int out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned int)p1; c3 = (unsigned int)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Two kernel nodes: a one-block fetch (the ONLY reader of host memory: 1248 blocks reading their record over PCIe took 98 us) copies
// record (counter % slots) into device memory, writes the AdamW scalars and counts the execution; the wide kernel behind it reads the
// device copy.
__global__ __launch_bounds__(64) void step_prologue_fetch_kernel(const PrologueRec* __restrict__ ring, int slots, long long* counter, PrologueRec* rec_dev,
                                                                float* hyper_dev) {
  const long long seq = counter[0];
  const uint4* r = reinterpret_cast<const uint4*>(ring + (seq % slots));
  if (threadIdx.x < 16) {
    const uint4 v = r[threadIdx.x];
    reinterpret_cast<uint4*>(rec_dev)[threadIdx.x] = v;
    if (threadIdx.x == 12 || threadIdx.x == 13) reinterpret_cast<uint4*>(hyper_dev)[threadIdx.x - 12] = v;      // bytes 192..223: hyper[8]
  }
  if (threadIdx.x == 0) counter[0] = seq + 1;       // the next execution (stream order) reads the next record
}

__global__ __launch_bounds__(256) void step_prologue_kernel(const PrologueRec* __restrict__ rec_dev, float* mask, int mask_n4) {
  __shared__ PrologueRec rec;
  if (threadIdx.x < 16) reinterpret_cast<uint4*>(&rec)[threadIdx.x] = reinterpret_cast<const uint4*>(rec_dev)[threadIdx.x];
  __syncthreads();
  const int b = blockIdx.x;
  if (b < PRO_COPY_BLOCKS) {
    int e = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i) e += (i < rec.n && b >= rec.first[i]) ? 1 : 0;
    if (rec.n > 0) {
      const int nb = (e + 1 < rec.n ? rec.first[e + 1] : PRO_COPY_BLOCKS) - rec.first[e];
      const uint4* __restrict__ s = reinterpret_cast<const uint4*>(rec.src[e]);
      uint4* __restrict__ d = reinterpret_cast<uint4*>(rec.dst[e]);
      const long long n = rec.n16[e], stride = (long long)nb * 256 * 4;
      for (long long i = ((long long)b - rec.first[e]) * 256 + threadIdx.x; i < n; i += stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + (long long)u * nb * 256 < n) v[u] = s ? s[i + (long long)u * nb * 256] : uint4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + (long long)u * nb * 256 < n) d[i + (long long)u * nb * 256] = v[u];
      }
    }
  } else if (rec.draw_mask) {      // element 4 g + j of the mask = word j of Philox(counter = (g, 0, ctr[0], ctr[1]), key) < mask_thr
    for (int g = (b - PRO_COPY_BLOCKS) * 256 + threadIdx.x; g < mask_n4; g += PRO_MASK_BLOCKS * 256) {
      unsigned int r[4];
      philox4x32_10((unsigned int)g, 0u, rec.ctr[0], rec.ctr[1], rec.key[0], rec.key[1], r);
      reinterpret_cast<float4*>(mask)[g] = float4{r[0] < rec.mask_thr ? 1.f : 0.f, r[1] < rec.mask_thr ? 1.f : 0.f,
                                                  r[2] < rec.mask_thr ? 1.f : 0.f, r[3] < rec.mask_thr ? 1.f : 0.f};
    }
  }
}

extern "C" int countr_step_prologue_record_bytes(void) { return (int)sizeof(PrologueRec); }
extern "C" int countr_step_prologue_copy_blocks(void) { return PRO_COPY_BLOCKS; }
extern "C" int countr_step_prologue(const void* ring, int slots, int64_t* counter, float* hyper_dev, float* mask, int mask_n, void* stream) {
  if (!ring || slots < 1 || !counter || !hyper_dev || (((uintptr_t)ring | (uintptr_t)counter | (uintptr_t)hyper_dev) & 15) ||
      (mask && ((mask_n & 3) || ((uintptr_t)mask & 15)))) {
    countr_set_error("countr_step_prologue: null / unaligned argument (16-byte pointers, mask_n % 4 == 0)"); return -1;
  }
  PrologueRec* rec_dev = reinterpret_cast<PrologueRec*>(counter + 2);       // counter: int64[2 + 32]: {executions, reserved, record copy}
  hipLaunchKernelGGL(step_prologue_fetch_kernel, dim3(1), dim3(64), 0, STREAM(stream), reinterpret_cast<const PrologueRec*>(ring), slots,
                     reinterpret_cast<long long*>(counter), rec_dev, hyper_dev);
  hipLaunchKernelGGL(step_prologue_kernel, dim3(PRO_COPY_BLOCKS + (mask ? PRO_MASK_BLOCKS : 0)), dim3(256), 0, STREAM(stream),
                     rec_dev, mask, mask ? mask_n / 4 : 0);
  COUNTR_LAUNCH_CHECK("countr_step_prologue");
}

extern "C" int countr_gelu_bwd(const void* dh, const void* pre, void* dpre, int64_t n, int dtype, void* stream) {
  if (!dh || !pre || !dpre || (n & 7)) { countr_set_error("countr_gelu_bwd: n must be a multiple of 8"); return -1; }
  const int nb = nblocks(n / 8, 256, 8192);
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, STREAM(stream), (const bf16_t*)dh, (const bf16_t*)pre, (bf16_t*)dpre, n / 8);
  else hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(nb), dim3(256), 0, STREAM(stream), (const float*)dh, (const float*)pre, (float*)dpre, n / 8);
  COUNTR_LAUNCH_CHECK("countr_gelu_bwd");
}

extern "C" int countr_colsum_nparts(void) { return 256; }
// workspace: fp32 [256][N]
extern "C" int countr_colsum(const void* x, float* out, float* workspace, int M, int N, int dtype, int accumulate, void* stream);

extern "C" int countr_cast_permute(const float* src, void* dst, int64_t n, int mode, int Co, int Ci, int taps, int dtype,
                                   void* stream) {
  if (!src || !dst) { countr_set_error("countr_cast_permute: null"); return -1; }
  const int nb = nblocks(n, 256, 4096);
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL(cast_permute_kernel<bf16_t>, dim3(nb), dim3(256), 0, STREAM(stream), src, (bf16_t*)dst, n, mode, Co, Ci, taps);
  else hipLaunchKernelGGL(cast_permute_kernel<float>, dim3(nb), dim3(256), 0, STREAM(stream), src, (float*)dst, n, mode, Co, Ci, taps);
  COUNTR_LAUNCH_CHECK("countr_cast_permute");
}

extern "C" int countr_conv_shadows(int n, const float* const* src, void* const* wf, void* const* wd, const int* co, const int* ci,
                                   const int* taps, int dtype, void* stream) {
  if (n < 1 || n > CS_MAX || !src || !wf || !wd || !co || !ci || !taps) { countr_set_error("countr_conv_shadows: 1..32 weights"); return -1; }
  ConvShadowTable t;
  t.n = n;
  int64_t big = 0;
  for (int i = 0; i < n; ++i) {
    if (!src[i] || !wd[i]) { countr_set_error("countr_conv_shadows: null"); return -1; }
    t.src[i] = src[i]; t.wf[i] = wf[i]; t.wd[i] = wd[i]; t.co[i] = co[i]; t.ci[i] = ci[i]; t.taps[i] = taps[i];
    const int64_t m = (int64_t)co[i] * ci[i] * taps[i];
    if (m > big) big = m;
  }
  for (int i = 0; i < n; ++i)
    if (taps[i] > 9) { countr_set_error("countr_conv_shadows: at most 9 taps"); return -1; }
  dim3 grid(nblocks(big, 32 * CS_CI * 9, 1024), n);   // one block per 32 x 8 x taps tile of the largest convolution
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL(conv_shadows_kernel<bf16_t>, grid, dim3(256), 0, STREAM(stream), t);
  else hipLaunchKernelGGL(conv_shadows_kernel<float>, grid, dim3(256), 0, STREAM(stream), t);
  COUNTR_LAUNCH_CHECK("countr_conv_shadows");
}

// Transposes of up to 96 16-bit matrices in ONE launch (blockIdx.y = matrix): dst[c][r] = src[r][c].  The refresh of the W^T shadows
// behind AdamW, which has just written the 16-bit shadow W itself: reading THAT instead of the fp32 master halves the bytes read, and the
// tiles move as 16-byte chunks both ways (64 x 64 elements through LDS: 128-byte runs in, 128-byte runs out).  The bits are those of
// countr_conv_shadows' taps = 1 form (a cast of the same fp32 value, transposed).
constexpr int TR_MAX = 96, TR_PITCH = 72;      // LDS row pitch in elements: 144 bytes, 16-byte aligned rows, odd multiple of 16 bytes
struct TransposeTable {
  const uint16_t* src[TR_MAX];
  uint16_t* dst[TR_MAX];
  int rows[TR_MAX], cols[TR_MAX];
};
__global__ __launch_bounds__(256) void transpose16_kernel(const TransposeTable t) {
  __shared__ __attribute__((aligned(16))) uint16_t tt[64 * TR_PITCH];
  const int e = blockIdx.y;
  const int R = t.rows[e], Cn = t.cols[e];
  const uint16_t* __restrict__ src = t.src[e];
  uint16_t* __restrict__ dst = t.dst[e];
  const int tc = Cn >> 6, nt = (R >> 6) * tc;
  for (int tl = blockIdx.x; tl < nt; tl += gridDim.x) {
    const int r0 = (tl / tc) * 64, c0 = (tl - (tl / tc) * tc) * 64;
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int idx = threadIdx.x + 256 * ps, r = idx >> 3, c8 = idx & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + r) * Cn + c0 + c8 * 8);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tt[(c8 * 8 + 2 * j) * TR_PITCH + r] = (uint16_t)(w[j] & 0xffffu);
        tt[(c8 * 8 + 2 * j + 1) * TR_PITCH + r] = (uint16_t)(w[j] >> 16);
      }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int idx = threadIdx.x + 256 * ps, c = idx >> 3, r8 = idx & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(tt + c * TR_PITCH + r8 * 8);
      *reinterpret_cast<uint4*>(dst + (int64_t)(c0 + c) * R + r0 + r8 * 8) = v;
    }
  }
}

extern "C" int countr_transpose16(int n, const void* const* src, void* const* dst, const int* rows, const int* cols, void* stream) {
  if (n < 1 || n > TR_MAX || !src || !dst || !rows || !cols) { countr_set_error("countr_transpose16: 1..96 matrices"); return -1; }
  TransposeTable t;
  int64_t big = 0;
  for (int i = 0; i < TR_MAX; ++i) {
    const int k = i < n ? i : 0;
    if (!src[k] || !dst[k] || rows[k] < 64 || cols[k] < 64 || (rows[k] % 64) || (cols[k] % 64) ||
        (((uintptr_t)src[k] | (uintptr_t)dst[k]) & 15)) {
      countr_set_error("countr_transpose16: rows and cols multiples of 64, 16-byte aligned matrices"); return -1;
    }
    t.src[i] = (const uint16_t*)src[k]; t.dst[i] = (uint16_t*)dst[k]; t.rows[i] = rows[k]; t.cols[i] = cols[k];
    const int64_t m = (int64_t)rows[k] * cols[k];
    if (m > big) big = m;
  }
  const int gx = (int)((big / 4096 + 3) / 4);       // up to four tiles of the largest matrix per block
  hipLaunchKernelGGL(transpose16_kernel, dim3(gx < 1 ? 1 : gx, n), dim3(256), 0, STREAM(stream), t);
  COUNTR_LAUNCH_CHECK("countr_transpose16");
}

extern "C" int countr_masked_mse_workspace_floats(int B) { return B * MSE_BLOCKS * 3; }
extern "C" int countr_masked_mse_amp(const float* pred, const float* gt, const float* mask, float* dpred, float* sums,
                                     float* workspace, int B, int HW, float grad_scale, const float* amp, void* stream);
extern "C" int countr_masked_mse(const float* pred, const float* gt, const float* mask, float* dpred, float* sums,
                                 float* workspace, int B, int HW, float grad_scale, void* stream) {
  return countr_masked_mse_amp(pred, gt, mask, dpred, sums, workspace, B, HW, grad_scale, nullptr, stream);
}
extern "C" int countr_masked_mse_amp(const float* pred, const float* gt, const float* mask, float* dpred, float* sums,
                                     float* workspace, int B, int HW, float grad_scale, const float* amp, void* stream) {
  if (!pred || !gt || !mask || !sums || !workspace) { countr_set_error("countr_masked_mse: null"); return -1; }
  hipLaunchKernelGGL(masked_mse_kernel, dim3(MSE_BLOCKS, B), dim3(256), 0, STREAM(stream), pred, gt, mask, dpred, workspace, B, HW, grad_scale, amp);
  hipLaunchKernelGGL(masked_mse_finish_kernel, dim3(1), dim3(64), 0, STREAM(stream), workspace, sums, B, MSE_BLOCKS);
  COUNTR_LAUNCH_CHECK("countr_masked_mse");
}

extern "C" int countr_adamw_gnorm_floats(void) { return 1 + 2048; }
extern "C" int countr_adamw_step_amp(float* p, const float* g, float* m, float* v, void* shadow_bf16, int nranges,
                                     const int64_t* starts, const int64_t* ends, const float* wds, const int* groups,
                                     const int* zero_grad, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                                     const float* hyper_dev, float* gnorm_ws, float* amp, void* stream);
extern "C" int countr_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int nranges,
                                 const int64_t* starts, const int64_t* ends, const float* wds, const int* groups,
                                 const int* zero_grad, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                                 const float* hyper_dev, float* gnorm_ws, void* stream) {
  return countr_adamw_step_amp(p, g, m, v, shadow_bf16, nranges, starts, ends, wds, groups, zero_grad, lr, beta1, beta2, eps, step, grad_scale,
                               hyper_dev, gnorm_ws, nullptr, stream);
}
extern "C" int countr_adamw_step_amp(float* p, const float* g, float* m, float* v, void* shadow_bf16, int nranges,
                                     const int64_t* starts, const int64_t* ends, const float* wds, const int* groups,
                                     const int* zero_grad, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                                     const float* hyper_dev, float* gnorm_ws, float* amp, void* stream) {
  if (!p || !g || !m || !v || nranges < 1 || nranges > 16 || (step < 1 && !hyper_dev)) { countr_set_error("countr_adamw_step: bad args (1..16 ranges, step >= 1)"); return -1; }
  AdamRanges R;
  R.n = nranges;
  int64_t total = 0;
  for (int i = 0; i < nranges; ++i) {
    R.start[i] = starts[i]; R.end[i] = ends[i]; R.wd[i] = wds[i]; total += ends[i] - starts[i];
    R.grp[i] = groups ? groups[i] : 0;
    R.zero[i] = zero_grad ? zero_grad[i] : 0;
    if (R.grp[i] < 0 || R.grp[i] > 2) { countr_set_error("countr_adamw_step: counter group must be 0, 1 or 2"); return -1; }
  }
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  const int nb = nblocks(total, 256, 2048);
  if (amp) hipLaunchKernelGGL(amp_check_kernel, dim3(nblocks(total, 1024, 2048)), dim3(256), 0, STREAM(stream), g, R, amp);
  hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, STREAM(stream), p, g, m, v, (bf16_t*)shadow_bf16, R, lr, beta1, beta2, eps, bc1, bc2, grad_scale, hyper_dev, gnorm_ws, amp);
  if (gnorm_ws) hipLaunchKernelGGL(gnorm_finish_kernel, dim3(1), dim3(1024), 0, STREAM(stream), gnorm_ws, nb, amp);
  else if (amp) hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, STREAM(stream), amp);
  COUNTR_LAUNCH_CHECK("countr_adamw_step");
}

extern "C" int countr_colsum(const void* x, float* out, float* workspace, int M, int N, int dtype, int accumulate, void* stream) {
  if (!x || !out || !workspace || (N & 7)) { countr_set_error("countr_colsum: N must be a multiple of 8"); return -1; }
  int CV = N / 8; if (CV > 32) CV = 32;
  while (256 % CV) --CV;                      // CV must divide 256 (N/8 is 64, 32, 16, ... for our shapes)
  const int ctiles = (N / 8 + CV - 1) / CV;
  int parts = 2048 / ctiles; if (parts > 256) parts = 256; if (parts < 1) parts = 1;   // >= ~2k workgroups on 256 CUs
  const int rl = 256 / CV;
  if (parts > (M + rl - 1) / rl) parts = (M + rl - 1) / rl;
  dim3 grid(ctiles, parts);
  if (dtype == COUNTR_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, STREAM(stream), (const bf16_t*)x, workspace, M, N, CV);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, STREAM(stream), (const float*)x, workspace, M, N, CV);
  return countr_colsum_partials(workspace, out, parts, N, accumulate, stream);
}
