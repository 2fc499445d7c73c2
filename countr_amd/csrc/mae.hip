// MAE-pretraining specific kernels (reference models_mae_noct.py): row gather used for random masking / unshuffle and
// their backward (:110-135, :163-170), and the all-patch pixel MSE with on-the-fly patchify (:84-96, :181-198).
#include "common.hpp"
#include "../../include/countr_hip.h"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

// dst[r,:] = (idx[r] >= 0 ? src[idx[r],:] : default_row[:]) + add[r % add_mod,:]     (4 columns per thread)
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void gather_rows_kernel(const TS* __restrict__ src, const int* __restrict__ idx, TD* __restrict__ dst,
                                                          const float* __restrict__ default_row, const float* __restrict__ add,
                                                          int add_mod, int rows, int cols) {
  const int c4 = cols >> 2;
  const int64_t total = (int64_t)rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / c4), c = (int)(i % c4) * 4;
    const int s = idx ? idx[r] : r;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (s >= 0) ld4<TS>(src + (int64_t)s * cols + c, v);
    else if (default_row) ld4<float>(default_row + c, v);
    if (add) {
      float a[4];
      ld4<float>(add + (int64_t)(add_mod > 0 ? r % add_mod : r) * cols + c, a);
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
    }
    st4<TD>(dst + (int64_t)r * cols + c, v);
  }
}

// one block per patch: target = patchify(imgs)[row] in (py, px, c) feature order, optional per-patch normalisation
// (unbiased variance, eps 1e-6), partial[row] = sum_e (pred - target)^2, dpred = 2 * gscale * (pred - target) / (F * rows)
template <typename TD>
__global__ __launch_bounds__(256) void patch_mse_kernel(const float* __restrict__ pred, const float* __restrict__ imgs, TD* __restrict__ dpred,
                                                        float* __restrict__ partial, int rows, int H, int W, int p, int norm_pix,
                                                        float gscale, const float* __restrict__ amp) {
  if (amp) gscale *= amp[0];      // dynamic loss scale (fp16 mode: countr_amp_*, elementwise.hip)
  __shared__ float sm[4];
  const int row = blockIdx.x, gw = W / p, L = (H / p) * gw, F = 3 * p * p;
  const int b = row / L, l = row % L, ph = l / gw, pw = l % gw;
  const float* img = imgs + (int64_t)b * 3 * H * W + (int64_t)(ph * p) * W + pw * p;
  float t[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = threadIdx.x + k * 256;
    t[k] = 0.f;
    if (e < F) {
      const int c = e % 3, px = (e / 3) % p, py = e / (3 * p);
      t[k] = img[(int64_t)c * H * W + py * W + px];
      s += t[k];
    }
  }
  if (norm_pix) {
    const float mean = block_sum<4>(s, sm) / (float)F;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (threadIdx.x + k * 256 < F) { t[k] -= mean; q += t[k] * t[k]; }
    const float var = block_sum<4>(q, sm) / (float)(F - 1);
    const float rs = 1.f / sqrtf(var + 1e-6f);
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] *= rs;
  }
  const float gs = 2.f * gscale / ((float)F * (float)rows);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < F) {
      const float d = pred[(int64_t)row * F + e] - t[k];
      acc += d * d;
      if (dpred) stf<TD>(dpred + (int64_t)row * F + e, d * gs);
    }
  }
  acc = block_sum<4>(acc, sm);
  if (threadIdx.x == 0) partial[row] = acc;
}

__global__ __launch_bounds__(256) void patch_mse_finish_kernel(const float* __restrict__ partial, float* __restrict__ loss, int rows, float inv) {
  __shared__ float sm[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < rows; i += 256) a += partial[i];
  a = block_sum<4>(a, sm);
  if (threadIdx.x == 0) loss[0] = a * inv;
}

// All index buffers of random_masking (models_mae_noct.py:119-132) from ids_shuffle [B][N] (the argsort of the noise) in one launch:
// thread (b, j) with s = ids_shuffle[b][j]: ids_restore[b][s] = j (the inverse permutation = argsort(ids_shuffle), :122); kept tokens
// (j < K): keep_pos = s, keep_src = s + b N, restore_src[b N + s] = j + b K, mask = 0; masked ones: mask_src[b (N-K) + j-K] = s + b N,
// restore_src[b N + s] = -1 (-> mask_token in the unshuffle gather), mask = 1.
__global__ __launch_bounds__(256) void mae_indices_kernel(const long long* __restrict__ ids_shuffle, long long* __restrict__ ids_restore,
                                                          int* __restrict__ keep_pos, int* __restrict__ keep_src,
                                                          int* __restrict__ restore_src, int* __restrict__ mask_src,
                                                          float* __restrict__ mask, int B, int N, int K) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * N) return;
  const int b = i / N, j = i - b * N;
  const int s = (int)ids_shuffle[i];
  ids_restore[b * N + s] = j;
  if (j < K) {
    keep_pos[b * K + j] = s;
    keep_src[b * K + j] = s + b * N;
    restore_src[b * N + s] = j + b * K;
    mask[b * N + s] = 0.f;
  } else {
    mask_src[b * (N - K) + j - K] = s + b * N;
    restore_src[b * N + s] = -1;
    mask[b * N + s] = 1.f;
  }
}

extern "C" int countr_mae_indices(const long long* ids_shuffle, long long* ids_restore, int* keep_pos, int* keep_src, int* restore_src,
                                  int* mask_src, float* mask, int B, int N, int K, void* stream) {
  if (!ids_shuffle || !ids_restore || !keep_pos || !keep_src || !restore_src || !mask || B < 1 || N < 1 || K < 1 || K > N || (K < N && !mask_src)) {
    countr_set_error("countr_mae_indices: bad args"); return -1;
  }
  hipLaunchKernelGGL(mae_indices_kernel, dim3((B * N + 255) / 256), dim3(256), 0, STREAM(stream), ids_shuffle, ids_restore, keep_pos,
                     keep_src, restore_src, mask_src, mask, B, N, K);
  COUNTR_LAUNCH_CHECK("countr_mae_indices");
}

extern "C" int countr_gather_rows(const void* src, const int* idx, void* dst, const float* default_row, const float* add, int add_mod,
                                  int rows, int cols, int src_dtype, int dst_dtype, void* stream) {
  if (!src || !dst || rows < 1 || cols < 4 || (cols & 3)) { countr_set_error("countr_gather_rows: null pointer or cols not a multiple of 4"); return -1; }
  const int64_t total = (int64_t)rows * (cols / 4);
  int nb = (int)((total + 255) / 256);
  if (nb > 8192) nb = 8192;
#define GR(TS, TD) hipLaunchKernelGGL((gather_rows_kernel<TS, TD>), dim3(nb), dim3(256), 0, STREAM(stream), (const TS*)src, idx, (TD*)dst, default_row, add, add_mod, rows, cols)
  if (src_dtype == COUNTR_BF16 && dst_dtype == COUNTR_BF16) GR(bf16_t, bf16_t);
  else if (src_dtype == COUNTR_BF16) GR(bf16_t, float);
  else if (dst_dtype == COUNTR_BF16) GR(float, bf16_t);
  else GR(float, float);
#undef GR
  COUNTR_LAUNCH_CHECK("countr_gather_rows");
}

extern "C" int countr_patch_mse_workspace_floats(int B, int H, int W, int patch) { return B * (H / patch) * (W / patch); }
extern "C" int countr_patch_mse_amp(const float* pred, const float* imgs, void* dpred, float* loss, float* workspace, int B, int H, int W,
                                    int patch, int norm_pix, float grad_scale, int dpred_dtype, const float* amp, void* stream) {
  if (!pred || !imgs || !loss || !workspace || patch < 1 || H % patch || W % patch || 3 * patch * patch > 1024) {
    countr_set_error("countr_patch_mse: bad args (H, W multiples of patch; 3*patch^2 <= 1024)"); return -1;
  }
  const int rows = B * (H / patch) * (W / patch), F = 3 * patch * patch;
  if (dpred_dtype == COUNTR_BF16)
    hipLaunchKernelGGL(patch_mse_kernel<bf16_t>, dim3(rows), dim3(256), 0, STREAM(stream), pred, imgs, (bf16_t*)dpred, workspace, rows, H, W, patch, norm_pix, grad_scale, amp);
  else
    hipLaunchKernelGGL(patch_mse_kernel<float>, dim3(rows), dim3(256), 0, STREAM(stream), pred, imgs, (float*)dpred, workspace, rows, H, W, patch, norm_pix, grad_scale, amp);
  hipLaunchKernelGGL(patch_mse_finish_kernel, dim3(1), dim3(256), 0, STREAM(stream), workspace, loss, rows, 1.f / ((float)F * (float)rows));
  COUNTR_LAUNCH_CHECK("countr_patch_mse");
}
extern "C" int countr_patch_mse(const float* pred, const float* imgs, void* dpred, float* loss, float* workspace, int B, int H, int W,
                                int patch, int norm_pix, float grad_scale, int dpred_dtype, void* stream) {
  return countr_patch_mse_amp(pred, imgs, dpred, loss, workspace, B, H, W, patch, norm_pix, grad_scale, dpred_dtype, nullptr, stream);
}
