// Sliding-window test path around SupervisedMAE.forward (reference: FSC_test_cross(few-shot).py:326-349, demo_zero.py:49-72): an
// image of height 384 is covered by 384-px windows with stride 128 (the last one snapped to w - 384); every window is an independent
// forward; the per-window densities are stitched back column by column -- columns a previous window already covered become
// old / 2 + new / 2, in window order.  Two data-movement kernels replace the ~70 torch launches of that path per 8 frames:
//   countr_window_gather  cuts the windows of several images straight into the engine's input batch [nw, 3, H, 384]
//   countr_window_blend   stitches the densities of n images of one width (+ per-image sums, the predicted counts x 60)
// fp32 in / out like the reference's tensors; results equal the torch slicing / blending bit for bit (x / 2 is exact).
#include "common.hpp"
#include "../../include/countr_hip.h"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

constexpr int WIN = 384;
constexpr int MAX_WINDOWS = 64, MAX_STARTS = 16;

struct GatherArgs {
  const float* frame[MAX_WINDOWS];   // image of window j: fp32 [3, H, W_j]
  int start[MAX_WINDOWS];            // its first column
  int width[MAX_WINDOWS];
};

// VEC: 4 columns per thread (every source row segment 16-byte aligned); otherwise one
template <bool VEC>
__global__ __launch_bounds__(256) void window_gather_kernel(const GatherArgs a, float* __restrict__ wins, int nw, int H) {
  constexpr int PER = VEC ? WIN / 4 : WIN;
  const int64_t total = (int64_t)nw * 3 * H * PER;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % PER);
    const int64_t row = i / PER;                  // (window, channel, y)
    const int j = (int)(row / (3 * H)), cy = (int)(row - (int64_t)j * 3 * H);
    const float* src = a.frame[j] + (int64_t)cy * a.width[j] + a.start[j];
    if (VEC) *reinterpret_cast<float4*>(wins + row * WIN + c * 4) = *reinterpret_cast<const float4*>(src + c * 4);
    else wins[row * WIN + c] = src[c];
  }
}

struct BlendArgs {
  int start[MAX_STARTS];
};

// one thread per output pixel (image i, row y, column x): the windows that cover x, in order
__global__ __launch_bounds__(256) void window_blend_kernel(const float* __restrict__ outs, const BlendArgs a, int nwin, int H, int W,
                                                           float* __restrict__ dm, float* __restrict__ partial, int blocks_per_image) {
  __shared__ float sm[4];
  const int i = blockIdx.y;
  const int64_t per = (int64_t)H * W;
  const int64_t chunk = (per + blocks_per_image - 1) / blocks_per_image;
  const int64_t p0 = (int64_t)blockIdx.x * chunk, p1 = min(per, p0 + chunk);
  float acc = 0.f;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
    const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
    float v = 0.f;
    int prev = -1;                                  // last column covered so far
    for (int k = 0; k < nwin; ++k) {
      const int s = a.start[k];
      if (x >= s && x < s + WIN) {
        const float o = outs[(((int64_t)i * nwin + k) * H + y) * WIN + (x - s)];
        v = (x <= prev) ? v * 0.5f + o * 0.5f : o;
      }
      prev = s + WIN - 1;
    }
    dm[(int64_t)i * per + p] = v;
    acc += v;
  }
  if (partial) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)i * blocks_per_image + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  }
}

// sums[i] = sum of the image's block partials, fixed order
__global__ __launch_bounds__(64) void window_sum_kernel(const float* __restrict__ partial, float* __restrict__ sums, int blocks_per_image) {
  const int i = blockIdx.x;
  float acc = 0.f;
  for (int b = threadIdx.x; b < blocks_per_image; b += 64) acc += partial[(int64_t)i * blocks_per_image + b];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) sums[i] = acc;
}

}  // namespace

extern "C" int countr_window_blend_blocks(int H, int W) {
  const int64_t per = (int64_t)H * W;
  return (int)max((int64_t)1, min((int64_t)128, per / 2048));
}

extern "C" int countr_window_gather(const void* const* frames, const int* widths, const int* starts, int nw, int H, float* wins, void* stream) {
  if (!frames || !widths || !starts || !wins || nw < 1 || nw > MAX_WINDOWS || H < 1) { countr_set_error("countr_window_gather: bad args (1..64 windows)"); return -1; }
  GatherArgs a;
  bool vec = (((uintptr_t)wins) & 15) == 0;
  for (int j = 0; j < MAX_WINDOWS; ++j) {
    const int s = j < nw ? j : nw - 1;
    if (!frames[s] || starts[s] < 0 || starts[s] + WIN > widths[s]) { countr_set_error("countr_window_gather: window outside its image"); return -1; }
    a.frame[j] = (const float*)frames[s]; a.start[j] = starts[s]; a.width[j] = widths[s];
    if ((((uintptr_t)frames[s]) & 15) || (widths[s] & 3) || (starts[s] & 3)) vec = false;
  }
  const int64_t total = (int64_t)nw * 3 * H * (vec ? WIN / 4 : WIN);
  const int blocks = (int)min((int64_t)65535, (total + 255) / 256);
  if (vec) hipLaunchKernelGGL(window_gather_kernel<true>, dim3(blocks), dim3(256), 0, STREAM(stream), a, wins, nw, H);
  else hipLaunchKernelGGL(window_gather_kernel<false>, dim3(blocks), dim3(256), 0, STREAM(stream), a, wins, nw, H);
  COUNTR_LAUNCH_CHECK("countr_window_gather");
}

extern "C" int countr_window_blend(const float* outs, int n, int nwin, const int* starts, int H, int W, float* dm, float* sums, float* workspace,
                                   void* stream) {
  if (!outs || !starts || !dm || n < 1 || nwin < 1 || nwin > MAX_STARTS || H < 1 || W < WIN || (sums && !workspace)) {
    countr_set_error("countr_window_blend: bad args (1..16 windows per image, W >= 384, sums need the workspace)"); return -1;
  }
  BlendArgs a;
  for (int k = 0; k < MAX_STARTS; ++k) {
    const int s = starts[k < nwin ? k : nwin - 1];
    if (s < 0 || s + WIN > W || (k > 0 && k < nwin && s <= starts[k - 1])) { countr_set_error("countr_window_blend: starts must increase inside the image"); return -1; }
    a.start[k] = s;
  }
  const int bpi = countr_window_blend_blocks(H, W);
  hipLaunchKernelGGL(window_blend_kernel, dim3(bpi, n), dim3(256), 0, STREAM(stream), outs, a, nwin, H, W, dm, sums ? workspace : nullptr, bpi);
  if (sums) hipLaunchKernelGGL(window_sum_kernel, dim3(n), dim3(64), 0, STREAM(stream), workspace, sums, bpi);
  COUNTR_LAUNCH_CHECK("countr_window_blend");
}
