// Does a pure-MFMA wave overlap with a pure-VALU (softmax-like) wave on the SAME SIMD?  One 512-thread workgroup per CU: waves 0-3
// and waves 4-7 share SIMDs 0-3.  Modes: 0 = both halves run the MIXED stream (16 MFMA interleaved with the softmax VALU of a tile),
// 1 = waves 0-3 pure MFMA (16 per iteration), waves 4-7 pure VALU (32 exp, 32 add, 16 cvt_pk, 16 max per iteration), 2 = as 1 but the
// roles swap every iteration (ping-pong with s_barrier between phases), 3 = only MFMA waves work, 4 = only VALU waves work.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_pingpong.hip -o tools/_ubench_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ void mfma16(f32x16_t (&a)[4], bf16x8_t x, bf16x8_t w) {
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a[i & 3], 0, 0, 0);
}
__device__ __forceinline__ void valu_tile(float (&s)[32], float& l0, float& l1, uint32_t (&p)[16], float& mx) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float e0 = __builtin_amdgcn_exp2f(s[2 * i]), e1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
    l0 += e0; l1 += e1;
    const f32x2_t v = {e0, e1};
    p[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    mx = fmaxf(fmaxf(mx, s[2 * i]), s[2 * i + 1]);
  }
}
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, uint64_t* stamps, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16_t a[4] = {};
  bf16x8_t x, w;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); w[i] = (__bf16)(i * 0.01f); }
  float s[32]; for (int i = 0; i < 32; ++i) s[i] = -0.01f * (threadIdx.x & 15) - i * 0.1f;
  float l0 = 0, l1 = 0, mx = -1e30f; uint32_t p[16] = {};
  const bool first = wave < 4;
  __syncthreads();
  const uint64_t c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        a[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a[i & 3], 0, 0, 0);
        const float e0 = __builtin_amdgcn_exp2f(s[2 * i]), e1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
        l0 += e0; l1 += e1;
        const f32x2_t v = {e0, e1};
        p[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
        mx = fmaxf(fmaxf(mx, s[2 * i]), s[2 * i + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 1 || MODE == 3 || MODE == 4) {
      if (first) { if (MODE != 4) mfma16(a, x, w); }
      else { if (MODE != 3) valu_tile(s, l0, l1, p, mx); }
    } else {
      const bool m = ((it & 1) == 0) == first;
      if (m) mfma16(a, x, w); else valu_tile(s, l0, l1, p, mx);
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) s[i] += 1e-6f * (float)(p[i & 15] & 1);   // keep the VALU chain alive across iterations
  }
  const uint64_t c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) stamps[blockIdx.x] = c1 - c0;
  out[blockIdx.x * 512 + threadIdx.x] = a[0][0] + a[1][1] + a[2][2] + a[3][3] + l0 + l1 + mx + (float)p[3];
}
int main() {
  float* out; uint64_t* st; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 8);
  const int iters = 2000;
  auto run = [&](auto kern, const char* name, int blocks) {
    kern<<<blocks, 512>>>(out, st, 10); hipDeviceSynchronize();
    kern<<<blocks, 512>>>(out, st, iters); hipDeviceSynchronize();
    uint64_t h[256]; hipMemcpy(h, st, blocks * 8, hipMemcpyDeviceToHost);
    printf("%-60s blocks %3d: %7.1f cycles per iteration\n", name, blocks, (double)h[0] / iters);
  };
  for (int blocks : {1, 256}) {
    run(k<0>, "0 both halves mixed (16 MFMA + tile VALU interleaved)", blocks);
    run(k<1>, "1 waves 0-3 pure MFMA (16), waves 4-7 pure VALU (tile)", blocks);
    run(k<2>, "2 ping-pong: roles swap every iteration, s_barrier", blocks);
    run(k<3>, "3 only the MFMA waves work", blocks);
    run(k<4>, "4 only the VALU waves work", blocks);
  }
  return 0;
}
