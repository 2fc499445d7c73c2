// What clock does the chip run at under matrix load?  One workgroup per CU (NW waves), each wave issues N independent-accumulator
// v_mfma_f32_32x32x16_bf16 back to back; s_memtime and s_memrealtime (100 MHz) bracket the loop, HIP events bracket the launch.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_clock.hip -o tools/_ubench_clock && tools/_ubench_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ void k(float* out, uint64_t* stamps, int iters) {
  f32x16_t a0 = {}, a1 = {}, a2 = {}, a3 = {};
  bf16x8_t x, w;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); w[i] = (__bf16)(i * 0.01f); }
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a3, 0, 0, 0);
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
  float* out; uint64_t* st; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 256 * 16);
  for (int nw : {4, 8}) for (int blocks : {1, 32, 256}) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 64 * nw>>>(out, st, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<<<blocks, 64 * nw>>>(out, st, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[512]; hipMemcpy(h, st, blocks * 16, hipMemcpyDeviceToHost);
    const double n_mfma = 4.0 * iters * nw * blocks;                      // per launch
    const double cyc_per_simd = 4.0 * iters * (nw / 4) * 32;              // matrix cycles one SIMD needs
    printf("waves/WG %d blocks %3d: %.3f ms  -> %.0f TF/s, implied matrix clock %.3f GHz | s_memtime delta %llu, s_memrealtime delta %llu (x10 ns = %.3f ms) -> s_memtime rate %.1f MHz\n",
           nw, blocks, ms, n_mfma * 32768 / ms / 1e9, cyc_per_simd / ms / 1e6, (unsigned long long)h[0], (unsigned long long)h[1], h[1] * 1e-5, h[0] / (h[1] * 1e-2));
  }
  return 0;
}
