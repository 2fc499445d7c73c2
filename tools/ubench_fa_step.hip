// Replica of one pipelined attention step's COMPUTE (csrc/flash_attn_fwd.hip, dh = 64, pre-scaled q) without any memory traffic: 8 QK^T
// MFMAs into Sn[2], 8 PV MFMAs into o[2] with P built from the previous Sn, the exp / row-sum / pack units in the kernel's slot schedule,
// the row max of Sn and the (never taken) rescale test.  512-thread workgroup = two waves per SIMD, one workgroup per CU.  VARIANT bits:
// 1 = no sched_barrier pins (the compiler interleaves), 2 = no row max / rescale test, 4 = no exp units, 8 = exp -> plain multiply.
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench_fa_step.hip -o tools/_ubench_fa_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { const f32x2_t v = {lo, hi}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
constexpr int unit_end[16] = {3, 4, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 16, 16, 16, 16};

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void k(float* out, uint64_t* stamps, int iters) {
  bf16x8_t qf[4], kf, kf1;
  for (int i = 0; i < 8; ++i) { kf[i] = (__bf16)(i * 0.01f + threadIdx.x * 1e-4f); kf1[i] = (__bf16)(i * 0.02f - threadIdx.x * 1e-4f); for (int s = 0; s < 4; ++s) qf[s][i] = (__bf16)(0.02f * i - 0.01f * s); }
  f32x16_t SA[2], SB[2], o[2] = {}, negm;
  for (int i = 0; i < 16; ++i) { SA[0][i] = -0.1f * i; SA[1][i] = -0.05f * i - 1.f; negm[i] = -2.f; }
  float l0 = 0, l1 = 0, mref = 0;
  uint32_t P[2][8] = {};
  auto step = [&](f32x16_t (&Sc)[2], f32x16_t (&Sn)[2], int it) {
    float mx = 0.f;
    { u32x4_t t = __builtin_bit_cast(u32x4_t, kf); t[0] = 0x3c003c00u + (unsigned)(it & 3); kf = __builtin_bit_cast(bf16x8_t, t); t = __builtin_bit_cast(u32x4_t, kf1); t[1] = 0x3c003c00u + (unsigned)(it & 3); kf1 = __builtin_bit_cast(bf16x8_t, t); }   // (keeps the QK^T MFMAs loop-variant)
    auto exp_unit = [&](int u) {
      const int blk = u >> 3, w = u & 7;
      const float p0 = (VARIANT & 8) ? Sc[blk][2 * w] * 1.0001f : __builtin_amdgcn_exp2f(Sc[blk][2 * w]), p1 = (VARIANT & 8) ? Sc[blk][2 * w + 1] * 0.9999f : __builtin_amdgcn_exp2f(Sc[blk][2 * w + 1]);
      l0 += p0; l1 += p1;
      P[blk][w] = pack2bf(p0, p1);
    };
    if (!(VARIANT & 4)) { exp_unit(0); exp_unit(1); }
    if (!(VARIANT & 1)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < 8) {
        const int blk = j & 1, ks = j >> 1;
        Sn[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(blk ? kf1 : kf, qf[ks], ks == 0 ? negm : Sn[blk], 0, 0, 0);
      } else {
        const int e = j - 8, blk = e / 4, s = (e / 2) & 1, d = e % 2;
        const u32x4_t pv = {P[blk][4 * s], P[blk][4 * s + 1], P[blk][4 * s + 2], P[blk][4 * s + 3]};
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d ? kf1 : kf, __builtin_bit_cast(bf16x8_t, pv), o[d], 0, 0, 0);
      }
      if (!(VARIANT & 4)) {
        const int u0 = (j == 0) ? 2 : unit_end[j - 1], u1 = unit_end[j];
        for (int u = u0; u < u1; ++u) exp_unit(u);
      }
      if (j >= 12 && !(VARIANT & 2)) {
        const int r0 = (j - 12) * 4;
        for (int r = r0; r < r0 + 4; ++r) mx = (r == 0) ? fmaxf(Sn[0][0], Sn[1][0]) : max3(mx, Sn[0][r], Sn[1][r]);
      }
      if (!(VARIANT & 1)) __builtin_amdgcn_sched_barrier(0);
    }
    if (!(VARIANT & 2)) {
      if (!__all(mx <= 8.0f)) {   // never taken with these inputs
        const float alpha = __builtin_amdgcn_exp2f(-mx);
        mref += mx; l0 *= alpha; l1 *= alpha;
        for (int d = 0; d < 2; ++d) for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
        for (int i = 0; i < 16; ++i) negm[i] = -mref;
      }
    }
  };
  __syncthreads();
  const uint64_t c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) { step(SA, SB, 2 * it); step(SB, SA, 2 * it + 1); }
  const uint64_t c1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + (threadIdx.x >> 6)] = c1 - c0;   // every wave: the SIMD's pace is its slower wave's
  out[blockIdx.x * 512 + threadIdx.x] = o[0][0] + o[1][1] + l0 + l1 + SA[0][3] + SB[1][2];
}
int main() {
  float* out; uint64_t* st; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 8 * 8);
  const int iters = 1000;
  auto run = [&](auto kern, const char* name) {
    for (int blocks : {1, 256}) {
      kern<<<blocks, 512>>>(out, st, 10); hipDeviceSynchronize();
      kern<<<blocks, 512>>>(out, st, iters); hipDeviceSynchronize();
      uint64_t h[2048]; hipMemcpy(h, st, blocks * 64, hipMemcpyDeviceToHost);
      uint64_t mn = ~0ull, mxv = 0; for (int i = 0; i < blocks * 8; ++i) { mn = h[i] < mn ? h[i] : mn; mxv = h[i] > mxv ? h[i] : mxv; }
      printf("%-58s blocks %3d: fastest wave %7.1f, slowest wave %7.1f cycles per step (two waves per SIMD, 16 MFMAs each: matrix bound 1024)\n", name, blocks,
             (double)mn / iters / 2, (double)mxv / iters / 2);
    }
  };
  run(k<0>, "0 the kernel's step (slot order pinned)");
  run(k<1>, "1 no sched_barrier pins");
  run(k<2>, "2 no row max / rescale test");
  run(k<4>, "4 no exp units");
  run(k<6>, "6 MFMAs only");
  run(k<8>, "8 v_exp_f32 replaced by v_mul_f32 (same instruction count)");
  run(k<9>, "9 = 8 without the sched_barrier pins");
  return 0;
}
