// Issue-rate microbenchmark for gfx950 (build: hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/_ubench_issue).
// One workgroup on one CU, W waves (4 = one per SIMD, 8 = two per SIMD, 12, 16); every wave runs the same straight-line
// block of instructions REPS times; reports shader cycles (s_memtime) per block for the slowest wave.  Used to price the
// attention inner loop: how many VALU / transcendental instructions fit beside one v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

#define REPS 200

// MODE: what one block contains
//  0: 32 independent v_fma_f32                      1: 32 v_exp_f32
//  2: 8 MFMA 32x32x16 (2 accumulators alternating)  3: 8 x {MFMA, 4 fma}        4: 8 x {MFMA, 8 fma}
//  5: 8 x {MFMA, 6 fma, 2 exp}                      6: 8 x {MFMA, 12 fma}       7: 8 x {MFMA, 8 fma, 2 exp, 1 ds_read_b128}
//  8: 32 v_add_f32 dependent chain                  9: 8 x {MFMA 16x16x32 x2, 8 fma}
// 10: 32 ds_read_b128                               11: 8 x {MFMA, 4 fma, 2 exp, 2 add, 1 cvt_pk}  (the attention mix per MFMA)
// 12: 16 v_pk_fma_f32                               13: 32 v_cvt_pk_bf16_f32
template <int MODE>
__global__ void ubench(float* out, uint64_t* cyc, float seed) {
  extern __shared__ char smem[];
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
  f32x16_t acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  bf16x8_t fa, fb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed * 2 + i); }
  typedef __attribute__((ext_vector_type(4))) float f32x4v;
  f32x4v ld = {0, 0, 0, 0};
  const char* lp = smem + (threadIdx.x & 63) * 16;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int r = 0; r < REPS; ++r) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed))
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]))
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed))
#define CVT(i, j) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[j]))
#define MF(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0)
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 32; ++k) FMA(k & 15);
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 32; ++k) EXP(k & 15);
    } else if (MODE == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { MF(acc0); MF(acc1); }
    } else if (MODE == 3 || MODE == 4 || MODE == 6) {
      constexpr int NF = MODE == 3 ? 4 : (MODE == 4 ? 8 : 12);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k & 1) MF(acc1); else MF(acc0);
#pragma unroll
        for (int j = 0; j < NF; ++j) FMA(j);
      }
    } else if (MODE == 5) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k & 1) MF(acc1); else MF(acc0);
#pragma unroll
        for (int j = 0; j < 6; ++j) FMA(j);
        EXP(6); EXP(7);
      }
    } else if (MODE == 7) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k & 1) MF(acc1); else MF(acc0);
        asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"((unsigned)(uintptr_t)lp));
#pragma unroll
        for (int j = 0; j < 8; ++j) FMA(j);
        EXP(8); EXP(9);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld));
      }
    } else if (MODE == 8) {
#pragma unroll
      for (int k = 0; k < 32; ++k) ADD(0);
    } else if (MODE == 9) {
      typedef __attribute__((ext_vector_type(4))) float f32x4_t;
      f32x4_t c0 = {acc0[0], acc0[1], acc0[2], acc0[3]}, c1 = {acc1[0], acc1[1], acc1[2], acc1[3]};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) FMA(j);
      }
      acc0[0] = c0[0]; acc1[0] = c1[0];
    } else if (MODE == 10) {
#pragma unroll
      for (int k = 0; k < 32; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"((unsigned)(uintptr_t)lp));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld));
    } else if (MODE == 11) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k & 1) MF(acc1); else MF(acc0);
        FMA(0); FMA(1); FMA(2); FMA(3);
        EXP(0); EXP(1);
        ADD(4); ADD(5);
        CVT(6, 7);
      }
    } else if (MODE == 12) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&a[(k & 7) * 2]) : "v"(*(double*)&a[0]));
    } else if (MODE == 13) {
#pragma unroll
      for (int k = 0; k < 32; ++k) CVT(k & 15, (k + 1) & 15);
    } else if (MODE >= 20 && MODE <= 24) {   // {MFMA, NF fillers of one kind}
      constexpr int NF = (MODE == 20 || MODE == 22) ? 8 : (MODE == 24 ? 16 : 12);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k & 1) MF(acc1); else MF(acc0);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          if (MODE <= 21) asm volatile("v_mov_b32 %0, %1" : "=v"(a[j & 15]) : "v"(seed));
          else if (MODE <= 23) ADD(j & 15);
          else FMA(j & 15);
        }
      }
    } else if (MODE == 25) {   // accumulator restarted from 0 every time (no SrcC read)
      f32x16_t z;
#pragma unroll
      for (int i = 0; i < 16; ++i) z[i] = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, z, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, z, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) FMA(j);
      }
    } else if (MODE == 26) {   // coarse grouping: 4 MFMA then 32 fma, twice
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        MF(acc0); MF(acc1); MF(acc0); MF(acc1);
#pragma unroll
        for (int j = 0; j < 32; ++j) FMA(j & 15);
      }
    } else if (MODE == 27) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k & 1) MF(acc1); else MF(acc0);
        EXP(0); EXP(1); EXP(2); EXP(3);
      }
    } else if (MODE == 28) {
#pragma unroll
      for (int k = 0; k < 32; ++k) asm volatile("v_exp_f16 %0, %0" : "+v"(a[k & 15]));
    } else if (MODE == 29) {   // 32 ds_read_b64_tr_b16
      typedef __attribute__((ext_vector_type(2))) float f32x2v;
      f32x2v l2 = {0, 0};
#pragma unroll
      for (int k = 0; k < 32; ++k) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(l2) : "v"((unsigned)(uintptr_t)(smem + (threadIdx.x & 63) * 8)));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l2));
      ld[0] += l2[0];
    } else if (MODE == 30) {   // 16 MFMA 16x16x32 (4 accumulators)
      typedef __attribute__((ext_vector_type(4))) float f32x4_t;
      f32x4_t c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = f32x4_t{acc0[i], acc0[i + 4], acc1[i], acc1[i + 4]};
#pragma unroll
      for (int k = 0; k < 16; ++k) c[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c[k & 3], 0, 0, 0);
      acc0[0] = c[0][0] + c[1][0]; acc1[0] = c[2][0] + c[3][0];
    } else if (MODE == 31) {   // the attention mix with the add/max work on packed ops: {MFMA, 2 fma, 2 exp, 1 pk_add, 1 max3, 1 cvt}
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k & 1) MF(acc1); else MF(acc0);
        FMA(0); FMA(1);
        EXP(0); EXP(1);
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[4]) : "v"(*(double*)&a[0]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[6]) : "v"(a[2]), "v"(a[3]));
        CVT(8, 9);
      }
    } else if (MODE == 32) {   // 8 barriers
#pragma unroll
      for (int k = 0; k < 8; ++k) __builtin_amdgcn_s_barrier();
    } else if (MODE == 33) {   // v_perm_b32 packing (truncating bf16 pack)
#pragma unroll
      for (int k = 0; k < 32; ++k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k & 15]) : "v"(a[(k + 1) & 15]), "v"(seed));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = ld[0] + ld[1];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE> void run(const char* what, int per_block) {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
  printf("%-58s", what);
  for (int waves : {4, 8, 12, 16}) {
    hipLaunchKernelGGL(ubench<MODE>, dim3(1), dim3(64 * waves), 16384, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL(ubench<MODE>, dim3(1), dim3(64 * waves), 16384, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    uint64_t h[64]; hipMemcpy(h, cyc, sizeof(uint64_t) * waves, hipMemcpyDeviceToHost);
    uint64_t mx = 0; for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("  W%-2d %7.1f", waves, (double)mx / REPS);
  }
  printf("   cycles per block of %d\n", per_block);
  hipFree(out); hipFree(cyc);
}

int main() {
  // s_memtime counts at a constant 100 MHz on some parts: calibrate against the wall clock with a long kernel
  run<0>("0: 32 independent v_fma_f32", 32);
  run<1>("1: 32 v_exp_f32", 32);
  run<8>("8: 32 dependent v_add_f32", 32);
  run<12>("12: 16 v_pk_fma_f32", 16);
  run<13>("13: 32 v_cvt_pk_bf16_f32", 32);
  run<2>("2: 8 MFMA 32x32x16", 8);
  run<3>("3: 8 x {MFMA, 4 fma}", 8);
  run<4>("4: 8 x {MFMA, 8 fma}", 8);
  run<6>("6: 8 x {MFMA, 12 fma}", 8);
  run<5>("5: 8 x {MFMA, 6 fma, 2 exp}", 8);
  run<11>("11: 8 x {MFMA, 4 fma, 2 exp, 2 add, 1 cvt} (attention mix)", 8);
  run<7>("7: 8 x {MFMA, ds_read_b128, 8 fma, 2 exp, wait}", 8);
  run<9>("9: 8 x {2 MFMA 16x16x32, 8 fma}", 8);
  run<10>("10: 32 ds_read_b128 + wait", 32);
  run<29>("29: 32 ds_read_b64_tr_b16 + wait", 32);
  run<20>("20: 8 x {MFMA, 8 v_mov}", 8);
  run<21>("21: 8 x {MFMA, 12 v_mov}", 8);
  run<22>("22: 8 x {MFMA, 8 v_add}", 8);
  run<23>("23: 8 x {MFMA, 12 v_add}", 8);
  run<24>("24: 8 x {MFMA, 16 fma}", 8);
  run<25>("25: 8 x {MFMA C=0, 8 fma}", 8);
  run<26>("26: 2 x {4 MFMA, 32 fma}", 8);
  run<27>("27: 8 x {MFMA, 4 exp}", 8);
  run<28>("28: 32 v_exp_f16", 32);
  run<30>("30: 16 MFMA 16x16x32", 16);
  run<31>("31: 8 x {MFMA, 2 fma, 2 exp, pk_add, max3, cvt}", 8);
  run<32>("32: 8 s_barrier", 8);
  run<33>("33: 32 v_perm_b32", 32);
  return 0;
}
