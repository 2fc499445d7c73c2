// Generic MFMA GEMM for gfx950 (MI355X).  Operands are addressed through four modes (row / col / 3x3-im2col-row / 3x3-im2col-col) so
// that nn.Linear forward, dgrad and wgrad, the unfused attention products and the 3x3 convolutions (fwd, dgrad, wgrad) all run here.
//   bf16: v_mfma_f32_16x16x32_bf16; tiles go global -> LDS by LDS-DMA (global_load_lds, XOR swizzle on the source address),
//         fragments come back with ds_read_b128 / ds_read_b64_tr_b16 issued from inline asm (counted lgkmcnt / vmcnt, one barrier
//         per k-tile).  Variants (launch()): 128x128 tile, 4 waves, 2 LDS stages, 2 workgroups per CU (big grids); the same tile
//         wave-specialised -- 4 compute + 4 loader waves on a 3-stage ring -- for grids of <= 256 tiles; 128x256 with 8 + 4 waves
//         for the 192x192 convolutions.  Epilogues (bias / GELU / fp32 residual / bf16 or fp32 out / pre-activation copy) are
//         compile-time variants staged through LDS into whole-row-segment stores.
//   fp32: v_mfma_f32_16x16x4_f32 (exact f32 fma chain), register-staged padded tiles -- the parity mode.
// Reference call sites: models_crossvit.py:62,65,84-92,115-127; models_mae_cross.py:47-100,138,152.
#include "common.hpp"
#include <type_traits>
#include <utility>
#include "../../include/countr_hip.h"
#include <stdlib.h>

// the kernel template lives in gemm_kernel.hpp; its instantiations in gemm_bf16_a.hip ((ROW, ROW), (ROW, COL), (COL, COL)),
// gemm_bf16_b.hip ((COL, ROW), (IM2ROW, ROW), (COL, IM2COL)) and gemm_f32.hip (all six, parity mode).  1 = not one of mine.
int countr_gemm_bf16_a(const countr_gemm_args& a, int ma, int mb, hipStream_t s);
int countr_gemm_bf16_b(const countr_gemm_args& a, int ma, int mb, hipStream_t s);
int countr_gemm_f32(const countr_gemm_args& a, int ma, int mb, hipStream_t s);

namespace {

int dispatch_bf16(const countr_gemm_args& a, int ma, int mb, hipStream_t s) {
  int rc = countr_gemm_bf16_a(a, ma, mb, s);
  if (rc == 1) rc = countr_gemm_bf16_b(a, ma, mb, s);
  if (rc == 1) { countr_set_error("countr_gemm: unsupported operand mode combination"); return -2; }
  return rc;
}
int dispatch_f32(const countr_gemm_args& a, int ma, int mb, hipStream_t s) {
  const int rc = countr_gemm_f32(a, ma, mb, s);
  if (rc == 1) { countr_set_error("countr_gemm: unsupported operand mode combination"); return -2; }
  return rc;
}

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int splitk,
                                     int64_t MN, int N, int taps, int accumulate, const float* __restrict__ rowsum_partial,
                                     float* __restrict__ rowsum_out, int M) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (rowsum_partial && i < M) {  // fused bias gradient: out_b[m] = sum_z rowsum_partial[z][m]
    float b = 0.f;
    for (int z = 0; z < splitk; ++z) b += rowsum_partial[(int64_t)z * M + i];
    rowsum_out[i] = accumulate ? rowsum_out[i] + b : b;
  }
  if (i >= MN) return;
  float s = 0.f;
  for (int z = 0; z < splitk; ++z) s += partial[(int64_t)z * MN + i];
  int64_t o = i;
  if (taps > 0) {  // [co][tap][ci] -> [co][ci][tap]
    const int cin = N / taps;
    const int64_t co = i / N;
    const int r = (int)(i - co * N);
    const int tap = r / cin, ci = r - tap * cin;
    o = co * N + (int64_t)ci * taps + tap;
  }
  out[o] = accumulate ? out[o] + s : s;
}

// Batched deferred reductions (one launch for many (partial slabs -> gradient) sums): table row e = {partial, out,
// nslabs | accumulate << 32 | wide << 33 | vec4 << 34, slab stride, count, N, taps, first block}, followed by the int32 map block -> entry.
__global__ void reduce_table_kernel(const long long* __restrict__ tab, int n) {
  // block -> entry map (int32 [total_blocks]) behind the n rows: one load instead of a scan over the first-block column, which
  // was a chain of up to n dependent scalar loads in every block (~10 us of the launch at n = 40)
  const int e = reinterpret_cast<const int*>(tab + (long long)n * 8)[blockIdx.x];
  const long long* t = tab + (long long)e * 8;
  const float* __restrict__ partial = reinterpret_cast<const float*>(t[0]);
  float* __restrict__ out = reinterpret_cast<float*>(t[1]);
  const int nslabs = (int)(t[2] & 0xffffffffll), accumulate = (int)((t[2] >> 32) & 1), wide = (int)((t[2] >> 33) & 1);
  const long long stride = t[3], count = t[4];
  const int N = (int)t[5], taps = (int)t[6];
  if (wide) {   // many slabs (LayerNorm block partials): 16 columns x 16 slab groups per block, LDS combine (no permutation)
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, gq = threadIdx.x >> 4;
    const long long col = ((long long)blockIdx.x - t[7]) * 16 + c;
    float s = 0.f;
    if (col < count)
      for (int z = gq; z < nslabs; z += 16) s += partial[(long long)z * stride + col];
    red[gq][c] = s;
    __syncthreads();
    if (gq == 0 && col < count) {
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += red[k][c];
      out[col] = accumulate ? out[col] + tot : tot;
    }
    return;
  }
  if ((t[2] >> 34) & 1) {   // count, stride, both pointers multiples of 4 floats: 16-byte loads, four slabs in flight per thread
    const long long i = (((long long)blockIdx.x - t[7]) * blockDim.x + threadIdx.x) * 4;
    if (i >= count) return;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    const float* __restrict__ p = partial + i;
    int z = 0;
#define COUNTR_ACC4(a, v) { const float4 _v = (v); a.x += _v.x; a.y += _v.y; a.z += _v.z; a.w += _v.w; }
    for (; z + 4 <= nslabs; z += 4) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (long long)z * stride);
      const float4 v1 = *reinterpret_cast<const float4*>(p + (long long)(z + 1) * stride);
      const float4 v2 = *reinterpret_cast<const float4*>(p + (long long)(z + 2) * stride);
      const float4 v3 = *reinterpret_cast<const float4*>(p + (long long)(z + 3) * stride);
      COUNTR_ACC4(a0, v0) COUNTR_ACC4(a1, v1) COUNTR_ACC4(a2, v2) COUNTR_ACC4(a3, v3)
    }
    for (; z < nslabs; ++z) COUNTR_ACC4(a0, *reinterpret_cast<const float4*>(p + (long long)z * stride))
#undef COUNTR_ACC4
    // same association as the scalar path: (s0 + s1) + (s2 + s3)
    float4 r = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    if (taps > 0) {  // [co][tap][ci] -> [co][ci][tap]; Cin % 4 == 0: the four elements share (co, tap), ci consecutive
      const int cin = N / taps;
      const long long co = i / N;
      const int rr = (int)(i - co * N);
      const int tap = rr / cin, ci = rr - tap * cin;
      float* o = out + co * N + (long long)ci * taps + tap;
      if (accumulate) { o[0] += r.x; o[taps] += r.y; o[2 * taps] += r.z; o[3 * taps] += r.w; }
      else { o[0] = r.x; o[taps] = r.y; o[2 * taps] = r.z; o[3 * taps] = r.w; }
    } else {
      float4* o = reinterpret_cast<float4*>(out + i);
      if (accumulate) { const float4 q = *o; r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w; }
      *o = r;
    }
    return;
  }
  const long long i = ((long long)blockIdx.x - t[7]) * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int z = 0;
  for (; z + 4 <= nslabs; z += 4) {
    s0 += partial[(long long)z * stride + i];
    s1 += partial[(long long)(z + 1) * stride + i];
    s2 += partial[(long long)(z + 2) * stride + i];
    s3 += partial[(long long)(z + 3) * stride + i];
  }
  for (; z < nslabs; ++z) s0 += partial[(long long)z * stride + i];
  const float s = (s0 + s1) + (s2 + s3);
  long long o = i;
  if (taps > 0) {  // [co][tap][ci] -> [co][ci][tap]
    const int cin = N / taps;
    const long long co = i / N;
    const int r = (int)(i - co * N);
    const int tap = r / cin, ci = r - tap * cin;
    o = co * N + (long long)ci * taps + tap;
  }
  out[o] = accumulate ? out[o] + s : s;
}

}  // namespace

extern "C" int countr_reduce_table(const long long* table, int n, int total_blocks, void* stream) {
  if (!table || n <= 0 || total_blocks <= 0) { countr_set_error("countr_reduce_table: bad args"); return -1; }
  hipLaunchKernelGGL(reduce_table_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table, n);
  COUNTR_LAUNCH_CHECK("countr_reduce_table");
}

int countr_lean_linear(const countr_gemm_args* a, hipStream_t s);   // linear.hip: 1 = does not qualify
int countr_lean_conv(const countr_gemm_args* a, hipStream_t s);
int countr_big_linear(const countr_gemm_args* a, hipStream_t s);    // gemm256.hip: 256 x 256 tiles, 8-phase schedule; 1 = does not qualify
int countr_big_conv(const countr_gemm_args* a, hipStream_t s);
int countr_lean_wgrad(const countr_gemm_args* a, int lin, hipStream_t s);      // conv_wgrad.hip
int countr_lean_wgrad_rowsum_slabs(const countr_gemm_args* a, int lin);

int countr_lean_wgrad_tiles(const countr_gemm_args* a, int lin);

extern "C" int countr_gemm_tiles(const countr_gemm_args* a, int dtype, int modeA, int modeB) {
  if (!a) return 0;
  if (dtype == COUNTR_BF16 && modeA == COUNTR_OP_COL && (modeB == COUNTR_OP_IM2COL || modeB == COUNTR_OP_COL))
    return countr_lean_wgrad_tiles(a, modeB == COUNTR_OP_COL);
  return ((a->M + 127) / 128) * ((a->N + 127) / 128);
}

int countr_lean_wgrad_group_form(const countr_gemm_args* items, int n);         // conv_wgrad.hip: 0 = the group does not qualify
int countr_lean_wgrad_group(const countr_gemm_args* items, int n, hipStream_t s);

// Tiles per split-K slab of the ONE launch countr_gemm_group would run for these n launches (what a caller divides the CU count by to pick
// a common splitk), or 0 when the group runs as n separate launches (then countr_gemm_tiles per launch applies).
extern "C" int countr_gemm_group_tiles(const countr_gemm_args* items, int n, int dtype, int modeA, int modeB) {
  if (!items || dtype != COUNTR_BF16 || modeA != COUNTR_OP_COL || modeB != COUNTR_OP_COL) return 0;
  const int form = countr_lean_wgrad_group_form(items, n);
  if (!form) return 0;
  int tiles = 0;
  for (int i = 0; i < n; ++i) tiles += (items[i].M / 128) * (items[i].N / (128 * form));
  return tiles;
}

// n independent launches of one kind, as countr_gemm would run them one after the other -- in ONE kernel launch where a grouped form
// exists (bf16 (COL, COL) split-K launches: the weight gradients of a transformer block's nn.Linear layers, conv_wgrad.hip), otherwise
// one by one.  Results equal those of the separate launches bit for bit (same splitk per launch: same accumulation order).
extern "C" int countr_gemm_group(const countr_gemm_args* items, int n, int dtype, int modeA, int modeB, void* stream) {
  if (!items || n < 1 || n > 10) { countr_set_error("countr_gemm_group: 1 to 10 launches"); return -1; }
  if (dtype == COUNTR_BF16 && modeA == COUNTR_OP_COL && modeB == COUNTR_OP_COL && n > 1) {
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) {
      const countr_gemm_args* a = &items[i];
      ok = a->A && a->B && a->partial && a->M > 0 && a->N > 0 && a->K > 0 && (a->M % 8) == 0 && (a->N % 8) == 0 && (!a->rowsum_partial || a->partial);
    }
    if (ok) {
      const int rc = countr_lean_wgrad_group(items, n, reinterpret_cast<hipStream_t>(stream));
      if (rc != 1) return rc;
    }
  }
  for (int i = 0; i < n; ++i) {
    const int rc = countr_gemm(&items[i], dtype, modeA, modeB, stream);
    if (rc != 0) return rc;
  }
  return 0;
}

extern "C" int countr_gemm_rowsum_slabs(const countr_gemm_args* a, int dtype, int modeA, int modeB) {
  if (!a) return 0;
  if (dtype == COUNTR_BF16 && modeA == COUNTR_OP_COL && (modeB == COUNTR_OP_IM2COL || modeB == COUNTR_OP_COL))
    return countr_lean_wgrad_rowsum_slabs(a, modeB == COUNTR_OP_COL);
  return a->splitk > 1 ? a->splitk : 1;
}

// 1 when countr_gemm on these arguments runs on a kernel whose epilogue writes a->gn_rows (the bf16 (IM2ROW, ROW) launches of the lean
// convolution kernels: shape, alignment and the environment switches decide), else 0: the caller then keeps the separate statistics pass
extern "C" int countr_gemm_gn_rows(const countr_gemm_args* a, int dtype, int modeA, int modeB) {
  if (!a || dtype != COUNTR_BF16 || modeA != COUNTR_OP_IM2ROW || modeB != COUNTR_OP_ROW || !a->A || !a->B || !a->C || a->partial || (a->N % 32)) return 0;
  countr_dry_run = 1;
  int rc = countr_big_conv(a, nullptr);
  if (rc == 1) rc = countr_lean_conv(a, nullptr);
  countr_dry_run = 0;
  return rc == 0;
}

extern "C" int countr_gemm(const countr_gemm_args* a, int dtype, int modeA, int modeB, void* stream) {
  if (!a || !a->A || !a->B || (!a->C && !a->partial)) { countr_set_error("countr_gemm: null pointer"); return -1; }
  if ((a->splitk > 1 && !a->partial) || (a->partial && a->nbatch > 1)) { countr_set_error("countr_gemm: bad split-K setup"); return -1; }
  if (a->rowsum_partial && (dtype != COUNTR_BF16 || !a->partial)) { countr_set_error("countr_gemm: rowsum_partial needs the bf16 split-K path"); return -1; }
  if (a->act < COUNTR_ACT_NONE || a->act > COUNTR_ACT_GELU_BWD || (a->act == COUNTR_ACT_GELU_BWD && (!a->C2 || a->partial))) {
    countr_set_error("countr_gemm: bad activation code (GELU_BWD needs the saved pre-activation in C2 and no split-K)"); return -1;
  }
  const int epc = dtype == COUNTR_BF16 ? 8 : 4;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->N & 3)) { countr_set_error("countr_gemm: bad shape (need N % 4 == 0)"); return -1; }
  if ((modeA == COUNTR_OP_ROW || modeB == COUNTR_OP_ROW) && (a->K % epc)) { countr_set_error("countr_gemm: K must be a multiple of the 16-byte chunk"); return -1; }
  if (modeA == COUNTR_OP_COL && (a->M % epc)) { countr_set_error("countr_gemm: M must be a multiple of the chunk for COL A"); return -1; }
  if (modeB == COUNTR_OP_COL && (a->N % epc)) { countr_set_error("countr_gemm: N must be a multiple of the chunk for COL B"); return -1; }
  if ((modeA == COUNTR_OP_IM2ROW || modeB == COUNTR_OP_IM2COL) && (a->Cin % 64 || a->H <= 0 || a->W <= 0)) { countr_set_error("countr_gemm: conv modes need Cin % 64 == 0"); return -1; }
  if (a->gn_rows && !(dtype == COUNTR_BF16 && modeA == COUNTR_OP_IM2ROW && modeB == COUNTR_OP_ROW)) {
    countr_set_error("countr_gemm: gn_rows is an output of the bf16 (IM2ROW, ROW) convolution launches only"); return -1;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == COUNTR_BF16 && modeA == COUNTR_OP_ROW && modeB == COUNTR_OP_ROW) {   // (GELU_BWD: the lean kernel only)
    const int rb = countr_big_linear(a, s);    // chip-filling grids of 256 x 256 tiles (fc1): the 8-phase kernel of gemm256.hip
    if (rb != 1) return rb;
    const int rc = countr_lean_linear(a, s);   // full-tile nn.Linear forward shapes: the lean kernel of linear.hip
    if (rc != 1) return rc;
  }
  if (dtype == COUNTR_BF16 && modeA == COUNTR_OP_IM2ROW && modeB == COUNTR_OP_ROW) {
    const int rb = countr_big_conv(a, s);      // ... on the 192x192 maps: 256 x 256 tiles, 8-phase schedule (gemm256.hip)
    if (rb != 1) return rb;
    const int rc = countr_lean_conv(a, s);     // 3x3 convolution forward / dgrad on the big maps: same kernel, im2row LDS-DMA addressing
    if (rc != 1) return rc;
  }
  if (dtype == COUNTR_BF16 && modeA == COUNTR_OP_COL && (modeB == COUNTR_OP_IM2COL || modeB == COUNTR_OP_COL)) {
    // weight (+ bias) gradient of a 3x3 convolution or an nn.Linear: both operands staged K-major, transposing fragment reads
    const int rc = countr_lean_wgrad(a, modeB == COUNTR_OP_COL, s);
    if (rc != 1) return rc;
  }
  if (a->rowsum_partial && a->rowsum_slabs > 0 && a->rowsum_slabs != (a->splitk > 1 ? a->splitk : 1)) {
    countr_set_error("countr_gemm: rowsum_slabs does not match countr_gemm_rowsum_slabs() for this launch"); return -1;
  }
  if (a->gn_rows) {
    countr_set_error("countr_gemm: gn_rows needs the lean bf16 convolution kernels ((IM2ROW, ROW), more than 256 output tiles, N % 128 == 0, bf16 output without activation)");
    return -1;
  }
  if (a->ln_xcopy || a->ln_stats_out || a->ln_stats || a->ln_colsum) {
    countr_set_error("countr_gemm: the LayerNorm-folding fields need the lean bf16 (ROW, ROW) kernel (N % 128 == 0, K % 64 == 0, aligned operands, bias)");
    return -1;
  }
  if (dtype == COUNTR_BF16) return dispatch_bf16(*a, modeA, modeB, s);
  if (dtype == COUNTR_F32) return dispatch_f32(*a, modeA, modeB, s);
  countr_set_error("countr_gemm: bad dtype");
  return -1;
}

// Finisher of a split-K FORWARD-type GEMM: out[m][n] = sum_z partial[z][m][n] (+ bias[n]) in the output dtype.  For launches with few
// tiles and a long K (decode_head0 forward: 72 tiles x 72 k-tiles, exemplar conv4 dgrad: 24 x 72) the k loop is cut over the idle CUs
// and this pass costs less than the serial loop it replaces.
template <typename TO>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ partial, TO* __restrict__ out,
                                                            const float* __restrict__ bias, int splitk, int64_t MN4, int N4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // one thread = 4 consecutive columns
  if (i >= MN4) return;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) ld4<float>(bias + (i % N4) * 4, s);
  for (int z = 0; z < splitk; ++z) {
    float v[4];
    ld4<float>(partial + ((int64_t)z * MN4 + i) * 4, v);
    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
  }
  st4<TO>(out + i * 4, s);
}

extern "C" int countr_splitk_finish(const float* partial, void* out, const float* bias, int splitk, int M, int N, int out_bf16,
                                    void* stream) {
  if (!partial || !out || splitk < 1 || M <= 0 || N <= 0 || (N & 3)) { countr_set_error("countr_splitk_finish: bad args (N % 4 == 0)"); return -1; }
  const int64_t MN4 = (int64_t)M * N / 4;
  const dim3 grid((unsigned)((MN4 + 255) / 256));
  if (out_bf16) hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partial, (bf16_t*)out, bias, splitk, MN4, N / 4);
  else hipLaunchKernelGGL(splitk_finish_kernel<float>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), partial, (float*)out, bias, splitk, MN4, N / 4);
  COUNTR_LAUNCH_CHECK("countr_splitk_finish");
}

extern "C" int countr_splitk_reduce(const float* partial, float* out, int splitk, int M, int N, int perm_taps,
                                    int accumulate, const float* rowsum_partial, float* rowsum_out, void* stream) {
  if (!partial || !out || splitk < 1 || (rowsum_partial && !rowsum_out)) { countr_set_error("countr_splitk_reduce: bad args"); return -1; }
  const int64_t MN = (int64_t)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), partial, out, splitk, MN, N, perm_taps, accumulate, rowsum_partial,
                     rowsum_out, M);
  COUNTR_LAUNCH_CHECK("countr_splitk_reduce");
}
