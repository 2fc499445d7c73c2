// Shared device helpers for the CounTR gfx950 kernels.  gfx950-only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit storage / matrix-operand type of the "half" paths.  One source, two libraries (countr_amd/build.py): libcountr_hip.so is
// built with bfloat16 (precision="bf16", the throughput mode: 8 exponent bits, no loss scaling) and libcountr_hip_f16.so with
// -DCOUNTR_HALF_FP16=1 = IEEE fp16 (precision="fp16": the reference's own autocast dtype -- torch.cuda.amp.autocast() defaults to fp16,
// FSC_finetune_cross.py:273-275,286 -- 10 mantissa bits instead of 7; v_mfma_f32_*_f16 runs at the bf16 rate on gfx950).  `bf16_t` is the
// raw 16-bit pattern in both; every conversion goes through the helpers below, every matrix instruction through COUNTR_MFMA_*.
#ifndef COUNTR_HALF_FP16
#define COUNTR_HALF_FP16 0
#endif
typedef uint16_t bf16_t;  // raw 16 bits (bfloat16, or fp16 in the COUNTR_HALF_FP16 build)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

#define WAVE 64

#if COUNTR_HALF_FP16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;      // (the operand vector type of the fp16 MFMAs)
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2_t;
#define COUNTR_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define COUNTR_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define COUNTR_H16_ONE_PAIR 0x3c003c00u                              // {1.0, 1.0}
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// both halves of a packed pair (lo = bits 0..15)
__device__ __forceinline__ void unpack2h(uint32_t w, float& lo, float& hi) {
  const bf16x2_t h = __builtin_bit_cast(bf16x2_t, w);
  lo = (float)h[0]; hi = (float)h[1];
}
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#define COUNTR_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define COUNTR_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define COUNTR_H16_ONE_PAIR 0x3f803f80u
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ void unpack2h(uint32_t w, float& lo, float& hi) {
  lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
}
#endif
// fp32 -> 16-bit round-to-nearest-even through the native conversion (one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 per pair on gfx950)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

// Generic scalar load/store by storage type (float or bf16_t).
template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// 4-element vector load/store (16 B for float, 8 B for bf16).
template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  unpack2h(t.x, v[0], v[1]);
  unpack2h(t.y, v[2], v[3]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}

// 8-element vector load/store (2x16 B for float, 16 B for bf16).
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
  float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  uint4 t = *reinterpret_cast<const uint4*>(p);
  unpack2h(t.x, v[0], v[1]);
  unpack2h(t.y, v[2], v[3]);
  unpack2h(t.z, v[4], v[5]);
  unpack2h(t.w, v[6], v[7]);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]),
                                            pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}

// Wave-level (64 lanes) butterfly reductions: every lane ends with the result.
// Cross-lane reductions on the VALU: DPP (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror) inside the 16-lane rows, then
// gfx950 v_permlane16_swap / v_permlane32_swap between rows.  hipcc lowers __shfl_xor to ds_bpermute: an LDS round trip and an
// lgkmcnt wait per step (6 dependent ones for a wave sum).  All 64 lanes must be active (callers keep these wave-uniform).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp_mov<0xB1>(x);    // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);    // quad_perm [2,3,0,1]
  x += dpp_mov<0x141>(x);   // row_half_mirror
  x += dpp_mov<0x140>(x);   // row_mirror
  return x;
}
// sum of four squares with the contractions written out: the LayerNorm producers of linear.hip and gemm256.hip must agree bit for bit
__device__ __forceinline__ float countr_sq4(float a, float b, float c, float d) {
  return __builtin_fmaf(a, a, b * b) + __builtin_fmaf(c, c, d * d);
}
__device__ __forceinline__ float row16_max(float x) {
  x = fmaxf(x, dpp_mov<0xB1>(x));
  x = fmaxf(x, dpp_mov<0x4E>(x));
  x = fmaxf(x, dpp_mov<0x141>(x));
  x = fmaxf(x, dpp_mov<0x140>(x));
  return x;
}
#define COUNTR_SWAP_OP(NAME, SWAP, OP)                                                                                      \
  __device__ __forceinline__ float NAME(float x) {                                                                          \
    float t;                                                                                                                \
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\t" SWAP " %0, %1\n\ts_nop 1\n\t" OP " %0, %0, %1" : "+v"(x), "=&v"(t));    \
    return x;                                                                                                               \
  }
COUNTR_SWAP_OP(xor16_sum, "v_permlane16_swap_b32", "v_add_f32")
COUNTR_SWAP_OP(xor32_sum, "v_permlane32_swap_b32", "v_add_f32")
COUNTR_SWAP_OP(xor16_max, "v_permlane16_swap_b32", "v_max_f32")
COUNTR_SWAP_OP(xor32_max, "v_permlane32_swap_b32", "v_max_f32")
#undef COUNTR_SWAP_OP
__device__ __forceinline__ float quad_sum(float x) { x += dpp_mov<0xB1>(x); x += dpp_mov<0x4E>(x); return x; }   // lanes 4k .. 4k+3
// sum over the lanes that share (lane % NCV), NCV = 8, 16 or 32 (row_ror:8 pairs lane i with i ^ 8 inside its row)
template <int NCV> __device__ __forceinline__ float stride_sum(float x) {
  static_assert(NCV == 8 || NCV == 16 || NCV == 32, "stride_sum");
  if (NCV <= 8) x += dpp_mov<0x128>(x);
  if (NCV <= 16) x = xor16_sum(x);
  return xor32_sum(x);
}
__device__ __forceinline__ float half_wave_sum(float v) { return xor16_sum(row16_sum(v)); }   // over lanes 0-31 / 32-63
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row16_max(v))); }

// Block-level sum for blocks of NW waves; `sm` must hold >= NW floats.  All threads get the result.
template <int NW> __device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += sm[i];
  return r;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// bf16 mode, GELU DERIVATIVE: erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below bf16 resolution) with hardware rcp / exp2 --
// about a third of the instructions of libm erff.  One exponential exp(-x^2/2) serves both the cdf (erf(x/sqrt2)) and the pdf of the
// derivative.  fp32 parity mode keeps erff.
__device__ __forceinline__ void gelu_fast_parts(float x, float& cdf, float& e) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);   // exp(-x^2 / 2)
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float half_tail = 0.5f * poly * e;                      // 0.5 * erfc(|x| / sqrt2)
  cdf = x >= 0.f ? 1.f - half_tail : half_tail;
}
// bf16-mode FORWARD GELU: x * Phi(x) as x * sigmoid(x (c0 + c1 x^2 + c2 x^4)), c = (1.59501576, 7.40113008e-2, -7.03034904e-4) fitted
// (minimax over [-9, 9], tools/fit_gelu.py: |error| <= 2.6e-5 absolute against the erf form; bf16 output resolution is 4e-3 relative) and
// pre-multiplied by -log2(e); the polynomial argument is clamped to |x| <= 8 (the fitted quartic turns around near |x| = 10).  5 VALU
// + 2 transcendentals per element (3 + 2 in the packed form of linear.hip) against ~15 + 2 for the erf form: the fc1 epilogue evaluates
// 14 M of them per encoder layer at ~5 cycles of issue per instruction and wave.  ONE formula for every bf16 GEMM epilogue, scalar and
// packed form bit-identical, so that a sample's result does not depend on which kernel its batch size selects.
#define COUNTR_GELU_K0 (-2.30112137f)
#define COUNTR_GELU_K1 (-0.10677575f)
#define COUNTR_GELU_K2 (1.01426498e-3f)
__device__ __forceinline__ float gelu_fast(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.f, 8.f);
  const float x2 = xc * xc;
  float t = __builtin_fmaf(x2, COUNTR_GELU_K2, COUNTR_GELU_K1);
  t = __builtin_fmaf(t, x2, COUNTR_GELU_K0);
  const float u = t * xc;
  return x * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(u) + 1.f);
}
__device__ __forceinline__ float gelu_fast_grad(float x) { float c, e; gelu_fast_parts(x, c, e); return fmaf(x * 0.3989422804014327f, e, c); }
template <typename T> __device__ __forceinline__ float gelu_t(float x) { if constexpr (sizeof(T) == 2) return gelu_fast(x); else return gelu_erf(x); }
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x) { if constexpr (sizeof(T) == 2) return gelu_fast_grad(x); else return gelu_erf_grad(x); }

// GroupNorm statistics from a convolution's epilogue (countr_gemm_args.gn_rows): the lane holds the eight ROUNDED 16-bit outputs of row m
// at columns [col, col + 8) -- the four lanes of a DPP quad hold one 32-channel block of the row (col % 32 == 8 (lane & 3)) -- and the
// block's {sum, sum of squares} goes to rows[m][N / 32][2].  One fixed tree per row (pairs inside the lane, then the quad butterfly): the
// value does not depend on the kernel, the tile or the batch a row is computed in (gemm256.hip and linear.hip share this function).
typedef __attribute__((ext_vector_type(4))) unsigned countr_u32x4_t;
// {sum, sum of squares} of the 32-channel block whose four 8-value pieces the lanes of a DPP quad hold (every lane of the quad gets both):
// THE tree of a pixel's GroupNorm partials, shared by the convolution epilogues and by the statistics pass that reads the map itself
// (norm.hip::gn_stats_tree_kernel), so that the two routes to a GroupNorm's statistics agree bit for bit
__device__ __forceinline__ void countr_gn_quad_sums(const countr_u32x4_t packed, float& s1, float& s2) {
  s1 = 0.f; s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float lo, hi;
    unpack2h(packed[e], lo, hi);
    s1 += lo + hi;
    s2 += __builtin_fmaf(lo, lo, hi * hi);
  }
  s1 += dpp_mov<0xB1>(s1); s2 += dpp_mov<0xB1>(s2);     // lane ^ 1
  s1 += dpp_mov<0x4E>(s1); s2 += dpp_mov<0x4E>(s2);     // lane ^ 2
}
__device__ __forceinline__ void countr_gn_row_partials(const countr_u32x4_t packed, float* __restrict__ rows, int m, int N, int col, int lane) {
  if (!rows) return;      // (uniform)
  float s1, s2;
  countr_gn_quad_sums(packed, s1, s2);
  if ((lane & 3) == 0) *reinterpret_cast<float2*>(rows + ((int64_t)m * (N >> 5) + (col >> 5)) * 2) = make_float2(s1, s2);
}

// != 0 while a query (countr_gemm_gn_rows) walks the launch path: the lean kernels' launch functions then return 0 without launching,
// so that "would this launch run on a kernel with that epilogue?" is answered by the selection code itself (defined in api.hip)
extern thread_local int countr_dry_run;

// Error plumbing shared by all translation units (defined in api.hip).
extern "C" void countr_set_error(const char* msg);
int countr_check_launch(const char* what);
// >= n zeros in device memory on the CURRENT device, allocated once by countr_init (api.hip) and read-only afterwards; nullptr (+ error
// text) when countr_init was not called for this device.  The launch paths never allocate.
#define COUNTR_ZERO_VEC_FLOATS 8192
const float* countr_zero_vec(int n);
#define COUNTR_LAUNCH_CHECK(what) return countr_check_launch(what)

// K order of the 3x3 convolutions' implicit GEMMs in linear.hip / gemm256.hip (K = 9 taps x Cin, k-tile = 64 channels of one tap):
// 1 = channel-chunk-major -- k-tile t is tap t % 9 of chunk t / 9, so nine consecutive k-tiles read the SAME 128-byte pieces of the
// map's pixels (shifted by a row / a pixel): the working set of an XCD's 32 workgroups over those nine k-tiles is ~1.1 MB and stays in
// its 4-MB L2; tap-major (0) walks the whole 4.4-MB footprint once per tap and re-fetched the 192x192 map 4.9 x per launch from the
// memory side (profiles/r4_conv_pmc.txt).  Both kernels use the same order (their results agree bit for bit).
#ifndef COUNTR_CONV_CHUNK_MAJOR
#define COUNTR_CONV_CHUNK_MAJOR 1
#endif
// (tap, channel offset) of k-tile t for Cin = 64 << cpt_log; t / 9 as (t * 57) >> 9 is exact for t < 512 (Cin <= 512: t <= 71;
// countr_lean_conv_rows / countr_big_conv refuse wider inputs)
__device__ __forceinline__ void countr_conv_ktile(int t, int Cin, int& tap, int& cb) {
#if COUNTR_CONV_CHUNK_MAJOR
  const int c = (t * 57) >> 9;
  tap = t - 9 * c;
  cb = c << 6;
#else
  const int k0 = t << 6;
  tap = k0 / Cin;
  cb = k0 - tap * Cin;
#endif
}

// The first workgroups of a GEMM launch (blockIdx.x < npf) warm the cache with a read-only range -- the next launch's weight
// panel -- and leave: block j of npf reads every npf-th 16-byte x blockDim slice; the values go nowhere (countr_gemm_args.prefetch).
__device__ __forceinline__ void countr_prefetch_range(const char* p, long long bytes, int j, int npf) {
  const uint4* __restrict__ s = reinterpret_cast<const uint4*>(p);
  const long long n = bytes >> 4, stride = (long long)npf * blockDim.x;
  uint4 acc = {0u, 0u, 0u, 0u};
  for (long long i = (long long)j * blockDim.x + threadIdx.x; i < n; i += 4 * stride) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * stride < n) { const uint4 v = s[i + u * stride]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  }
  asm volatile("" ::"v"(acc.x), "v"(acc.y), "v"(acc.z), "v"(acc.w));
}
// workgroups a launch of `tiles` tiles puts IN FRONT of its tile workgroups for the hint (a multiple of 8: workgroup b runs on XCD b % 8,
// and the tile order is XCD-aware): the idle CUs of a single-round grid, or 32 CU slots for the first microseconds of a bigger one
static inline int countr_prefetch_blocks(long tiles, const void* p, long long bytes) {
  if (!p || bytes < 16 || ((uintptr_t)p & 15) || (bytes & 15)) return 0;
  if (tiles >= 256) return 32;
  const long spare = ((256 - tiles) / 8) * 8;
  return (int)(spare < 64 ? spare : 64);
}
