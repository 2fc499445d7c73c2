// 256 x 256 workgroup tile on the 8-phase schedule (hand-written for gfx950): out = epilogue(A W^T + b) for the launches whose grid of
// 256 x 256 tiles still fills the chip -- the 3x3 convolutions of the density head on the 192x192 / 96x96 maps (forward, and dgrad through
// the dgrad-form weights: models_mae_cross.py:80-100,185-197) and the encoder's fc1 (models_crossvit.py:62) -- where the 128x256 forms of
// linear.hip are bound by L2 -> LDS bytes per flop (profiles/r3_linear_stamps.txt, r3_conv_wgrad_microbench.txt): a 256 x 256 x 64
// k-tile stages 64 KB for 2048 matrix cycles per SIMD, 2/3 of the 128x256 form's bytes per flop.
//
// Structure (cdna_hip_programming.md, "The 256^2 8-phase template"; round 3's 256x256 attempt had the tile but not this schedule):
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator registers), ALL waves stage
//     and multiply; the two waves of a SIMD are one wave of each M half, and the M halves run ONE BARRIER APART: while one half issues
//     its 8 MFMAs of a phase (s_setprio 1), the other issues its fragment reads and its two LDS-DMA pieces.
//   * a k-tile = 4 phases, one output quadrant (64 x 32 x K 64 = 8 MFMAs per wave) each; fragments: A rows lo (8 ds_read_b128) + W lo (4)
//     in phase 1, W hi (4) in phase 2, A rows hi (8) in phase 3, nothing in phase 4 (24 reads per k-tile: both W halves stay in registers).
//   * staging: a k-tile is four 16-KB UNITS that match those reads -- A-lo (rows 0-63 of both M halves), W-lo (columns 0-31 of the four
//     N quarters), W-hi, A-hi -- two k-tile buffers = 128 KB; every phase issues ONE unit (two 1-KiB MUBUF LDS-DMA pieces per wave) into
//     the region whose last read lies two or more phases back, five phases before its first read: four units (64 KB) are in flight per
//     CU at any time, and the only wait is ONE s_waitcnt vmcnt(8) per phase (never 0 inside the loop), raw s_barrier + lgkmcnt(0).
//   * LDS image as in linear.hip: 128-byte rows, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) (swizzle on the DMA source address and
//     on the read), W rows read in the permuted order that makes a lane's 16 accumulator registers 16 consecutive output columns.
//   * epilogue: fp32 accumulators -> LDS (two passes of 64 rows per wave) -> whole 128-byte row segments; bias, LayerNorm fold, GELU on the
//     read-back side.
#include "common.hpp"
#include "../../include/countr_hip.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

#ifndef G256_ABL
#define G256_ABL 0   // timing experiments (results are WRONG): 1 = no MFMA, 2 = no fragment reads, 3 = no DMA after the prologue
#endif

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void* lds_vptr_t;
typedef __attribute__((address_space(3))) const char* lds_cptr_t;

enum { EPI_BF16 = 0, EPI_GELU = 1, EPI_RES = 2 };   // RES: fp32 out = acc + bias + resid (+ the LayerNorm producer's bf16 copy and row partials)

struct BigArgs {
  const char* A;       // [M, lda] bf16, or CONV: NHWC map [B, H, Wd, Cin]
  const char* W;       // [N, ldw] bf16
  char* C;             // bf16 [M, ldc]
  char* C2;            // EPI_GELU: optional bf16 pre-activation copy
  const float* bias;   // [N]
  const float* resid;  // EPI_RES: fp32 [*, ldres], row m % res_mod when res_mod > 0
  char* xcopy;         // EPI_RES, LayerNorm producer (optional): bf16 copy of the fp32 output [M, ldc]
  float* stats_out;    // ... and [M][N/64][2] = {sum, sum of squares} of the output row over each 64-column block
  int ldres, res_mod;
  int M, N, K;
  int lda, ldw, ldc;   // elements
  int tilesN;
  const char* pf; long long pf_bytes; int npf;   // workgroups < npf only read this range and leave (cache warm-up hint of countr_gemm_args)
  int launch_tiles;    // tiles this launch covers (the first ones of the problem; the rest may run as a tail launch of linear.hip's kernel)
  int H, Wd, Cin, cpt_log;   // CONV: k-tiles per tap = Cin / 64 = 1 << cpt_log
  const float* stats_in;     // LN consumer (see linear.hip): [M][K/64][2]
  const float* colsum;       // [N]
  float ln_eps;
  float* stamps;             // G256_STAMP builds
  float* gn_rows;            // CONV (optional): [M][N/32][2] = {sum, sum of squares} of every 32-channel block of the ROUNDED output row
};

constexpr int UNIT = 16384, B_BASE = 65536;
constexpr int OPITCH = 64 * 4 + 16;             // staging pitch of a 64-column fp32 row
constexpr int REGION = 64 * OPITCH;             // one wave's staging region (64 rows x 64 columns)
constexpr int LNST_OFF = 8 * REGION;            // row statistics behind the staging regions (the ring is smaller)
constexpr int LDS_BYTES = LNST_OFF + 256 * 8;

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ uint32_t lds_u32(const char* p) { return (uint32_t)(uintptr_t)(lds_cptr_t)p; }

template <int OFF> __device__ __forceinline__ bf16x8_t ds_read128(uint32_t a) {
  bf16x8_t v;
#if G256_ABL == 2
  v = __builtin_bit_cast(bf16x8_t, u32x4_t{a, (uint32_t)OFF, 1u, 2u});
#else
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
#endif
  return v;
}
__device__ __forceinline__ void wait4(bf16x8_t (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
}
__device__ __forceinline__ void wait8(bf16x8_t (&f)[2][4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[0][3]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]), "+v"(f[1][3]));
}
__device__ __forceinline__ void wait12(bf16x8_t (&f)[2][4], bf16x8_t (&g)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[0][3]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]), "+v"(f[1][3]),
               "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]));
}

// packed form of common.hpp's gelu_fast (same operations on fp32 pairs: bit-identical results; copy of linear.hip's)
__device__ __forceinline__ f32x2_t gelu_sig2(f32x2_t x) {
  const f32x2_t xc = {__builtin_amdgcn_fmed3f(x[0], -8.f, 8.f), __builtin_amdgcn_fmed3f(x[1], -8.f, 8.f)};
  const f32x2_t x2 = xc * xc;
  const f32x2_t k0 = {COUNTR_GELU_K0, COUNTR_GELU_K0}, k1 = {COUNTR_GELU_K1, COUNTR_GELU_K1}, k2 = {COUNTR_GELU_K2, COUNTR_GELU_K2};
  f32x2_t t = __builtin_elementwise_fma(x2, k2, k1);
  t = __builtin_elementwise_fma(t, x2, k0);
  const f32x2_t u = t * xc;
  const f32x2_t one = {1.f, 1.f};
  const f32x2_t e = {__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
  const f32x2_t d = e + one;
  const f32x2_t r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return x * r;
}

#ifndef G256_PH
#define G256_PH 2     // phases per k-tile: 4 = one output quadrant (8 MFMAs) per phase, 2 = two quadrants (16 MFMAs) per phase
#endif
#ifndef G256_REC
#define G256_REC 2    // stamp builds: the pair of k-tiles (2 G256_REC, 2 G256_REC + 1) whose phases are recorded
#endif

template <bool CONV, int EPI, bool LN>
__global__ __launch_bounds__(512, 2) void g256_kernel(const BigArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < g.npf) { countr_prefetch_range(g.pf, g.pf_bytes, (int)blockIdx.x, g.npf); return; }
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  // XCD-aware tile order (workgroup b runs on XCD b % 8): every XCD owns one contiguous range of the (tile_m, tile_n) space
  int lt;
  {
    const int bt = (int)blockIdx.x - g.npf;      // (npf % 8 == 0)
    const int nt = g.launch_tiles, q = nt >> 3, r = nt & 7, x = bt & 7, j = bt >> 3;
    lt = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
  }
  const int tile_m = lt / g.tilesN, tile_n = lt - tile_m * g.tilesN;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int ntiles = g.K >> 6;      // even (host check)

  // ---- epilogue operands first: plain loads, consumed after the main loop (older than every LDS-DMA: the counted waits cover them)
  const int ccol = (lane & 7) * 8, rrow = lane >> 3;
  const int sn0 = n0 + wc * 64;
  float bcol[8], ccs[LN ? 8 : 1];
  {
    const float4 b0 = *reinterpret_cast<const float4*>(g.bias + sn0 + ccol), b1 = *reinterpret_cast<const float4*>(g.bias + sn0 + ccol + 4);
    bcol[0] = b0.x; bcol[1] = b0.y; bcol[2] = b0.z; bcol[3] = b0.w; bcol[4] = b1.x; bcol[5] = b1.y; bcol[6] = b1.z; bcol[7] = b1.w;
    if constexpr (LN) {
      const float4 c0 = *reinterpret_cast<const float4*>(g.colsum + sn0 + ccol), c1 = *reinterpret_cast<const float4*>(g.colsum + sn0 + ccol + 4);
      ccs[0] = c0.x; ccs[1] = c0.y; ccs[2] = c0.z; ccs[3] = c0.w; ccs[4] = c1.x; ccs[5] = c1.y; ccs[6] = c1.z; ccs[7] = c1.w;
    }
  }
  // LN consumer: the row partials of this tile's 256 rows, one row per thread of the first four waves; loaded by inline asm (hipcc
  // would wait vmcnt(0) for a plain load that has LDS-DMA behind it) and reduced behind the prologue's DMA issue
  u32x4_t lnv[LN ? 6 : 1];
  int ln_nb2 = 0;
  if constexpr (LN) {
    if (tid < 256) {
      const int m = min(m0 + tid, g.M - 1);
      const int nblk = g.K >> 6;
      ln_nb2 = nblk >> 1;     // float4 = two 64-column blocks; nblk even, <= 12 (host check)
      const char* sp = reinterpret_cast<const char*>(g.stats_in + (int64_t)m * nblk * 2);
#pragma unroll
      for (int b2 = 0; b2 < 6; ++b2) {
        const char* p = sp + (b2 < ln_nb2 ? b2 : 0) * 16;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lnv[b2]) : "v"(p) : "memory");
      }
    }
  }

  // ---- LDS-DMA set-up.  Unit (kind, u) of buffer b at (kind ? B_BASE : 0) + (2 u + b) UNIT, 128 rows of 128 bytes; a wave stages rows
  // [64 j + 8 w, +8) of every unit (j = 0, 1): lane -> (row, 16-byte slot), source chunk = slot ^ ((row >> 1) & 7).
  //   A unit u, row 64 j + r  = tile row 128 j + 64 u + r       (j = the M half that reads it, u = lo / hi rows of its wave tiles)
  //   W unit u, row 64 j + r  = tile column 128 j + 64 (r >> 5) + 32 u + (r & 31)
  const int l8 = lane >> 3;
  const int dchunk = (lane & 7) ^ (((w & 1) << 2) | (lane >> 4));
  const int arow = w * 8 + l8;                                   // + 64 i, pass i = 2 j + u
  const int bcolr = (w >> 2) * 64 + (w & 3) * 8 + l8;            // + 128 j + 32 u
  const uint32_t voffA = CONV ? (uint32_t)(((m0 + arow) * g.Cin + dchunk * 8) * 2) : (uint32_t)((arow * g.lda + dchunk * 8) * 2);
  const uint32_t voffB = (uint32_t)((bcolr * g.ldw + dchunk * 8) * 2);
  uint32_t rowmask = 0;          // bit i: this lane's row of pass i exists (ragged M: the others stage zeros)
#pragma unroll
  for (int i = 0; i < 4; ++i) rowmask |= (m0 + arow + 64 * i < g.M) ? (1u << i) : 0u;
  uint32_t vmask[CONV ? 4 : 1];  // CONV: bit t of vmask[i] <=> tap t of this lane's pixel of pass i lies inside the image
  if constexpr (CONV) {
    const int m = m0 + arow;
    int x = m % g.Wd, y = (m / g.Wd) % g.H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t vm = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if ((rowmask >> i & 1u) && (unsigned)(y + t / 3 - 1) < (unsigned)g.H && (unsigned)(x + t % 3 - 1) < (unsigned)g.Wd) vm |= 1u << t;
      vmask[i] = vm;
      x += 64;
      while (x >= g.Wd) { x -= g.Wd; y = (y + 1 == g.H) ? 0 : y + 1; }
    }
  }
  // MUBUF LDS-DMA (buffer_load_dwordx4 ... offen lds): descriptor base + 32-bit lane offset (loop invariant) + scalar offset; a lane
  // whose offset lies outside the descriptor stages ZEROS (ragged rows, the convolution's padding taps, k-tiles behind the last one).
  // CONV: the descriptor base sits (Wd + 1) pixels in front of the map so that every tap shift is a non-negative scalar offset.
  const int64_t cshift = CONV ? (int64_t)(g.Wd + 1) * g.Cin * 2 : 0;
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(CONV ? g.A - cshift : g.A + (int64_t)m0 * g.lda * 2), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (int64_t)n0 * g.ldw * 2), 0, 0x7ffffff0, 0x00020000);
  const uint32_t passA = (uint32_t)(CONV ? g.Cin : g.lda) * 128u;     // 64 rows further, bytes
  const uint32_t passB = (uint32_t)g.ldw * 64u;                       // 32 columns further, bytes
  char* const dma_dst = smem + w * 1024;
  // one unit of k-tile t into buffer BUF: KIND 0 = A, 1 = W; U = lo / hi
  auto issue = [&](auto KIND, auto UU, auto BB, int t) {
    constexpr int kind = decltype(KIND)::value, u = decltype(UU)::value, buf = decltype(BB)::value;
#if G256_ABL == 3
    if (t >= 2) return;
#endif
    const uint32_t kill = t < ntiles ? 0u : 0x80000000u;
    char* dst = dma_dst + (kind ? B_BASE : 0) + (2 * u + buf) * UNIT;
    if constexpr (kind == 0) {
      if constexpr (CONV) {
        int tap, cb;
        countr_conv_ktile(t, g.Cin, tap, cb);
        const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;
        const uint32_t so = (uint32_t)(cshift + ((int64_t)((ty - 1) * g.Wd + (tx - 1)) * g.Cin + cb) * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = 2 * j + u;
          // (named variable on purpose: a conditional expression written as the builtin's argument makes the HOST pass of hipcc 7.2
          // drop the kernel's stub silently -- linear.hip)
          const uint32_t vo = ((vmask[i] >> tap) & 1u) ? (voffA | kill) : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_vptr_t)(dst + j * 8192), 16, vo, so + i * passA, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = 2 * j + u;
          const uint32_t vo = (rowmask >> i & 1u) ? (voffA | kill) : 0x80000000u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_vptr_t)(dst + j * 8192), 16, vo, t * 128 + i * passA, 0, 0);
        }
      }
    } else {
      const uint32_t vo = voffB | kill;
      uint32_t kb = (uint32_t)t * 128u;            // byte offset of k-tile t inside a W row ([tap][Cin] for a convolution)
      if constexpr (CONV) { int tap, cb; countr_conv_ktile(t, g.Cin, tap, cb); kb = (uint32_t)(tap * g.Cin + cb) * 2u; }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_vptr_t)(dst + j * 8192), 16, vo, kb + (4 * j + u) * passB, 0, 0);
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

  // ---- prologue: the stream of units is A-lo, W-lo, W-hi, A-hi per k-tile; six units are ahead of the first phase
  issue(I0{}, I0{}, I0{}, 0); issue(I1{}, I0{}, I0{}, 0); issue(I1{}, I1{}, I0{}, 0); issue(I0{}, I1{}, I0{}, 0);
  issue(I0{}, I0{}, I1{}, 1); issue(I1{}, I0{}, I1{}, 1);
#if G256_PH == 2
  issue(I1{}, I1{}, I1{}, 1);      // (two-phase form: the three units a k-tile's first phase reads travel together, seven units ahead)
#endif
  if constexpr (LN) {
    // the six partial loads (older than the DMA pieces) have returned
#if G256_PH == 2
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#endif
    if (tid < 256) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int b2 = 0; b2 < 6; ++b2) {
        asm volatile("" : "+v"(lnv[b2]));       // (defined only behind the wait above)
        if (b2 < ln_nb2) {
          s1 += __uint_as_float(lnv[b2][0]) + __uint_as_float(lnv[b2][2]);
          s2 += __uint_as_float(lnv[b2][1]) + __uint_as_float(lnv[b2][3]);
        }
      }
      // (explicit fma: the two kernels that serve this epilogue -- linear.hip, gemm256.hip -- must not differ by a compiler's contraction choice)
      const float inv = 1.f / (float)g.K, mean = s1 * inv, ex2 = s2 * inv, var = fmaxf(__builtin_fmaf(-mean, mean, ex2), 0.f);
      const float rs = rsqrtf(var + g.ln_eps);
      const uint32_t sa = lds_u32(smem) + LNST_OFF + tid * 8;
      asm volatile("ds_write_b64 %0, %1" ::"v"(sa), "v"(make_float2(mean, rs)) : "memory");   // (asm: a compiler-visible LDS store would wait vmcnt(0))
    }
  }

  // ---- fragment addresses: A rows (wr half, natural order), W rows (wc quarter, permuted order p(l31)); k-step kk = chunks 2 kk + lh
  const int l31 = lane & 31, lh = lane >> 5;
  const int prow = ((l31 >> 2) & 1) * 16 + ((l31 >> 3) & 3) * 4 + (l31 & 3);
  const int swx = (l31 >> 1) & 7, sww = (prow >> 1) & 7;
  uint32_t xad[4], wad[4];
  {
    const uint32_t sb = lds_u32(smem);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      xad[kk] = sb + (uint32_t)((wr * 64 + l31) * 128 + (((kk * 2 + lh) ^ swx) << 4));
      wad[kk] = sb + (uint32_t)(B_BASE + (wc * 32 + prow) * 128 + (((kk * 2 + lh) ^ sww) << 4));
    }
  }
  f32x16_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  bf16x8_t xa[2][4], wlo[4], whi[4];

#define G_SB __builtin_amdgcn_sched_barrier(0)
#define G_BAR __builtin_amdgcn_s_barrier()
#if G256_ABL == 1
#define G_MM(TM, TN, WF, XF) { const u32x4_t a_ = __builtin_bit_cast(u32x4_t, WF), b_ = __builtin_bit_cast(u32x4_t, XF); \
        acc[TM][TN][0] += __uint_as_float(a_[0] ^ b_[0]); acc[TM][TN][5] += __uint_as_float(a_[1] ^ b_[1]); \
        acc[TM][TN][10] += __uint_as_float(a_[2] ^ b_[2]); acc[TM][TN][15] += __uint_as_float(a_[3] ^ b_[3]); }
#else
#define G_MM(TM, TN, WF, XF) acc[TM][TN] = COUNTR_MFMA_32X32X16(WF, XF, acc[TM][TN], 0, 0, 0)
#endif
  // one output quadrant: rows half AH (wave-tile rows 64 AH + [0, 64)), column tile TN, fragments WF[kk] x xa[tm2][kk]
#ifndef G256_ORDER
#define G256_ORDER 1
#endif
#if G256_ORDER == 0     // the two accumulators alternate (a dependent MFMA every other slot)
#define G_QUAD(AH, TN, WF)                                                     \
  G_MM(2 * AH, TN, WF[0], xa[0][0]); G_MM(2 * AH + 1, TN, WF[0], xa[1][0]);    \
  G_MM(2 * AH, TN, WF[1], xa[0][1]); G_MM(2 * AH + 1, TN, WF[1], xa[1][1]);    \
  G_MM(2 * AH, TN, WF[2], xa[0][2]); G_MM(2 * AH + 1, TN, WF[2], xa[1][2]);    \
  G_MM(2 * AH, TN, WF[3], xa[0][3]); G_MM(2 * AH + 1, TN, WF[3], xa[1][3]);
#else                   // one accumulator's four k-steps back to back (the accumulate-forwarding path), then the other's
#define G_QUAD(AH, TN, WF)                                                     \
  G_MM(2 * AH, TN, WF[0], xa[0][0]); G_MM(2 * AH, TN, WF[1], xa[0][1]);        \
  G_MM(2 * AH, TN, WF[2], xa[0][2]); G_MM(2 * AH, TN, WF[3], xa[0][3]);        \
  G_MM(2 * AH + 1, TN, WF[0], xa[1][0]); G_MM(2 * AH + 1, TN, WF[1], xa[1][1]); \
  G_MM(2 * AH + 1, TN, WF[2], xa[1][2]); G_MM(2 * AH + 1, TN, WF[3], xa[1][3]);
#endif
#define G_P1 __builtin_amdgcn_s_setprio(1)
#define G_P0 __builtin_amdgcn_s_setprio(0)
#define G_VM8 asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

#ifdef G256_STAMP
  // s_memtime anatomy (tools/stamp_g256.py): four stamps per phase -- start of the M part, in front of the mid barrier, behind it, behind
  // the MFMAs -- of k-tiles 2 G256_REC and 2 G256_REC + 1, kept in SGPRs and moved into lanes of ONE VGPR a phase later, behind that
  // phase's own lgkmcnt(0) (an s_memtime returns through lgkmcnt: consuming it earlier would drain the fragment reads as well)
  uint32_t stq[8][4] = {};
  uint32_t rec = 0;
  const uint64_t sk0 = __builtin_readcyclecounter(), sr0 = wall_clock64();
#define GSTQ(P, I) stq[P][I] = (uint32_t)__builtin_readcyclecounter()
#define GREC1(P, I) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(rec) : "s"(stq[P][I]), "n"(4 * (P) + (I)))
#define GREC(P, TT) if (((TT) >> 1) == G256_REC) { GREC1(P, 0); GREC1(P, 1); GREC1(P, 2); GREC1(P, 3); }
#else
#define GSTQ(P, I)
#define GREC(P, TT)
#endif
  // one k-tile (buffer BUF) = four phases; in its M part a phase reads fragments, issues ONE unit and waits until at most four units
  // (eight pieces of this wave) are in flight: the unit read in phase p + 1 was issued five phases earlier and has been waited for by
  // EVERY wave before the barrier in front of that read (both M halves: the later half passes one more barrier first)
  auto ktile = [&](auto BB, int t) {
    constexpr int buf = decltype(BB)::value;
    using NB = std::integral_constant<int, buf ^ 1>;
    constexpr int P0 = 4 * buf;
    // phase 1: A-lo + W-lo of this k-tile; issue W-hi of k-tile t + 1
    {
      GSTQ(P0, 0);
      sfor<2>([&](auto TM2) { sfor<4>([&](auto KK) { constexpr int tm2 = decltype(TM2)::value, kk = decltype(KK)::value;
        xa[tm2][kk] = ds_read128<(0 + buf) * UNIT + tm2 * 4096>(xad[kk]); }); });
      sfor<4>([&](auto KK) { constexpr int kk = decltype(KK)::value; wlo[kk] = ds_read128<(0 + buf) * UNIT>(wad[kk]); });
      issue(I1{}, I1{}, NB{}, t + 1);
      G_VM8;
      GSTQ(P0, 1);
      G_BAR;
      GSTQ(P0, 2);
      wait12(xa, wlo); G_SB;
      GREC((P0 + 7) & 7, t - 1);
      G_P1; G_QUAD(0, 0, wlo); G_P0; G_SB;
      GSTQ(P0, 3);
      G_BAR;
    }
    // phase 2: W-hi; issue A-hi of k-tile t + 1
    {
      GSTQ(P0 + 1, 0);
      sfor<4>([&](auto KK) { constexpr int kk = decltype(KK)::value; whi[kk] = ds_read128<(2 + buf) * UNIT>(wad[kk]); });
      issue(I0{}, I1{}, NB{}, t + 1);
      G_VM8;
      GSTQ(P0 + 1, 1);
      G_BAR;
      GSTQ(P0 + 1, 2);
      wait4(whi); G_SB;
      GREC(P0, t);
      G_P1; G_QUAD(0, 1, whi); G_P0; G_SB;
      GSTQ(P0 + 1, 3);
      G_BAR;
    }
    // phase 3: A-hi; issue A-lo of k-tile t + 2 (this buffer: its A-lo / W-lo regions were last read in phase 1)
    {
      GSTQ(P0 + 2, 0);
      sfor<2>([&](auto TM2) { sfor<4>([&](auto KK) { constexpr int tm2 = decltype(TM2)::value, kk = decltype(KK)::value;
        xa[tm2][kk] = ds_read128<(2 + buf) * UNIT + tm2 * 4096>(xad[kk]); }); });
      issue(I0{}, I0{}, BB, t + 2);
      G_VM8;
      GSTQ(P0 + 2, 1);
      G_BAR;
      GSTQ(P0 + 2, 2);
      wait8(xa); G_SB;
      GREC(P0 + 1, t);
      G_P1; G_QUAD(1, 1, whi); G_P0; G_SB;
      GSTQ(P0 + 2, 3);
      G_BAR;
    }
    // phase 4: no reads; issue W-lo of k-tile t + 2
    {
      GSTQ(P0 + 3, 0);
      issue(I1{}, I0{}, BB, t + 2);
      G_VM8;
      GSTQ(P0 + 3, 1);
      G_BAR;
      GSTQ(P0 + 3, 2);
      G_SB;
      GREC(P0 + 2, t);
      G_P1; G_QUAD(1, 0, wlo); G_P0; G_SB;
      GSTQ(P0 + 3, 3);
      G_BAR;
    }
  };

#if G256_PH == 2
  // Two phases per k-tile (16 MFMAs = 512 matrix cycles each: half the barriers, and the partner half's M part -- which costs the
  // multiplying wave ~40 cycles per segment whatever its length -- is paid half as often).  X: A-lo, W-lo, W-hi (16 reads) -> quadrants
  // (lo, lo), (lo, hi); issues A-hi of k-tile t + 1.  Y: A-hi (8 reads) -> (hi, hi), (hi, lo); issues A-lo, W-lo, W-hi of k-tile t + 2
  // into the regions X has just read: every wave retires its fragment reads (lgkmcnt(0)) BEFORE the mid barrier, so a region may be
  // restaged one phase after its last read.  A unit is waited for (vmcnt(8): at most four units in flight) in the phase before its read,
  // two phases after its issue.
  auto ktile2 = [&](auto BB, int t) {
    constexpr int buf = decltype(BB)::value;
    using NB = std::integral_constant<int, buf ^ 1>;
    constexpr int P0 = 2 * buf;
    {
      GSTQ(P0, 0);
      sfor<2>([&](auto TM2) { sfor<4>([&](auto KK) { constexpr int tm2 = decltype(TM2)::value, kk = decltype(KK)::value;
        xa[tm2][kk] = ds_read128<(0 + buf) * UNIT + tm2 * 4096>(xad[kk]); }); });
      sfor<4>([&](auto KK) { constexpr int kk = decltype(KK)::value; wlo[kk] = ds_read128<(0 + buf) * UNIT>(wad[kk]); });
      sfor<4>([&](auto KK) { constexpr int kk = decltype(KK)::value; whi[kk] = ds_read128<(2 + buf) * UNIT>(wad[kk]); });
      issue(I0{}, I1{}, NB{}, t + 1);
      wait12(xa, wlo); wait4(whi);
      G_VM8;
      GSTQ(P0, 1);
      G_BAR;
      GSTQ(P0, 2);
      G_SB;
      GREC((P0 + 3) & 3, t - 1);
      G_P1; G_QUAD(0, 0, wlo); G_QUAD(0, 1, whi); G_P0; G_SB;
      GSTQ(P0, 3);
      G_BAR;
    }
    {
      GSTQ(P0 + 1, 0);
      sfor<2>([&](auto TM2) { sfor<4>([&](auto KK) { constexpr int tm2 = decltype(TM2)::value, kk = decltype(KK)::value;
        xa[tm2][kk] = ds_read128<(2 + buf) * UNIT + tm2 * 4096>(xad[kk]); }); });
      issue(I0{}, I0{}, BB, t + 2); issue(I1{}, I0{}, BB, t + 2); issue(I1{}, I1{}, BB, t + 2);
      wait8(xa);
      G_VM8;
      GSTQ(P0 + 1, 1);
      G_BAR;
      GSTQ(P0 + 1, 2);
      G_SB;
      GREC(P0, t);
      G_P1; G_QUAD(1, 1, whi); G_QUAD(1, 0, wlo); G_P0; G_SB;
      GSTQ(P0 + 1, 3);
      G_BAR;
    }
  };
#endif

  G_VM8;          // A-lo(0), W-lo(0) of this wave have landed
  G_BAR;          // ... of every wave
  if (wr == 1) G_BAR;     // the second M half runs one barrier behind the first
  for (int t = 0; t < ntiles; t += 2) {
#if G256_PH == 2
    ktile2(I0{}, t);
    ktile2(I1{}, t + 1);
#else
    ktile(I0{}, t);
    ktile(I1{}, t + 1);
#endif
  }
  if (wr == 0) G_BAR;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the masked pieces behind the last k-tile
#ifdef G256_STAMP
  if (g.stamps) {
    const uint32_t cyc = (uint32_t)(__builtin_readcyclecounter() - sk0), rt = (uint32_t)(wall_clock64() - sr0);
    uint32_t v = rec;
    if (lane == 32) v = cyc;
    if (lane == 33) v = rt;
    if (lane == 34) v = (uint32_t)ntiles;
    reinterpret_cast<uint32_t*>(g.stamps)[((int64_t)blockIdx.x * 8 + w) * 64 + lane] = v;
  }
#endif
  __syncthreads();     // every wave is past its last fragment read and its last DMA: the ring becomes the staging area

  // ---- epilogue: wave w stages 64 rows x 64 columns of raw accumulators per pass in its own region, then reads them back as row
  // segments (lane -> 8 columns, 8 lanes per row): bias / LayerNorm fold / GELU / rounding on the read-back side, 16-byte stores
  char* const reg = smem + w * REGION;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int tm2 = 0; tm2 < 2; ++tm2) {
      char* dst = reg + (tm2 * 32 + l31) * OPITCH + lh * 64;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f4_t*>(dst + tn * 128 + q * 16) =
              f4_t{acc[2 * p + tm2][tn][4 * q], acc[2 * p + tm2][tn][4 * q + 1], acc[2 * p + tm2][tn][4 * q + 2], acc[2 * p + tm2][tn][4 * q + 3]};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const char* src = reg + rrow * OPITCH + ccol * 4;
    const int mrow = wr * 128 + p * 64 + rrow;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ml = mrow + 8 * j, m = m0 + ml;
      if (m >= g.M) continue;            // ragged last tile
      const f4_t v0 = *reinterpret_cast<const f4_t*>(src + j * 8 * OPITCH);
      const f4_t v1 = *reinterpret_cast<const f4_t*>(src + j * 8 * OPITCH + 16);
      f32x2_t pp[4];
      if constexpr (LN) {   // rstd (acc - mean colsum) + bias
        const float2 st = *reinterpret_cast<const float2*>(smem + LNST_OFF + ml * 8);
        const float rs = st.y, tt = -st.x * st.y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pp[e][0] = __builtin_fmaf(e < 2 ? v0[2 * e] : v1[2 * e - 4], rs, __builtin_fmaf(tt, ccs[2 * e], bcol[2 * e]));
          pp[e][1] = __builtin_fmaf(e < 2 ? v0[2 * e + 1] : v1[2 * e - 3], rs, __builtin_fmaf(tt, ccs[2 * e + 1], bcol[2 * e + 1]));
        }
      } else {
        pp[0] = f32x2_t{v0[0] + bcol[0], v0[1] + bcol[1]}; pp[1] = f32x2_t{v0[2] + bcol[2], v0[3] + bcol[3]};
        pp[2] = f32x2_t{v1[0] + bcol[4], v1[1] + bcol[5]}; pp[3] = f32x2_t{v1[2] + bcol[6], v1[3] + bcol[7]};
      }
      if constexpr (EPI == EPI_RES) {
        // same operations in the same order as linear.hip's EPI_RES epilogue ((acc + bias) + resid; row partials as a balanced tree over
        // 4-column leaves in column order): a row's result does not depend on which of the two kernels its batch size selects
        const float* rp = g.resid + (int64_t)(g.res_mod > 0 ? m % g.res_mod : m) * g.ldres + sn0 + ccol;
        const f4_t r0 = *reinterpret_cast<const f4_t*>(rp), r1 = *reinterpret_cast<const f4_t*>(rp + 4);
        f4_t a = v0 + f4_t{bcol[0], bcol[1], bcol[2], bcol[3]}, b = v1 + f4_t{bcol[4], bcol[5], bcol[6], bcol[7]};
        a += r0; b += r1;
        float* cp = reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + sn0 + ccol;
        *reinterpret_cast<f4_t*>(cp) = a;
        *reinterpret_cast<f4_t*>(cp + 4) = b;
        if (g.xcopy) {
          *reinterpret_cast<u32x4_t*>(g.xcopy + ((int64_t)m * g.ldc + sn0 + ccol) * 2) = u32x4_t{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
          float s1 = ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
          float s2 = countr_sq4(a[0], a[1], a[2], a[3]) + countr_sq4(b[0], b[1], b[2], b[3]);
          s1 += dpp_mov<0xB1>(s1); s2 += dpp_mov<0xB1>(s2);     // lanes l ^ 1
          s1 += dpp_mov<0x4E>(s1); s2 += dpp_mov<0x4E>(s2);     // lanes l ^ 2
          s1 += dpp_mov<0x141>(s1); s2 += dpp_mov<0x141>(s2);   // the other quad of the row's 8 lanes (row_half_mirror)
          if ((lane & 7) == 0) *reinterpret_cast<float2*>(g.stats_out + ((int64_t)m * (g.N >> 6) + (sn0 >> 6)) * 2) = make_float2(s1, s2);
        }
        continue;
      }
      const int64_t o = ((int64_t)m * g.ldc + sn0 + ccol) * 2;
      if constexpr (EPI == EPI_GELU) {
        if (g.C2) *reinterpret_cast<u32x4_t*>(g.C2 + o) = u32x4_t{pack2bf(pp[0][0], pp[0][1]), pack2bf(pp[1][0], pp[1][1]), pack2bf(pp[2][0], pp[2][1]), pack2bf(pp[3][0], pp[3][1])};
#pragma unroll
        for (int e = 0; e < 4; ++e) pp[e] = gelu_sig2(pp[e]);
      }
      const u32x4_t packed = u32x4_t{pack2bf(pp[0][0], pp[0][1]), pack2bf(pp[1][0], pp[1][1]), pack2bf(pp[2][0], pp[2][1]), pack2bf(pp[3][0], pp[3][1])};
      *reinterpret_cast<u32x4_t*>(g.C + o) = packed;
      if constexpr (CONV && EPI == EPI_BF16) countr_gn_row_partials(packed, g.gn_rows, m, g.N, sn0 + ccol, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <bool CONV, int EPI, bool LN>
int launch_big(const BigArgs& a0, hipStream_t s) {
  if (countr_dry_run) return 0;     // (a selection query: countr_gemm_gn_rows)
  BigArgs a = a0;
  a.npf = countr_prefetch_blocks(a.launch_tiles, a.pf, a.pf_bytes);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&g256_kernel<CONV, EPI, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((g256_kernel<CONV, EPI, LN>), dim3(a.launch_tiles + a.npf), dim3(512), LDS_BYTES, s, a);
  COUNTR_LAUNCH_CHECK("countr_gemm(256x256 8-phase)");
}

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

}  // namespace

// nn.Linear forward (bf16 out with optional GELU / pre-activation copy / LayerNorm-fold consumer, or fp32 out = acc + bias + residual with the
// optional LayerNorm producer outputs: attn.proj / mlp.fc2 of the encoder at 32 windows -- inference 8.00 -> 7.88 ms) on 256 x 256 tiles.  Returns 1 when the
// launch does not qualify or the 128-row forms of linear.hip are expected to be faster (the caller then tries those).
int countr_big_linear(const countr_gemm_args* a, hipStream_t s) {
  const int mode = env_int("COUNTR_G256", 1);      // 0: never; 1: where it is expected to win; 2: wherever it qualifies (tests)
  if (mode == 0) return 1;
  if (a->partial || a->nbatch > 1 || a->alpha != 1.0f || a->rowsum_partial) return 1;
  if (a->M < 1 || (a->N % 256) || (a->K % 128) || a->K < 256 || a->N > 8192) return 1;
  const bool res = !a->out_bf16;      // fp32 out = acc + bias + residual (proj / fc2), optionally a LayerNorm producer
  if (res) {
    if (!a->resid || a->act != COUNTR_ACT_NONE || a->C2 || (a->ldc % 4) || (a->ldres % 4) || ((uintptr_t)a->resid & 15)) return 1;
    if ((a->ln_xcopy != nullptr) != (a->ln_stats_out != nullptr) || ((uintptr_t)a->ln_xcopy & 15) || ((uintptr_t)a->ln_stats_out & 7)) return 1;
    if (a->ln_stats || a->ln_colsum) return 1;
  } else if (a->resid || a->ln_xcopy || a->ln_stats_out) return 1;
  if ((a->lda % 8) || (a->ldb % 8) || (a->ldc % 8) || (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C | (uintptr_t)a->C2) & 15)) return 1;
  if ((int64_t)256 * a->lda * 2 + (int64_t)a->K * 2 >= (int64_t)0x7f000000ll || (int64_t)256 * a->ldb * 2 + (int64_t)a->K * 2 >= (int64_t)0x7f000000ll) return 1;
  if (a->bias && ((uintptr_t)a->bias & 15)) return 1;
  const bool ln_in = a->ln_stats != nullptr || a->ln_colsum != nullptr;
  if (ln_in && (!a->ln_stats || !a->ln_colsum || (((uintptr_t)a->ln_stats | (uintptr_t)a->ln_colsum) & 15) || a->ln_nblk != a->K / 64 || a->K > 768)) return 1;
  int epi;
  if (res) epi = EPI_RES;
#ifdef G256_STAMP
  else if (a->act == COUNTR_ACT_NONE) epi = EPI_BF16;      // stamp builds: C2 carries the stamp buffer
#else
  else if (a->act == COUNTR_ACT_NONE && !a->C2) epi = EPI_BF16;
#endif
  else if (a->act == COUNTR_ACT_GELU) epi = EPI_GELU;
  else return 1;
  const long tiles = (long)((a->M + 255) / 256) * (a->N / 256);
  if (mode == 1) {
    // one round of 256 x 256 tiles that uses most of the chip, or many rounds: fc1 at B = 8 is 216 tiles.  (qkv at B = 8 would be 162
    // tiles on 256 CUs: the 192 x 256 form of linear.hip keeps it.)
    const long rounds = (tiles + 255) / 256;
    if (tiles < 200 || tiles * 100 < rounds * 256 * 80) return 1;
  }
  if (!a->bias && a->N > COUNTR_ZERO_VEC_FLOATS) return 1;
  const float* bias = a->bias ? a->bias : countr_zero_vec(a->N);      // the per-device vector of zeros countr_init allocated
  if (!bias) return -1;
  BigArgs g;
  g.A = (const char*)a->A; g.W = (const char*)a->B; g.C = (char*)a->C; g.C2 = (char*)a->C2; g.bias = bias;
  g.M = a->M; g.N = a->N; g.K = a->K; g.lda = (int)a->lda; g.ldw = (int)a->ldb; g.ldc = (int)a->ldc; g.tilesN = a->N / 256;
  g.launch_tiles = (int)tiles; g.pf = (const char*)a->prefetch; g.pf_bytes = a->prefetch_bytes;
  g.resid = a->resid; g.xcopy = (char*)a->ln_xcopy; g.stats_out = a->ln_stats_out; g.ldres = (int)a->ldres; g.res_mod = a->res_mod;
  g.H = g.Wd = g.Cin = g.cpt_log = 0; g.stats_in = a->ln_stats; g.colsum = a->ln_colsum; g.ln_eps = a->ln_eps; g.stamps = nullptr; g.gn_rows = nullptr;
#ifdef G256_STAMP
  g.stamps = (float*)a->C2; g.C2 = nullptr;
#endif
  if (epi == EPI_RES) return launch_big<false, EPI_RES, false>(g, s);
  if (ln_in) return epi == EPI_BF16 ? launch_big<false, EPI_BF16, true>(g, s) : launch_big<false, EPI_GELU, true>(g, s);
  return epi == EPI_BF16 ? launch_big<false, EPI_BF16, false>(g, s) : launch_big<false, EPI_GELU, false>(g, s);
}

int countr_lean_conv_rows(const countr_gemm_args* a, hipStream_t s, int row0);     // linear.hip

// 3x3 convolution forward / dgrad as implicit GEMM (A = IM2ROW view of an NHWC bf16 map, B = [Cout][9 Cin] weights) on 256 x 256 tiles.
// Split rounds: a grid of 4.5 rounds of workgroups (the 192x192 map at B = 8: 1152 tiles on 256 CUs) costs five tile-times, the last
// one on half the chip.  When the tiles behind the last FULL round are at most half a round, those rows run as a tail launch of
// linear.hip's 128 x 256 kernel instead (twice the workgroups at half the work each: one full round of a ~0.6x tile-time).  Results
// do not depend on the split: the two kernels agree bit for bit (tests/test_gemm_gpu.py).
int countr_big_conv(const countr_gemm_args* a, hipStream_t s) {
  const int mode = env_int("COUNTR_G256", 1);
  if (mode == 0) return 1;
#ifndef G256_STAMP
  if (a->C2) return 1;
#endif
  if (a->partial || a->nbatch > 1 || a->alpha != 1.0f || a->rowsum_partial || a->resid || a->act != COUNTR_ACT_NONE || !a->out_bf16) return 1;
  if (a->M < 1 || (a->N % 256) || a->N > 4096 || a->K != 9 * a->Cin || a->H < 2 || a->W < 2) return 1;
  if (a->Cin != 64 && a->Cin != 128 && a->Cin != 256 && a->Cin != 512) return 1;     // k-tiles per tap a power of two, 9 Cin / 64 even
  if (a->Cin == 64) return 1;                                                          // (9 k-tiles: odd)
  if ((a->ldb % 8) || (a->ldc % 8) || (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15)) return 1;
  if ((int64_t)(a->M + 2 * a->W + 2 + 256) * a->Cin * 2 >= (int64_t)0x7f000000ll || (int64_t)a->N * a->ldb * 2 >= (int64_t)0x7f000000ll) return 1;
  if (a->bias && ((uintptr_t)a->bias & 15)) return 1;
  if (a->gn_rows && ((uintptr_t)a->gn_rows & 7)) return 1;
  const long tilesN = a->N / 256, tiles = (long)((a->M + 255) / 256) * tilesN;
  // split rounds: full rounds here, at most half a round of tiles behind them on the 128-row kernel
  long head = tiles;
  {
    const long full = (tiles / 256) * 256, rem = tiles - full;
    // (rem >= 96: the tail launch -- 2 rem workgroups of the 128-row kernel -- must itself fill most of a round; 96x96 at B = 8 has
    // rem = 32: 64 tail workgroups cost a whole tile-time on a quarter of the chip, 62 + 31 us against 89 us on the 128-row kernel alone)
    if (full >= 256 && rem >= 96 && rem <= 128 && (full % tilesN) == 0 && mode != 0) head = full;
  }
  if (mode == 1 && head == tiles) {
    const long rounds = (tiles + 255) / 256;
    if (tiles < 200 || tiles * 100 < rounds * 256 * 80) return 1;
  }
  if (!a->bias && a->N > COUNTR_ZERO_VEC_FLOATS) return 1;
  const float* bias = a->bias ? a->bias : countr_zero_vec(a->N);      // the per-device vector of zeros countr_init allocated
  if (!bias) return -1;
  if (head < tiles) {
    // the tail first checks that it qualifies (same conditions as ours, plus its own) by launching: it runs BEHIND the head on the stream
    // either way, so launch order on the host is free; launching it first lets a refusal fall back to the single launch
    const int rc = countr_lean_conv_rows(a, s, (int)(head / tilesN) * 256);
    if (rc < 0) return rc;
    if (rc == 1) head = tiles;
  }
  BigArgs g;
  g.A = (const char*)a->A; g.W = (const char*)a->B; g.C = (char*)a->C; g.C2 = nullptr; g.bias = bias;
  g.M = a->M; g.N = a->N; g.K = a->K; g.lda = 0; g.ldw = (int)a->ldb; g.ldc = (int)a->ldc; g.tilesN = (int)tilesN;
  g.launch_tiles = (int)head; g.pf = nullptr; g.pf_bytes = 0;
  g.resid = nullptr; g.xcopy = nullptr; g.stats_out = nullptr; g.ldres = 0; g.res_mod = 0;
  g.H = a->H; g.Wd = a->W; g.Cin = a->Cin; g.cpt_log = a->Cin == 128 ? 1 : a->Cin == 256 ? 2 : 3;
  g.stats_in = nullptr; g.colsum = nullptr; g.ln_eps = 0.f; g.stamps = nullptr; g.gn_rows = a->gn_rows;
#ifdef G256_STAMP
  g.stamps = (float*)a->C2;
#endif
  return launch_big<true, EPI_BF16, false>(g, s);
}
