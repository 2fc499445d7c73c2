// Lean nn.Linear forward for gfx950: out = epilogue(x W^T + b), x [M,K] bf16 row-major, W [N,K] bf16 row-major, FULL tiles only
// (M % 128 == 0, N % 128 == 0, K % 64 == 0, 16-byte aligned operands).  countr_gemm() routes the (ROW, ROW) bf16 launches that qualify
// here; everything else (ragged shapes, fp32 parity mode, dgrad / wgrad / convolutions) stays on gemm_kernel in gemm.hip.
//
// Why a second kernel.  The generic kernel's workgroup spends ~14 k cycles outside its main loop on the encoder shapes
// (profiles/r2_gemm_kernel_timeline.txt): bounds-checked, 64-bit, per-element-predicated address arithmetic issued by ONE wave per SIMD
// (a lone wave issues a VALU every ~5.5 cycles, profiles/r2_issue_microbench.txt), 16-wide MFMAs whose issue slots hide almost nothing, a
// libm-shaped GELU (~20 VALU per element), and loader waves that have left before the epilogue starts.  Here:
//   * v_mfma_f32_32x32x16_bf16 (half the matrix instructions per flop; ~4 issue slots of cover under each), 64x64 wave tile = 2x2 tiles,
//     fragments by ds_read_b128 from XOR-swizzled 128-byte rows (swizzle on the LDS-DMA source address, slot = chunk ^ ((row >> 1) & 7));
//   * W rows are read in the permuted order p(8a + 4h + b) = 16h + 4a + b, so that a lane's 16 accumulator registers of a tile are 16
//     CONSECUTIVE output columns;
//   * no bounds checks, 32-bit offsets from scalar bases, bias / residual fetched before the main loop;
//   * epilogue: raw fp32 accumulators -> LDS -> row segments; bias, GELU, residual, rounding happen on the read-back side, where every
//     wave of the workgroup takes part (in the wave-specialised form the four loader waves do half of it);
//   * GELU as x * sigmoid(x (c0 + c1 x^2 + c2 x^4)) on packed fp32 pairs: |error| <= 2.6e-5 absolute against the erf form (bf16 output
//     resolution is 4e-3 relative), 6 VALU + 2 transcendentals per element pair-half instead of ~20.
// Launch forms: <NLD = 4> 4 compute + 4 loader waves, 3-stage LDS ring, one workgroup per CU (grids of <= 256 tiles);
//               <NLD = 0> 4 waves that stage and multiply, 2 stages, two workgroups per CU (bigger grids).
// Reference call sites: models_crossvit.py:62,65 (Mlp), :84,92 (Attention qkv / proj), :115-127 (CrossAttention), models_mae_cross.py:152.
#include "common.hpp"
#include "../../include/countr_hip.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

#ifndef LIN_ABL
#define LIN_ABL 0   // timing experiments (tools/exp_lin.sh, results are WRONG): 1 = no MFMA, 2 = no fragment reads, 3 = no DMA after the first tiles
#endif

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void* lds_vptr_t;
typedef const __attribute__((address_space(1))) void* glb_vptr_t;
typedef __attribute__((address_space(3))) const char* lds_cptr_t;

enum { EPI_BF16 = 0, EPI_GELU = 1, EPI_RES = 2, EPI_GBWD = 3 };   // GBWD: out = acc * GELU'(C2), C2 = saved bf16 pre-activation (fc2 input gradient)

struct LinArgs {
  const char* A;       // [M, lda] bf16
  const char* W;       // [N, ldw] bf16
  char* C;             // bf16 or fp32 [M, ldc]
  char* C2;            // EPI_GELU: optional bf16 pre-activation copy
  const float* bias;   // [N] or null
  const float* resid;  // EPI_RES: fp32 [*, ldres]
  int M, N, K;
  int lda, ldw, ldc, ldres;   // elements
  int res_mod, tilesN;
  const char* pf; long long pf_bytes; int launch_tiles, npf;   // workgroups < npf only read [pf, pf + pf_bytes) and leave (cache warm-up hint); launch_tiles tile workgroups follow
  int tile_m0;         // first row tile of this launch (a launch may cover the LAST row tiles of a problem only: gemm256.hip's split rounds)
  int H, Wd, Cin;      // CONV (3x3, pad 1, NHWC): A is the map [B, H, Wd, Cin], row m = pixel, k = tap * Cin + c
  // LayerNorm folding (LN template flag).  Producer (EPI_RES): xcopy = bf16 copy of the fp32 output, stats_out[m][N/64][2] = {sum, sum of
  // squares} of the fp32 output row over each 64-column block.  Consumer (EPI_BF16 / EPI_GELU): A is that bf16 copy of the UN-normalised
  // rows, W carries gamma, stats_in[m][K/64][2] the producer's partials, colsum[n] = sum_k W[n][k]:
  //   out = rstd_m (acc - mean_m colsum_n) + bias_n  ==  LayerNorm(x)_m W0^T + b0   (W = gamma o W0, bias = b0 + W0 beta).
  char* xcopy;
  float* stats_out;
  const float* stats_in;
  const float* colsum;
  float ln_eps;
  float* gn_rows;      // CONV (optional): GroupNorm row partials of the rounded output (common.hpp::countr_gn_row_partials)
};

constexpr int STAGE_BYTES = 32768, B_OFF = 16384;
constexpr int OPITCH = 64 * 4 + 16;   // staging pitch of a 64-column fp32 row

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ uint32_t lds_u32(const char* p) { return (uint32_t)(uintptr_t)(lds_cptr_t)p; }

template <int OFF> __device__ __forceinline__ bf16x8_t ds_read128(uint32_t a) {
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
// wait until at most PENDING LDS reads are outstanding; the fragments are threaded through so that their MFMAs stay behind the wait
template <int PENDING> __device__ __forceinline__ void frag_wait(bf16x8_t (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(PENDING));
}

template <int PENDING> __device__ __forceinline__ void frag_wait5(bf16x8_t (&f)[4], bf16x8_t& f2) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f2) : "n"(PENDING));
}

// packed form of common.hpp's gelu_fast (same operations on fp32 pairs: bit-identical results)
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

// WMB = 128-row blocks per workgroup tile (1: 128x128, 4 compute waves; 2: 256x128, 8 compute waves -- 2/3 of the staged bytes per MFMA,
// for grids that still fill the chip with half as many tiles); NLD = loader waves (0: every wave stages and multiplies).
// CONV: the A operand is the 3x3 im2row view of an NHWC map (implicit GEMM, forward and -- with dgrad-form weights -- dgrad of the
// density-head / exemplar convolutions): only the LDS-DMA source addresses differ, the tile in LDS and everything behind it is the same.
// TM = 32-row MFMA tiles per compute wave (2: 64x64 wave tiles; 3: 96x64 -- the 192 x 256 workgroup tile, 8 compute + 4 loader waves on
// a TWO-stage ring of 56 KB: 1.5 x the work of a 256x128 / 128x256 tile per workgroup, for shapes whose 256x128 grid is a little over one
// round of 256 workgroups -- qkv at B = 8: 324 workgroups = two rounds, 216 of these = one -- and 86 % of the staged bytes per MFMA).
template <int WMB, int NLD, int EPI, int STAGES, bool CONV = false, int WNB = 1, bool LN = false, int TM = 2>
__global__ __launch_bounds__(256 * WMB * WNB + 64 * NLD, NLD ? ((STAGES == 2 && TM == 2) ? 4 : (4 * WMB * WNB + NLD) / 4) : 2) void lin_kernel(const LinArgs g) {
  constexpr bool SPEC = NLD > 0;
  constexpr bool T3 = TM == 3;
  constexpr bool TWO = SPEC && STAGES == 2 && !T3;     // wave-specialised with a 2-stage ring: TWO workgroups per CU (64 KB, 128 VGPRs each)
  static_assert(TM == 2 || TM == 3, "wave tile = 64x64 or 96x64");
  static_assert(!T3 || (SPEC && STAGES == 2 && WMB == 1 && WNB == 2 && NLD == 4 && EPI != EPI_RES), "192x256 form: bf16 outputs, 2-stage ring");
  static_assert(SPEC ? (STAGES >= 2 && STAGES <= 5) : (STAGES == 2 && WMB == 1), "ring depth / plain form");
  static_assert(!TWO || (WMB == 1 && WNB == 1 && NLD == 4), "two-per-CU form");
  static_assert(WMB * WNB <= 2 && (SPEC || WMB * WNB == 1), "tile = 128x128, 256x128, 128x256 or 192x256");
  constexpr int NCW = 4 * WMB * WNB;            // compute waves, (2 WMB) x (2 WNB) sub-tiles of (32 TM) x 64
  constexpr int NLW = SPEC ? NLD : 4;           // waves that stage tiles
  static_assert(NLW == 4, "four staging waves");
  constexpr int RW = 32 * TM;                   // rows of a compute wave's sub-tile
  constexpr int BMt = 2 * RW * WMB, BNt = 128 * WNB, SUBN = 2 * WNB;   // SUBN = 64-column sub-tiles per row of sub-tiles
  constexpr int A_BYTES = BMt * 128, STAGE_BYTES = A_BYTES + BNt * 128;
  constexpr int PA = BMt / (8 * NLW), PB = BNt / (8 * NLW);   // 1-KiB pieces per operand per staging wave per k-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  if ((int)blockIdx.x < g.npf) { countr_prefetch_range(g.pf, g.pf_bytes, (int)blockIdx.x, g.npf); return; }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = SPEC && wv >= NCW;
  const int cw = loader ? wv - NCW : wv;         // index inside its role (plain form: both)
  // XCD-aware tile order (workgroup b runs on XCD b % 8): every XCD owns one contiguous range of the (tile_m, tile_n) space
  int lt;
  {
    const int bt = (int)blockIdx.x - g.npf;      // (npf % 8 == 0: the XCD of tile workgroup bt is bt % 8)
    const int nt = g.launch_tiles, q = nt >> 3, r = nt & 7, x = bt & 7, j = bt >> 3;
    lt = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
  }
  const int tile_q = lt / g.tilesN, tile_n = lt - tile_q * g.tilesN, tile_m = tile_q + g.tile_m0;
  const int m0 = tile_m * BMt, n0 = tile_n * BNt;
  const int ntiles = g.K >> 6;

  // ---- LDS-DMA set-up (staging waves): piece (pass i, wave w) = rows [8 NLW i + 8 w, +8) of the operand tile, lane -> (row, 16-byte slot)
  const int drow = cw * 8 + (lane >> 3);
  const int dchunk = (lane & 7) ^ (((cw & 1) << 2) | (lane >> 4));        // = slot ^ ((row >> 1) & 7), pass-independent
  const uint32_t voffA = CONV ? (uint32_t)(((m0 + drow) * g.Cin + dchunk * 8) * 2) : (uint32_t)((drow * g.lda + dchunk * 8) * 2);
  const uint32_t voffB = (uint32_t)((drow * g.ldw + dchunk * 8) * 2);
  // Ragged M (M % tile rows != 0): bit i of rowmask <=> this lane's row of pass i exists; the other lanes stage zeros (an offset outside
  // the descriptor) and the epilogue skips their rows.  CONV: bit t of vmask[i] <=> tap t of this lane's pixel in pass i lies inside
  // the image (zero padding otherwise); the pixel of pass i + 1 is 8 NLW pixels further in raster order
  uint32_t rowmask = 0;
#pragma unroll
  for (int i = 0; i < PA; ++i) rowmask |= (m0 + drow + 8 * NLW * i < g.M) ? (1u << i) : 0u;
  uint32_t vmask[CONV ? PA : 1];
  if constexpr (CONV) {
    if (!SPEC || loader) {
      const int m = m0 + drow;
      int x = m % g.Wd, y = (m / g.Wd) % g.H;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        uint32_t vm = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
          if ((rowmask >> i & 1u) && (unsigned)(y + t / 3 - 1) < (unsigned)g.H && (unsigned)(x + t % 3 - 1) < (unsigned)g.Wd) vm |= 1u << t;
        vmask[i] = vm;
        x += 8 * NLW;
        while (x >= g.Wd) { x -= g.Wd; y = (y + 1 == g.H) ? 0 : y + 1; }
      }
    }
  }
  // The staging loads are MUBUF LDS-DMA (buffer_load_dwordx4 ... offen lds): address = descriptor base + 32-bit lane offset (loop
  // invariant) + scalar offset (k-tile and pass advance, SALU only) -- no vector address arithmetic per piece -- and a lane whose
  // offset lies beyond the descriptor's size gets ZEROS (the convolution's padding taps) instead of needing a second source pointer.
  // (The 64-bit-address form, global_load_lds with a VGPR pair per lane, issued at ~95 cycles per piece and wave and was what bounded
  // the wave-specialised loop: profiles/r3_linear_stamps.txt.)
  // CONV: the descriptor base sits (Wd + 1) pixels in front of the map so that every tap shift is a non-negative scalar offset.
  const int64_t cshift = CONV ? (int64_t)(g.Wd + 1) * g.Cin * 2 : 0;
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(CONV ? g.A - cshift : g.A + (int64_t)m0 * g.lda * 2), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc((void*)(g.W + (int64_t)n0 * g.ldw * 2), 0, 0x7ffffff0, 0x00020000);
  const uint32_t passA = (uint32_t)(CONV ? g.Cin : g.lda) * (16u * NLW), passB = (uint32_t)g.ldw * (16u * NLW);   // 8 NLW rows further, bytes
  auto issue = [&](int t, int slot) {
#if LIN_ABL == 3
    if (t >= 2) return;
#endif
    char* dst = smem + slot * STAGE_BYTES + cw * 1024;
    uint32_t kb = (uint32_t)t * 128u;              // byte offset of k-tile t inside a W row
    if constexpr (CONV) {
      // k-tile t = 64 channels of one tap (64 | Cin): scalar offset = tap shift + channel block, per-lane pixel offset, and one bit
      // test per piece for the zero padding (offset 2^31 is out of the descriptor's range: the DMA writes zeros)
      int tap, cb;
      countr_conv_ktile(t, g.Cin, tap, cb);
      kb = (uint32_t)(tap * g.Cin + cb) * 2u;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const uint32_t so = (uint32_t)(cshift + ((int64_t)(dy * g.Wd + dx) * g.Cin + cb) * 2);
      const uint32_t bit = 1u << tap;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        // (named variable on purpose: with the conditional expression written as the builtin's argument the HOST pass of hipcc 7.2
        // silently drops the kernel's stub -- undefined symbol at load time, no diagnostic)
        const uint32_t vo = (vmask[i] & bit) ? voffA : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_vptr_t)(dst + i * NLW * 1024), 16, vo, so + i * passA, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const uint32_t vo = (rowmask >> i & 1u) ? voffA : 0x80000000u;     // (a named variable: see the note in the CONV branch)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_vptr_t)(dst + i * NLW * 1024), 16, vo, t * 128 + i * passA, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_vptr_t)(dst + A_BYTES + i * NLW * 1024), 16, voffB, kb + i * passB, 0, 0);
  };

  // loader waves start the ring before anything else; the memory clobber keeps the prefetch loads below BEHIND these in issue order
  // (the first wait of the loader loop counts them: vmcnt = the younger tiles + NPRE younger loads)
  if constexpr (SPEC) {
    if (loader) {
#pragma unroll
      for (int s_ = 0; s_ < STAGES - 1; ++s_)
        if (s_ < ntiles) issue(s_, s_);
    }
    asm volatile("" ::: "memory");
  }

  // ---- read-back geometry (epilogue).  A unit = 32 rows x 64 columns of one compute wave's 64x64 sub-tile.  SPEC: compute wave c
  // finishes rows [0, 32) of its own sub-tile, loader l rows [32, 64) of sub-tiles l, l + 4 (, ...); plain: both halves of its own.
  constexpr bool OBF = EPI != EPI_RES;
  constexpr int RING = STAGES * STAGE_BYTES, STG = SPEC ? NCW * 64 * OPITCH : 0, LNST_OFF = RING > STG ? RING : STG;   // row statistics behind ring / staging
  constexpr int NUNITS = SPEC ? NCW / 4 : 1;           // read-back units of a loader wave (a compute wave has one)
  const int sub0 = loader ? (cw & 3) : wv;             // first sub-tile this wave reads back (NCW is a multiple of 4: same column half)
  const int sn0 = n0 + (sub0 % SUBN) * 64;             // (sub0 + 4 u shares the column block: SUBN divides 4)
  // bf16 out: lane -> 8 columns (2 fp32 chunks), 8 lanes per row;  fp32 out: lane -> 4 columns, 16 lanes per row
  const int ccol = OBF ? (lane & 7) * 8 : (lane & 15) * 4;
  const int rrow = OBF ? (lane >> 3) : (lane >> 4);     // + 8 j (bf16) / 4 j (fp32)
  constexpr int RSTEP = OBF ? 8 : 4, NIT_HALF = 32 / RSTEP;     // iterations per 32 rows
  const int half0 = loader ? 1 : 0;
  float bcol[OBF ? 8 : 4];
  {
    const float4 b0 = *reinterpret_cast<const float4*>(g.bias + sn0 + ccol);
    bcol[0] = b0.x; bcol[1] = b0.y; bcol[2] = b0.z; bcol[3] = b0.w;
    if constexpr (OBF) {
      const float4 b1 = *reinterpret_cast<const float4*>(g.bias + sn0 + ccol + 4);
      bcol[4] = b1.x; bcol[5] = b1.y; bcol[6] = b1.z; bcol[7] = b1.w;
    }
  }
  constexpr bool LNIN = LN && OBF, LNOUT = LN && !OBF;
  float ccs[LNIN ? 8 : 1];
  if constexpr (LNIN) {
    const float4 c0 = *reinterpret_cast<const float4*>(g.colsum + sn0 + ccol), c1 = *reinterpret_cast<const float4*>(g.colsum + sn0 + ccol + 4);
    ccs[0] = c0.x; ccs[1] = c0.y; ccs[2] = c0.z; ccs[3] = c0.w; ccs[4] = c1.x; ccs[5] = c1.y; ccs[6] = c1.z; ccs[7] = c1.w;
    // mean / rstd of the tile's rows from the producer's 64-column partials -> LDS behind the ring (read by the epilogue).  The compute
    // waves do it (they wait for the first k-tile anyway and issue no DMA of their own in the wave-specialised forms)
    if (!loader && tid < BMt) {
      const int m = m0 + tid;
      float s1 = 0.f, s2 = 0.f;
      if (m < g.M) {
        const int nblk = g.K >> 6;
        const float4* sp = reinterpret_cast<const float4*>(g.stats_in + (int64_t)m * nblk * 2);
        if (nblk == 12) {   // K = 768 (ViT-B): all six loads in flight at once (the runtime-bound loop below waits for each in turn: +4 us)
          float4 v[6];
#pragma unroll
          for (int b2 = 0; b2 < 6; ++b2) v[b2] = sp[b2];
#pragma unroll
          for (int b2 = 0; b2 < 6; ++b2) { s1 += v[b2].x + v[b2].z; s2 += v[b2].y + v[b2].w; }
        } else {
          for (int b2 = 0; b2 < nblk / 2; ++b2) { const float4 v = sp[b2]; s1 += v.x + v.z; s2 += v.y + v.w; }
          if (nblk & 1) { const float2 v = *reinterpret_cast<const float2*>(g.stats_in + ((int64_t)m * nblk + nblk - 1) * 2); s1 += v.x; s2 += v.y; }
        }
      }
      // (explicit fma: the two kernels that serve this epilogue -- linear.hip, gemm256.hip -- must not differ by a compiler's contraction choice)
      const float inv = 1.f / (float)g.K, mean = s1 * inv, ex2 = s2 * inv, var = fmaxf(__builtin_fmaf(-mean, mean, ex2), 0.f);
      *reinterpret_cast<float2*>(smem + LNST_OFF + tid * 8) = make_float2(mean, rsqrtf(var + g.ln_eps));
    }
  }
  // residual prefetch (SPEC): the row segments this wave will add, fetched before the main loop
  typedef __attribute__((ext_vector_type(4))) float f4_t;
  constexpr bool RPRE = SPEC && !TWO && EPI == EPI_RES;
  f4_t rpre[RPRE ? 8 : 1];          // first unit only (a loader's further units fetch at use: register budget)
  if constexpr (RPRE) {
    {
      constexpr int u = 0;
      const int mrow0 = m0 + ((sub0 + 4 * u) / SUBN) * RW + half0 * 32 + rrow;
      if (g.res_mod > 0) {   // row modulo (pos-embed adds): one division, then steps of 4 rows with a conditional wrap
        int mr = mrow0 % g.res_mod;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rpre[8 * u + j] = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(g.resid + (int64_t)mr * g.ldres + sn0 + ccol));
          mr += 4; if (mr >= g.res_mod) mr -= g.res_mod;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)   // (rows beyond a ragged M are clamped: fetched, never stored)
          rpre[8 * u + j] = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(g.resid + (int64_t)min(mrow0 + 4 * j, g.M - 1) * g.ldres + sn0 + ccol));
      }
    }
  }

  // ---- fragment addresses (compute waves): byte offsets inside a stage for k-step kk; tile tm / tn = +4096 in the offset field
  const int wm = cw / SUBN, wn = cw % SUBN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int prow = ((l31 >> 2) & 1) * 16 + ((l31 >> 3) & 3) * 4 + (l31 & 3);            // W row permutation p(l31)
  const int swx = (l31 >> 1) & 7, sww = (prow >> 1) & 7;
  uint32_t xoff[4], woff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    xoff[kk] = (uint32_t)((wm * RW + l31) * 128 + (((kk * 2 + lh) ^ swx) << 4));   // (RW / 2 is a multiple of 8: the swizzle term does not see wm)
    woff[kk] = (uint32_t)(A_BYTES + (wn * 64 + prow) * 128 + (((kk * 2 + lh) ^ sww) << 4));
  }
  f32x16_t acc[TM][2];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  bf16x8_t fr[(TWO || T3) ? 2 : 4][4];   // [set][x0, x1, w0, w1]; set = k-step (two-per-CU and 192-row forms: alternating)
  bf16x8_t fr2[T3 ? 2 : 1];              // [set] x2: the third 32-row tile of the 96-row wave tile
  uint32_t sbase = lds_u32(smem);
  uint32_t xa, wa;
  auto addr = [&](auto KK) { constexpr int kk = decltype(KK)::value; xa = sbase + xoff[kk]; wa = sbase + woff[kk]; };
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
#if LIN_ABL == 2
#define LIN_RD(SET, Q, OFF, A) fr[SET][Q] = __builtin_bit_cast(bf16x8_t, u32x4_t{A, (uint32_t)(OFF), (uint32_t)(Q), 1u})
#else
#define LIN_RD(SET, Q, OFF, A) fr[SET][Q] = ds_read128<OFF>(A)
#endif
#if LIN_ABL == 1
#define LIN_MM(SET, TM, TN) { const u32x4_t a_ = __builtin_bit_cast(u32x4_t, fr[SET][2 + TN]), b_ = __builtin_bit_cast(u32x4_t, fr[SET][TM]); \
        acc[TM][TN][0] += __uint_as_float(a_[0] ^ b_[0]); acc[TM][TN][5] += __uint_as_float(a_[1] ^ b_[1]); \
        acc[TM][TN][10] += __uint_as_float(a_[2] ^ b_[2]); acc[TM][TN][15] += __uint_as_float(a_[3] ^ b_[3]); }
#else
#define LIN_MM(SET, TM, TN) acc[TM][TN] = COUNTR_MFMA_32X32X16(fr[SET][2 + TN], fr[SET][TM], acc[TM][TN], 0, 0, 0)
#endif
#define LIN_SB __builtin_amdgcn_sched_barrier(0)
  // MFMAs of set U with the four reads of set R (addresses xa / wa) between them
#define LIN_STEP_RD(U, R) LIN_MM(U, 0, 0); LIN_SB; LIN_RD(R, 0, 0, xa); LIN_SB; LIN_MM(U, 0, 1); LIN_SB; LIN_RD(R, 2, 0, wa); LIN_SB; \
                          LIN_MM(U, 1, 0); LIN_SB; LIN_RD(R, 1, 4096, xa); LIN_SB; LIN_MM(U, 1, 1); LIN_SB; LIN_RD(R, 3, 4096, wa); LIN_SB
#define LIN_STEP(U) LIN_MM(U, 0, 0); LIN_MM(U, 0, 1); LIN_MM(U, 1, 0); LIN_MM(U, 1, 1); LIN_SB
#define LIN_RD4(SET) LIN_RD(SET, 0, 0, xa); LIN_RD(SET, 1, 4096, xa); LIN_RD(SET, 2, 0, wa); LIN_RD(SET, 3, 4096, wa)
  // 96-row wave tile: the third x fragment and its two MFMAs
#if LIN_ABL == 2
#define LIN_RDX2(SET) fr2[SET] = __builtin_bit_cast(bf16x8_t, u32x4_t{xa, 8192u, 9u, 1u})
#else
#define LIN_RDX2(SET) fr2[SET] = ds_read128<8192>(xa)
#endif
#if LIN_ABL == 1
#define LIN_MM2(SET, TN) { const u32x4_t a_ = __builtin_bit_cast(u32x4_t, fr[SET][2 + TN]), b_ = __builtin_bit_cast(u32x4_t, fr2[SET]); \
        acc[TM - 1][TN][0] += __uint_as_float(a_[0] ^ b_[0]); acc[TM - 1][TN][7] += __uint_as_float(a_[2] ^ b_[3]); }
#else
#define LIN_MM2(SET, TN) acc[TM - 1][TN] = COUNTR_MFMA_32X32X16(fr[SET][2 + TN], fr2[SET], acc[TM - 1][TN], 0, 0, 0)
#endif
#define LIN_STEP6(U) LIN_MM(U, 0, 0); LIN_MM(U, 0, 1); LIN_MM(U, 1, 0); LIN_MM(U, 1, 1); LIN_MM2(U, 0); LIN_MM2(U, 1); LIN_SB
#ifdef LIN_STAMP   // s_memtime anatomy (tools/stamp_lin.py): loaders [1] load wait [2] barrier [3] DMA issue; compute waves [2] barrier
#define LSTAMP(x) const uint64_t x = __builtin_readcyclecounter()
#else
#define LSTAMP(x)
#endif

  if constexpr (SPEC) {
    if (loader) {
      // loader waves: keep STAGES-1 tiles in flight; tile t has landed when at most (STAGES-2) tiles' worth of loads are outstanding
      constexpr int PER = PA + PB;
      constexpr int NPRE = (OBF ? 2 : 1) + (RPRE ? 8 : 0) + (LNIN ? 2 : 0);   // bias / residual loads issued after the first STAGES-1 tiles
      static_assert((STAGES - 2) * PER + NPRE <= 63, "vmcnt immediate");
      int islot = STAGES - 1;
#ifdef LIN_STAMP
      uint64_t sk1 = 0, sk2 = 0, sk3 = 0;
      const uint64_t sk0 = __builtin_readcyclecounter(), sr0 = wall_clock64();
#endif
      for (int t = 0; t < ntiles; ++t) {
        LSTAMP(ua);
        if (t == 0 && STAGES - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER + NPRE) : "memory");
        else if (t + STAGES - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LSTAMP(ub);
        __builtin_amdgcn_s_barrier();
        LSTAMP(uc);
        if (t + STAGES - 1 < ntiles) issue(t + STAGES - 1, islot);
        islot = (islot + 1 == STAGES) ? 0 : islot + 1;
#ifdef LIN_STAMP
        { LSTAMP(ud); sk1 += ub - ua; sk2 += uc - ub; sk3 += ud - uc; }
#endif
      }
#ifdef LIN_STAMP
      if (lane == 0 && g.C2 && EPI != EPI_GELU) {
        float* d = reinterpret_cast<float*>(g.C2) + ((int64_t)blockIdx.x * (NCW + NLD) + wv) * 8;
        d[0] = (float)(__builtin_readcyclecounter() - sk0); d[1] = (float)sk1; d[2] = (float)sk2; d[3] = (float)sk3;
        d[5] = (float)ntiles; d[6] = 1.f; d[7] = (float)(wall_clock64() - sr0);
      }
#endif
      __builtin_amdgcn_s_barrier();   // ring free (the compute waves' last in-loop barrier): every wave is past its last tile
    } else {
      // compute waves.  One fragment set per k-step of a tile; the reads of k-step s+2 are issued one by one BETWEEN the MFMAs of
      // k-step s (a ds_read_b128 issues under the 32 cycles of the MFMA in front of it), so a set has a full k-step (128 matrix
      // cycles) to land.  At the tile boundary: k-step 2's MFMAs (no reads), every read of tile t landed, barrier t+1 (the loaders may
      // now refill tile t's slot; tile t+1 is visible), then k-step 3's MFMAs carry the EIGHT reads of tile t+1's k-steps 0 and 1.
      // The loop body is the same for every tile, the last included: its barrier is the "ring free" barrier the loaders join after
      // their loop, and its eight look-ahead reads fetch stale ring bytes that nobody uses (a peeled last tile made the register
      // allocator copy all 64 accumulator registers once per tile).
      int slot = 0;
#ifdef LIN_STAMP
      uint64_t ck2 = 0;
      const uint64_t ck0 = __builtin_readcyclecounter(), cr0 = wall_clock64();
#endif
      __builtin_amdgcn_s_barrier();
      if constexpr (TWO) {
        // 128-VGPR budget: two fragment sets, reads one k-step ahead in a block; the other workgroup on the CU covers the bubbles
        addr(K0{}); LIN_RD4(0); LIN_SB;
        for (int t = 0; t < ntiles; ++t) {
          addr(K1{}); LIN_RD4(1); frag_wait<4>(fr[0]); LIN_STEP(0);
          addr(K2{}); LIN_RD4(0); frag_wait<4>(fr[1]); LIN_STEP(1);
          addr(K3{}); LIN_RD4(1); frag_wait<4>(fr[0]); LIN_STEP(0);
          frag_wait<0>(fr[1]);
          slot ^= 1;
          __builtin_amdgcn_s_barrier();
          sbase = lds_u32(smem) + slot * STAGE_BYTES;
          addr(K0{}); LIN_RD4(0); LIN_SB;
          LIN_STEP(1);
        }
        frag_wait<0>(fr[0]);
      } else if constexpr (T3) {
        // 168-VGPR budget (12 waves), 96 accumulator registers: two fragment sets, reads one k-step ahead in a block; the SIMD's other
        // compute wave covers the bubbles.  Two stages: the loaders refill the stage a tile has just left while the next is multiplied
        addr(K0{}); LIN_RD4(0); LIN_RDX2(0); LIN_SB;
        for (int t = 0; t < ntiles; ++t) {
          addr(K1{}); LIN_RD4(1); LIN_RDX2(1); frag_wait5<5>(fr[0], fr2[0]); LIN_STEP6(0);
          addr(K2{}); LIN_RD4(0); LIN_RDX2(0); frag_wait5<5>(fr[1], fr2[1]); LIN_STEP6(1);
          addr(K3{}); LIN_RD4(1); LIN_RDX2(1); frag_wait5<5>(fr[0], fr2[0]); LIN_STEP6(0);
          frag_wait5<0>(fr[1], fr2[1]);
          slot ^= 1;
          __builtin_amdgcn_s_barrier();
          sbase = lds_u32(smem) + slot * STAGE_BYTES;
          addr(K0{}); LIN_RD4(0); LIN_RDX2(0); LIN_SB;
          LIN_STEP6(1);
        }
        frag_wait5<0>(fr[0], fr2[0]);
      } else {
      addr(K0{}); LIN_RD4(0);
      addr(K1{}); LIN_RD4(1);
      LIN_SB;
      for (int t = 0; t < ntiles; ++t) {
        addr(K2{}); frag_wait<4>(fr[0]); LIN_STEP_RD(0, 2);
        addr(K3{}); frag_wait<4>(fr[1]); LIN_STEP_RD(1, 3);
        frag_wait<4>(fr[2]); LIN_STEP(2);
        frag_wait<0>(fr[3]);
        slot = (slot + 1 == STAGES) ? 0 : slot + 1;
        LSTAMP(va);
        __builtin_amdgcn_s_barrier();
#ifdef LIN_STAMP
        { LSTAMP(vb); ck2 += vb - va; }
#endif
        sbase = lds_u32(smem) + slot * STAGE_BYTES;
        addr(K0{});
        LIN_MM(3, 0, 0); LIN_SB; LIN_RD(0, 0, 0, xa); LIN_RD(0, 2, 0, wa); LIN_SB;
        LIN_MM(3, 0, 1); LIN_SB; LIN_RD(0, 1, 4096, xa); LIN_RD(0, 3, 4096, wa); LIN_SB;
        addr(K1{});
        LIN_MM(3, 1, 0); LIN_SB; LIN_RD(1, 0, 0, xa); LIN_RD(1, 2, 0, wa); LIN_SB;
        LIN_MM(3, 1, 1); LIN_SB; LIN_RD(1, 1, 4096, xa); LIN_RD(1, 3, 4096, wa); LIN_SB;
      }
      frag_wait<0>(fr[0]); frag_wait<0>(fr[1]);   // the stale look-ahead reads must have returned before their registers are reused
      }
#ifdef LIN_STAMP
      if (lane == 0 && g.C2 && EPI != EPI_GELU) {
        float* d = reinterpret_cast<float*>(g.C2) + ((int64_t)blockIdx.x * (NCW + NLD) + wv) * 8;
        d[0] = (float)(__builtin_readcyclecounter() - ck0); d[2] = (float)ck2; d[5] = (float)ntiles; d[6] = 2.f; d[7] = (float)(wall_clock64() - cr0);
      }
#endif
    }
  } else {
    // plain form, two stages: tile t+1 streams in while tile t is multiplied (two workgroups per CU cover each other's waits)
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < ntiles; ++t) {
      const int cur = t & 1;
      sbase = lds_u32(smem) + cur * STAGE_BYTES;
      addr(K0{}); LIN_RD4(0);
      if (t + 1 < ntiles) issue(t + 1, cur ^ 1);
      addr(K1{}); LIN_RD4(1); frag_wait<4>(fr[0]); LIN_STEP(0);
      addr(K2{}); LIN_RD4(0); frag_wait<4>(fr[1]); LIN_STEP(1);
      addr(K3{}); LIN_RD4(1); frag_wait<4>(fr[0]); LIN_STEP(0);
      frag_wait<0>(fr[1]); LIN_STEP(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue.  Staging: compute wave w owns smem + w * REGION; row r of its sub-tile at r * OPITCH (SPEC: all 64 rows, plain: 32).
  constexpr int REGION = (SPEC ? 64 : 32) * OPITCH;
  // (launch_lin sizes the dynamic LDS as max(ring, NCW * REGION): the two-per-CU form's 64-KB ring is smaller than its staging)
  auto stage_out = [&](int tm, int rslot) {   // lane: row rslot*32 + l31 of the wave's region (plain: l31), columns tn*32 + 16 lh + [0, 16)
    char* dst = smem + wv * REGION + ((SPEC ? rslot * 32 : 0) + l31) * OPITCH + lh * 64;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f4_t*>(dst + tn * 128 + q * 16) = f4_t{acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]};
  };
  // rows [32 half, +32) of sub-tile `sub`'s staging region = rows [32 tmrow, +32) of its RW-row wave tile; j0 = index of its rpre[0]
  auto finish_half = [&](int sub, int half, int j0, auto USE_RPRE, int tmrow) {
    const char* src = smem + sub * REGION + ((SPEC ? half * 32 : 0) + rrow) * OPITCH + ccol * 4;
    const int mrow = m0 + (sub / SUBN) * RW + tmrow * 32 + rrow;
#pragma unroll
    for (int j = 0; j < NIT_HALF; ++j) {
      const int m = mrow + RSTEP * j;
      if (m >= g.M) continue;            // ragged last tile
      if constexpr (OBF) {
        const f4_t v0 = *reinterpret_cast<const f4_t*>(src + j * RSTEP * OPITCH);
        const f4_t v1 = *reinterpret_cast<const f4_t*>(src + j * RSTEP * OPITCH + 16);
        f32x2_t p[4];
        if constexpr (LNIN) {   // rstd (acc - mean colsum) + bias
          const float2 st = *reinterpret_cast<const float2*>(smem + LNST_OFF + (m - m0) * 8);
          const float rs = st.y, t = -st.x * st.y;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p[e][0] = __builtin_fmaf(e < 2 ? v0[2 * e] : v1[2 * e - 4], rs, __builtin_fmaf(t, ccs[2 * e], bcol[2 * e]));
            p[e][1] = __builtin_fmaf(e < 2 ? v0[2 * e + 1] : v1[2 * e - 3], rs, __builtin_fmaf(t, ccs[2 * e + 1], bcol[2 * e + 1]));
          }
        } else {
          p[0] = f32x2_t{v0[0] + bcol[0], v0[1] + bcol[1]}; p[1] = f32x2_t{v0[2] + bcol[2], v0[3] + bcol[3]};
          p[2] = f32x2_t{v1[0] + bcol[4], v1[1] + bcol[5]}; p[3] = f32x2_t{v1[2] + bcol[6], v1[3] + bcol[7]};
        }
        const int64_t o = ((int64_t)m * g.ldc + sn0 + ccol) * 2;
        if constexpr (EPI == EPI_GBWD) {   // autograd of GELU fused into the fc2 input gradient: the same derivative as countr_gelu_bwd, applied to the fp32 sum
          const u32x4_t h = *reinterpret_cast<const u32x4_t*>(g.C2 + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float hlo, hhi;
            unpack2h(h[e], hlo, hhi);
            p[e][0] *= gelu_fast_grad(hlo);
            p[e][1] *= gelu_fast_grad(hhi);
          }
        }
        if constexpr (EPI == EPI_GELU) {
          if (g.C2) *reinterpret_cast<u32x4_t*>(g.C2 + o) = u32x4_t{pack2bf(p[0][0], p[0][1]), pack2bf(p[1][0], p[1][1]), pack2bf(p[2][0], p[2][1]), pack2bf(p[3][0], p[3][1])};
#pragma unroll
          for (int e = 0; e < 4; ++e) p[e] = gelu_sig2(p[e]);
        }
        const u32x4_t packed = u32x4_t{pack2bf(p[0][0], p[0][1]), pack2bf(p[1][0], p[1][1]), pack2bf(p[2][0], p[2][1]), pack2bf(p[3][0], p[3][1])};
        *reinterpret_cast<u32x4_t*>(g.C + o) = packed;
        if constexpr (CONV && EPI == EPI_BF16) countr_gn_row_partials(packed, g.gn_rows, m, g.N, sn0 + ccol, lane);
      } else {
        f4_t v = *reinterpret_cast<const f4_t*>(src + j * RSTEP * OPITCH);
        v += f4_t{bcol[0], bcol[1], bcol[2], bcol[3]};
        if constexpr (decltype(USE_RPRE)::value) v += rpre[j0 + j];
        else v += *reinterpret_cast<const f4_t*>(g.resid + (int64_t)(g.res_mod > 0 ? m % g.res_mod : m) * g.ldres + sn0 + ccol);
        *reinterpret_cast<f4_t*>(g.C + ((int64_t)m * g.ldc + sn0 + ccol) * 4) = v;
        if constexpr (LNOUT) {   // bf16 copy for the next GEMM's A operand + this 64-column block's row partials (16 lanes = one row)
          *reinterpret_cast<uint2*>(g.xcopy + ((int64_t)m * g.ldc + sn0 + ccol) * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
          const float s1 = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
          const float s2 = row16_sum(countr_sq4(v[0], v[1], v[2], v[3]));
          if ((lane & 15) == 0) *reinterpret_cast<float2*>(g.stats_out + ((int64_t)m * (g.N >> 6) + (sn0 >> 6)) * 2) = make_float2(s1, s2);
        }
      }
    }
  };
  if constexpr (SPEC) {
    if (!loader) { stage_out(0, 0); stage_out(1, 1); }
    __syncthreads();
    if (!loader) finish_half(sub0, 0, 0, std::bool_constant<RPRE>{}, 0);
    else if (cw < 4) {
      finish_half(sub0, 1, 0, std::bool_constant<RPRE>{}, 1);
#pragma unroll
      for (int u = 1; u < NUNITS; ++u) finish_half(sub0 + 4 * u, 1, 0, std::false_type{}, 1);
    }
    if constexpr (T3) {   // third 32-row tile of the 96-row wave tiles: a second pass through the same regions
      __syncthreads();
      if (!loader) stage_out(2, 0);
      __syncthreads();
      if (!loader) finish_half(sub0, 0, 0, std::false_type{}, 2);
    }
  } else {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      stage_out(tm, tm);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      finish_half(sub0, tm, 0, std::false_type{}, tm);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int WMB, int NLD, int EPI, int STAGES, bool CONV = false, int WNB = 1, bool LN = false, int TM = 2>
int launch_lin(const LinArgs& a0, hipStream_t s) {
  LinArgs a = a0;
  constexpr int BMt = 64 * TM * WMB;
  constexpr int ring = STAGES * (BMt + 128 * WNB) * 128, staging = NLD ? 4 * WMB * WNB * 64 * OPITCH : 0;   // the epilogue's staging regions reuse the ring
  constexpr int lds = (ring > staging ? ring : staging) + ((LN && EPI != EPI_RES) ? BMt * 8 : 0);   // + the consumer's row statistics
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_kernel<WMB, NLD, EPI, STAGES, CONV, WNB, LN, TM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  if (countr_dry_run) return 0;     // (a selection query: countr_gemm_gn_rows)
  a.launch_tiles = ((a.M + BMt - 1) / BMt - a.tile_m0) * a.tilesN;
  a.npf = countr_prefetch_blocks(a.launch_tiles, a.pf, a.pf_bytes);
  hipLaunchKernelGGL((lin_kernel<WMB, NLD, EPI, STAGES, CONV, WNB, LN, TM>), dim3(a.launch_tiles + a.npf), dim3(256 * WMB * WNB + 64 * NLD), lds, s, a);
  COUNTR_LAUNCH_CHECK("countr_gemm(lean linear)");
}

}  // namespace

// Returns 1 when the launch does not qualify (the caller then uses gemm_kernel), otherwise the launch status (0 / < 0).
int countr_lean_linear(const countr_gemm_args* a, hipStream_t s) {
  { const char* e = getenv("COUNTR_LEAN"); if (e && atoi(e) == 0) return 1; }   // read per call: A/B inside one process (not on a replay path)
  if (a->partial || a->nbatch > 1 || a->alpha != 1.0f || a->rowsum_partial) return 1;
  if (a->M < 1 || (a->N % 128) || (a->K % 64) || a->K < 128) return 1;      // any M: rows beyond it stage zeros and are not stored
  if ((a->lda % 8) || (a->ldb % 8) || (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C | (uintptr_t)a->C2) & 15)) return 1;
  if ((int64_t)128 * a->lda * 2 + (int64_t)a->K * 2 >= (int64_t)0x7f000000ll || (int64_t)256 * a->ldb * 2 + (int64_t)a->K * 2 >= (int64_t)0x7f000000ll) return 1;   // per-tile descriptor offsets
  if (a->bias && ((uintptr_t)a->bias & 15)) return 1;
  const float* zero_bias = nullptr;     // no bias (input gradients): the epilogue adds the per-device vector of zeros (countr_init)
  if (!a->bias) {
    if (a->N > COUNTR_ZERO_VEC_FLOATS) return 1;
    if (!(zero_bias = countr_zero_vec(a->N))) return -1;
  }
  int epi;
  if (a->out_bf16) {
    if (a->resid || (a->ldc % 8)) return 1;
#ifdef LIN_STAMP
    if (a->act == COUNTR_ACT_NONE) epi = EPI_BF16;     // stamp builds: C2 carries the debug buffer
#else
    if (a->act == COUNTR_ACT_NONE && !a->C2) epi = EPI_BF16;
#endif
    else if (a->act == COUNTR_ACT_GELU) epi = EPI_GELU;
    else if (a->act == COUNTR_ACT_GELU_BWD && a->C2 && !a->ln_stats && !a->ln_colsum && !a->ln_xcopy && !a->ln_stats_out) epi = EPI_GBWD;
    else return 1;
  } else {
#ifndef LIN_STAMP
    if (a->C2) return 1;
#endif
    if (!a->resid || a->act != COUNTR_ACT_NONE || (a->ldc % 4) || (a->ldres % 4) || ((uintptr_t)a->resid & 15)) return 1;
    epi = EPI_RES;
  }
  const bool ln_out = a->ln_xcopy != nullptr || a->ln_stats_out != nullptr, ln_in = a->ln_stats != nullptr || a->ln_colsum != nullptr;
  if (ln_out && (epi != EPI_RES || !a->ln_xcopy || !a->ln_stats_out || ((uintptr_t)a->ln_xcopy & 15) || ((uintptr_t)a->ln_stats_out & 7))) return 1;
  // (K % 128 == 0: the consumer reads its row partials [K / 64][2] with 16-byte loads, aligned for every row only when K / 64 is even)
  if (ln_in && (epi == EPI_RES || !a->ln_stats || !a->ln_colsum || (((uintptr_t)a->ln_stats | (uintptr_t)a->ln_colsum) & 15) || a->ln_nblk != a->K / 64 || (a->K % 128))) return 1;
  LinArgs g;
  g.xcopy = (char*)a->ln_xcopy; g.stats_out = a->ln_stats_out; g.stats_in = a->ln_stats; g.colsum = a->ln_colsum; g.ln_eps = a->ln_eps;
  g.gn_rows = nullptr;
  g.A = (const char*)a->A; g.W = (const char*)a->B; g.C = (char*)a->C; g.C2 = (char*)a->C2; g.bias = a->bias ? a->bias : zero_bias; g.resid = a->resid;
  g.M = a->M; g.N = a->N; g.K = a->K; g.lda = (int)a->lda; g.ldw = (int)a->ldb; g.ldc = (int)a->ldc; g.ldres = (int)a->ldres;
  g.res_mod = a->res_mod; g.tilesN = a->N / 128; g.tile_m0 = 0; g.H = g.Wd = g.Cin = 0;
  g.pf = (const char*)a->prefetch; g.pf_bytes = a->prefetch_bytes; g.launch_tiles = 0; g.npf = 0;
  const long tiles = (long)((a->M + 127) / 128) * g.tilesN;
#define LIN_LAUNCH(WMB, NLD, ST)                                                                    \
  {                                                                                                 \
    if (ln_in || ln_out) {                                                                          \
      if (epi == EPI_BF16) return launch_lin<WMB, NLD, EPI_BF16, ST, false, 1, true>(g, s);         \
      if (epi == EPI_GELU) return launch_lin<WMB, NLD, EPI_GELU, ST, false, 1, true>(g, s);         \
      return launch_lin<WMB, NLD, EPI_RES, ST, false, 1, true>(g, s);                               \
    }                                                                                               \
    if (epi == EPI_BF16) return launch_lin<WMB, NLD, EPI_BF16, ST>(g, s);                           \
    if (epi == EPI_GELU) return launch_lin<WMB, NLD, EPI_GELU, ST>(g, s);                           \
    if (epi == EPI_GBWD) return launch_lin<WMB, NLD, EPI_GBWD, ST>(g, s);                           \
    return launch_lin<WMB, NLD, EPI_RES, ST>(g, s);                                                 \
  }
  // one workgroup per CU at most: 4 compute + 4 loader waves on a 3-stage ring.  (Measured and dropped: 4- and 5-stage rings and 8 loader
  // waves -- the loaders are not waiting for data, their LDS-DMA instructions ISSUE at ~95 cycles each: 32 KB per k-tile at ~42 B/clk per
  // CU is what bounds this form, profiles/r3_linear_stamps.txt)
  if (tiles <= 256) LIN_LAUNCH(1, 4, 3)
  // 192 x 256 tiles (96x64 wave tiles, TM = 3) where they need fewer tile-times than the 256x128 grid: qkv at B = 8 (M = 4608, N = 2304) is
  // 324 workgroups of 256x128 = two rounds of 2 units, or 216 of 192x256 = one round of 3
  {
    const long g256 = (long)((a->M + 255) / 256) * (a->N / 128), g192 = (long)((a->M + 191) / 192) * (a->N / 256);
    // rounds x tile work (in 128x128 units: 2 vs 3) of the two grids on 256 CUs
    const long span256 = ((g256 + 255) / 256) * 2, span192 = ((g192 + 255) / 256) * 3;
    // ... and on big grids (>= 4 rounds) also when it needs up to 10 % more tile-times: its unit is cheaper (86 % of the staged bytes
    // per MFMA) -- zero-shot inference at 32 windows: 8.42-8.46 -> 8.27-8.37 ms with qkv AND fc1 on this form
    const long slack = g256 >= 1024 ? 110 : 100;
    if (epi != EPI_RES && (a->N % 256) == 0 && g256 > 256 && span192 * 100 <= span256 * slack) {
      g.tilesN = a->N / 256;
      if (ln_in) {
        if (epi == EPI_BF16) return launch_lin<1, 4, EPI_BF16, 2, false, 2, true, 3>(g, s);
        return launch_lin<1, 4, EPI_GELU, 2, false, 2, true, 3>(g, s);
      }
      if (epi == EPI_BF16) return launch_lin<1, 4, EPI_BF16, 2, false, 2, false, 3>(g, s);
      if (epi == EPI_GBWD) return launch_lin<1, 4, EPI_GBWD, 2, false, 2, false, 3>(g, s);
      return launch_lin<1, 4, EPI_GELU, 2, false, 2, false, 3>(g, s);
    }
  }
  // bigger grids: 256x128 tiles, 8 compute + 4 loader waves, 144-KB ring -- 2/3 of the staged bytes per MFMA.  (Measured and removed:
  // 128x128 wave-specialised on a 2-stage ring with two workgroups per CU, and the plain form where every wave stages and multiplies
  // -- finetune step 5.01 / 5.03 ms against 4.99 on one box, profiles/r3_step_ab.txt.)
  LIN_LAUNCH(2, 4, 3)
#undef LIN_LAUNCH
}

// 3x3 convolution forward / dgrad as implicit GEMM (A = IM2ROW view of an NHWC bf16 map, B = [Cout][9 Cin] weights): 256 x 128 tiles,
// 8 compute + 4 loader waves.  Returns 1 when the launch does not qualify (gemm_kernel then runs it).
// row0 > 0 (a multiple of 128, N % 256 == 0): only the rows [row0, M) -- the tail of a launch whose first rows ran on gemm256.hip's tiles.
int countr_lean_conv_rows(const countr_gemm_args* a, hipStream_t s, int row0);
int countr_lean_conv(const countr_gemm_args* a, hipStream_t s) { return countr_lean_conv_rows(a, s, 0); }

int countr_lean_conv_rows(const countr_gemm_args* a, hipStream_t s, int row0) {
  { const char* e = getenv("COUNTR_LEAN"); if (e && atoi(e) == 0) return 1; }
  { const char* e = getenv("COUNTR_LEAN_CONV"); if (e && atoi(e) == 0) return 1; }
  if (row0 < 0 || (row0 % 128) || row0 >= a->M || (row0 > 0 && (a->N % 256))) return 1;
#ifndef LIN_STAMP
  if (a->C2) return 1;
#endif
  if (a->partial || a->nbatch > 1 || a->alpha != 1.0f || a->rowsum_partial || a->resid || a->act != COUNTR_ACT_NONE || !a->out_bf16) return 1;
  if (a->M < 1 || (a->N % 128) || (a->Cin % 64) || a->Cin > 512 || a->K != 9 * a->Cin || a->H < 2 || a->W < 2) return 1;
  if ((a->ldb % 8) || (a->ldc % 8) || (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15)) return 1;
  if ((int64_t)(a->M + 2 * a->W + 2) * a->Cin * 2 >= (int64_t)0x7f000000ll || (int64_t)a->N * a->ldb * 2 >= (int64_t)0x7f000000ll || a->N > 4096) return 1;
  if (a->bias && ((uintptr_t)a->bias & 15)) return 1;
  const long tiles = (long)((a->M + 127) / 128) * (a->N / 128);
  if (tiles <= 256 && row0 == 0) return 1;     // small maps: the generic kernel's split-K / wave-specialised 128x128 forms
  if (a->gn_rows && ((uintptr_t)a->gn_rows & 7)) return 1;
  const float* zero_bias = nullptr;
  if (!a->bias && !(zero_bias = countr_zero_vec(a->N))) return -1;
  LinArgs g;
  g.A = (const char*)a->A; g.W = (const char*)a->B; g.C = (char*)a->C; g.C2 = nullptr; g.bias = a->bias ? a->bias : zero_bias; g.resid = nullptr;
#ifdef LIN_STAMP
  g.C2 = (char*)a->C2;   // stamp builds: the debug buffer
#endif
  g.M = a->M; g.N = a->N; g.K = a->K; g.lda = 0; g.ldw = (int)a->ldb; g.ldc = (int)a->ldc; g.ldres = 0;
  g.res_mod = 0; g.tilesN = a->N / 128; g.tile_m0 = row0 / 128; g.H = a->H; g.Wd = a->W; g.Cin = a->Cin;
  g.pf = nullptr; g.pf_bytes = 0; g.launch_tiles = 0; g.npf = 0;
  g.xcopy = nullptr; g.stats_out = nullptr; g.stats_in = nullptr; g.colsum = nullptr; g.ln_eps = 0.f;
  g.gn_rows = a->gn_rows;
  // 128 x 256 tiles when the width allows (the density head's 256 output channels in ONE workgroup: the im2row operand -- nine taps of
  // a map that does not fit the L2 -- is then staged once per row block, not once per column tile: 192x192 366 vs 402 us), else 256 x 128
  // (the 192 x 256 form on the convolutions: faster back to back -- 349 -> 331 us -- and not in the step, where the im2row operand
  // comes from HBM: measured and removed)
  if ((a->N % 256) == 0) { g.tilesN = a->N / 256; return launch_lin<1, 4, EPI_BF16, 3, true, 2>(g, s); }
  return launch_lin<2, 4, EPI_BF16, 3, true>(g, s);
}
