// Lean weight gradient of a 3x3 convolution (NHWC, pad 1) for gfx950: dW[co][tap][ci] = sum_p dy[p][co] * x[p + tap][ci], the
// (COL, IM2COL) split-K launches of countr_gemm() whose shapes qualify (bf16, Cout % 128 == 0, Cin % 128 == 0, pixels % 64 == 0).
// Autograd's convolution_backward weight/bias outputs for models_mae_cross.py:85-100 (density head) and :47-67 (exemplar CNN).
//
// Both operands are K-major here -- the contraction index is the PIXEL, the slow dimension of both NHWC maps -- so a k-tile is 64
// pixels x 128 channels of each map:
//   * staged by MUBUF LDS-DMA (buffer_load_dwordx4 ... lds) exactly as it lies in memory, 256-byte pixel rows, 16-byte chunk c of pixel
//     row k stored at slot c ^ ((k & 3) << 2); the padding taps of the im2col operand are lanes whose offset is pushed outside the
//     descriptor (the DMA writes zeros), decided by ONE comparison pair per (lane, pass) from incrementally updated pixel coordinates
//     (the generic kernel's im2col loader spends ~16 VALU instructions and a 64-bit address per 1-KiB piece: 1303 issue cycles per
//     k-tile against 512 matrix cycles, profiles/r2_gemm_kernel_timeline.txt);
//   * read back with ds_read_b64_tr_b16: a 16-lane group transposes a [4 pixels][16 channels] block, two reads = one 32x32x16 MFMA
//     operand (8 consecutive pixels of one channel per lane).  With the XOR above the 32 lanes of a half-wave touch 32 distinct 8-byte
//     bank pairs: conflict-free.
// Workgroup = 4 WNB compute waves (64x64 sub-tiles, 2x2 MFMA tiles) + 4 loader waves, 3-stage ring, tile 128 (Cout) x 128 WNB (Cin
// of one tap); grid = tiles x splitk, an XCD owning one or two k-ranges.  Output: raw fp32 partial[z][Cout][9 Cin] (countr_reduce_table
// / countr_splitk_reduce finish and permute to OIHW).
// Fused bias gradient (rowsum_partial): sum_p dy[p][co] = the same dy fragments times a matrix of ones.  One extra MFMA per k-step
// would cost a wave 25 %, so the k-tiles of a (z, 32-row block) are dealt round-robin to the NC = tilesN * WNB waves that hold that
// block's fragments anyway (+1.4 % each); every one writes its own slab: rowsum_partial[z * NC + i][Cout]
// (countr_gemm_rowsum_slabs() tells the caller how many slabs a launch writes).
// Round 5, the big maps (cwg3_body below): where the rows of the map are whole k-tiles (W % 64 == 0, or W % 96 == 0 with 96-pixel k-tiles)
// a workgroup computes the THREE taps of one kernel row from ONE staged row segment of the input map -- a third fewer staged bytes and
// a sixth fewer fragment reads per MFMA; the 192 x 192 and 96 x 96 layers of the density head.
#include "common.hpp"
#include "../../include/countr_hip.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

#ifndef CWG_ABL
#define CWG_ABL 0   // timing experiments (tools/exp_cwg.sh, results are WRONG): 1 = no MFMA, 2 = no fragment reads, 3 = no DMA after the first tiles
#endif

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) void* lds_vptr_t;
typedef __attribute__((address_space(3))) const char* lds_cptr_t;

struct CwgArgs {
  const char* dy;      // [P, lda] bf16 (NHWC output gradient; LIN: dy [rows, lda])
  const char* x;       // [P, ldb] bf16 (NHWC input map; LIN: x [rows, ldb])
  float* part;         // [Z][Cout][N]
  float* rowsum;       // [Z * NC][Cout] or null
  int Cout, Cin, H, Wd;   // Cout = rows of the output (M of the GEMM); Cin / H / Wd: convolution only
  int lda, ldb;        // row pitches of dy and x in elements (convolution: Cout, Cin)
  int N;               // 9 Cin (LIN: in_features)
  int tiles, tilesN;   // tiles = (Cout / 128) * tilesN
  int nkt, per, Z;     // k-tiles (P / 64), k-tiles per z, slabs
};

__device__ __forceinline__ uint32_t lds_u32(const char* p) { return (uint32_t)(uintptr_t)(lds_cptr_t)p; }

template <int OFF> __device__ __forceinline__ s16x4_t ds_tr(uint32_t a) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}

struct Frag { s16x4_t lo, hi; };   // k = 8 lh + [0, 4), + [4, 8)
__device__ __forceinline__ bf16x8_t frag_bits(const Frag& f) {
  return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// wait until at most PENDING LDS instructions are outstanding; the set's fragments are threaded through so their MFMAs stay behind it
template <int PENDING> __device__ __forceinline__ void frag_wait(Frag (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi)
               : "n"(PENDING));
}

// LIN: the same kernel as the weight gradient of an nn.Linear, dW[n][k] = sum_r dy[r][n] x[r][k] (the (COL, COL) split-K launches): the
// contraction index is the token row, again the slow dimension of both operands -- no taps, no padding, nothing else differs.
// joint (z, tile) order of a launch, one contiguous range per XCD (workgroup b runs on XCD b % 8)
__device__ __forceinline__ int cwg_virtual_index() {
  const int nt = gridDim.x, q = nt >> 3, r = nt & 7, xc = blockIdx.x & 7, j = blockIdx.x >> 3;
  return (xc < r ? xc * (q + 1) : r * (q + 1) + (xc - r) * q) + j;
}

// v: this workgroup's index among the g.tiles * g.Z (tile, slab) pairs of the problem g
template <int WNB, bool LIN>
__device__ __forceinline__ void cwg_body(const CwgArgs& g, const int v) {
  constexpr int STAGES = 3, NCW = 4 * WNB, NLW = 4, SUBN = 2 * WNB;
  constexpr int A_BYTES = 64 * 256, STAGE_BYTES = A_BYTES * (1 + WNB);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wv >= NCW;
  const int cw = loader ? wv - NCW : wv;
  const int z = v / g.tiles, lt = v - z * g.tiles;
  const int tile_m = lt / g.tilesN, tile_n = lt - tile_m * g.tilesN;
  const int m0 = tile_m * 128, n0 = tile_n * (128 * WNB);
  const int tap = LIN ? 4 : n0 / g.Cin, ci0 = LIN ? n0 : n0 - tap * g.Cin;
  const int dyt = tap / 3 - 1, dxt = tap - (tap / 3) * 3 - 1;
  const int kt0 = min(z * g.per, g.nkt), kt1 = min(kt0 + g.per, g.nkt);
  const int ntiles = kt1 - kt0;

  // ---- staging (loader waves).  Piece (pass i, wave w) = pixel rows [16 i + 4 w, +4) of a 128-channel operand tile; lane -> (row, slot)
  const int krow = lane >> 4, c8 = (lane & 15) ^ (krow << 2);
  const uint32_t voffA = (uint32_t)(((cw * 4 + krow) * g.lda + c8 * 8) * 2);
  const uint32_t voffB = (uint32_t)(((cw * 4 + krow) * g.ldb + c8 * 8) * 2);
  const int64_t cshift = LIN ? 0 : (int64_t)(g.Wd + 1) * g.ldb * 2;     // descriptor base (Wd + 1) pixels in front of the map: every tap shift >= 0
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(g.dy + ((int64_t)kt0 * 64 * g.lda + m0) * 2), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(g.x + ((int64_t)kt0 * 64 * g.ldb + ci0) * 2 - cshift), 0, 0x7ffffff0, 0x00020000);
  const uint32_t tapoff = LIN ? 0u : (uint32_t)(cshift + (int64_t)(dyt * g.Wd + dxt) * g.ldb * 2);
  const uint32_t passA = (uint32_t)g.lda * 32u, passB = (uint32_t)g.ldb * 32u;       // 16 pixel rows further, bytes
  const uint32_t tileA = (uint32_t)g.lda * 128u, tileB = (uint32_t)g.ldb * 128u;     // 64 pixel rows further
  int px[LIN ? 1 : 4], py[LIN ? 1 : 4];   // pixel coordinates of this lane's row in pass i of the NEXT tile to be issued
  if (!LIN && loader) {
    const int p = kt0 * 64 + cw * 4 + krow;
    int xx = p % g.Wd, yy = (p / g.Wd) % g.H;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      px[i] = xx; py[i] = yy;
      xx += 16;
      while (xx >= g.Wd) { xx -= g.Wd; yy = (yy + 1 == g.H) ? 0 : yy + 1; }
    }
  }
  auto issue = [&](int t, int slot_) {   // called with t = 0, 1, 2, ... in order (the coordinates advance by one tile per call)
    char* dst = smem + slot_ * STAGE_BYTES + cw * 1024;
#if CWG_ABL == 3
    if (t >= 2) return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_vptr_t)(dst + i * 4096), 16, voffA, t * tileA + i * passA, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t vo = voffB;
      if constexpr (!LIN) {
        const bool ok = (unsigned)(py[i] + dyt) < (unsigned)g.H && (unsigned)(px[i] + dxt) < (unsigned)g.Wd;
        vo = ok ? voffB : 0x80000000u;   // (a named variable: hipcc 7.2 drops the kernel stub for a conditional written as the argument)
      }
#pragma unroll
      for (int j = 0; j < WNB; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_vptr_t)(dst + A_BYTES + j * A_BYTES + i * 4096), 16, vo,
                                                 tapoff + t * tileB + i * passB + j * 256, 0, 0);
      if constexpr (!LIN) {
        px[i] += 64;
        while (px[i] >= g.Wd) { px[i] -= g.Wd; py[i] = (py[i] + 1 == g.H) ? 0 : py[i] + 1; }
      }
    }
  };
  if (loader) {
#pragma unroll
    for (int s_ = 0; s_ < STAGES - 1; ++s_)
      if (s_ < ntiles) issue(s_, s_);
  }

  // ---- fragment addresses (compute waves)
  const int wm = cw / SUBN, wn = cw % SUBN;
  const int l15 = lane & 15, q16 = (lane >> 4) & 1, lh = lane >> 5, rr = l15 >> 2, bb = l15 & 3;
  uint32_t offA[2], offB[2];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const int ca = wm * 8 + t2 * 4 + q16 * 2 + (bb >> 1), cb = (wn & 1) * 8 + t2 * 4 + q16 * 2 + (bb >> 1);
    offA[t2] = (uint32_t)((lh * 8 + rr) * 256 + ((ca ^ (rr << 2)) << 4) + (bb & 1) * 8);
    offB[t2] = (uint32_t)(A_BYTES + (wn >> 1) * A_BYTES + (lh * 8 + rr) * 256 + ((cb ^ (rr << 2)) << 4) + (bb & 1) * 8);
  }
  f32x16_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  // bias-gradient turn: contributor i = (tile_n * SUBN + wn) >> 1 of the NC that hold rows [32 tmr, +32) of this 64-row block
  const int rid = tile_n * SUBN + wn, tmr = rid & 1, ridx = rid >> 1, NC = (g.tilesN * SUBN) >> 1;
  const bool want_rs = g.rowsum != nullptr;
  f32x16_t accb;
#pragma unroll
  for (int e = 0; e < 16; ++e) accb[e] = 0.f;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(COUNTR_H16_ONE_PAIR, COUNTR_H16_ONE_PAIR, COUNTR_H16_ONE_PAIR, COUNTR_H16_ONE_PAIR));

  Frag fr[4][4];   // [set = k-step][a0, a1, b0, b1]
#if CWG_ABL == 2
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { fr[a][b].lo = s16x4_t{(short)lane, 1, 2, 3}; fr[a][b].hi = s16x4_t{3, 2, 1, (short)lane}; }
#endif
  uint32_t a0, a1, b0, b1;
  auto addr = [&](uint32_t sbase) { a0 = sbase + offA[0]; a1 = sbase + offA[1]; b0 = sbase + offB[0]; b1 = sbase + offB[1]; };
#if CWG_ABL == 2
#define CWG_RD(SET, Q, KK, A) { fr[SET][Q].lo[0] += (short)(A); }
#else
#define CWG_RD(SET, Q, KK, A) { fr[SET][Q].lo = ds_tr<(KK) * 4096>(A); fr[SET][Q].hi = ds_tr<(KK) * 4096 + 1024>(A); }
#endif
#if CWG_ABL == 1
#define CWG_MM(SET, TM, TN) acc[TM][TN][(SET) * 4 + (TM) * 2 + (TN)] += (float)fr[SET][TM].lo[0] * (float)fr[SET][2 + TN].hi[1]
#else
#define CWG_MM(SET, TM, TN) acc[TM][TN] = COUNTR_MFMA_32X32X16(frag_bits(fr[SET][TM]), frag_bits(fr[SET][2 + TN]), acc[TM][TN], 0, 0, 0)
#endif
#ifdef CWG_NOSB
#define CWG_SB
#else
#define CWG_SB __builtin_amdgcn_sched_barrier(0)
#endif
#define CWG_BIAS(SET) if (mine) { accb = COUNTR_MFMA_32X32X16(frag_bits(tmr ? fr[SET][1] : fr[SET][0]), ones, accb, 0, 0, 0); CWG_SB; }
  // MFMAs of set U with the reads of set R = k-step KK of the stage at a0 / a1 / b0 / b1 between them
#define CWG_STEP_RD(U, R, KK) CWG_MM(U, 0, 0); CWG_SB; CWG_RD(R, 0, KK, a0); CWG_SB; CWG_MM(U, 0, 1); CWG_SB; CWG_RD(R, 2, KK, b0); CWG_SB; \
                              CWG_MM(U, 1, 0); CWG_SB; CWG_RD(R, 1, KK, a1); CWG_SB; CWG_MM(U, 1, 1); CWG_SB; CWG_RD(R, 3, KK, b1); CWG_SB; CWG_BIAS(U)
#define CWG_STEP(U) CWG_MM(U, 0, 0); CWG_MM(U, 0, 1); CWG_MM(U, 1, 0); CWG_MM(U, 1, 1); CWG_SB; CWG_BIAS(U)
#define CWG_RD4(SET, KK) CWG_RD(SET, 0, KK, a0); CWG_RD(SET, 1, KK, a1); CWG_RD(SET, 2, KK, b0); CWG_RD(SET, 3, KK, b1)

  if (loader) {
    // keep STAGES-1 tiles in flight; tile t has landed when at most (STAGES-2) tiles' worth of loads are outstanding
    constexpr int PER = 4 + 4 * WNB;
    int islot = STAGES - 1;
    for (int t = 0; t < ntiles; ++t) {
      if (t + STAGES - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + STAGES - 1 < ntiles) issue(t + STAGES - 1, islot);
      islot = (islot + 1 == STAGES) ? 0 : islot + 1;
    }
    __builtin_amdgcn_s_barrier();   // the compute waves' last in-loop barrier
    return;
  }

  // compute waves: one fragment set per k-step; the reads of k-step s+2 go between the MFMAs of k-step s; at the tile boundary k-step 2
  // runs bare, every read of the tile has landed, barrier (the loaders may refill the slot, the next tile is visible), and k-step 3's
  // MFMAs carry the reads of the next tile's k-steps 0 and 1.  Same body for every tile: the last one's look-ahead reads fetch stale
  // ring bytes nobody uses (linear.hip has the history of this loop).
  int slot = 0, tmod = 0;
  __builtin_amdgcn_s_barrier();
  addr(lds_u32(smem));
  CWG_RD4(0, 0);
  CWG_RD4(1, 1);
  CWG_SB;
  for (int t = 0; t < ntiles; ++t) {
    const bool mine = want_rs && tmod == ridx;
    tmod = (tmod + 1 == NC) ? 0 : tmod + 1;
    frag_wait<8>(fr[0]); CWG_STEP_RD(0, 2, 2);
    frag_wait<8>(fr[1]); CWG_STEP_RD(1, 3, 3);
    frag_wait<8>(fr[2]); CWG_STEP(2);
    frag_wait<0>(fr[3]);
    slot = (slot + 1 == STAGES) ? 0 : slot + 1;
    __builtin_amdgcn_s_barrier();
    addr(lds_u32(smem) + slot * STAGE_BYTES);
    CWG_MM(3, 0, 0); CWG_SB; CWG_RD(0, 0, 0, a0); CWG_RD(0, 2, 0, b0); CWG_SB;
    CWG_MM(3, 0, 1); CWG_SB; CWG_RD(0, 1, 0, a1); CWG_RD(0, 3, 0, b1); CWG_SB;
    CWG_MM(3, 1, 0); CWG_SB; CWG_RD(1, 0, 1, a0); CWG_RD(1, 2, 1, b0); CWG_SB;
    CWG_MM(3, 1, 1); CWG_SB; CWG_RD(1, 1, 1, a1); CWG_RD(1, 3, 1, b1); CWG_SB;
    CWG_BIAS(3)
  }
  frag_wait<0>(fr[0]); frag_wait<0>(fr[1]);   // the stale look-ahead reads must be back before their registers are reused

  // ---- raw fp32 partial sums: accumulator register e of lane (l31, lh) = row (e & 3) + 8 (e >> 2) + 4 lh, column l31 of its 32x32 tile
  const int l31 = lane & 31;
  float* out = g.part + (int64_t)z * g.Cout * g.N + (int64_t)(m0 + wm * 64 + 4 * lh) * g.N + n0 + wn * 64 + l31;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        out[(int64_t)(tm * 32 + (e & 3) + 8 * (e >> 2)) * g.N + tn * 32] = acc[tm][tn][e];
  if (want_rs && l31 == 0) {
    float* rs = g.rowsum + (int64_t)(z * NC + ridx) * g.Cout + m0 + wm * 64 + tmr * 32 + 4 * lh;
#pragma unroll
    for (int e = 0; e < 16; ++e) rs[(e & 3) + 8 * (e >> 2)] = accb[e];
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Form 3 (round 5): the THREE taps of one kernel row per workgroup.  For maps whose rows are whole k-tiles (W % 64 == 0: the 192 x 192
// layer, 43 of the head's 58 GF) the 64 pixels of a k-tile lie in one image row, so the im2col operands of the taps (dy, -1), (dy, 0),
// (dy, +1) are ONE row segment of the input map read at three pixel offsets.  The segment is staged once -- 72 pixel rows: the k-tile's
// 64 + a 4-row piece on either side for the +-1 neighbours (80 rows are issued so that every loader wave issues the same count; the
// last 8 are lanes outside the descriptor) -- in the same 256-byte-row / XOR-slot image as the other forms, and the three B fragments
// of a k-step are transposing reads at row offsets 3, 4, 5 with the slot XOR following the row ((rr + dx) & 3: the 16-lane group still
// touches four consecutive rows, so the reads stay conflict-free).  Tile = 128 (Cout) x 128 (Cin) x 3 taps: 36 KB staged per 192
// MFMAs (form 2: 48 KB per 128) and a dy fragment serves three MFMAs.  8 compute waves (64 x 32 x 3 taps each: 96 accumulator
// registers) + 4 loader waves, 3-stage ring of 36 KB.  Padding: image rows outside [0, H) and pixels outside [0, W) are lanes pushed
// outside the descriptor, as everywhere.  Bias gradient: 12 waves on a CU leave a wave 168 registers, which the 96 accumulators + two
// fragment sets fill, so the row sums of dy are taken by the LOADER waves (idle between their issue and the next barrier): the
// k-tiles of a (slab, tile_m) are dealt round-robin to its tilesN workgroups, whose loader wave w multiplies rows [32 w, +32) of the
// staged dy tile by a ones fragment (4 MFMAs in 1 of tilesN k-tiles), each workgroup writing its own slab.
// KS = k-steps (16 pixels) per k-tile: 4 (W % 64 == 0) or 6 (W % 96 == 0: the 96 x 96 layer, one image row per k-tile)
constexpr int C3_STAGES = 3;
constexpr int c3_stage_bytes(int ks) { return ks * 16 * 256 + (ks * 16 + 16) * 256; }
template <int PENDING> __device__ __forceinline__ void frag_wait1(Frag& f) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.lo), "+v"(f.hi) : "n"(PENDING)); }

struct Frag5 { Frag f[5]; };   // a0, a1 (two 32-row tiles of dy), b(-1), b(0), b(+1)
template <int PENDING> __device__ __forceinline__ void frag_wait5(Frag5& s) {
  asm volatile("s_waitcnt lgkmcnt(%10)"
               : "+v"(s.f[0].lo), "+v"(s.f[0].hi), "+v"(s.f[1].lo), "+v"(s.f[1].hi), "+v"(s.f[2].lo), "+v"(s.f[2].hi), "+v"(s.f[3].lo), "+v"(s.f[3].hi),
                 "+v"(s.f[4].lo), "+v"(s.f[4].hi)
               : "n"(PENDING));
}

template <int KS>
__device__ __forceinline__ void cwg3_body(const CwgArgs& g, const int v) {
  constexpr int KT = KS * 16, C3_A = KT * 256, C3_STAGE = c3_stage_bytes(KS);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wv >= 8;
  const int cw = loader ? wv - 8 : wv;
  const int z = v / g.tiles, lt = v - z * g.tiles;
  const int tile_m = lt / g.tilesN, tile_n = lt - tile_m * g.tilesN;
  const int nci = g.Cin >> 7;
  const int row3 = tile_n / nci, ci0 = (tile_n - row3 * nci) * 128;
  const int dyt = row3 - 1, m0 = tile_m * 128;
  const int kt0 = min(z * g.per, g.nkt), kt1 = min(kt0 + g.per, g.nkt);
  const int ntiles = kt1 - kt0;

  if (loader) {
    const int krow = lane >> 4, c8 = (lane & 15) ^ (krow << 2);
    const uint32_t voffA = (uint32_t)(((cw * 4 + krow) * g.lda + c8 * 8) * 2);
    const uint32_t voffB = (uint32_t)(((cw * 4 + krow) * g.ldb + c8 * 8) * 2);
    const int64_t cshift = (int64_t)(g.Wd + 8) * g.ldb * 2;          // descriptor base in front of the map: every shift (dy W - 4 pixels) >= 0
    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc((void*)(g.dy + ((int64_t)kt0 * KT * g.lda + m0) * 2), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc((void*)(g.x + ((int64_t)kt0 * KT * g.ldb + ci0) * 2 - cshift), 0, 0x7ffffff0, 0x00020000);
    const uint32_t tapoff = (uint32_t)(cshift + (int64_t)(dyt * g.Wd - 4) * g.ldb * 2);
    const uint32_t passA = (uint32_t)g.lda * 32u, passB = (uint32_t)g.ldb * 32u, tileA = (uint32_t)g.lda * (2u * KT), tileB = (uint32_t)g.ldb * (2u * KT);
    int x0 = (int)(((int64_t)kt0 * KT) % g.Wd), y = (int)((((int64_t)kt0 * KT) / g.Wd) % g.H);      // of the NEXT tile to be issued
    auto issue = [&](int t, int slot_) {      // called with t = 0, 1, 2, ... in order
      char* dst = smem + slot_ * C3_STAGE + cw * 1024;
#pragma unroll
      for (int i = 0; i < KS; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_vptr_t)(dst + i * 4096), 16, voffA, t * tileA + i * passA, 0, 0);
      const bool rowok = (unsigned)(y + dyt) < (unsigned)g.H;
#pragma unroll
      for (int i = 0; i < KS + 1; ++i) {
        const int r = 16 * i + 4 * cw + krow, xx = x0 - 4 + r;
        const bool ok = rowok && r < KT + 8 && (unsigned)xx < (unsigned)g.Wd;
        const uint32_t vo = ok ? voffB : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_vptr_t)(dst + C3_A + i * 4096), 16, vo, tapoff + t * tileB + i * passB, 0, 0);
      }
      x0 += KT;
      if (x0 >= g.Wd) { x0 = 0; y = (y + 1 == g.H) ? 0 : y + 1; }
    };
#pragma unroll
    for (int s_ = 0; s_ < C3_STAGES - 1; ++s_)
      if (s_ < ntiles) issue(s_, s_);
    constexpr int PER = 2 * KS + 1;
    const int l15 = lane & 15, q16 = (lane >> 4) & 1, lh = lane >> 5, rr = l15 >> 2, bb = l15 & 3;
    const uint32_t offR = (uint32_t)((lh * 8 + rr) * 256 + (((cw * 4 + q16 * 2 + (bb >> 1)) ^ (rr << 2)) << 4) + (bb & 1) * 8);
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(COUNTR_H16_ONE_PAIR, COUNTR_H16_ONE_PAIR, COUNTR_H16_ONE_PAIR, COUNTR_H16_ONE_PAIR));
    const bool want_rs = g.rowsum != nullptr;
    f32x16_t accb;
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[e] = 0.f;
    int islot = C3_STAGES - 1, slot = 0, tmod = 0;
    for (int t = 0; t < ntiles; ++t) {
      if (t + C3_STAGES - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((C3_STAGES - 2) * PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + C3_STAGES - 1 < ntiles) issue(t + C3_STAGES - 1, islot);
      islot = (islot + 1 == C3_STAGES) ? 0 : islot + 1;
      if (want_rs && tmod == tile_n) {       // (stage `slot` is complete behind the barrier and is overwritten only behind the next one)
        const uint32_t sb = lds_u32(smem) + slot * C3_STAGE + offR;
        Frag fb[KS];
        fb[0].lo = ds_tr<0>(sb); fb[0].hi = ds_tr<1024>(sb);
        fb[1].lo = ds_tr<4096>(sb); fb[1].hi = ds_tr<4096 + 1024>(sb);
        fb[2].lo = ds_tr<8192>(sb); fb[2].hi = ds_tr<8192 + 1024>(sb);
        fb[3].lo = ds_tr<12288>(sb); fb[3].hi = ds_tr<12288 + 1024>(sb);
        if constexpr (KS == 6) {
          fb[4].lo = ds_tr<16384>(sb); fb[4].hi = ds_tr<16384 + 1024>(sb);
          fb[5].lo = ds_tr<20480>(sb); fb[5].hi = ds_tr<20480 + 1024>(sb);
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) { frag_wait1<0>(fb[kk]); accb = COUNTR_MFMA_32X32X16(frag_bits(fb[kk]), ones, accb, 0, 0, 0); }
      }
      tmod = (tmod + 1 == g.tilesN) ? 0 : tmod + 1;
      slot = (slot + 1 == C3_STAGES) ? 0 : slot + 1;
    }
    __builtin_amdgcn_s_barrier();
    if (want_rs && (lane & 31) == 0) {
      float* rs = g.rowsum + (int64_t)(z * g.tilesN + tile_n) * g.Cout + m0 + cw * 32 + 4 * lh;
#pragma unroll
      for (int e = 0; e < 16; ++e) rs[(e & 3) + 8 * (e >> 2)] = accb[e];
    }
    return;
  }

  // ---- compute waves: wave (wm, wn) owns rows [64 wm, +64) of the tile's Cout and channels [32 wn, +32) of its Cin, for the three taps
  const int wm = cw >> 2, wn = cw & 3;
  const int l15 = lane & 15, q16 = (lane >> 4) & 1, lh = lane >> 5, rr = l15 >> 2, bb = l15 & 3;
  uint32_t offA[2], offB[3];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const int ca = wm * 8 + t2 * 4 + q16 * 2 + (bb >> 1);
    offA[t2] = (uint32_t)((lh * 8 + rr) * 256 + ((ca ^ (rr << 2)) << 4) + (bb & 1) * 8);
  }
  const int cb = wn * 4 + q16 * 2 + (bb >> 1);
#pragma unroll
  for (int d = 0; d < 3; ++d) {      // dx = d - 1: staged row 4 + dx + (k within the tile)
    const int row = 3 + d + lh * 8 + rr, sw = (rr + d + 3) & 3;
    offB[d] = (uint32_t)(C3_A + row * 256 + ((cb ^ (sw << 2)) << 4) + (bb & 1) * 8);
  }
  f32x16_t acc[3][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[d][tm][e] = 0.f;
  Frag5 fs[2];
#define C3_RD(SET, KK)                                                                                                    \
  { fs[SET].f[0].lo = ds_tr<(KK) * 4096>(sb + offA[0]); fs[SET].f[0].hi = ds_tr<(KK) * 4096 + 1024>(sb + offA[0]);         \
    fs[SET].f[1].lo = ds_tr<(KK) * 4096>(sb + offA[1]); fs[SET].f[1].hi = ds_tr<(KK) * 4096 + 1024>(sb + offA[1]);         \
    fs[SET].f[2].lo = ds_tr<(KK) * 4096>(sb + offB[0]); fs[SET].f[2].hi = ds_tr<(KK) * 4096 + 1024>(sb + offB[0]);         \
    fs[SET].f[3].lo = ds_tr<(KK) * 4096>(sb + offB[1]); fs[SET].f[3].hi = ds_tr<(KK) * 4096 + 1024>(sb + offB[1]);         \
    fs[SET].f[4].lo = ds_tr<(KK) * 4096>(sb + offB[2]); fs[SET].f[4].hi = ds_tr<(KK) * 4096 + 1024>(sb + offB[2]); }
#define C3_MM(SET)                                                                                                        \
  { _Pragma("unroll") for (int d = 0; d < 3; ++d) { _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                         \
      acc[d][tm] = COUNTR_MFMA_32X32X16(frag_bits(fs[SET].f[tm]), frag_bits(fs[SET].f[2 + d]), acc[d][tm], 0, 0, 0); } }
  int slot = 0;
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < ntiles; ++t) {
    const uint32_t sb = lds_u32(smem) + slot * C3_STAGE;
    C3_RD(0, 0);
    C3_RD(1, 1); frag_wait5<10>(fs[0]); C3_MM(0);
    C3_RD(0, 2); frag_wait5<10>(fs[1]); C3_MM(1);
    C3_RD(1, 3); frag_wait5<10>(fs[0]); C3_MM(0);
    if constexpr (KS == 6) {
      C3_RD(0, 4); frag_wait5<10>(fs[1]); C3_MM(1);
      C3_RD(1, 5); frag_wait5<10>(fs[0]); C3_MM(0);
    }
    frag_wait5<0>(fs[1]); C3_MM(1);
    slot = (slot + 1 == C3_STAGES) ? 0 : slot + 1;
    __builtin_amdgcn_s_barrier();
  }
#undef C3_RD
#undef C3_MM
  // ---- raw fp32 partial sums: accumulator register e of lane (l31, lh) = row (e & 3) + 8 (e >> 2) + 4 lh, column l31 of its 32x32 tile
  const int l31 = lane & 31;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float* out = g.part + (int64_t)z * g.Cout * g.N + (int64_t)(m0 + wm * 64 + 4 * lh) * g.N + (row3 * 3 + d) * g.Cin + ci0 + wn * 32 + l31;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int e = 0; e < 16; ++e) out[(int64_t)(tm * 32 + (e & 3) + 8 * (e >> 2)) * g.N] = acc[d][tm][e];
  }
}

template <int KS>
__global__ __launch_bounds__(768) void cwg3_kernel(const CwgArgs g) { cwg3_body<KS>(g, cwg_virtual_index()); }

template <int KS>
int launch_cwg3(const CwgArgs& a, hipStream_t s) {
  constexpr int lds = C3_STAGES * c3_stage_bytes(KS);     // 108 KB (KS = 4), 156 KB (KS = 6)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cwg3_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((cwg3_kernel<KS>), dim3(a.tiles * a.Z), dim3(768), lds, s, a);
  COUNTR_LAUNCH_CHECK("countr_gemm(conv wgrad, three taps per workgroup)");
}

template <int WNB, bool LIN>
__global__ __launch_bounds__(256 * WNB + 256, (4 * WNB + 4) / 4) void cwg_kernel(const CwgArgs g) {
  cwg_body<WNB, LIN>(g, cwg_virtual_index());
}

// Several nn.Linear weight gradients in ONE launch (countr_gemm_group): the workgroups of problem i are [start[i], start[i + 1]) of the
// XCD-ordered index.  Why: the four weight gradients of a transformer block are 16-72 tiles each -- alone each needs 3-16 split-K slabs
// to fill the chip (fp32 partials written and summed again); together they fill it with one or two.
constexpr int CWG_GROUP_MAX = 10;
struct CwgGroup {
  CwgArgs it[CWG_GROUP_MAX];
  int start[CWG_GROUP_MAX + 1];
};
template <int WNB>
__global__ __launch_bounds__(256 * WNB + 256, (4 * WNB + 4) / 4) void cwg_group_kernel(const CwgGroup grp) {
  const int v = cwg_virtual_index();
  int p = 0;
#pragma unroll
  for (int i = 1; i < CWG_GROUP_MAX; ++i)
    if (v >= grp.start[i]) p = i;
  cwg_body<WNB, true>(grp.it[p], v - grp.start[p]);
}

template <int WNB>
int launch_cwg_group(const CwgGroup& g, hipStream_t s) {
  constexpr int lds = 3 * 64 * 256 * (1 + WNB);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cwg_group_kernel<WNB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((cwg_group_kernel<WNB>), dim3(g.start[CWG_GROUP_MAX]), dim3(256 * WNB + 256), lds, s, g);
  COUNTR_LAUNCH_CHECK("countr_gemm_group(lean linear wgrad)");
}

template <int WNB, bool LIN>
int launch_cwg(const CwgArgs& a, hipStream_t s) {
  constexpr int lds = 3 * 64 * 256 * (1 + WNB);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cwg_kernel<WNB, LIN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((cwg_kernel<WNB, LIN>), dim3(a.tiles * a.Z), dim3(256 * WNB + 256), lds, s, a);
  COUNTR_LAUNCH_CHECK("countr_gemm(lean conv wgrad)");
}

// k-tile of form 3 in pixels: a divisor of the map's width (64 where it divides, else 96)
static int cwg3_ktile(int W) { return (W % 64) == 0 ? 64 : 96; }

// Contributors that share the bias-gradient of one (slab, row block) = rowsum slabs per split-K slab
static int cwg_contributors(const countr_gemm_args* a, int form) { return form == 3 ? 3 * (a->Cin / 128) : a->N / 128; }

// form: 0 = does not qualify, 1 = 128x128 tiles, 2 = 128x256 tiles, 3 = 128 x 128 x the three taps of a kernel row (conv only).
// lin: the (COL, COL) launch of an nn.Linear weight gradient
int cwg_form(const countr_gemm_args* a, bool lin) {
  { const char* e = getenv("COUNTR_LEAN"); if (e && atoi(e) == 0) return 0; }
  { const char* e = getenv(lin ? "COUNTR_LEAN_LWGRAD" : "COUNTR_LEAN_WGRAD"); if (e && atoi(e) == 0) return 0; }
  if (!a->partial || a->nbatch > 1 || a->alpha != 1.0f || a->bias || a->resid || a->C2 || a->act != COUNTR_ACT_NONE) return 0;
  if (a->ln_xcopy || a->ln_stats_out || a->ln_stats || a->ln_colsum) return 0;
  if ((a->M % 128) || (a->N % 128) || (a->K % 64) || a->K < 64 || a->ldc != a->N) return 0;
  if (lin) {
    if ((a->lda % 8) || (a->ldb % 8) || a->lda < a->M || a->ldb < a->N) return 0;
    if ((int64_t)a->K * a->ldb * 2 >= (int64_t)0x7f000000ll || (int64_t)a->K * a->lda * 2 >= (int64_t)0x7f000000ll) return 0;
  } else {
    if ((a->Cin % 128) || a->N != 9 * a->Cin || a->H < 2 || a->W < 2 || a->lda != a->M) return 0;
    if ((int64_t)(a->K + 2 * a->W + 2) * a->Cin * 2 >= (int64_t)0x7f000000ll || (int64_t)a->K * a->M * 2 >= (int64_t)0x7f000000ll) return 0;
  }
  if ((((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->partial) & 15)) return 0;
  // 128 x 256 tiles (a whole tap of a 256-channel map per workgroup: 48 KB staged per 32 MFMAs instead of 32 KB per 16) when the
  // operand has the columns for it AND a chip-filling split still leaves every workgroup >= 8 k-tiles -- 192x192: 316 vs 391 us with
  // the slab count that fills the chip in either form (countr_gemm_tiles() is what a caller divides 256 by),
  // profiles/r3_conv_wgrad_microbench.txt
  const int wide = lin ? a->N : a->Cin;
  const long t256 = (long)(a->M / 128) * (a->N / 256);
  int form = ((wide % 256) == 0 && t256 > 0 && (long)(a->K / 64) * t256 >= 8 * 256) ? 2 : 1;
  // three taps of a kernel row per workgroup (conv only) where the map's rows are whole k-tiles and the k-range is long enough that
  // 12 Cout Cin / 128^2 tiles x a chip-filling split leave >= 16 k-tiles each: the 192 x 192 layer
  bool can3 = !lin && ((a->W % 64) == 0 || (a->W % 96) == 0);
  { const char* e = getenv("COUNTR_LEAN_WGRAD3"); if (e && atoi(e) == 0) can3 = false; }      // (A/B switch: the round-4 selection)
  if (can3 && (long)(a->K / 64) * (a->M / 128) * 3 * (a->Cin / 128) >= 16 * 256) form = 3;
  { const char* e = getenv("COUNTR_LEAN_WGRAD_FORM"); if (e) form = atoi(e); }
  if (form == 3 && !can3) form = 2;
  if (form == 2 && (wide % 256)) form = 1;
  if (form < 1 || form > 3) form = 1;
  if (a->rowsum_partial && a->rowsum_slabs != (a->splitk > 1 ? a->splitk : 1) * cwg_contributors(a, form)) return 0;   // caller sized another layout
  return form;
}

}  // namespace

// Slabs of rowsum_partial a (COL, IM2COL) / (COL, COL) bf16 split-K launch writes ([slabs][M]): splitk on the generic kernel, splitk x NC here.
int countr_lean_wgrad_rowsum_slabs(const countr_gemm_args* a, int lin) {
  countr_gemm_args b = *a;
  b.rowsum_partial = nullptr;       // (the layout check is what this call answers)
  if (!b.partial) b.partial = reinterpret_cast<float*>(16);   // (a sizing call may come before the workspace exists)
  const int form = cwg_form(&b, lin != 0);
  const int sk = a->splitk > 1 ? a->splitk : 1;
  if (!form) return sk;
  return sk * cwg_contributors(a, form);     // forms 1, 2: NC = tilesN * SUBN / 2 = (N / (128 WNB)) * WNB; form 3: tilesN
}

// Output tiles (workgroups per split-K slab) of such a launch: what the caller divides the CU count by to pick splitk.
int countr_lean_wgrad_tiles(const countr_gemm_args* a, int lin) {
  countr_gemm_args b = *a;
  b.rowsum_partial = nullptr;
  if (!b.partial) b.partial = reinterpret_cast<float*>(16);
  const int form = cwg_form(&b, lin != 0);
  if (!form) return ((a->M + 127) / 128) * ((a->N + 127) / 128);
  if (form == 3) return (a->M / 128) * 3 * (a->Cin / 128);
  return (a->M / 128) * (a->N / (128 * form));
}

static void cwg_fill(CwgArgs& g, const countr_gemm_args* a, bool lin, int form) {
  g.dy = (const char*)a->A; g.x = (const char*)a->B; g.part = a->partial; g.rowsum = a->rowsum_partial;
  g.Cout = a->M; g.Cin = lin ? 0 : a->Cin; g.H = lin ? 0 : a->H; g.Wd = lin ? 0 : a->W; g.N = a->N;
  g.lda = lin ? (int)a->lda : a->M; g.ldb = lin ? (int)a->ldb : a->Cin;
  g.tilesN = form == 3 ? 3 * (a->Cin / 128) : a->N / (128 * form); g.tiles = (a->M / 128) * g.tilesN;
  g.nkt = a->K / (form == 3 ? cwg3_ktile(a->W) : 64); g.Z = a->splitk > 1 ? a->splitk : 1;
  g.per = (g.nkt + g.Z - 1) / g.Z;
}

// Tile width (in 128-column units) a group of n (COL, COL) split-K launches runs at in ONE launch, or 0 when it does not qualify (the
// caller then launches them one by one): 2 to 10 nn.Linear weight gradients, each qualifying on its own; 256-column tiles when every
// problem has the columns for them.  (The accumulation order of an output element does not depend on the tile width.)
int countr_lean_wgrad_group_form(const countr_gemm_args* items, int n) {
  if (n < 2 || n > CWG_GROUP_MAX) return 0;
  int form = 2;
  for (int i = 0; i < n; ++i) {
    countr_gemm_args b = items[i];
    if (!b.partial) b.partial = reinterpret_cast<float*>(16);   // (sizing calls come before the workspaces exist)
    if (b.rowsum_partial) b.rowsum_slabs = (b.splitk > 1 ? b.splitk : 1) * (b.N / 128);
    if (!cwg_form(&b, true)) return 0;
    if (b.N % 256) form = 1;
  }
  { const char* e = getenv("COUNTR_LEAN_WGRAD_FORM"); if (e && atoi(e) == 1) form = 1; }
  return form;
}

// Returns 1 when the group does not qualify, otherwise the launch status.
int countr_lean_wgrad_group(const countr_gemm_args* items, int n, hipStream_t s) {
  const int form = countr_lean_wgrad_group_form(items, n);
  if (!form) return 1;
  CwgGroup grp;
  int total = 0;
  for (int i = 0; i < CWG_GROUP_MAX; ++i) {
    const countr_gemm_args* a = &items[i < n ? i : n - 1];
    cwg_fill(grp.it[i], a, true, form);
    grp.start[i] = i < n ? total : 0x7fffffff;      // (workgroup v runs the LAST problem whose start is <= v: entries past the group never match)
    if (i < n) {
      if (a->rowsum_partial && a->rowsum_slabs != grp.it[i].Z * (a->N / 128)) return 1;   // caller sized another layout
      total += grp.it[i].tiles * grp.it[i].Z;
    }
  }
  grp.start[CWG_GROUP_MAX] = total;     // the grid
  return form == 2 ? launch_cwg_group<2>(grp, s) : launch_cwg_group<1>(grp, s);
}

// Returns 1 when the launch does not qualify (gemm_kernel then runs it), otherwise the launch status.
int countr_lean_wgrad(const countr_gemm_args* a, int lin, hipStream_t s) {
  const int form = cwg_form(a, lin != 0);
  if (!form) return 1;
  CwgArgs g;
  cwg_fill(g, a, lin != 0, form);
  if (lin) return form == 2 ? launch_cwg<2, true>(g, s) : launch_cwg<1, true>(g, s);
  if (form == 3) return cwg3_ktile(a->W) == 64 ? launch_cwg3<4>(g, s) : launch_cwg3<6>(g, s);
  return form == 2 ? launch_cwg<2, false>(g, s) : launch_cwg<1, false>(g, s);
}
