// gemm_kernel.hpp -- the generic MFMA GEMM kernel template of gemm.hip and its tile-shape selection, shared by the translation units
// that instantiate it (gemm_bf16_a.hip, gemm_bf16_b.hip, gemm_f32.hip: one ~19 000-instruction kernel per (dtype, operand modes,
// variant) -- 19 of them -- compiled in parallel instead of in one 97-second unit).
#pragma once
#include "common.hpp"
#include <type_traits>
#include <utility>
#include "../../include/countr_hip.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;  // fp32 path / default tile
constexpr int ROW_PITCH = 144;      // 128 B of K + 16 B pad
constexpr int OP_BYTES = 18432;     // per operand per stage (max over layouts), fp32 path

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
  static constexpr int EPC = 8;     // elements per 16-byte chunk
  static constexpr int BK = 64;
  static constexpr int COL_PITCH = 272;  // 128 rows * 2 B + 16
};
template <> struct Cfg<float> {
  static constexpr int EPC = 4;
  static constexpr int BK = 32;
  static constexpr int COL_PITCH = 528;  // 128 rows * 4 B + 16
};

struct OpDesc {
  const char* ptr;
  int64_t ld;   // elements
  int rows;     // valid rows of this operand (M or N)
  int H, W, C;  // conv geometry
};

// ---------------------------------------------------------------------------------------------
// Global -> register -> LDS staging, one specialisation per addressing mode.
// Each thread owns 4 chunks (16 B) of the 128 x BK operand tile.
// ---------------------------------------------------------------------------------------------
template <typename T, int MODE> struct Loader;

template <typename T> struct Loader<T, COUNTR_OP_ROW> {
  static constexpr int EPC = Cfg<T>::EPC;
  const char* rp[4];
  int kc;
  __device__ void init(const OpDesc& d, int row0, int kstart, int tid) {
    kc = (tid & 7) * EPC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row0 + (tid >> 3) + 32 * i;
      rp[i] = (r < d.rows) ? d.ptr + (int64_t)r * d.ld * sizeof(T) : nullptr;
    }
  }
  __device__ void load(int k0, int kend, uint4 (&v)[4]) {
    const int k = k0 + kc;
    const bool kok = (k + EPC) <= kend;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = make_uint4(0, 0, 0, 0);
      if (kok && rp[i]) v[i] = *reinterpret_cast<const uint4*>(rp[i] + (int64_t)k * sizeof(T));
    }
  }
  __device__ void store(char* lds, int tid, const uint4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(lds + ((tid >> 3) + 32 * i) * ROW_PITCH + (tid & 7) * 16) = v[i];
  }
};

template <typename T> struct Loader<T, COUNTR_OP_COL> {
  static constexpr int EPC = Cfg<T>::EPC;
  static constexpr int CPR = 128 / EPC;       // chunks per k-row
  static constexpr int KSTEP = 256 / CPR;     // k-rows covered per pass
  const char* base;
  int64_t ldb;
  __device__ void init(const OpDesc& d, int row0, int kstart, int tid) {
    const int r0 = row0 + (tid % CPR) * EPC;
    base = (r0 + EPC <= d.rows) ? d.ptr + (int64_t)r0 * sizeof(T) : nullptr;
    ldb = d.ld * (int64_t)sizeof(T);
  }
  __device__ void load(int k0, int kend, uint4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + (threadIdx.x / CPR) + KSTEP * i;
      v[i] = make_uint4(0, 0, 0, 0);
      if (base && k < kend) v[i] = *reinterpret_cast<const uint4*>(base + (int64_t)k * ldb);
    }
  }
  __device__ void store(char* lds, int tid, const uint4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(lds + ((tid / CPR) + KSTEP * i) * Cfg<T>::COL_PITCH + (tid % CPR) * 16) = v[i];
  }
};

// rows = pixels (b,y,x) of an NHWC map, k = tap*C + c, 3x3 window, zero padding 1.
template <typename T> struct Loader<T, COUNTR_OP_IM2ROW> {
  static constexpr int EPC = Cfg<T>::EPC;
  const char* ptr;
  int pix[4], py[4], px[4];  // linear pixel, y, x ; pix = -1 when the row is out of range
  int H, W, C, kc;
  __device__ void init(const OpDesc& d, int row0, int kstart, int tid) {
    ptr = d.ptr; H = d.H; W = d.W; C = d.C;
    kc = (tid & 7) * EPC;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = row0 + (tid >> 3) + 32 * i;
      pix[i] = (m < d.rows) ? m : -1;
      px[i] = m % W;
      py[i] = (m / W) % H;
    }
  }
  __device__ void load(int k0, int kend, uint4 (&v)[4]) {
    const int tap = k0 / C;                 // wave-uniform: BK divides C
    const int ci = k0 - tap * C + kc;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const bool kok = (k0 + kc + EPC) <= kend;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = make_uint4(0, 0, 0, 0);
      const int yy = py[i] + dy, xx = px[i] + dx;
      if (kok && pix[i] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
        v[i] = *reinterpret_cast<const uint4*>(
            ptr + ((int64_t)(pix[i] + dy * W + dx) * C + ci) * sizeof(T));
    }
  }
  __device__ void store(char* lds, int tid, const uint4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(lds + ((tid >> 3) + 32 * i) * ROW_PITCH + (tid & 7) * 16) = v[i];
  }
};

// rows = tap*C + c, k = pixel (b,y,x): the wgrad view of the same gather.
template <typename T> struct Loader<T, COUNTR_OP_IM2COL> {
  static constexpr int EPC = Cfg<T>::EPC;
  static constexpr int CPR = 128 / EPC;
  static constexpr int KSTEP = 256 / CPR;
  static constexpr int BK = Cfg<T>::BK;
  const char* ptr;
  int py[4], px[4];
  int H, W, C, ci, dy, dx;
  bool colok;
  __device__ void init(const OpDesc& d, int row0, int kstart, int tid) {
    ptr = d.ptr; H = d.H; W = d.W; C = d.C;
    const int r0 = row0 + (tid % CPR) * EPC;
    colok = (r0 + EPC) <= d.rows;
    const int tap = r0 / C;
    ci = r0 - tap * C;
    dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = kstart + (tid / CPR) + KSTEP * i;
      px[i] = p % W;
      py[i] = (p / W) % H;
    }
  }
  __device__ void load(int k0, int kend, uint4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = k0 + (threadIdx.x / CPR) + KSTEP * i;
      v[i] = make_uint4(0, 0, 0, 0);
      const int yy = py[i] + dy, xx = px[i] + dx;
      if (colok && p < kend && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
        v[i] = *reinterpret_cast<const uint4*>(ptr + ((int64_t)(p + dy * W + dx) * C + ci) * sizeof(T));
      // advance this chunk's pixel by one K tile
      px[i] += BK;
      while (px[i] >= W) { px[i] -= W; py[i] = (py[i] + 1 == H) ? 0 : py[i] + 1; }
    }
  }
  __device__ void store(char* lds, int tid, const uint4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(lds + ((tid / CPR) + KSTEP * i) * Cfg<T>::COL_PITCH + (tid % CPR) * 16) = v[i];
  }
};


// ---------------------------------------------------------------------------------------------
// bf16 path: direct-to-LDS staging (global_load_lds, 16 B per lane, no VGPR round trip, no ds_write).
// The DMA destination is lane-linear (wave-uniform base + lane*16), so tiles are stored unpadded and the
// bank-conflict swizzle is applied to the per-lane SOURCE address and again on the fragment read:
//   row-like tile [128 rows][8 chunks]  : chunk c of row r lives in slot  c ^ swz_row(r)
//   col-like tile [64 k-rows][16 chunks]: chunk c of k-row k lives in slot c ^ swz_col(k)
// Masked chunks (padding pixels, ragged edges) read a 16-byte zero page instead, so every lane always issues.
// ---------------------------------------------------------------------------------------------
__device__ uint4 g_zero_page[2] = {};

typedef __attribute__((address_space(3))) void* lds_vptr_t;
typedef const __attribute__((address_space(1))) void* glb_vptr_t;
__device__ __forceinline__ void dma16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_vptr_t)g, (lds_vptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz_col(int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); }
// row-like tiles: a 16-lane fragment group reads rows  base + (i>>2)*16 + (i&3)  (i = 0..15), i.e. row bits {0,1,4,5}
// vary.  Bit 0 selects the half of the 256-B bank row (rows are 128 B); bits {1,4,5} feed the 3-bit chunk swizzle, so the
// 16 lanes hit 16 distinct 16-byte slots (conflict-free ds_read_b128).
__device__ __forceinline__ int swz_row(int r) { return ((r >> 1) & 1) | (((r >> 4) & 3) << 1); }


// ROWS = tile rows of this operand (128 or 256), NW = waves in the workgroup.  A wave-instruction moves one 1-KiB group:
// row-like tiles: 8 rows x 128 B; col-like tiles (ROWS == 128 only): 4 k-rows x 256 B.  Group index = pass * NW + wave.
template <int MODE, int ROWS, int NW> struct DmaLoader;

template <int ROWS, int NW> struct DmaLoader<COUNTR_OP_ROW, ROWS, NW> {
  static constexpr int PASSES = ROWS / (8 * NW);
  static constexpr bool HAS_FAST = true;
  const char* rp[PASSES];
  int kc[PASSES];  // swizzled k-chunk (elements)
  uint32_t voff[PASSES];   // fast path: byte offset of this lane's chunk from the operand base (row clamped into the matrix)
  const char* ubase;
  __device__ void init(const OpDesc& d, int row0, int kstart, int wave, int lane) {
    ubase = d.ptr;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int rl = (i * NW + wave) * 8 + (lane >> 3);
      kc[i] = ((lane & 7) ^ swz_row(rl)) * 8;
      const int r = row0 + rl;
      rp[i] = (r < d.rows) ? d.ptr + (int64_t)r * d.ld * 2 : nullptr;
      voff[i] = (uint32_t)(((int64_t)min(r, d.rows - 1) * d.ld + kc[i]) * 2);
    }
  }
  // Fast path (full k-tiles, operand < 4 GiB): the per-tile address is a UNIFORM base (SGPR pair, advanced with scalar adds)
  // plus a loop-invariant 32-bit lane offset -> no per-piece vector address arithmetic or bounds selects in the loop.  Rows
  // past the matrix are clamped to its last row: they only feed output rows / columns that the epilogue never stores.
  __device__ __forceinline__ void issue_fast(int k0, char* lds, int wave) {
    const char* ub = ubase + (int64_t)k0 * 2;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) dma16(ub + voff[i], lds + (i * NW + wave) * 1024);
  }
  __device__ void issue(int k0, int kend, char* lds, int wave) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
#if defined(COUNTR_ABL) && COUNTR_ABL == 4   // timing experiment: no per-tile address arithmetic (always the first k-tile)
      const void* src = (const void*)rp[i];
#else
      const int k = k0 + kc[i];
      const void* src = ((k + 8) <= kend && rp[i]) ? (const void*)(rp[i] + (int64_t)k * 2) : (const void*)g_zero_page;
#endif
      dma16(src, lds + (i * NW + wave) * 1024);
    }
  }
};

template <int ROWS, int NW> struct DmaLoader<COUNTR_OP_COL, ROWS, NW> {
  static_assert(ROWS == 128, "K-strided operands are only staged as 128-row tiles");
  static constexpr int PASSES = 16 / NW;
  static constexpr bool HAS_FAST = true;
  const char* base[PASSES];
  int krow[PASSES];
  int64_t ldb;
  uint32_t voff[PASSES];
  const char* ubase;
  __device__ void init(const OpDesc& d, int row0, int kstart, int wave, int lane) {
    ldb = d.ld * 2;
    ubase = d.ptr;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      krow[i] = (i * NW + wave) * 4 + (lane >> 4);
      const int r0 = row0 + ((lane & 15) ^ swz_col(krow[i])) * 8;
      base[i] = (r0 + 8 <= d.rows) ? d.ptr + (int64_t)r0 * 2 : nullptr;
      voff[i] = (uint32_t)((int64_t)krow[i] * ldb + (int64_t)min(r0, d.rows - 8) * 2);
    }
  }
  __device__ __forceinline__ void issue_fast(int k0, char* lds, int wave) {   // see the row-like loader
    const char* ub = ubase + (int64_t)k0 * ldb;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) dma16(ub + voff[i], lds + (i * NW + wave) * 1024);
  }
  __device__ void issue(int k0, int kend, char* lds, int wave) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int k = k0 + krow[i];
      const void* src = (base[i] && k < kend) ? (const void*)(base[i] + (int64_t)k * ldb) : (const void*)g_zero_page;
      dma16(src, lds + (i * NW + wave) * 1024);
    }
  }
};

template <int ROWS, int NW> struct DmaLoader<COUNTR_OP_IM2ROW, ROWS, NW> {
  static constexpr int PASSES = ROWS / (8 * NW);
  static constexpr bool HAS_FAST = true;
  const char* ptr;
  int pix[PASSES], py[PASSES], px[PASSES], kc[PASSES];
  int H, W, C;
  int64_t voff[PASSES];     // fast path: byte offset of (pixel, k-chunk) from the tensor base
  uint32_t vmask[PASSES];   // fast path: bit t set <=> tap t of this lane's pixel lies inside the image (zero padding otherwise)
  __device__ void init(const OpDesc& d, int row0, int kstart, int wave, int lane) {
    ptr = d.ptr; H = d.H; W = d.W; C = d.C;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int rl = (i * NW + wave) * 8 + (lane >> 3);
      kc[i] = ((lane & 7) ^ swz_row(rl)) * 8;
      const int m = row0 + rl;
      pix[i] = (m < d.rows) ? m : -1;
      px[i] = m % W;
      py[i] = (m / W) % H;
      voff[i] = ((int64_t)m * C + kc[i]) * 2;
      uint32_t vm = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = py[i] + t / 3 - 1, xx = px[i] + t % 3 - 1;
        if (m < d.rows && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) vm |= 1u << t;
      }
      vmask[i] = vm;
    }
  }
  // Fast path (full k-tiles): uniform base = tensor + tap shift + channel chunk (scalar arithmetic), loop-invariant lane offset,
  // and one bit test per piece for the zero padding instead of recomputing the tap geometry per lane.
  __device__ __forceinline__ void issue_fast(int k0, char* lds, int wave) {
    const int tap = k0 / C;
    const int cbase = k0 - tap * C;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const char* ub = ptr + ((int64_t)(dy * W + dx) * C + cbase) * 2;
    const uint32_t bit = 1u << tap;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const void* src = (vmask[i] & bit) ? (const void*)(ub + voff[i]) : (const void*)g_zero_page;
      dma16(src, lds + (i * NW + wave) * 1024);
    }
  }
  __device__ void issue(int k0, int kend, char* lds, int wave) {
    const int tap = k0 / C;
    const int cbase = k0 - tap * C;
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int yy = py[i] + dy, xx = px[i] + dx;
      const bool ok = (k0 + kc[i] + 8) <= kend && pix[i] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const void* src = ok ? (const void*)(ptr + ((int64_t)(pix[i] + dy * W + dx) * C + cbase + kc[i]) * 2) : (const void*)g_zero_page;
      dma16(src, lds + (i * NW + wave) * 1024);
    }
  }
};

template <int ROWS, int NW> struct DmaLoader<COUNTR_OP_IM2COL, ROWS, NW> {
  static_assert(ROWS == 128, "K-strided operands are only staged as 128-row tiles");
  static constexpr int PASSES = 16 / NW;
  static constexpr bool HAS_FAST = true;
  const char* ptr;
  int py[PASSES], px[PASSES], krow[PASSES], ci[PASSES], dy[PASSES], dx[PASSES];
  int H, W, C;
  bool colok[PASSES];
  int64_t voff[PASSES];   // fast path: byte offset of (k-row, tap shift, channel chunk) from the pixel-block base
  int stepx, stepy;       // 64 pixels further = (stepy rows, stepx columns)
  __device__ void init(const OpDesc& d, int row0, int kstart, int wave, int lane) {
    ptr = d.ptr; H = d.H; W = d.W; C = d.C;
    stepy = 64 / W; stepx = 64 - stepy * W;
    stepy %= H;   // rows wrap per image: with the step reduced modulo H one conditional subtraction per advance is enough
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      krow[i] = (i * NW + wave) * 4 + (lane >> 4);
      const int r0 = row0 + ((lane & 15) ^ swz_col(krow[i])) * 8;
      colok[i] = (r0 + 8) <= d.rows;
      const int tap = r0 / C;
      ci[i] = r0 - tap * C;
      dy[i] = tap / 3 - 1; dx[i] = tap - (tap / 3) * 3 - 1;
      const int p = kstart + krow[i];
      px[i] = p % W;
      py[i] = (p / W) % H;
      voff[i] = ((int64_t)(krow[i] + dy[i] * W + dx[i]) * C + ci[i]) * 2;
    }
  }
  // Fast path (full k-tiles, sequential k order): uniform base = first pixel of the k-tile, loop-invariant lane offset; the pixel
  // coordinates advance by a fixed (rows, columns) step with one conditional wrap each instead of a data-dependent loop.
  __device__ __forceinline__ void issue_fast(int k0, char* lds, int wave) {
    const char* ub = ptr + (int64_t)k0 * C * 2;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int yy = py[i] + dy[i], xx = px[i] + dx[i];
      const bool ok = colok[i] && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const void* src = ok ? (const void*)(ub + voff[i]) : (const void*)g_zero_page;
      dma16(src, lds + (i * NW + wave) * 1024);
      px[i] += stepx; py[i] += stepy;
      if (px[i] >= W) { px[i] -= W; py[i] += 1; }
      if (py[i] >= H) py[i] -= H;
    }
  }
  __device__ void issue(int k0, int kend, char* lds, int wave) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int p = k0 + krow[i];
      const int yy = py[i] + dy[i], xx = px[i] + dx[i];
      const bool ok = colok[i] && p < kend && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const void* src = ok ? (const void*)(ptr + ((int64_t)(p + dy[i] * W + dx[i]) * C + ci[i]) * 2) : (const void*)g_zero_page;
      dma16(src, lds + (i * NW + wave) * 1024);
      px[i] += 64;
      while (px[i] >= W) { px[i] -= W; py[i] = (py[i] + 1 == H) ? 0 : py[i] + 1; }
    }
  }
};

constexpr bool is_rowlike(int mode) { return mode == COUNTR_OP_ROW || mode == COUNTR_OP_IM2ROW; }

// ---------------------------------------------------------------------------------------------
// LDS -> MFMA fragments.  `row` is the tile-local row this lane contributes (already including the
// N-side permutation), `row4` the first of the 4 consecutive rows this lane addresses for a
// transpose read.
// ---------------------------------------------------------------------------------------------
// The reads are INLINE ASM on purpose: for compiler-visible ds_reads the backend inserts s_waitcnt vmcnt(0) in front of
// them whenever an LDS-DMA load is in flight (it cannot prove that the DMA destination and the read do not alias), which
// serialises "stream tile t+1 / multiply tile t" completely.  With opaque reads the kernel owns both counters: vmcnt for
// the DMA stages (explicit s_waitcnt + barrier before a stage is read) and lgkmcnt for the fragments (lds_wait below).
typedef __attribute__((address_space(3))) const char* lds_cptr_t;
__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lds_cptr_t)p; }

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// row-like fragment read with the tile-row displacement in the instruction's offset field: for the permuted row order the swizzle
// term of a lane is the same for all 16-row tiles of a wave (they differ in row bits 2, 3 and 6 only), so one base address per
// (operand, k-step) serves every tile -> ~4 instead of ~24 vector adds per k-tile.
template <int OFF> __device__ __forceinline__ bf16x8_t lds_read_b128_off(uint32_t a) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}

template <int MODE> struct FragReads { static constexpr int N = is_rowlike(MODE) ? 1 : 2; };  // LDS instructions per fragment

template <int MODE>
__device__ __forceinline__ bf16x8_t frag_bf16(uint32_t lds, int row, int row4, int kk, int lane) {
  const int g = lane >> 4, i = lane & 15;
  if constexpr (is_rowlike(MODE)) {
    bf16x8_t v;
    const uint32_t a = lds + row * 128 + (((kk * 4 + g) ^ swz_row(row)) << 4);
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
    return v;
  } else {
    // K-major image [k][row]: two hardware-transposing reads of a [4 k][16 rows] block each.
    const int k0 = kk * 32 + g * 8 + (i >> 2), k1 = k0 + 4;
    const uint32_t a0 = lds + k0 * 256 + (((row4 >> 3) ^ swz_col(k0)) << 4) + (row4 & 7) * 2;
    const uint32_t a1 = lds + k1 * 256 + (((row4 >> 3) ^ swz_col(k1)) << 4) + (row4 & 7) * 2;
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
  }
}

// Wait until at most PENDING LDS reads are outstanding; the fragments are threaded through the asm so that the MFMAs
// consuming them cannot be scheduled above the wait.
template <int PENDING>
__device__ __forceinline__ void lds_wait(bf16x8_t (&x)[4], bf16x8_t (&w)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])
               : "n"(PENDING));
}

// fp32: 4 consecutive MFMA k-steps' operands; lane (i, g) holds k = c16*16 + 4*g + s, s = 0..3.
template <int MODE>
__device__ __forceinline__ float4 frag_f32(const char* lds, int row, int c16, int lane) {
  const int g = lane >> 4;
  if constexpr (is_rowlike(MODE)) {
    return *reinterpret_cast<const float4*>(lds + row * ROW_PITCH + (c16 * 16 + g * 4) * 4);
  } else {
    const char* p = lds + (c16 * 16 + g * 4) * Cfg<float>::COL_PITCH + row * 4;
    float4 r;
    r.x = *reinterpret_cast<const float*>(p);
    r.y = *reinterpret_cast<const float*>(p + Cfg<float>::COL_PITCH);
    r.z = *reinterpret_cast<const float*>(p + 2 * Cfg<float>::COL_PITCH);
    r.w = *reinterpret_cast<const float*>(p + 3 * Cfg<float>::COL_PITCH);
    return r;
  }
}

// TMW = 16-row MFMA tiles per wave along M (4: 64x64 wave tile, 8: 128x64 wave tile -> half the staged bytes and 25 % fewer
// fragment reads per MFMA; used for the 256x256 workgroup tile).
// SPEC = wave specialisation: the workgroup gets WM*WN extra LOADER waves (wave w + WM*WN shares its SIMD with compute wave w)
// that do nothing but stream tiles into a STAGES-deep LDS ring, while the compute waves only read fragments and issue MFMAs.  A
// 1-KiB LDS-DMA piece costs its wave ~60 issue cycles, about as much per k-tile as the tile's MFMAs: in one instruction stream
// the two serialise, in two streams on the same SIMD they overlap.
template <typename T, int MA, int MB, int STAGES, int WM, int WN, int TMW = 4, int NLD = 0>
__global__ __launch_bounds__(64 * (WM * WN + NLD)) void gemm_kernel(const countr_gemm_args g, const int skew_mul) {
  constexpr bool SPEC = NLD > 0;   // NLD loader waves behind the WM*WN compute waves
  static_assert(TMW == 4 || TMW == 8, "wave tile is 64x64 or 128x64");
  static_assert(!SPEC || (sizeof(T) == 2 && STAGES >= 3), "wave specialisation: bf16 path, >= 3 LDS stages");
  constexpr int BMt = 16 * TMW * WM, BNt = 64 * WN, NW = WM * WN;
  constexpr int SA = BMt * 128, SB = BNt * 128;  // bf16 stage bytes per operand
  constexpr int BK = Cfg<T>::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // stage s: A tile at smem + 2*s*OP_BYTES, B tile right behind it

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef COUNTR_GEMM_STAMP   // kernel-level timeline per workgroup (absolute s_memtime): entry, loop start, loop end, exit
  const uint64_t tl_entry = __builtin_readcyclecounter();
  uint64_t tl_loop0 = 0, tl_loop1 = 0, tl_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TLX(k) tl_x[k] = __builtin_readcyclecounter()
#else
#define TLX(k)
#endif
  const bool loader_wave = SPEC && (tid >> 6) >= WM * WN;
  const int wave = loader_wave ? (tid >> 6) - WM * WN : (tid >> 6);   // index inside its role
  const int tilesN = (g.N + BNt - 1) / BNt;
  // XCD-aware tile order: workgroup id b runs on XCD b % 8 (observed; speed only).  Give every XCD one contiguous range
  // of the (tile_m, tile_n) space so the tiles sharing an A row-panel / B panel sit behind the same L2 instead of being
  // re-fetched over the fabric by all 8 XCDs (fc2 4608x768x3072: ~208 MB -> ~66 MB per launch).  Bijective for any count.
  // With a z dimension (split-K slabs, batched GEMMs) the dispatch order is x fastest, then z: the XCD of a workgroup is
  // (blockIdx.x + gridDim.x * blockIdx.z) % 8, and the remap runs over the joint (z, tile) space -- an XCD then holds ~1/8 of the
  // (z, tile) pairs in z-major order, i.e. the tiles of ONE or two k-ranges (one or two batches) instead of a few tiles of every one:
  // the operand panels of a k-range are fetched by one L2, not by all eight.
  int lt, zz;
  {
#ifndef COUNTR_ZMAP
#define COUNTR_ZMAP 1
#endif
    const int nx = gridDim.x;
#if COUNTR_ZMAP
    const int lin = blockIdx.x + nx * blockIdx.z, nt_ = nx * gridDim.z;
#else
    const int lin = blockIdx.x, nt_ = nx;
#endif
    const int q = nt_ >> 3, r = nt_ & 7, x = lin & 7, j = lin >> 3;
    const int v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
#if COUNTR_ZMAP
    zz = v / nx; lt = v - zz * nx;
#else
    zz = blockIdx.z; lt = v;
#endif
  }
  const int tile_m = lt / tilesN, tile_n = lt - tile_m * tilesN;
  const int m0 = tile_m * BMt, n0 = tile_n * BNt;

  // batch / split-K decode
  int kstart = 0, kend = g.K;
  int64_t offA = 0, offB = 0, offC = 0;
  const int z = zz;
  const bool split = g.partial != nullptr;  // raw fp32 partial sums (split-K, also with splitk == 1)
  if (split) {
    const int nsplit = g.splitk > 1 ? g.splitk : 1;
    const int tiles = (g.K + BK - 1) / BK;
    const int per = (tiles + nsplit - 1) / nsplit;
    kstart = z * per * BK;
    kend = min(g.K, kstart + per * BK);
  } else if (g.nbatch > 1) {
    const int b0 = z / g.nb1, b1 = z - b0 * g.nb1;
    offA = b0 * g.sA0 + b1 * g.sA1;
    offB = b0 * g.sB0 + b1 * g.sB1;
    offC = b0 * g.sC0 + b1 * g.sC1;
  }

  OpDesc dA{reinterpret_cast<const char*>(g.A) + offA * (int64_t)sizeof(T), g.lda, g.M, g.H, g.W, g.Cin};
  OpDesc dB{reinterpret_cast<const char*>(g.B) + offB * (int64_t)sizeof(T), g.ldb, g.N, g.H, g.W, g.Cin};

  f32x4_t acc[TMW][4];
#pragma unroll
  for (int a = 0; a < TMW; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int wm0 = (wave / WN) * (16 * TMW), wn0 = (wave % WN) * 64;
  const int li = lane & 15;
  // N-side row permutation: MFMA output row i of tile tn is column wn0 + (i>>2)*16 + tn*4 + (i&3),
  // so that a lane ends up holding 16 consecutive output columns (vector stores in the epilogue).
  const int nrow_base = wn0 + (li >> 2) * 16 + (li & 3);
  const int nrow4_base = wn0 + (li & 3) * 16;
  // M-side: row-like bf16 operands use the same permuted row order as the N side (conflict-free swizzle, see swz_row);
  // the lane's output row for tile tm follows.  K-strided / fp32 operands keep consecutive rows.
  constexpr bool MPERM = (sizeof(T) == 2) && is_rowlike(MA);
  auto mrow = [&](int tm) { return MPERM ? (wm0 + (tm >> 2) * 64 + (li >> 2) * 16 + (tm & 3) * 4 + (li & 3)) : (wm0 + tm * 16 + li); };
  const int ntiles = (kend > kstart) ? (kend - kstart + BK - 1) / BK : 0;

  // Residual prefetch.  The fp32 residual tile of a (bias + residual -> fp32) GEMM is known before the first k-tile, but the
  // epilogue used to fetch it at the very end: with one wave per SIMD and 8 KB in flight per wave that was two exposed memory
  // round trips per workgroup (in-step proj 19.5 vs 10.3 us, fc2 40.5 vs 27.7 us against the same GEMMs with bf16 output).  The
  // compute waves now issue all 16 row-segment loads of their 64x64 sub-tile up front, in the staged epilogue's (row, chunk)
  // order, and the values wait in registers under the whole main loop.
  typedef __attribute__((ext_vector_type(4))) float resid4_t;
  constexpr bool RPRE_OK = sizeof(T) == 2 && TMW == 4 && (NLD == 0 || NLD == WM * WN);   // 256-VGPR budgets only
  const bool resid_pre = RPRE_OK && !loader_wave && !split && g.resid != nullptr && !g.out_bf16 && g.act == COUNTR_ACT_NONE && g.C2 == nullptr &&
                         g.nbatch <= 1 && ((g.ldc & 3) == 0) && (((uintptr_t)g.C & 15) == 0) && ((g.N & 3) == 0) && ((g.ldres & 3) == 0) &&
                         (((uintptr_t)g.resid & 15) == 0);
  auto stage_row = [&](int h, int r) {   // staged-epilogue row r (0..31) of half h -> row of the wave's 64-row sub-tile
    return MPERM ? (((2 * h + ((r >> 2) & 1)) >> 2) * 64 + (r >> 3) * 16 + ((2 * h + ((r >> 2) & 1)) & 3) * 4 + (r & 3)) : (h * 32 + r);
  };
  resid4_t rpre[RPRE_OK ? 2 : 1][RPRE_OK ? 8 : 1];
  TLX(0);
  if constexpr (RPRE_OK) {
    if (resid_pre) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int idx = lane + 64 * j, r = idx >> 4, cc = idx & 15;
          const int m = m0 + wm0 + stage_row(h, r), n = n0 + wn0 + cc * 4;
          rpre[h][j] = resid4_t{0.f, 0.f, 0.f, 0.f};
          if (m < g.M && n < g.N)
            rpre[h][j] = __builtin_nontemporal_load(reinterpret_cast<const resid4_t*>(g.resid + (int64_t)(g.res_mod > 0 ? (m % g.res_mod) : m) * g.ldres + n));
        }
      asm volatile("" ::: "memory");
    }
  }

  if constexpr (sizeof(T) == 2) {
    // ---------------- bf16: LDS-DMA staging, two stages, tile t+1 in flight while tile t is multiplied
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // k-skew: workgroups walk the k-tiles in rotated order so that, at any moment, the workgroups of an XCD read different
    // 128-byte k-columns.  In lock step they would all hit the few L2 channels that one k-column of a matrix with a
    // power-of-two-ish leading dimension maps to (partition camping).  Only the fp32 summation order changes.
    const int kskew = (MB == COUNTR_OP_IM2COL || ntiles == 0) ? 0 : (int)(((unsigned)lt * (unsigned)skew_mul) % (unsigned)ntiles);
    // (A channel-chunk-outer / tap-inner walk of the convolutions' k-tiles was tried for L2 locality and removed in round 2: FETCH_SIZE
    // 122 vs 115 MB x 2 per launch and 376 vs 373 us -- no effect; fabric traffic is 1.6 x the algorithmic input either way,
    // profiles/r2_gemm_conv192_pmc.txt.)
    // ... round 4: with the faster kernels of linear.hip / gemm256.hip it does pay in the step (4.62 -> 4.58 ms), and this kernel follows
    // their order on unsplit convolutions so that a sample's result does not depend on which kernel its batch size selects
    // (common.hpp: COUNTR_CONV_CHUNK_MAJOR)
    const bool chunk_major = COUNTR_CONV_CHUNK_MAJOR && MA == COUNTR_OP_IM2ROW && kstart == 0 && kend == g.K && g.K == 9 * g.Cin && ntiles <= 79;
    auto ktile = [&](int t) {
      int tt = t + kskew; if (tt >= ntiles) tt -= ntiles;
      if (chunk_major) { const int c = (tt * 57) >> 9; return (tt - 9 * c) * g.Cin + c * BK; }
      return kstart + tt * BK;
    };
    constexpr int NLW = SPEC ? NLD : NW;   // waves that stage tiles
    DmaLoader<MA, BMt, NLW> la;
    DmaLoader<MB, BNt, NLW> lb;
    TLX(1);
    la.init(dA, m0, kstart, wv, lane);
    lb.init(dB, n0, kstart, wv, lane);
    TLX(2);
    // uniform-base addressing when every k-tile of this launch is full and the operands span < 4 GiB (see DmaLoader); the
    // whole main loop is instantiated twice so that the fast variant carries no per-lane pointer selects
    const bool fullk = ((kend - kstart) % BK) == 0;
    const bool okA = !decltype(la)::HAS_FAST || (((int64_t)g.M * dA.ld * 2 < (int64_t)0xffff0000ll) && (MA != COUNTR_OP_COL || (int64_t)g.K * dA.ld * 2 < (int64_t)0xffff0000ll));
    const bool okB = !decltype(lb)::HAS_FAST || (((int64_t)g.N * dB.ld * 2 < (int64_t)0xffff0000ll) && (MB != COUNTR_OP_COL || (int64_t)g.K * dB.ld * 2 < (int64_t)0xffff0000ll));
    const bool fast_addr = fullk && okA && okB && (decltype(la)::HAS_FAST || decltype(lb)::HAS_FAST);
#ifndef COUNTR_ABL
#define COUNTR_ABL 0   // ablation builds (tools/ablate_gemm.sh): 1 = no MFMA, 2 = no fragment reads, 3 = DMA only for tile 0
#endif
    // optional fused bias gradient: sum_k A(m, k), accumulated by the waves of the first N-tile column only
    const bool do_rowsum = g.rowsum_partial != nullptr && tile_n == 0 && (wave % WN) == 0;
    f32x4_t accb[TMW];
#pragma unroll
    for (int q = 0; q < TMW; ++q) accb[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // One k-tile = NS steps of 16 MFMAs: step s covers k-step kk = s / NH (32 of the 64 staged k) and the 64-row half
    // h = s % NH of the wave tile.  Step s+1's fragments are requested before step s's MFMAs are issued (the W fragments
    // of a k-step are shared by its halves), so LDS round trips hide under the matrix pipe.
    constexpr int NH = TMW / 4, NS = 2 * NH;
    constexpr int XR = 4 * FragReads<MA>::N, WR = 4 * FragReads<MB>::N;   // LDS instructions per step for x / w fragments
    // Fragment pipeline.  set_tile() names the staged tile the NEXT request() reads; step<s, NEXT> first issues the reads of what
    // follows (NEXT = 1: step s+1 of the same tile; 2: step 0 of the tile named by set_tile -- the loop is rotated so that these
    // first fragments of tile t+1 fly under the last MFMA step of tile t instead of behind an idle matrix pipe after every
    // barrier; 0: nothing), waits for its own fragments and issues its 16 MFMAs.
    uint32_t sa = 0, sb = 0;
    bf16x8_t xf[2][4], wf[2][4];
    auto set_tile = [&](const char* sa_, const char* sb_) { sa = lds_addr(sa_); sb = lds_addr(sb_); };
      auto request = [&](auto S) {
        constexpr int s = decltype(S)::value, kk = s / NH, h = s % NH;
#if COUNTR_ABL == 2
        for (int q = 0; q < 4; ++q) { xf[s & 1][q] = __builtin_bit_cast(bf16x8_t, make_uint4(lane, q, kk, 1)); if (h == 0) wf[kk][q] = xf[s & 1][q]; }
#else
        if constexpr (MPERM) {
          const int r0 = mrow(0);
          const uint32_t xa = sa + r0 * 128 + (((kk * 4 + (lane >> 4)) ^ swz_row(r0)) << 4);
          static_for<4>([&](auto TM) { constexpr int tm = decltype(TM)::value; xf[s & 1][tm] = lds_read_b128_off<(h * 64 + tm * 4) * 128>(xa); });
        } else {
#pragma unroll
          for (int tm = 0; tm < 4; ++tm)
            xf[s & 1][tm] = frag_bf16<MA>(sa, mrow(h * 4 + tm), wm0 + (h * 4 + tm) * 16 + (li & 3) * 4, kk, lane);
        }
        if constexpr (h == 0) {
          if constexpr (is_rowlike(MB)) {
            const uint32_t wa = sb + nrow_base * 128 + (((kk * 4 + (lane >> 4)) ^ swz_row(nrow_base)) << 4);
            static_for<4>([&](auto TN) { constexpr int tn = decltype(TN)::value; wf[kk][tn] = lds_read_b128_off<tn * 4 * 128>(wa); });
          } else {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
              wf[kk][tn] = frag_bf16<MB>(sb, nrow_base + tn * 4, nrow4_base + tn * 4, kk, lane);
          }
        }
#endif
      };
      auto step = [&](auto S, auto NEXT_, auto WAIT_) {
        constexpr int s = decltype(S)::value, kk = s / NH, h = s % NH, NEXT = decltype(NEXT_)::value;
        constexpr bool WAIT = decltype(WAIT_)::value;   // false: the caller already waited for this step's fragments
        static_assert(NEXT != 1 || s + 1 < NS, "no next step in this tile");
        static_assert(NEXT != 2 || (((s & 1) == 1) && kk == 1), "the next tile's first request writes xf[0] / wf[0]");
        if constexpr (NEXT == 1) request(std::integral_constant<int, s + 1>{});
        if constexpr (NEXT == 2) request(std::integral_constant<int, 0>{});
#if COUNTR_ABL != 2
        constexpr int nxt = NEXT == 1 ? XR + (((s + 1) % NH) == 0 ? WR : 0) : (NEXT == 2 ? XR + WR : 0);   // reads allowed to stay in flight
        if constexpr (WAIT) lds_wait<(nxt > 15 ? 15 : nxt)>(xf[s & 1], wf[kk]);                           // lgkmcnt is a 4-bit counter
#endif
#if COUNTR_ABL == 1
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 a = __builtin_bit_cast(uint4, xf[s & 1][q]), b = __builtin_bit_cast(uint4, wf[kk][q]);
          acc[h * 4 + q][0][0] += __uint_as_float(a.x ^ b.x); acc[h * 4 + q][1][1] += __uint_as_float(a.y ^ b.y);
          acc[h * 4 + q][2][2] += __uint_as_float(a.z ^ b.z); acc[h * 4 + q][3][3] += __uint_as_float(a.w ^ b.w);
        }
#else
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn)
            acc[h * 4 + tm][tn] = COUNTR_MFMA_16X16X32(wf[kk][tn], xf[s & 1][tm], acc[h * 4 + tm][tn], 0, 0, 0);
#endif
        if (do_rowsum) {  // wave-uniform: row sums of the M-side operand = bias gradient of a wgrad GEMM
          const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(COUNTR_H16_ONE_PAIR, COUNTR_H16_ONE_PAIR, COUNTR_H16_ONE_PAIR, COUNTR_H16_ONE_PAIR));
#pragma unroll
          for (int tm = 0; tm < 4; ++tm)
            accb[h * 4 + tm] = COUNTR_MFMA_16X16X32(ones, xf[s & 1][tm], accb[h * 4 + tm], 0, 0, 0);
        }
      };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using YES = std::true_type; using NO = std::false_type;
    auto wait_frags = [&](auto S) {   // all LDS reads issued so far have landed (fragments of step S threaded through)
      constexpr int s = decltype(S)::value;
      lds_wait<0>(xf[s & 1], wf[s / NH]);
    };
    // steps 0 .. NS-2 of the tile whose step-0 fragments have been requested already
    auto steps_but_last = [&] {
      step(I0{}, I1{}, YES{});
      if constexpr (NS > 2) {
        step(I1{}, I1{}, YES{});
        step(I2{}, I1{}, YES{});
      }
    };
    // classic form: one whole tile, nothing in flight across its ends
    auto mma_tile = [&](const char* sa_, const char* sb_) {
      set_tile(sa_, sb_);
      request(I0{});
      steps_but_last();
      step(std::integral_constant<int, NS - 1>{}, I0{}, YES{});
    };
    auto main_loop = [&](auto FAST) {
      constexpr bool fast = decltype(FAST)::value;
      auto issueA = [&](int k0, char* lds) {
        if constexpr (fast && decltype(la)::HAS_FAST) la.issue_fast(k0, lds, wv); else la.issue(k0, kend, lds, wv);
      };
      auto issueB = [&](int k0, char* lds) {
        if constexpr (fast && decltype(lb)::HAS_FAST) lb.issue_fast(k0, lds, wv); else lb.issue(k0, kend, lds, wv);
      };
    if constexpr (STAGES == 1) {
      // Single LDS stage (32 KB): up to 5 workgroups stay resident per CU and hide each other's DMA latency.
      // Chosen by the host for big grids (>= ~3 workgroups per CU), where it beats per-workgroup double buffering.
      for (int t = 0; t < ntiles; ++t) {
        issueA(ktile(t), smem);
        issueB(ktile(t), smem + SA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mma_tile(smem, smem + SA);
        __builtin_amdgcn_s_barrier();  // all fragment reads of this tile are consumed before it is overwritten
      }
    } else if constexpr (SPEC) {
      // loader waves: wait for tile t (counted vmcnt, loads retire in order) -> barrier -> refill the slot that tile t-1 used;
      // compute waves: barrier -> multiply tile t.  One workgroup barrier per k-tile orders both hand-offs: a compute wave
      // arrives only after it issued tile t-1's MFMAs (their fragments were read), a loader only after tile t has landed.
      constexpr int PER = DmaLoader<MA, BMt, NLW>::PASSES + DmaLoader<MB, BNt, NLW>::PASSES;
      static_assert((STAGES - 2) * PER <= 63, "vmcnt immediate");
#ifdef COUNTR_GEMM_STAMP   // loaders: [1] load wait, [2] barrier, [3] DMA issue; compute waves: [2] barrier, [4] fragments + MFMA
      uint64_t sk1 = 0, sk2 = 0, sk3 = 0, sk4 = 0;
      const uint64_t sk0 = __builtin_readcyclecounter();
#define SSTAMP(x) const uint64_t x = __builtin_readcyclecounter()
#else
#define SSTAMP(x)
#endif
      if (loader_wave) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
          if (s < ntiles) {
            issueA(ktile(s), smem + s * (SA + SB));
            issueB(ktile(s), smem + s * (SA + SB) + SA);
          }
        int islot = STAGES - 1;
        for (int t = 0; t < ntiles; ++t) {
          SSTAMP(ua);
          if (t + STAGES - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          SSTAMP(ub);
          __builtin_amdgcn_s_barrier();
          SSTAMP(uc);
          if (t + STAGES - 1 < ntiles) {
            char* nxt = smem + islot * (SA + SB);
            issueA(ktile(t + STAGES - 1), nxt);
            issueB(ktile(t + STAGES - 1), nxt + SA);
          }
          islot = (islot + 1 == STAGES) ? 0 : islot + 1;
#ifdef COUNTR_GEMM_STAMP
          { SSTAMP(ud); sk1 += ub - ua; sk2 += uc - ub; sk3 += ud - uc; }
#endif
        }
      } else {
        // rotated: [steps 0..NS-2 of tile t] -> last step's fragments landed -> barrier t+1 -> request tile t+1's first fragments
        // -> last step's MFMAs of tile t.  (All LDS reads of tile t have completed before the barrier that lets the loaders
        // refill its slot; the MFMAs behind it only read registers.)
        int slot = 0;
        if (ntiles > 0) {
          __builtin_amdgcn_s_barrier();
          set_tile(smem, smem + SA);
          request(I0{});
        }
        for (int t = 0; t + 1 < ntiles; ++t) {
          SSTAMP(ua);
          steps_but_last();
          wait_frags(std::integral_constant<int, NS - 1>{});
          slot = (slot + 1 == STAGES) ? 0 : slot + 1;
          SSTAMP(ub);
          __builtin_amdgcn_s_barrier();
          set_tile(smem + slot * (SA + SB), smem + slot * (SA + SB) + SA);
          SSTAMP(uc);
          step(std::integral_constant<int, NS - 1>{}, I2{}, NO{});
#ifdef COUNTR_GEMM_STAMP
          { SSTAMP(ud); sk2 += uc - ub; sk4 += (ub - ua) + (ud - uc); }
#endif
        }
        if (ntiles > 0) {   // last tile: nothing follows
          steps_but_last();
          step(std::integral_constant<int, NS - 1>{}, I0{}, YES{});
        }
      }
#ifdef COUNTR_GEMM_STAMP
      if (g.nbatch == 1 && g.sC1 && lane == 0) {
        float* d = reinterpret_cast<float*>(g.sC1) + ((int64_t)blockIdx.x * (NW + NLD) + (tid >> 6)) * 8;
        d[0] = (float)(__builtin_readcyclecounter() - sk0); d[1] = (float)sk1; d[2] = (float)sk2; d[3] = (float)sk3; d[4] = (float)sk4;
        d[5] = (float)ntiles; d[6] = loader_wave ? 1.f : 2.f;
      }
#endif
    } else if constexpr (STAGES >= 3) {
      // Deep pipeline for SMALL grids (<= 1 workgroup per CU, nothing else to hide the DMA round trip): STAGES-1 tiles are in
      // flight while one is multiplied.  Counted vmcnt: DMA loads retire in order, so "at most (STAGES-2) tiles' worth of
      // loads outstanding" means tile t has landed.
      constexpr int PER = DmaLoader<MA, BMt, NW>::PASSES + DmaLoader<MB, BNt, NW>::PASSES;   // DMA instructions per wave per tile
      static_assert((STAGES - 2) * PER <= 63, "vmcnt immediate");
#pragma unroll
      for (int s = 0; s < STAGES - 1; ++s)
        if (s < ntiles) {
          issueA(ktile(s), smem + s * (SA + SB));
          issueB(ktile(s), smem + s * (SA + SB) + SA);
        }
      int slot = 0, islot = STAGES - 1;
      for (int t = 0; t < ntiles; ++t) {
        if (t + STAGES - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // tile t visible to all waves; every wave is done with the slot refilled below
        set_tile(smem + slot * (SA + SB), smem + slot * (SA + SB) + SA);
        request(I0{});                  // first fragments first: their LDS round trip runs under the DMA issue
        if (t + STAGES - 1 < ntiles) {
          char* nxt = smem + islot * (SA + SB);
          issueA(ktile(t + STAGES - 1), nxt);
          issueB(ktile(t + STAGES - 1), nxt + SA);
        }
        steps_but_last();
        step(std::integral_constant<int, NS - 1>{}, I0{}, YES{});
        slot = (slot + 1 == STAGES) ? 0 : slot + 1;
        islot = (islot + 1 == STAGES) ? 0 : islot + 1;
      }
    } else {
      // Two stages (64 KB, 2 workgroups per CU): tile t+1 streams in while tile t is multiplied.
      if (ntiles > 0) {
        issueA(ktile(0), smem);
        issueB(ktile(0), smem + SA);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // (Letting the second wave group of an 8-wave workgroup issue its share of tile t+1 in the MIDDLE of tile t was measured on the
      // experimental 256x256 tile and changes nothing: profiles/r2_gemm_256x256_experiment.txt.)
#ifdef COUNTR_GEMM_STAMP   // s_memtime anatomy (tools/stamp_gemm.py): per-wave cycles in DMA issue / MFMA steps / load wait / barrier
      uint64_t tki = 0, tkm = 0, tkw = 0, tkb = 0;
      const uint64_t tk0 = __builtin_readcyclecounter();
#define STAMP(x) const uint64_t x = __builtin_readcyclecounter()
#else
#define STAMP(x)
#endif
      for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        char* nxt = smem + (cur ^ 1) * (SA + SB);
        const bool more = t + 1 < ntiles && (COUNTR_ABL != 3);
        auto issue_next = [&] {
          if (more) {
            issueA(ktile(t + 1), nxt);
            issueB(ktile(t + 1), nxt + SA);
          }
        };
        STAMP(ta);
        set_tile(smem + cur * (SA + SB), smem + cur * (SA + SB) + SA);
        request(I0{});                        // first fragments first: their LDS round trip runs under the DMA issue below
        issue_next();
        STAMP(tb);
        steps_but_last();
        step(std::integral_constant<int, NS - 1>{}, I0{}, YES{});
        STAMP(tc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(td);
        __builtin_amdgcn_s_barrier();
#ifdef COUNTR_GEMM_STAMP
        { STAMP(te); tki += tb - ta; tkm += tc - tb; tkw += td - tc; tkb += te - td; }
#endif
      }
#ifdef COUNTR_GEMM_STAMP
      if (g.nbatch == 1 && g.sC1 && lane == 0) {   // (stamp builds only: the unused batch stride carries the debug buffer address)
        float* d = reinterpret_cast<float*>(g.sC1) + ((int64_t)blockIdx.x * NW + wv) * 8;
        d[0] = (float)(__builtin_readcyclecounter() - tk0); d[1] = (float)tki; d[2] = (float)tkm; d[3] = (float)tkw; d[4] = (float)tkb; d[5] = (float)ntiles;
      }
#endif
    }
    };
#ifdef COUNTR_GEMM_STAMP
    tl_loop0 = __builtin_readcyclecounter();
#endif
    if (fast_addr) main_loop(std::true_type{}); else main_loop(std::false_type{});
#ifdef COUNTR_GEMM_STAMP
    tl_loop1 = __builtin_readcyclecounter();
#endif
    // deep rings have no barrier behind the last tile: one here (all waves, loaders included) frees the ring for the staged epilogue
    if constexpr (STAGES >= 3) __builtin_amdgcn_s_barrier();
    if (loader_wave) return;   // no barrier after this point
    if (do_rowsum && (lane >> 4) == 0) {
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm) {
        const int m = m0 + mrow(tm);
        if (m < g.M) g.rowsum_partial[(int64_t)z * g.M + m] = accb[tm][0];
      }
    }
  } else {
    // ---------------- fp32 parity path: register-staged, padded tiles (2x2 waves only)
    static_assert(sizeof(T) == 2 || (WM == 2 && WN == 2 && TMW == 4), "fp32 path is 128x128 only");
    Loader<T, MA> la;
    Loader<T, MB> lb;
    la.init(dA, m0, kstart, tid);
    lb.init(dB, n0, kstart, tid);
    uint4 va[4], vb[4];
    if (ntiles > 0) {
      la.load(kstart, kend, va);
      lb.load(kstart, kend, vb);
      la.store(smem, tid, va);
      lb.store(smem + OP_BYTES, tid, vb);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      const int cur = t & 1;
      const bool more = (t + 1) < ntiles;
      if (more) {
        la.load(kstart + (t + 1) * BK, kend, va);
        lb.load(kstart + (t + 1) * BK, kend, vb);
      }
      const char* sa = smem + cur * 2 * OP_BYTES;
      const char* sb = sa + OP_BYTES;
#pragma unroll
      for (int c16 = 0; c16 < 2; ++c16) {
        float4 xf[4], wf[4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) xf[tm] = frag_f32<MA>(sa, wm0 + tm * 16 + li, c16, lane);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) wf[tn] = frag_f32<MB>(sb, nrow_base + tn * 4, c16, lane);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tn].x, xf[tm].x, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tn].y, xf[tm].y, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tn].z, xf[tm].z, acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tn].w, xf[tm].w, acc[tm][tn], 0, 0, 0);
          }
      }
      if (more) {
        la.store(smem + (cur ^ 1) * 2 * OP_BYTES, tid, va);
        lb.store(smem + (cur ^ 1) * 2 * OP_BYTES + OP_BYTES, tid, vb);
      }
      __syncthreads();
    }
  }

  // ---------------- epilogue: lane (j = lane&15, gq = lane>>4) owns, per tm, row m and the 16
  // consecutive columns nb .. nb+15 (acc[tm][tn][reg] -> column nb + tn*4 + reg).
  const int gq = lane >> 4;
  const int nb = n0 + wn0 + gq * 16;
  if (split) {
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm) {
      const int m = m0 + mrow(tm);
      if (m >= g.M) continue;
      float* dst = g.partial + ((int64_t)z * g.M + m) * g.N + nb;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        if (nb + tn * 4 < g.N) {
          const float v[4] = {acc[tm][tn][0], acc[tm][tn][1], acc[tm][tn][2], acc[tm][tn][3]};
          st4<float>(dst + tn * 4, v);
        }
      }
    }
    return;
  }
  // The epilogue is instantiated for the option combinations the engines use (bias / residual / GELU / output type as
  // compile-time flags) plus one fully general version: with every option tested at run time inside the (row, column) loops the
  // compiler produced ~4400 instructions of branches and waits, and a plain bias add cost 9 us on a 30-us GEMM.
  auto epilogue = [&](auto BIAS_, auto RESID_, auto ACT_, auto OBF_, auto GENERIC_) {
    constexpr bool GENERIC = decltype(GENERIC_)::value;
    constexpr bool BIASC = decltype(BIAS_)::value, RESIDC = decltype(RESID_)::value, OBFC = decltype(OBF_)::value;
    constexpr int ACTC = decltype(ACT_)::value;
    const bool has_bias = GENERIC ? (g.bias != nullptr) : BIASC;
    const bool has_resid = GENERIC ? (g.resid != nullptr) : RESIDC;
    const int act = GENERIC ? g.act : ACTC;
    const bool obf = GENERIC ? (g.out_bf16 != 0) : OBFC;
    float bv[4][4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      bv[tn][0] = bv[tn][1] = bv[tn][2] = bv[tn][3] = 0.f;
      if (has_bias && nb + tn * 4 < g.N) ld4<float>(g.bias + nb + tn * 4, bv[tn]);
    }
    TLX(3);
    if constexpr (!GENERIC && sizeof(T) == 2 && (TMW % 2 == 0)) {
      // Staged epilogue: a lane's natural stores are 8-byte (bf16) / 16-byte (fp32) pieces of 16 different rows per instruction
      // (4 lanes share a row): measured 8-24 % of a forward GEMM (COUNTR_ABL=5).  Instead the wave writes 32 rows x 64 columns of
      // finished values (bias / GELU applied) into its private slice of the now idle LDS ring and stores whole row segments:
      // 16-byte chunks, 8 (bf16) or 16 (fp32) consecutive lanes per 128 / 256-byte segment; the fp32 residual is read the same way.
      using OT = std::conditional_t<OBFC, bf16_t, float>;
      constexpr int ES = (int)sizeof(OT), EPC = 16 / ES, CPRW = 64 / EPC, PITCHO = 64 * ES + 16;
      constexpr bool C2OK = OBFC && ACTC == COUNTR_ACT_GELU;   // training fc1: bf16 pre-activation copy staged beside the output
      const bool al = (g.C2 == nullptr || (C2OK && ((uintptr_t)g.C2 & 15) == 0)) && g.nbatch <= 1 && (((int64_t)g.ldc * ES) & 15) == 0 && ((uintptr_t)g.C & 15) == 0 && (g.N % EPC) == 0 &&
                      (!RESIDC || ((g.ldres & 3) == 0 && ((uintptr_t)g.resid & 15) == 0));
      if (al) {
        char* ost = smem + wave * (C2OK ? 64 : 32) * PITCHO;
        const bool copy2 = C2OK && g.C2 != nullptr;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        typedef __attribute__((ext_vector_type(4))) float f32x4v_t;
#pragma unroll
        for (int h = 0; h < TMW / 2; ++h) {
#pragma unroll
          for (int tl = 0; tl < 2; ++tl) {
            const int tm = 2 * h + tl;
            const int r = MPERM ? ((li >> 2) * 8 + tl * 4 + (li & 3)) : (tl * 16 + li);
            char* dst = ost + r * PITCHO + gq * 16 * ES;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = acc[tm][tn][e] * g.alpha + bv[tn][e];
              if constexpr (C2OK) {
                if (copy2) *reinterpret_cast<uint2*>(dst + 32 * PITCHO + tn * 8) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
              }
              if (ACTC == COUNTR_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_t<T>(v[e]);
              }
              if constexpr (OBFC) *reinterpret_cast<uint2*>(dst + tn * 8) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
              else *reinterpret_cast<f32x4v_t*>(dst + tn * 16) = f32x4v_t{v[0], v[1], v[2], v[3]};
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (h == 0) TLX(4);
#pragma unroll
          for (int j = 0; j < (32 * CPRW) / 64; ++j) {
            const int idx = lane + 64 * j, r = idx / CPRW, cc = idx % CPRW;
            const int row = MPERM ? (((2 * h + ((r >> 2) & 1)) >> 2) * 64 + (r >> 3) * 16 + ((2 * h + ((r >> 2) & 1)) & 3) * 4 + (r & 3)) : (h * 32 + r);
            const int m = m0 + wm0 + row, n = n0 + wn0 + cc * EPC;
            if (m < g.M && n < g.N) {
              if constexpr (OBFC) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ost + r * PITCHO + cc * 16);
                *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(g.C) + offC + (int64_t)m * g.ldc + n) = v;
                if constexpr (C2OK) {
                  if (copy2)
                    *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(g.C2) + offC + (int64_t)m * g.ldc + n) =
                        *reinterpret_cast<const u32x4_t*>(ost + (32 + r) * PITCHO + cc * 16);
                }
              } else {
                f32x4v_t v = *reinterpret_cast<const f32x4v_t*>(ost + r * PITCHO + cc * 16);
                if constexpr (RESIDC) {
                  if constexpr (RPRE_OK) {
                    if (resid_pre) v += rpre[h][j];
                    else v += *reinterpret_cast<const f32x4v_t*>(g.resid + (int64_t)(g.res_mod > 0 ? (m % g.res_mod) : m) * g.ldres + n);
                  } else {
                    v += *reinterpret_cast<const f32x4v_t*>(g.resid + (int64_t)(g.res_mod > 0 ? (m % g.res_mod) : m) * g.ldres + n);
                  }
                }
                *reinterpret_cast<f32x4v_t*>(reinterpret_cast<float*>(g.C) + offC + (int64_t)m * g.ldc + n) = v;
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (h == 0) TLX(5);
        }
        TLX(6);
        return;
      }
    }
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm) {
      const int m = m0 + mrow(tm);
      if (m >= g.M) continue;
      const int64_t crow = offC + (int64_t)m * g.ldc;
      const float* rrow = has_resid ? g.resid + (int64_t)(g.res_mod > 0 ? (m % g.res_mod) : m) * g.ldres : nullptr;
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int n = nb + tn * 4;
        if (n >= g.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[tm][tn][e] * g.alpha + bv[tn][e];
        if (act == COUNTR_ACT_GELU_BWD) {   // dgrad of fc2 fused with GELU': C2 is the saved pre-activation (INPUT, layout of C)
          float h[4];
          if (obf) ld4<bf16_t>(reinterpret_cast<const bf16_t*>(g.C2) + crow + n, h);
          else ld4<float>(reinterpret_cast<const float*>(g.C2) + crow + n, h);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_t<T>(h[e]);
        } else if (g.C2) {                  // pre-activation copy (training fc1): rare, tested at run time in every version
          if (obf) st4<bf16_t>(reinterpret_cast<bf16_t*>(g.C2) + crow + n, v);
          else st4<float>(reinterpret_cast<float*>(g.C2) + crow + n, v);
        }
        if (act == COUNTR_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_t<T>(v[e]);
        }
        if (has_resid) {
          float r[4];
          ld4<float>(rrow + n, r);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += r[e];
        }
#if COUNTR_ABL == 5   // timing experiment: no output stores (only an unreachable one keeps the values alive)
        if (v[0] == 123.456f)
#endif
        {
          if (obf) st4<bf16_t>(reinterpret_cast<bf16_t*>(g.C) + crow + n, v);
          else st4<float>(reinterpret_cast<float*>(g.C) + crow + n, v);
        }
      }
    }
  };
  using std::true_type; using std::false_type;
  using A0 = std::integral_constant<int, COUNTR_ACT_NONE>; using A1 = std::integral_constant<int, COUNTR_ACT_GELU>;
  const bool hb = g.bias != nullptr, hr = g.resid != nullptr;
  if (g.out_bf16) {
    if (g.act == COUNTR_ACT_NONE && !hr) { if (hb) epilogue(true_type{}, false_type{}, A0{}, true_type{}, false_type{}); else epilogue(false_type{}, false_type{}, A0{}, true_type{}, false_type{}); }
    else if (g.act == COUNTR_ACT_GELU && hb && !hr) epilogue(true_type{}, false_type{}, A1{}, true_type{}, false_type{});
    else epilogue(false_type{}, false_type{}, A0{}, false_type{}, true_type{});
  } else {
    if (g.act == COUNTR_ACT_NONE && hb && hr) epilogue(true_type{}, true_type{}, A0{}, false_type{}, false_type{});
    else if (g.act == COUNTR_ACT_NONE && hb && !hr) epilogue(true_type{}, false_type{}, A0{}, false_type{}, false_type{});
    else if (g.act == COUNTR_ACT_NONE && !hb && !hr) epilogue(false_type{}, false_type{}, A0{}, false_type{}, false_type{});
    else if (g.act == COUNTR_ACT_NONE && !hb && hr) epilogue(false_type{}, true_type{}, A0{}, false_type{}, false_type{});
    else epilogue(false_type{}, false_type{}, A0{}, false_type{}, true_type{});
  }
#ifdef COUNTR_GEMM_STAMP
  if (g.nbatch == 1 && g.sC1 && tid == 0) {   // [workgroup][4] x uint64 behind the per-wave records (float offset 400000)
    uint64_t* t = reinterpret_cast<uint64_t*>(reinterpret_cast<float*>(g.sC1) + 400000) + (int64_t)blockIdx.x * 4;
    const uint64_t tl_pre = __builtin_readcyclecounter();   // before waiting for the store acknowledgements
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[0] = tl_entry; t[1] = tl_loop0; t[2] = tl_loop1; t[3] = __builtin_readcyclecounter();
    uint64_t* x = reinterpret_cast<uint64_t*>(reinterpret_cast<float*>(g.sC1) + 500000) + (int64_t)blockIdx.x * 8;
    for (int k = 0; k < 7; ++k) x[k] = tl_x[k];
    x[7] = tl_pre;
  }
#endif
}

template <typename T, int MA, int MB, int STAGES, int WM, int WN, int TMW = 4, int NLD = 0>
int launch_variant(const countr_gemm_args& a, hipStream_t s) {
  constexpr int BMt = 16 * TMW * WM, BNt = 64 * WN;
  constexpr int lds_bytes = sizeof(T) == 2 ? STAGES * (BMt + BNt) * 128 : 4 * OP_BYTES;
  const int tilesM = (a.M + BMt - 1) / BMt, tilesN = (a.N + BNt - 1) / BNt;
  const int zdim = a.partial ? (a.splitk > 1 ? a.splitk : 1) : (a.nbatch > 1 ? a.nbatch : 1);
  dim3 grid(tilesM * tilesN, 1, zdim);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, MA, MB, STAGES, WM, WN, TMW, NLD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<T, MA, MB, STAGES, WM, WN, TMW, NLD>), grid, dim3(64 * (WM * WN + NLD)), lds_bytes, s, a, 0);
  COUNTR_LAUNCH_CHECK("countr_gemm");
}

// Tile-shape / stage selection (bf16).  Bigger workgroup tiles re-use each staged operand for more MFMAs (the kernel is
// bound by the global->LDS path, see the ablation in DESIGN.md) but need enough tiles to fill 256 CUs.
template <typename T, int MA, int MB>
int launch(const countr_gemm_args& a, hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    const int zdim = a.partial ? (a.splitk > 1 ? a.splitk : 1) : (a.nbatch > 1 ? a.nbatch : 1);
    const int ksplit = a.partial ? (a.splitk > 1 ? a.splitk : 1) : 1;
    const int ktiles = (a.K / ksplit + 63) / 64;
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * zdim;
    // Double-buffered 128x128 tiles everywhere (tile t+1 streams in under tile t's MFMAs); launches that cannot give every CU a
    // second workgroup anyway (<= 256 workgroups: most (ROW, COL) dgrads, the small convolutions) run wave-specialised: 4 loader + 4
    // compute waves per workgroup on a 3-stage ring ((row, col) dgrads -20...-25 %, conv 24x24 54.6 -> 34.8 us).  The shapes that
    // dominate a step never get here: linear.hip / gemm256.hip / conv_wgrad.hip take them.  (Measured and removed, numbers in
    // profiles/HISTORY.md: 1 / 3 / 4 LDS stages, 64x128 / 256x128 / 256x256 / 128x256 tiles, a 4-stage and a 2-workgroup ring.)
    if (t128 <= 256 && ktiles >= 3) {
      // convolution wgrad: the im2col gather costs its loader ~16 VALU instructions per 1-KiB piece: 8 loader waves halve the
      // per-wave share (192^2 480 -> 463 us, 96^2 117 -> 107.5, 24^2 25.5 -> 24.2)
      if constexpr (MB == COUNTR_OP_IM2COL) return launch_variant<T, MA, MB, 3, 2, 2, 4, 8>(a, s);
      return launch_variant<T, MA, MB, 3, 2, 2, 4, 4>(a, s);
    }
  }
  return launch_variant<T, MA, MB, 2, 2, 2>(a, s);
}

}  // namespace
