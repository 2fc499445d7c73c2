// Fused self-attention forward for gfx950, software-pipelined (bf16 in/out, fp32 softmax + accumulation).
// Reference: Attention.forward, models_crossvit.py:82-94 (== timm Attention): softmax(q k^T * dh^-0.5) v on a packed
// qkv [B, N, 3, H, dh]; output [B, N, H*dh].  N = 576 (288 kept tokens in MAE pretraining), dh = 64 (encoder) or 32 (decoder).
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); a wave owns 32 rows; two workgroups per CU, i.e. two waves per
// SIMD.  What differs from the first-generation kernel (flash_attn.hip, kept for the backward) is the instruction stream of a
// wave.  There a K/V tile was processed as QK^T (MFMA only) -> softmax (200 VALU, no MFMA) -> PV (MFMA only).  Here:
//   * S^T = K Q^T and O^T += V^T P^T use v_mfma_f32_32x32x16_bf16: a lane owns ONE query row (l & 31) and 16 of the 32 keys of a
//     block, so the row max needs one v_permlane32_swap per tile and m / l are scalars per lane.
//   * the loop is skewed by one tile: step t issues the 2*KS MFMAs of S(t+1) = K(t+1) Q^T and the PV MFMAs of tile t, and between
//     consecutive MFMAs the exp2 / row-sum / bf16-pack work of S(t) (then the row max of S(t+1)) in fixed slots
//     (__builtin_amdgcn_sched_barrier pins the interleave; fragment reads are issued two slots ahead of their MFMA).
//   * P never leaves the lane: PV consumes the keys in the order the S^T accumulator holds them (k-slot j of half h <-> key
//     16 s + 4 h + (j & 3) + 8 (j >> 2)), V^T fragments are read in that order with ds_read_b64_tr_b16.
//   * waves whose 32 query rows lie beyond N (N = 576 = 4.5 x 128) only help staging and skip all MFMA / VALU work.
//   * deferred rescale (reference max kept while the running max grows by <= 8 in log2 units), decided after ALL PV MFMAs of the
//     pending tile were issued and before the next tile is exponentiated.
// K runs one tile ahead of V ("stage j" = {K(j+1), V(j)}).  Two staging paths:
//   * dh = 64 (DMA): stages stream by LDS-DMA (global_load_lds, 16 B per lane, no VGPRs, no ds_write) into a 4-slot ring of
//     unpadded 128-byte rows, two stages ahead of their use; the bank-conflict swizzle sits on the per-lane SOURCE address and again
//     on the fragment reads (K: chunk ^= (row >> 1) & 7 -> ds_read_b128 of 32 distinct rows conflict-free; V: chunk ^= 4 for rows
//     with bit 1 set -> the four consecutive rows of a transposing read cover the 64 banks once).  Fragment reads are inline asm
//     with hand-counted lgkmcnt (a compiler-visible ds_read gets s_waitcnt vmcnt(0) in front of it while a DMA is in flight); a
//     stage is published by a counted s_waitcnt vmcnt + one raw s_barrier per step.
//   * dh = 32: register-staged double buffer (global_load at the top of a step, ds_write at its end, one barrier per step), K
//     rows pitched dh*2+16 bytes, V rows 64 bytes.
// Issue model behind the slot layout (tools/ubench_issue.hip, profiles/r2_issue_microbench.txt): with two waves per SIMD a
// 32x32x16 MFMA occupies the matrix pipe for 32 cycles but hides only ~12 cycles of VALU work; v_fma 2.8, v_exp_f32 8.3,
// v_cvt_pk_bf16_f32 4.7 cycles per wave instruction.  A tile costs a wave 16 MFMAs and ~570 VALU cycles (half of them the 32
// v_exp_f32), so the loop is VALU-bound by construction at dh = 64.
#include "common.hpp"
#ifndef COUNTR_FA_NOPIN
#define COUNTR_FA_NOPIN 0   // experiments: 1 = no end-of-slot sched_barrier in the pipelined step, 2 = one every fourth slot, 3 = none at all
#endif
#ifndef COUNTR_FA_PRIO
#define COUNTR_FA_PRIO 0
#endif
// Round 6: no running row max in the steady state.  P = exp2(S - m_ref) with m_ref = the row max of the FIRST key tile: a later score
// that exceeds it by d just makes that P 2^d -- exact in fp32 / bfloat16 (a power-of-two factor that cancels in O / l) as long as
// nothing overflows, and l >= max P tells afterwards whether anything could have.  So the loop carries no v_max3 chain, no cross-lane
// max and no rescale test (13 % of a step's issue cycles, profiles/r3_fa_step_microbench.txt: 1611 -> 1398); after the last tile every
// lane checks l < 2^64 and O finite, the workgroup ORs the verdicts through one LDS word, and a workgroup with a miss (a row whose
// later scores exceed the first tile's max by more than 64 in the log2 domain, or non-finite data) runs its strip AGAIN on the exact
// deferred-rescale loop below -- same result as before, twice the time, for inputs no trained attention produces.  bfloat16 build only:
// an fp16 P overflows at 2^16, so the fp16 library keeps the running max.  -DCOUNTR_FA_NOMAX=0: the round-2..5 kernel.
#ifndef COUNTR_FA_NOMAX
#define COUNTR_FA_NOMAX (!COUNTR_HALF_FP16)
#endif
#include <stdlib.h>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
typedef __attribute__((address_space(3))) void* lds_vptr_t;
typedef const __attribute__((address_space(1))) void* glb_vptr_t;
typedef __attribute__((address_space(3))) const char* lds_cptr_t;

template <typename F, int... I>
__device__ __forceinline__ void fa_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void fa_static_for(F&& f) { fa_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int DH, bool DMA> struct FaCfg {
  static constexpr int KP = DMA ? DH * 2 : DH * 2 + 16;                 // K row pitch (bytes); padded: odd multiple of 16
  static constexpr int VP = DMA ? DH * 2 : (DH == 64 ? 192 : 64);       // V row pitch (bytes); padded: +-64 mod 256
  static constexpr int KT = 64 * KP, VT = 64 * VP, STAGE = KT + VT;
  static constexpr int NSLOT = DMA ? 4 : 2;
  static constexpr int OP = DH * 2 + 16;                                // pitch of the output staging rows
  static constexpr int OST = DMA ? 0 : NSLOT * STAGE;                   // DMA: the output staging aliases the (drained) ring
  static constexpr int LDS = DMA ? NSLOT * STAGE : NSLOT * STAGE + 4 * 32 * OP;
  static constexpr int LDS_ALL = LDS + 16;                              // + the workgroup's "fast path overflowed" word (COUNTR_FA_NOMAX)
  static_assert(!DMA || (DH == 64 && 4 * 32 * OP <= NSLOT * STAGE), "DMA path is laid out for 128-byte rows");
};

// units of exp work (2 scores each) finished by the end of MFMA slot j; slots = 2*KS QK^T MFMAs then 4*DB PV MFMAs
// NM (no running max, see COUNTR_FA_NOMAX): the slots MAX0 .. NS-1 carry no row-max work, so the exp units are spread over all slots
// (deadlines: the PV MFMA of slot NQK + e reads units 4 (e / DB) .. +3, which must be complete by the end of the slot before it).
template <int DH, bool NM> struct FaSched;
template <> struct FaSched<64, false> {
  static constexpr int NS = 16, MAX0 = 12;   // row max of S(t+1) spread over slots MAX0 .. NS-1
  static constexpr int PRE = 2;              // units done under the latency of the first fragment reads, ahead of slot 0
  static constexpr int unit_end[16] = {3, 4, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 16, 16, 16, 16};
};
template <> struct FaSched<64, true> {
  static constexpr int NS = 16, MAX0 = 12;
  static constexpr int PRE = 2;
  static constexpr int unit_end[16] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 16, 16};
};
template <> struct FaSched<32, false> {
  static constexpr int NS = 8, MAX0 = 6;
  static constexpr int PRE = 2;
  static constexpr int unit_end[8] = {4, 7, 10, 12, 14, 16, 16, 16};
};
template <> struct FaSched<32, true> {
  static constexpr int NS = 8, MAX0 = 6;
  static constexpr int PRE = 2;
  static constexpr int unit_end[8] = {4, 6, 8, 10, 12, 14, 16, 16};
};

// Built with -fno-slp-vectorize -fno-honor-nans (countr_amd/build.py): plain -O3 SLP-packs adjacent fp32 adds into v_pk_add_f32
// (slower beside MFMAs, and it collected the row-sum adds of a whole tile into one dependent chain) and wraps every fmaxf of an
// MFMA result in a canonicalising v_max.  The arithmetic stays compiler-visible: hipcc pads the trans-use and MFMA-result
// hazards only for instructions it can see.
__device__ __forceinline__ float max3(float a, float b, float c3) { return fmaxf(fmaxf(a, b), c3); }

__device__ __forceinline__ uint32_t fa_lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lds_cptr_t)p; }
template <int OFF> __device__ __forceinline__ bf16x8_t fa_read_b128(uint32_t a) {
  bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ bf16x8_t fa_read_tr(uint32_t a) {   // two transposing reads 8 rows apart
  s16x4_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 8 * 128));
  const s16x8_t r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, r);
}
// at most PENDING LDS reads outstanding; the fragment is threaded through so that its MFMA cannot move above the wait
template <int PENDING> __device__ __forceinline__ void fa_lds_wait(bf16x8_t& f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(PENDING)); }
template <int PENDING> __device__ __forceinline__ void fa_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PENDING) : "memory"); }

// ABL (timing experiments only, compiled in by -DCOUNTR_FA_ABL_BUILD=<k>; 1-6 give wrong results): 1 = K/V staged in the prologue only (no loads,
// LDS stores or barriers in the steps), 2 = 1 + no v_exp, 3 = 1 + no MFMA, 4 = staging and barriers only, 5 = empty kernel,
// 6 = one tile only (prologue + epilogue), 7 = correct output + s_memtime stamps of wave 0 written to lse (tools/stamp_attn.py),
// 8 = no fragment reads in the steps (MFMAs on stale registers)
// PRE: q already carries scale * log2(e) (the frozen encoder's q projection is pre-scaled at weight-packing time, engine.py): the
// scores come out of the MFMA in the exp2 domain, and because the QK^T accumulators start at -m_ref instead of 0 the softmax
// needs no subtract / scale FMA at all (one VALU instruction per score less: 32 of ~135 per tile and wave).
template <int DH, bool RAGGED, int ABL = 0, bool PRE = false>
__global__ __launch_bounds__(256, 2) void fa_fwd_pipe_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                             float* __restrict__ lse, int N, int H, float c /* scale*log2e */) {
  constexpr bool DMA = DH == 64;
  using C = FaCfg<DH, DMA>;
  constexpr bool NOMAX = COUNTR_FA_NOMAX && !COUNTR_HALF_FP16 && (ABL == 0 || ABL == 7);
  constexpr int KS = DH / 16;        // QK^T k-steps (16 channels each)
  constexpr int DB = DH / 32;        // 32-channel blocks of O^T
  constexpr int CPR = DH / 8;        // 16-byte chunks per K/V row
  constexpr int RPP = 256 / CPR;     // rows staged per pass of the 256 threads (register path)
  constexpr int PASSES = 64 / RPP;
  constexpr int NQK = 2 * KS;        // MFMA slots of QK^T
#ifndef COUNTR_FA_LA
#define COUNTR_FA_LA 5
#endif
#ifndef COUNTR_FA_SPLIT
#define COUNTR_FA_SPLIT 0
#endif
  constexpr bool SPLIT = DMA && COUNTR_FA_SPLIT;
  constexpr int LA = DMA ? COUNTR_FA_LA : 2;   // fragment look-ahead in MFMA slots (register-staged dh = 32 path: compiler-scheduled reads)
  constexpr bool STAGING = !((ABL >= 1 && ABL <= 3) || ABL == 9 || ABL == 10);   // 9 = 1 + no VALU at all in the steps, 10 = 1 + no row max / rescale
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, ql = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  volatile int* const redo = reinterpret_cast<volatile int*>(smem + C::LDS);   // NOMAX: set by any lane whose fast-path sums left the safe range
  if (NOMAX && tid == 0) *redo = 0;                                             // (published by the prologue's barriers)
  const int qblocks = (N + 127) >> 7;
  // XCD-aware mapping (speed only): all query blocks of one (batch, head) get workgroup ids congruent mod 8 so that its K/V
  // is fetched into ONE XCD's L2.
  int bh, qb;
  const int nbh = gridDim.x / qblocks;
  if ((nbh & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    bh = xcd * (nbh >> 3) + j / qblocks;
    qb = j - (j / qblocks) * qblocks;
  } else {
    bh = blockIdx.x / qblocks;
    qb = blockIdx.x - bh * qblocks;
  }
  const int b = bh / H, h = bh - b * H;
  const int64_t rs = (int64_t)3 * H * DH;   // row stride (elements) of the packed qkv
  const bf16_t* qp = qkv + (int64_t)b * N * rs + h * DH;
  const bf16_t* kp = qp + H * DH;
  const bf16_t* vp = kp + H * DH;
  const int q0 = qb * 128 + wave * 32;
  const bool active = (ABL == 4) ? false : q0 < N;   // wave-uniform
  const int T = (ABL == 6) ? 1 : (N + 63) >> 6;
  if (ABL == 5 && N > 0) return;
  uint64_t tk0 = 0, tkc = 0, tks = 0, tkb = 0, tkp = 0;
  if (ABL == 7) tk0 = __builtin_readcyclecounter();
#if COUNTR_FA_PRIO   // experiment (round-3 verdict, 3e): static priority by wave slot -- the two waves of a SIMD alternate instead of contending
  {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(4, 0, 4)" : "=s"(hw));    // HW_ID.wave_id: the slot of this wave on its SIMD
    if (hw & 1) __builtin_amdgcn_s_setprio(COUNTR_FA_PRIO);
  }
#endif

  // ---- Q^T fragments (MFMA B operand): lane (ql, hh) holds channels 16 ks + 8 hh .. +7 of query q0 + ql
  bf16x8_t qf[KS];
  {
    const int q = q0 + ql;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4_t v = {0, 0, 0, 0};
      if (q < N) v = *reinterpret_cast<const u32x4_t*>(qp + (int64_t)q * rs + ks * 16 + hh * 8);
      qf[ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  // ================================================================ staging
  // register path: thread (srow, scc) moves 16-byte chunk scc of rows srow + RPP * pass
  const int srow = tid / CPR, scc = tid % CPR;
  u32x4_t kreg[PASSES], vreg[PASSES];
  auto gload = [&](const bf16_t* base, int tile, u32x4_t (&reg)[PASSES]) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int key = tile * 64 + srow + ps * RPP;
      if (!RAGGED || key < N) reg[ps] = *reinterpret_cast<const u32x4_t*>(base + (int64_t)key * rs + scc * 8);
      else reg[ps] = u32x4_t{0, 0, 0, 0};
    }
  };
  auto lstore = [&](char* dst, int pitch, const u32x4_t (&reg)[PASSES]) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) *reinterpret_cast<u32x4_t*>(dst + (srow + ps * RPP) * pitch + scc * 16) = reg[ps];
  };
  auto Kslot = [&](int slot) { return smem + slot * C::STAGE; };
  auto Vslot = [&](int slot) { return smem + slot * C::STAGE + C::KT; };
  // DMA path: wave w streams the 1-KiB pieces 2w, 2w+1 (8 rows each) of the K tile and of the V tile of a stage.  Lane L of piece
  // p lands in row R = 8p + (L >> 3), 16-byte slot L & 7, and fetches the chunk that slot holds under the swizzle.
  // The pieces are MUBUF LDS-DMA loads (buffer_load_dwordx4 ... offen lds): descriptor base = this (batch, head)'s K or V, a loop-
  // invariant 32-bit lane offset, the tile advance as a scalar offset -- no vector address arithmetic per piece -- and key rows beyond N
  // (ragged last tile) get an offset outside the descriptor, for which the DMA writes zeros.  (Round 2 used global_load_lds with a
  // 64-bit address pair per lane: ~95 issue cycles per piece and wave, four pieces per step in every wave's instruction stream.)
  uint32_t koff[2], voff[2];
  int srowd[2];
  if (DMA) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int R = 8 * (2 * wave + p) + (lane >> 3), cs = lane & 7;
      srowd[p] = R;
      koff[p] = (uint32_t)(R * (int)rs + ((cs ^ ((R >> 1) & 7)) << 3)) * 2u;
      voff[p] = (uint32_t)(R * (int)rs + ((cs ^ (((R >> 1) & 1) << 2)) << 3)) * 2u;
    }
  }
  const __amdgpu_buffer_rsrc_t srdK = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdV = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, 0x7ffffff0, 0x00020000);
  auto dma_tile = [&](const __amdgpu_buffer_rsrc_t srd, int tile, const uint32_t (&off)[2], char* dst) {
    const uint32_t so = (uint32_t)tile * 64u * (uint32_t)rs * 2u;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      uint32_t vo = off[p];
      if (RAGGED && tile * 64 + srowd[p] >= N) vo = 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_vptr_t)(dst + (2 * wave + p) * 1024), 16, vo, so, 0, 0);
    }
  };
  // stage j = {K(j+1), V(j)} -> ring slot j & 3
  auto dma_stage = [&](int j) {
    if (j + 1 < T) dma_tile(srdK, j + 1, koff, Kslot(j & 3));
    dma_tile(srdV, j, voff, Vslot(j & 3));
  };
  auto stage_count = [&](int j) { return j < T ? (j + 1 < T ? 4 : 2) : 0; };   // DMA instructions of stage j per wave
  auto vm_wait_pending = [&](int pending) {   // all but the newest `pending` DMA instructions of this wave have landed
    if (pending >= 4) fa_vm_wait<4>();
    else if (pending >= 2) fa_vm_wait<2>();
    else fa_vm_wait<0>();
  };

  // ================================================================ fragment reads
  // register path (compiler-visible)
  auto kfrag = [&](const char* K, int blk, int ks) {   // K rows 32 blk + ql, channels 16 ks + 8 hh .. +7
    return *reinterpret_cast<const bf16x8_t*>(K + (blk * 32 + ql) * C::KP + ks * 32 + hh * 16);
  };
  const int vlane = (4 * hh + ((lane & 15) >> 2)) * C::VP + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
  auto vfrag = [&](const char* V, int blk, int s, int d) {   // V^T: channels 32 d + (lane & 31), keys 32 blk + 16 s + kappa(hh, j)
    const char* a = V + (blk * 32 + s * 16) * C::VP + d * 64 + vlane;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a + 8 * C::VP));
    s16x8_t vv;
    vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3]; vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, vv);
  };
  // DMA path: per-lane byte offsets inside a slot (the swizzle folded in); tile row / block displacements go in the offset field
  uint32_t kbase[KS], vbase[DB];
  if (DMA) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kbase[ks] = ql * 128 + (((2 * ks + hh) ^ ((ql >> 1) & 7)) << 4);
    const int i = lane & 15, fl = (i >> 3) & 1;
#pragma unroll
    for (int d = 0; d < DB; ++d) vbase[d] = C::KT + (4 * hh + (i >> 2)) * 128 + ((d ^ fl) << 6) + ((lane >> 4) & 1) * 32 + (i & 3) * 8;
  }

  f32x16_t o[DB];
  float mref = 0.f, l0 = 0.f, l1 = 0.f;
  f32x16_t negm;     // PRE: -m_ref in every accumulator slot (all 16 scores of a lane belong to one query row)
  auto reset_state = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
    mref = 0.f; l0 = 0.f; l1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  };

  auto mask_tail = [&](f32x16_t (&S)[2], int tile) {   // keys >= N of the ragged last tile do not exist
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (tile * 64 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) S[blk][r] = -INFINITY;
  };
  // deferred rescale; every PV MFMA of the pending tile has been issued and l already contains its row sums.
  // mx: row max of the NEXT tile's scores Sn -- raw (scale c applies) or, PRE, already relative to m_ref in the exp2 domain.
  auto rescale_for = [&](float mx, f32x16_t (&Sn)[2]) {
    const float excess = PRE ? mx : mx * c - mref;     // how far the next tile's max lies above the reference max
    if (!__all(excess <= 8.0f)) {
      const float delta = fmaxf(excess, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      mref += delta;
      l0 *= alpha;
      l1 *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
      if (PRE) {   // the next tile's scores were accumulated against the old reference
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int i = 0; i < 16; ++i) Sn[blk][i] -= delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[i] = -mref;
      }
    }
  };

  // exp unit u (0..15) of tile scores S: registers 2u', 2u'+1 of block u >> 3 -> P word, row sums
  uint32_t P[2][8];
  auto exp_unit = [&](const f32x16_t (&S)[2], int u) {
    const int blk = u >> 3, w = u & 7;
    float p0 = PRE ? S[blk][2 * w] : __builtin_fmaf(S[blk][2 * w], c, -mref);
    float p1 = PRE ? S[blk][2 * w + 1] : __builtin_fmaf(S[blk][2 * w + 1], c, -mref);
    if (ABL != 2) { p0 = __builtin_amdgcn_exp2f(p0); p1 = __builtin_amdgcn_exp2f(p1); }
    l0 += p0;
    l1 += p1;
    P[blk][w] = pack2bf(p0, p1);
  };
  // split form (COUNTR_FA_SPLIT): the two v_exp_f32 of unit u are issued in one MFMA slot, the instructions that consume them (row
  // sums, bf16 pack) in the next, so that an in-order wave never waits for a transcendental result
  float pe[16][2];
  auto exp_only = [&](const f32x16_t (&S)[2], int u) {
    const int blk = u >> 3, w = u & 7;
    float p0 = PRE ? S[blk][2 * w] : __builtin_fmaf(S[blk][2 * w], c, -mref);
    float p1 = PRE ? S[blk][2 * w + 1] : __builtin_fmaf(S[blk][2 * w + 1], c, -mref);
    pe[u][0] = __builtin_amdgcn_exp2f(p0); pe[u][1] = __builtin_amdgcn_exp2f(p1);
  };
  auto exp_finish = [&](int u) {
    const int blk = u >> 3, w = u & 7;
    l0 += pe[u][0];
    l1 += pe[u][1];
    P[blk][w] = pack2bf(pe[u][0], pe[u][1]);
  };
  auto pfrag = [&](int blk, int s) {
    const u32x4_t v = {P[blk][4 * s], P[blk][4 * s + 1], P[blk][4 * s + 2], P[blk][4 * s + 3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };

  // ================================================================ prologue: S(0) = K(0) Q^T, its row max
  // DMA: K(0) -> K area of slot 3 (stage "-1"), stages 0 and 1 -> slots 0, 1 (all in flight together; stage 1 stays in flight).
  // registers: K(0) -> K area of slot 1, {K(1), V(0)} -> slot 0 (all loads issued before the first wait).
  f32x16_t SA[2], SB[2];
  auto prologue = [&]() __attribute__((always_inline)) {
  reset_state();
  const char* K0;
  if (DMA) {
    dma_tile(srdK, 0, koff, Kslot(3));
    dma_stage(0);
    if (1 < T) dma_stage(1);
    // K(0) first: S(0) and its row max are computed while stages 0 and 1 are still landing
    { const int pend = stage_count(0) + stage_count(1); if (pend >= 8) fa_vm_wait<8>(); else if (pend >= 6) fa_vm_wait<6>(); else if (pend >= 4) fa_vm_wait<4>(); else fa_vm_wait<2>(); }
    __builtin_amdgcn_s_barrier();
    K0 = Kslot(3);
  } else {
    u32x4_t k1reg[PASSES];
    gload(kp, 0, kreg);
    gload(vp, 0, vreg);
    if (T > 1) gload(kp, 1, k1reg);
    lstore(Kslot(1), C::KP, kreg);
    lstore(Vslot(0), C::VP, vreg);
    if (T > 1) lstore(Kslot(0), C::KP, k1reg);
    __syncthreads();
    K0 = Kslot(1);
  }
  if (active) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int i = 0; i < 16; ++i) SA[blk][i] = 0.f;
    if (DMA) {
      const uint32_t kb = fa_lds_addr(K0);
      bf16x8_t f[2 * KS];
      fa_static_for<2 * KS>([&](auto J) { constexpr int j = J; f[j] = fa_read_b128<(j & 1) * 4096>(kb + kbase[j >> 1]); });
      fa_static_for<2 * KS>([&](auto J) {
        constexpr int j = J;
        fa_lds_wait<2 * KS - 1 - j>(f[j]);
        __builtin_amdgcn_sched_barrier(0);
        SA[j & 1] = COUNTR_MFMA_32X32X16(f[j], qf[j >> 1], SA[j & 1], 0, 0, 0);
      });
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) SA[blk] = COUNTR_MFMA_32X32X16(kfrag(K0, blk, ks), qf[ks], SA[blk], 0, 0, 0);
    }
    if (RAGGED && T == 1) mask_tail(SA, 0);
    float mx = fmaxf(SA[0][0], SA[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = max3(mx, SA[0][r], SA[1][r]);
    mref = xor32_max(mx) * (PRE ? 1.f : c);
    if (PRE) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < 16; ++i) SA[blk][i] -= mref;
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = -mref;
    }
  }
  if (DMA) {   // stage 0 = {K(1), V(0)} has landed for every wave (stage 1 stays in flight)
    vm_wait_pending(stage_count(1));
    __builtin_amdgcn_s_barrier();
  } else {
    __syncthreads();   // register path: the K area of slot 1 is rewritten at the end of step 0
  }
  };   // prologue

  // ================================================================ one pipelined step (tile t, t + 1 < T)
  // Sc = S(t) -> P, Sn = S(t+1) = K(t+1) Q^T, O += V(t)^T P^T, rescale decision for tile t + 1 (NM: none -- m_ref stays the first
  // tile's row max).  Reads stage t.
  auto step = [&](auto NMt, const int t, f32x16_t (&Sc)[2], f32x16_t (&Sn)[2]) __attribute__((always_inline)) {
    constexpr bool NM = decltype(NMt)::value;
    using SC = FaSched<DH, NM>;
    const int slot = DMA ? (t & 3) : (t & 1);
    if (STAGING) {
      if (DMA) {
        if (t + 2 < T) dma_stage(t + 2);   // slot (t+2)&3 was last read in step t-2: two barriers ago
      } else {
        if (t + 2 < T) gload(kp, t + 2, kreg);
        gload(vp, t + 1, vreg);
      }
    }
    uint64_t ta = 0, tb = 0, tc = 0;
    if (ABL == 7) ta = __builtin_readcyclecounter();
    if (active) {
      const char* K = Kslot(slot);
      const char* V = Vslot(slot);
      const uint32_t sb = fa_lds_addr(Kslot(slot));
      uint32_t ka[KS], va[DB];
      if (DMA) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ka[ks] = sb + kbase[ks];
#pragma unroll
        for (int d = 0; d < DB; ++d) va[d] = sb + vbase[d];
      }
      bf16x8_t fr[SC::NS];   // operand A of slot j (lives LA slots)
      // Fragment look-ahead.  Round 2 requested a fragment two MFMA slots ahead of its use: with eight waves reading the CU's LDS a
      // ds_read_b128 / transposing pair comes back after 130-300 cycles, two slots are 64-100 cycles of issue, so most slots began
      // with a wait and the matrix pipe idled (the loop ran 1740 cycles per step against ~1000 for the same 16 MFMA + 108 VALU per
      // wave WITHOUT LDS reads: tools/ubench_pingpong.hip mode 0).  LA slots ahead costs 4 VGPRs per slot of the 50 this kernel has spare.
      constexpr auto lds_after = [](int j) { int n = 0; for (int k = 1; k <= LA; ++k) if (j + k < SC::NS) n += (j + k < NQK) ? 1 : 2; return n; };
      auto issue = [&](auto J) {   // fragment read(s) of MFMA slot j
        constexpr int j = J;
        if constexpr (ABL == 8) { fr[j] = qf[j & (KS - 1)]; return; }     // timing experiment: no fragment reads
        if constexpr (j < NQK) {
          if constexpr (DMA) fr[j] = fa_read_b128<(j & 1) * 4096>(ka[j >> 1]);
          else fr[j] = kfrag(K, j & 1, j >> 1);
        } else {
          constexpr int e = j - NQK, blk = e / (2 * DB), s = (e / DB) & 1, d = e % DB;   // PV slot: key block, k-step, channel block
          if constexpr (DMA) fr[j] = fa_read_tr<(blk * 32 + s * 16) * 128>(va[d]);
          else fr[j] = vfrag(V, blk, s, d);
        }
      };
      fa_static_for<LA>([&](auto J) { issue(J); });
      float mx = 0.f;
      if (ABL != 9) {
#pragma unroll
        for (int u = 0; u < SC::PRE; ++u) { if (SPLIT) exp_only(Sc, u); else exp_unit(Sc, u); }
      }
      __builtin_amdgcn_sched_barrier(0);
      fa_static_for<SC::NS>([&](auto J) {
        constexpr int j = J;
        if constexpr (j + LA < SC::NS) issue(std::integral_constant<int, j + LA>{});
        if constexpr (DMA) {   // LDS instructions issued after the reads of slot j: those of slots j+1 .. j+LA
          constexpr int pend = lds_after(j);
          static_assert(pend <= 15, "lgkmcnt is a 4-bit counter");
          fa_lds_wait<pend>(fr[j]);
#if COUNTR_FA_NOPIN != 3
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if constexpr (j < NQK) {
          constexpr int blk = j & 1, ks = j >> 1;
          if constexpr (ABL == 3) {
            Sn[blk][ks] = __builtin_bit_cast(float, (int)fr[j][0] | ((int)qf[ks][0] << 16));
          } else if constexpr (ks == 0) {
            f32x16_t z;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.f;
            Sn[blk] = COUNTR_MFMA_32X32X16(fr[j], qf[ks], PRE ? negm : z, 0, 0, 0);
          } else {
            Sn[blk] = COUNTR_MFMA_32X32X16(fr[j], qf[ks], Sn[blk], 0, 0, 0);
          }
        } else {
          constexpr int e = j - NQK, blk = e / (2 * DB), s = (e / DB) & 1, d = e % DB;
          if constexpr (ABL == 3) o[d][e] += __builtin_bit_cast(float, (int)fr[j][0] | ((int)pfrag(blk, s)[0] << 16));
          else o[d] = COUNTR_MFMA_32X32X16(fr[j], pfrag(blk, s), o[d], 0, 0, 0);
        }
        constexpr int u0 = (j == 0) ? SC::PRE : SC::unit_end[j == 0 ? 0 : j - 1], u1 = SC::unit_end[j];
        if constexpr (ABL != 9) {
          if constexpr (SPLIT) {   // exponentials of this slot's units first, then the consumers of the previous slot's
            constexpr int f0 = (j == 0) ? 0 : ((j == 1) ? SC::PRE : SC::unit_end[j - 2]);
#pragma unroll
            for (int u = u0; u < u1; ++u) exp_only(Sc, u);
#pragma unroll
            for (int u = f0; u < u0; ++u) exp_finish(u);
            if constexpr (j + 1 == SC::NS) {
#pragma unroll
              for (int u = u0; u < 16; ++u) exp_finish(u);
            }
          } else {
#pragma unroll
            for (int u = u0; u < u1; ++u) exp_unit(Sc, u);
          }
        }
        if constexpr (j >= SC::MAX0 && ABL != 9 && ABL != 10) {   // row max of S(t+1), a share per slot
          constexpr int PER = 16 / (SC::NS - SC::MAX0), r0 = (j - SC::MAX0) * PER;
          if (RAGGED && j == SC::MAX0 && t + 2 == T) mask_tail(Sn, t + 1);
          if constexpr (!NM) {
#pragma unroll
            for (int r = r0; r < r0 + PER; ++r) mx = (r == 0) ? fmaxf(Sn[0][0], Sn[1][0]) : max3(mx, Sn[0][r], Sn[1][r]);
          }
        }
#if COUNTR_FA_NOPIN == 0
        __builtin_amdgcn_sched_barrier(0);
#elif COUNTR_FA_NOPIN == 2
        if constexpr ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // a pin every fourth slot only
#endif
      });
      if constexpr (!NM) {
        if (ABL != 9 && ABL != 10) rescale_for(xor32_max(mx), Sn);
      }
    }
    if (ABL == 7) tb = __builtin_readcyclecounter();
    if (STAGING) {
      if (DMA) {
        vm_wait_pending(stage_count(t + 2));   // stage t+1 has landed (this wave's part); stage t+2 stays in flight
        if (ABL == 7) tc = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
      } else {
        if (t + 2 < T) lstore(Kslot(slot ^ 1), C::KP, kreg);
        lstore(Vslot(slot ^ 1), C::VP, vreg);
        if (ABL == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tc = __builtin_readcyclecounter(); }
        __syncthreads();
      }
    }
    if (ABL == 7) { const uint64_t td = __builtin_readcyclecounter(); tkc += tb - ta; tks += tc - tb; tkb += td - tc; }
  };
  // ---- last tile: no next scores
  auto tail = [&](const int t, f32x16_t (&Sc)[2]) __attribute__((always_inline)) {
    if (active) {
      const int slot = DMA ? (t & 3) : (t & 1);
      if (DMA) {
        const uint32_t sb = fa_lds_addr(Kslot(slot));
        bf16x8_t f[4 * DB];
        fa_static_for<4 * DB>([&](auto E) {
          constexpr int e = E, blk = e / (2 * DB), s = (e / DB) & 1, d = e % DB;
          f[e] = fa_read_tr<(blk * 32 + s * 16) * 128>(sb + vbase[d]);
        });
#pragma unroll
        for (int u = 0; u < 8; ++u) exp_unit(Sc, u);
        __builtin_amdgcn_sched_barrier(0);
        fa_static_for<4 * DB>([&](auto E) {   // the PV MFMAs of key block 0 run under the exp work of key block 1
          constexpr int e = E, blk = e / (2 * DB), s = (e / DB) & 1, d = e % DB;
          fa_lds_wait<2 * (4 * DB - 1 - e)>(f[e]);
          __builtin_amdgcn_sched_barrier(0);
          o[d] = COUNTR_MFMA_32X32X16(f[e], pfrag(blk, s), o[d], 0, 0, 0);
          if constexpr (e < 2 * DB) {
            constexpr int per = 8 / (2 * DB);
#pragma unroll
            for (int u = 8 + e * per; u < 8 + (e + 1) * per; ++u) exp_unit(Sc, u);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) exp_unit(Sc, u);
        const char* V = Vslot(slot);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int d = 0; d < DB; ++d) o[d] = COUNTR_MFMA_32X32X16(vfrag(V, blk, s, d), pfrag(blk, s), o[d], 0, 0, 0);
      }
    }
  };

  auto attend = [&](auto NMt) __attribute__((always_inline)) {   // prologue + all key tiles
    prologue();
    if (ABL == 7) tkp = __builtin_readcyclecounter();
    int t = 0;
    for (; t + 2 < T; t += 2) {
      step(NMt, t, SA, SB);
      step(NMt, t + 1, SB, SA);
    }
    if (t + 1 < T) {
      step(NMt, t, SA, SB);
      tail(t + 1, SB);
    } else {
      tail(t, SA);
    }
  };
  uint64_t tke = 0;
  if constexpr (NOMAX) {
    attend(std::true_type{});
    if (ABL == 7) tke = __builtin_readcyclecounter();
    if (active) {   // did every P, l and O of this lane stay finite and far from the top of the fp32 range?
      float z = 0.f;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) z = __builtin_fmaf(o[d][i], 0.f, z);      // NaN iff some O is inf / NaN
      if (!(l0 + l1 < 0x1p64f) || !(z == 0.f)) *redo = 1;
    }
    __syncthreads();   // publishes the verdict; also: every wave is done with the last V tile (the output staging aliases the ring)
    if (*redo != 0) {  // workgroup-uniform: the strip again, on the exact loop (running max, deferred rescale)
      attend(std::false_type{});
      if (DMA) __builtin_amdgcn_s_barrier();
    }
  } else {
    attend(std::false_type{});
    if (ABL == 7) tke = __builtin_readcyclecounter();
    // ---- epilogue: normalise, stage the wave's [32][DH] bf16 block through its private LDS rows, store whole rows (16-byte chunks)
    if (DMA) __builtin_amdgcn_s_barrier();   // the staging rows alias the ring: every wave is done with the last V tile (no DMA pending)
  }
  if (active) {
    char* ost = smem + C::OST + wave * 32 * C::OP;
    const float lt = xor32_sum(l0 + l1);
    const float inv = 1.f / lt;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const uint2 pk = make_uint2(pack2bf(o[d][4 * rg] * inv, o[d][4 * rg + 1] * inv), pack2bf(o[d][4 * rg + 2] * inv, o[d][4 * rg + 3] * inv));
        *reinterpret_cast<uint2*>(ost + ql * C::OP + (d * 32 + 8 * rg + 4 * hh) * 2) = pk;
      }
    const int q = q0 + ql;
    if (ABL != 7 && lse && hh == 0 && q < N) lse[((int64_t)b * H + h) * N + q] = (mref + log2f(lt)) * 0.6931471805599453f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < (32 * CPR) / 64; ++j) {
      const int idx = lane + 64 * j, r = idx / CPR, cc = idx % CPR;
      const uint4 v = *reinterpret_cast<const uint4*>(ost + r * C::OP + cc * 16);
      if (q0 + r < N) *reinterpret_cast<uint4*>(out + ((int64_t)b * N + q0 + r) * (H * DH) + h * DH + cc * 8) = v;
    }
  }
  if (ABL == 7 && lse && wave == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t tend = __builtin_readcyclecounter();
    if (lane == 0) {
      float* d = lse + blockIdx.x * 8;
      d[0] = (float)(tend - tk0); d[1] = (float)(tkp - tk0); d[2] = (float)tkc; d[3] = (float)tks; d[4] = (float)tkb;
      d[5] = (float)(tend - tke); d[6] = (float)(tk0 & 0xffffff); d[7] = (float)(tke - tkp);
    }
  }
}

template <int DH>
int launch_fa_fwd_pipe(const void* qkv, void* out, float* lse, int B, int N, int H, float c, hipStream_t s) {
  using C = FaCfg<DH, DH == 64>;
  dim3 grid(B * H * ((N + 127) / 128)), block(256);
#ifdef COUNTR_FA_ABL_BUILD     // timing experiments (bash tools/exp_file.sh flash_attn_fwd abl<k> -DCOUNTR_FA_ABL_BUILD=<k>): results are wrong
  if (DH == 64 && N % 64 == 0) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pipe_kernel<64, false, COUNTR_FA_ABL_BUILD>), hipFuncAttributeMaxDynamicSharedMemorySize, FaCfg<64, true>::LDS_ALL);
    hipLaunchKernelGGL((fa_fwd_pipe_kernel<64, false, COUNTR_FA_ABL_BUILD>), grid, block, (FaCfg<64, true>::LDS_ALL), s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, c);
    COUNTR_LAUNCH_CHECK("countr_attn_fwd (ablation)");
  }
#endif
  // the dh = 64 ring is exactly 64 KiB; with the verdict word behind it the workgroup's LDS is a little over the default limit
  if (c <= 0.f) {   // pre-scaled q (see PRE): built for the encoder shape class only
    if (DH != 64 || N % 64) { countr_set_error("countr_attn_fwd: scale <= 0 (pre-scaled q) needs head_dim 64 and N % 64 == 0"); return -1; }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pipe_kernel<64, false, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_ALL);
    hipLaunchKernelGGL((fa_fwd_pipe_kernel<64, false, 0, true>), grid, block, C::LDS_ALL, s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, 1.f);
    COUNTR_LAUNCH_CHECK("countr_attn_fwd");
  }
  if (N % 64) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pipe_kernel<DH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_ALL);
    hipLaunchKernelGGL((fa_fwd_pipe_kernel<DH, true>), grid, block, C::LDS_ALL, s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, c);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_fwd_pipe_kernel<DH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_ALL);
    hipLaunchKernelGGL((fa_fwd_pipe_kernel<DH, false>), grid, block, C::LDS_ALL, s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, c);
  }
  COUNTR_LAUNCH_CHECK("countr_attn_fwd");
}

}  // namespace

// Called by countr_attn_fwd (flash_attn.hip).  dh must be 32 or 64.
int countr_attn_fwd_pipelined(const void* qkv, void* out, float* lse, int B, int N, int H, int dh, float scale, hipStream_t s) {
  const float c = scale * 1.4426950408889634f;
  if (dh == 64) return launch_fa_fwd_pipe<64>(qkv, out, lse, B, N, H, c, s);
  return launch_fa_fwd_pipe<32>(qkv, out, lse, B, N, H, c, s);
}
