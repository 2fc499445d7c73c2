// Fused self-attention for gfx950: the public forward entry (the kernel is in flash_attn_fwd.hip) and the fused BACKWARD.
// Reference: Attention.forward, models_crossvit.py:82-94 (== timm Attention): softmax(q k^T * dh^-0.5) v on a
// packed qkv [B, N, 3, H, dh]; output [B, N, H*dh].  N = 576, dh = 64 (encoder) or 32 (decoder).
// Shared layout facts (both passes of the backward use the forward's operand forms):
//   * S^T = K Q^T with the streamed tile as the MFMA A operand (ds_read_b128) and the resident fragment as B, so a lane holds
//     4 keys x 1 query row per tile: row reductions need two xor-steps (lanes l, l^16, l^32, l^48).
//   * exp2 domain: p = exp2(s * c - m), c = scale * log2(e).
//   * PV-type accumulations take their second operand from the lane's own registers by choosing the MFMA k-slot order
//     kappa(g, e) = {4g..4g+3, 16+4g..16+4g+3}; the transposed tile is read with ds_read_b64_tr_b16 in the same key order.
#include "common.hpp"
#include "../../include/countr_hip.h"
#include <stdlib.h>

namespace {

constexpr int FA_BQ = 128;   // query rows per workgroup
constexpr int FA_BKV = 64;   // keys per tile
// Bytes per LDS row of a staged [rows][DH] bf16 tile.  Row pitch = 32 mod 64 bytes makes BOTH fragment reads conflict-free over
// the 64 LDS banks: ds_read_b128 (16 rows x one 16-byte chunk per 16-lane group) and ds_read_b64_tr_b16 (8 rows x 4 chunks of
// 8 bytes per 32-lane group).  The former DH*2 + 16 pitch was 2-way conflicted on both (tools/lds_banks.py enumerates them).
#ifndef COUNTR_FA_PAD
#define COUNTR_FA_PAD 32
#endif
constexpr int fa_pitch(int dh) { return dh * 2 + COUNTR_FA_PAD; }

}  // namespace

// qkv: bf16 [B, N, 3, H, dh] packed (row stride 3*H*dh); out: bf16 [B, N, H*dh]; lse: optional fp32 [B, H, N]
// (natural-log sum-exp of the scaled scores, kept for a fused backward).  dh must be 32 or 64.
int countr_attn_fwd_pipelined(const void* qkv, void* out, float* lse, int B, int N, int H, int dh, float scale, hipStream_t s);   // flash_attn_fwd.hip

extern "C" int countr_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int dh, float scale, void* stream) {
  if (!qkv || !out || B <= 0 || N <= 0 || H <= 0) { countr_set_error("countr_attn_fwd: bad args"); return -1; }
  if (dh != 32 && dh != 64) { countr_set_error("countr_attn_fwd: head_dim must be 32 or 64"); return -1; }
  return countr_attn_fwd_pipelined(qkv, out, lse, B, N, H, dh, scale, reinterpret_cast<hipStream_t>(stream));
}

// =====================================================================================================
// Fused attention backward (autograd of Attention.forward, models_crossvit.py:84-91) without materialising P.
//   P_ij = exp2(s_ij * c - lse2_i),  dV_j = sum_i P_ij dO_i,  dP_ij = dO_i . V_j,  delta_i = dO_i . O_i,
//   dS_ij = P_ij (dP_ij - delta_i),  dQ_i = scale * sum_j dS_ij K_j,  dK_j = scale * sum_i dS_ij Q_i.
// One template, two passes in ONE launch (no atomics, deterministic):
//   MODE 0 (dQ)    : a workgroup owns 128 query rows (Q, dO fragments resident in VGPRs), streams K/V tiles through LDS;
//                    delta of its own rows from the resident dO and O.
//   MODE 1 (dK,dV) : a workgroup owns 128 keys (K, V fragments resident), streams Q/dO tiles (+ lse, and delta = dO . O computed from the
//                    O tile while staging) through LDS.
// Both passes use the forward kernel's layouts: "S-type" products X[streamed][resident] = T R^T with the streamed tile as
// the MFMA A operand (ds_read_b128) and the resident fragment as B; "PV-type" accumulations acc^T += T^T Y with T^T read by
// ds_read_b64_tr_b16 in the k-slot order kappa(g,e) = {4g+e, 16+4g+e}, so Y (P or dS) is packed from the lane's own registers.
// =====================================================================================================
namespace {

// bid / nblk: this pass's workgroup index and count (the launch holds both passes: flash_attn_bwd_kernel below)
template <int DH, int MODE, bool RAGGED>
__device__ __forceinline__ void fa_bwd_body(char* smem, const int bid, const int nblk, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ outp,
                                            const bf16_t* __restrict__ dout, const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int N, int H,
                                            float scale) {
  constexpr int KS = DH / 32, DT = DH / 16, PITCH = fa_pitch(DH), TILE = FA_BKV * PITCH, CPR = DH / 8;
  constexpr int PASSES = (FA_BKV * CPR) / 256;
  constexpr int STAGE = 2 * TILE + 512;  // two streamed tiles + (MODE 1) 64 lse2 + 64 delta floats
  const float c = scale * 1.4426950408889634f;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int rblocks = (N + FA_BQ - 1) / FA_BQ;
  int bh, rb;
  const int nbh = nblk / rblocks;
  if ((nbh & 7) == 0) {   // (nblk is then a multiple of 8: bid & 7 is the XCD in both halves of the launch)
    const int xcd = bid & 7, j = bid >> 3;
    bh = xcd * (nbh >> 3) + j / rblocks;
    rb = j - (j / rblocks) * rblocks;
  } else {
    bh = bid / rblocks;
    rb = bid - bh * rblocks;
  }
  const int b = bh / H, h = bh - b * H;
  const int64_t rs = (int64_t)3 * H * DH, ro = (int64_t)H * DH;
  const bf16_t* qp = qkv + (int64_t)b * N * rs + h * DH;
  const bf16_t* kp = qp + H * DH;
  const bf16_t* vp = kp + H * DH;
  const bf16_t* dop = dout + (int64_t)b * N * ro + h * DH;
  const bf16_t* op = outp + (int64_t)b * N * ro + h * DH;
  const float* lsep = lse + (int64_t)bh * N;
  const int r0 = rb * FA_BQ + wave * 32;  // first resident row (query for MODE 0, key for MODE 1) of this wave
  const bool active = r0 < N;             // wave-uniform

  // ---- resident fragments: R1 (Q | K) and R2 (dO | V), MFMA B-operand layout: lane (li, g) holds row r0 + rt*16 + li
  const bf16_t* r1p = (MODE == 0) ? qp : kp;
  const int64_t r1s = rs;
  const bf16_t* r2p = (MODE == 0) ? dop : vp;
  const int64_t r2s = (MODE == 0) ? ro : rs;
  bf16x8_t r1[2][KS], r2[2][KS];
  float lse2[2] = {0.f, 0.f}, dl[2] = {0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int r = r0 + rt * 16 + li;
    float dot = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 a = make_uint4(0, 0, 0, 0), bq = make_uint4(0, 0, 0, 0);
      if (r < N) {
        a = *reinterpret_cast<const uint4*>(r1p + (int64_t)r * r1s + ks * 32 + g * 8);
        bq = *reinterpret_cast<const uint4*>(r2p + (int64_t)r * r2s + ks * 32 + g * 8);
        if (MODE == 0) {  // delta_i = dO_i . O_i
          float dv[8], ov[8];
          ld8<bf16_t>(dop + (int64_t)r * ro + ks * 32 + g * 8, dv);
          ld8<bf16_t>(op + (int64_t)r * ro + ks * 32 + g * 8, ov);
#pragma unroll
          for (int e = 0; e < 8; ++e) dot += dv[e] * ov[e];
        }
      }
      r1[rt][ks] = __builtin_bit_cast(bf16x8_t, a);
      r2[rt][ks] = __builtin_bit_cast(bf16x8_t, bq);
    }
    if (MODE == 0) {
      dot += __shfl_xor(dot, 16, 64);
      dot += __shfl_xor(dot, 32, 64);
      dl[rt] = dot;
      lse2[rt] = (r < N) ? lsep[r] * 1.4426950408889634f : 0.f;
    }
  }

  constexpr int NACC = (MODE == 0) ? 1 : 2;
  f32x4_t acc[NACC][DT][2];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { acc[a][dt][0] = f32x4_t{0, 0, 0, 0}; acc[a][dt][1] = f32x4_t{0, 0, 0, 0}; }

  // ---- streamed tiles: T1 (K | Q), T2 (V | dO)
  const bf16_t* t1p = (MODE == 0) ? kp : qp;
  const bf16_t* t2p = (MODE == 0) ? vp : dop;
  const int64_t t2s = (MODE == 0) ? rs : ro;
  const int ntiles = (N + FA_BKV - 1) / FA_BKV;
  uint4 t1reg[PASSES], t2reg[PASSES], oreg[MODE == 1 ? PASSES : 1];
  float streg = 0.f;
  auto gload = [&](int t) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int cidx = tid + 256 * ps;
      const int row = t * FA_BKV + cidx / CPR, cc = cidx % CPR;
      t1reg[ps] = make_uint4(0, 0, 0, 0);
      t2reg[ps] = make_uint4(0, 0, 0, 0);
      if constexpr (MODE == 1) oreg[ps] = make_uint4(0, 0, 0, 0);
      if (row < N) {
        t1reg[ps] = *reinterpret_cast<const uint4*>(t1p + (int64_t)row * rs + cc * 8);
        t2reg[ps] = *reinterpret_cast<const uint4*>(t2p + (int64_t)row * t2s + cc * 8);
        if constexpr (MODE == 1) oreg[ps] = *reinterpret_cast<const uint4*>(op + (int64_t)row * ro + cc * 8);
      }
    }
    if (MODE == 1 && tid < 64) {
      const int row = t * FA_BKV + tid;
      streg = (row < N) ? lsep[row] * 1.4426950408889634f : 0.f;
    }
  };
  auto lstore = [&](int stage) {
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int cidx = tid + 256 * ps;
      const int off = (cidx / CPR) * PITCH + (cidx % CPR) * 16;
      *reinterpret_cast<uint4*>(base + off) = t1reg[ps];
      *reinterpret_cast<uint4*>(base + TILE + off) = t2reg[ps];
      if constexpr (MODE == 1) {   // delta of the streamed rows: dO . O, the row's CPR chunk holders are consecutive lanes
        const uint32_t* dw = reinterpret_cast<const uint32_t*>(&t2reg[ps]);
        const uint32_t* ow = reinterpret_cast<const uint32_t*>(&oreg[ps]);
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dlo, dhi, olo, ohi;
          unpack2h(dw[e], dlo, dhi);
          unpack2h(ow[e], olo, ohi);
          dot = __builtin_fmaf(dlo, olo, dot);
          dot = __builtin_fmaf(dhi, ohi, dot);
        }
        dot += dpp_mov<0xB1>(dot);                          // lanes l ^ 1
        dot += dpp_mov<0x4E>(dot);                          // lanes l ^ 2
        if (CPR == 8) dot += dpp_mov<0x141>(dot);           // the other quad of the row's 8 lanes
        if ((cidx % CPR) == 0) reinterpret_cast<float*>(base + 2 * TILE)[64 + cidx / CPR] = dot;
      }
    }
    if (MODE == 1 && tid < 64) reinterpret_cast<float*>(base + 2 * TILE)[tid] = streg;
  };

  gload(0);
  lstore(0);
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;

  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) gload(t + 1);
    const char* T1 = smem + (t & 1) * STAGE;
    const char* T2 = T1 + TILE;
    const float* ST = reinterpret_cast<const float*>(T1 + 2 * TILE);

    if (active) {   // (a wave whose 32 resident rows all lie beyond N -- N = 288: three of twelve -- only helps staging)
    // ---- S-type products: x1 = T1 R1^T (scores), x2 = T2 R2^T (dP)
    f32x4_t x1[4][2], x2[4][2];
#pragma unroll
    for (int st = 0; st < 4; ++st) { x1[st][0] = x1[st][1] = x2[st][0] = x2[st][1] = f32x4_t{0, 0, 0, 0}; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(T1 + (st * 16 + li) * PITCH + (ks * 32 + g * 8) * 2);
        const bf16x8_t a2 = *reinterpret_cast<const bf16x8_t*>(T2 + (st * 16 + li) * PITCH + (ks * 32 + g * 8) * 2);
        x1[st][0] = COUNTR_MFMA_16X16X32(a1, r1[0][ks], x1[st][0], 0, 0, 0);
        x1[st][1] = COUNTR_MFMA_16X16X32(a1, r1[1][ks], x1[st][1], 0, 0, 0);
        x2[st][0] = COUNTR_MFMA_16X16X32(a2, r2[0][ks], x2[st][0], 0, 0, 0);
        x2[st][1] = COUNTR_MFMA_16X16X32(a2, r2[1][ks], x2[st][1], 0, 0, 0);
      }

    // ---- P and dS (x1 <- P, x2 <- dS); streamed index of element (st, reg) is t*64 + st*16 + g*4 + reg
    const bool ragged = RAGGED && (t + 1) * FA_BKV > N;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      float sl[4] = {0.f, 0.f, 0.f, 0.f}, sd[4] = {0.f, 0.f, 0.f, 0.f};
      if (MODE == 1) {
        ld4<float>(ST + st * 16 + g * 4, sl);
        ld4<float>(ST + 64 + st * 16 + g * 4, sd);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float l2 = (MODE == 0) ? lse2[rt] : sl[r];
          const float dd = (MODE == 0) ? dl[rt] : sd[r];
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(x1[st][rt][r], c, -l2));
          if (ragged && (t * FA_BKV + st * 16 + g * 4 + r >= N)) p = 0.f;
          x1[st][rt][r] = p;
          x2[st][rt][r] = p * (x2[st][rt][r] - dd);
        }
    }
    bf16x8_t pf[2][2], dsf[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const uint4 pk = make_uint4(pack2bf(x2[2 * ps][rt][0], x2[2 * ps][rt][1]), pack2bf(x2[2 * ps][rt][2], x2[2 * ps][rt][3]),
                                    pack2bf(x2[2 * ps + 1][rt][0], x2[2 * ps + 1][rt][1]),
                                    pack2bf(x2[2 * ps + 1][rt][2], x2[2 * ps + 1][rt][3]));
        dsf[rt][ps] = __builtin_bit_cast(bf16x8_t, pk);
        if (MODE == 1) {
          const uint4 pp = make_uint4(pack2bf(x1[2 * ps][rt][0], x1[2 * ps][rt][1]), pack2bf(x1[2 * ps][rt][2], x1[2 * ps][rt][3]),
                                      pack2bf(x1[2 * ps + 1][rt][0], x1[2 * ps + 1][rt][1]),
                                      pack2bf(x1[2 * ps + 1][rt][2], x1[2 * ps + 1][rt][3]));
          pf[rt][ps] = __builtin_bit_cast(bf16x8_t, pp);
        }
      }

    // ---- PV-type accumulations: acc0^T += T1^T dS  (dQ^T | dK^T),  MODE 1 also acc1^T += T2^T P (dV^T)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (ps * 32 + 4 * g + (li >> 2)) * PITCH + (dt * 16 + (li & 3) * 4) * 2;
        {
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(T1 + off));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(T1 + off + 16 * PITCH));
          s16x8_t vv;
          vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3]; vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
          const bf16x8_t tf = __builtin_bit_cast(bf16x8_t, vv);
          acc[0][dt][0] = COUNTR_MFMA_16X16X32(tf, dsf[0][ps], acc[0][dt][0], 0, 0, 0);
          acc[0][dt][1] = COUNTR_MFMA_16X16X32(tf, dsf[1][ps], acc[0][dt][1], 0, 0, 0);
        }
        if (MODE == 1) {
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(T2 + off));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(T2 + off + 16 * PITCH));
          s16x8_t vv;
          vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3]; vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
          const bf16x8_t tf = __builtin_bit_cast(bf16x8_t, vv);
          acc[NACC - 1][dt][0] = COUNTR_MFMA_16X16X32(tf, pf[0][ps], acc[NACC - 1][dt][0], 0, 0, 0);
          acc[NACC - 1][dt][1] = COUNTR_MFMA_16X16X32(tf, pf[1][ps], acc[NACC - 1][dt][1], 0, 0, 0);
        }
      }

    }   // active
    if (more) lstore((t + 1) & 1);
    __syncthreads();
  }
  if (!active) return;      // (no barrier behind this point)

  // ---- epilogue: lane (li, g) owns resident row r0 + rt*16 + li and channels dt*16 + 4g .. +3; as in the forward the wave's
  // [32][DH] block goes through its slice of the (now idle) tile ring and out as whole rows in 16-byte chunks
  char* ost = smem + wave * 32 * PITCH;
  auto emit = [&](const f32x4_t (&a)[DT][2], float sc, int64_t coloff) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        *reinterpret_cast<uint2*>(ost + (rt * 16 + li) * PITCH + (dt * 16 + g * 4) * 2) =
            make_uint2(pack2bf(a[dt][rt][0] * sc, a[dt][rt][1] * sc), pack2bf(a[dt][rt][2] * sc, a[dt][rt][3] * sc));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < (32 * CPR) / 64; ++j) {
      const int idx = lane + 64 * j, r = idx / CPR, cc = idx % CPR;
      const uint4 v = *reinterpret_cast<const uint4*>(ost + r * PITCH + cc * 16);
      if (r0 + r < N) *reinterpret_cast<uint4*>(dqkv + ((int64_t)b * N + r0 + r) * rs + h * DH + coloff + cc * 8) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  if (MODE == 0) {
    emit(acc[0], scale, 0);                      // dQ
  } else {
    emit(acc[0], scale, (int64_t)H * DH);        // dK
    emit(acc[NACC - 1], 1.f, (int64_t)2 * H * DH);   // dV
  }
}

// ONE launch for both passes: the first half of the grid runs the dK / dV pass (the longer one: two accumulators), the second half the dQ
// pass.  The dK / dV pass recomputes delta for the rows it streams (an extra read of the O tile), so the passes do not depend on
// each other: at N = 288 (MAE encoder, 8 images) a pass is 288 workgroups of five short tiles -- two back-to-back launches were two
// latency-bound rounds on half of the chip's slots.
template <int DH, bool RAGGED>
__global__ __launch_bounds__(256, 2) void flash_attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ outp,
                                                             const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dqkv, int N, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int half = gridDim.x >> 1;
  if ((int)blockIdx.x < half) fa_bwd_body<DH, 1, RAGGED>(smem, blockIdx.x, half, qkv, outp, dout, lse, dqkv, N, H, scale);
  else fa_bwd_body<DH, 0, RAGGED>(smem, blockIdx.x - half, half, qkv, outp, dout, lse, dqkv, N, H, scale);
}

template <int DH>
int launch_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int N,
                    int H, float scale, hipStream_t s) {
  const int rblocks = (N + FA_BQ - 1) / FA_BQ;
  dim3 grid(2 * B * H * rblocks), block(256);
  const size_t lds = 2 * (2 * FA_BKV * fa_pitch(DH) + 512);
  (void)delta;   // (the passes no longer exchange delta through memory; the argument stays in the ABI)
  if (N % FA_BKV)
    hipLaunchKernelGGL((flash_attn_bwd_kernel<DH, true>), grid, block, lds, s, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, scale);
  else
    hipLaunchKernelGGL((flash_attn_bwd_kernel<DH, false>), grid, block, lds, s, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, scale);
  COUNTR_LAUNCH_CHECK("countr_attn_bwd");
}

}  // namespace

// Backward of countr_attn_fwd.  qkv, out, lse as given to / produced by the forward; dout bf16 [B, N, H*dh];
// delta: fp32 [B, H, N], unused since both passes compute it (kept in the ABI); dqkv: bf16 [B, N, 3, H, dh] (fully overwritten).
extern "C" int countr_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                               int B, int N, int H, int dh, float scale, void* stream) {
  if (!qkv || !out || !dout || !lse || !delta || !dqkv || B <= 0 || N <= 0 || H <= 0) { countr_set_error("countr_attn_bwd: bad args"); return -1; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dh == 64) return launch_attn_bwd<64>(qkv, out, dout, lse, delta, dqkv, B, N, H, scale, s);
  if (dh == 32) return launch_attn_bwd<32>(qkv, out, dout, lse, delta, dqkv, B, N, H, scale, s);
  countr_set_error("countr_attn_bwd: head_dim must be 32 or 64");
  return -1;
}
