// Fused self-attention forward for gfx950 (bf16 in/out, fp32 softmax + accumulation).
// Reference: Attention.forward, models_crossvit.py:82-94 (== timm Attention): softmax(q k^T * dh^-0.5) v on a
// packed qkv [B, N, 3, H, dh]; output [B, N, H*dh].  N = 576, dh = 64 (encoder) or 32 (decoder).
//
// Structure (one workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows):
//   * K/V tiles of 64 keys are register-staged into double-buffered LDS (one barrier per tile).
//   * S^T = K Q^T with v_mfma_f32_16x16x32_bf16 (A = K fragment from LDS, B = Q fragment held in VGPRs), so a
//     lane holds 4 keys x 1 query row per tile: row max / sum need only two xor-shuffles (lanes l, l^16, l^32, l^48).
//   * online softmax in exp2 domain: p = exp2(s * c - m), c = scale * log2(e) folded into one FMA.
//   * O^T += V^T P^T: the P operand is built from the lane's own S^T registers by choosing the MFMA k-slot
//     order kappa(g, e) = {4g..4g+3, 16+4g..16+4g+3} (sum over keys is order independent) and V^T fragments are
//     read in the same key order with ds_read_b64_tr_b16 from the row-major V tile: no cross-lane traffic for P.
#include "common.cuh"
#include "../../include/countr_hip.h"

namespace {

constexpr int FA_BQ = 128;   // query rows per workgroup
constexpr int FA_BKV = 64;   // keys per tile

template <int DH>
__global__ __launch_bounds__(256) void flash_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                             float* __restrict__ lse, int N, int H, float c /* scale*log2e */) {
  constexpr int KS = DH / 32;          // k-steps over head dim for QK^T
  constexpr int DT = DH / 16;          // 16-wide tiles of the head dim for O
  constexpr int PITCH = DH * 2 + 16;   // bytes per LDS row
  constexpr int TILE = FA_BKV * PITCH;
  constexpr int CPR = DH / 8;          // 16-byte chunks per row
  constexpr int PASSES = (FA_BKV * CPR) / 256;  // staging passes (2 for dh=64, 1 for dh=32)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // stage s: K at s*2*TILE, V behind it

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int qblocks = (N + FA_BQ - 1) / FA_BQ;
  // XCD-aware mapping: workgroup id b runs on XCD b % 8 (observed dispatch order; affects speed only).  All query
  // blocks of one (batch, head) are given ids congruent mod 8 so that its K/V tiles are fetched into ONE XCD's L2
  // instead of once per query block (PMC: 78 MB -> 21 MB fabric reads per launch at B=8, H=12).
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
  const int64_t rs = (int64_t)3 * H * DH;  // row stride (elements) of the packed qkv
  const bf16_t* qp = qkv + (int64_t)b * N * rs + h * DH;
  const bf16_t* kp = qp + H * DH;
  const bf16_t* vp = kp + H * DH;
  const int q0 = qb * FA_BQ + wave * 32;

  bf16x8_t qf[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + qt * 16 + li;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < N) v = *reinterpret_cast<const uint4*>(qp + (int64_t)q * rs + ks * 32 + g * 8);
      qf[qt][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
  }

  f32x4_t o[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { o[dt][0] = f32x4_t{0, 0, 0, 0}; o[dt][1] = f32x4_t{0, 0, 0, 0}; }
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};

  const int ntiles = (N + FA_BKV - 1) / FA_BKV;
  uint4 kreg[PASSES], vreg[PASSES];
  auto gload = [&](int t) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int cidx = tid + 256 * ps;
      const int key = t * FA_BKV + cidx / CPR, cc = cidx % CPR;
      kreg[ps] = make_uint4(0, 0, 0, 0);
      vreg[ps] = make_uint4(0, 0, 0, 0);
      if (key < N) {
        kreg[ps] = *reinterpret_cast<const uint4*>(kp + (int64_t)key * rs + cc * 8);
        vreg[ps] = *reinterpret_cast<const uint4*>(vp + (int64_t)key * rs + cc * 8);
      }
    }
  };
  auto lstore = [&](int stage) {
    char* ks_ = smem + stage * 2 * TILE;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int cidx = tid + 256 * ps;
      const int off = (cidx / CPR) * PITCH + (cidx % CPR) * 16;
      *reinterpret_cast<uint4*>(ks_ + off) = kreg[ps];
      *reinterpret_cast<uint4*>(ks_ + TILE + off) = vreg[ps];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) gload(t + 1);
    const char* Ks = smem + (t & 1) * 2 * TILE;
    const char* Vs = Ks + TILE;

    // ---- S^T[kt][qt] : 16 keys x 16 queries per MFMA tile
    f32x4_t s[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) { s[kt][0] = f32x4_t{0, 0, 0, 0}; s[kt][1] = f32x4_t{0, 0, 0, 0}; }

#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (kt * 16 + li) * PITCH + (ks * 32 + g * 8) * 2);
        s[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[kt][0], 0, 0, 0);
        s[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[kt][1], 0, 0, 0);
      }

    if ((t + 1) * FA_BKV > N) {  // ragged last tile: keys >= N do not exist
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t * FA_BKV + kt * 16 + g * 4 + r >= N) { s[kt][0][r] = -INFINITY; s[kt][1][r] = -INFINITY; }
    }

    // ---- online softmax (exp2 domain), P packed straight into PV operands
    bf16x8_t pf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // deferred rescale: while the running max grows by <= 8 (log2 units) keep the old reference max -- P stays
      // <= 2^8, exact in the fp32 row sum and well inside bf16 range -- and skip the O / l rescale for this tile.
      const float mloc = mx * c;
      float mref = m[qt];
      if (!__all(mloc - mref <= 8.0f)) {
        const float mnew = fmaxf(mref, mloc);
        const float alpha = __builtin_amdgcn_exp2f(mref - mnew);
        m[qt] = mnew;
        mref = mnew;
        l[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          o[dt][qt][0] *= alpha; o[dt][qt][1] *= alpha; o[dt][qt][2] *= alpha; o[dt][qt][3] *= alpha;
        }
      }
      float rsum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][r], c, -mref));
          s[kt][qt][r] = p;
          rsum += p;
        }
      l[qt] += rsum;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const uint4 pk = make_uint4(pack2bf(s[2 * ps][qt][0], s[2 * ps][qt][1]), pack2bf(s[2 * ps][qt][2], s[2 * ps][qt][3]),
                                    pack2bf(s[2 * ps + 1][qt][0], s[2 * ps + 1][qt][1]),
                                    pack2bf(s[2 * ps + 1][qt][2], s[2 * ps + 1][qt][3]));
        pf[qt][ps] = __builtin_bit_cast(bf16x8_t, pk);
      }
    }

    // ---- O^T[dt][qt] += V^T P^T  (k-slot order kappa(g,e) = {4g+e, 16+4g+e})
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;

#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const char* a0 = Vs + (ps * 32 + 4 * g + (li >> 2)) * PITCH + (dt * 16 + (li & 3) * 4) * 2;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)a0);
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a0 + 16 * PITCH));
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        s16x8_t vv;
        vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3];
        vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vv);
        o[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][ps], o[dt][0], 0, 0, 0);
        o[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][ps], o[dt][1], 0, 0, 0);
      }


    if (more) lstore((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: normalise and store; lane (li, g) owns query q and channels dt*16 + 4g .. +3
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const int q = q0 + qt * 16 + li;
    if (q >= N) continue;
    const float inv = 1.f / lt;
    bf16_t* orow = out + ((int64_t)b * N + q) * (H * DH) + h * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const float v[4] = {o[dt][qt][0] * inv, o[dt][qt][1] * inv, o[dt][qt][2] * inv, o[dt][qt][3] * inv};
      st4<bf16_t>(orow + dt * 16 + g * 4, v);
    }
    if (lse && g == 0) lse[((int64_t)b * H + h) * N + q] = (m[qt] + log2f(lt)) * 0.6931471805599453f;
  }
}

}  // namespace

// qkv: bf16 [B, N, 3, H, dh] packed (row stride 3*H*dh); out: bf16 [B, N, H*dh]; lse: optional fp32 [B, H, N]
// (natural-log sum-exp of the scaled scores, kept for a fused backward).  dh must be 32 or 64.
extern "C" int countr_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int dh, float scale, void* stream) {
  if (!qkv || !out || B <= 0 || N <= 0 || H <= 0) { countr_set_error("countr_attn_fwd: bad args"); return -1; }
  const float c = scale * 1.4426950408889634f;
  const int qblocks = (N + FA_BQ - 1) / FA_BQ;
  dim3 grid(B * H * qblocks), block(256);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dh == 64) {
    hipLaunchKernelGGL(flash_attn_fwd_kernel<64>, grid, block, 4 * FA_BKV * (64 * 2 + 16), s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, c);
  } else if (dh == 32) {
    hipLaunchKernelGGL(flash_attn_fwd_kernel<32>, grid, block, 4 * FA_BKV * (32 * 2 + 16), s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, c);
  } else {
    countr_set_error("countr_attn_fwd: head_dim must be 32 or 64");
    return -1;
  }
  COUNTR_LAUNCH_CHECK("countr_attn_fwd");
}
