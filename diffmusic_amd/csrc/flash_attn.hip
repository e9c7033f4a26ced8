// Fused multi-head attention forward (no materialised score matrix) for the U-Net transformer blocks.
//
// The reference reaches this through diffusers' Attention processor inside UNet2DConditionModel
// (diffmusic/pipelines/pipeline_musicldm.py:696-703 calls the U-Net twice per step on the CFG batch); the
// U-Net is never differentiated on the hot path (scheduling_dps.py:195-212 differentiates the decoder
// only), so a forward-only kernel is enough here.  The VAE mid-block attention needs its probabilities
// for the backward pass and stays on the GEMM path (blocks.h attention_core with P_keep).
//
// o[b, i, h*dh:(h+1)*dh] = softmax_j(q_i . k_j * scale + colbias[b, j]) v_j
//
// One workgroup = 4 waves, each owning QT tiles of 16 queries of one (batch, head) (QT = 2: 128 queries per workgroup; QT = 1: 64, taken
// when the grid would not fill the chip).  Keys are walked in blocks of 64.  Everything is computed transposed so that a lane owns ONE
// query column:
//   S^T = K Q^T   (MFMA A = K fragment from LDS, B = Q fragment held in registers)
//          -> lane (lr, lq) holds S^T[key 16t+4lq+e][query lr]: the row statistics of a query live in one lane
//             (+ the three lanes lr+16, lr+32, lr+48: two row swaps per block for the running maximum),
//   O^T += V^T P^T (MFMA A = V^T fragment, B = the exponentiated accumulators re-used in place:
//             the k slots of a 32-key MFMA step are defined as {16t0+4lq+j} u {16t1+4lq+j}, which is exactly what
//             the S^T accumulators of two key tiles hold, so P never moves between lanes or through LDS).
// K and V tiles sit row-major [key][d] in LDS (global -> registers -> one ds_write_b128 per 16-byte chunk); the V^T fragment comes out of
// ds_read_b64_tr_b16, whose 4-key x 16-column gather per 16-lane group delivers exactly the k slots above.
// exp2 with the scale folded into one FMA; running max / sum per lane (online softmax).  The loop is VALU-bound
// (per 64 keys and query tile: 16 exp at 8 issue cycles, ~50 plain VALU at 4); the matrix work is 16 MFMAs per 64 keys at dh = 32, QT = 2.
#include "dmx_common.h"
#include "kernels.h"
#include "conv_pair.h"

namespace {

constexpr int FA_KB = 64;     // keys per block
// queries: QT tiles of 16 per wave, 4 waves per workgroup (QT = 2: 128 queries per workgroup; QT = 1: 64 -- twice the workgroups when the
// grid would leave the chip half empty, at twice the K / V fragment reads per query)

template <int DK, int DT>     // DK: 32-wide k steps of q.k (dh <= 32*DK); DT: 16-wide tiles of the output dim (dh <= 16*DT)
struct FaCfg {
  static constexpr int KPITCH = DK * 64 + 32;          // bytes per key row of the K tile (+32: conflict-free ds_read_b128)
  // V tile: row-major [key][d] like K (one ds_write_b128 per chunk); the V^T fragments come out of ds_read_b64_tr_b16.  Pitch = 8 dwords
  // mod 16: the 8 key rows x 32 bytes that one 32-lane half gathers fall into 64 distinct banks
  static constexpr int VPITCH = DT * 32 + (DT % 2 == 0 ? 32 : 0);
  static constexpr int KBYTES = FA_KB * KPITCH, VBYTES = FA_KB * VPITCH;
  static constexpr int KCH = FA_KB * DK * 4, VCH = DT * 16 * 8;      // 16-byte chunks per tile
  static constexpr int KIT = (KCH + 255) / 256, VIT = (VCH + 255) / 256;
  static constexpr int LDS = 2 * (KBYTES + VBYTES);
};

struct FaParams {
  const act_t* q; const act_t* k; const act_t* v; act_t* o;
  const float* colbias;
  int Nq, Nk, C, heads, dh, ldq, ldk, ldv;      // C = row stride of o; ldq / ldk / ldv = row strides of q / k / v (a fused QKV buffer has 3C)
  float c;                    // scale * log2(e)
};

typedef unsigned fa_u2 __attribute__((ext_vector_type(2)));
typedef short fa_s4 __attribute__((ext_vector_type(4)));
typedef fa_s4 __attribute__((address_space(3))) fa_lds_s4;
// NOTE this file is compiled with -fno-honor-nans -fno-slp-vectorize (build.py FILE_FLAGS): fmaxf on MFMA results then needs no quieting
// v_max_f32 x, x in front (15 extra instructions per query tile and block) and nested fmaxf becomes v_max3_f32; and the row sums / rescales stay
// one-value VALU -- beside MFMAs a v_pk_add_f32 / v_pk_mul_f32 costs more issue time than the two plain instructions it replaces
// (MI355X_MICROARCH.md, per-instruction constants).  Plain C on purpose, no inline asm: the compiler's hazard recogniser has to see every
// instruction that reads an MFMA result.
__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float vmax2(float a, float b) { return fmaxf(a, b); }
// maximum over the lanes {l, l ^ 16, l ^ 32, l ^ 48}
__device__ __forceinline__ float rows_max(float v) {
  fa_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// (dh = 32: four waves per SIMD -- 128 registers -- hold the 1024 workgroups of the level-1 U-Net attention in ONE round of the chip; at the
//  160 registers the compiler takes when left alone three fit and a third of the grid runs as a second round)
template <int DK, int DT, int QT>
__global__ __launch_bounds__(256, (DK == 1 ? 4 : 2)) void flash_attn_fwd_kernel(const FaParams P) {
  using F = FaCfg<DK, DT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = blockIdx.y, b = z / P.heads, h = z - b * P.heads;
  const int q0 = (blockIdx.x * 4 + wave) * (16 * QT);
  const int dh = P.dh, C = P.C, Nk = P.Nk;
  const int ldq = P.ldq, ldk = P.ldk;
  const act_t* qb = P.q + (long long)b * P.Nq * ldq + h * dh;
  const act_t* kb = P.k + (long long)b * Nk * ldk + h * dh;
  const act_t* vb = P.v + (long long)b * Nk * P.ldv + h * dh;
  const float NEG = -__builtin_huge_valf();

  // ---- Q fragments (B operand): lane (lr, lq) holds q[query 16*qt + lr][32*ks + 8*lq .. +8]
  frag8_t qf[QT][DK];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) {
      const int qi = q0 + qt * 16 + lr, d = ks * 32 + lq * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qi < P.Nq && d < dh) v = *reinterpret_cast<const uint4*>(qb + (long long)qi * ldq + d);
      qf[qt][ks] = __builtin_bit_cast(frag8_t, v);
    }

  // ---- K / V^T tiles: global -> registers -> LDS, double buffered
  uint4 kreg[F::KIT], vreg[F::VIT];
  auto gload = [&](int blk) {
    const int key0 = blk * FA_KB;
#pragma unroll
    for (int it = 0; it < F::KIT; ++it) {
      const int c = tid + it * 256, row = c / (DK * 4), d = (c % (DK * 4)) * 8;
      kreg[it] = make_uint4(0, 0, 0, 0);
      if (c < F::KCH && key0 + row < Nk && d < dh) kreg[it] = *reinterpret_cast<const uint4*>(kb + (long long)(key0 + row) * ldk + d);
    }
#pragma unroll
    for (int it = 0; it < F::VIT; ++it) {
      // V arrives row-major (key, d) like K and stays so in LDS: chunk c = 8 d-values of one key
      const int c = tid + it * 256, kl = c / (2 * DT), d = (c - kl * (2 * DT)) * 8;
      vreg[it] = make_uint4(0, 0, 0, 0);
      if (c < F::VCH && d < dh && key0 + kl < Nk) vreg[it] = *reinterpret_cast<const uint4*>(vb + (long long)(key0 + kl) * P.ldv + d);
    }
  };
  auto lstore = [&](int buf) {
    char* kt = smem + buf * (F::KBYTES + F::VBYTES);
    char* vt = kt + F::KBYTES;
#pragma unroll
    for (int it = 0; it < F::KIT; ++it) {
      const int c = tid + it * 256;
      if (c < F::KCH) *reinterpret_cast<uint4*>(kt + (c / (DK * 4)) * F::KPITCH + (c % (DK * 4)) * 16) = kreg[it];
    }
#pragma unroll
    for (int it = 0; it < F::VIT; ++it) {
      const int c = tid + it * 256, kl = c / (2 * DT), d = (c - kl * (2 * DT)) * 8;
      if (c < F::VCH) *reinterpret_cast<uint4*>(vt + kl * F::VPITCH + d * 2) = vreg[it];     // keys past Nk and d >= dh are zeros
    }
  };

  f32x4 oacc[DT][QT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) oacc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) m_run[qt] = NEG, l_run[qt] = 0.f;

  const int nblk = (Nk + FA_KB - 1) / FA_KB;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    const int buf = blk & 1;
    if (blk + 1 < nblk) gload(blk + 1);
    const char* kt = smem + buf * (F::KBYTES + F::VBYTES);
    const char* vt = kt + F::KBYTES;
    // ---- S^T = K Q^T for 4 key tiles x 2 query tiles
    f32x4 s[4][QT];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      frag8_t kf[DK];
#pragma unroll
      for (int ks = 0; ks < DK; ++ks) kf[ks] = *reinterpret_cast<const frag8_t*>(kt + (t * 16 + lr) * F::KPITCH + ks * 64 + lq * 16);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) a = DMX_MFMA16(kf[ks], qf[qt][ks], a);
        s[t][qt] = a;
      }
    }
    // ---- logits in the log2 domain, key mask / additive key bias
    const int key0 = blk * FA_KB;
    const float c = P.c;
    const bool tail = key0 + FA_KB > Nk;
    // generic blocks (bias or key tail) scale the logits here; full unbiased blocks keep them raw and fold the scale into the one
    // FMA in front of the exponential (c > 0: max(c * s) = c * max(s)) -- 32 multiplies fewer per block in a VALU-bound loop
    const bool raw = !(P.colbias || tail);
    if (!raw) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int kbase = key0 + t * 16 + lq * 4;
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (P.colbias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bb[e] = (kbase + e < Nk) ? P.colbias[(long long)b * Nk + kbase + e] * 1.4426950408889634f : 0.f;
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int e = 0; e < 4; ++e) s[t][qt][e] = (kbase + e < Nk) ? __builtin_fmaf(s[t][qt][e], c, bb[e]) : NEG;
      }
    }
    const float cs = raw ? c : 1.f;                    // what is still to be applied to the stored logits
    // ---- online softmax per query (= per lane column); the four lanes of a query share the maximum.  The arithmetic runs on
    // 4-vectors so that the compiler can use the packed fp32 forms (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32)
    frag8_t pf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      // 16 logits -> 8 v_max3 / v_max, then the four lanes of the query through the row swaps of gfx950 (v_permlane16_swap /
      // v_permlane32_swap: VALU, no trip through the LDS pipe)
      float mx = vmax3(s[0][qt][0], s[0][qt][1], s[0][qt][2]);
      mx = vmax3(mx, s[0][qt][3], s[1][qt][0]);
      mx = vmax3(mx, s[1][qt][1], s[1][qt][2]);
      mx = vmax3(mx, s[1][qt][3], s[2][qt][0]);
      mx = vmax3(mx, s[2][qt][1], s[2][qt][2]);
      mx = vmax3(mx, s[2][qt][3], s[3][qt][0]);
      mx = vmax3(mx, s[3][qt][1], s[3][qt][2]);
      mx = rows_max(vmax2(mx, s[3][qt][3]));
      const float m_new = fmaxf(m_run[qt], mx * cs);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);     // first block: exp2(-inf) = 0
      m_run[qt] = m_new;
      float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][qt][e], cs, -m_new));     // masked keys: -inf * 1 - m = -inf -> p = 0
          s[t][qt][e] = pv;
          sum[e] = t == 0 ? pv : sum[e] + pv;
        }
      l_run[qt] = __builtin_fmaf(l_run[qt], alpha, (sum[0] + sum[1]) + (sum[2] + sum[3]));
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) oacc[dt][qt][e] *= alpha;
      // P^T fragments: k slots (lq, j) of MFMA step kk <-> keys 32kk + {4lq + j, 16 + 4lq + j}
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const uint4 u = make_uint4(pack2a(s[2 * kk][qt][0], s[2 * kk][qt][1]), pack2a(s[2 * kk][qt][2], s[2 * kk][qt][3]),
                                   pack2a(s[2 * kk + 1][qt][0], s[2 * kk + 1][qt][1]), pack2a(s[2 * kk + 1][qt][2], s[2 * kk + 1][qt][3]));
        pf[qt][kk] = __builtin_bit_cast(frag8_t, u);
      }
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        // transposed read: the 16 lanes of group lq gather keys 32kk + 4lq .. + 3 (lane 4q + p: key row q, columns 4p .. 4p + 3 of this dt)
        // and lane lr receives column 16dt + lr of those four keys -- the k slots (lq, j) of this MFMA step; second half: 16 keys on
        const char* vrow = vt + (kk * 32 + lq * 4 + (lr >> 2)) * F::VPITCH + (dt * 16 + (lr & 3) * 4) * 2;
        const fa_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_s4*)vrow);
        const fa_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((fa_lds_s4*)(vrow + 16 * F::VPITCH));
        const frag8_t vf = __builtin_bit_cast(frag8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) oacc[dt][qt] = DMX_MFMA16(vf, pf[qt][kk], oacc[dt][qt]);
      }
    if (blk + 1 < nblk) lstore(buf ^ 1);     // the other buffer was last read before the previous barrier
    __syncthreads();
  }

  // ---- normalise and store: lane holds o[query 16qt + lr][16dt + 4lq + e]
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = l_run[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    const int qi = q0 + qt * 16 + lr;
    if (qi >= P.Nq) continue;
    act_t* orow = P.o + ((long long)b * P.Nq + qi) * C + h * dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + lq * 4;
      if (d < dh)
        *reinterpret_cast<uint2*>(orow + d) = make_uint2(pack2a(oacc[dt][qt][0] * inv, oacc[dt][qt][1] * inv),
                                                         pack2a(oacc[dt][qt][2] * inv, oacc[dt][qt][3] * inv));
    }
  }
}

template <int DK, int DT>
int launch_fa(const FaParams& P, int Z, hipStream_t st) {
  using F = FaCfg<DK, DT>;
  static const int force_qt = [] { const char* e = getenv("DMX_FLASH_QT"); return e ? atoi(e) : 0; }();
  // at most one workgroup per CU at 128 queries each (or half-empty workgroups): halve the query tile.  Measured, 2B = 16 / 8 U-Net shapes:
  // 256 workgroups 27.7 -> 25.0 us, 96 workgroups 8.4 -> 6.6 us, but 384 workgroups 9.0 -> 11.5 us
  const bool small = force_qt ? force_qt == 1 : ((long long)cdiv(P.Nq, 128) * Z <= 256 || P.Nq <= 64);
  if (small) {
    dim3 grid((unsigned)cdiv(P.Nq, 64), (unsigned)Z, 1);
    hipLaunchKernelGGL((flash_attn_fwd_kernel<DK, DT, 1>), grid, dim3(256), F::LDS, st, P);
  } else {
    dim3 grid((unsigned)cdiv(P.Nq, 128), (unsigned)Z, 1);
    hipLaunchKernelGGL((flash_attn_fwd_kernel<DK, DT, 2>), grid, dim3(256), F::LDS, st, P);
  }
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}

}  // namespace

bool dmx_flash_attn_ok(int dh, int C) { return dh >= 8 && dh <= 96 && (dh & 7) == 0 && (C & 7) == 0; }

// q (B, Nq, ldq), k (B, Nk, ldk), v (B, Nk, ldv) channels-last (row strides >= C; 0 = C) with `heads` heads of dh = C / heads; o (B, Nq, C).
// colbias: optional additive key bias (B, Nk) fp32.  V is transposed by the LDS read (ds_read_b64_tr_b16): no V^T tensor, no transpose launch.
int dmx_flash_attn_fwd(const act_t* q, const act_t* k, const act_t* v, act_t* o, const float* colbias, int B, int Nq, int Nk,
                       int C, int heads, float scale, hipStream_t st, int ldq, int ldk, int ldv) {
  const int dh = C / heads;
  if (ldq <= 0) ldq = C;
  if (ldk <= 0) ldk = C;
  if (ldv <= 0) ldv = C;
  if (!dmx_flash_attn_ok(dh, C) || Nq < 1 || Nk < 1 || (ldq & 7) || (ldk & 7) || (ldv & 7)) return DMX_ERR_SHAPE;
  FaParams P;
  P.q = q; P.k = k; P.v = v; P.o = o; P.colbias = colbias;
  P.Nq = Nq; P.Nk = Nk; P.C = C; P.heads = heads; P.dh = dh; P.ldq = ldq; P.ldk = ldk; P.ldv = ldv;
  P.c = scale * 1.4426950408889634f;
  const int Z = B * heads;
  const int rec = dmx_prof_open(st);
  int rc;
  if (dh <= 32) rc = launch_fa<1, 2>(P, Z, st);
  else if (dh <= 48) rc = launch_fa<2, 3>(P, Z, st);
  else if (dh <= 64) rc = launch_fa<2, 4>(P, Z, st);
  else if (dh <= 80) rc = launch_fa<3, 5>(P, Z, st);
  else rc = launch_fa<3, 6>(P, Z, st);
  dmx_prof_close(rec, st, 4.0 * Z * (double)Nq * Nk * dh, 2.0 * Z * (2.0 * Nq + 2.0 * Nk) * dh, Nq, Nk, dh, 1, 0, 30);
  return rc;
}
