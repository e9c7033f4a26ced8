// Fused GEMM epilogues shared by the implicit-GEMM kernels (gemm_conv.hip) and the fused resblock-pair kernel
// (conv_pair.hip).  Device-only; include inside a .hip translation unit.
#pragma once
#include "dmx_common.h"
#include <type_traits>

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}


// Fused epilogue shared by both kernels: lane holds n = n0 + j*16 + 4*lq + {0..3} for pixel row m = m0 + i*16 + lr.
template <int FM, int FN>
__device__ __forceinline__ void gemm_epilogue(const GemmDesc& p, f32x4 (&acc)[FM][FN], int m0, int n0, int lr, int lq,
                                              long long coff, int HqWq) {
  const int flags = p.flags;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + i * 16 + lr;
    if (m >= p.M) continue;
    const int b = m / HqWq, rem = m - b * HqWq;
    const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
    const long long orow = ((long long)b * p.Ho + (qy * p.osy + p.ooy)) * p.Wo + (qx * p.osx + p.oox);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + j * 16 + lq * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (flags & EPI_MASK) {
        const uint2 xr = *reinterpret_cast<const uint2*>(p.X + coff + orow * p.ldx + n);
        const float s = p.mask_slope;
        v[0] *= (alo(xr.x) > 0.f) ? 1.f : s;
        v[1] *= (ahi(xr.x) > 0.f) ? 1.f : s;
        v[2] *= (alo(xr.y) > 0.f) ? 1.f : s;
        v[3] *= (ahi(xr.y) > 0.f) ? 1.f : s;
      }
      if (flags & EPI_BIAS) {
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
      if (flags & EPI_ROWBIAS) {
        const float4 bb = *reinterpret_cast<const float4*>(p.rowbias + (long long)b * (p.ldrb ? p.ldrb : p.N) + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
      if (flags & EPI_RESID) {
        const uint2 rr = *reinterpret_cast<const uint2*>(p.R + coff + orow * p.ldr + n);
        float r0 = alo(rr.x), r1 = ahi(rr.x), r2 = alo(rr.y), r3 = ahi(rr.y);
        if (flags & EPI_RESID_INV) {
          const float is = p.resid_inv_slope;
          r0 = r0 > 0.f ? r0 : r0 * is; r1 = r1 > 0.f ? r1 : r1 * is; r2 = r2 > 0.f ? r2 : r2 * is; r3 = r3 > 0.f ? r3 : r3 * is;
        }
        v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
      }
      const float al = p.alpha;
      v[0] *= al; v[1] *= al; v[2] *= al; v[3] *= al;
      if (flags & EPI_F32OUT) {
        float* cp = reinterpret_cast<float*>(p.C) + coff + orow * p.ldc + n;
        if (flags & EPI_ACCUM) {
          const float4 o = *reinterpret_cast<const float4*>(cp);
          v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
        }
        if (flags & EPI_TANH) { v[0] = tanhf(v[0]); v[1] = tanhf(v[1]); v[2] = tanhf(v[2]); v[3] = tanhf(v[3]); }
        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        act_t* cp = reinterpret_cast<act_t*>(p.C) + coff + orow * p.ldc + n;
        if (flags & EPI_ACCUM) {
          const uint2 o = *reinterpret_cast<const uint2*>(cp);
          v[0] += alo(o.x); v[1] += ahi(o.x); v[2] += alo(o.y); v[3] += ahi(o.y);
        }
        if (flags & EPI_TANH) { v[0] = tanhf(v[0]); v[1] = tanhf(v[1]); v[2] = tanhf(v[2]); v[3] = tanhf(v[3]); }
        if (!(flags & EPI_NO_C)) *reinterpret_cast<uint2*>(cp) = make_uint2(pack2a(v[0], v[1]), pack2a(v[2], v[3]));
      }
      if (flags & EPI_LRELU2) {
        const float s = p.act_slope;
        const float a0 = v[0] > 0.f ? v[0] : v[0] * s, a1 = v[1] > 0.f ? v[1] : v[1] * s;
        const float a2 = v[2] > 0.f ? v[2] : v[2] * s, a3 = v[3] > 0.f ? v[3] : v[3] * s;
        *reinterpret_cast<uint2*>(p.C2 + coff + orow * p.ldc2 + n) = make_uint2(pack2a(a0, a1), pack2a(a2, a3));
      }
    }
  }
}

// sign bits of 8 packed 16-bit values: bit e set <=> element e > 0 (sign clear and non-zero; fp16 and bf16 alike).
// Branch-free on the packed words (the epilogue is VALU-bound): a half is positive iff its magnitude is >= 1 ulp -- adding 0x7fff to
// the 15 magnitude bits carries into bit 15 exactly then, and never across halves -- and its own sign bit is clear.
__device__ __forceinline__ unsigned dmx_pos8(const uint4& v) {
  auto pos = [](uint32_t u) -> uint32_t { return (((u & 0x7fff7fffu) + 0x7fff7fffu) & ~u) & 0x80008000u; };   // flags at bits 15 and 31
  const uint32_t w = (pos(v.x) >> 15) | (pos(v.y) >> 13) | (pos(v.z) >> 11) | (pos(v.w) >> 9);               // low halves: bits 0,2,4,6; high: 16,18,20,22
  return (w & 0x55u) | ((w >> 15) & 0xaau);
}

// rows per epilogue chunk: 16 * (largest of 4,3,2,1 that divides FM); 80- and 32-row chunks both measured slower
#ifndef DMX_GNB_HB
#define DMX_GNB_HB 4          // rows of the saved GroupNorm input in flight per lane in the backward-sums epilogue (8: one round trip per chunk)
#endif
#ifndef DMX_EPI_IB
#define DMX_EPI_IB 4
#endif
template <int FM>
struct EpiChunk { static constexpr int IB = (FM % DMX_EPI_IB == 0) ? DMX_EPI_IB : (FM % 4 == 0) ? 4 : (FM % 3 == 0) ? 3 : (FM % 2 == 0) ? 2 : 1; static constexpr int CH = IB * 16; };

// ---------------------------------------------------------------------------------------------
// LDS-staged epilogue for fp16 outputs.  The MFMA accumulator layout gives a lane 4 channels of one pixel
// (8-byte pieces, 32-byte runs per row); written straight to HBM that is 1/4 of a cache line per row and the
// epilogue ran at ~2 TB/s, 35-65 % of a HiFi-GAN layer's time.  Here every tensor the epilogue touches (mask
// source X, residual R, previous C, outputs C and C2) moves between HBM and a wave-private LDS tile in full
// row segments (16 B per lane, TN*2-byte contiguous runs = whole cache lines) and is exchanged with the
// accumulator layout through LDS.  Rows are handled in chunks of CH <= 64 to fit 8 waves in the stage buffers.
// BITS: instantiate the sign-bit tape paths (EPI_MASKBITS / EPI_BITS2).  They live in SEPARATE kernel instantiations (the launcher
// picks by flag): merely having them in the common epilogue cost every other layer ~5 % (measured on the VAE), executed or not,
// and two copies inside one kernel pushed the 256x256 tile into scratch.
template <int FM, int FN, int EM>
__device__ __forceinline__ void gemm_epilogue_lds_impl(const GemmDesc& p, f32x4 (&acc)[FM][FN], int m0, int n0, int lane,
                                                       long long coff, int HqWq, char* wl, int mlimit,
                                                       const uint2 (&rpre)[FM * FN], const bool use_rpre) {
  // rpre: the residual tile already in registers in accumulator layout ([FM][FN] 8-byte pieces; conv_pair.hip takes it from
  // its LDS slab) -- the residual tensor is then not read from HBM again
  constexpr bool BITS = EM == 1;          // sign-bit tape paths
  constexpr bool SOFT = EM == 2;          // fused softmax backward (EPI_SOFTBWD)
  constexpr bool GEGLU = EM == 3;         // value * gelu(gate) of interleaved fragment pairs, half-width output (EPI_GEGLU)
  constexpr bool ROWS = EM == 4;          // per-row partial sums of the output for a LayerNorm folded into the consumer (EPI_ROWSTATS)
  static_assert(!ROWS || FN == 2 || FN == 4, "row statistics come in slots of 32 columns");
  constexpr bool GNS = EM == 5;           // GroupNorm partial sums of the output tile for the consumer's GroupNorm (EPI_GNSTATS)
  constexpr bool GNB = EM == 6;           // GroupNorm BACKWARD partial sums: this launch produces dy of a GroupNorm(+SiLU) (EPI_GNBWD)
  static_assert(!GEGLU || FN % 2 == 0, "GEGLU pairs accumulator fragments");
  constexpr int CH = EpiChunk<FM>::CH;
  constexpr int IB = EpiChunk<FM>::IB;
  constexpr int TNB = FN * 32;            // bytes per tile row
  constexpr int PITCH = TNB + 16;
  constexpr int CPR = TNB / 16;           // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;           // rows per wave-instruction in the row-major phase
  constexpr int NIT = CH / RPI;           // wave-instructions per tensor and chunk
  static_assert(CH % RPI == 0, "chunk rows must be a multiple of the rows per wave-instruction");
  const int lr = lane & 15, lq = lane >> 4;
  const int rr = lane / CPR, cch = lane - rr * CPR;
  int* tab = reinterpret_cast<int*>(wl + CH * PITCH);   // output row index per tile row (-1: out of range), general map only
  const int flags = p.flags;
  const int ncol = n0 + cch * 8;
  const bool col_ok = ncol < p.N;
  const int mend = p.M < mlimit ? p.M : mlimit;
  // stride-1 convolutions and plain GEMMs write GEMM row m to output row m: no index table, no divisions
  const bool ident = p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.Ho == p.Hq && p.Wo == p.Wq;
#define DMX_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  // the channel bias depends on the column only: fetch it once, ahead of everything else (inside the chunk loop each chunk
  // paid an exposed L2 round trip for it)
  float4 bcol[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + j * 16 + lq * 4;
    bcol[j] = ((flags & EPI_BIAS) && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // EPI_GNSTATS: (sum v, sum v^2) of the STORED 16-bit values per 4-channel quad and SLOT of GN_SLOT = 64 consecutive rows of an image (the
  // image's last slot may be shorter).  A slot is one row chunk of this epilogue: the statistics instantiations are the tiles with 64-row
  // chunks and 64-column wave tiles (static_assert below), whose lanes own the same rows and columns of a chunk in the same order, so a
  // slot's two sums are the same bits whichever of those tiles produced it -- the statistics of a clip depend neither on the batch size
  // (other M, other tile choice) nor on its position in the batch.  A wave tile never holds rows of two images: either the rows per image
  // (HqWq) are a multiple of its rows, or the launch tiles M per image (gemm_glds_kernel) and `mlimit` ends the tile at its image's last
  // row.  A lane of the row-major phase owns 8 channels = 2 quads of RPI-strided rows: gs = (s, q) of its two quads, reduced over the lanes
  // of a column chunk and written at the end of every row chunk; gn_parts_kernel (elementwise.hip) Chan-combines the slots.  Taken from the
  // LDS-staged tile stage_out leaves behind (like emit_bits): the values GroupNorm will read, no accumulator registers held longer.
  float gs[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int GN_SLOT = 64;
  static_assert(!(GNS || GNB) || (CH == GN_SLOT && FN == 4), "GroupNorm statistics: 64-row chunks of 64-column wave tiles only (one summation order)");
  const int gn_b0 = (GNS || GNB) ? m0 / HqWq : 0;
  static_for<0, FM / IB>([&](auto H) {
    constexpr int h = decltype(H)::value;
    // ---- output row of each tile row this lane touches in the row-major phases
    int orows[NIT];          // output row (< 2^31 rows per tensor) or -1
    if (ident) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int m = m0 + h * CH + it * RPI + rr;
        orows[it] = m < mend ? m : -1;
      }
    } else {
      if (lane < CH) {
        const int m = m0 + h * CH + lane;
        int orow = -1;
        if (m < mend) {
          const int b = m / HqWq, rem = m - b * HqWq;
          const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
          orow = (b * p.Ho + (qy * p.osy + p.ooy)) * p.Wo + (qx * p.osx + p.oox);
        }
        tab[lane] = orow;
      }
      DMX_LDS_SYNC();
#pragma unroll
      for (int it = 0; it < NIT; ++it) orows[it] = tab[it * RPI + rr];
      DMX_LDS_SYNC();
    }
    // row-major global -> LDS -> accumulator-layout pieces, combined into acc by `f`.  All loads of a chunk are issued back to
    // back from always-valid addresses (out-of-range rows read element 0 and are zeroed): one exposed latency per tensor.
    auto stage_in = [&](const act_t* G, int ld, auto&& f) {
      {
        uint4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const bool ok = orows[it] >= 0 && col_ok;
          const long long off = ok ? (long long)orows[it] * ld + ncol : 0ll;
          v[it] = *reinterpret_cast<const uint4*>(G + coff + off);
          if (!ok) v[it] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) *reinterpret_cast<uint4*>(wl + (it * RPI + rr) * PITCH + cch * 16) = v[it];
      }
      DMX_LDS_SYNC();
#pragma unroll
      for (int ii = 0; ii < IB; ++ii)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const uint2 q = *reinterpret_cast<const uint2*>(wl + (ii * 16 + lr) * PITCH + (j * 16 + lq * 4) * 2);
          f(acc[h * IB + ii][j], alo(q.x), ahi(q.x), alo(q.y), ahi(q.y));
        }
      DMX_LDS_SYNC();
    };
    auto stage_out = [&](act_t* G, int ld, auto&& f) {
      // (GEGLU: only the even fragments carry results; they are packed into the left half of the tile rows and leave as N / 2 columns)
      constexpr int JS = GEGLU ? 2 : 1;
      static_for<0, IB>([&](auto II) {
        static_for<0, FN / JS>([&](auto JJ) {
          constexpr int ii = decltype(II)::value, j = decltype(JJ)::value * JS;
          float o[4];
          f(acc[h * IB + ii][j], o);
          *reinterpret_cast<uint2*>(wl + (ii * 16 + lr) * PITCH + ((j / JS) * 16 + lq * 4) * 2) = make_uint2(pack2a(o[0], o[1]), pack2a(o[2], o[3]));
        });
      });
      DMX_LDS_SYNC();
      constexpr int HN = NIT > 4 ? NIT / 2 : NIT;      // row-major read / store in groups of <= 4 instructions (register pressure)
#pragma unroll
      for (int g0 = 0; g0 < NIT; g0 += HN) {
      uint4 v[HN];
#pragma unroll
      for (int it = 0; it < HN; ++it) v[it] = *reinterpret_cast<const uint4*>(wl + ((g0 + it) * RPI + rr) * PITCH + cch * 16);
      // (no explicit wait here: the stores depend on v through registers, and a later LDS write of this wave cannot pass
      //  these reads -- LDS operations of one wave complete in order.  A "memory"-clobbering asm at this point made hipcc
      //  keep v[] in scratch: every output byte was written twice.)
      const int ocol = GEGLU ? (n0 >> 1) + cch * 8 : ncol;
      const bool ocol_ok = GEGLU ? (cch < CPR / 2 && ocol < (p.N >> 1)) : col_ok;
#pragma unroll
      for (int it = 0; it < HN; ++it) {
        if (orows[g0 + it] >= 0 && ocol_ok) *reinterpret_cast<uint4*>(G + coff + (long long)orows[g0 + it] * ld + ocol) = v[it];
      }
      }
    };
    // EPI_BITS2: a separate sweep over the packed tile stage_out left in LDS -- one byte per 16-byte chunk, bit e <=> channel ncol + e > 0,
    // the backward sweep's leaky-relu' mask.  (Taking the bytes from the registers of the store loop instead was measured in round 3:
    // no gain forward, 5-8 % slower C = 32 backward instances through register allocation; scripts/dev/pair_bench.py.)
    auto emit_gn = [&]() {
#ifndef DMX_BF16
      typedef _Float16 dmx_h2 __attribute__((ext_vector_type(2)));
      const dmx_h2 one = {(_Float16)1.0f, (_Float16)1.0f};
#endif
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const uint4 v = *reinterpret_cast<const uint4*>(wl + (it * RPI + rr) * PITCH + cch * 16);
        if (orows[it] >= 0 && col_ok) {
#ifndef DMX_BF16
          const dmx_h2 a0 = __builtin_bit_cast(dmx_h2, v.x), a1 = __builtin_bit_cast(dmx_h2, v.y);
          const dmx_h2 a2 = __builtin_bit_cast(dmx_h2, v.z), a3 = __builtin_bit_cast(dmx_h2, v.w);
          const float s0 = __builtin_amdgcn_fdot2(a0, one, __builtin_amdgcn_fdot2(a1, one, 0.f, false), false);
          const float q0 = __builtin_amdgcn_fdot2(a0, a0, __builtin_amdgcn_fdot2(a1, a1, 0.f, false), false);
          const float s1 = __builtin_amdgcn_fdot2(a2, one, __builtin_amdgcn_fdot2(a3, one, 0.f, false), false);
          const float q1 = __builtin_amdgcn_fdot2(a2, a2, __builtin_amdgcn_fdot2(a3, a3, 0.f, false), false);
#else                                                            // bf16 build: no 16-bit dot product of this format; fp32 sums of the unpacked values
          const float f0 = alo(v.x), f1 = ahi(v.x), f2 = alo(v.y), f3 = ahi(v.y), f4 = alo(v.z), f5 = ahi(v.z), f6 = alo(v.w), f7 = ahi(v.w);
          const float s0 = (f0 + f1) + (f2 + f3), s1 = (f4 + f5) + (f6 + f7);
          const float q0 = __builtin_fmaf(f0, f0, __builtin_fmaf(f1, f1, __builtin_fmaf(f2, f2, f3 * f3)));
          const float q1 = __builtin_fmaf(f4, f4, __builtin_fmaf(f5, f5, __builtin_fmaf(f6, f6, f7 * f7)));
#endif
          gs[0] += s0; gs[1] += q0; gs[2] += s1; gs[3] += q1;
        }
      }
    };
    // EPI_GNBWD: this launch's output is dy, the gradient w.r.t. the OUTPUT of a GroupNorm (+SiLU) whose input x and per-(image, channel)
    // scale / shift / mean are on the tape: the two backward sums per group -- sum dxh and sum dxh (x - mean) with dxh = dy silu'(x scale +
    // shift) scale (gn_partial_kernel<1>'s, with the rstd factor left to the finalize kernel) -- are taken here from the staged tile and the
    // matching rows of x, per wave tile / image / 4-channel quad like the forward sums: the standalone pass over x and dy disappears.
    auto emit_gnb = [&]() {
      {
        const int bb = gn_b0;
        float sc[8], sf[8];
        {
          const float* scp = p.gnb_scale + (long long)bb * p.N + ncol;
          const float* sfp = p.gnb_shift + (long long)bb * p.N + ncol;
          const float4 a0 = col_ok ? *reinterpret_cast<const float4*>(scp) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 a1 = col_ok ? *reinterpret_cast<const float4*>(scp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 c0 = col_ok ? *reinterpret_cast<const float4*>(sfp) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 c1 = col_ok ? *reinterpret_cast<const float4*>(sfp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
          sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
          sf[0] = c0.x; sf[1] = c0.y; sf[2] = c0.z; sf[3] = c0.w; sf[4] = c1.x; sf[5] = c1.y; sf[6] = c1.z; sf[7] = c1.w;
        }
        // group mean of each of the lane's two quads (a quad lies inside one group: channels per group % 4 == 0).  The second sum is taken
        // of dxh (x - mean), like gn_partial_kernel<1>: sum dxh x - mean sum dxh afterwards cancels catastrophically in fp32 when the
        // group's mean dominates (measured: the VAE input-gradient of a clip moved by 3.5 % between batch compositions)
        const int ngrp = p.N / p.gnb_cpg;
        const float mu[2] = {col_ok ? p.gnb_stats[((long long)bb * ngrp + ncol / p.gnb_cpg) * 2] : 0.f,
                             col_ok ? p.gnb_stats[((long long)bb * ngrp + (ncol + 4) / p.gnb_cpg) * 2] : 0.f};
        // rows of x in flight per lane: a divisor of NIT, at most 4 (8 spilled on the 256 x 256 and 512 x 128 tiles)
        constexpr int HB = DMX_GNB_HB > 4 && NIT % DMX_GNB_HB == 0 ? DMX_GNB_HB : (NIT % 4 == 0 ? 4 : (NIT % 3 == 0 ? 3 : (NIT % 2 == 0 ? 2 : 1)));
#pragma unroll
        for (int g0 = 0; g0 < NIT; g0 += HB) {
          uint4 xv[HB];
          bool ok[HB];
#pragma unroll
          for (int it = 0; it < HB; ++it) {
            ok[it] = orows[g0 + it] >= 0 && col_ok;
            xv[it] = *reinterpret_cast<const uint4*>(p.gnb_x + (ok[it] ? (long long)orows[g0 + it] * p.gnb_ldx + ncol : 0ll));
          }
#pragma unroll
          for (int it = 0; it < HB; ++it) {
            const uint4 dv = *reinterpret_cast<const uint4*>(wl + ((g0 + it) * RPI + rr) * PITCH + cch * 16);
            if (ok[it]) {
              const float xf[8] = {alo(xv[it].x), ahi(xv[it].x), alo(xv[it].y), ahi(xv[it].y), alo(xv[it].z), ahi(xv[it].z), alo(xv[it].w), ahi(xv[it].w)};
              const float df[8] = {alo(dv.x), ahi(dv.x), alo(dv.y), ahi(dv.y), alo(dv.z), ahi(dv.z), alo(dv.w), ahi(dv.w)};
              float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float dz = df[e];
                if (p.gnb_silu) {
                  const float z = __builtin_fmaf(xf[e], sc[e], sf[e]);
                  const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-z));
                  dz *= sg * (1.f + z * (1.f - sg));
                }
                const float dxh = dz * sc[e];
                a1[e >> 2] += dxh;
                a2[e >> 2] = __builtin_fmaf(dxh, xf[e] - mu[e >> 2], a2[e >> 2]);
              }
              gs[0] += a1[0]; gs[1] += a2[0]; gs[2] += a1[1]; gs[3] += a2[1];
            }
          }
        }
      }
    };
    auto emit_bits = [&]() {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const uint4 v = *reinterpret_cast<const uint4*>(wl + (it * RPI + rr) * PITCH + cch * 16);
        if (orows[it] >= 0 && col_ok) p.B2[(long long)orows[it] * p.ldb2 + (ncol >> 3)] = (unsigned char)dmx_pos8(v);
      }
    };
    if (BITS && (flags & EPI_MASKBITS)) {
      // mask from the sign-bit tensor: FN * 2 bytes cover this wave's 16 * FN columns of one row; the four lanes that share a row
      // (lq = 0..3) read the same bytes and pick their nibble -- no LDS round trip, 1/16 of the bytes of the 16-bit mask source
      static_assert(FN == 2 || FN == 4, "sign-bit masks: wave tiles of 32 or 64 columns");
      const float sl = p.mask_slope;
      const bool span_ok = n0 + FN * 16 <= p.N;
      const int sh0 = 8 * (lq >> 1) + 4 * (lq & 1);             // bit position of this lane's nibble inside a 16-column pair
      // all IB mask words are fetched before the first one is used: with load and use in one loop body every row fragment paid its own
      // L2 round trip (in-kernel stamps of the pair kernel, scripts/dev/r03_pair_stamps.py)
      uint32_t q0s[IB], q1s[IB];
      int orow_s[IB];
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
        const int r = ii * 16 + lr;
        if (ident) { const int m = m0 + h * CH + r; orow_s[ii] = m < mend ? m : -1; }
        else orow_s[ii] = tab[r];
      }
      if (span_ok) {                                            // (wave-uniform) straight-line loads from always-valid rows: one round trip for all
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
          const unsigned char* src = p.XB + (long long)(orow_s[ii] >= 0 ? orow_s[ii] : 0) * p.ldxb + (n0 >> 3);
          if constexpr (FN == 4) { const uint2 q = *reinterpret_cast<const uint2*>(src); q0s[ii] = q.x; q1s[ii] = q.y; }
          else { q0s[ii] = *reinterpret_cast<const uint32_t*>(src); q1s[ii] = 0xffffffffu; }
        }
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) if (orow_s[ii] < 0) { q0s[ii] = 0xffffffffu; q1s[ii] = 0xffffffffu; }
      } else {                                                  // N tail (narrow layers: N = 8 / 16 inside a 32-column wave tile)
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
          uint32_t q0 = 0xffffffffu, q1 = 0xffffffffu;
          if (orow_s[ii] >= 0) {
            const unsigned char* src = p.XB + (long long)orow_s[ii] * p.ldxb + (n0 >> 3);
            q0 = q1 = 0u;
#pragma unroll 1                 // rare path: keep it rolled (unrolled, its byte loads were hoisted and cost ~50 VGPRs everywhere)
            for (int k = 0; k < FN * 2; ++k) {
              const uint32_t by = n0 + k * 8 < p.N ? (uint32_t)src[k] : 0xffu;
              if (k < 4) q0 |= by << (8 * k); else q1 |= by << (8 * (k - 4));
            }
          }
          q0s[ii] = q0; q1s[ii] = q1;
        }
      }
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
        const uint32_t q0 = q0s[ii], q1 = q1s[ii];
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const unsigned nib = ((j < 2 ? q0 : q1) >> (sh0 + 16 * (j & 1)));
          f32x4& a = acc[h * IB + ii][j];
          a[0] *= (nib & 1u) ? 1.f : sl; a[1] *= (nib & 2u) ? 1.f : sl; a[2] *= (nib & 4u) ? 1.f : sl; a[3] *= (nib & 8u) ? 1.f : sl;
        }
      }
    } else if (SOFT && (flags & EPI_SOFTBWD)) {
      // dS = P * (dP - delta[row]): delta is one fp32 per GEMM row of this batch (coff = z * sCo since Zi == 1)
      const float* rv = p.rowbias + (p.sCo ? coff / p.sCo : 0ll) * (long long)p.M;
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
        const int m = m0 + h * CH + ii * 16 + lr;
        const float dl = m < mend ? rv[m] : 0.f;
#pragma unroll
        for (int j = 0; j < FN; ++j) { f32x4& a = acc[h * IB + ii][j]; a[0] -= dl; a[1] -= dl; a[2] -= dl; a[3] -= dl; }
      }
      stage_in(p.X, p.ldx, [&](f32x4& a, float x0, float x1, float x2, float x3) { a[0] *= x0; a[1] *= x1; a[2] *= x2; a[3] *= x3; });
    } else if (flags & EPI_MASK) {
      const float sl = p.mask_slope;
      stage_in(p.X, p.ldx, [&](f32x4& a, float x0, float x1, float x2, float x3) {
        a[0] *= x0 > 0.f ? 1.f : sl; a[1] *= x1 > 0.f ? 1.f : sl; a[2] *= x2 > 0.f ? 1.f : sl; a[3] *= x3 > 0.f ? 1.f : sl;
      });
    }
    if (flags & (EPI_BIAS | EPI_ROWBIAS)) {
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
        int bimg = 0;
        if (flags & EPI_ROWBIAS) {
          const int m = m0 + h * CH + ii * 16 + lr;
          bimg = (m < p.M ? m : 0) / HqWq;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = n0 + j * 16 + lq * 4;
          if (n >= p.N) continue;
          f32x4& a = acc[h * IB + ii][j];
          if (flags & EPI_BIAS) {
            const float4 bb = bcol[j];
            a[0] += bb.x; a[1] += bb.y; a[2] += bb.z; a[3] += bb.w;
          }
          if (flags & EPI_ROWBIAS) {
            const float4 bb = *reinterpret_cast<const float4*>(p.rowbias + (long long)bimg * (p.ldrb ? p.ldrb : p.N) + n);
            a[0] += bb.x; a[1] += bb.y; a[2] += bb.z; a[3] += bb.w;
          }
        }
      }
    }
    if constexpr (GEGLU) {
      // value * gelu_erf(gate): fragment j holds the values, j + 1 the gates of the same 16 channels (fp32, before any rounding)
#pragma unroll
      for (int ii = 0; ii < IB; ++ii)
#pragma unroll
        for (int j = 0; j < FN; j += 2) {
          f32x4& a = acc[h * IB + ii][j];
          const f32x4& g = acc[h * IB + ii][j + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] *= 0.5f * g[e] * (1.f + erff(g[e] * 0.70710678118654752f));
        }
    }
    if ((flags & EPI_RESID) && use_rpre) {
      const float is = (flags & EPI_RESID_INV) ? p.resid_inv_slope : 1.f;
#pragma unroll
      for (int ii = 0; ii < IB; ++ii)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const uint2 q = rpre[(h * IB + ii) * FN + j];
          const float x0 = alo(q.x), x1 = ahi(q.x), x2 = alo(q.y), x3 = ahi(q.y);
          f32x4& a = acc[h * IB + ii][j];
          a[0] += fminf(x0, x0 * is); a[1] += fminf(x1, x1 * is); a[2] += fminf(x2, x2 * is); a[3] += fminf(x3, x3 * is);
        }
    } else if (flags & EPI_RESID) {
      const float is = (flags & EPI_RESID_INV) ? p.resid_inv_slope : 1.f;
      stage_in(p.R, p.ldr, [&](f32x4& a, float x0, float x1, float x2, float x3) {
        // x > 0 ? x : x * is  ==  min(x, x * is) for is >= 1 (is = 1 / leaky slope, or exactly 1 for a plain residual)
        a[0] += fminf(x0, x0 * is); a[1] += fminf(x1, x1 * is); a[2] += fminf(x2, x2 * is); a[3] += fminf(x3, x3 * is);
      });
    }
    if (p.alpha != 1.f) {
      const float al = p.alpha;
#pragma unroll
      for (int ii = 0; ii < IB; ++ii)
#pragma unroll
        for (int j = 0; j < FN; ++j) { f32x4& a = acc[h * IB + ii][j]; a[0] *= al; a[1] *= al; a[2] *= al; a[3] *= al; }
    }
    if (flags & EPI_ACCUM)
      stage_in(reinterpret_cast<const act_t*>(p.C), p.ldc, [&](f32x4& a, float x0, float x1, float x2, float x3) { a[0] += x0; a[1] += x1; a[2] += x2; a[3] += x3; });
    if (flags & EPI_TANH) {
#pragma unroll
      for (int ii = 0; ii < IB; ++ii)
#pragma unroll
        for (int j = 0; j < FN; ++j) { f32x4& a = acc[h * IB + ii][j]; a[0] = tanhf(a[0]); a[1] = tanhf(a[1]); a[2] = tanhf(a[2]); a[3] = tanhf(a[3]); }
    }
    if constexpr (ROWS) {
      // EPI_ROWSTATS: (sum v, sum v^2) of every output row over each 32-column slot of this wave's columns, from the final fp32 values
      // (what the LayerNorm folded into the next projection needs; ln_apply in gemm_tile.h adds the slots of a row).  A lane holds 4
      // values per fragment of row 16 ii + lr: two fragments make a slot, two xor-shuffles over lq finish it, lane lq == 0 stores it.
      if (flags & EPI_ROWSTATS) {
        float2* rso = reinterpret_cast<float2*>(p.rowstats_out);
        const int ns = p.nslots;
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
          const int m = m0 + h * CH + ii * 16 + lr;
#pragma unroll
          for (int sl = 0; sl < FN / 2; ++sl) {
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const int j = sl * 2 + jj;
              const f32x4& a = acc[h * IB + ii][j];
              if (n0 + j * 16 + lq * 4 < p.N) {
                sv += (a[0] + a[1]) + (a[2] + a[3]);
                qv += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
              }
            }
            sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
            qv += __shfl_xor(qv, 16, 64); qv += __shfl_xor(qv, 32, 64);
            const int slot = (n0 >> 5) + sl;
            if (lq == 0 && m < mend && slot < ns) rso[(long long)m * ns + slot] = make_float2(sv, qv);
          }
        }
      }
    }
    const bool bits2 = (flags & EPI_BITS2) != 0;
    if (!(flags & EPI_NO_C)) {
      stage_out(reinterpret_cast<act_t*>(p.C), p.ldc, [&](const f32x4& a, float (&o)[4]) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; });
      if constexpr (BITS) { if (bits2 && !(flags & EPI_LRELU2)) emit_bits(); }
      if constexpr (GNS) { if (flags & EPI_GNSTATS) emit_gn(); }
      if constexpr (GNB) { if (flags & EPI_GNBWD) emit_gnb(); }
      if constexpr (GNS || GNB) {
        if ((flags & (EPI_GNSTATS | EPI_GNBWD)) && m0 + h * CH < mend) {          // this chunk = slot (m0 + h CH - image start) / 64 of its image
#pragma unroll
          for (int o = CPR; o < 64; o <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) gs[k] += __shfl_xor(gs[k], o, 64);
          }
          if (rr == 0 && col_ok) {
            const int nq = p.N >> 2, slots = (HqWq + GN_SLOT - 1) / GN_SLOT + 1;
            float* dst = p.gn_part + (((long long)gn_b0 * slots + (m0 + h * CH - gn_b0 * HqWq) / GN_SLOT) * nq + (ncol >> 2)) * 2;
            *reinterpret_cast<float4*>(dst) = make_float4(gs[0], gs[1], gs[2], gs[3]);
          }
        }
        gs[0] = gs[1] = gs[2] = gs[3] = 0.f;
      }
      DMX_LDS_SYNC();
    }
    if (flags & EPI_LRELU2) {
      const float sl = p.act_slope;
      stage_out(p.C2, p.ldc2, [&](const f32x4& a, float (&o)[4]) {
        // leaky-relu with 0 <= slope <= 1 (checked at launch): max(v, v * slope)
        o[0] = fmaxf(a[0], a[0] * sl); o[1] = fmaxf(a[1], a[1] * sl); o[2] = fmaxf(a[2], a[2] * sl); o[3] = fmaxf(a[3], a[3] * sl);
      });
      if constexpr (BITS) { if (bits2) emit_bits(); }
      DMX_LDS_SYNC();
    }
  });
#undef DMX_LDS_SYNC
}

template <int FM, int FN, int EM>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmDesc& p, f32x4 (&acc)[FM][FN], int m0, int n0, int lane,
                                                  long long coff, int HqWq, char* wl, int mlimit = 0x7fffffff) {
  const uint2 none[FM * FN] = {};
  gemm_epilogue_lds_impl<FM, FN, EM>(p, acc, m0, n0, lane, coff, HqWq, wl, mlimit, none, false);
}

}  // namespace
