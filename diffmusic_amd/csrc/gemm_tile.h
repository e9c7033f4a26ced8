// Tile kernels of the implicit-GEMM family (see gemm_conv.hip for the overview): the register-staged 4-wave kernel, the LDS-DMA
// tile (`glds_tile`) and their launch templates.  Included by gemm_conv.hip (the common instantiations and the dispatcher) and by
// gemm_ln.hip (the LayerNorm-folding instantiations used by the U-Net's transformer blocks), so that the two sets compile in parallel.
#pragma once
#include "dmx_common.h"
#include <type_traits>
#include <cstdlib>

#include "gemm_epilogue.h"
#include "conv_pair.h"

namespace {

constexpr int BK = 64;  // 16-bit elements per K-step (128 B per tile row)
// fragment steps the lagging half of the register-bound 320-row tile carries across the barrier: 0 = that tile stays unstaggered
// (measured on one device: 2 or 4 carried steps cost it 2 %; its 256 registers hold no more than 4)
constexpr int DMX_DEF_BIG = 0;

// ---- LayerNorm folded into the projection that consumes it (EPI_LNFOLD; the U-Net's LN -> QKV / Q / FF1 pairs).
//   LN(x) . W^T + b  =  rstd * (x . W'^T - mean * colsum) + b',   W' = W diag(gamma),  colsum[n] = sum_k W'[n, k],  b' = b + W beta
// The GEMM runs on the RAW rows with the stock K loop; mean / rstd of a row come from per-row partial sums (sum x, sum x^2 per 32
// columns) that the GEMM which PRODUCED the rows wrote from its epilogue (EPI_ROWSTATS, gemm_epilogue.h).  Lane (lr, lq) owns the
// accumulators of rows 16 i + lr: the four lq lanes of a row split its slots and two xor-shuffles complete the sums in exactly the lanes
// that need them.  No normalised tensor, no LayerNorm launch, nothing added to the K loop.
// (Round 4 first gathered the statistics from the activation fragments inside the K loop with v_dot2c_f32_f16: the 8-wave tiles lost the
//  5-6 us per launch that the removed layernorm kernels had cost, and the FN = 2 tiles produced wrong accumulators in lanes 48-63 on
//  full grids -- VALU beside the hand-scheduled asm fragment reads; scripts/dev/r04_lnfold_debug.py.)
// prologue part (all threads of the workgroup; runs while the first tiles are in flight, when registers are free): (mean, rstd) of the BM
// rows of the tile and the BN row sums of the packed weights go to LDS -- s_ln[BM] float2, s_cs[BN] float.  After the K loop ln_apply
// needs two short LDS reads per row fragment instead of global round trips with 128+ accumulators live (fetched there, the statistics
// cost the 256 x 256 tile 7 us per launch -- more than the layernorm launch they replace; profiles/r04_unet_lnfold.log).
template <int BM, int BN, int NT>
__device__ __forceinline__ void ln_prologue(const GemmDesc& p, int m0, int n0, float2* s_ln, float* s_cs) {
  // TPR adjacent lanes share a row and split its slots; each issues up to 8 loads back to back (one memory round trip for the usual
  // <= 32 slots: a one-load-at-a-time loop over 20 slots cost the small tiles 4-5 us of exposed latency per launch)
  constexpr int TPR = NT / BM >= 4 ? 4 : (NT / BM >= 2 ? 2 : 1);
  const int tid = threadIdx.x, ns = p.nslots;
  const int r = tid / TPR, sub = tid - r * TPR;
  const float inv_c = 1.f / (float)p.K;
  if (r < BM) {                                   // (whole TPR-lane groups take the branch together: the shuffles below stay inside a group)
    const int m = m0 + r;
    const bool row_ok = m < p.M;
    const float2* rs = reinterpret_cast<const float2*>(p.rowstats_in) + (long long)(row_ok ? m : 0) * ns;
    float si = 0.f, qi = 0.f;
    for (int t0 = 0; t0 < ns; t0 += TPR * 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u * TPR + sub;
        v[u] = (row_ok && t < ns) ? rs[t] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { si += v[u].x; qi += v[u].y; }
    }
    if constexpr (TPR >= 2) { si += __shfl_xor(si, 1, 64); qi += __shfl_xor(qi, 1, 64); }
    if constexpr (TPR >= 4) { si += __shfl_xor(si, 2, 64); qi += __shfl_xor(qi, 2, 64); }
    const float mean = si * inv_c;
    if (sub == 0) s_ln[r] = make_float2(mean, rsqrtf(fmaxf(qi * inv_c - mean * mean, 0.f) + p.ln_eps));
  }
  // row sums of the packed weights and the folded bias (b + W beta; added here rather than by the epilogue's EPI_BIAS path, which costs a
  // large tile ~2 us): s_cs[0 .. BN) and s_cs[BN .. 2 BN)
  for (int c = tid; c < BN; c += NT) {
    const bool ok = n0 + c < p.N;
    s_cs[c] = ok ? p.colsum[n0 + c] : 0.f;
    s_cs[BN + c] = (ok && p.bias) ? p.bias[n0 + c] : 0.f;
  }
}
// after the K loop: rows 16 i + lr of the wave (tile row r0), columns 16 j + 4 lq .. + 3 (tile column c0)
template <int FM, int FN, int BN>
__device__ __forceinline__ void ln_apply(f32x4 (&acc)[FM][FN], const float2* s_ln, const float* s_cs, int r0, int c0, int lr, int lq) {
#pragma unroll
  for (int j = 0; j < FN; ++j) {            // (column fragment outermost: 8 live registers of column data instead of 8 FN)
    const float4 cs = *reinterpret_cast<const float4*>(s_cs + c0 + j * 16 + lq * 4);
    const float4 bs = *reinterpret_cast<const float4*>(s_cs + BN + c0 + j * 16 + lq * 4);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const float2 st = s_ln[r0 + i * 16 + lr];
      const float mean = st.x, rstd = st.y;
      f32x4& a = acc[i][j];
      a[0] = __builtin_fmaf(rstd, a[0] - mean * cs.x, bs.x); a[1] = __builtin_fmaf(rstd, a[1] - mean * cs.y, bs.y);
      a[2] = __builtin_fmaf(rstd, a[2] - mean * cs.z, bs.z); a[3] = __builtin_fmaf(rstd, a[3] - mean * cs.w, bs.w);
    }
  }
}

template <int BM, int BN, int WM, int WN, int EM, bool LNF = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmDesc p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int A_PER = BM * 8 / NT, B_PER = BN * 8 / NT;
  constexpr int ROWS_PER_PASS = NT / 8;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(A_PER >= 1 && B_PER >= 1, "tile too small for the thread count");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  short2* s_taps = reinterpret_cast<short2*>(smem + 2 * STAGE);

  const int tid = threadIdx.x;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int z = blockIdx.y;
  const int zo = z / p.Zi, zi = z - zo * p.Zi;
  const act_t* __restrict__ Ab = p.A + zo * p.sAo + zi * p.sAi;
  const act_t* __restrict__ Wb = p.W + zo * p.sWo + zi * p.sWi;
  const long long coff = zo * p.sCo + zi * p.sCi;

  if (tid < DMX_MAX_TAPS) s_taps[tid] = make_short2(p.tdy[tid], p.tdx[tid]);
  float2* s_ln = reinterpret_cast<float2*>(smem + 2 * STAGE + DMX_MAX_TAPS * 4);      // LNF: (mean, rstd) of the BM rows, then BN weight row sums
  float* s_cs = reinterpret_cast<float*>(smem + 2 * STAGE + DMX_MAX_TAPS * 4 + BM * 8);
  int m0t = tm * BM, mlim = p.M;          // first GEMM row of this tile / first row it does not own
  if constexpr (EM == 5 || EM == 6) {     // GroupNorm partial sums: image-aligned M tiling where a wave tile would straddle images (gemm_glds_kernel)
    const int P = p.Hq * p.Wq;
    if (P % TM != 0) {
      const int tpi = (P + BM - 1) / BM, b = tm / tpi;
      m0t = b * P + (tm - b * tpi) * BM;
      mlim = (b + 1) * P;
    }
  }
  if constexpr (LNF) ln_prologue<BM, BN, NT>(p, m0t, tn * BN, s_ln, s_cs);

  // ---- per-thread gather bookkeeping (rows are fixed over the K loop)
  const int cc = tid & 7, r0 = tid >> 3;
  const int cpt = p.Ci >> 3;  // 16-byte chunks per tap
  const int HqWq = p.Hq * p.Wq;
  const act_t* a_base[A_PER];
  int a_iy[A_PER], a_ix[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int m = m0t + r0 + i * ROWS_PER_PASS;
    if (m < mlim) {
      const int b = m / HqWq, rem = m - b * HqWq;
      const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
      a_base[i] = Ab + (long long)b * p.Hi * p.Wi * p.lda;
      a_iy[i] = qy * p.sy;
      a_ix[i] = qx * p.sx;
    } else {
      a_base[i] = Ab;
      a_iy[i] = -(1 << 20);
      a_ix[i] = 0;
    }
  }
  const act_t* w_base[B_PER];
  bool w_ok[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int n = tn * BN + r0 + i * ROWS_PER_PASS;
    w_ok[i] = n < p.N;
    w_base[i] = Wb + (long long)(w_ok[i] ? n : 0) * p.ldw;
  }
  __syncthreads();  // taps visible

  uint4 ra[A_PER], rb[B_PER];
  const int kchunks = p.K >> 3;
  auto load_tile = [&](int ks) {
    const int kc = ks * 8 + cc;
    const int tap = kc / cpt;
    const int cin = (kc - tap * cpt) << 3;
    const bool kval = kc < kchunks;
    const short2 t = s_taps[tap & (DMX_MAX_TAPS - 1)];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int iy = a_iy[i] + t.x, ix = a_ix[i] + t.y;
      const bool ok = kval && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      ra[i] = make_uint4(0, 0, 0, 0);
      if (ok) ra[i] = *reinterpret_cast<const uint4*>(a_base[i] + (long long)(iy * p.Wi + ix) * p.lda + cin);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      rb[i] = make_uint4(0, 0, 0, 0);
      if (kval && w_ok[i]) rb[i] = *reinterpret_cast<const uint4*>(w_base[i] + (kc << 3));
    }
  };
  auto store_tile = [&](int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int r = r0 + i * ROWS_PER_PASS;
      *reinterpret_cast<uint4*>(sa + r * 128 + ((cc ^ (r & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int r = r0 + i * ROWS_PER_PASS;
      *reinterpret_cast<uint4*>(sb + r * 128 + ((cc ^ (r & 7)) << 4)) = rb[i];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int lr = lane & 15, lq = lane >> 4;

  f32x4 acc[FM][FN];
  // EPI_BIASINIT as in glds_tile below.  Here it is not about speed: the two kernel families must ROUND alike (bias first, then the
  // products in K order), or a layer that runs on this kernel at batch 1 and on an LDS-DMA tile at batch 8 stops being bit-identical
  // across batch sizes -- and one flipped leaky-relu sign in the vocoder tape moves its input gradient by percents
  // (tests/test_gpu_batch_parity.py)
  if (p.flags & EPI_BIASINIT) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = tn * BN + wn * TN + j * 16 + lq * 4;
      float bb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bb[e] = n + e < p.N ? p.bias[n + e] : 0.f;        // (N need not be a multiple of 4 on this kernel)
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = f32x4{bb[0], bb[1], bb[2], bb[3]};
    }
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) load_tile(ks + 1);
    const char* sa = smem + cur * STAGE + (wm * TM + lr) * 128;
    const char* sb = smem + cur * STAGE + A_BYTES + (wn * TN + lr) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int sw = ((kk * 4 + lq) ^ (lr & 7)) << 4;
      frag8_t af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const frag8_t*>(sa + i * 16 * 128 + sw);
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = *reinterpret_cast<const frag8_t*>(sb + j * 16 * 128 + sw);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = DMX_MFMA16(wf[j], af[i], acc[i][j]);
    }
    if (ks + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
  if constexpr (LNF) ln_apply<FM, FN, BN>(acc, s_ln, s_cs, wm * TM, wn * TN, lr, lq);

  if ((p.flags & EPI_F32OUT) || ((p.N | p.ldc | p.ldr | p.ldx | p.ldc2) & 7)) {     // direct path: fp32 out or rows not 16-B granular
    gemm_epilogue<FM, FN>(p, acc, m0t + wm * TM, tn * BN + wn * TN, lr, lq, coff, HqWq);
  } else {
    constexpr int EPI_CH = EpiChunk<FM>::CH;
    constexpr int EPI_WAVE_BYTES = EPI_CH * (FN * 32 + 16) + EPI_CH * 12;
    static_assert(EPI_WAVE_BYTES * WM * WN <= 2 * (BM + BN) * 128, "epilogue staging does not fit the stage buffers");
    gemm_epilogue_lds<FM, FN, EM>(p, acc, m0t + wm * TM, tn * BN + wn * TN, lane, coff, HqWq, smem + wave * EPI_WAVE_BYTES, mlim);
  }
}

// ---------------------------------------------------------------------------------------------
// Large-tile kernel: 8 waves, 256 x {256,128} x 64 tile, both operands brought in by LDS-DMA
// (buffer_load_dwordx4 ... lds): no VGPR staging, no ds_write.  Each wave-instruction fills 1 KiB =
// 8 tile rows x 128 B; the LDS image is lane-linear, so the XOR swizzle that makes the ds_read_b128
// fragment reads conflict-free is applied to the per-lane SOURCE chunk (lane l fetches logical chunk
// (l&7)^(l>>3) of row l>>3).  Conv zero padding, M/N/K tails: the lane's buffer offset is sent out of
// range and the hardware range check returns zeros.  Two LDS stages, one barrier per K-step.
constexpr unsigned OOB = 0x80000000u;   // == num_records of the descriptors below

#ifdef DMX_GEMM_STAMPS
// diagnostic build only (scripts/dev/r03_gemm_stamps.py): 100 MHz wall-clock stamps of the phases of the first 8192 workgroups of the
// last launch: entry | ring prologue landed | K loop done | epilogue's last store issued | stores acknowledged, and the hardware id
// (XCC / SE / CU) the workgroup ran on
__device__ unsigned long long g_gemm_stamps[8192 * 6];
#define DMX_GSTAMP(i) do { gstamp_v[i] = wall_clock64(); } while (0)
#else
#define DMX_GSTAMP(i) do { } while (0)
#endif

template <int BM, int BN, int WM, int WN, int NSTAGE, int EM, bool LNF = false>
__device__ __forceinline__ void glds_tile(const GemmDesc& p, char* smem, const int m0, const int tn, const int mlim) {
  // mlim: first GEMM row this tile does NOT own (p.M, or the end of the tile's image under the image-aligned tiling of the
  // GroupNorm-statistics instantiations): rows from there on are neither fetched nor stored
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int A_ISS = BM / 8 / NW, B_ISS = BN / 8 / NW;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(A_ISS >= 1 && B_ISS >= 1, "tile too small");

  constexpr int PER = A_ISS + B_ISS;              // LDS-DMA instructions per thread per stage
  constexpr int KEEP = (NSTAGE - 2) * PER;        // loads allowed in flight when the next tile must have landed

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#ifdef DMX_GEMM_STAMPS
  unsigned long long gstamp_v[5] = {0, 0, 0, 0, 0};
#endif
  DMX_GSTAMP(0);
  const int z = blockIdx.y;
  const int zo = z / p.Zi, zi = z - zo * p.Zi;
  const act_t* Ab = p.A + zo * p.sAo + zi * p.sAi;
  const act_t* Wb = p.W + zo * p.sWo + zi * p.sWi;
  const int nsplit = p.ksplit > 1 ? p.ksplit : 1, sp = blockIdx.z;     // split-K: this workgroup owns K steps [ks0, ks1)
  const long long coff = zo * p.sCo + zi * p.sCi + (nsplit > 1 ? (long long)sp * p.M * p.N : 0ll);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<act_t*>(Ab), 0, OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<act_t*>(Wb), 0, (unsigned)p.N * (unsigned)p.ldw * 2u, 0x00020000);

  // tap table in a REGISTER: lane j (< 16) holds (dy, dx) of tap j, looked up with v_readlane / ds_bpermute.  It must not live
  // in LDS: the compiler puts an s_waitcnt vmcnt(0) in front of every LDS read it can see (it may alias an in-flight LDS-DMA
  // target), and a table read at the top of the K step drained the whole ring once per step -- with 3 or 4 stages only one
  // tile was ever in flight under the MFMAs.
  int tapreg = 0;
#pragma unroll
  for (int j = 0; j < DMX_MAX_TAPS; ++j)
    if (lane == j) tapreg = (int)(unsigned char)p.tdy[j] | ((int)(unsigned char)p.tdx[j] << 8);

  const int lrow = lane >> 3, cc = (lane & 7) ^ lrow;   // logical 16-B chunk this lane fetches
  const int cpt = p.Ci >> 3;
  const int HqWq = p.Hq * p.Wq;
  const unsigned lda2 = (unsigned)p.lda * 2u;
  // per fetched row: byte offset of its chunk at tap (0, 0) / channel group 0, and a 16-bit mask of the taps that fall inside the
  // input for this row (conv zero padding, rows past M: mask 0) -- two masks per register.  The K loop then needs one bit test
  // per row and step instead of two coordinate adds and two range compares, and 3 instead of 10 registers for A_ISS = 5.
  unsigned a_lin[A_ISS], a_mask2[(A_ISS + 1) / 2];
  {
    int r_iy[A_ISS], r_ix[A_ISS];            // prologue only: the K loop keeps a_lin and the masks
    unsigned r_mask[A_ISS];
#pragma unroll
    for (int i = 0; i < A_ISS; ++i) {
      const int m = m0 + (i * NW + wave) * 8 + lrow;
      r_mask[i] = 0u; a_lin[i] = 0u;
      r_iy[i] = -(1 << 20); r_ix[i] = 0;
      if (m < mlim) {
        int b = 0, qy = 0, qx = m;                                   // plain GEMM rows (one "image" of M x 1 pixels): no divisions
        // (hoisting the divisions out -- image / pixel of the tile's first row once, rows by offset -- leaves the 3.3-6 us a workgroup
        //  spends before its first MFMA unchanged: that time is the latency of the ring's first loads; scripts/dev/r03_gemm_stamps.py)
        if (!(p.Hq == 1 && p.Wq >= p.M)) {
          b = m / HqWq; const int rem = m - b * HqWq;
          qy = rem / p.Wq; qx = rem - qy * p.Wq;
        }
        r_iy[i] = qy * p.sy; r_ix[i] = qx * p.sx;
        a_lin[i] = (unsigned)b * (unsigned)(p.Hi * p.Wi) * lda2 + (unsigned)(r_iy[i] * p.Wi + r_ix[i]) * lda2 + ((unsigned)cc << 4);
      }
    }
    for (int t = 0; t < p.ntaps; ++t) {       // wave-uniform trip count: a plain GEMM pays for one tap, not sixteen
      const int tv = __builtin_amdgcn_readlane(tapreg, t);
      const int dy = (int)(signed char)(tv & 0xff), dx = (int)(signed char)((tv >> 8) & 0xff);
#pragma unroll
      for (int i = 0; i < A_ISS; ++i)
        if ((unsigned)(r_iy[i] + dy) < (unsigned)p.Hi && (unsigned)(r_ix[i] + dx) < (unsigned)p.Wi) r_mask[i] |= 1u << t;
    }
#pragma unroll
    for (int i = 0; i < (A_ISS + 1) / 2; ++i) a_mask2[i] = 0u;
#pragma unroll
    for (int i = 0; i < A_ISS; ++i) a_mask2[i >> 1] |= r_mask[i] << ((i & 1) * 16);
  }
  // weight rows: instruction j fetches row tn * BN + (j * NW + wave) * 8 + lrow; rows past N lie beyond the descriptor's range
  // (num_records = N rows) and come back as zeros
  const unsigned ldw2 = (unsigned)p.ldw * 2u;
  const unsigned w_row = (unsigned)(tn * BN + wave * 8 + lrow) * ldw2;
  const unsigned w_step = (unsigned)(NW * 8) * ldw2;

  const int kchunks = p.K >> 3;
  // K order: with Ci % 64 == 0 a K-step is one (tap, 64-channel group); walk the taps innermost so the ~(BM + halo)
  // input rows of a channel group are re-read from L2 by consecutive K-steps instead of once per pass over all channels
  // (single-tap GEMMs take the same division-free walk: the channel groups of their only tap, the last one possibly partial)
  const bool tap_inner = (p.Ci & 63) == 0 || p.ntaps == 1;
  const int cgroups = (p.Ci + 63) >> 6;
  const int nk_all = (p.K + BK - 1) / BK;
  const int ks_per = (nk_all + nsplit - 1) / nsplit;
  const int ks0 = sp * ks_per, ks1 = ks0 + ks_per < nk_all ? ks0 + ks_per : nk_all;
  // tap-inner walk without divisions: (tap, channel group) of the NEXT step to be issued; issue() is called with consecutive steps
  int i_cg = ks0 / p.ntaps, i_tp = ks0 - i_cg * p.ntaps;
  // One K step's LDS-DMA is split in two parts so that its PER instructions can be spread over the fragment steps (an LDS-DMA
  // issued back to back with seven others and a burst of ds_reads costs the wave 100-185 cycles, one slipped between MFMA groups
  // 25-60; MI355X_MICROARCH.md): issue_begin() computes what depends on the K step only, issue_one<I>() sends instruction I.
  bool q_kval = false;          // this lane's chunk of the step exists (K tail, split-K range, partial last channel group)
  int q_tp = 0;                 // tap of this lane's chunk (wave-uniform on the tap-inner walk)
  unsigned q_adel = 0, q_wdel = 0;   // byte deltas added to the per-row A offsets / weight-row offsets
  char* q_base = nullptr;
  auto issue_begin = [&](int ksl, int stage) {
    q_base = smem + stage * STAGE;
    if (tap_inner) {
      // everything that depends on the K step is wave-uniform here: one readlane for the tap, one scalar byte delta for all rows
      const int tp = __builtin_amdgcn_readfirstlane(i_tp), cgi = __builtin_amdgcn_readfirstlane(i_cg);
      q_kval = cgi < cgroups && ksl + ks0 < ks1 && cgi * 8 + cc < cpt;     // (lane term: partial last group when Ci % 64 != 0)
      const int tv = __builtin_amdgcn_readlane(tapreg, tp);
      const int dy = (int)(signed char)(tv & 0xff), dx = (int)(signed char)((tv >> 8) & 0xff);
      q_tp = tp;
      q_adel = (unsigned)((dy * p.Wi + dx) * (int)lda2) + ((unsigned)cgi << 7);           // tap shift + 64-channel group, bytes
      q_wdel = (unsigned)(tp * cpt + cgi * 8 + cc) << 4;                                      // weight-row byte offset of this chunk
      if (++i_tp == p.ntaps) { i_tp = 0; ++i_cg; }
    } else {
      const int ks = ksl + ks0;
      const int kc = ks * 8 + cc;
      const int tap = kc / cpt;
      q_kval = kc < kchunks && ks < ks1;
      const int tv = __builtin_amdgcn_ds_bpermute((tap & (DMX_MAX_TAPS - 1)) << 2, tapreg);
      const int dy = (int)(signed char)(tv & 0xff), dx = (int)(signed char)((tv >> 8) & 0xff);
      q_tp = tap & (DMX_MAX_TAPS - 1);
      q_adel = (unsigned)((dy * p.Wi + dx) * (int)lda2) + ((unsigned)(kc - tap * cpt) << 4) - ((unsigned)cc << 4);   // a_lin already holds cc * 16
      q_wdel = (unsigned)kc << 4;
    }
  };
  auto issue_one = [&](auto I) {
    constexpr int i = decltype(I)::value;
    if constexpr (i < A_ISS) {
      const bool ok = q_kval && ((a_mask2[i >> 1] >> ((i & 1) * 16 + q_tp)) & 1u);
      const unsigned voff = ok ? a_lin[i] + q_adel : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(q_base + (i * NW + wave) * 1024), 16, voff, 0, 0, 0);
    } else {
      constexpr int j = i - A_ISS;
      const unsigned voff = q_kval ? w_row + (unsigned)j * w_step + q_wdel : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(q_base + A_BYTES + (j * NW + wave) * 1024), 16, voff, 0, 0, 0);
    }
  };
  auto issue = [&](int ksl, int stage) {          // whole step at once (ring prologue)
    issue_begin(ksl, stage);
    static_for<0, PER>(issue_one);
  };

  const int wm = wave / WN, wn = wave - wm * WN;
  const int lr = lane & 15, lq = lane >> 4;
  f32x4 acc[FM][FN];
  // EPI_BIASINIT (set by the dispatcher in place of EPI_BIAS when nothing in the epilogue comes before the bias): the accumulators START
  // at the channel bias -- the same number of register moves as clearing them, the bias registers die before the K loop, and the
  // epilogue loses its bias term (2 us per 320 x 256 tile, profiles/r03_epilogue_ablation.log) -- loaded ahead of the ring's LDS-DMA
  // issues so that the in-order vmcnt of their first use does not wait for a tile
  if (p.flags & EPI_BIASINIT) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = tn * BN + wn * TN + j * 16 + lq * 4;
      const float4 bb = n < p.N ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // NSTAGE-deep ring: tiles ks+1 .. ks+NSTAGE-1 are in flight while tile ks is consumed.  Every iteration issues
  // exactly one stage (past the end the offsets are out of range -> zero fill, no memory traffic), so the vmcnt
  // counts are compile-time constants; the raw s_barrier keeps the younger loads in flight across it.
  const int nk = ks1 > ks0 ? ks1 - ks0 : 0;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s) issue(s, s);
  float2* s_ln = reinterpret_cast<float2*>(smem + NSTAGE * STAGE);                   // LNF: behind the ring (launch_glds_t sizes it)
  float* s_cs = reinterpret_cast<float*>(smem + NSTAGE * STAGE + BM * 8);
  if constexpr (LNF) ln_prologue<BM, BN, NW * 64>(p, m0, tn * BN, s_ln, s_cs);       // (under the latency of the ring's first tiles)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
  __builtin_amdgcn_s_barrier();
  DMX_GSTAMP(1);

  // Hand-scheduled fragment pipeline.  hipcc sinks every ds_read next to its consumer (read 2, wait, 4-8 MFMAs: the LDS
  // latency is exposed 16x per K-step and the matrix pipe idles ~50 %).  Here the reads are inline asm issued two steps
  // (2*FN MFMAs) ahead of their use, with counted lgkmcnt waits computed at compile time from the issue order (LDS
  // returns in order), and sched_barrier fences so the MFMAs of a step cannot be hoisted above its wait.
  //
  // Stagger (8-wave tiles: two waves per SIMD that run the same program with one barrier per K step): left alone the two
  // partners move in lockstep -- both wait for their first fragments behind the barrier, both compute addresses, both issue
  // LDS-DMA at the same time, and the matrix pipe idles through all of it.  The second-dispatched half of the workgroup
  // (waves NW/2 ..) therefore runs up to HALF A K STEP BEHIND: between two barriers it first issues the last DEF fragment steps
  // (all kk = 1) of the previous K step -- their fragments were read, and waited for, before the barrier and stay in registers
  // across it -- with the LDS-DMA issues and the new stage's first reads underneath, then the first NS - DEF steps of the
  // current K step, during which it also reads the fragments it will carry across the next barrier.  Its partner's exposed
  // latency behind the barrier is covered by the carried MFMAs and vice versa.  Per-wave accumulation order is unchanged:
  // results are bit-identical to the unstaggered schedule (MI355X_MICROARCH.md, two waves per SIMD, item 9).
  constexpr int NS = 2 * FM;                                   // steps per K-step: (kk, i), FN MFMAs each
  constexpr int P0 = FN + 2;                                   // prologue reads: wf0[0..FN), af[0], af[1]
  // DEF = fragment steps the lagging half carries across the barrier: half a K step where the registers allow it (4 * DEF +
  // 4 * FN carried VGPRs next to the FM * FN * 4 accumulators), DMX_DEF_BIG on the register-bound 320-row tile
  constexpr bool ROOMY = FM * FN * 4 + 4 * FM + 8 * FN + 64 <= 256;
  constexpr bool STAGGER = NW == 8 && FM >= 2 && (ROOMY || DMX_DEF_BIG > 0);
  constexpr int DEF = ROOMY ? FM : (DMX_DEF_BIG > 0 ? (DMX_DEF_BIG < FM ? DMX_DEF_BIG : FM) : 1);
  constexpr int LOWN = NS - DEF;                               // steps of the current K step the lagging half runs before the barrier
  static_assert(DEF <= FM && DEF >= 1, "carried steps must all be kk = 1 steps");
  frag8_t wf0[FN], wf1[FN], af[NS];
#define DMX_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  auto kloop = [&](auto LAGT) {
    constexpr bool LAG = decltype(LAGT)::value;
    int cur = 0, nxt = NSTAGE - 1;
    auto mfma_step = [&wf0, &wf1, &af, &acc](auto ST) {
      constexpr int st = decltype(ST)::value;
      constexpr int kk = st / FM, i = st - kk * FM;
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = DMX_MFMA16(kk ? wf1[j] : wf0[j], af[st], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int ks = 0; ks < nk; ++ks) {
      const char* sa = smem + cur * STAGE + (wm * TM + lr) * 128;
      const char* sb = smem + cur * STAGE + A_BYTES + (wn * TN + lr) * 128;
      const unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(sa);
      const unsigned b0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(sb);
      const unsigned sw0 = ((0 * 4 + lq) ^ (lr & 7)) << 4, sw1 = ((1 * 4 + lq) ^ (lr & 7)) << 4;
      const unsigned aA0 = a0 + sw0, aA1 = a0 + sw1, aB0 = b0 + sw0, aB1 = b0 + sw1;
      __builtin_amdgcn_sched_barrier(0);
      // (register-bound tiles: the lagging half reads its first fragments only AFTER the carried MFMAs, whose operands then die first)
      constexpr bool P0_LATE = LAG && DEF < FM;
      auto prologue_reads = [&wf0, &af, aA0, aA1, aB0]() {
        static_for<0, FN>([&wf0, aB0](auto J) { constexpr int j = decltype(J)::value; DMX_DSR(wf0[j], aB0, j * 2048); });
        DMX_DSR(af[0], aA0, 0);
        if constexpr (FM > 1) { DMX_DSR(af[1], aA0, 2048); } else { DMX_DSR(af[1], aA1, 0); }
      };
      if constexpr (!P0_LATE) prologue_reads();
      // reads of fragment step st: af of step st + 2; the kk = 1 weights trickle in during kk = 0; the lagging half's last DEF - 2
      // own steps also fetch af[LOWN + 2 ..] (every fragment it carries must be in registers before the barrier)
      auto read_step = [&wf1, &af, aA0, aA1, aB1](auto ST) {
        constexpr int st = decltype(ST)::value;
        if constexpr (st + 2 < NS) {
          constexpr int s2 = st + 2, k2 = s2 / FM, i2 = s2 - k2 * FM;
          if constexpr (k2 == 0) { DMX_DSR(af[s2], aA0, i2 * 2048); } else { DMX_DSR(af[s2], aA1, i2 * 2048); }
        }
        if constexpr (LAG && st >= LOWN - (DEF - 2) && st < LOWN) {
          constexpr int s3 = st + DEF, i3 = s3 - FM;
          DMX_DSR(af[s3], aA1, i3 * 2048);
        }
        constexpr int w_lo = st < FM ? st * FN / FM : 0, w_hi = st < FM ? (st + 1) * FN / FM : 0;
        static_for<w_lo, w_hi>([&wf1, aB1](auto J) { constexpr int j = decltype(J)::value; DMX_DSR(wf1[j], aB1, j * 2048); });
      };
      // issue-order bookkeeping (all constexpr): reads issued through a step and the newest one the step depends on
      auto wait_step = [](auto ST) {
        constexpr int st = decltype(ST)::value;
        constexpr auto r = [](int t) {
          const int trickle = t < FM ? (t + 1) * FN / FM - t * FN / FM : 0;
          return (t + 2 < NS ? 1 : 0) + (LAG && t >= LOWN - (DEF - 2) && t < LOWN ? 1 : 0) + trickle;
        };
        constexpr auto issued_through = [r](int t) { int n = P0; for (int q = 0; q <= t; ++q) n += r(q); return n; };
        constexpr int total = issued_through(st);
        constexpr int pos_af = st == 0 ? FN + 1 : (st == 1 ? FN + 2 : issued_through(st - 3 < 0 ? -1 : st - 3) + 1);
        constexpr int pos_w = st < FM ? FN : issued_through(FM - 1);
        constexpr int need = pos_af > pos_w ? pos_af : pos_w;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(total - need) : "memory");
        __builtin_amdgcn_sched_barrier(0);
      };
      // The address arithmetic of tile ks + NSTAGE - 1 runs right behind the prologue fragment reads (under their latency).
      __builtin_amdgcn_sched_barrier(0);
      issue_begin(ks + NSTAGE - 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      // the PER LDS-DMA instructions of the step are slipped in behind the MFMA groups of SPREAD fragment steps
      constexpr int SPREAD = NS / 2 > 0 ? NS / 2 : 1;      // (all at the top of the step, or over 3 steps: 5 % slower on the 320-row tile)
      if constexpr (LAG) {
        // carried steps of the previous K step (fragments in registers since before the barrier), LDS-DMA issues underneath
        if (ks > 0) {
          static_for<LOWN, NS>([&](auto ST) {
            constexpr int st = decltype(ST)::value;
            mfma_step(ST);
            static_for<(st - LOWN) * PER / DEF, (st - LOWN + 1) * PER / DEF>(issue_one);
            __builtin_amdgcn_sched_barrier(0);
          });
        } else {
          static_for<0, PER>(issue_one);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (P0_LATE) { prologue_reads(); __builtin_amdgcn_sched_barrier(0); }
        static_for<0, LOWN>([&](auto ST) {
          read_step(ST);
          wait_step(ST);
          mfma_step(ST);
        });
      } else {
        static_for<0, NS>([&](auto ST) {
          constexpr int st = decltype(ST)::value;
          read_step(ST);
          wait_step(ST);
          mfma_step(ST);
          if constexpr (st < SPREAD) {
            static_for<st * PER / SPREAD, (st + 1) * PER / SPREAD>(issue_one);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      }
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(KEEP) : "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 == NSTAGE ? 0 : cur + 1;
      nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }
    if constexpr (LAG) {
      __builtin_amdgcn_sched_barrier(0);
      if (nk > 0) static_for<LOWN, NS>(mfma_step);     // the last K step's carried steps
    }
  };
  if constexpr (STAGGER) {
    if (wave >= NW / 2) kloop(std::true_type{}); else kloop(std::false_type{});
  } else {
    kloop(std::false_type{});
  }
#undef DMX_DSR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the zero-fill tail ...
  __builtin_amdgcn_s_barrier();                          // ... of every wave before the stage buffers are reused by the epilogue
  DMX_GSTAMP(2);
  if constexpr (LNF) ln_apply<FM, FN, BN>(acc, s_ln, s_cs, wm * TM, wn * TN, lr, lq);
  if ((p.flags & EPI_F32OUT) || ((p.N | p.ldc | p.ldr | p.ldx | p.ldc2) & 7)) {     // direct path: fp32 out or rows not 16-B granular
    gemm_epilogue<FM, FN>(p, acc, m0 + wm * TM, tn * BN + wn * TN, lr, lq, coff, HqWq);
  } else {
    constexpr int EPI_CH = EpiChunk<FM>::CH;
    constexpr int EPI_WAVE_BYTES = EPI_CH * (FN * 32 + 16) + EPI_CH * 12;
    static_assert(EPI_WAVE_BYTES * WM * WN <= 2 * (BM + BN) * 128, "epilogue staging does not fit the stage buffers");
    gemm_epilogue_lds<FM, FN, EM>(p, acc, m0 + wm * TM, tn * BN + wn * TN, lane, coff, HqWq, smem + wave * EPI_WAVE_BYTES, mlim);
  }
#ifdef DMX_GEMM_STAMPS
  DMX_GSTAMP(3);                                         // every store of the epilogue issued ...
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ... and acknowledged
  DMX_GSTAMP(4);
  {
    const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (tid == NW * 64 - 64 && wg < 8192) {              // lane 0 of the LAST wave (the lagging half: it leaves the K loop last)
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
#pragma unroll
      for (int i = 0; i < 5; ++i) g_gemm_stamps[wg * 6 + i] = gstamp_v[i];
      g_gemm_stamps[wg * 6 + 5] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    }
  }
#endif
}

// XCD-aware remap (blocks are dealt round-robin over the 8 XCDs): give each XCD a contiguous run of logical tiles so the
// N-tiles of one M-tile and neighbouring M-tiles (shared A rows / halos, same weights) hit the same L2.  Bijective for any grid.
__device__ __forceinline__ int xcd_remap(int bid) {
  const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int EM, bool LNF = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_glds_kernel(const GemmDesc p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x);
  int m0 = (bid / tiles_n) * BM, mlim = p.M;
  if constexpr (EM == 5 || EM == 6) {
    // GroupNorm partial sums are kept per wave tile and image.  Where the rows of an image are no multiple of the wave tile's, a flat M
    // tiling would cut every image's slots at another place -- a clip's statistics would depend on its position in the batch in the last
    // bits -- so the M tiles restart at every image (the image's last tile is partial; rows past its end are masked like rows past M)
    const int P = p.Hq * p.Wq;
    if (P % (BM / WM) != 0) {
      const int tpi = (P + BM - 1) / BM, tm = bid / tiles_n, b = tm / tpi;
      m0 = b * P + (tm - b * tpi) * BM;
      mlim = (b + 1) * P;
    }
  }
  glds_tile<BM, BN, WM, WN, NSTAGE, EM, LNF>(p, smem, m0, bid % tiles_n, mlim);
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int EM, bool LNF = false>
int launch_glds_t(const GemmDesc& d, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  constexpr int SMEM = NSTAGE * (BM + BN) * 128 + (LNF ? BM * 8 + BN * 8 : 0);       // (the tap table of this kernel lives in a register, not in LDS)
  static_assert(SMEM <= 160 * 1024, "tile ring exceeds the 160 KiB of LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<BM, BN, WM, WN, NSTAGE, EM, LNF>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  long long tiles = (long long)cdiv(d.M, BM) * cdiv(d.N, BN);
  if constexpr (EM == 5 || EM == 6) {                  // image-aligned M tiling (see gemm_glds_kernel)
    const int P = d.Hq * d.Wq;
    if (P % (BM / WM) != 0) tiles = (long long)(d.M / P) * cdiv(P, BM) * cdiv(d.N, BN);
  }
  dim3 grid((unsigned)tiles, (unsigned)d.Z, (unsigned)(d.ksplit > 1 ? d.ksplit : 1));
  static const bool bias_init = getenv("DMX_NO_BIAS_INIT") == nullptr;
  if (bias_init && (d.flags & EPI_BIAS) && d.bias && !(d.flags & (EPI_MASK | EPI_MASKBITS | EPI_SOFTBWD | EPI_LNFOLD))) {
    // nothing precedes the bias in the epilogue's order (mask -> bias -> residual -> alpha ...): start the accumulators at it instead
    GemmDesc q = d;
    q.flags = (q.flags & ~EPI_BIAS) | EPI_BIASINIT;
    hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, NSTAGE, EM, LNF>), grid, dim3(NT), SMEM, stream, q);
    return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, WM, WN, NSTAGE, EM, LNF>), grid, dim3(NT), SMEM, stream, d);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}
template <int BM, int BN, int WM, int WN, int EM, bool LNF = false>
int launch_cfg_t(const GemmDesc& d, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  constexpr int SMEM = 2 * (BM + BN) * 128 + DMX_MAX_TAPS * 4 + (LNF ? BM * 8 + BN * 8 : 0);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, WM, WN, EM, LNF>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  long long tiles = (long long)cdiv(d.M, BM) * cdiv(d.N, BN);
  if constexpr (EM == 5 || EM == 6) {                  // image-aligned M tiling (see gemm_kernel)
    const int P = d.Hq * d.Wq;
    if (P % (BM / WM) != 0) tiles = (long long)(d.M / P) * cdiv(P, BM) * cdiv(d.N, BN);
  }
  dim3 grid((unsigned)tiles, (unsigned)d.Z, 1);
  static const bool bias_init = getenv("DMX_NO_BIAS_INIT") == nullptr;
  if (bias_init && (d.flags & EPI_BIAS) && d.bias && !(d.flags & (EPI_MASK | EPI_MASKBITS | EPI_SOFTBWD | EPI_LNFOLD))) {
    GemmDesc q = d;                           // same rule as launch_glds_t: every kernel of the family rounds alike
    q.flags = (q.flags & ~EPI_BIAS) | EPI_BIASINIT;
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, EM, LNF>), grid, dim3(NT), SMEM, stream, q);
    return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, EM, LNF>), grid, dim3(NT), SMEM, stream, d);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}
}  // namespace
