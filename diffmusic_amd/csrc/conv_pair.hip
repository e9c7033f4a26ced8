// Fused pair of stride-1 dilated 1-D convolutions for the narrow (C = 32 / 64 / 128) HiFi-GAN resblock stages.
//
// A HiFi-GAN resblock step is conv1 -> leaky-relu -> conv2 -> (+ residual)  (transformers
// modeling_speecht5.py HifiGanResidualBlock.forward; reached from the reference through
// BaseOperator.inverse_transform, diffmusic/inverse_problem/operator.py:126-130), and its input-gradient
// is dgrad(conv2) -> leaky-relu' -> dgrad(conv1) -> (+ residual).  At C <= 64 each convolution moves
// 2.5 tensors through HBM for a few hundred FLOP per byte: as separate implicit-GEMM launches these layers
// ran at 2-3x their HBM time (profiles/r01_gemm_shapes_v2.csv).  Here one workgroup
//   1. loads the time slab it needs (256 + halo rows, channels-last fp16) into LDS once,
//   2. runs the first convolution out of LDS (every tap is a row-shifted view of the same slab, so the
//      im2col re-reads of the generic kernel disappear), applies its pointwise tail and leaves the 256-row
//      intermediate in LDS (forward also streams it to HBM for the tape; backward never stores it),
//   3. runs the second convolution from that intermediate, and
//   4. finishes with the shared LDS-staged epilogue (residual / averaging / leaky-relu of the stored tensor).
// HBM traffic per pair drops from 5 to 3 tensors (forward) and from 7 to 4 (backward).
// The weights of one tap (C x C) stream through a 3-stage LDS ring, prefetched two taps ahead.
// MFMA operand roles are the same as in gemm_conv.hip (weight = A operand), so the accumulator layout and
// the epilogue are shared.  The same kernel with `single` set is a plain slab convolution.
#include "gemm_epilogue.h"
#include "conv_pair.h"

namespace {

constexpr int PAIR_HALO = 50;            // max (lo + hi) halo rows of one stage
constexpr int PAIR_ROWS = 256;           // intermediate rows per workgroup
constexpr unsigned PAIR_OOB = 0x80000000u;

struct PairParams {
  GemmDesc a, b;
  int T, nb, BMo;                        // frames per clip, workgroups per clip, output rows per workgroup
  int loA, hiA, loB, hiB;
  int off0A, dA, off0B, dB;              // slab row read by tap j of a stage: off0 + j * d  (taps are affine: j*dil - pad or pad - j*dil)
  int single;
  int r_from_slab;                       // stage B's residual is stage A's input: take it from the LDS slab, not from HBM
  int a_tape_bits_only;                  // forward: the activated intermediate leaves the kernel as sign bits only (a.B2), not as a tensor
};

#ifdef DMX_PAIR_STAMPS
// diagnostic build only (scripts/dev/r03_pair_stamps.sh): 100 MHz wall-clock stamps of the phases of the first 4096 workgroups
__device__ unsigned long long g_pair_stamps[4096 * 8];
// (kept in registers and written once at the end: a stamp stored on the spot is a global store the next compiler-inserted vmcnt(0) waits for)
#define DMX_STAMP(i) do { stamp_v[i] = wall_clock64(); } while (0)
#else
#define DMX_STAMP(i) do { } while (0)
#endif

template <int C>
struct PairCfg {
  // +32 B: a ds_read_b128 lane group is 8 rows at one k-quarter plus 8 other rows at the next (MI355X_MICROARCH.md, LDS);
  // with a pitch of 2C+32 bytes (24 / 40 / 72 dwords) those 16 four-dword spans hit 16 different bank quads
  static constexpr int PITCH = 2 * C + 32;
  static constexpr int CPR = C / 8;                // 16-byte chunks per row
  static constexpr int WN = C > 64 ? C / 64 : 1;   // column groups of 64 channels (C = 128: two), one wave each per row group
  static constexpr int NW = 4 * WN;                // waves: 4 row groups of 64 rows x WN column groups
  static constexpr int NT = NW * 64;
  static constexpr int FN = C / 16 / WN;           // 16-wide n tiles per wave
  static constexpr int SLAB_ROWS = PAIR_ROWS + PAIR_HALO;
  static constexpr int SLAB_BYTES = SLAB_ROWS * PITCH;
  static constexpr int TILE = C * 128;             // weights of one K step: C rows x 64 k, 128-byte rows, XOR-swizzled chunks
  static constexpr int NS = C == 128 ? 4 : 3;      // weight ring slots (C = 128: four, one barrier per TWO K steps; see run_stage)
  static constexpr int PER = TILE / 1024 / NW;     // LDS-DMA instructions per wave per K step
  static constexpr int SLAB_IT = (SLAB_ROWS * CPR + NT - 1) / NT;
  static constexpr int BITS_IT = PAIR_ROWS * CPR / NT;
  static constexpr int BITS_BYTES = PAIR_ROWS * CPR;
  static constexpr int BIAS_BYTES = 2 * C * 4;     // both stages' channel biases (fp32): read from LDS wherever a stage needs them (see pair_body)
  static constexpr int LDS_BYTES = SLAB_BYTES + NS * TILE + BITS_BYTES + BIAS_BYTES;
  static constexpr int MIN_WAVES = C == 32 ? 3 : 2;   // waves per SIMD the register allocation must leave room for (what the LDS footprint allows)
  static_assert(PER >= 1 && PAIR_ROWS * CPR % NT == 0, "tile / thread-count mismatch");
};

template <int FN>
struct PairFrags { frag8_t x[2][4]; frag8_t w[2][FN]; };

#define DMX_PAIR_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// LDS accesses of the tail between the two stages, as inline asm for the same reason as the fragment reads: while stage B's first weight
// tiles are in flight (LDS-DMA issued in stage A's last steps) every LDS access the compiler can SEE is preceded by an s_waitcnt vmcnt(0)
// -- the tail (residual pieces out of the slab, the intermediate written over it, the mask bits, the tape read-back) then started with
// the full latency of that prefetch exposed.  The asm forms carry their own lgkmcnt waits; the slab and the ring are disjoint.
#define DMX_LDS_ADDR(p) ((unsigned)(size_t)(__attribute__((address_space(3))) const char*)(p))
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));      // (register-class operands of the asm forms: HIP's uint2 / uint4 are structs)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pair_lds_w64(const void* p, uint2 v) {
  const u32x2_t w = {v.x, v.y};
  asm volatile("ds_write_b64 %0, %1" ::"v"(DMX_LDS_ADDR(p)), "v"(w) : "memory");
}
#define DMX_PAIR_DSR64(dst, p) asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(DMX_LDS_ADDR(p)))
#define DMX_PAIR_DSRU8(dst, p) asm volatile("ds_read_u8 %0, %1" : "=v"(dst) : "v"(DMX_LDS_ADDR(p)))
__device__ __forceinline__ void pair_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// waits for outstanding LDS reads and ties up to 16 destination registers so that no consumer is scheduled above the wait
template <typename T>
__device__ __forceinline__ void pair_tie8(T (&r)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : : "memory");
}
template <typename T>
__device__ __forceinline__ void pair_tie16(T (&r)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]),
                 "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
               :
               : "memory");
}
template <typename T, int N>
__device__ __forceinline__ void pair_tie(T (&r)[N]) {
  static_assert(N == 8 || N == 16, "4 row fragments x 2 or 4 column fragments");
  if constexpr (N == 8) pair_tie8(r); else pair_tie16(r);
}

// the fragment reads are inline asm (the compiler would otherwise sink them next to their MFMAs and put a vmcnt(0) in front
// of every LDS read that may alias an LDS-DMA target); this wait ties the registers so no consumer is scheduled above it
__device__ __forceinline__ void pair_settle(PairFrags<4>& f) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f.x[0][0]), "+v"(f.x[0][1]), "+v"(f.x[0][2]), "+v"(f.x[0][3]), "+v"(f.x[1][0]), "+v"(f.x[1][1]), "+v"(f.x[1][2]),
                 "+v"(f.x[1][3]), "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.w[0][2]), "+v"(f.w[0][3]), "+v"(f.w[1][0]), "+v"(f.w[1][1]),
                 "+v"(f.w[1][2]), "+v"(f.w[1][3])
               :
               : "memory");
}
__device__ __forceinline__ void pair_settle(PairFrags<2>& f) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f.x[0][0]), "+v"(f.x[0][1]), "+v"(f.x[0][2]), "+v"(f.x[0][3]), "+v"(f.x[1][0]), "+v"(f.x[1][1]), "+v"(f.x[1][2]),
                 "+v"(f.x[1][3]), "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.w[1][0]), "+v"(f.w[1][1])
               :
               : "memory");
}

// one workgroup of the fused pair: `bid` of `nwg` workgroups of the launch (or of this problem's share of a grouped launch)
template <int C>
__device__ __forceinline__ void pair_body(const PairParams& P, char* smem, int bid, const int nwg) {
  using K = PairCfg<C>;
  constexpr int PITCH = K::PITCH, CPR = K::CPR, FN = K::FN, NS = K::NS, PER = K::PER, TILE = K::TILE, NT = K::NT, NW = K::NW, WN = K::WN;
  char* ring = smem;                                   // 1 KiB-aligned LDS-DMA targets first
  char* slab = smem + NS * TILE;
  unsigned char* s_bits = reinterpret_cast<unsigned char*>(slab + K::SLAB_BYTES);
  float* s_bias = reinterpret_cast<float*>(slab + K::SLAB_BYTES + K::BITS_BYTES);     // [2][C]: stage A's, stage B's

  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave / WN, cg = wave - rg * WN;       // row group (64 intermediate rows) and column group (64 channels) of this wave
#ifdef DMX_PAIR_STAMPS
  const int stamp_id = bid;
  unsigned long long stamp_v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DMX_STAMP(0);
  {  // XCD-aware remap: neighbouring time tiles (shared halos, same weights) land on the same L2
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int T = P.T;
  const int b = bid / P.nb, tq = bid - b * P.nb;
  const int t0 = tq * P.BMo;
  const bool single = P.single != 0;
  const int kA = single ? 0 : P.a.ntaps, kB = P.b.ntaps;
  const int stepsA = (kA * C + 63) / 64, stepsB = (kB * C + 63) / 64, total = stepsA + stepsB;

  // ---- weights: one 64-deep K step (C rows x 128 B) per LDS-DMA tile, three tiles ahead of the MFMAs
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<act_t*>(single ? P.b.W : P.a.W), 0, PAIR_OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<act_t*>(P.b.W), 0, PAIR_OOB, 0x00020000);
  const int lrow = lane >> 3, cc = (lane & 7) ^ lrow;   // lane fetches logical 16-B chunk cc of tile row (8j + lrow): swizzled image
  auto issue_dma = [&](int g) {
    const bool inA = g < stepsA;
    const int sidx = inA ? g : g - stepsA;
    const int Kel = inA ? P.a.K : P.b.K;
    const unsigned ldw2 = (unsigned)(inA ? P.a.ldw : P.b.ldw) * 2u;
    const int kel = sidx * 64 + cc * 8;
    const bool ok = g < total && kel < Kel;
    char* dst = ring + (g % NS) * TILE;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int j = i * NW + wave;
      const unsigned voff = ok ? (unsigned)(j * 8 + lrow) * ldw2 + (unsigned)kel * 2u : PAIR_OOB;
      if (inA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, voff, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, voff, 0, 0, 0);
    }
  };
  issue_dma(0);
  issue_dma(1);
  issue_dma(2);

  // ---- phase 1: time slab (zero outside the clip) and the leaky-relu' sign bits of the stage-A mask
  {
    const act_t* src = single ? P.b.A : P.a.A;
    const int nrows = PAIR_ROWS + (single ? P.loB + P.hiB : P.loA + P.hiA);
    const int tfirst = t0 - P.loB - (single ? 0 : P.loA);
    uint4 sv[K::SLAB_IT];
#pragma unroll
    for (int it = 0; it < K::SLAB_IT; ++it) {
      const int c = tid + it * NT, row = c / CPR, piece = c % CPR, t = tfirst + row;
      sv[it] = make_uint4(0, 0, 0, 0);
      if (row < nrows && t >= 0 && t < T) sv[it] = *reinterpret_cast<const uint4*>(src + ((long long)b * T + t) * C + piece * 8);
    }
    const bool masked = !single && (P.a.flags & EPI_MASK);
    const bool maskbits = !single && (P.a.flags & EPI_MASKBITS);
    // sign-bit mask source (1 byte per 8 channels, the layout s_bits uses): a plain copy of PAIR_ROWS x CPR bytes
    constexpr int BPT = K::BITS_BYTES / NT;             // bytes per thread: 4 (C = 32) or 8 (C = 64 / 128), never crossing a row
    static_assert(BPT == 4 || BPT == 8, "bit-mask copy granularity");
    static_assert(CPR % BPT == 0, "a thread's mask bytes must stay inside one row");
    uint32_t mb[2] = {0u, 0u};
    if (maskbits) {
      const int c = tid * BPT, row = c / CPR, piece = c % CPR, t = t0 - P.loB + row;
      if (t >= 0 && t < T) {
        const unsigned char* src = P.a.XB + ((long long)b * T + t) * P.a.ldxb + piece;
        mb[0] = *reinterpret_cast<const uint32_t*>(src);
        if (BPT == 8) mb[1] = *reinterpret_cast<const uint32_t*>(src + 4);
      }
    }
#pragma unroll
    for (int it = 0; it < K::SLAB_IT; ++it) {
      const int c = tid + it * NT, row = c / CPR, piece = c % CPR;
      if (row < K::SLAB_ROWS) *reinterpret_cast<uint4*>(slab + row * PITCH + piece * 16) = sv[it];
    }
    if (masked) {           // 16-bit mask source (generic callers; the HiFi-GAN executor passes sign bits): reduced to bits on the fly,
#pragma unroll 1            // one load at a time -- holding all of them next to the slab registers cost the C = 64 instance a wave per SIMD
      for (int it = 0; it < K::BITS_IT; ++it) {
        const int c = tid + it * NT, row = c / CPR, piece = c % CPR, t = t0 - P.loB + row;
        uint4 m = make_uint4(0, 0, 0, 0);
        if (t >= 0 && t < T) m = *reinterpret_cast<const uint4*>(P.a.X + ((long long)b * T + t) * C + piece * 8);
        s_bits[c] = (unsigned char)dmx_pos8(m);
      }
    }
    if (maskbits) {
      uint32_t* d = reinterpret_cast<uint32_t*>(s_bits + tid * BPT);
      d[0] = mb[0];
      if (BPT == 8) d[1] = mb[1];
    }
  }
  // both stages' channel biases -> LDS, under the slab's round trip: a stage that needs its bias while LDS-DMA or tape stores are in flight
  // would otherwise wait for ALL of them (vmcnt counts in order) -- stage B's accumulator start used to wait for the tape's store
  // acknowledgements and the weight prefetch, the backward tail for the prefetch
  if (tid < 2 * C) {
    const GemmDesc& d = tid < C ? P.a : P.b;
    const int ch = tid < C ? tid : tid - C;
    s_bias[tid] = (!(single && tid < C) && d.bias && (d.flags & (EPI_BIAS | EPI_BIASINIT))) ? d.bias[ch] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  DMX_STAMP(1);

  f32x4 acc[4][FN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < FN; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // EPI_BIASINIT: the stage's accumulators start at its channel bias (the bias registers die here, before the K loop)
  auto init_acc = [&](const GemmDesc& d) {
    if (C < 128 && (d.flags & EPI_BIASINIT)) {
      f32x4 bb[FN];
#pragma unroll
      for (int n = 0; n < FN; ++n)
        asm volatile("ds_read_b128 %0, %1" : "=v"(bb[n]) : "v"(DMX_LDS_ADDR(s_bias + (&d == &P.b ? C : 0) + cg * 64 + n * 16 + lq * 4)));
      if constexpr (FN == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb[0]), "+v"(bb[1]) : : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb[0]), "+v"(bb[1]), "+v"(bb[2]), "+v"(bb[3]) : : "memory");
#pragma unroll
      for (int n = 0; n < FN; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][n] = bb[n];
    } else {
      zero_acc();
    }
  };
  using Frags = PairFrags<FN>;
  const unsigned xlane = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(slab + (rg * 64 + lr) * PITCH + lq * 16);
  const unsigned wlane0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(ring + (cg * 64 + lr) * 128 + (((0 * 4 + lq) ^ (lr & 7)) << 4));
  const unsigned wlane1 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(ring + (cg * 64 + lr) * 128 + (((1 * 4 + lq) ^ (lr & 7)) << 4));
  // fragments of K step s of a stage (global step g): x rows shifted by the tap, weights from ring tile g
  auto load_frags = [&](Frags& f, bool isB, int sidx, int g) {
    const int off0 = isB ? P.off0B : P.off0A, dd = isB ? P.dB : P.dA, kk = isB ? kB : kA;
    unsigned xa0, xa1;
    if constexpr (C >= 64) {
      constexpr int SPT = C / 64;                                    // K steps per tap (C = 128: one per 64-channel half)
      const int tp = sidx / SPT, kh = sidx - tp * SPT;
      xa0 = xlane + (unsigned)((off0 + tp * dd) * PITCH + kh * 128);
      xa1 = xa0 + 64u;
    } else {
      const int tp0 = 2 * sidx, tp1 = (2 * sidx + 1 < kk) ? 2 * sidx + 1 : kk - 1;   // a padded tap has zero weights: any valid rows
      xa0 = xlane + (unsigned)((off0 + tp0 * dd) * PITCH);
      xa1 = xlane + (unsigned)((off0 + tp1 * dd) * PITCH);
    }
    const unsigned wo = (unsigned)((g % NS) * TILE);
    const unsigned wa0 = wlane0 + wo, wa1 = wlane1 + wo;
    static_for<0, 4>([&f, xa0, xa1](auto I) {
      constexpr int i = decltype(I)::value;
      DMX_PAIR_DSR(f.x[0][i], xa0, i * 16 * PITCH);
      DMX_PAIR_DSR(f.x[1][i], xa1, i * 16 * PITCH);
    });
    static_for<0, FN>([&f, wa0, wa1](auto N) {
      constexpr int n = decltype(N)::value;
      DMX_PAIR_DSR(f.w[0][n], wa0, n * 2048);
      DMX_PAIR_DSR(f.w[1][n], wa1, n * 2048);
    });
  };
  auto mfma_half = [&](const Frags& f, int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < FN; ++n) acc[i][n] = DMX_MFMA16(f.w[ks][n], f.x[ks][i], acc[i][n]);
  };
  // one K step: the next step's fragments and the weight tile three steps ahead are in flight under this step's MFMAs.  The
  // LDS-DMA goes out between the two MFMA halves (slot g % NS is free since the barrier that ended step g - 1): next to the burst
  // of fragment reads at the top of the step its issue cost the wave 2-3x as much
  auto step = [&](Frags& cur, Frags& nxt, bool isB, int sidx, int n, int g) {
    const bool more = sidx + 1 < n;
    if (more) load_frags(nxt, isB, sidx + 1, g + 1);
    mfma_half(cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    issue_dma(g + 3);
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(cur, 1);
    if (more) pair_settle(nxt);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");      // tile g+2 has landed (tile g+3 may still be in flight)
    __builtin_amdgcn_s_barrier();
  };
  auto run_stage = [&](bool isB, int n, int g0) {
    Frags fa, fb;
    load_frags(fa, isB, 0, g0);
    pair_settle(fa);
    // the first step below re-fills ring tile g0 % NS (weights of step g0 + 3) right away: every wave must have taken its step-g0
    // fragments out of it first.  (Inside the loop the end-of-step barrier gives this guarantee; without this one ~2 % of the
    // outputs changed from run to run.)
    __builtin_amdgcn_s_barrier();
    if constexpr (C == 128) {
      // C = 128 (one 8-wave workgroup per CU): K steps run in PAIRS with one barrier per pair.  The ring has four slots: while a pair
      // (g, g + 1) reads the fragments of tiles g + 1 and g + 2, the weights of tiles g + 3 and g + 4 are fetched into the slots of
      // tiles g - 1 and g (both fully read before the barrier that started the pair); both fetches go out in the FIRST step of the pair
      // and must have landed at the barrier that ends it (vmcnt(0): more than a whole K step of latency cover).  Half the barriers,
      // half the all-waves-wait points of the one-barrier-per-step schedule the narrower instances keep.
      int sidx = 0;
      for (; sidx + 1 < n; sidx += 2) {
        const int g = g0 + sidx;
        load_frags(fb, isB, sidx + 1, g + 1);
        mfma_half(fa, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue_dma(g + 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(fa, 1);
        __builtin_amdgcn_sched_barrier(0);
        issue_dma(g + 4);
        __builtin_amdgcn_sched_barrier(0);
        pair_settle(fb);
        const bool more = sidx + 2 < n;
        if (more) load_frags(fa, isB, sidx + 2, g + 2);
        mfma_half(fb, 0);
        mfma_half(fb, 1);
        if (more) pair_settle(fa);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // tiles g + 3 and g + 4 have landed
        __builtin_amdgcn_s_barrier();
      }
      if (sidx < n) {                                                    // odd tail: a single step, its one fetch awaited in full
        mfma_half(fa, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue_dma(g0 + sidx + 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(fa, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else {
      for (int sidx = 0; sidx < n; sidx += 2) {
        step(fa, fb, isB, sidx, n, g0 + sidx);
        if (sidx + 1 < n) step(fb, fa, isB, sidx + 1, n, g0 + sidx + 1);
      }
    }
  };

  uint2 rpre[4 * FN];
  unsigned tape_bits[2] = {0u, 0u};       // sign bytes of the activated intermediate (forward tape), one per 16-byte chunk this thread read back
  bool tape_bits_valid = false;
#pragma unroll
  for (int i = 0; i < 4 * FN; ++i) rpre[i] = make_uint2(0, 0);
  if (!single) {
    // ---- stage A: 256 intermediate rows, row i <-> t = t0 - loB + i
    init_acc(P.a);
    run_stage(false, stepsA, 0);
    DMX_STAMP(2);
    // stage A's channel bias: fetched ONCE, here, for all row fragments.  Inside the tail loop below every iteration paid its own L2
    // round trip (hipcc puts an s_waitcnt vmcnt(0) in front of each LDS access of a kernel that uses LDS-DMA, so a load issued in one
    // iteration could not overlap the next): 8-16 serial round trips = 2.6-3.9 us of a 15-25 us workgroup (in-kernel stamps,
    // scripts/dev/r03_pair_stamps.py)
    const int fa = P.a.flags;
    float4 abias[FN];
    {
      f32x4 ab[FN];
#pragma unroll
      for (int n = 0; n < FN; ++n) asm volatile("ds_read_b128 %0, %1" : "=v"(ab[n]) : "v"(DMX_LDS_ADDR(s_bias + cg * 64 + n * 16 + lq * 4)));
      if constexpr (FN == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ab[0]), "+v"(ab[1]) : : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ab[0]), "+v"(ab[1]), "+v"(ab[2]), "+v"(ab[3]) : : "memory");
#pragma unroll
      for (int n = 0; n < FN; ++n) abias[n] = (fa & EPI_BIAS) ? make_float4(ab[n][0], ab[n][1], ab[n][2], ab[n][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (P.r_from_slab) {
      // the residual of stage B is the input of stage A: output row r (t = t0 + r) is slab row r + loA + loB; keep this lane's
      // accumulator-layout pieces in registers before the slab is overwritten by the intermediate
      u32x2_t rp[4 * FN];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < FN; ++n)
          DMX_PAIR_DSR64(rp[i * FN + n], slab + (rg * 64 + i * 16 + lr + P.loA + P.loB) * PITCH + (cg * 64 + n * 16 + lq * 4) * 2);
      pair_tie(rp);
#pragma unroll
      for (int k = 0; k < 4 * FN; ++k) rpre[k] = make_uint2(rp[k][0], rp[k][1]);
      // these rows reach loA + loB rows into the NEXT row group's region, which that group's waves overwrite below:
      // everybody must have taken its residual before anybody stores the intermediate
      __builtin_amdgcn_s_barrier();
    }
    DMX_STAMP(6);
    // pointwise tail of stage A, written over the (now dead) input slab; rows outside the clip are the zero padding of stage B
    const float aslope = P.a.act_slope, mslope = P.a.mask_slope;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rg * 64 + i * 16 + lr;
      const int t = t0 - P.loB + row;
      const bool inside = t >= 0 && t < T;
      // the mask bytes of this row fragment: FN LDS reads in flight, one wait (asm: see pair_lds_w64)
      unsigned mbits[FN];
#pragma unroll
      for (int n = 0; n < FN; ++n) mbits[n] = 0u;
      if (fa & (EPI_MASK | EPI_MASKBITS)) {
#pragma unroll
        for (int n = 0; n < FN; ++n) DMX_PAIR_DSRU8(mbits[n], s_bits + row * CPR + ((cg * 64 + n * 16 + lq * 4) >> 3));
        if constexpr (FN == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mbits[0]), "+v"(mbits[1]) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mbits[0]), "+v"(mbits[1]), "+v"(mbits[2]), "+v"(mbits[3]) : : "memory");
      }
#pragma unroll
      for (int n = 0; n < FN; ++n) {
        const int ch = cg * 64 + n * 16 + lq * 4;
        float v[4] = {acc[i][n][0], acc[i][n][1], acc[i][n][2], acc[i][n][3]};
        if (fa & (EPI_MASK | EPI_MASKBITS)) {
          const unsigned bits = mbits[n] >> (ch & 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= ((bits >> e) & 1u) ? 1.f : mslope;
        }
        {
          const float4 bb = abias[n];
          v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
        if (fa & EPI_LRELU2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * aslope;
        }
        if (!inside) { v[0] = v[1] = v[2] = v[3] = 0.f; }
        pair_lds_w64(slab + row * PITCH + ch * 2, make_uint2(pack2a(v[0], v[1]), pack2a(v[2], v[3])));
      }
    }
    pair_lds_wait();
    __builtin_amdgcn_s_barrier();
    DMX_STAMP(7);
    if (fa & EPI_LRELU2) {   // forward: the activated intermediate is part of the tape -> HBM, owned rows only
      // backward only needs its SIGN (the leaky-relu' mask): with EPI_BITS2 one byte per 8 channels goes out (a.B2) and the
      // 16-bit tensor itself is written only when a caller still wants it (a.C2 non-null and not bits-only)
      const bool bits = (fa & EPI_BITS2) != 0, full = !P.a_tape_bits_only;
      // all LDS reads first, then all global stores (interleaved, each read waited for the previous iteration's store to complete:
      // the same compiler-inserted vmcnt(0) as above)
      u32x4_t tq[K::BITS_IT];
#pragma unroll
      for (int it = 0; it < K::BITS_IT; ++it) {
        const int c = tid + it * NT, row = c / CPR, piece = c % CPR;
        asm volatile("ds_read_b128 %0, %1" : "=v"(tq[it]) : "v"(DMX_LDS_ADDR(slab + row * PITCH + piece * 16)));
      }
      static_assert(K::BITS_IT == 4 || K::BITS_IT == 8, "tape read-back granularity");
      if constexpr (K::BITS_IT == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]) : : "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]), "+v"(tq[4]), "+v"(tq[5]), "+v"(tq[6]), "+v"(tq[7]) : : "memory");
      uint4 tv[K::BITS_IT];
#pragma unroll
      for (int it = 0; it < K::BITS_IT; ++it) tv[it] = make_uint4(tq[it][0], tq[it][1], tq[it][2], tq[it][3]);
#pragma unroll
      for (int it = 0; it < K::BITS_IT; ++it) {
        const int c = tid + it * NT, row = c / CPR, piece = c % CPR, t = t0 - P.loB + row;
        if (row >= P.loB && row < P.loB + P.BMo && t < T) {
          if (full) *reinterpret_cast<uint4*>(P.a.C2 + ((long long)b * T + t) * C + piece * 8) = tv[it];
        }
        // the sign bytes wait in two registers and leave at the very end of the kernel: stored here, stage B's counted vmcnt waits (the
        // counter is in order) sat on their acknowledgements
        if constexpr (C < 128) {
          if (bits) tape_bits[it >> 2] |= (dmx_pos8(tv[it]) & 0xffu) << (8 * (it & 3));
        } else {                           // (C = 128 is at its 256 registers: its bytes leave here)
          if (bits && row >= P.loB && row < P.loB + P.BMo && t < T) P.a.B2[((long long)b * T + t) * P.a.ldb2 + piece] = (unsigned char)dmx_pos8(tv[it]);
        }
      }
      tape_bits_valid = bits && C < 128;
    }
  }

  // ---- stage B: output row r <-> t = t0 + r reads intermediate rows r + (tap offset + loB)
  DMX_STAMP(3);
  init_acc(P.b);
  run_stage(true, stepsB, stepsA);
  __syncthreads();
  DMX_STAMP(4);
  {
    constexpr int EPI_WAVE_BYTES = 64 * (FN * 32 + 16) + 64 * 12;
    static_assert(EPI_WAVE_BYTES * NW <= K::SLAB_BYTES, "epilogue staging does not fit the slab");
    const int mbase = b * T + t0;
    const int tend = t0 + P.BMo < T ? t0 + P.BMo : T;
    gemm_epilogue_lds_impl<4, FN, 1>(P.b, acc, mbase + rg * 64, cg * 64, lane, 0, T, slab + wave * EPI_WAVE_BYTES, b * T + tend, rpre,
                                        !single && P.r_from_slab != 0);
  }
  if (tape_bits_valid) {
#pragma unroll
    for (int it = 0; it < K::BITS_IT; ++it) {
      const int c = tid + it * NT, row = c / CPR, piece = c % CPR, t = t0 - P.loB + row;
      if (row >= P.loB && row < P.loB + P.BMo && t < T)
        P.a.B2[((long long)b * T + t) * P.a.ldb2 + piece] = (unsigned char)((tape_bits[it >> 2] >> (8 * (it & 3))) & 0xffu);
    }
  }
  DMX_STAMP(5);
#ifdef DMX_PAIR_STAMPS
  if (threadIdx.x == 0 && stamp_id < 4096)
    for (int i = 0; i < 8; ++i) g_pair_stamps[stamp_id * 8 + i] = stamp_v[i];
#endif
}

template <int C>
__global__ __launch_bounds__(PairCfg<C>::NT, PairCfg<C>::MIN_WAVES) void conv_pair_kernel(const PairParams P) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  pair_body<C>(P, smem, blockIdx.x, gridDim.x);
}

// Grouped launch: up to DMX_PAIR_GROUP independent pairs of the same width (the k = 3 / 7 / 11 branches of one HiFi-GAN resblock
// step) as ONE grid, problem j = blockIdx.y, longest problem first.  Alone, each branch is ~5 rounds of workgroups over the 256
// CUs and two of the three end with a nearly empty sixth round (1288 and 1304 workgroups for 1280 slots at C = 128: 16 % of their
// time); as one grid the short k = 3 workgroups fill the tails of the long ones.  gridDim.x is a multiple of 8 (the surplus
// workgroups of the shorter problems exit at once), so blockIdx.x & 7 is still the XCD the remap assumes.
constexpr int DMX_PAIR_GROUP = 3;
struct PairGroup { PairParams p[DMX_PAIR_GROUP]; int n[DMX_PAIR_GROUP]; };
template <int C>
__global__ __launch_bounds__(PairCfg<C>::NT, PairCfg<C>::MIN_WAVES) void conv_pair_group_kernel(const PairGroup G) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int j = blockIdx.y, n = G.n[j];
  if ((int)blockIdx.x >= n) return;
  pair_body<C>(G.p[j], smem, blockIdx.x, n);
}

struct Halo { int lo, hi; bool ok; int d; };
Halo halo_of(const GemmDesc& d) {
  Halo h{0, 0, true, d.ntaps > 1 ? d.tdx[1] - d.tdx[0] : 0};
  for (int j = 0; j < d.ntaps; ++j) {
    if (d.tdx[j] != d.tdx[0] + j * h.d) h.ok = false;      // taps must be affine in j
    if (d.tdy[j] != 0) h.ok = false;
    if (-d.tdx[j] > h.lo) h.lo = -d.tdx[j];
    if (d.tdx[j] > h.hi) h.hi = d.tdx[j];
  }
  if (h.lo + h.hi > PAIR_HALO) h.ok = false;
  return h;
}

bool stage_ok(const GemmDesc& d, int C, int T, int allowed_flags) {
  if (d.N != C || d.Ci != C || d.lda != C || d.K != d.ntaps * C || d.ldw < d.K) return false;
  if (d.ntaps < 1 || d.ntaps > 16 || d.Z != 1 || d.Hq != 1 || d.Hi != 1 || d.Ho != 1) return false;
  if (d.Wq != T || d.Wi != T || d.Wo != T || d.sx != 1 || d.osx != 1 || d.oox != 0 || d.ooy != 0) return false;
  if (d.flags & ~allowed_flags) return false;
  if (d.M % T != 0) return false;
  return true;
}

bool g_pair_enabled = getenv("DMX_NO_PAIR") == nullptr;

}  // namespace

bool dmx_conv_pair_eligible(const GemmDesc* a, const GemmDesc& b) {
  if (!g_pair_enabled) return false;
  const int C = b.N, T = b.Wq;
  static const bool c128 = getenv("DMX_NO_PAIR128") == nullptr;
  if (C != 32 && C != 64 && !(C == 128 && c128)) return false;
  if (T < 1) return false;
  const int bflags = EPI_BIAS | EPI_RESID | EPI_RESID_INV | EPI_ACCUM | EPI_MASK | EPI_LRELU2 | EPI_NO_C | EPI_MASKBITS | EPI_BITS2;
  if (!stage_ok(b, C, T, bflags) || !halo_of(b).ok) return false;
  if ((b.flags & EPI_LRELU2) && !(b.act_slope >= 0.f && b.act_slope <= 1.f)) return false;
  if ((b.flags & EPI_RESID_INV) && !(b.resid_inv_slope >= 1.f)) return false;
  if (b.ldc != C || ((b.flags & EPI_RESID) && b.ldr != C) || ((b.flags & EPI_MASK) && b.ldx != C) || ((b.flags & EPI_LRELU2) && b.ldc2 != C))
    return false;
  if ((b.flags & EPI_MASKBITS) && (!b.XB || b.ldxb * 8 < C)) return false;
  if ((b.flags & EPI_BITS2) && (!b.B2 || b.ldb2 * 8 < C)) return false;
  if (a) {
    if (!stage_ok(*a, C, T, EPI_BIAS | EPI_LRELU2 | EPI_NO_C | EPI_MASK | EPI_MASKBITS | EPI_BITS2) || !halo_of(*a).ok) return false;
    if (a->M != b.M || a->alpha != 1.f) return false;
    if ((a->flags & EPI_LRELU2) && !(a->flags & EPI_BITS2) && (a->ldc2 != C || !a->C2)) return false;    // a tape must leave the kernel
    if ((a->flags & EPI_LRELU2) && a->C2 && a->ldc2 != C) return false;
    if ((a->flags & EPI_BITS2) && (!(a->flags & EPI_LRELU2) || !a->B2 || a->ldb2 * 8 < C)) return false;
    if ((a->flags & EPI_MASK) && (a->ldx != C || !a->X)) return false;
    if ((a->flags & EPI_MASKBITS) && (!a->XB || a->ldxb * 8 < C || (a->flags & EPI_MASK))) return false;
    const Halo hb = halo_of(b);
    if (PAIR_ROWS - hb.lo - hb.hi < 128) return false;
  }
  return true;
}

namespace {
// fills the kernel parameters of one pair; returns its workgroup count, FLOPs and algorithmic bytes
long long pair_params(const GemmDesc* a, const GemmDesc& b, PairParams& P, double& fl, double& by) {
  memset(&P, 0, sizeof(P));
  P.b = b;
  P.single = a ? 0 : 1;
  if (a) P.a = *a;
  const Halo hb = halo_of(b);
  P.loB = hb.lo; P.hiB = hb.hi;
  P.off0B = b.tdx[0] + hb.lo; P.dB = hb.d;
  if (a) { const Halo ha = halo_of(*a); P.loA = ha.lo; P.hiA = ha.hi; P.off0A = a->tdx[0] + ha.lo; P.dA = ha.d; }
  P.T = b.Wq;
  P.BMo = a ? PAIR_ROWS - hb.lo - hb.hi : PAIR_ROWS;
  P.nb = cdiv(P.T, P.BMo);
  // channel biases that nothing precedes in the epilogue order (no leaky-relu' mask: the forward pairs) become the INITIAL value of the
  // stage's accumulators (EPI_BIASINIT): stage A loses an L2 round trip between its K loop and its tail, stage B the epilogue's bias term
  static const bool bias_init = getenv("DMX_NO_BIAS_INIT") == nullptr;
  if (bias_init && b.N < 128) {       // (the C = 128 instances spill 5 registers with it)
    if ((P.b.flags & EPI_BIAS) && P.b.bias && !(P.b.flags & (EPI_MASK | EPI_MASKBITS))) P.b.flags = (P.b.flags & ~EPI_BIAS) | EPI_BIASINIT;
    if (a && (P.a.flags & EPI_BIAS) && P.a.bias && !(P.a.flags & (EPI_MASK | EPI_MASKBITS))) P.a.flags = (P.a.flags & ~EPI_BIAS) | EPI_BIASINIT;
  }
  P.r_from_slab = (a && (b.flags & EPI_RESID) && b.R == a->A && b.ldr == b.N) ? 1 : 0;
  P.a_tape_bits_only = (a && (a->flags & EPI_BITS2) && !a->C2) ? 1 : 0;
  const int nclips = b.M / P.T;
  const int C = b.N;
  fl = 2.0 * b.M * (double)b.N * b.K;
  by = 2.0 * b.M * (double)C * 2.0;                     // input + output
  const double bits_by = b.M * (double)C / 8.0;
  if (a) {
    fl += 2.0 * a->M * (double)a->N * a->K;
    if ((a->flags & EPI_LRELU2) && !P.a_tape_bits_only) by += 2.0 * b.M * (double)C;
    if (a->flags & EPI_BITS2) by += bits_by;
    if (a->flags & EPI_MASK) by += 2.0 * b.M * (double)C;
    if (a->flags & EPI_MASKBITS) by += bits_by;
  }
  if ((b.flags & EPI_RESID) && !P.r_from_slab) by += 2.0 * b.M * (double)C;
  if (b.flags & EPI_MASK) by += 2.0 * b.M * (double)C;
  if (b.flags & EPI_MASKBITS) by += bits_by;
  if (b.flags & EPI_BITS2) by += bits_by;
  if (b.flags & EPI_ACCUM) by += 2.0 * b.M * (double)C;
  if ((b.flags & EPI_LRELU2) && !(b.flags & EPI_NO_C)) by += 2.0 * b.M * (double)C;
  return (long long)nclips * P.nb;
}
}  // namespace

// a == nullptr: plain slab convolution of `b`.  Otherwise b.A must be the tensor stage `a` produces
// (a.C2 when a carries EPI_LRELU2, else a.C); it is taken from LDS and, in the latter case, never written.
// (DMX_PAIR_EXTRA_LDS: occupancy experiments only -- unused LDS bytes added to the request so that fewer workgroups share a CU)
static int pair_extra_lds() { static const int v = [] { const char* e = getenv("DMX_PAIR_EXTRA_LDS"); return e ? atoi(e) : 0; }(); return v; }
int dmx_conv_pair_launch(const GemmDesc* a, const GemmDesc& b, hipStream_t st) {
  if (!dmx_conv_pair_eligible(a, b)) return DMX_ERR_SHAPE;
  PairParams P;
  double fl, by;
  const long long grid = pair_params(a, b, P, fl, by);
  if (grid > 0x7fffffffLL) return DMX_ERR_SHAPE;
  const int C = b.N;
  const int rec = dmx_prof_open(st);
  auto launch = [&](auto tag) {
    constexpr int CC = decltype(tag)::value;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv_pair_kernel<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, PairCfg<CC>::LDS_BYTES + (CC < 128 ? pair_extra_lds() : 0)); attr = true; }
    hipLaunchKernelGGL(conv_pair_kernel<CC>, dim3((unsigned)grid), dim3(PairCfg<CC>::NT), PairCfg<CC>::LDS_BYTES + (CC < 128 ? pair_extra_lds() : 0), st, P);
  };
  if (C == 32) launch(std::integral_constant<int, 32>{});
  else if (C == 64) launch(std::integral_constant<int, 64>{});
  else launch(std::integral_constant<int, 128>{});
  dmx_prof_close(rec, st, fl, by, b.M, b.N, (a ? a->K : 0) + b.K, b.ntaps, b.flags, a ? 21 : 20);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}

// n (<= 3) independent fused pairs of the same width in one grid (see conv_pair_group_kernel).  The pairs must not depend on each
// other's outputs and must not accumulate into the same tensor.  Falls back to n launches when the shapes differ in width.
int dmx_conv_pair_group_launch(int n, const GemmDesc* const* a, const GemmDesc* const* b, hipStream_t st) {
  if (n < 1 || n > DMX_PAIR_GROUP) return DMX_ERR_SHAPE;
  static const bool off = getenv("DMX_NO_PAIR_GROUP") != nullptr;
  bool same = !off && n > 1;
  for (int j = 0; j < n; ++j) {
    if (!a[j] || !b[j] || !dmx_conv_pair_eligible(a[j], *b[j])) return DMX_ERR_SHAPE;
    if (b[j]->N != b[0]->N) same = false;
    for (int i = 0; i < j; ++i)                               // no two problems may write the same tensor
      if (b[j]->C == b[i]->C || (b[j]->C2 && b[j]->C2 == b[i]->C2)) same = false;
  }
  if (!same) {
    for (int j = 0; j < n; ++j) { const int rc = dmx_conv_pair_launch(a[j], *b[j], st); if (rc != DMX_OK) return rc; }
    return DMX_OK;
  }
  // longest problem first: order by K steps per workgroup (stage A + stage B)
  int order[DMX_PAIR_GROUP] = {0, 1, 2};
  auto work = [&](int j) { return a[j]->K + b[j]->K; };
  for (int i = 0; i < n; ++i)
    for (int k = i + 1; k < n; ++k)
      if (work(order[k]) > work(order[i])) { const int t = order[i]; order[i] = order[k]; order[k] = t; }
  PairGroup G;
  memset(&G, 0, sizeof(G));
  double fl = 0.0, by = 0.0;
  long long gmax = 0;
  int Ksum = 0;
  for (int i = 0; i < n; ++i) {
    const int j = order[i];
    double f1, b1;
    const long long g = pair_params(a[j], *b[j], G.p[i], f1, b1);
    if (g > 0x7ffffff0LL) return DMX_ERR_SHAPE;
    G.n[i] = (int)g;
    if (g > gmax) gmax = g;
    fl += f1; by += b1; Ksum += a[j]->K + b[j]->K;
  }
  const unsigned gx = (unsigned)((gmax + 7) & ~7ll);
  const int C = b[0]->N;
  const int rec = dmx_prof_open(st);
  auto launch = [&](auto tag) {
    constexpr int CC = decltype(tag)::value;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv_pair_group_kernel<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, PairCfg<CC>::LDS_BYTES); attr = true; }
    hipLaunchKernelGGL(conv_pair_group_kernel<CC>, dim3(gx, (unsigned)n), dim3(PairCfg<CC>::NT), PairCfg<CC>::LDS_BYTES, st, G);
  };
  if (C == 32) launch(std::integral_constant<int, 32>{});
  else if (C == 64) launch(std::integral_constant<int, 64>{});
  else launch(std::integral_constant<int, 128>{});
  dmx_prof_close(rec, st, fl, by, b[0]->M, C, Ksum, b[0]->ntaps, b[0]->flags, 22);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}

#ifdef DMX_PAIR_STAMPS
extern "C" int dmx_pair_stamps_read(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pair_stamps), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
#endif
