// Pointwise / reduction kernels around the contractions (all HBM-bound): GroupNorm(+SiLU) fwd+bwd,
// LayerNorm, row softmax fwd+bwd, GEGLU, nearest upsample fwd+bwd, transposes, channel concat and
// the layout conversions at the stage boundaries.  Activations are channels-last bf16; every
// thread moves 16 bytes (8 channels) per access and reductions use wave64 shuffles + LDS.
#include "dmx_common.h"
#include <cstdlib>
#include "kernels.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  f[0] = alo(u.x); f[1] = ahi(u.x); f[2] = alo(u.y); f[3] = ahi(u.y);
  f[4] = alo(u.z); f[5] = ahi(u.z); f[6] = alo(u.w); f[7] = ahi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack2a(f[0], f[1]), pack2a(f[2], f[3]), pack2a(f[4], f[5]), pack2a(f[6], f[7]));
}
__device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
__device__ __forceinline__ float dsilu_f(float z) {
  const float s = 1.f / (1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// ------------------------------------------------------------------------------ GroupNorm
// x: (B, P, C).  Block = cpr*rpb threads (cpr = C/8 chunk columns, rpb pixel rows in flight);
// grid = (nchunk, B).  MODE 0: stats of x.  MODE 1: backward sums of dxhat and dxhat*xhat.
template <int MODE>
__global__ void gn_partial_kernel(const act_t* __restrict__ x, const act_t* __restrict__ dy,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  const float* __restrict__ stats, float* __restrict__ partial,
                                  int P, int C, int G, int rpb, int ppb, int silu) {
  extern __shared__ float sh[];  // [rpb][C][2]
  const int cpr = C >> 3;
  const int col = threadIdx.x % cpr, row = threadIdx.x / cpr;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * ppb, p1 = min(P, p0 + ppb);
  const int c0 = col << 3;
  float s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
  // (backward sums: s2 accumulates dxhat * (x - mean); the group's rstd is a common factor and is applied once at the end -- eight
  //  registers and one multiply per element less inside the loop, which is what lets it keep two rows in flight)
  float sc[8], sf[8], mu[8];
  if (MODE == 1) {
    const int cpg = C / G;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[i] = scale[(long long)b * C + c0 + i];
      sf[i] = shift[(long long)b * C + c0 + i];
      mu[i] = stats[((long long)b * G + (c0 + i) / cpg) * 2];
    }
  }
  const act_t* xb = x + (long long)b * P * C;
  const act_t* db = MODE == 1 ? dy + (long long)b * P * C : nullptr;
  auto body = [&](const uint4& xv, const uint4& dv) {
    float f[8];
    unpack8(xv, f);
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1[i] += f[i]; s2[i] += f[i] * f[i]; }
    } else {
      float d[8];
      unpack8(dv, d);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float z = f[i] * sc[i] + sf[i];
        const float dz = silu ? d[i] * dsilu_f(z) : d[i];
        const float dxh = dz * sc[i];              // = dz*gamma*rstd ; divide rstd out in finalize
        s1[i] += dxh; s2[i] += dxh * (f[i] - mu[i]);
      }
    }
  };
  int p = p0 + row;
  for (; MODE == 0 && p + 3 * rpb < p1; p += 4 * rpb) {   // stats pass: four independent 16-B loads in flight per thread
    // (the backward sums unrolled FOUR rows deep measured 2x slower: registers / occupancy; they take the two-row loop below)
    uint4 xv[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xv[u] = *reinterpret_cast<const uint4*>(xb + (long long)(p + u * rpb) * C + c0);
      if (MODE == 1) dv[u] = *reinterpret_cast<const uint4*>(db + (long long)(p + u * rpb) * C + c0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) body(xv[u], dv[u]);
  }
  for (; MODE == 1 && p + rpb < p1; p += 2 * rpb) {       // backward sums: two rows x two tensors = four 16-B loads in flight per thread
    uint4 xv[2], dv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      xv[u] = *reinterpret_cast<const uint4*>(xb + (long long)(p + u * rpb) * C + c0);
      dv[u] = *reinterpret_cast<const uint4*>(db + (long long)(p + u * rpb) * C + c0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) body(xv[u], dv[u]);
  }
  for (; p < p1; p += rpb) {
    uint4 dv = make_uint4(0, 0, 0, 0);
    if (MODE == 1) dv = *reinterpret_cast<const uint4*>(db + (long long)p * C + c0);
    body(*reinterpret_cast<const uint4*>(xb + (long long)p * C + c0), dv);
  }
  if (MODE == 1) {
    const int cpg = C / G;
#pragma unroll
    for (int i = 0; i < 8; ++i) s2[i] *= stats[((long long)b * G + (c0 + i) / cpg) * 2 + 1];
  }
  float* my = sh + ((long long)row * C + c0) * 2;
#pragma unroll
  for (int i = 0; i < 8; ++i) { my[2 * i] = s1[i]; my[2 * i + 1] = s2[i]; }
  __syncthreads();
  const int cpg = C / G;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int r = 0; r < rpb; ++r)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += sh[((long long)r * C + c) * 2]; q += sh[((long long)r * C + c) * 2 + 1]; }
    float* out = partial + (((long long)b * gridDim.x + blockIdx.x) * G + g) * 2;
    if (MODE == 0) {
      const float n = (float)(p1 - p0) * cpg;
      const float m = n > 0 ? a / n : 0.f;
      out[0] = m;                       // local mean
      out[1] = fmaxf(q - a * m, 0.f);   // local M2
    } else {
      out[0] = a; out[1] = q;
    }
  }
}

// Block-wide reduction over the k index of values laid out (k, g): thread t holds g = t % G.  Lanes t and t + 32 of a wave
// share g when G == 32; waves are combined through `sh` (blockDim/64 x G floats).  Returns the total for group t % G to every thread.
__device__ __forceinline__ float gn_group_total(float v, float* sh, int G) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
  if (G <= 32) {
    for (int o = 32; o >= G; o >>= 1) v += __shfl_xor(v, o, 64);
  }
  __syncthreads();
  if (lane < G || G > 32) sh[w * 64 + lane] = v;
  __syncthreads();
  float r = 0.f;
  const int g = tid % G;
  if (G <= 32) { for (int i = 0; i < nw; ++i) r += sh[i * 64 + g]; }
  else { for (int i = 0; i < nw; ++i) r += sh[i * 64 + (lane % G)]; }
  return r;
}

// combine partials -> stats[b,g] = (mean, rstd); scale[b,c] = rstd*gamma, shift[b,c] = beta - mean*rstd*gamma.
// One 1024-thread block per image; thread t owns group t % G and chunks t / G, t / G + 1024 / G, ...: every pass over the
// partials is one fully coalesced sweep with all loads in flight at once (the old 8-lanes-per-group walk exposed ~64
// dependent cache-line latencies per launch: 16 us for a kernel that moves 128 KB).
constexpr int GN_FIN_MAXIT = 32;
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ stats,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   int P, int C, int G, int nchunk, int ppb, float eps) {
  __shared__ float sh[16 * 64];
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.x, cpg = C / G;
  const int g = threadIdx.x % G, kk = threadIdx.x / G, kstep = blockDim.x / G;
  float2 pv[GN_FIN_MAXIT];
  float nbv[GN_FIN_MAXIT];
#pragma unroll
  for (int it = 0; it < GN_FIN_MAXIT; ++it) {
    const int k = kk + it * kstep;
    pv[it] = make_float2(0.f, 0.f);
    nbv[it] = 0.f;
    if (k < nchunk) {
      pv[it] = *reinterpret_cast<const float2*>(partial + (((long long)b * nchunk + k) * G + g) * 2);
      nbv[it] = (float)(min(P, (k + 1) * ppb) - k * ppb) * cpg;
    }
  }
  float sw = 0.f, sn = 0.f;
#pragma unroll
  for (int it = 0; it < GN_FIN_MAXIT; ++it) { sw += nbv[it] * pv[it].x; sn += nbv[it]; }
  sw = gn_group_total(sw, sh, G);
  sn = gn_group_total(sn, sh, G);
  const float mean = sw / sn;
  float m2 = 0.f;
#pragma unroll
  for (int it = 0; it < GN_FIN_MAXIT; ++it) { const float d = pv[it].x - mean; m2 += pv[it].y + nbv[it] * d * d; }
  m2 = gn_group_total(m2, sh, G);
  if (threadIdx.x < G) {
    const float rstd = rsqrtf(m2 / sn + eps);
    s_mean[g] = mean; s_rstd[g] = rstd;
    stats[((long long)b * G + g) * 2] = mean;
    stats[((long long)b * G + g) * 2 + 1] = rstd;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int gg = c / cpg;
    const float a = s_rstd[gg] * gamma[c];
    scale[(long long)b * C + c] = a;
    shift[(long long)b * C + c] = beta[c] - s_mean[gg] * a;
  }
}

// y = act(x*scale + shift)
__global__ void gn_apply_kernel(const act_t* __restrict__ x, const float* __restrict__ scale,
                                const float* __restrict__ shift, act_t* __restrict__ y,
                                int P, int C, int rpb, int ppb, int silu) {
  const int cpr = C >> 3;
  const int col = threadIdx.x % cpr, row = threadIdx.x / cpr;
  const int b = blockIdx.y, c0 = col << 3;
  const int p0 = blockIdx.x * ppb, p1 = min(P, p0 + ppb);
  float sc[8], sf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sc[i] = scale[(long long)b * C + c0 + i]; sf[i] = shift[(long long)b * C + c0 + i]; }
  const long long base = (long long)b * P * C + c0;
  auto body = [&](const uint4& xv, long long off) {
    float f[8];
    unpack8(xv, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float z = f[i] * sc[i] + sf[i];
      f[i] = silu ? silu_f(z) : z;
    }
    *reinterpret_cast<uint4*>(y + off) = pack8(f);
  };
  int p = p0 + row;
  for (; p + 3 * rpb < p1; p += 4 * rpb) {
    uint4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const uint4*>(x + base + (long long)(p + u * rpb) * C);
#pragma unroll
    for (int u = 0; u < 4; ++u) body(xv[u], base + (long long)(p + u * rpb) * C);
  }
  for (; p < p1; p += rpb) body(*reinterpret_cast<const uint4*>(x + base + (long long)p * C), base + (long long)p * C);
}

// Two-launch GroupNorm for the mid-size U-Net levels (1000 .. 4000 pixels): gn_partial_kernel<0> as above, then this kernel, whose
// prologue combines the (at most GN_FUSED_MAXCHUNK) chunk partials of its image -- the same Chan combine gn_finalize_kernel does --
// and keeps scale / shift of all channels in LDS.  Every workgroup repeats the tiny combine (<= 8 KB of partials from L2), which
// is cheaper than a third launch; workgroup 0 of an image also writes stats / scale / shift for callers that keep a tape.
constexpr int GN_FUSED_MAXCHUNK = 32;
__global__ __launch_bounds__(256) void gn_finalize_apply_kernel(const act_t* __restrict__ x, const float* __restrict__ partial,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ stats, float* __restrict__ scale, float* __restrict__ shift,
                                         act_t* __restrict__ y, int P, int C, int G, int nchunk, int rpb, int ppb, float eps, int silu) {
  __shared__ float s_red[GN_FUSED_MAXCHUNK][64];
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.y, cpg = C / G, tid = threadIdx.x;
  const int cpr = C >> 3;
  const int col = tid % cpr, row = tid / cpr, c0 = col << 3;
  const bool active = row < rpb;
  const int p0 = blockIdx.x * ppb, p1 = min(P, p0 + ppb);
  const long long base = (long long)b * P * C + c0;
  // everything that does not depend on the statistics is requested first: this thread's affine parameters, its first rows, and
  // the chunk partials (at most 8 (chunk, group) slots per thread), so that one memory round trip
  // covers the whole prologue
  float ga[8], be[8];
  uint4 xv0[4];
  int p = p0 + row;
  const bool first4 = active && p + 3 * rpb < p1;
  if (active) {
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + c0), g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + c0), b1 = *reinterpret_cast<const float4*>(beta + c0 + 4);
    ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w; ga[4] = g1.x; ga[5] = g1.y; ga[6] = g1.z; ga[7] = g1.w;
    be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w;
    if (first4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) xv0[u] = *reinterpret_cast<const uint4*>(x + base + (long long)(p + u * rpb) * C);
    }
  }
  constexpr int PV_MAX = 8;                       // the launcher keeps nchunk * G <= 8 * blockDim
  float2 pv[PV_MAX];
  const int nslots = nchunk * G, nth = blockDim.x;
#pragma unroll
  for (int q = 0; q < PV_MAX; ++q) {
    const int i = tid + q * nth;
    pv[q] = make_float2(0.f, 0.f);
    if (i < nslots) pv[q] = *reinterpret_cast<const float2*>(partial + ((long long)b * nslots + i) * 2);   // slot i = (chunk i / G, group i % G)
  }
#pragma unroll
  for (int q = 0; q < PV_MAX; ++q) {
    const int i = tid + q * nth;
    if (i < nslots) {
      const int k = i / G, g = i - k * G;
      const float nb = (float)(min(P, (k + 1) * ppb) - k * ppb) * cpg;
      s_red[k][g] = nb * pv[q].x;
    }
  }
  __syncthreads();
  if (tid < G) {
    float sw = 0.f;
    for (int k = 0; k < nchunk; ++k) sw += s_red[k][tid];
    s_mean[tid] = sw / ((float)P * cpg);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PV_MAX; ++q) {
    const int i = tid + q * nth;
    if (i < nslots) {
      const int k = i / G, g = i - k * G;
      const float nb = (float)(min(P, (k + 1) * ppb) - k * ppb) * cpg;
      const float d = pv[q].x - s_mean[g];
      s_red[k][g] = pv[q].y + nb * d * d;
    }
  }
  __syncthreads();
  if (tid < G) {
    float m2 = 0.f;
    for (int k = 0; k < nchunk; ++k) m2 += s_red[k][tid];
    const float rstd = rsqrtf(m2 / ((float)P * cpg) + eps);
    s_rstd[tid] = rstd;
    if (blockIdx.x == 0) { stats[((long long)b * G + tid) * 2] = s_mean[tid]; stats[((long long)b * G + tid) * 2 + 1] = rstd; }
  }
  __syncthreads();
  if (!active) return;
  float sc[8], sf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int gg = (c0 + i) / cpg;
    sc[i] = s_rstd[gg] * ga[i];
    sf[i] = be[i] - s_mean[gg] * sc[i];
  }
  if (blockIdx.x == 0 && row == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { scale[(long long)b * C + c0 + i] = sc[i]; shift[(long long)b * C + c0 + i] = sf[i]; }
  }
  auto body = [&](const uint4& xv, long long off) {
    float f[8];
    unpack8(xv, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float z = f[i] * sc[i] + sf[i];
      f[i] = silu ? silu_f(z) : z;
    }
    *reinterpret_cast<uint4*>(y + off) = pack8(f);
  };
  if (first4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) body(xv0[u], base + (long long)(p + u * rpb) * C);
    p += 4 * rpb;
  }
  for (; p + 3 * rpb < p1; p += 4 * rpb) {
    uint4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const uint4*>(x + base + (long long)(p + u * rpb) * C);
#pragma unroll
    for (int u = 0; u < 4; ++u) body(xv[u], base + (long long)(p + u * rpb) * C);
  }
  for (; p < p1; p += rpb) body(*reinterpret_cast<const uint4*>(x + base + (long long)p * C), base + (long long)p * C);
}

// ---- GroupNorm from PRODUCER-written partial sums (EPI_GNSTATS, gemm_epilogue.h): no statistics pass over the tensor.
// A region = the partial sums one producer launch left for (part of) the normalised tensor: per image, per wave tile of `tm` GEMM rows
// (slot j = tile index inside the image; a tile that straddles two images is the last slot of one and slot 0 of the next), per 4-channel
// quad: (sum v, sum v^2) of the stored 16-bit values.  Up to 8 regions make up a tensor: the four output-parity launches of an
// upsample-folded convolution (each covers a quarter of the pixels), the two sources of a skip concatenation (qoff = quad offset of the
// source's channel 0 in the normalised tensor; a group may straddle the seam).  Every (region, slot, group) item becomes (n, mean, M2) --
// M2 = sum v^2 - (sum v)^2 / n over at most tm x C/G values -- and the items are Chan-combined in two sweeps (mean first, then
// M2 + n (mean_i - mean)^2), like the chunk partials of gn_partial_kernel / gn_finalize_kernel.
// rows of image b inside slot j of region R, and the slot count of image b
// (the producers lay their wave tiles out per image -- an image whose row count is no multiple of the slot rows gets its own, image-aligned
// M tiling, gemm_glds_kernel -- so slot j of ANY image covers its rows [j tm, (j + 1) tm): a clip's statistics do not depend on its
// position in the batch or on the other clips)
__device__ __forceinline__ int gn_parts_rows(const GnRegion& R, int b, int j) {
  (void)b;
  const int left = R.P - j * R.tm;
  return left < R.tm ? (left > 0 ? left : 0) : R.tm;
}
__device__ __forceinline__ int gn_parts_nslots(const GnRegion& R, int b) {
  (void)b;
  return (R.P + R.tm - 1) / R.tm;
}
// quads [q0, q1) of region R that belong to group g (cpg channels per group)
__device__ __forceinline__ void gn_parts_quads(const GnRegion& R, int g, int cpg, int& q0, int& q1) {
  const int gq0 = (g * cpg) >> 2, gq1 = ((g + 1) * cpg) >> 2;
  q0 = max(gq0, R.qoff) - R.qoff;
  q1 = min(gq1, R.qoff + R.cq) - R.qoff;
}
// Block-wide: (mean, rstd) of every group of image b into s_mean / s_rstd.  Thread t < (nth / G) * G owns group t % G and the slots
// t / G, t / G + nth / G, ... of every region.  STAGED: the image's partial sums were copied to LDS first (`s_buf`, region r at float2
// offset soff[r], slot-major like the global layout) -- one coalesced round trip for the whole workgroup instead of a dependent load per
// item (the direct form cost the fused mid-size plan 8 us per launch).  `s_red` holds nth floats.
template <bool STAGED>
__device__ __forceinline__ void gn_parts_combine(const GnParts& sp, int b, int C, int G, float eps, float* s_red, float* s_mean, float* s_rstd,
                                                 const float2* s_buf, const int* soff) {
  const int tid = threadIdx.x, nth = blockDim.x, nslice = nth / G, cpg = C / G;
  const bool act = tid < nslice * G;
  const int g = tid % G, slice = tid / G;
  auto sweep = [&](auto&& f) {
    if (!act) return;
    for (int r = 0; r < sp.n; ++r) {
      const GnRegion& R = sp.r[r];
      int q0, q1;
      gn_parts_quads(R, g, cpg, q0, q1);
      if (q1 <= q0) continue;
      const int ns = gn_parts_nslots(R, b), slots = (R.P + R.tm - 1) / R.tm + 1;
      const float2* src = STAGED ? s_buf + soff[r] : reinterpret_cast<const float2*>(R.part) + (long long)b * slots * R.nq;
      const int nq = q1 - q0;
      if (!STAGED && nq <= 2) {
        // straight from global memory (the big tensors, whose partial sums do not fit LDS): 8 slots x <= 2 quads of loads in flight per
        // thread, then the same additions in the same order as the plain loop below (a thread of the 64000-pixel tensors walks 31 slots:
        // one dependent round trip each cost the one-workgroup-per-image launch 23 us)
        for (int j0 = slice; j0 < ns; j0 += 8 * nslice) {
          float2 v[8][2];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * nslice;
#pragma unroll
            for (int q = 0; q < 2; ++q) v[u][q] = (j < ns && q < nq) ? src[j * R.nq + q0 + q] : make_float2(0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * nslice;
            if (j >= ns) break;
            const int rows = gn_parts_rows(R, b, j);
            if (rows <= 0) continue;
            float sv = 0.f, qv = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) if (q < nq) { sv += v[u][q].x; qv += v[u][q].y; }
            f(sv, qv, (float)rows * 4.f * (float)nq);
          }
        }
        continue;
      }
      for (int j = slice; j < ns; j += nslice) {
        const int rows = gn_parts_rows(R, b, j);
        if (rows <= 0) continue;
        float sv = 0.f, qv = 0.f;
        for (int q = q0; q < q1; ++q) { const float2 v = src[j * R.nq + q]; sv += v.x; qv += v.y; }
        f(sv, qv, (float)rows * 4.f * (float)(q1 - q0));
      }
    }
  };
  auto reduce = [&](float v) -> float {        // sum over the slices of a group; result valid in threads < G
    __syncthreads();
    if (act) s_red[slice * G + g] = v;
    __syncthreads();
    float t = 0.f;
    if (tid < G) for (int k = 0; k < nslice; ++k) t += s_red[k * G + tid];
    return t;
  };
  float a = 0.f, n = 0.f;
  sweep([&](float sv, float qv, float nv) { a += sv; n += nv; });
  const float st = reduce(a), nt = reduce(n);
  if (tid < G) s_mean[tid] = st / fmaxf(nt, 1.f);
  __syncthreads();
  const float mean = s_mean[g];
  float m2 = 0.f;
  sweep([&](float sv, float qv, float nv) {
    const float mi = sv / nv, d = mi - mean;
    m2 += fmaxf(qv - sv * mi, 0.f) + nv * d * d;
  });
  const float m2t = reduce(m2);
  if (tid < G) s_rstd[tid] = rsqrtf(m2t / fmaxf(nt, 1.f) + eps);
  __syncthreads();
}

// grid (nchunk, B).  y != nullptr: finalize + apply in one launch (every workgroup repeats the small combine; workgroup 0 of an image also
// writes stats / scale / shift for callers that keep a tape).  y == nullptr: statistics only (launched with one workgroup per image).
template <int NT>
__global__ __launch_bounds__(NT) void gn_parts_kernel(const act_t* __restrict__ x, act_t* __restrict__ y, const GnParts sp,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ stats, float* __restrict__ scale, float* __restrict__ shift,
                                                     int P, int C, int G, int rpb, int ppb, float eps, int silu, int stage_f2) {
  __shared__ float s_red[NT];
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ int s_off[8];
  extern __shared__ __attribute__((aligned(16))) float2 s_stage[];     // stage_f2 > 0: room for the image's partial sums (float2 count)
  const int b = blockIdx.y, cpg = C / G, tid = threadIdx.x;
  // what does not depend on the statistics is requested first (this thread's affine parameters and its first rows of x), so that one memory
  // round trip covers them together with the partial sums
  const int cpr = C >> 3;
  const int col = tid % cpr, row = tid / cpr, c0 = col << 3;
  const bool active = y && row < rpb;
  const int p0 = blockIdx.x * ppb, p1 = min(P, p0 + ppb);
  const long long base = (long long)b * P * C + c0;
  float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, b0 = g0, b1 = g0;
  uint4 xv0[4];
  int p = p0 + row;
  const bool first4 = active && p + 3 * rpb < p1;
  if (active) {
    g0 = *reinterpret_cast<const float4*>(gamma + c0); g1 = *reinterpret_cast<const float4*>(gamma + c0 + 4);
    b0 = *reinterpret_cast<const float4*>(beta + c0); b1 = *reinterpret_cast<const float4*>(beta + c0 + 4);
    if (first4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) xv0[u] = *reinterpret_cast<const uint4*>(x + base + (long long)(p + u * rpb) * C);
    }
  }
  if (stage_f2 > 0) {
    int off = 0;
    for (int r = 0; r < sp.n; ++r) {
      const GnRegion& R = sp.r[r];
      const int cnt = gn_parts_nslots(R, b) * R.nq, slots = (R.P + R.tm - 1) / R.tm + 1;
      const float2* src = reinterpret_cast<const float2*>(R.part) + (long long)b * slots * R.nq;
      // batches of 8 loads per thread in flight (a load-then-store loop waits for every load before its LDS write: one round trip each)
      for (int i0 = 0; i0 < cnt; i0 += NT * 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT + tid; v[u] = i < cnt ? src[i] : make_float2(0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT + tid; if (i < cnt) s_stage[off + i] = v[u]; }
      }
      if (tid == 0) s_off[r] = off;
      off += cnt;
    }
    __syncthreads();
    gn_parts_combine<true>(sp, b, C, G, eps, s_red, s_mean, s_rstd, s_stage, s_off);
  } else {
    gn_parts_combine<false>(sp, b, C, G, eps, s_red, s_mean, s_rstd, nullptr, nullptr);
  }
  if (blockIdx.x == 0) {
    if (tid < G) { stats[((long long)b * G + tid) * 2] = s_mean[tid]; stats[((long long)b * G + tid) * 2 + 1] = s_rstd[tid]; }
    for (int c = tid; c < C; c += NT) {
      const float a = s_rstd[c / cpg] * gamma[c];
      scale[(long long)b * C + c] = a;
      shift[(long long)b * C + c] = beta[c] - s_mean[c / cpg] * a;
    }
  }
  if (!active) return;
  float sc[8], sf[8];
  {
    const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gg = (c0 + i) / cpg;
      sc[i] = s_rstd[gg] * ga[i];
      sf[i] = be[i] - s_mean[gg] * sc[i];
    }
  }
  auto body = [&](const uint4& xv, long long off) {
    float f[8];
    unpack8(xv, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float z = f[i] * sc[i] + sf[i];
      f[i] = silu ? silu_f(z) : z;
    }
    *reinterpret_cast<uint4*>(y + off) = pack8(f);
  };
  if (first4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) body(xv0[u], base + (long long)(p + u * rpb) * C);
    p += 4 * rpb;
  }
  for (; p + 3 * rpb < p1; p += 4 * rpb) {
    uint4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const uint4*>(x + base + (long long)(p + u * rpb) * C);
#pragma unroll
    for (int u = 0; u < 4; ++u) body(xv[u], base + (long long)(p + u * rpb) * C);
  }
  for (; p < p1; p += rpb) body(*reinterpret_cast<const uint4*>(x + base + (long long)p * C), base + (long long)p * C);
}

// backward finalize: per (b,c) coefficients k0, k1 with dx = scale*dy*act'(z) + k0 + k1*x
__global__ __launch_bounds__(1024) void gn_bwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ stats,
                                       float* __restrict__ k0, float* __restrict__ k1,
                                       int P, int C, int G, int nchunk) {
  __shared__ float sh[16 * 64];
  __shared__ float s_c1[64], s_c2[64];
  const int b = blockIdx.x, cpg = C / G;
  const int g = threadIdx.x % G, kk = threadIdx.x / G, kstep = blockDim.x / G;
  float a = 0.f, q = 0.f;
#pragma unroll
  for (int it = 0; it < GN_FIN_MAXIT; ++it) {
    const int k = kk + it * kstep;
    if (k < nchunk) {
      const float2 pp = *reinterpret_cast<const float2*>(partial + (((long long)b * nchunk + k) * G + g) * 2);
      a += pp.x; q += pp.y;
    }
  }
  a = gn_group_total(a, sh, G);
  q = gn_group_total(q, sh, G);
  if (threadIdx.x < G) {
    const float n = (float)P * cpg;
    s_c1[g] = a / n;   // mean(dz*gamma*rstd)          (rstd already folded in via `scale`)
    s_c2[g] = q / n;   // mean(dz*gamma*rstd * xhat)
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int gg = c / cpg;
    const float mean = stats[((long long)b * G + gg) * 2], rstd = stats[((long long)b * G + gg) * 2 + 1];
    // dx = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat)), with rstd*dxhat == scale*dz
    // => dx = scale*dz - c1 - c2*rstd*(x-mean)
    k0[(long long)b * C + c] = -s_c1[gg] + s_c2[gg] * rstd * mean;
    k1[(long long)b * C + c] = -s_c2[gg] * rstd;
  }
}

// backward finalize from PRODUCER-written partial sums (EPI_GNBWD, gemm_epilogue.h): per slot and 4-channel quad (sum dxh, sum dxh (x - mean))
// with dxh = dy act'(z) gamma rstd; per group  c1 = sum dxh / n,  c2 = rstd * sum dxh (x - mean) / n,  then k0 / k1 as above.
// One workgroup per image; thread t owns group t % G and the slots t / G, t / G + blockDim / G, ... (plain sums: no conditioning issue,
// every term is a product of O(1) factors); batches of 8 loads in flight per thread.
__global__ __launch_bounds__(1024) void gn_bwd_parts_finalize_kernel(const GnParts sp, const float* __restrict__ stats,
                                                                      float* __restrict__ k0, float* __restrict__ k1, int P, int C, int G) {
  __shared__ float s_red[1024];
  __shared__ float s_c1[64], s_c2[64];
  const int b = blockIdx.x, cpg = C / G, tid = threadIdx.x, nth = blockDim.x, nslice = nth / G;
  const bool act = tid < nslice * G;
  const int g = tid % G, slice = tid / G;
  float a1 = 0.f, a2 = 0.f;
  if (act) {
    for (int r = 0; r < sp.n; ++r) {
      const GnRegion& R = sp.r[r];
      int q0, q1;
      gn_parts_quads(R, g, cpg, q0, q1);
      if (q1 <= q0) continue;
      const int ns = gn_parts_nslots(R, b), slots = (R.P + R.tm - 1) / R.tm + 1, nq = q1 - q0;
      const float2* src = reinterpret_cast<const float2*>(R.part) + (long long)b * slots * R.nq;
      const int items = ((ns - slice + nslice - 1) / nslice) * nq;             // (slot, quad) pairs of this thread
      for (int i0 = 0; i0 < items; i0 += 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u, j = slice + (i / nq) * nslice, q = q0 + i % nq;
          v[u] = (i < items && j < ns && gn_parts_rows(R, b, j) > 0) ? src[j * R.nq + q] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a1 += v[u].x; a2 += v[u].y; }
      }
    }
  }
  auto reduce = [&](float v) -> float {
    __syncthreads();
    if (act) s_red[slice * G + g] = v;
    __syncthreads();
    float t = 0.f;
    if (tid < G) for (int k = 0; k < nslice; ++k) t += s_red[k * G + tid];
    return t;
  };
  const float t1 = reduce(a1), t2 = reduce(a2);
  if (tid < G) {
    const float n = (float)P * cpg;
    const float mean = stats[((long long)b * G + tid) * 2], rstd = stats[((long long)b * G + tid) * 2 + 1];
    s_c1[tid] = t1 / n;
    s_c2[tid] = rstd * t2 / n;                 // (the producer summed dxh (x - mean))
    (void)mean;
  }
  __syncthreads();
  for (int c = tid; c < C; c += nth) {
    const int gg = c / cpg;
    const float mean = stats[((long long)b * G + gg) * 2], rstd = stats[((long long)b * G + gg) * 2 + 1];
    k0[(long long)b * C + c] = -s_c1[gg] + s_c2[gg] * rstd * mean;
    k1[(long long)b * C + c] = -s_c2[gg] * rstd;
  }
}

__global__ void gn_bwd_apply_kernel(const act_t* __restrict__ x, const act_t* __restrict__ dy,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ k0, const float* __restrict__ k1,
                                    const act_t* __restrict__ add, act_t* __restrict__ dx,
                                    int P, int C, int rpb, int ppb, int silu) {
  const int cpr = C >> 3;
  const int col = threadIdx.x % cpr, row = threadIdx.x / cpr;
  const int b = blockIdx.y, c0 = col << 3;
  const int p0 = blockIdx.x * ppb, p1 = min(P, p0 + ppb);
  float sc[8], sf[8], a0[8], a1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long long o = (long long)b * C + c0 + i;
    sc[i] = scale[o]; sf[i] = shift[o]; a0[i] = k0[o]; a1[i] = k1[o];
  }
  const long long base = (long long)b * P * C + c0;
  auto body = [&](const uint4& xv, const uint4& dv, const uint4& rv, long long off) {
    float f[8], d[8], r[8];
    unpack8(xv, f);
    unpack8(dv, d);
    if (add) unpack8(rv, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float z = f[i] * sc[i] + sf[i];
      const float dz = silu ? d[i] * dsilu_f(z) : d[i];
      float v = sc[i] * dz + a0[i] + a1[i] * f[i];
      if (add) v += r[i];
      f[i] = v;
    }
    *reinterpret_cast<uint4*>(dx + off) = pack8(f);
  };
  int p = p0 + row;
  for (; p + 1 * rpb < p1; p += 2 * rpb) {             // 2 rows x 3 tensors = six 16-B loads in flight per thread
    uint4 xv[2], dv[2], rv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long off = base + (long long)(p + u * rpb) * C;
      xv[u] = *reinterpret_cast<const uint4*>(x + off);
      dv[u] = *reinterpret_cast<const uint4*>(dy + off);
      rv[u] = add ? *reinterpret_cast<const uint4*>(add + off) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) body(xv[u], dv[u], rv[u], base + (long long)(p + u * rpb) * C);
  }
  for (; p < p1; p += rpb) {
    const long long off = base + (long long)p * C;
    body(*reinterpret_cast<const uint4*>(x + off), *reinterpret_cast<const uint4*>(dy + off),
         add ? *reinterpret_cast<const uint4*>(add + off) : make_uint4(0, 0, 0, 0), off);
  }
}

// ------------------------------------------------------------------------------ LayerNorm (rows, C)
__global__ void layernorm_kernel(const act_t* __restrict__ x, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, act_t* __restrict__ y, int rows, int C, float eps) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= rows) return;
  const act_t* xr = x + (long long)wave * C;
  float s = 0.f, q = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(xr + c), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += f[i]; q += f[i] * f[i]; }
  }
  s = wave_sum(s); q = wave_sum(q);
  const float mean = s / C, rstd = rsqrtf(fmaxf(q / C - mean * mean, 0.f) + eps);
  for (int c = lane * 8; c < C; c += 512) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(xr + c), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (f[i] - mean) * rstd * gamma[c + i] + beta[c + i];
    *reinterpret_cast<uint4*>(y + (long long)wave * C + c) = pack8(f);
  }
}

// ------------------------------------------------------------------------------ softmax
// S fp32 (rows, N) ld=lds -> P fp16 (rows, N) ld=ldp; optional additive column bias (mask).
// Single pass: TPR threads own one row and keep it in registers (<= 16 values per thread, N <= 16*TPR).
template <int TPR>
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ S, act_t* __restrict__ P, const float* __restrict__ colbias,
                                                      long long rows, int N, long long lds, long long ldp, int rows_per_bias) {
  __shared__ float sh[16];
  constexpr int RPB = 256 / TPR;
  const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TPR;
  const int t = threadIdx.x % TPR;
  const bool live = row < rows;
  const float* s = S + (live ? row : 0) * lds;
  const float* cb = colbias ? colbias + ((live ? row : 0) / rows_per_bias) * N : nullptr;
  float v[16];
  float mx = -3.0e38f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = (q * TPR + t) * 4;
    float4 x = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    if (live && c < N) {
      x = *reinterpret_cast<const float4*>(s + c);
      if (cb) { const float4 bb = *reinterpret_cast<const float4*>(cb + c); x.x += bb.x; x.y += bb.y; x.z += bb.z; x.w += bb.w; }
    }
    v[q * 4] = x.x; v[q * 4 + 1] = x.y; v[q * 4 + 2] = x.z; v[q * 4 + 3] = x.w;
    mx = fmaxf(mx, fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w)));
  }
  if (TPR == 64) mx = wave_max(mx); else mx = block_max(mx, sh);
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) { v[q] = __expf(v[q] - mx); sum += v[q]; }
  if (TPR == 64) sum = wave_sum(sum); else sum = block_sum(sum, sh);
  const float inv = 1.f / sum;
  if (!live) return;
  act_t* p = P + row * ldp;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = (q * TPR + t) * 4;
    if (c < N) *reinterpret_cast<uint2*>(p + c) = make_uint2(pack2a(v[q * 4] * inv, v[q * 4 + 1] * inv), pack2a(v[q * 4 + 2] * inv, v[q * 4 + 3] * inv));
    else if (c < ldp) *reinterpret_cast<uint2*>(p + c) = make_uint2(0u, 0u);      // zero the key padding (ldp = pad8(N))
  }
}

// fp16-score variant (in place allowed): S and P share the (rows, ldp) fp16 buffer written by the score GEMM.
template <int TPR>
__global__ __launch_bounds__(256) void softmax_act_kernel(const act_t* __restrict__ S, act_t* __restrict__ P, const float* __restrict__ colbias,
                                                          long long rows, int N, long long ldp, int rows_per_bias) {
  __shared__ float sh[16];
  constexpr int RPB = 256 / TPR;
  const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TPR;
  const int t = threadIdx.x % TPR;
  const bool live = row < rows;
  const act_t* s = S + (live ? row : 0) * ldp;
  const float* cb = colbias ? colbias + ((live ? row : 0) / rows_per_bias) * N : nullptr;
  float v[16];
  float mx = -3.0e38f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = (q * TPR + t) * 4;
    float x0 = -3.0e38f, x1 = -3.0e38f, x2 = -3.0e38f, x3 = -3.0e38f;
    if (live && c < N) {
      const uint2 u = *reinterpret_cast<const uint2*>(s + c);
      x0 = alo(u.x); x1 = ahi(u.x); x2 = alo(u.y); x3 = ahi(u.y);
      if (cb) { const float4 bb = *reinterpret_cast<const float4*>(cb + c); x0 += bb.x; x1 += bb.y; x2 += bb.z; x3 += bb.w; }
    }
    v[q * 4] = x0; v[q * 4 + 1] = x1; v[q * 4 + 2] = x2; v[q * 4 + 3] = x3;
    mx = fmaxf(mx, fmaxf(fmaxf(x0, x1), fmaxf(x2, x3)));
  }
  if (TPR == 64) mx = wave_max(mx); else mx = block_max(mx, sh);
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) { v[q] = __expf(v[q] - mx); sum += v[q]; }
  if (TPR == 64) sum = wave_sum(sum); else sum = block_sum(sum, sh);
  const float inv = 1.f / sum;
  if (!live) return;
  act_t* p = P + row * ldp;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = (q * TPR + t) * 4;
    if (c < N) *reinterpret_cast<uint2*>(p + c) = make_uint2(pack2a(v[q * 4] * inv, v[q * 4 + 1] * inv), pack2a(v[q * 4 + 2] * inv, v[q * 4 + 3] * inv));
    else if (c < ldp) *reinterpret_cast<uint2*>(p + c) = make_uint2(0u, 0u);
  }
}

// ------------------------------------------------------------------------------ GEGLU: (rows, 2I) -> (rows, I)
__global__ void geglu_kernel(const act_t* __restrict__ x, act_t* __restrict__ y, long long rows, int I) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = I >> 3;
  if (idx >= rows * cpr) return;
  const long long r = idx / cpr;
  const int c = (int)(idx - r * cpr) << 3;
  float a[8], g[8];
  unpack8(*reinterpret_cast<const uint4*>(x + r * 2 * I + c), a);
  unpack8(*reinterpret_cast<const uint4*>(x + r * 2 * I + I + c), g);
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] *= 0.5f * g[i] * (1.f + erff(g[i] * 0.70710678118654752f));
  *reinterpret_cast<uint4*>(y + r * I + c) = pack8(a);
}

__global__ void silu_kernel(const act_t* __restrict__ x, act_t* __restrict__ y, long long n8) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  float f[8];
  unpack8(reinterpret_cast<const uint4*>(x)[idx], f);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = silu_f(f[i]);
  reinterpret_cast<uint4*>(y)[idx] = pack8(f);
}

// ------------------------------------------------------------------------------ nearest upsample (B,Hi,Wi,C)->(B,Ho,Wo,C)
__global__ void upsample_nearest_kernel(const act_t* __restrict__ x, act_t* __restrict__ y,
                                        int B, int Hi, int Wi, int Ho, int Wo, int C) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = C >> 3;
  const long long total = (long long)B * Ho * Wo * cpr;
  if (idx >= total) return;
  const int c = (int)(idx % cpr) << 3;
  long long pix = idx / cpr;
  const int ox = (int)(pix % Wo); pix /= Wo;
  const int oy = (int)(pix % Ho); const int b = (int)(pix / Ho);
  const int iy = min((int)floorf(oy * ((float)Hi / Ho)), Hi - 1), ix = min((int)floorf(ox * ((float)Wi / Wo)), Wi - 1);
  *reinterpret_cast<uint4*>(y + (((long long)b * Ho + oy) * Wo + ox) * C + c) =
      *reinterpret_cast<const uint4*>(x + (((long long)b * Hi + iy) * Wi + ix) * C + c);
}
// exact x2 backward: dx[iy,ix] = sum of the 2x2 block
__global__ void upsample2x_bwd_kernel(const act_t* __restrict__ dy, act_t* __restrict__ dx, int B, int Hi, int Wi, int C) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = C >> 3;
  const long long total = (long long)B * Hi * Wi * cpr;
  if (idx >= total) return;
  const int c = (int)(idx % cpr) << 3;
  long long pix = idx / cpr;
  const int ix = (int)(pix % Wi); pix /= Wi;
  const int iy = (int)(pix % Hi); const int b = (int)(pix / Hi);
  const int Ho = Hi * 2, Wo = Wi * 2;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int dyi = 0; dyi < 2; ++dyi)
#pragma unroll
    for (int dxi = 0; dxi < 2; ++dxi) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(dy + (((long long)b * Ho + 2 * iy + dyi) * Wo + 2 * ix + dxi) * C + c), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += f[i];
    }
  *reinterpret_cast<uint4*>(dx + (((long long)b * Hi + iy) * Wi + ix) * C + c) = pack8(acc);
}

// ------------------------------------------------------------------------------ batched transpose
// in[z][r][c] (row stride ldi, batch strides) -> out[z][c][r] (row stride ldo).  32x32 LDS tiles.
__global__ void transpose_kernel(const act_t* __restrict__ in, act_t* __restrict__ out, int R, int Cc,
                                 long long ldi, long long ldo, int Zi, long long sIo, long long sIi,
                                 long long sOo, long long sOi) {
  __shared__ act_t tile[32][33];
  const int z = blockIdx.z, zo = z / Zi, zi = z - zo * Zi;
  const act_t* ib = in + zo * sIo + zi * sIi;
  act_t* ob = out + zo * sOo + zi * sOi;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < Cc) ? ib[(long long)r * ldi + c] : (act_t)0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < R && c < Cc) ob[(long long)c * ldo + r] = tile[tx][j];
  }
}

// Vector variant for 16-byte-granular operands (row strides, batch strides and base pointers multiples of 8 elements): 64 x 64
// tiles, 16-B global loads along the input rows and 16-B global stores along the output rows; the element transposition goes
// through LDS with a 66-element pitch (33 dwords: rows 8 apart fall on different banks).  Output columns past R inside the last
// 8-wide chunk are written as zeros (the callers pad the row pitch of the transposed tensor to a multiple of 8).
__global__ __launch_bounds__(256) void transpose64_kernel(const act_t* __restrict__ in, act_t* __restrict__ out, int R, int Cc,
                                                          long long ldi, long long ldo, int Zi, long long sIo, long long sIi,
                                                          long long sOo, long long sOi) {
  __shared__ act_t tile[64][66];
  const int z = blockIdx.z, zo = z / Zi, zi = z - zo * Zi;
  const act_t* ib = in + zo * sIo + zi * sIi;
  act_t* ob = out + zo * sOo + zi * sOi;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int id = threadIdx.x + k * 256, row = id >> 3, cc = id & 7;
    const int r = r0 + row, c = c0 + cc * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < R && c < Cc) {
      if (c + 8 <= Cc) {
        v = *reinterpret_cast<const uint4*>(ib + (long long)r * ldi + c);
      } else {
        act_t t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = c + e < Cc ? ib[(long long)r * ldi + c + e] : (act_t)0;
        v = make_uint4(t[0] | ((uint32_t)t[1] << 16), t[2] | ((uint32_t)t[3] << 16), t[4] | ((uint32_t)t[5] << 16), t[6] | ((uint32_t)t[7] << 16));
      }
    }
    uint32_t* d = reinterpret_cast<uint32_t*>(&tile[row][cc * 8]);      // 132-byte rows: 4-byte aligned only
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int id = threadIdx.x + k * 256, orow = id >> 3, rc = id & 7;
    const int c = c0 + orow, r = r0 + rc * 8;
    if (c < Cc && r < R) {
      act_t t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = tile[rc * 8 + e][orow];       // rows past R hold zeros (loaded as such)
      *reinterpret_cast<uint4*>(ob + (long long)c * ldo + r) =
          make_uint4(t[0] | ((uint32_t)t[1] << 16), t[2] | ((uint32_t)t[3] << 16), t[4] | ((uint32_t)t[5] << 16), t[6] | ((uint32_t)t[7] << 16));
    }
  }
}

// 128 x 128 tiles for the big (N x N) score-matrix transposes of the VAE mid attention backward: 256-byte runs on both the read
// and the write side (the 64 x 64 tile's 128-byte runs reached 1.8 TB/s on a 4000 x 4000 x 8 tensor).  Same edge handling.
__global__ __launch_bounds__(256) void transpose128_kernel(const act_t* __restrict__ in, act_t* __restrict__ out, int R, int Cc,
                                                           long long ldi, long long ldo, int Zi, long long sIo, long long sIi,
                                                           long long sOo, long long sOi) {
  __shared__ act_t tile[128][130];
  const int z = blockIdx.z, zo = z / Zi, zi = z - zo * Zi;
  const act_t* ib = in + zo * sIo + zi * sIi;
  act_t* ob = out + zo * sOo + zi * sOi;
  const int c0 = blockIdx.x * 128, r0 = blockIdx.y * 128;
  uint4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int id = threadIdx.x + k * 256, row = id >> 4, cc = id & 15;
    const int r = r0 + row, c = c0 + cc * 8;
    v[k] = make_uint4(0, 0, 0, 0);
    if (r < R && c < Cc) {
      if (c + 8 <= Cc) {
        v[k] = *reinterpret_cast<const uint4*>(ib + (long long)r * ldi + c);
      } else {
        act_t t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = c + e < Cc ? ib[(long long)r * ldi + c + e] : (act_t)0;
        v[k] = make_uint4(t[0] | ((uint32_t)t[1] << 16), t[2] | ((uint32_t)t[3] << 16), t[4] | ((uint32_t)t[5] << 16), t[6] | ((uint32_t)t[7] << 16));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int id = threadIdx.x + k * 256, row = id >> 4, cc = id & 15;
    uint32_t* d = reinterpret_cast<uint32_t*>(&tile[row][cc * 8]);      // 260-byte rows: 4-byte aligned only
    d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int id = threadIdx.x + k * 256, orow = id >> 4, rc = id & 15;
    const int c = c0 + orow, r = r0 + rc * 8;
    if (c < Cc && r < R) {
      act_t t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = tile[rc * 8 + e][orow];       // rows past R hold zeros (loaded as such)
      *reinterpret_cast<uint4*>(ob + (long long)c * ldo + r) =
          make_uint4(t[0] | ((uint32_t)t[1] << 16), t[2] | ((uint32_t)t[3] << 16), t[4] | ((uint32_t)t[5] << 16), t[6] | ((uint32_t)t[7] << 16));
    }
  }
}

// out[r] = sum_c a[r, c] * b[r, c] (fp32): the delta term of the attention backward, delta = rowsum(dO * O) = rowsum(dP * P)
__global__ __launch_bounds__(256) void rowdot_kernel(const act_t* __restrict__ a, const act_t* __restrict__ b, float* __restrict__ out,
                                                     long long rows, int C, long long lda, long long ldb) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    float fa[8], fb[8];
    unpack8(*reinterpret_cast<const uint4*>(a + r * lda + c), fa);
    unpack8(*reinterpret_cast<const uint4*>(b + r * ldb + c), fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += fa[e] * fb[e];
  }
  acc = wave_sum(acc);
  if (lane == 0) out[r] = acc;
}

// copy (rows, C) block into a wider channels-last tensor at channel offset (concat / slice)
__global__ void copy_channels_kernel(const act_t* __restrict__ src, act_t* __restrict__ dst, long long rows,
                                     int C, int lds, int ldd, int soff, int doff) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = C >> 3;
  if (idx >= rows * cpr) return;
  const long long r = idx / cpr;
  const int c = (int)(idx - r * cpr) << 3;
  *reinterpret_cast<uint4*>(dst + r * ldd + doff + c) = *reinterpret_cast<const uint4*>(src + r * lds + soff + c);
}

// channel concat of two channels-last tensors in one launch: dst[r] = [a[r] (Ca channels) | b[r] (Cb channels)]
__global__ void concat2_kernel(const act_t* __restrict__ a, const act_t* __restrict__ b, act_t* __restrict__ dst, long long rows, int Ca, int Cb) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cpr = (Ca + Cb) >> 3, ca8 = Ca >> 3;
  if (idx >= rows * cpr) return;
  const long long r = idx / cpr;
  const int c = (int)(idx - r * cpr);
  const uint4 v = c < ca8 ? *reinterpret_cast<const uint4*>(a + r * Ca + (c << 3)) : *reinterpret_cast<const uint4*>(b + r * Cb + ((c - ca8) << 3));
  *reinterpret_cast<uint4*>(dst + r * (Ca + Cb) + (c << 3)) = v;
}

// y = a*x (+ b*y0)  over bf16
__global__ void axpby_kernel(const act_t* __restrict__ x, const act_t* __restrict__ y0, act_t* __restrict__ y,
                             float a, float b, long long n8) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  float f[8], g[8];
  unpack8(reinterpret_cast<const uint4*>(x)[idx], f);
  if (y0) {
    unpack8(reinterpret_cast<const uint4*>(y0)[idx], g);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = a * f[i] + b * g[i];
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] *= a;
  }
  reinterpret_cast<uint4*>(y)[idx] = pack8(f);
}

// fp32 NCHW (B,C,H,W) * scale -> bf16 NHWC padded to Cp channels ;  and the reverse (+ scale)
__global__ void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ x, act_t* __restrict__ y, int B, int C, int HW,
                                             int Cp, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * HW * Cp) return;
  const int c = (int)(idx % Cp);
  const long long pix = idx / Cp;
  const int b = (int)(pix / HW), p = (int)(pix - (long long)b * HW);
  y[idx] = c < C ? f2a(x[((long long)b * C + c) * HW + p] * scale) : (act_t)0;
}
__global__ void nhwc_bf16_to_nchw_f32_kernel(const act_t* __restrict__ x, float* __restrict__ y, int B, int C, int HW,
                                             int Cp, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * C * HW) return;
  const int p = (int)(idx % HW);
  const long long bc = idx / HW;
  const int b = (int)(bc / C), c = (int)(bc - (long long)b * C);
  y[idx] = a2f(x[((long long)b * HW + p) * Cp + c]) * scale;
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, act_t* __restrict__ y, long long n, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) y[idx] = f2a(x[idx] * scale);
}
__global__ void bf16_to_f32_kernel(const act_t* __restrict__ x, float* __restrict__ y, long long n, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) y[idx] = a2f(x[idx]) * scale;
}
// (rows, ld) bf16 column `col` <-> (rows) fp32 vector : used for the 1-channel tensors (waveform, mel)
__global__ void extract_col_kernel(const act_t* __restrict__ x, float* __restrict__ y, long long rows, int ld, int col) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < rows) y[idx] = a2f(x[idx * ld + col]);
}
// sinusoidal timestep embedding [cos | sin] (flip_sin_to_cos) -> bf16 (B, dim)
__global__ void timestep_embed_kernel(const float* __restrict__ t, act_t* __restrict__ y, int B, int dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * dim) return;
  const int b = idx / dim, j = idx - b * dim, half = dim >> 1;
  const int k = j < half ? j : j - half;
  const float e = t[b] * __expf(-9.210340371976184f * (float)k / (float)half);
  y[idx] = f2a(j < half ? cosf(e) : sinf(e));
}

// (rows, ld) fp32 column -> (rows) fp32
__global__ void gather_col_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int ld, int col) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < rows) y[idx] = x[idx * ld + col];
}
// gz[r, 0] = dwav[r] * (1 - wav8[r,0]^2) ; gz[r, 1..7] = 0   (bf16, 8 channels)
__global__ void tanh_bwd_pad8_kernel(const float* __restrict__ dwav, const float* __restrict__ wav8, act_t* __restrict__ gz,
                                     long long rows) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows) return;
  const float w = wav8[idx * 8];
  const float g = dwav[idx] * (1.f - w * w);
  reinterpret_cast<uint4*>(gz)[idx] = make_uint4((uint32_t)f2a(g), 0u, 0u, 0u);
}
// v (rows) fp32 -> (rows, 8) bf16 with channel 0 = v*scale
__global__ void scatter_col_pad8_kernel(const float* __restrict__ v, act_t* __restrict__ y, long long rows, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows) return;
  reinterpret_cast<uint4*>(y)[idx] = make_uint4((uint32_t)f2a(v[idx] * scale), 0u, 0u, 0u);
}

// (rows, ld) fp32 column -> (rows) fp16
__global__ void gather_col_f32_to_act_kernel(const float* __restrict__ x, act_t* __restrict__ y, long long rows, int ld, int col) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < rows) y[idx] = f2a(x[idx * ld + col]);
}
// (rows) fp16 -> (rows, 8) fp16 with channel 0 = v
__global__ void pad_col8_act_kernel(const act_t* __restrict__ v, act_t* __restrict__ y, long long rows) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows) return;
  reinterpret_cast<uint4*>(y)[idx] = make_uint4((uint32_t)v[idx], 0u, 0u, 0u);
}

// Single-launch GroupNorm (+SiLU) for small images (the U-Net levels: 64 .. 1000 pixels): one workgroup per (group, image)
// keeps its P x C/G slice in registers, two-pass mean / variance, normalises and writes.  Replaces three launches
// (partial, finalize, apply) whose cost at these sizes is launch latency, not bytes.
constexpr int GN_SMALL_MAXU = 16;      // 4-channel units per thread
__global__ __launch_bounds__(256) void gn_small_kernel(const act_t* __restrict__ x, act_t* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ stats, float* __restrict__ scale, float* __restrict__ shift,
                                                       int P, int C, int G, float eps, int silu) {
  __shared__ float sh[16];
  const int g = blockIdx.x, b = blockIdx.y, cpg = C / G, U = cpg >> 2, c0 = g * cpg;
  const int units = P * U;
  const act_t* xb = x + (long long)b * P * C + c0;
  uint2 v[GN_SMALL_MAXU];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < GN_SMALL_MAXU; ++i) {
    const int u = threadIdx.x + i * 256;
    v[i] = make_uint2(0, 0);
    if (u < units) {
      const int px = u / U, cu = u - px * U;
      v[i] = *reinterpret_cast<const uint2*>(xb + (long long)px * C + cu * 4);
      sum += alo(v[i].x) + ahi(v[i].x) + alo(v[i].y) + ahi(v[i].y);
    }
  }
  const float n = (float)units * 4.f;
  const float mean = block_sum(sum, sh) / n;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < GN_SMALL_MAXU; ++i) {
    if (threadIdx.x + i * 256 < units) {
      const float d0 = alo(v[i].x) - mean, d1 = ahi(v[i].x) - mean, d2 = alo(v[i].y) - mean, d3 = ahi(v[i].y) - mean;
      m2 += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
  }
  const float rstd = rsqrtf(block_sum(m2, sh) / n + eps);
  if (threadIdx.x == 0) {
    stats[((long long)b * G + g) * 2] = mean;
    stats[((long long)b * G + g) * 2 + 1] = rstd;
  }
  if (threadIdx.x < cpg) {
    const int c = c0 + threadIdx.x;
    const float a = rstd * gamma[c];
    scale[(long long)b * C + c] = a;
    shift[(long long)b * C + c] = beta[c] - mean * a;
  }
  if (!y) return;
  act_t* yb = y + (long long)b * P * C + c0;
#pragma unroll
  for (int i = 0; i < GN_SMALL_MAXU; ++i) {
    const int u = threadIdx.x + i * 256;
    if (u < units) {
      const int px = u / U, cu = u - px * U;
      const float4 ga = *reinterpret_cast<const float4*>(gamma + c0 + cu * 4);
      const float4 be = *reinterpret_cast<const float4*>(beta + c0 + cu * 4);
      float z0 = (alo(v[i].x) - mean) * rstd * ga.x + be.x, z1 = (ahi(v[i].x) - mean) * rstd * ga.y + be.y;
      float z2 = (alo(v[i].y) - mean) * rstd * ga.z + be.z, z3 = (ahi(v[i].y) - mean) * rstd * ga.w + be.w;
      if (silu) { z0 = silu_f(z0); z1 = silu_f(z1); z2 = silu_f(z2); z3 = silu_f(z3); }
      *reinterpret_cast<uint2*>(yb + (long long)px * C + cu * 4) = make_uint2(pack2a(z0, z1), pack2a(z2, z3));
    }
  }
}

inline void gn_geom(int P, int C, int& nt, int& rpb, int& nchunk, int& ppb) {
  const int cpr = C >> 3;
  rpb = cpr >= 256 ? 1 : 256 / cpr;
  nt = cpr * rpb;
  nchunk = cdiv(P, rpb * 8);
  if (nchunk > DMX_GN_MAX_CHUNKS) nchunk = DMX_GN_MAX_CHUNKS;
  if (nchunk < 1) nchunk = 1;
  ppb = cdiv(P, nchunk);
  nchunk = cdiv(P, ppb);
}

}  // namespace

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH)

size_t dmx_gn_part_floats(int B, int P, int N) { return (size_t)B * ((size_t)(P + 31) / 32 + 1) * (size_t)(N / 4) * 2; }
size_t dmx_gn_scratch_floats(int B, int C, int G) {
  return (size_t)B * DMX_GN_MAX_CHUNKS * G * 2;
}

int dmx_groupnorm_fwd(const act_t* x, act_t* y, const float* gamma, const float* beta, float* stats, float* scale,
                      float* shift, float* partial, int B, int P, int C, int G, float eps, int silu, hipStream_t st, const GnParts* parts) {
  if ((C & 7) || C % G || G > 64 || (G & (G - 1)) || C > 2048) return DMX_ERR_SHAPE;
  const int cpg = C / G;
  if (parts && parts->n > 0 && (cpg & 3) == 0) {
    // the producers of x left its partial sums (EPI_GNSTATS): combine them, no pass over x for the statistics
    int nt, rpb, nchunk, ppb;
    gn_geom(P, C, nt, rpb, nchunk, ppb);
    // float2 elements of one image's partial sums (worst image: one slot more than P / tm)
    long long f2 = 0;
    for (int r = 0; r < parts->n; ++r) {
      if ((long long)B * parts->r[r].P >= (1ll << 31) / 2) return DMX_ERR_SHAPE;        // (32-bit row arithmetic in the combine)
      f2 += (long long)(parts->r[r].P / parts->r[r].tm + 2) * parts->r[r].nq;
    }
    const int stage = f2 * 8 <= 144 * 1024 ? (int)f2 : 0;                           // staged through LDS when it fits
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_parts_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_parts_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
      attr_set = true;
    }
    if (y && P <= 8192 && (C >> 3) <= 256) {      // mid-size images: finalize + apply in one launch
      if (nchunk > GN_FUSED_MAXCHUNK) { nchunk = GN_FUSED_MAXCHUNK; ppb = cdiv(P, nchunk); nchunk = cdiv(P, ppb); }
      hipLaunchKernelGGL(gn_parts_kernel<256>, dim3(nchunk, B), dim3(256), (size_t)stage * 8, st, x, y, *parts, gamma, beta, stats, scale, shift,
                         P, C, G, rpb, ppb, eps, silu, stage);
      return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(gn_parts_kernel<1024>, dim3(1, B), dim3(1024), (size_t)stage * 8, st, x, (act_t*)nullptr, *parts, gamma, beta, stats, scale,
                       shift, P, C, G, rpb, ppb, eps, silu, stage);
    if (y) hipLaunchKernelGGL(gn_apply_kernel, dim3(nchunk, B), dim3(nt), 0, st, x, scale, shift, y, P, C, rpb, ppb, silu);
    return CHECK_LAUNCH();
  }
  static const bool small_ok = getenv("DMX_NO_GN_SMALL") == nullptr, fused_ok = getenv("DMX_NO_GN_FUSED") == nullptr;
  // one workgroup per (group, image) holds its slice in registers: one launch, but its 8-byte pieces of 2C-byte rows are the
  // worst case for the memory pipeline -- it wins only where the tensor is tiny (<= 512 pixels: launch latency rules)
  const bool small_fits = (cpg & 3) == 0 && cpg <= 256 && (long long)P * (cpg >> 2) <= 256ll * GN_SMALL_MAXU && (long long)B * G >= 128;
  const bool mid = y && fused_ok && P >= 512 && P <= 8192 && (C >> 3) <= 256;
  if (small_ok && small_fits && !mid) {
    hipLaunchKernelGGL(gn_small_kernel, dim3(G, B), dim3(256), 0, st, x, y, gamma, beta, stats, scale, shift, P, C, G, eps, silu);
    return CHECK_LAUNCH();
  }
  int nt, rpb, nchunk, ppb;
  gn_geom(P, C, nt, rpb, nchunk, ppb);
  if (mid) {
    // mid-size images: row-coalesced statistics, then finalize + apply in one launch (at most GN_FUSED_MAXCHUNK chunks per image)
    const int nt2 = nt < 64 ? 64 : nt;
    int cmax = GN_FUSED_MAXCHUNK;
    while (cmax > 1 && cmax * G > 8 * nt2) --cmax;
    if (nchunk > cmax) { nchunk = cmax; ppb = cdiv(P, nchunk); nchunk = cdiv(P, ppb); }
    hipLaunchKernelGGL(gn_partial_kernel<0>, dim3(nchunk, B), dim3(nt), (size_t)rpb * C * 2 * sizeof(float), st, x,
                       (const act_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, partial,
                       P, C, G, rpb, ppb, 0);
    hipLaunchKernelGGL(gn_finalize_apply_kernel, dim3(nchunk, B), dim3(nt2), 0, st, x, partial, gamma, beta,
                       stats, scale, shift, y, P, C, G, nchunk, rpb, ppb, eps, silu);
    return CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(gn_partial_kernel<0>, dim3(nchunk, B), dim3(nt), (size_t)rpb * C * 2 * sizeof(float), st, x,
                     (const act_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, partial,
                     P, C, G, rpb, ppb, 0);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(1024), 0, st, partial, gamma, beta, stats, scale, shift, P, C, G,
                     nchunk, ppb, eps);
  if (y) hipLaunchKernelGGL(gn_apply_kernel, dim3(nchunk, B), dim3(nt), 0, st, x, scale, shift, y, P, C, rpb, ppb, silu);
  return CHECK_LAUNCH();
}

int dmx_groupnorm_bwd(const act_t* x, const act_t* dy, const act_t* add, act_t* dx, const float* stats,
                      const float* scale, const float* shift, float* k0, float* k1, float* partial, int B, int P, int C,
                      int G, int silu, hipStream_t st, const GnParts* parts) {
  if ((C & 7) || C % G || G > 64 || (G & (G - 1)) || C > 2048) return DMX_ERR_SHAPE;
  int nt, rpb, nchunk, ppb;
  gn_geom(P, C, nt, rpb, nchunk, ppb);
  if (parts && parts->n > 0 && ((C / G) & 3) == 0) {
    // the dgrad launch that produced dy left the two backward sums per slot and quad (EPI_GNBWD): no pass over x and dy for them
    for (int r = 0; r < parts->n; ++r) if ((long long)B * parts->r[r].P >= (1ll << 30)) return DMX_ERR_SHAPE;
    hipLaunchKernelGGL(gn_bwd_parts_finalize_kernel, dim3(B), dim3(1024), 0, st, *parts, stats, k0, k1, P, C, G);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nchunk, B), dim3(nt), 0, st, x, dy, scale, shift, k0, k1, add, dx, P, C,
                       rpb, ppb, silu);
    return CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(gn_partial_kernel<1>, dim3(nchunk, B), dim3(nt), (size_t)rpb * C * 2 * sizeof(float), st, x, dy,
                     scale, shift, stats, partial, P, C, G, rpb, ppb, silu);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(1024), 0, st, partial, stats, k0, k1, P, C, G, nchunk);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nchunk, B), dim3(nt), 0, st, x, dy, scale, shift, k0, k1, add, dx, P, C,
                     rpb, ppb, silu);
  return CHECK_LAUNCH();
}

int dmx_layernorm_fwd(const act_t* x, act_t* y, const float* gamma, const float* beta, int rows, int C, float eps,
                      hipStream_t st) {
  if (C & 7) return DMX_ERR_SHAPE;
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, gamma, beta, y, rows, C, eps);
  return CHECK_LAUNCH();
}

int dmx_softmax_fwd(const float* S, act_t* P, const float* colbias, long long rows, int N, long long lds, long long ldp,
                    int rows_per_bias, hipStream_t st) {
  if ((N & 3) || (lds & 3) || (ldp & 3) || N > 4096) return DMX_ERR_SHAPE;
  const int rpb = rows_per_bias < 1 ? 1 : rows_per_bias;
  if (N <= 1024)
    hipLaunchKernelGGL(softmax_kernel<64>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, P, colbias, rows, N, lds, ldp, rpb);
  else
    hipLaunchKernelGGL(softmax_kernel<256>, dim3((unsigned)rows), dim3(256), 0, st, S, P, colbias, rows, N, lds, ldp, rpb);
  return CHECK_LAUNCH();
}
int dmx_softmax_act(const act_t* S, act_t* P, const float* colbias, long long rows, int N, long long ldp, int rows_per_bias,
                    hipStream_t st) {
  if ((N & 3) || (ldp & 3) || N > 4096) return DMX_ERR_SHAPE;
  const int rpb = rows_per_bias < 1 ? 1 : rows_per_bias;
  if (N <= 1024)
    hipLaunchKernelGGL(softmax_act_kernel<64>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, P, colbias, rows, N, ldp, rpb);
  else
    hipLaunchKernelGGL(softmax_act_kernel<256>, dim3((unsigned)rows), dim3(256), 0, st, S, P, colbias, rows, N, ldp, rpb);
  return CHECK_LAUNCH();
}
int dmx_geglu(const act_t* x, act_t* y, long long rows, int I, hipStream_t st) {
  if (I & 7) return DMX_ERR_SHAPE;
  const long long n = rows * (I >> 3);
  hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, rows, I);
  return CHECK_LAUNCH();
}
int dmx_silu(const act_t* x, act_t* y, long long n, hipStream_t st) {
  if (n & 7) return DMX_ERR_SHAPE;
  hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, x, y, n / 8);
  return CHECK_LAUNCH();
}
int dmx_upsample_nearest(const act_t* x, act_t* y, int B, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t st) {
  if (C & 7) return DMX_ERR_SHAPE;
  const long long n = (long long)B * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(upsample_nearest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, B, Hi, Wi, Ho, Wo, C);
  return CHECK_LAUNCH();
}
int dmx_upsample2x_bwd(const act_t* dy, act_t* dx, int B, int Hi, int Wi, int C, hipStream_t st) {
  if (C & 7) return DMX_ERR_SHAPE;
  const long long n = (long long)B * Hi * Wi * (C >> 3);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, dx, B, Hi, Wi, C);
  return CHECK_LAUNCH();
}
int dmx_transpose(const act_t* in, act_t* out, int R, int C, long long ldi, long long ldo, int Z, int Zi, long long sIo,
                  long long sIi, long long sOo, long long sOi, hipStream_t st) {
  const bool vec = !((ldi | ldo | sIo | sIi | sOo | sOi) & 7) && !(((uintptr_t)in | (uintptr_t)out) & 15) && ldo >= (long long)((R + 7) & ~7);
  if (vec && R >= 1024 && C >= 1024) {
    hipLaunchKernelGGL(transpose128_kernel, dim3(cdiv(C, 128), cdiv(R, 128), Z), dim3(256), 0, st, in, out, R, C, ldi, ldo, Zi, sIo, sIi, sOo, sOi);
    return CHECK_LAUNCH();
  }
  if (vec) {
    hipLaunchKernelGGL(transpose64_kernel, dim3(cdiv(C, 64), cdiv(R, 64), Z), dim3(256), 0, st, in, out, R, C, ldi, ldo, Zi, sIo, sIi, sOo, sOi);
    return CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32), Z), dim3(256), 0, st, in, out, R, C, ldi, ldo, Zi,
                     sIo, sIi, sOo, sOi);
  return CHECK_LAUNCH();
}
int dmx_rowdot(const act_t* a, const act_t* b, float* out, long long rows, int C, long long lda, long long ldb, hipStream_t st) {
  if ((C & 7) || (lda & 7) || (ldb & 7)) return DMX_ERR_SHAPE;
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a, b, out, rows, C, lda, ldb);
  return CHECK_LAUNCH();
}
int dmx_copy_channels(const act_t* src, act_t* dst, long long rows, int C, int lds, int ldd, int soff, int doff,
                      hipStream_t st) {
  if ((C & 7) || (lds & 7) || (ldd & 7) || (soff & 7) || (doff & 7)) return DMX_ERR_SHAPE;
  const long long n = rows * (C >> 3);
  hipLaunchKernelGGL(copy_channels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, rows, C, lds,
                     ldd, soff, doff);
  return CHECK_LAUNCH();
}
int dmx_concat2(const act_t* a, const act_t* b, act_t* dst, long long rows, int Ca, int Cb, hipStream_t st) {
  if ((Ca & 7) || (Cb & 7)) return DMX_ERR_SHAPE;
  const long long n = rows * ((Ca + Cb) >> 3);
  hipLaunchKernelGGL(concat2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, dst, rows, Ca, Cb);
  return CHECK_LAUNCH();
}
int dmx_axpby(const act_t* x, const act_t* y0, act_t* y, float a, float b, long long n, hipStream_t st) {
  if (n & 7) return DMX_ERR_SHAPE;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, x, y0, y, a, b, n / 8);
  return CHECK_LAUNCH();
}
int dmx_nchw_f32_to_nhwc_bf16(const float* x, act_t* y, int B, int C, int HW, int Cp, float scale, hipStream_t st) {
  const long long n = (long long)B * HW * Cp;
  hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, B, C, HW, Cp, scale);
  return CHECK_LAUNCH();
}
int dmx_nhwc_bf16_to_nchw_f32(const act_t* x, float* y, int B, int C, int HW, int Cp, float scale, hipStream_t st) {
  const long long n = (long long)B * C * HW;
  hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, B, C, HW, Cp, scale);
  return CHECK_LAUNCH();
}
int dmx_f32_to_bf16(const float* x, act_t* y, long long n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n, scale);
  return CHECK_LAUNCH();
}
int dmx_bf16_to_f32(const act_t* x, float* y, long long n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n, scale);
  return CHECK_LAUNCH();
}
int dmx_extract_col(const act_t* x, float* y, long long rows, int ld, int col, hipStream_t st) {
  hipLaunchKernelGGL(extract_col_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, x, y, rows, ld, col);
  return CHECK_LAUNCH();
}
int dmx_timestep_embed(const float* t, act_t* y, int B, int dim, hipStream_t st) {
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(cdiv(B * dim, 256)), dim3(256), 0, st, t, y, B, dim);
  return CHECK_LAUNCH();
}

int dmx_gather_col_f32(const float* x, float* y, long long rows, int ld, int col, hipStream_t st) {
  hipLaunchKernelGGL(gather_col_f32_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, x, y, rows, ld, col);
  return CHECK_LAUNCH();
}
int dmx_tanh_bwd_pad8(const float* dwav, const float* wav8, act_t* gz, long long rows, hipStream_t st) {
  hipLaunchKernelGGL(tanh_bwd_pad8_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, dwav, wav8, gz, rows);
  return CHECK_LAUNCH();
}
int dmx_scatter_col_pad8(const float* v, act_t* y, long long rows, float scale, hipStream_t st) {
  hipLaunchKernelGGL(scatter_col_pad8_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, v, y, rows, scale);
  return CHECK_LAUNCH();
}

int dmx_gather_col_f32_to_act(const float* x, act_t* y, long long rows, int ld, int col, hipStream_t st) {
  hipLaunchKernelGGL(gather_col_f32_to_act_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, x, y, rows, ld, col);
  return CHECK_LAUNCH();
}
int dmx_pad_col8_act(const act_t* v, act_t* y, long long rows, hipStream_t st) {
  hipLaunchKernelGGL(pad_col8_act_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, v, y, rows);
  return CHECK_LAUNCH();
}
