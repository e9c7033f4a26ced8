// CLAP HTS-AT (Swin) audio tower, forward + input-gradient, for the style-guidance operator (BASELINE.json configs[4]).
//
// Reference intent: diffmusic/inverse_problem/operator.py:253-271 (`StyleGuidanceOperator.transform` = a Gram matrix of CLAP audio
// features; unrunnable there, run.py:213-214).  The network is transformers' `ClapAudioModel` (HTS-AT: BatchNorm over the mel bins ->
// bicubic time stretch to 1024 frames -> 256 x 256 "image" -> 4 x 4 patch embedding -> four Swin stages of window-8 attention with
// relative position bias and shifted windows, patch merging in between -> LayerNorm), restated here on the library's own kernels:
// every Linear is the LDS-DMA implicit GEMM of gemm_conv.hip (forward and dgrad), and what surrounds them is in this file --
// LayerNorm rows forward / backward (also over the 2 x 2 gathered rows of a patch merging), erf-GELU forward / backward, window
// attention forward / backward (64 tokens x head dim 24 per window and head: one wave each, scores in registers), the fused input
// stage (BatchNorm + bicubic + patch convolution + LayerNorm) and its transpose, and the Gram matrix with its gradient.
// Activations are 16-bit channels-last token rows like the U-Net's; softmax, statistics and the Gram matrix are fp32.
// Tape: per block its input, q|k|v, the post-attention hidden state and the MLP pre-activation; LayerNorm statistics and attention
// probabilities are recomputed in the backward pass.
#include "blocks.h"
#include <map>

namespace {

constexpr int WS = 8, WT = WS * WS;          // window side / tokens per window
constexpr int HD = 24;                       // head dim of every HTS-AT stage (96 / 4 = 192 / 8 = 384 / 16 = 768 / 32)

// ------------------------------------------------------------------------------------------------ LayerNorm over (gathered) rows
// A row is C channels of one token, or (merge) the 4C channels of the 2 x 2 tokens a patch merging concatenates:
// [ (2oy, 2ox) | (2oy + 1, 2ox) | (2oy, 2ox + 1) | (2oy + 1, 2ox + 1) ]  (transformers ClapAudioPatchMerging.forward)
struct RowSrc { const act_t* x; int C, merge, H, W; };
__device__ __forceinline__ long long row_chunk_off(const RowSrc& s, long long r, int chunk) {
  if (!s.merge) return r * s.C + chunk * 8;
  const int cpt = s.C >> 3, part = chunk / cpt, within = chunk - part * cpt;
  const int Wo = s.W >> 1, Ho = s.H >> 1;
  const int ox = (int)(r % Wo);
  const long long t = r / Wo;
  const int oy = (int)(t % Ho);
  const long long b = t / Ho;
  const int iy = 2 * oy + (part & 1), ix = 2 * ox + (part >> 1);
  return ((b * s.H + iy) * s.W + ix) * (long long)s.C + within * 8;
}
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  f[0] = alo(v.x); f[1] = ahi(v.x); f[2] = alo(v.y); f[3] = ahi(v.y); f[4] = alo(v.z); f[5] = ahi(v.z); f[6] = alo(v.w); f[7] = ahi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2a(f[0], f[1]), pack2a(f[2], f[3]), pack2a(f[4], f[5]), pack2a(f[6], f[7]));
}
__device__ __forceinline__ float group_sum(float v, int tpr) {
  for (int o = tpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int LN_MAXCH = 3;       // 8-channel chunks per thread: rows of up to 64 * 3 * 8 = 1536 channels (the last patch merging: 4 * 384)

// y = (x - mean) rstd gamma + beta over rows of Cw channels; tpr (a power of two <= 64) threads per row
template <typename OutT>
__global__ __launch_bounds__(256) void ln_rows_fwd_kernel(const RowSrc s, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          OutT* __restrict__ y, long long rows, int Cw, int tpr, float eps) {
  const int rpb = 256 / tpr, sub = threadIdx.x % tpr, nch = Cw >> 3;
  const long long r = (long long)blockIdx.x * rpb + threadIdx.x / tpr;
  const bool valid = r < rows;
  float v[LN_MAXCH][8];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    const int ch = sub + k * tpr;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
    if (valid && ch < nch) {
      unpack8(*reinterpret_cast<const uint4*>(s.x + row_chunk_off(s, r, ch)), v[k]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[k][e];
    }
  }
  const float mean = group_sum(sum, tpr) / (float)Cw;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    const int ch = sub + k * tpr;
    if (valid && ch < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(group_sum(sq, tpr) / (float)Cw + eps);
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    const int ch = sub + k * tpr;
    if (valid && ch < nch) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mean) * rstd * gamma[ch * 8 + e] + beta[ch * 8 + e];
      if constexpr (sizeof(OutT) == 4) {
        float* dst = reinterpret_cast<float*>(y) + r * Cw + ch * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
      } else {
        *reinterpret_cast<uint4*>(reinterpret_cast<act_t*>(y) + r * Cw + ch * 8) = pack8(o);
      }
    }
  }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma (+ add), written at the rows' SOURCE positions (a patch merging scatters
// its four tokens back; every token belongs to exactly one merged row).  Statistics are recomputed from x.
template <typename DyT>
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(const RowSrc s, const float* __restrict__ gamma, const DyT* __restrict__ dy,
                                                          const act_t* __restrict__ add, act_t* __restrict__ dx, long long rows, int Cw,
                                                          int tpr, float eps) {
  const int rpb = 256 / tpr, sub = threadIdx.x % tpr, nch = Cw >> 3;
  const long long r = (long long)blockIdx.x * rpb + threadIdx.x / tpr;
  const bool valid = r < rows;
  float v[LN_MAXCH][8], g[LN_MAXCH][8];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    const int ch = sub + k * tpr;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[k][e] = 0.f; g[k][e] = 0.f; }
    if (valid && ch < nch) {
      unpack8(*reinterpret_cast<const uint4*>(s.x + row_chunk_off(s, r, ch)), v[k]);
      if constexpr (sizeof(DyT) == 4) {
        const float* src = reinterpret_cast<const float*>(dy) + r * Cw + ch * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        g[k][0] = a.x; g[k][1] = a.y; g[k][2] = a.z; g[k][3] = a.w; g[k][4] = b.x; g[k][5] = b.y; g[k][6] = b.z; g[k][7] = b.w;
      } else {
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const act_t*>(dy) + r * Cw + ch * 8), g[k]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { sum += v[k][e]; g[k][e] *= gamma[ch * 8 + e]; }
    }
  }
  const float mean = group_sum(sum, tpr) / (float)Cw;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    const int ch = sub + k * tpr;
    if (valid && ch < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(group_sum(sq, tpr) / (float)Cw + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    const int ch = sub + k * tpr;
    if (valid && ch < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[k][e] = (v[k][e] - mean) * rstd; sg += g[k][e]; sgx += g[k][e] * v[k][e]; }
    }
  }
  const float mg = group_sum(sg, tpr) / (float)Cw, mgx = group_sum(sgx, tpr) / (float)Cw;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    const int ch = sub + k * tpr;
    if (valid && ch < nch) {
      const long long off = row_chunk_off(s, r, ch);
      float o[8], a[8];
      if (add) unpack8(*reinterpret_cast<const uint4*>(add + off), a);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (g[k][e] - mg - v[k][e] * mgx) + (add ? a[e] : 0.f);
      *reinterpret_cast<uint4*>(dx + off) = pack8(o);
    }
  }
}

int ln_tpr(int Cw) {
  const int nch = Cw >> 3;
  int tpr = 1;
  while (tpr < nch && tpr < 64) tpr <<= 1;
  return tpr;
}
template <typename OutT>
int ln_rows_fwd(const RowSrc& s, const float* gamma, const float* beta, OutT* y, long long rows, float eps, hipStream_t st) {
  const int Cw = s.merge ? 4 * s.C : s.C;
  if ((s.C & 7) || Cw > 64 * LN_MAXCH * 8) return DMX_ERR_SHAPE;
  const int tpr = ln_tpr(Cw), rpb = 256 / tpr;
  hipLaunchKernelGGL(ln_rows_fwd_kernel<OutT>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, s, gamma, beta, y, rows, Cw, tpr, eps);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}
template <typename DyT>
int ln_rows_bwd(const RowSrc& s, const float* gamma, const DyT* dy, const act_t* add, act_t* dx, long long rows, float eps, hipStream_t st) {
  const int Cw = s.merge ? 4 * s.C : s.C;
  if ((s.C & 7) || Cw > 64 * LN_MAXCH * 8 || (s.merge && add)) return DMX_ERR_SHAPE;
  const int tpr = ln_tpr(Cw), rpb = 256 / tpr;
  hipLaunchKernelGGL(ln_rows_bwd_kernel<DyT>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, s, gamma, dy, add, dx, rows, Cw, tpr, eps);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------ erf-GELU
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float u) {
  return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
}
__global__ void gelu_fwd_kernel(const act_t* __restrict__ u, act_t* __restrict__ y, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float f[8];
  unpack8(reinterpret_cast<const uint4*>(u)[i], f);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = gelu_f(f[e]);
  reinterpret_cast<uint4*>(y)[i] = pack8(f);
}
__global__ void gelu_bwd_kernel(const act_t* __restrict__ u, const act_t* __restrict__ dy, act_t* __restrict__ du, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float f[8], d[8];
  unpack8(reinterpret_cast<const uint4*>(u)[i], f);
  unpack8(reinterpret_cast<const uint4*>(dy)[i], d);
#pragma unroll
  for (int e = 0; e < 8; ++e) d[e] *= gelu_df(f[e]);
  reinterpret_cast<uint4*>(du)[i] = pack8(d);
}

// ------------------------------------------------------------------------------------------------ window attention
// One wave per (window, head); lane i is query token i of the window (and, in the backward pass, key token i).  The window's tokens are
// gathered straight from the token-major q|k|v rows: window partition, the cyclic shift of the odd blocks and their inverse are index
// arithmetic.  Scores = q k^T / sqrt(24) + relative position bias (+ -100 between tokens of different shift regions), softmax in fp32.
struct WinGeom { int B, H, W, C, heads, shift; };
__device__ __forceinline__ void win_token(const WinGeom& g, int win, int i, int& n, int& region) {
  const int nwx = g.W / WS;
  const int wy = win / nwx, wx = win - wy * nwx;
  const int ys = wy * WS + (i >> 3), xs = wx * WS + (i & 7);           // position in the shifted image
  const int y = (ys + g.shift) % g.H, x = (xs + g.shift) % g.W;        // torch.roll(-shift): shifted[ys] = image[(ys + shift) mod H]
  n = y * g.W + x;
  const int rh = (ys >= g.H - WS) + (ys >= g.H - g.shift), rw = (xs >= g.W - WS) + (xs >= g.W - g.shift);
  region = g.shift > 0 ? rh * 3 + rw : 0;
}
__device__ __forceinline__ void load_head(const act_t* p, float (&f)[HD]) {      // 24 consecutive 16-bit values (48 bytes, 16-byte aligned)
#pragma unroll
  for (int k = 0; k < HD / 8; ++k) {
    float t[8];
    unpack8(reinterpret_cast<const uint4*>(p)[k], t);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[k * 8 + e] = t[e];
  }
}
__device__ __forceinline__ void store_head(act_t* p, const float (&f)[HD]) {
#pragma unroll
  for (int k = 0; k < HD / 8; ++k) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = f[k * 8 + e];
    reinterpret_cast<uint4*>(p)[k] = pack8(t);
  }
}

__global__ __launch_bounds__(64) void win_attn_fwd_kernel(const act_t* __restrict__ qkv, act_t* __restrict__ out,
                                                          const float* __restrict__ bias_table, const WinGeom g) {
  __shared__ float sk[WT][HD + 1], sv[WT][HD + 1];
  __shared__ float sb[(2 * WS - 1) * (2 * WS - 1)];
  __shared__ int sreg[WT];
  const int i = threadIdx.x, head = blockIdx.y;
  const int nw = (g.H / WS) * (g.W / WS);
  const int b = blockIdx.x / nw, win = blockIdx.x - b * nw;
  int n, region;
  win_token(g, win, i, n, region);
  const long long tok = (long long)b * g.H * g.W + n;
  const act_t* row = qkv + tok * 3 * g.C + head * HD;
  float q[HD], t[HD];
  load_head(row, q);
  load_head(row + g.C, t);
#pragma unroll
  for (int d = 0; d < HD; ++d) sk[i][d] = t[d];
  load_head(row + 2 * g.C, t);
#pragma unroll
  for (int d = 0; d < HD; ++d) sv[i][d] = t[d];
  for (int k = i; k < (2 * WS - 1) * (2 * WS - 1); k += 64) sb[k] = bias_table[k * g.heads + head];
  sreg[i] = region;
  __syncthreads();
  const float scale = 0.20412414523193154f;                 // 1 / sqrt(24)
#pragma unroll
  for (int d = 0; d < HD; ++d) q[d] *= scale;
  const int iy = i >> 3, ix = i & 7;
  float s[WT];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < WT; ++j) {
    float a = sb[(iy - (j >> 3) + WS - 1) * (2 * WS - 1) + (ix - (j & 7) + WS - 1)];
#pragma unroll
    for (int d = 0; d < HD; ++d) a = __builtin_fmaf(q[d], sk[j][d], a);
    if (sreg[j] != region) a += -100.f;
    s[j] = a;
    mx = fmaxf(mx, a);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < WT; ++j) { s[j] = __expf(s[j] - mx); sum += s[j]; }
  const float inv = 1.f / sum;
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < WT; ++j) {
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = __builtin_fmaf(s[j], sv[j][d], o[d]);
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] *= inv;
  store_head(out + tok * g.C + head * HD, o);
}

// dq | dk | dv of one (window, head) from the tape's q|k|v and the gradient of the attention output (token-major, C wide)
__global__ __launch_bounds__(64) void win_attn_bwd_kernel(const act_t* __restrict__ qkv, const act_t* __restrict__ dout,
                                                          act_t* __restrict__ dqkv, const float* __restrict__ bias_table, const WinGeom g) {
  __shared__ float sq[WT][HD + 1], sk[WT][HD + 1], sv[WT][HD + 1], sdo[WT][HD + 1];
  __shared__ float sp[WT][WT + 1], sds[WT][WT + 1];
  __shared__ float sb[(2 * WS - 1) * (2 * WS - 1)];
  __shared__ int sreg[WT];
  const int i = threadIdx.x, head = blockIdx.y;
  const int nw = (g.H / WS) * (g.W / WS);
  const int b = blockIdx.x / nw, win = blockIdx.x - b * nw;
  int n, region;
  win_token(g, win, i, n, region);
  const long long tok = (long long)b * g.H * g.W + n;
  const act_t* row = qkv + tok * 3 * g.C + head * HD;
  const float scale = 0.20412414523193154f;
  float q[HD], dO[HD], t[HD];
  load_head(row, q);
#pragma unroll
  for (int d = 0; d < HD; ++d) { q[d] *= scale; sq[i][d] = q[d]; }
  load_head(row + g.C, t);
#pragma unroll
  for (int d = 0; d < HD; ++d) sk[i][d] = t[d];
  load_head(row + 2 * g.C, t);
#pragma unroll
  for (int d = 0; d < HD; ++d) sv[i][d] = t[d];
  load_head(dout + tok * g.C + head * HD, dO);
#pragma unroll
  for (int d = 0; d < HD; ++d) sdo[i][d] = dO[d];
  for (int k = i; k < (2 * WS - 1) * (2 * WS - 1); k += 64) sb[k] = bias_table[k * g.heads + head];
  sreg[i] = region;
  __syncthreads();
  const int iy = i >> 3, ix = i & 7;
  // this lane's row of scores / probabilities lives in LDS (sp[i][.], stride 65: lane i and column j hit bank (i + j) mod 64), not in 64
  // registers: with them the kernel spilled 4.6 KB per lane
  float mx = -3.0e38f;
#pragma unroll 4
  for (int j = 0; j < WT; ++j) {
    float a = sb[(iy - (j >> 3) + WS - 1) * (2 * WS - 1) + (ix - (j & 7) + WS - 1)];
#pragma unroll
    for (int d = 0; d < HD; ++d) a = __builtin_fmaf(q[d], sk[j][d], a);
    if (sreg[j] != region) a += -100.f;
    sp[i][j] = a;
    mx = fmaxf(mx, a);
  }
  float sum = 0.f;
#pragma unroll 4
  for (int j = 0; j < WT; ++j) { const float e = __expf(sp[i][j] - mx); sp[i][j] = e; sum += e; }
  const float inv = 1.f / sum;
  // dP[j] = dO . v_j ; delta = sum_j P[j] dP[j] ; dS[j] = P[j] (dP[j] - delta)
  float delta = 0.f;
#pragma unroll 4
  for (int j = 0; j < WT; ++j) {
    const float pj = sp[i][j] * inv;
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) a = __builtin_fmaf(dO[d], sv[j][d], a);
    sp[i][j] = pj;
    sds[i][j] = a;                                   // dP for now
    delta = __builtin_fmaf(pj, a, delta);
  }
  float dq[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) dq[d] = 0.f;
#pragma unroll 4
  for (int j = 0; j < WT; ++j) {
    const float ds = sp[i][j] * (sds[i][j] - delta);
    sds[i][j] = ds;
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = __builtin_fmaf(ds, sk[j][d], dq[d]);
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) dq[d] *= scale;
  __syncthreads();
  // lane i as KEY i: dk = sum_q dS[q][i] (scale q_q) ; dv = sum_q P[q][i] dO_q
  float dk[HD], dv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  for (int qi = 0; qi < WT; ++qi) {
    const float ds = sds[qi][i], p = sp[qi][i];
#pragma unroll
    for (int d = 0; d < HD; ++d) { dk[d] = __builtin_fmaf(ds, sq[qi][d], dk[d]); dv[d] = __builtin_fmaf(p, sdo[qi][d], dv[d]); }
  }
  act_t* drow = dqkv + tok * 3 * g.C + head * HD;
  store_head(drow, dq);
  store_head(drow + g.C, dk);
  store_head(drow + 2 * g.C, dv);
}

// ------------------------------------------------------------------------------------------------ input stage
// mel (B, frames, 64) fp32 log-mel -> BatchNorm2d over the mel bins (eval) -> bicubic stretch of the time axis to 1024 frames
// (align_corners = True; taps and weights tabulated on the host, torch's cubic convolution with A = -0.75) -> the 256 x 256 image
// img[c * 64 + f][tt] = X[c * 256 + tt][f] (ClapAudioEncoder.reshape_mel2img) -> Conv2d(1, E, 4, stride 4) -> LayerNorm(E): one thread per
// token; a channel of the convolution is 16 multiply-adds of LDS-resident weights and is recomputed where it is needed (statistics, output,
// backward sums) rather than held in E registers.  E <= 128, a multiple of 8.
constexpr int EMB_MAX = 128;
struct EmbedParams {
  const float* mel; const int* tidx; const float* tw; const float* bn_a; const float* bn_b;
  const float* Wp; const float* bp; const float* ln_g; const float* ln_b;
  int B, frames, E, bins, grid;          // grid = tokens per image side (64), bins = mel bins (64)
  float eps;
};
// the 16 BatchNorm'ed, time-stretched pixels of one token (4 mel bins x 4 stretched frames; pix[dy * 4 + dx])
__device__ __forceinline__ void embed_pixels(const EmbedParams& p, int b, int tok, float (&pix)[16]) {
  const int ty = tok / p.grid, tx = tok - ty * p.grid;
  const int cpb = p.bins / 4;                               // token rows per time chunk
  const int c = ty / cpb, f0 = (ty - c * cpb) * 4;
  const int T = p.grid * 4;                                 // image columns = frames per chunk (256)
  const float* m = p.mel + (long long)b * p.frames * p.bins + f0;
  const float4 ba = *reinterpret_cast<const float4*>(p.bn_a + f0), bb = *reinterpret_cast<const float4*>(p.bn_b + f0);
#pragma unroll
  for (int dx = 0; dx < 4; ++dx) {
    const int tt = c * T + tx * 4 + dx;
    const int4 id = reinterpret_cast<const int4*>(p.tidx)[tt];
    const float4 w = reinterpret_cast<const float4*>(p.tw)[tt];
    const float4 a0 = *reinterpret_cast<const float4*>(m + (long long)id.x * p.bins), a1 = *reinterpret_cast<const float4*>(m + (long long)id.y * p.bins);
    const float4 a2 = *reinterpret_cast<const float4*>(m + (long long)id.z * p.bins), a3 = *reinterpret_cast<const float4*>(m + (long long)id.w * p.bins);
    pix[0 * 4 + dx] = ba.x * (w.x * a0.x + w.y * a1.x + w.z * a2.x + w.w * a3.x) + bb.x;
    pix[1 * 4 + dx] = ba.y * (w.x * a0.y + w.y * a1.y + w.z * a2.y + w.w * a3.y) + bb.y;
    pix[2 * 4 + dx] = ba.z * (w.x * a0.z + w.y * a1.z + w.z * a2.z + w.w * a3.z) + bb.z;
    pix[3 * 4 + dx] = ba.w * (w.x * a0.w + w.y * a1.w + w.z * a2.w + w.w * a3.w) + bb.w;
  }
}
// channel n of the patch convolution (sW: [E][16] weights, then E biases); recomputed wherever it is needed instead of kept in E registers
__device__ __forceinline__ float embed_chan(const float* sW, int E, int n, const float (&pix)[16]) {
  float a = sW[E * 16 + n];
#pragma unroll
  for (int k = 0; k < 16; ++k) a = __builtin_fmaf(sW[n * 16 + k], pix[k], a);
  return a;
}
__device__ __forceinline__ void embed_load_weights(const EmbedParams& p, float* sW) {
  const int E = p.E;
  for (int k = threadIdx.x; k < E * 16; k += blockDim.x) sW[k] = p.Wp[k];
  for (int k = threadIdx.x; k < E; k += blockDim.x) { sW[E * 16 + k] = p.bp[k]; sW[E * 17 + k] = p.ln_g[k]; sW[E * 18 + k] = p.ln_b[k]; }
  __syncthreads();
}
__device__ __forceinline__ void embed_stats(const float* sW, int E, const float (&pix)[16], float eps, float& mean, float& rstd) {
  float s1 = 0.f;
  for (int n = 0; n < E; ++n) s1 += embed_chan(sW, E, n, pix);
  mean = s1 / (float)E;
  float s2 = 0.f;
  for (int n = 0; n < E; ++n) { const float d = embed_chan(sW, E, n, pix) - mean; s2 = __builtin_fmaf(d, d, s2); }
  rstd = rsqrtf(s2 / (float)E + eps);
}
__global__ __launch_bounds__(128) void embed_fwd_kernel(const EmbedParams p, act_t* __restrict__ tokens) {
  __shared__ float sW[EMB_MAX * 19];
  embed_load_weights(p, sW);
  const int E = p.E, ntok = p.grid * p.grid;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)p.B * ntok) return;
  const int b = (int)(gid / ntok), tok = (int)(gid - (long long)b * ntok);
  float pix[16];
  embed_pixels(p, b, tok, pix);
  float mean, rstd;
  embed_stats(sW, E, pix, p.eps, mean, rstd);
  act_t* dst = tokens + gid * E;
  for (int k = 0; k < E / 8; ++k) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int n = k * 8 + e; o[e] = (embed_chan(sW, E, n, pix) - mean) * rstd * sW[E * 17 + n] + sW[E * 18 + n]; }
    reinterpret_cast<uint4*>(dst)[k] = pack8(o);
  }
}
// d tokens -> d image (B, 1024, 64) fp32, already multiplied by the BatchNorm scale (every pixel belongs to exactly one token)
__global__ __launch_bounds__(128) void embed_bwd_kernel(const EmbedParams p, const act_t* __restrict__ dtok, float* __restrict__ dimg) {
  __shared__ float sW[EMB_MAX * 19];
  embed_load_weights(p, sW);
  const int E = p.E, ntok = p.grid * p.grid;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)p.B * ntok) return;
  const int b = (int)(gid / ntok), tok = (int)(gid - (long long)b * ntok);
  float pix[16];
  embed_pixels(p, b, tok, pix);
  float mean, rstd;
  embed_stats(sW, E, pix, p.eps, mean, rstd);
  const act_t* src = dtok + gid * E;
  float gsum = 0.f, gxsum = 0.f;
  for (int k = 0; k < E / 8; ++k) {          // LayerNorm backward sums: g = d token * gamma
    float d[8];
    unpack8(reinterpret_cast<const uint4*>(src)[k], d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = k * 8 + e;
      const float xh = (embed_chan(sW, E, n, pix) - mean) * rstd, gg = d[e] * sW[E * 17 + n];
      gsum += gg; gxsum = __builtin_fmaf(gg, xh, gxsum);
    }
  }
  const float mg = gsum / (float)E, mgx = gxsum / (float)E;
  float dp[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) dp[k] = 0.f;
  for (int k = 0; k < E / 8; ++k) {
    float d[8];
    unpack8(reinterpret_cast<const uint4*>(src)[k], d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = k * 8 + e;
      const float xh = (embed_chan(sW, E, n, pix) - mean) * rstd, gg = d[e] * sW[E * 17 + n];
      const float dpre = rstd * (gg - mg - xh * mgx);
#pragma unroll
      for (int q = 0; q < 16; ++q) dp[q] = __builtin_fmaf(sW[n * 16 + q], dpre, dp[q]);
    }
  }
  const int ty = tok / p.grid, tx = tok - ty * p.grid;
  const int cpb = p.bins / 4;
  const int c = ty / cpb, f0 = (ty - c * cpb) * 4;
  const int T = p.grid * 4;
  const float4 ba = *reinterpret_cast<const float4*>(p.bn_a + f0);
#pragma unroll
  for (int dx = 0; dx < 4; ++dx) {
    const int tt = c * T + tx * 4 + dx;
    float* dst = dimg + ((long long)b * (4 * T) + tt) * p.bins + f0;
    *reinterpret_cast<float4*>(dst) = make_float4(ba.x * dp[0 * 4 + dx], ba.y * dp[1 * 4 + dx], ba.z * dp[2 * 4 + dx], ba.w * dp[3 * 4 + dx]);
  }
}
// transpose of the bicubic stretch: dmel[b][k][f] = scale[b] * sum over the stretched frames t that read frame k of w * dimg[b][t][f]
__global__ void interp_bwd_kernel(const float* __restrict__ dimg, const int* __restrict__ kstart, const int* __restrict__ kt,
                                  const float* __restrict__ kw, const float* __restrict__ scale, float* __restrict__ dmel, int B, int frames,
                                  int bins, int Tout) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)B * frames * bins) return;
  const int f = (int)(gid % bins);
  const long long r = gid / bins;
  const int k = (int)(r % frames), b = (int)(r / frames);
  float a = 0.f;
  for (int e = kstart[k]; e < kstart[k + 1]; ++e) a = __builtin_fmaf(kw[e], dimg[((long long)b * Tout + kt[e]) * bins + f], a);
  dmel[gid] = a * (scale ? scale[b] : 1.f);
}

// ------------------------------------------------------------------------------------------------ Gram matrix of token features
// G[b] = F[b]^T F[b] / T with F (B, T, C) fp32 (T = 64 tokens, C = 768): 64 x 64 tile per workgroup, all T rows of both column panels in LDS
__global__ __launch_bounds__(256) void gram_fwd_kernel(const float* __restrict__ F, float* __restrict__ G, int T, int C) {
  extern __shared__ float sm[];
  float* sa = sm;                 // [T][64]
  float* sbb = sm + T * 64;       // [T][64]
  const int b = blockIdx.z, i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const float* Fb = F + (long long)b * T * C;
  for (int k = threadIdx.x; k < T * 64; k += 256) {
    const int t = k >> 6, c = k & 63;
    sa[k] = i0 + c < C ? Fb[(long long)t * C + i0 + c] : 0.f;
    sbb[k] = j0 + c < C ? Fb[(long long)t * C + j0 + c] : 0.f;
  }
  __syncthreads();
  const int ti = (threadIdx.x >> 4) * 4, tj = (threadIdx.x & 15) * 4;
  float acc[4][4] = {};
  for (int t = 0; t < T; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(sa + t * 64 + ti), c = *reinterpret_cast<const float4*>(sbb + t * 64 + tj);
    const float av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_fmaf(av[u], cv[v], acc[u][v]);
  }
  const float inv = 1.f / (float)T;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (i0 + ti + u < C && j0 + tj + v < C) G[((long long)b * C + i0 + ti + u) * C + j0 + tj + v] = acc[u][v] * inv;
}
// dF[b][t][c] = (1 / T) sum_c' F[b][t][c'] (dG[b][c'][c] + dG[b][c][c'])
__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* __restrict__ F, const float* __restrict__ dG, float* __restrict__ dF, int T, int C) {
  __shared__ float sf[64][33];          // F[t][c' chunk]
  __shared__ float ss[32][65];          // S[c' chunk][c tile]
  const int b = blockIdx.z, t0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const float* Fb = F + (long long)b * T * C;
  const float* Gb = dG + (long long)b * C * C;
  const int tt = (threadIdx.x >> 4) * 4, tc = (threadIdx.x & 15) * 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < C; k0 += 32) {
    for (int k = threadIdx.x; k < 64 * 32; k += 256) {
      const int t = k >> 5, c = k & 31;
      sf[t][c] = (t0 + t < T && k0 + c < C) ? Fb[(long long)(t0 + t) * C + k0 + c] : 0.f;
    }
    for (int k = threadIdx.x; k < 32 * 64; k += 256) {           // dG[c'][c]: rows c', coalesced over c
      const int cp = k >> 6, c = k & 63;
      ss[cp][c] = (k0 + cp < C && c0 + c < C) ? Gb[(long long)(k0 + cp) * C + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 32 * 64; k += 256) {           // + dG[c][c']: rows c, coalesced over c' (the transposed walk of the first
      const int c = k >> 5, cp = k & 31;                         // form read one element per 3 KB row: 170 us per launch)
      if (k0 + cp < C && c0 + c < C) ss[cp][c] += Gb[(long long)(c0 + c) * C + k0 + cp];
    }
    __syncthreads();
#pragma unroll 4
    for (int cp = 0; cp < 32; ++cp) {
      float fv[4], sv4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { fv[u] = sf[tt + u][cp]; sv4[u] = ss[cp][tc + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_fmaf(fv[u], sv4[v], acc[u][v]);
    }
    __syncthreads();
  }
  const float inv = 1.f / (float)T;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (t0 + tt + u < T && c0 + tc + v < C) dF[((long long)b * T + t0 + tt + u) * C + c0 + tc + v] = acc[u][v] * inv;
}

// torch's bicubic coefficients (UpSampleKernel.h cubic_convolution1 / 2, A = -0.75)
inline void cubic_coeffs(double t, double (&w)[4]) {
  const double A = -0.75;
  auto c1 = [&](double x) { return ((A + 2) * x - (A + 3)) * x * x + 1; };
  auto c2 = [&](double x) { return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A; };
  w[0] = c2(t + 1.0); w[1] = c1(t); w[2] = c1(1.0 - t); w[3] = c2(2.0 - t);
}

struct InterpTables { int* tidx = nullptr; float* tw = nullptr; int* kstart = nullptr; int* kt = nullptr; float* kw = nullptr; };

}  // namespace

// ==================================================================================================== executor
struct HtsatBlock {
  int C = 0, heads = 0, shift = 0;
  int ln1_g = -1, ln1_b = -1, ln2_g = -1, ln2_b = -1, rpb = -1;
  ConvLayer q, k, v, qkv, proj, fc1, fc2;
};
struct HtsatStage {
  int C = 0, H = 0, W = 0, heads = 0;
  std::vector<HtsatBlock> blocks;
  bool merge = false;
  int mg = -1, mb = -1;
  ConvLayer reduction;
};
struct HtsatBlockTape { const act_t* x_in = nullptr; act_t* qkv = nullptr; act_t* x_mid = nullptr; act_t* u = nullptr; };

struct Htsat : Model {
  dmx_htsat_config cfg;
  int bn_w = -1, bn_b = -1, bn_m = -1, bn_v = -1, pe_w = -1, pe_b = -1, pe_g = -1, pe_bb = -1, fn_g = -1, fn_b = -1;
  float *bn_a = nullptr, *bn_c = nullptr;        // BatchNorm folded to a * x + c per mel bin
  std::vector<HtsatStage> stages;
  std::map<int, InterpTables> tables;            // per input frame count
  int grid0 = 0, Cfinal = 0, Tfinal = 0;
  // tape
  int B = 0, frames = 0;
  bool have_tape = false;
  const float* t_mel = nullptr;
  std::vector<std::vector<HtsatBlockTape>> tape;
  std::vector<const act_t*> t_stage_out;         // input of each stage's patch merging
  const act_t* t_final_in = nullptr;

  explicit Htsat(const dmx_htsat_config& c) : cfg(c) {
    kind = DMX_MODEL_HTSAT;
    const std::string p = "audio_encoder.";
    bn_w = ps.add(p + "batch_norm.weight", {c.num_mel_bins}); bn_b = ps.add(p + "batch_norm.bias", {c.num_mel_bins});
    bn_m = ps.add(p + "batch_norm.running_mean", {c.num_mel_bins}); bn_v = ps.add(p + "batch_norm.running_var", {c.num_mel_bins});
    const int E = c.embed_dim;
    pe_w = ps.add(p + "patch_embed.proj.weight", {E, 1, 4, 4}); pe_b = ps.add(p + "patch_embed.proj.bias", {E});
    pe_g = ps.add(p + "patch_embed.norm.weight", {E}); pe_bb = ps.add(p + "patch_embed.norm.bias", {E});
    grid0 = c.spec_size / 4;
    int C = E, H = grid0, W = grid0;
    for (int s = 0; s < c.num_stages; ++s) {
      HtsatStage st;
      st.C = C; st.H = H; st.W = W; st.heads = c.num_heads[s];
      for (int j = 0; j < c.depths[s]; ++j) {
        HtsatBlock b;
        b.C = C; b.heads = st.heads;
        b.shift = (j % 2 == 1 && (H < W ? H : W) > WS) ? WS / 2 : 0;     // (set_shift_and_window_size: no shift when the grid is one window)
        const std::string bp = p + "layers." + std::to_string(s) + ".blocks." + std::to_string(j);
        b.ln1_g = ps.add(bp + ".layernorm_before.weight", {C}); b.ln1_b = ps.add(bp + ".layernorm_before.bias", {C});
        b.rpb = ps.add(bp + ".attention.self.relative_position_bias_table", {(2 * WS - 1) * (2 * WS - 1), st.heads});
        b.q = make_linear(ps, bp + ".attention.self.query", C, C, true, true);
        b.k = make_linear(ps, bp + ".attention.self.key", C, C, true, true);
        b.v = make_linear(ps, bp + ".attention.self.value", C, C, true, true);
        b.proj = make_linear(ps, bp + ".attention.output.dense", C, C, true, true);
        b.ln2_g = ps.add(bp + ".layernorm_after.weight", {C}); b.ln2_b = ps.add(bp + ".layernorm_after.bias", {C});
        b.fc1 = make_linear(ps, bp + ".intermediate.dense", C, 4 * C, true, true);
        b.fc2 = make_linear(ps, bp + ".output.dense", 4 * C, C, true, true);
        st.blocks.push_back(b);
      }
      st.merge = s + 1 < c.num_stages;
      if (st.merge) {
        const std::string dp = p + "layers." + std::to_string(s) + ".downsample";
        st.reduction = make_linear(ps, dp + ".reduction", 4 * C, 2 * C, false, true);
        st.mg = ps.add(dp + ".norm.weight", {4 * C}); st.mb = ps.add(dp + ".norm.bias", {4 * C});
      }
      stages.push_back(st);
      if (st.merge) { C *= 2; H /= 2; W /= 2; }
    }
    Cfinal = C; Tfinal = H * W;
    fn_g = ps.add(p + "norm.weight", {C}); fn_b = ps.add(p + "norm.bias", {C});
  }
  ~Htsat() override {
    for (auto& kv : tables) {
      (void)hipFree(kv.second.tidx); (void)hipFree(kv.second.tw); (void)hipFree(kv.second.kstart); (void)hipFree(kv.second.kt); (void)hipFree(kv.second.kw);
    }
  }

  int finalize(hipStream_t st) override {
    for (auto& s : stages) {
      for (auto& b : s.blocks) {
        const ConvLayer* src[3] = {&b.q, &b.k, &b.v};
        CTRY(pack_linear_stack(ps, b.qkv, src, 3, st));
        CTRY(pack_layer(ps, b.proj, st)); CTRY(pack_layer(ps, b.fc1, st)); CTRY(pack_layer(ps, b.fc2, st));
      }
      if (s.merge) CTRY(pack_layer(ps, s.reduction, st));
    }
    // BatchNorm2d (eval) per mel bin: a = w / sqrt(var + eps), c = b - mean a   (host: 4 x 64 floats, load time)
    const int nb = cfg.num_mel_bins;
    std::vector<float> w(nb), b(nb), m(nb), v(nb), a(nb), c(nb);
    if (hipStreamSynchronize(st) != hipSuccess) return DMX_ERR_LAUNCH;
    (void)hipMemcpy(w.data(), ps.dev(bn_w), nb * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), ps.dev(bn_b), nb * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(m.data(), ps.dev(bn_m), nb * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(v.data(), ps.dev(bn_v), nb * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < nb; ++i) { a[i] = (float)((double)w[i] / std::sqrt((double)v[i] + (double)cfg.bn_eps)); c[i] = b[i] - m[i] * a[i]; }
    bn_a = (float*)ps.dalloc(nb * 4); bn_c = (float*)ps.dalloc(nb * 4);
    if (!bn_a || !bn_c) return DMX_ERR_PARAM;
    (void)hipMemcpy(bn_a, a.data(), nb * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bn_c, c.data(), nb * 4, hipMemcpyHostToDevice);
    return DMX_OK;
  }

  // bicubic tables of one input length (built once per length; load-time-like: allocates and copies synchronously)
  int get_tables(int frames_, InterpTables** out) {
    auto it = tables.find(frames_);
    if (it == tables.end()) {
      const int Tout = cfg.spec_size * (cfg.spec_size / cfg.num_mel_bins);
      std::vector<int> tidx(Tout * 4);
      std::vector<float> tw(Tout * 4);
      std::vector<std::vector<std::pair<int, float>>> rev(frames_);
      for (int t = 0; t < Tout; ++t) {
        int i0 = t; double w[4] = {0, 1, 0, 0};
        if (frames_ != Tout) {
          // area_pixel_compute_source_index with align_corners: scale = (in - 1) / (out - 1) in the tensor's (float) precision
          const float scale = Tout > 1 ? (float)(frames_ - 1) / (float)(Tout - 1) : 0.f;
          const float real = scale * (float)t;
          i0 = (int)std::floor(real);
          cubic_coeffs((double)(real - (float)i0), w);
        }
        for (int k = 0; k < 4; ++k) {
          int id = i0 - 1 + k; id = id < 0 ? 0 : (id > frames_ - 1 ? frames_ - 1 : id);
          tidx[t * 4 + k] = id; tw[t * 4 + k] = (float)w[k];
          if (w[k] != 0.0) rev[id].push_back({t, (float)w[k]});
        }
      }
      std::vector<int> kstart(frames_ + 1, 0), kt; std::vector<float> kw;
      for (int k = 0; k < frames_; ++k) { kstart[k] = (int)kt.size(); for (auto& e : rev[k]) { kt.push_back(e.first); kw.push_back(e.second); } }
      kstart[frames_] = (int)kt.size();
      InterpTables T;
      auto up = [](const void* h, size_t bytes, void** d) { return hipMalloc(d, bytes ? bytes : 4) == hipSuccess && hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess; };
      if (!up(tidx.data(), tidx.size() * 4, (void**)&T.tidx) || !up(tw.data(), tw.size() * 4, (void**)&T.tw) ||
          !up(kstart.data(), kstart.size() * 4, (void**)&T.kstart) || !up(kt.data(), kt.size() * 4, (void**)&T.kt) ||
          !up(kw.data(), kw.size() * 4, (void**)&T.kw)) { dmx_set_error("htsat: interpolation tables: out of device memory"); return DMX_ERR_PARAM; }
      it = tables.emplace(frames_, T).first;
    }
    *out = &it->second;
    return DMX_OK;
  }

  EmbedParams embed_params(const float* mel, const InterpTables& T) const {
    EmbedParams p;
    p.mel = mel; p.tidx = T.tidx; p.tw = T.tw; p.bn_a = bn_a; p.bn_b = bn_c;
    p.Wp = ps.dev(pe_w); p.bp = ps.dev(pe_b); p.ln_g = ps.dev(pe_g); p.ln_b = ps.dev(pe_bb);
    p.B = B; p.frames = frames; p.E = cfg.embed_dim; p.bins = cfg.num_mel_bins; p.grid = grid0; p.eps = cfg.ln_eps;
    return p;
  }

  int check_shape(int B_, int frames_) const {
    const int Tout = cfg.spec_size * (cfg.spec_size / cfg.num_mel_bins);
    if (B_ < 1 || frames_ < 2 || frames_ > Tout) { dmx_set_error("htsat: 2 <= frames <= %d (the wav size must not exceed the Swin input size)", Tout); return DMX_ERR_SHAPE; }
    return DMX_OK;
  }

  // mel (B, frames, 64) fp32 -> feat (B, T = 64, C = 768) fp32 token features (the tower's final LayerNorm output, token order of the last grid)
  int forward(const float* mel, int B_, int frames_, float* feat, bool keep, void* ws, size_t wsb, hipStream_t st) {
    dry = (ws == nullptr);
    CTRY(check_shape(B_, frames_));
    arena.reset(ws, dry ? (size_t)-1 : wsb);
    Ctx cx{&arena, st, dry, nullptr};
    Arena& A = arena;
    B = B_; frames = frames_; have_tape = false; t_mel = mel;
    InterpTables* T = nullptr;
    if (!dry) CTRY(get_tables(frames, &T));
    const int E = cfg.embed_dim;
    long long N = (long long)grid0 * grid0;
    act_t* cur = A.bf((size_t)B * N * E);
    if (!dry) {
      const EmbedParams p = embed_params(mel, *T);
      const unsigned nb = (unsigned)(((long long)B * N + 127) / 128);
      hipLaunchKernelGGL(embed_fwd_kernel, dim3(nb), dim3(128), 0, st, p, cur);
    }
    tape.assign(stages.size(), {});
    t_stage_out.assign(stages.size(), nullptr);
    for (size_t s = 0; s < stages.size(); ++s) {
      const HtsatStage& S = stages[s];
      const int C = S.C;
      N = (long long)S.H * S.W;
      const long long rows = (long long)B * N;
      tape[s].resize(S.blocks.size());
      for (size_t j = 0; j < S.blocks.size(); ++j) {
        const HtsatBlock& b = S.blocks[j];
        HtsatBlockTape& t = tape[s][j];
        t.x_in = cur;
        t.qkv = A.bf((size_t)rows * 3 * C);
        t.x_mid = A.bf((size_t)rows * C);
        t.u = A.bf((size_t)rows * 4 * C);
        act_t* x_out = A.bf((size_t)rows * C);
        const size_t mk = A.mark();
        act_t* y = A.bf((size_t)rows * C);
        act_t* ao = A.bf((size_t)rows * C);
        act_t* gl = A.bf((size_t)rows * 4 * C);
        CRUN(ln_rows_fwd<act_t>(RowSrc{cur, C, 0, S.H, S.W}, ps.dev(b.ln1_g), ps.dev(b.ln1_b), y, rows, cfg.ln_eps, st));
        Epi e0;
        CRUN(linear_fwd(b.qkv, y, C, t.qkv, 3 * C, rows, e0, st));
        if (!dry) {
          const WinGeom g{B, S.H, S.W, C, b.heads, b.shift};
          hipLaunchKernelGGL(win_attn_fwd_kernel, dim3((unsigned)(B * (S.H / WS) * (S.W / WS)), (unsigned)b.heads), dim3(64), 0, st, t.qkv, ao, ps.dev(b.rpb), g);
        }
        Epi er; er.flags = EPI_RESID; er.R = cur;
        CRUN(linear_fwd(b.proj, ao, C, t.x_mid, C, rows, er, st));
        CRUN(ln_rows_fwd<act_t>(RowSrc{t.x_mid, C, 0, S.H, S.W}, ps.dev(b.ln2_g), ps.dev(b.ln2_b), y, rows, cfg.ln_eps, st));
        CRUN(linear_fwd(b.fc1, y, C, t.u, 4 * C, rows, e0, st));
        if (!dry) {
          const long long n8 = rows * 4 * C / 8;
          hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, t.u, gl, n8);
        }
        Epi er2; er2.flags = EPI_RESID; er2.R = t.x_mid;
        CRUN(linear_fwd(b.fc2, gl, 4 * C, x_out, C, rows, er2, st));
        A.release(mk);
        cur = x_out;
      }
      t_stage_out[s] = cur;
      if (S.merge) {
        const long long mrows = rows / 4;
        act_t* nxt = A.bf((size_t)mrows * 2 * C);
        const size_t mk = A.mark();
        act_t* ym = A.bf((size_t)mrows * 4 * C);
        CRUN(ln_rows_fwd<act_t>(RowSrc{cur, C, 1, S.H, S.W}, ps.dev(S.mg), ps.dev(S.mb), ym, mrows, 1e-5f, st));
        Epi e0;
        CRUN(linear_fwd(S.reduction, ym, 4 * C, nxt, 2 * C, mrows, e0, st));
        A.release(mk);
        cur = nxt;
      }
    }
    t_final_in = cur;
    CRUN(ln_rows_fwd<float>(RowSrc{cur, Cfinal, 0, 1, Tfinal}, ps.dev(fn_g), ps.dev(fn_b), feat, (long long)B * Tfinal, cfg.ln_eps, st));
    if (!dry && hipGetLastError() != hipSuccess) return DMX_ERR_LAUNCH;
    CHECK_WS("htsat");
    have_tape = keep && !dry;
    return DMX_OK;
  }

  // dfeat (B, T, C) fp32 -> dmel (B, frames, 64) fp32, times scale[b] when `scale` is given (the caller's unscaling of a normalised gradient)
  int backward(const float* dfeat, const float* scale, float* dmel, hipStream_t st) {
    if (!have_tape) { dmx_set_error("htsat backward without a forward that kept its state"); return DMX_ERR_STATE; }
    dry = false;
    Ctx cx{&arena, st, false, nullptr};
    Arena& A = arena;
    const size_t mk_all = A.mark();
    InterpTables* T = nullptr;
    CTRY(get_tables(frames, &T));
    act_t* d = A.bf((size_t)B * Tfinal * Cfinal);
    CRUN(ln_rows_bwd<float>(RowSrc{t_final_in, Cfinal, 0, 1, Tfinal}, ps.dev(fn_g), dfeat, nullptr, d, (long long)B * Tfinal, cfg.ln_eps, st));
    for (int s = (int)stages.size() - 1; s >= 0; --s) {
      const HtsatStage& S = stages[s];
      const int C = S.C;
      const long long rows = (long long)B * S.H * S.W;
      if (S.merge) {
        const long long mrows = rows / 4;
        act_t* dym = A.bf((size_t)mrows * 4 * C);
        act_t* dx = A.bf((size_t)rows * C);
        Epi e0;
        CRUN(linear_bwd(S.reduction, d, 2 * C, dym, 4 * C, mrows, e0, st));
        CRUN(ln_rows_bwd<act_t>(RowSrc{t_stage_out[s], C, 1, S.H, S.W}, ps.dev(S.mg), dym, nullptr, dx, mrows, 1e-5f, st));
        d = dx;
      }
      for (int j = (int)S.blocks.size() - 1; j >= 0; --j) {
        const HtsatBlock& b = S.blocks[j];
        const HtsatBlockTape& t = tape[s][j];
        act_t* dmid = A.bf((size_t)rows * C);
        act_t* din = A.bf((size_t)rows * C);
        const size_t mk = A.mark();
        act_t* dg = A.bf((size_t)rows * 4 * C);
        act_t* dy = A.bf((size_t)rows * C);
        act_t* dqkv = A.bf((size_t)rows * 3 * C);
        Epi e0;
        CRUN(linear_bwd(b.fc2, d, C, dg, 4 * C, rows, e0, st));
        {
          const long long n8 = rows * 4 * C / 8;
          hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, t.u, dg, dg, n8);
        }
        CRUN(linear_bwd(b.fc1, dg, 4 * C, dy, C, rows, e0, st));
        CRUN(ln_rows_bwd<act_t>(RowSrc{t.x_mid, C, 0, S.H, S.W}, ps.dev(b.ln2_g), dy, d, dmid, rows, cfg.ln_eps, st));
        CRUN(linear_bwd(b.proj, dmid, C, dy, C, rows, e0, st));                 // d attention output
        {
          const WinGeom g{B, S.H, S.W, C, b.heads, b.shift};
          hipLaunchKernelGGL(win_attn_bwd_kernel, dim3((unsigned)(B * (S.H / WS) * (S.W / WS)), (unsigned)b.heads), dim3(64), 0, st, t.qkv, dy, dqkv, ps.dev(b.rpb), g);
        }
        CRUN(linear_bwd(b.qkv, dqkv, 3 * C, dy, C, rows, e0, st));
        CRUN(ln_rows_bwd<act_t>(RowSrc{t.x_in, C, 0, S.H, S.W}, ps.dev(b.ln1_g), dy, dmid, din, rows, cfg.ln_eps, st));
        A.release(mk);
        d = din;
      }
    }
    // input stage
    const int Tout = cfg.spec_size * (cfg.spec_size / cfg.num_mel_bins);
    float* dimg = A.f32((size_t)B * Tout * cfg.num_mel_bins);
    {
      const EmbedParams p = embed_params(t_mel, *T);
      const long long ntok = (long long)B * grid0 * grid0;
      const unsigned nb = (unsigned)((ntok + 127) / 128);
      hipLaunchKernelGGL(embed_bwd_kernel, dim3(nb), dim3(128), 0, st, p, d, dimg);
      const long long n = (long long)B * frames * cfg.num_mel_bins;
      hipLaunchKernelGGL(interp_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dimg, T->kstart, T->kt, T->kw, scale, dmel, B, frames,
                         cfg.num_mel_bins, Tout);
    }
    const bool over = A.overflow;
    A.release(mk_all);
    if (over) { dmx_set_error("htsat backward: workspace too small (size it with dmx_htsat_workspace_bytes)"); return DMX_ERR_WORKSPACE; }
    return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
  }

  // test hook: a tape tensor of the last forward (which: 0 = block input, 1 = q|k|v, 2 = hidden state after attention, 3 = MLP pre-activation;
  // block == depth of the stage: the stage's output before patch merging).  Returns the element count (16-bit elements), 0 if unknown.
  size_t tape_tensor(int s, int j, int which, const act_t** p) const {
    if (!have_tape || s < 0 || s >= (int)stages.size()) return 0;
    const HtsatStage& S = stages[s];
    const size_t rows = (size_t)B * S.H * S.W;
    if (j == (int)S.blocks.size()) { *p = t_stage_out[s]; return rows * S.C; }
    if (j < 0 || j > (int)S.blocks.size()) return 0;
    const HtsatBlockTape& t = tape[s][j];
    switch (which) {
      case 0: *p = t.x_in; return rows * S.C;
      case 1: *p = t.qkv; return rows * 3 * S.C;
      case 2: *p = t.x_mid; return rows * S.C;
      case 3: *p = t.u; return rows * 4 * S.C;
      default: return 0;
    }
  }

  // backward scratch on top of the forward's footprint (the forward's dry run only sees the forward)
  size_t bwd_extra_bytes(int B_) const {
    size_t worst = 0, chain = 0;
    chain += align_up((size_t)B_ * Tfinal * Cfinal * 2, 256);
    for (int s = (int)stages.size() - 1; s >= 0; --s) {
      const HtsatStage& S = stages[s];
      const size_t rows = (size_t)B_ * S.H * S.W, C = S.C;
      if (S.merge) chain += align_up(rows / 4 * 4 * C * 2, 256) + align_up(rows * C * 2, 256);
      for (size_t j = 0; j < S.blocks.size(); ++j) {
        chain += 2 * align_up(rows * C * 2, 256);
        const size_t tmp = align_up(rows * 4 * C * 2, 256) + align_up(rows * C * 2, 256) + align_up(rows * 3 * C * 2, 256);
        if (chain + tmp > worst) worst = chain + tmp;
      }
    }
    const int Tout = cfg.spec_size * (cfg.spec_size / cfg.num_mel_bins);
    chain += align_up((size_t)B_ * Tout * cfg.num_mel_bins * 4, 256);
    return (chain > worst ? chain : worst) + 4096;
  }
};

Model* dmx_make_htsat(const dmx_htsat_config* c) {
  if (c->num_stages < 1 || c->num_stages > 4 || c->window_size != WS || c->patch_size != 4 || (c->embed_dim % 8 || c->embed_dim < 8 || c->embed_dim > EMB_MAX) ||
      c->num_mel_bins % 4 || c->spec_size % c->num_mel_bins || c->spec_size % (4 * WS)) {
    dmx_set_error("htsat: unsupported configuration (window 8, patch 4, embed width a multiple of 8 up to 128, <= 4 stages)");
    return nullptr;
  }
  int C = c->embed_dim, side = c->spec_size / 4;
  for (int s = 0; s < c->num_stages; ++s) {
    if (C % c->num_heads[s] || C / c->num_heads[s] != HD || side % WS || c->depths[s] < 1) {
      dmx_set_error("htsat: stage %d: head dim must be %d and the token grid a multiple of the window", s, HD);
      return nullptr;
    }
    if (s + 1 < c->num_stages) { C *= 2; side /= 2; }
  }
  return new Htsat(*c);
}
size_t dmx_htsat_ws_impl(Model* m, int B, int frames) {
  Htsat* h = static_cast<Htsat*>(m);
  if (h->forward(nullptr, B, frames, nullptr, false, nullptr, 0, nullptr) != DMX_OK) return 0;
  return h->arena.peak + h->bwd_extra_bytes(B);
}
int dmx_htsat_fwd_impl(Model* m, const float* mel, int B, int frames, float* feat, int keep, void* ws, size_t wsb, hipStream_t st) {
  return static_cast<Htsat*>(m)->forward(mel, B, frames, feat, keep != 0, ws, wsb, st);
}
int dmx_htsat_bwd_impl(Model* m, const float* dfeat, const float* scale, float* dmel, hipStream_t st) {
  return static_cast<Htsat*>(m)->backward(dfeat, scale, dmel, st);
}
size_t dmx_htsat_tape_impl(Model* m, int stage, int block, int which, const act_t** p) {
  return static_cast<Htsat*>(m)->tape_tensor(stage, block, which, p);
}
void dmx_htsat_dims_impl(Model* m, int* tokens, int* channels) {
  Htsat* h = static_cast<Htsat*>(m);
  *tokens = h->Tfinal; *channels = h->Cfinal;
}

int dmx_gram_fwd_impl(const float* F, float* G, int B, int T, int C, hipStream_t st) {
  if (T < 1 || T > 96 || C < 1) return DMX_ERR_SHAPE;
  const size_t smem = (size_t)T * 128 * sizeof(float);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 128 * 4); attr = true; }
  hipLaunchKernelGGL(gram_fwd_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B), dim3(256), smem, st, F, G, T, C);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}
int dmx_gram_bwd_impl(const float* F, const float* dG, float* dF, int B, int T, int C, hipStream_t st) {
  if (T < 1 || C < 1) return DMX_ERR_SHAPE;
  hipLaunchKernelGGL(gram_bwd_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((T + 63) / 64), (unsigned)B), dim3(256), 0, st, F, dG, dF, T, C);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}
