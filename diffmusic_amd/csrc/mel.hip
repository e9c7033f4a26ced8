// STFT / mel measurement path in fp32 (reference: torchaudio MelSpectrogram + AmplitudeToDB built in
// diffmusic/inverse_problem/operator.py:23-33, MelScale :143-147, torch.stft :162-170; loss and
// autograd through them in scheduling_dps.py:202-212).
//
// The DFT is a dense fp32 contraction on the matrix cores (v_mfma_f32_32x32x2_f32): frames are gathered
// straight from the (reflect-padded) waveform into LDS tiles -- the framed signal, the complex spectrum
// workspace of torch.stft and the power spectrogram are never materialised as separate HBM round trips
// beyond one fp32 spectrum buffer that the backward pass re-uses.  The window is folded into the
// twiddle table, rows interleave (re, im) so |X|^2 is a lane-local operation downstream.
#include "dmx_common.h"
#include "kernels.h"
#include <math.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int TB = 64;      // tile rows/cols
constexpr int TK = 32;      // k-step
constexpr int LDT = 36;     // LDS row stride in floats (144 B: conflict-free 16-B reads)

// reflect-padded sample of clip `w` (length L) at padded index p (pad = n_fft/2 each side)
__device__ __forceinline__ float wav_reflect(const float* __restrict__ w, int L, int p, int pad) {
  int s = p - pad;
  if (s < 0) s = -s;
  if (s >= L) s = 2 * (L - 1) - s;
  return w[s];
}

// C[m][n] = sum_k A[m][k] * W[n][k]   (fp32 MFMA).  GATHER: A[m][k] = wavpad[b][f*hop + k], m = b*T + f.
template <bool GATHER>
__global__ __launch_bounds__(256) void f32_gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                          float* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                          int ldc, int T, int L, int hop, int pad) {
  __shared__ __attribute__((aligned(16))) float sA[TB * LDT];
  __shared__ __attribute__((aligned(16))) float sW[TB * LDT];
  const int tid = threadIdx.x;
  const int tiles_n = (N + TB - 1) / TB;
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int lrow = tid >> 2, seg = (tid & 3) * 8;       // each thread stages 8 consecutive k of one row
  const int m = tm * TB + lrow, n = tn * TB + lrow;
  const float* arow = nullptr;
  int fbase = 0;
  bool a_ok = m < M, a_fast = false;
  if (a_ok) {
    if (GATHER) {
      const int b = m / T, f = m - b * T;
      arow = A + (long long)b * lda;                     // lda = waveform row stride
      fbase = f * hop;                                   // padded index of k=0
      a_fast = (fbase - pad >= 0) && (fbase - pad + K <= L) && ((((long long)b * lda + fbase - pad) & 3) == 0);
    } else {
      arow = A + (long long)m * lda;
    }
  }
  const bool w_ok = n < N;
  const float* wrow = W + (long long)(w_ok ? n : 0) * ldw;
  float4 ra0, ra1, rw0, rw1;
  auto load = [&](int k0) {
    const int k = k0 + seg;
    ra0 = ra1 = rw0 = rw1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a_ok && k < K) {
      if (!GATHER) {
        ra0 = *reinterpret_cast<const float4*>(arow + k);
        ra1 = *reinterpret_cast<const float4*>(arow + k + 4);
      } else if (a_fast) {
        const float* p = arow + (fbase - pad + k);
        ra0 = *reinterpret_cast<const float4*>(p);
        ra1 = *reinterpret_cast<const float4*>(p + 4);
      } else {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = wav_reflect(arow, L, fbase + k + i, pad);
        ra0 = make_float4(t[0], t[1], t[2], t[3]);
        ra1 = make_float4(t[4], t[5], t[6], t[7]);
      }
    }
    if (w_ok && k < K) {
      rw0 = *reinterpret_cast<const float4*>(wrow + k);
      rw1 = *reinterpret_cast<const float4*>(wrow + k + 4);
    }
  };
  auto store = [&]() {
    *reinterpret_cast<float4*>(&sA[lrow * LDT + seg]) = ra0;
    *reinterpret_cast<float4*>(&sA[lrow * LDT + seg + 4]) = ra1;
    *reinterpret_cast<float4*>(&sW[lrow * LDT + seg]) = rw0;
    *reinterpret_cast<float4*>(&sW[lrow * LDT + seg + 4]) = rw1;
  };
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;               // wave tile: W rows wr*32.., A rows (output cols) wc*32..
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nk = (K + TK - 1) / TK;
  load(0);
  for (int ks = 0; ks < nk; ++ks) {
    __syncthreads();
    store();
    __syncthreads();
    if (ks + 1 < nk) load((ks + 1) * TK);
    const float* pw = &sW[(wr * 32 + l31) * LDT + lh * 16];
    const float* pa = &sA[(wc * 32 + l31) * LDT + lh * 16];
    float wv[16], av[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 x = *reinterpret_cast<const float4*>(pw + q * 4);
      const float4 y = *reinterpret_cast<const float4*>(pa + q * 4);
      wv[q * 4] = x.x; wv[q * 4 + 1] = x.y; wv[q * 4 + 2] = x.z; wv[q * 4 + 3] = x.w;
      av[q * 4] = y.x; av[q * 4 + 1] = y.y; av[q * 4 + 2] = y.z; av[q * 4 + 3] = y.w;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[kk], av[kk], acc, 0, 0, 0);
  }
  // lane holds output column (A row) mo = .. + l31 and 16 W-rows: (r&3) + 8*(r>>2) + 4*lh
  const int mo = tm * TB + wc * 32 + l31;
  if (mo < M) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int no = tn * TB + wr * 32 + 8 * g + 4 * lh;
      if (no < N) *reinterpret_cast<float4*>(C + (long long)mo * ldc + no) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
  }
}

// table[2k][n] = w[n] cos(2 pi k n / N), table[2k+1][n] = -w[n] sin(..), zero rows above 2*bins; tableT its transpose
__global__ void stft_tables_kernel(float* __restrict__ table, float* __restrict__ tableT, int n_fft, int bins, int Npad, int Kpad,
                                   int hann) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Npad * n_fft) return;
  const int n = (int)(idx % n_fft), r = (int)(idx / n_fft);
  float v = 0.f;
  if (r < 2 * bins) {
    const int k = r >> 1;
    const int ph = (int)(((long long)k * n) % n_fft);                 // exact phase reduction
    const double ang = 6.283185307179586476925286766559 * (double)ph / (double)n_fft;
    const double w = hann ? 0.5 - 0.5 * cos(6.283185307179586476925286766559 * (double)n / (double)n_fft) : 1.0;
    v = (float)((r & 1) ? -w * sin(ang) : w * cos(ang));
  }
  table[idx] = v;
  if (r < Kpad) tableT[(long long)n * Kpad + r] = v;
}

// X (B*T, ldx) interleaved re/im -> mel_lin (B*T, n_mels) = fb^T * (|X|^2 or |X|), mel_out = dB/clamp
constexpr int FR = 16;
__global__ __launch_bounds__(256) void mel_fwd_kernel(const float* __restrict__ X, const float* __restrict__ fb,
                                                      float* __restrict__ mel_lin, float* __restrict__ mel_out, int rows, int ldx,
                                                      int bins, int n_mels, int power2, int to_db, float lo, float hi) {
  extern __shared__ float sP[];   // [FR][bins]
  const int r0 = blockIdx.x * FR;
  for (int i = threadIdx.x; i < FR * bins; i += blockDim.x) {
    const int f = i / bins, k = i - f * bins;
    float p = 0.f;
    if (r0 + f < rows) {
      const float2 x = *reinterpret_cast<const float2*>(X + (long long)(r0 + f) * ldx + 2 * k);
      p = x.x * x.x + x.y * x.y;
      if (!power2) p = sqrtf(p);
    }
    sP[i] = p;
  }
  __syncthreads();
  const int mcol = threadIdx.x % n_mels, fg = threadIdx.x / n_mels;     // n_mels = 64 -> 4 frame groups
  constexpr int per = FR / 4;                                          // 256 threads / 64 mels = 4 frame groups
  float acc[per];
#pragma unroll
  for (int j = 0; j < per; ++j) acc[j] = 0.f;
  for (int k = 0; k < bins; ++k) {
    const float w = fb[(long long)k * n_mels + mcol];
#pragma unroll
    for (int j = 0; j < per; ++j) acc[j] += w * sP[(fg * per + j) * bins + k];
  }
#pragma unroll
  for (int j = 0; j < per; ++j) {
    const int r = r0 + fg * per + j;
    if (r >= rows) continue;
    const float v = acc[j];
    mel_lin[(long long)r * n_mels + mcol] = v;
    float o = to_db ? 10.f * log10f(fmaxf(v, 1e-10f)) : v;
    o = fminf(fmaxf(o, lo), hi);
    mel_out[(long long)r * n_mels + mcol] = o;
  }
}

// dmel (rows, n_mels) wrt mel_out -> Y (rows, ldy) with Y[2k] = c*dP*re, Y[2k+1] = c*dP*im
__global__ __launch_bounds__(256) void mel_bwd_kernel(const float* __restrict__ X, const float* __restrict__ fb,
                                                      const float* __restrict__ mel_lin, const float* __restrict__ dmel,
                                                      float* __restrict__ Y, int rows, int ldx, int ldy, int bins, int n_mels,
                                                      int power2, int to_db, float lo, float hi) {
  extern __shared__ float sD[];   // [FR][n_mels]
  const int r0 = blockIdx.x * FR;
  for (int i = threadIdx.x; i < FR * n_mels; i += blockDim.x) {
    const int f = i / n_mels, mcol = i - f * n_mels;
    float d = 0.f;
    if (r0 + f < rows) {
      const float v = mel_lin[(long long)(r0 + f) * n_mels + mcol];
      d = dmel[(long long)(r0 + f) * n_mels + mcol];
      float o = to_db ? 10.f * log10f(fmaxf(v, 1e-10f)) : v;
      if (o < lo || o > hi) d = 0.f;                                   // clamp(min,max) passes gradient inside only
      if (to_db) d = (v > 1e-10f) ? d * (4.342944819032518f / v) : 0.f; // d/dv 10 log10(max(v,1e-10))
    }
    sD[i] = d;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ldy / 2; k += blockDim.x) {
    float dp[FR];
#pragma unroll
    for (int j = 0; j < FR; ++j) dp[j] = 0.f;
    if (k < bins) {
      for (int mcol = 0; mcol < n_mels; ++mcol) {
        const float w = fb[(long long)k * n_mels + mcol];
        if (w != 0.f) {
#pragma unroll
          for (int j = 0; j < FR; ++j) dp[j] += w * sD[j * n_mels + mcol];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < FR; ++j) {
      if (r0 + j >= rows) continue;
      float2 y = make_float2(0.f, 0.f);
      if (k < bins) {
        const float2 x = *reinterpret_cast<const float2*>(X + (long long)(r0 + j) * ldx + 2 * k);
        if (power2) {
          y = make_float2(2.f * dp[j] * x.x, 2.f * dp[j] * x.y);
        } else {
          const float mag = sqrtf(x.x * x.x + x.y * x.y);
          const float s = mag > 0.f ? dp[j] / mag : 0.f;
          y = make_float2(s * x.x, s * x.y);
        }
      }
      *reinterpret_cast<float2*>(Y + (long long)(r0 + j) * ldy + 2 * k) = y;
    }
  }
}

// |X| (B, bins, T) layout of torch.stft(...).abs() from X (B*T, ldx)
__global__ void stft_mag_kernel(const float* __restrict__ X, float* __restrict__ mag, int B, int T, int bins, int ldx) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * bins * T) return;
  const int t = (int)(idx % T);
  const int k = (int)((idx / T) % bins);
  const int b = (int)(idx / ((long long)T * bins));
  const float2 x = *reinterpret_cast<const float2*>(X + ((long long)b * T + t) * ldx + 2 * k);
  mag[idx] = sqrtf(x.x * x.x + x.y * x.y);
}

// d|X| (B, bins, T) -> Y (B*T, ldy) with Y[2k] = g * re / |X|, Y[2k+1] = g * im / |X| (0 where |X| = 0, the subgradient
// torch.abs uses); columns past 2*bins are zeroed so that the transposed STFT can consume the full row
__global__ void stft_mag_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dmag, float* __restrict__ Y, int B, int T,
                                    int bins, int ldx, int ldy) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int half = ldy >> 1;
  if (idx >= (long long)B * T * half) return;
  const int k = (int)(idx % half);
  const long long r = idx / half;
  float2 y = make_float2(0.f, 0.f);
  if (k < bins) {
    const int t = (int)(r % T), b = (int)(r / T);
    const float2 x = *reinterpret_cast<const float2*>(X + r * ldx + 2 * k);
    const float m = sqrtf(x.x * x.x + x.y * x.y);
    const float c = m > 0.f ? dmag[((long long)b * bins + k) * T + t] / m : 0.f;
    y = make_float2(c * x.x, c * x.y);
  }
  *reinterpret_cast<float2*>(Y + r * ldy + 2 * k) = y;
}

// overlap-add of dframe (B*T, n_fft) back onto the waveform, folding the reflect padding
__global__ void overlap_add_kernel(const float* __restrict__ dframe, float* __restrict__ dwav, long long out_stride, int B, int T, int L,
                                   int n_fft, int hop, int pad, int accumulate) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * L) return;
  const int t = (int)(idx % L), b = (int)(idx / L);
  int cand[3];
  int nc = 0;
  cand[nc++] = t + pad;
  if (t >= 1 && t <= pad) cand[nc++] = pad - t;
  { const int p = 2 * (L - 1) + pad - t; if (p >= L + pad && p < L + 2 * pad) cand[nc++] = p; }
  float acc = 0.f;
  const float* df = dframe + (long long)b * T * n_fft;
  for (int c = 0; c < nc; ++c) {
    const int p = cand[c];
    int f0 = (p - (n_fft - 1) + hop - 1) / hop;
    if (p - (n_fft - 1) <= 0) f0 = 0;
    int f1 = p / hop;
    if (f1 > T - 1) f1 = T - 1;
    for (int f = f0; f <= f1; ++f) acc += df[(long long)f * n_fft + (p - f * hop)];
  }
  float* o = dwav + (long long)b * out_stride + t;
  if (accumulate) *o += acc; else *o = acc;
}

// per-clip L2 of (ref - pred) and its gradient wrt pred: one block per clip
__global__ __launch_bounds__(1024) void l2_loss_grad_kernel(const float* __restrict__ ref, const float* __restrict__ pred,
                                                            float* __restrict__ loss, float* __restrict__ dpred, long long n,
                                                            long long ref_stride, float gscale) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  const float* r = ref + (long long)b * ref_stride;
  const float* p = pred + (long long)b * n;
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) { const float d = r[i] - p[i]; s += d * d; }
  s = block_sum(s, sh);
  const float nrm = sqrtf(s);
  if (threadIdx.x == 0) loss[b] = nrm;
  if (dpred) {
    const float inv = nrm > 0.f ? gscale / nrm : 0.f;
    float* g = dpred + (long long)b * n;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) g[i] = -(r[i] - p[i]) * inv;
  }
}

// long rows (the 768 x 768 Gram matrices of the style operator: 590 k elements per clip -- one workgroup per clip took 440 us): per-chunk
// partial sums parked in the first slots of the clip's dpred row, a finalize launch that adds them in a fixed order, and a gradient launch
// that overwrites the row (stream order: the partial sums are consumed before the gradient lands on them); no atomics
constexpr int L2_CHUNK = 8192;
__global__ __launch_bounds__(256) void l2_partial_kernel(const float* __restrict__ ref, const float* __restrict__ pred, float* __restrict__ dpred,
                                                         long long n, long long ref_stride) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const long long i0 = (long long)blockIdx.x * L2_CHUNK, i1 = i0 + L2_CHUNK < n ? i0 + L2_CHUNK : n;
  const float* r = ref + (long long)b * ref_stride;
  const float* p = pred + (long long)b * n;
  float s = 0.f;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) { const float d = r[i] - p[i]; s = __builtin_fmaf(d, d, s); }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) dpred[(long long)b * n + blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void l2_finalize_kernel(const float* __restrict__ dpred, float* __restrict__ loss, long long n, int nchunk) {
  const int b = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < nchunk; c += 64) s += dpred[(long long)b * n + c];
  s = wave_sum(s);
  if (threadIdx.x == 0) loss[b] = sqrtf(s);
}
__global__ __launch_bounds__(256) void l2_grad_kernel(const float* __restrict__ ref, const float* __restrict__ pred, const float* __restrict__ loss,
                                                      float* __restrict__ dpred, long long n, long long ref_stride, float gscale) {
  const int b = blockIdx.y;
  const long long i0 = (long long)blockIdx.x * L2_CHUNK, i1 = i0 + L2_CHUNK < n ? i0 + L2_CHUNK : n;
  const float* r = ref + (long long)b * ref_stride;
  const float* p = pred + (long long)b * n;
  const float nrm = loss[b], inv = nrm > 0.f ? gscale / nrm : 0.f;
  float* g = dpred + (long long)b * n;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) g[i] = -(r[i] - p[i]) * inv;
}

// y[b, t] = x[b, t] * mask[t] for t < L, 0 for L <= t < Ly  (row strides xs / ys); mask may be null (copy)
__global__ void mask_mul_kernel(const float* __restrict__ x, long long xs, const float* __restrict__ mask, float* __restrict__ y,
                                long long ys, int B, int L, int Ly) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Ly) return;
  const int t = (int)(idx % Ly), b = (int)(idx / Ly);
  float v = 0.f;
  if (t < L) { v = x[(long long)b * xs + t]; if (mask) v *= mask[t]; }
  y[(long long)b * ys + t] = v;
}

// mag (B, bins, T) -> mel (B*T, n_mels) = clamp(fb^T mag)   (MelScale on a given magnitude, operator.py:153-154)
__global__ void melscale_kernel(const float* __restrict__ mag, const float* __restrict__ fb, float* __restrict__ mel, int B, int T,
                                int bins, int n_mels, float lo, float hi) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * T * n_mels) return;
  const int mcol = (int)(idx % n_mels);
  const int t = (int)((idx / n_mels) % T);
  const int b = (int)(idx / ((long long)n_mels * T));
  float acc = 0.f;
  for (int k = 0; k < bins; ++k) acc += fb[(long long)k * n_mels + mcol] * mag[((long long)b * bins + k) * T + t];
  mel[idx] = fminf(fmaxf(acc, lo), hi);
}

// per-clip scale so that max|x| == target: x *= s ; inv_scale[b] = 1/s  (keeps fp16 gradients in range)
__global__ __launch_bounds__(1024) void absmax_normalize_kernel(float* __restrict__ x, float* __restrict__ inv_scale, long long n,
                                                                float target) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  float* p = x + (long long)b * n;
  float m = 0.f;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(p[i]));
  m = block_max(m, sh);
  const float s = (m > 0.f && isfinite(m)) ? target / m : 1.f;
  if (threadIdx.x == 0) inv_scale[b] = 1.f / s;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) p[i] *= s;
}

// The same in four short launches for long clips (one 1024-thread workgroup per clip walked 160 000 samples twice in 69 us):
// memset(max) -> per-chunk maxima folded with an integer atomicMax (|x| >= 0: the bit pattern orders like the value, and a maximum does
// not depend on the order of its operands -- still bit-reproducible) -> scale -> max replaced by 1 / scale.
__global__ __launch_bounds__(256) void absmax_chunk_kernel(const float* __restrict__ x, unsigned* __restrict__ maxbits, long long n, long long per) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const float* p = x + (long long)b * n;
  const long long i0 = (long long)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  float m = 0.f;
  for (long long i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
    if (i + 4 <= i1) { const float4 v = *reinterpret_cast<const float4*>(p + i); m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w))); }
    else for (long long k = i; k < i1; ++k) m = fmaxf(m, fabsf(p[k]));
  }
  m = block_max(m, sh);
  if (threadIdx.x == 0) atomicMax(maxbits + b, __float_as_uint(m));
}
__device__ __forceinline__ float absmax_scale_of(float m, float target) { return (m > 0.f && isfinite(m)) ? target / m : 1.f; }
__global__ __launch_bounds__(256) void absmax_scale_kernel(float* __restrict__ x, const float* __restrict__ maxv, long long n, long long per, float target) {
  const int b = blockIdx.y;
  float* p = x + (long long)b * n;
  const float s = absmax_scale_of(maxv[b], target);
  const long long i0 = (long long)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  for (long long i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
    if (i + 4 <= i1) { float4 v = *reinterpret_cast<float4*>(p + i); v.x *= s; v.y *= s; v.z *= s; v.w *= s; *reinterpret_cast<float4*>(p + i) = v; }
    else for (long long k = i; k < i1; ++k) p[k] *= s;
  }
}
__global__ void absmax_finish_kernel(float* __restrict__ inv_scale, int B, float target) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) inv_scale[b] = 1.f / absmax_scale_of(inv_scale[b], target);
}

}  // namespace

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH)

int dmx_stft_tables(float* table, float* tableT, int n_fft, int bins, int Npad, int Kpad, int hann, hipStream_t st) {
  const long long n = (long long)Npad * n_fft;
  hipMemsetAsync(tableT, 0, (size_t)n_fft * Kpad * sizeof(float), st);
  hipLaunchKernelGGL(stft_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, table, tableT, n_fft, bins, Npad, Kpad, hann);
  return CHECK_LAUNCH();
}
// X (B*T, Npad) = frames(wav) . table^T
int dmx_stft_fwd(const float* wav, long long wav_stride, const float* table, float* X, int B, int L, int T, int n_fft, int hop, int Npad,
                 hipStream_t st) {
  const int M = B * T;
  const long long tiles = (long long)cdiv(M, TB) * cdiv(Npad, TB);
  hipLaunchKernelGGL(f32_gemm_nt_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, st, wav, table, X, M, Npad, n_fft, (int)wav_stride, n_fft, Npad,
                     T, L, hop, n_fft / 2);
  return CHECK_LAUNCH();
}
// dframe (B*T, n_fft) = Y (B*T, Kpad) . tableT^T
int dmx_stft_bwd_frames(const float* Y, const float* tableT, float* dframe, int M, int n_fft, int Kpad, hipStream_t st) {
  const long long tiles = (long long)cdiv(M, TB) * cdiv(n_fft, TB);
  hipLaunchKernelGGL(f32_gemm_nt_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, st, Y, tableT, dframe, M, n_fft, Kpad, Kpad, Kpad, n_fft,
                     0, 0, 0, 0);
  return CHECK_LAUNCH();
}
int dmx_mel_fwd(const float* X, const float* fb, float* mel_lin, float* mel_out, int rows, int ldx, int bins, int n_mels, int power2,
                int to_db, float lo, float hi, hipStream_t st) {
  if (n_mels != 64) return DMX_ERR_SHAPE;
  hipLaunchKernelGGL(mel_fwd_kernel, dim3(cdiv(rows, FR)), dim3(256), (size_t)FR * bins * sizeof(float), st, X, fb, mel_lin, mel_out, rows,
                     ldx, bins, n_mels, power2, to_db, lo, hi);
  return CHECK_LAUNCH();
}
int dmx_mel_bwd(const float* X, const float* fb, const float* mel_lin, const float* dmel, float* Y, int rows, int ldx, int ldy, int bins,
                int n_mels, int power2, int to_db, float lo, float hi, hipStream_t st) {
  hipLaunchKernelGGL(mel_bwd_kernel, dim3(cdiv(rows, FR)), dim3(256), (size_t)FR * n_mels * sizeof(float), st, X, fb, mel_lin, dmel, Y, rows,
                     ldx, ldy, bins, n_mels, power2, to_db, lo, hi);
  return CHECK_LAUNCH();
}
int dmx_stft_mag(const float* X, float* mag, int B, int T, int bins, int ldx, hipStream_t st) {
  const long long n = (long long)B * bins * T;
  hipLaunchKernelGGL(stft_mag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, mag, B, T, bins, ldx);
  return CHECK_LAUNCH();
}
int dmx_stft_mag_bwd(const float* X, const float* dmag, float* Y, int B, int T, int bins, int ldx, int ldy, hipStream_t st) {
  const long long n = (long long)B * T * (ldy >> 1);
  hipLaunchKernelGGL(stft_mag_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, dmag, Y, B, T, bins, ldx, ldy);
  return CHECK_LAUNCH();
}
int dmx_overlap_add(const float* dframe, float* dwav, long long out_stride, int B, int T, int L, int n_fft, int hop, int accumulate,
                    hipStream_t st) {
  const long long n = (long long)B * L;
  hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dframe, dwav, out_stride, B, T, L, n_fft, hop,
                     n_fft / 2, accumulate);
  return CHECK_LAUNCH();
}
int dmx_l2_loss_grad(const float* ref, const float* pred, float* loss, float* dpred, int B, long long n, long long ref_stride, float gscale,
                     hipStream_t st) {
  if (dpred && n >= 8 * L2_CHUNK) {
    const int nchunk = (int)((n + L2_CHUNK - 1) / L2_CHUNK);
    hipLaunchKernelGGL(l2_partial_kernel, dim3(nchunk, B), dim3(256), 0, st, ref, pred, dpred, n, ref_stride);
    hipLaunchKernelGGL(l2_finalize_kernel, dim3(B), dim3(64), 0, st, dpred, loss, n, nchunk);
    hipLaunchKernelGGL(l2_grad_kernel, dim3(nchunk, B), dim3(256), 0, st, ref, pred, loss, dpred, n, ref_stride, gscale);
    return CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(l2_loss_grad_kernel, dim3(B), dim3(1024), 0, st, ref, pred, loss, dpred, n, ref_stride, gscale);
  return CHECK_LAUNCH();
}
int dmx_mask_mul(const float* x, long long xs, const float* mask, float* y, long long ys, int B, int L, int Ly, hipStream_t st) {
  const long long n = (long long)B * Ly;
  hipLaunchKernelGGL(mask_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xs, mask, y, ys, B, L, Ly);
  return CHECK_LAUNCH();
}
int dmx_melscale(const float* mag, const float* fb, float* mel, int B, int T, int bins, int n_mels, float lo, float hi, hipStream_t st) {
  const long long n = (long long)B * T * n_mels;
  hipLaunchKernelGGL(melscale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, fb, mel, B, T, bins, n_mels, lo, hi);
  return CHECK_LAUNCH();
}
int dmx_absmax_normalize(float* x, float* inv_scale, int B, long long n, float target, hipStream_t st) {
  if (n < 32768 || (n & 3) || ((uintptr_t)x & 15)) {            // short or unaligned clips: one workgroup per clip
    hipLaunchKernelGGL(absmax_normalize_kernel, dim3(B), dim3(1024), 0, st, x, inv_scale, n, target);
    return CHECK_LAUNCH();
  }
  const int chunks = (int)((n + 8191) / 8192);                  // 8192 samples per workgroup (a multiple of 4: chunks stay 16-byte aligned)
  const long long per = 8192;
  (void)hipMemsetAsync(inv_scale, 0, (size_t)B * sizeof(float), st);
  hipLaunchKernelGGL(absmax_chunk_kernel, dim3(chunks, B), dim3(256), 0, st, x, reinterpret_cast<unsigned*>(inv_scale), n, per);
  hipLaunchKernelGGL(absmax_scale_kernel, dim3(chunks, B), dim3(256), 0, st, x, inv_scale, n, per, target);
  hipLaunchKernelGGL(absmax_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, st, inv_scale, B, target);
  return CHECK_LAUNCH();
}
