// Fused STFT -> |X|^2 -> mel -> dB (-> L2 loss) and its hand-written backward, n_fft = 1024, fp32.
//
// Reference: torchaudio MelSpectrogram(16000, 1024, 160, 1024, n_mels=64, power=2) + AmplitudeToDB built in
// diffmusic/inverse_problem/operator.py:23-33 (and MelScale on |STFT|, :143-147 / :162-170), the loss
// torch.linalg.norm(transform(y) - transform(A(x))) and torch.autograd.grad through all of it in
// diffmusic/schedulers/scheduling_dps.py:202-212.
//
// One wave owns one frame at a time: 1024 windowed samples (reflect padding and the optional measurement mask applied on load,
// 256-byte coalesced reads, the next frame's samples fetched under this frame's arithmetic; neighbouring frames overlap by 86 % and
// are served by L2) go through an in-place radix-4 Stockham FFT in LDS (five passes), |X|^2 stays in LDS, the mel filterbank is applied from a compacted table
// (the <= 2 % non-zero weights of the triangular bank, one mel column per lane, 256-byte reads) and the dB / clamp tail runs in
// registers.  Forward writes the (B, T, 64) mel tensor and per-workgroup partial sums of (ref - mel)^2; nothing else touches HBM
// (no framed signal, no complex spectrum, no power spectrogram: the dense-DFT path moved ~50x the algorithmic bytes).
//
// Backward recomputes the frame's spectrum instead of reading a saved one (two FFTs per frame are cheaper than 8 KB of HBM per
// frame), pushes d(loss)/d(mel) through clamp, dB, filterbank and |.|^2, runs the adjoint of the windowed real DFT as one more
// 1024-point FFT with conjugated twiddles, and overlap-adds in LDS: a workgroup owns a chunk of OUTPUT samples and walks every
// frame that touches it (reflect-padded positions fold back onto their source sample), each wave into its own accumulator in a
// fixed frame order, the four accumulators summed in wave order -- no atomics, bit-reproducible.  Algorithmic traffic: forward
// reads the waveform and writes mel, backward reads waveform + reference mel and writes the waveform gradient = 2.4 MB per
// 10 s clip and guided step.
#include "dmx_common.h"
#include "kernels.h"
#include <cstring>

namespace {

constexpr int NF = 1024;            // n_fft (the only size this file handles; other sizes take the dense-DFT path of mel.hip)
constexpr int NB = NF / 2 + 1;      // one-sided bins
constexpr int NM = 64;              // mel columns = lanes of a wave
constexpr int PADH = NF / 2;        // reflect padding on each side (center=True)

struct SmParams {
  const float* wav; long long wav_stride;
  const float* mask;                 // optional (L): y = wav * mask (the inpainting operator, operator.py:126-128)
  const float* ref; long long ref_stride;      // reference mel (B or 1, T, 64); stride 0 = one reference for every clip
  const float* dmel;                 // backward: explicit d(loss)/d(mel) (B, T, 64) instead of the L2 gradient against ref
  float* mel_out;                    // forward: (B, T, 64) or null
  float* partial;                    // per-clip, per-workgroup sums of (ref - mel)^2: forward writes, backward reads
  int nparts;                        // workgroups per clip of the forward launch that filled `partial`
  float* loss;                       // backward: loss[b] = sqrt(sum of partials) (written by the clip's first workgroup) or null
  float* dwav; long long dwav_stride; int Lfull, accumulate;
  float gscale;
  int B, L, T, hop, power2, to_db;
  float lo, hi;
  const float2* tw;                  // exp(-2 pi i m / 1024), m = 0 .. 1023
  const float* win;                  // analysis window (1024)
  const int* klo; const int* klen; const float* fbc; int kmax;       // forward: mel column m sums bins klo[m] .. klo[m] + klen[m]
  const int* mlo; const int* mlen; const float* fbr; int mmax;       // backward: bin k sums mel columns mlo[k] .. (row pitch NBP)
  int chunk;                         // backward: output samples per workgroup
};
constexpr int NBP = 576;             // padded bin count of the transposed compact bank (9 * 64)

__device__ __forceinline__ int fold_reflect(int s, int L) {
  if (s < 0) s = -s;
  if (s >= L) s = 2 * (L - 1) - s;
  return s;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// radix-4 butterfly, natural output order y_r = sum_q v_q exp(-/+ 2 pi i r q / 4)
template <bool INV>
__device__ __forceinline__ void bfly4(float2& v0, float2& v1, float2& v2, float2& v3) {
  const float2 a = make_float2(v0.x + v2.x, v0.y + v2.y), b = make_float2(v0.x - v2.x, v0.y - v2.y);
  const float2 c = make_float2(v1.x + v3.x, v1.y + v3.y);
  const float2 d0 = make_float2(v1.x - v3.x, v1.y - v3.y);
  const float2 d = INV ? make_float2(-d0.y, d0.x) : make_float2(d0.y, -d0.x);        // * (+i) : * (-i)
  v0 = make_float2(a.x + c.x, a.y + c.y);
  v2 = make_float2(a.x - c.x, a.y - c.y);
  v1 = make_float2(b.x + d.x, b.y + d.y);
  v3 = make_float2(b.x - d.x, b.y - d.y);
}

// One Stockham pass over the wave's 1024 complex points, IN PLACE: every lane first reads the 16 inputs of its four butterflies
// (j = lane + 64 m), then writes their 16 outputs.  LDS instructions of one wave execute in issue order and all 64 lanes issue
// together, so every read of the pass precedes every write of the pass; the fences keep the compiler from mixing the two groups.
template <int NS, bool INV>
__device__ __forceinline__ void fft_pass(float2* buf, const float2* s_tw, int lane) {
  float2 v[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int j = lane + 64 * m;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[m][q] = buf[j + 256 * q];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int j = lane + 64 * m;
    const int k = j & (NS - 1);
    if (NS > 1) {
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        float2 w = s_tw[q * k * (256 / NS)];
        if (INV) w.y = -w.y;
        v[m][q] = cmul(v[m][q], w);
      }
    }
    bfly4<INV>(v[m][0], v[m][1], v[m][2], v[m][3]);
    const int j0 = ((j - k) << 2) + k;
#pragma unroll
    for (int q = 0; q < 4; ++q) buf[j0 + q * NS] = v[m][q];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <bool INV>
__device__ __forceinline__ void fft1024(float2* buf, const float2* s_tw, int lane) {
  fft_pass<1, INV>(buf, s_tw, lane);
  fft_pass<4, INV>(buf, s_tw, lane);
  fft_pass<16, INV>(buf, s_tw, lane);
  fft_pass<64, INV>(buf, s_tw, lane);
  fft_pass<256, INV>(buf, s_tw, lane);
}

// masked, reflect-padded samples n = lane + 64 j of frame f of clip b -> registers (frames past the end: zeros, nothing is read)
__device__ __forceinline__ void fetch_frame(const SmParams& P, int b, int f, float (&x)[16], int lane) {
  const float* w = P.wav + (long long)b * P.wav_stride;
  const int p0 = f * P.hop - PADH;
  const bool live = f < P.T;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int s = fold_reflect(p0 + lane + 64 * j, P.L);
    float v = 0.f;
    if (live) { v = w[s]; if (P.mask) v *= P.mask[s]; }
    x[j] = v;
  }
}
// registers -> LDS with the analysis window (kept in registers: lane l only ever needs win[l + 64 j]); imaginary parts zero
__device__ __forceinline__ void put_frame(const float (&x)[16], const float (&win)[16], float2* buf, int lane) {
#pragma unroll
  for (int j = 0; j < 16; ++j) buf[lane + 64 * j] = make_float2(x[j] * win[j], 0.f);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// |X|^2 (or |X|) of the 513 one-sided bins -> pw[k]; then mel column `lane`: v = sum_k fb[k][lane] * pw[k]
__device__ __forceinline__ float power_and_mel(const SmParams& P, const float2* buf, float* pw, int lane) {
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int k = lane + 64 * j;
    if (k < NB) {
      const float2 x = buf[k];
      float p = x.x * x.x + x.y * x.y;
      if (!P.power2) p = sqrtf(p);
      pw[k] = p;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int k0 = P.klo[lane], kn = P.klen[lane];
  float acc = 0.f;
  for (int i = 0; i < P.kmax; ++i) {                       // wave-uniform trip count; bins in increasing order
    const float wgt = P.fbc[i * NM + lane];
    if (i < kn) acc += wgt * pw[k0 + i];
  }
  return acc;
}

__device__ __forceinline__ float mel_tail(const SmParams& P, float v) {
  float o = P.to_db ? 10.f * log10f(fmaxf(v, 1e-10f)) : v;
  return fminf(fmaxf(o, P.lo), P.hi);
}

constexpr int FWD_FPW = 4;           // frames per wave of the forward kernel (16 per workgroup)

__global__ __launch_bounds__(256) void stft_mel_fwd_kernel(const SmParams P) {
  __shared__ float2 s_tw[NF];
  __shared__ float2 s_buf[4][NF];
  __shared__ float s_pw[4][NBP];
  __shared__ float s_red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  for (int i = tid; i < NF; i += 256) s_tw[i] = P.tw[i];
  float win[16], xs[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) win[j] = P.win[lane + 64 * j];
  __syncthreads();
  float2* buf = s_buf[wave];
  float* pw = s_pw[wave];
  float sq = 0.f;
  const int f0 = blockIdx.x * (4 * FWD_FPW);
  fetch_frame(P, b, f0 + wave, xs, lane);
  for (int i = 0; i < FWD_FPW; ++i) {
    const int f = f0 + wave + 4 * i;                       // the four waves walk neighbouring frames together (shared cache lines)
    if (f >= P.T) break;                                   // wave-uniform
    put_frame(xs, win, buf, lane);
    if (i + 1 < FWD_FPW) fetch_frame(P, b, f + 4, xs, lane);      // in flight under this frame's FFT
    fft1024<false>(buf, s_tw, lane);
    const float v = power_and_mel(P, buf, pw, lane);
    const float o = mel_tail(P, v);
    const long long idx = ((long long)b * P.T + f) * NM + lane;
    if (P.mel_out) P.mel_out[idx] = o;
    if (P.ref) {
      const float d = P.ref[(long long)b * P.ref_stride + (long long)f * NM + lane] - o;
      sq += d * d;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (P.partial) {                                         // deterministic: lanes by butterfly, waves in order
    sq = wave_sum(sq);
    if (lane == 0) s_red[wave] = sq;
    __syncthreads();
    if (tid == 0) P.partial[(long long)b * gridDim.x + blockIdx.x] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
  }
}

// output samples per workgroup (8 hops of 160): 15 frames touch a chunk, ~4 per wave; with 20 KB of accumulators two workgroups share a
// CU (a 2560-sample chunk needs 27 % fewer FFTs per sample but leaves one workgroup per CU alone with its load latencies)
constexpr int BWD_MAX_CHUNK = 1280;

__global__ __launch_bounds__(256) void stft_mel_bwd_kernel(const SmParams P) {
  __shared__ float2 s_tw[NF];
  __shared__ float2 s_buf[4][NF];
  __shared__ float s_pw[4][NBP];
  __shared__ float s_dv[4][NM];
  __shared__ float s_acc[4][BWD_MAX_CHUNK];
  __shared__ float s_inv;
  // ~71.7 KB of static LDS: legal because gfx950 has 160 KiB per CU (other gfx9 parts stop at 64 KiB); two workgroups share a CU
  static_assert(sizeof(float2) * NF * 5 + sizeof(float) * (4 * NBP + 4 * NM + 4 * BWD_MAX_CHUNK + 1) <= 80 * 1024,
                "stft_mel_bwd_kernel: two workgroups per CU must fit the 160 KiB of gfx950 LDS");
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "stft_mel.hip sizes its LDS for gfx950 (160 KiB per CU)"
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int L = P.L, hop = P.hop;
  const int s0 = blockIdx.x * P.chunk, s1 = min(s0 + P.chunk, L);
  for (int i = tid; i < NF; i += 256) s_tw[i] = P.tw[i];
  float win[16], xs[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) win[j] = P.win[lane + 64 * j];
  for (int i = tid; i < 4 * BWD_MAX_CHUNK; i += 256) (&s_acc[0][0])[i] = 0.f;
  if (tid == 0) {
    float inv = 0.f;
    if (!P.dmel) {                                         // || ref - mel ||_2 of this clip from the forward launch's partial sums, fixed order
      float s = 0.f;
      for (int i = 0; i < P.nparts; ++i) s += P.partial[(long long)b * P.nparts + i];
      const float nrm = sqrtf(s);
      inv = nrm > 0.f ? P.gscale / nrm : 0.f;
      if (P.loss && blockIdx.x == 0) P.loss[b] = nrm;
    }
    s_inv = inv;
  }
  __syncthreads();
  const float inv = s_inv;
  // frames that reach this chunk: directly, or through the reflect padding at either end of the clip
  int flo, fhi;
  {
    const int a0 = s0 + PADH - (NF - 1);                   // first padded position whose frame may still cover s0
    flo = a0 <= 0 ? 0 : (a0 + hop - 1) / hop;
    fhi = min(P.T - 1, (s1 - 1 + PADH) / hop);
    if (s0 <= PADH && s1 > 1) flo = 0;                     // left reflections live in padded positions [0, 511]: frames from 0 on
    if (s1 - 1 >= L - 1 - PADH && s0 <= L - 2) fhi = P.T - 1;   // right reflections: up to the last frame
  }
  float2* buf = s_buf[wave];
  float* pw = s_pw[wave];
  float* dv = s_dv[wave];
  float* acc = s_acc[wave];
  fetch_frame(P, b, flo + wave, xs, lane);
  for (int f = flo + wave; f <= fhi; f += 4) {
    put_frame(xs, win, buf, lane);
    if (f + 4 <= fhi) fetch_frame(P, b, f + 4, xs, lane);         // in flight under this frame's two FFTs
    fft1024<false>(buf, s_tw, lane);
    const float v = power_and_mel(P, buf, pw, lane);
    // d(loss)/d(mel_out) -> d/d(mel_lin): clamp passes the gradient inside (lo, hi) only, dB is 10 / ln 10 / v above the 1e-10 floor
    const long long idx = ((long long)b * P.T + f) * NM + lane;
    const float o_raw = P.to_db ? 10.f * log10f(fmaxf(v, 1e-10f)) : v;
    float d;
    if (P.dmel) d = P.dmel[idx];
    else d = -(P.ref[(long long)b * P.ref_stride + (long long)f * NM + lane] - fminf(fmaxf(o_raw, P.lo), P.hi)) * inv;
    if (o_raw < P.lo || o_raw > P.hi) d = 0.f;
    if (P.to_db) d = (v > 1e-10f) ? d * (4.342944819032518f / v) : 0.f;
    dv[lane] = d;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // d/dP[k] = sum_m fb[k][m] dv[m];  G[k] = d/dX[k] = 2 dP X (power) or dP X / |X| (magnitude); bins above 512 are zero
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = lane + 64 * j;
      float2 g = make_float2(0.f, 0.f);
      if (k < NB) {
        const int m0 = P.mlo[k], mn = P.mlen[k];
        float dp = 0.f;
        for (int i = 0; i < mn; ++i) dp += P.fbr[i * NBP + k] * dv[m0 + i];
        const float2 x = buf[k];
        if (P.power2) {
          g = make_float2(2.f * dp * x.x, 2.f * dp * x.y);
        } else {
          const float mag = pw[k];
          const float c = mag > 0.f ? dp / mag : 0.f;
          g = make_float2(c * x.x, c * x.y);
        }
      }
      buf[k] = g;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // adjoint of the windowed real DFT: dframe[n] = win[n] * Re sum_k G[k] exp(+2 pi i k n / 1024)
    fft1024<true>(buf, s_tw, lane);
    // overlap-add into this wave's accumulator.  Three sub-passes with distinct targets each (a frame at the clip edge holds a
    // sample AND its mirror image): interior positions, left reflections, right reflections
    const int p0 = f * hop - PADH;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int n = lane + 64 * j;
        const int sr = p0 + n;                             // unfolded sample index
        int s;
        bool ok;
        if (pass == 0) { s = sr; ok = sr >= 0 && sr < L; }
        else if (pass == 1) { s = -sr; ok = sr < 0; }
        else { s = 2 * (L - 1) - sr; ok = sr >= L; }
        if (ok && s >= s0 && s < s1) acc[s - s0] += buf[n].x * win[j];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (pass == 0 && p0 >= 0 && p0 + NF <= L) break;     // interior frame: no reflections (wave-uniform)
    }
  }
  __syncthreads();
  float* out = P.dwav + (long long)b * P.dwav_stride;
  for (int i = tid; i < s1 - s0; i += 256) {
    float g = ((s_acc[0][i] + s_acc[1][i]) + s_acc[2][i]) + s_acc[3][i];
    if (P.mask) g *= P.mask[s0 + i];
    if (P.accumulate) out[s0 + i] += g; else out[s0 + i] = g;
  }
  if (blockIdx.x == gridDim.x - 1 && !P.accumulate)        // samples past the clip (the vocoder's tail) get no gradient
    for (int i = L + tid; i < P.Lfull; i += 256) out[i] = 0.f;
}

}  // namespace

static void fill_common(SmParams& P, const DmxStftMelTables& t, int B, int L, int hop, int power2, int to_db, float lo, float hi) {
  P.B = B; P.L = L; P.hop = hop; P.T = 1 + L / hop; P.power2 = power2; P.to_db = to_db; P.lo = lo; P.hi = hi;
  P.tw = t.tw; P.win = t.win; P.klo = t.klo; P.klen = t.klen; P.fbc = t.fbc; P.kmax = t.kmax;
  P.mlo = t.mlo; P.mlen = t.mlen; P.fbr = t.fbr; P.mmax = t.mmax;
}

int dmx_stft_mel_parts(int L, int hop) { return cdiv(1 + L / hop, 4 * FWD_FPW); }

int dmx_stft_mel_fwd(const DmxStftMelTables& t, const float* wav, long long wav_stride, const float* mask, const float* ref, long long ref_stride,
                     float* mel_out, float* partial, int B, int L, int hop, int power2, int to_db, float lo, float hi, hipStream_t st) {
  if (L < NF / 2 + 1) return DMX_ERR_SHAPE;
  SmParams P;
  memset(&P, 0, sizeof(P));
  fill_common(P, t, B, L, hop, power2, to_db, lo, hi);
  P.wav = wav; P.wav_stride = wav_stride; P.mask = mask; P.ref = ref; P.ref_stride = ref_stride; P.mel_out = mel_out;
  P.partial = ref ? partial : nullptr;
  hipLaunchKernelGGL(stft_mel_fwd_kernel, dim3(dmx_stft_mel_parts(L, hop), B), dim3(256), 0, st, P);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}

int dmx_stft_mel_bwd(const DmxStftMelTables& t, const float* wav, long long wav_stride, const float* mask, const float* ref, long long ref_stride,
                     const float* dmel, const float* partial, float gscale, float* loss, float* dwav, long long dwav_stride, int Lfull,
                     int accumulate, int B, int L, int hop, int power2, int to_db, float lo, float hi, hipStream_t st) {
  if (L < 2 * NF || (!dmel && (!ref || !partial)) || Lfull < L) return DMX_ERR_SHAPE;   // (left and right reflection zones must not meet)
  SmParams P;
  memset(&P, 0, sizeof(P));
  fill_common(P, t, B, L, hop, power2, to_db, lo, hi);
  P.wav = wav; P.wav_stride = wav_stride; P.mask = mask; P.ref = ref; P.ref_stride = ref_stride; P.dmel = dmel;
  P.partial = const_cast<float*>(partial); P.nparts = dmx_stft_mel_parts(L, hop); P.gscale = gscale; P.loss = loss;
  P.dwav = dwav; P.dwav_stride = dwav_stride; P.Lfull = Lfull; P.accumulate = accumulate;
  int chunk = (BWD_MAX_CHUNK / hop) * hop;                 // whole hops per workgroup
  if (chunk < hop) chunk = BWD_MAX_CHUNK;                  // (hop > 2560: any chunking is correct, frames are found from sample ranges)
  P.chunk = chunk;
  hipLaunchKernelGGL(stft_mel_bwd_kernel, dim3(cdiv(L, chunk), B), dim3(256), 0, st, P);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}
