// FIR operators of the measurement path (fp32): torchaudio sinc-hann polyphase resampling
// (SuperResolutionOperator.forward, diffmusic/inverse_problem/operator.py:203-205) and the 5000-tap
// reverberation conv1d (MusicDereverberationOperator.forward, operator.py:244-250), each with its
// hand-written transpose for the guidance gradient.
//   forward : out[j*nw + p] = sum_t h[p][t] * in[j*og + t - off]        (zero outside [0, Lin))
//   backward: din[i]        = sum_p sum_t h[p][t] * dout[j*nw + p],  j*og + t - off == i
#include "dmx_common.h"
#include "kernels.h"

namespace {

// 1:1 case (og = nw = 1), long filters: LDS-tiled, 4 outputs per thread, 512-tap chunks
constexpr int FT = 256, FO = 4, FCH = 512;
__global__ __launch_bounds__(FT) void fir_dense_kernel(const float* __restrict__ in, long long in_stride, const float* __restrict__ h,
                                                       float* __restrict__ out, long long out_stride, int Lin, int Lout, int taps,
                                                       int off) {
  __shared__ float sh[FCH];
  __shared__ float sx[FT * FO + FCH];
  const int b = blockIdx.y, base = blockIdx.x * FT * FO;
  const float* x = in + (long long)b * in_stride;
  float acc[FO] = {0.f, 0.f, 0.f, 0.f};
  for (int t0 = 0; t0 < taps; t0 += FCH) {
    const int nt = min(FCH, taps - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < FCH; i += FT) sh[i] = i < nt ? h[t0 + i] : 0.f;
    for (int i = threadIdx.x; i < FT * FO + FCH; i += FT) {
      const int s = base + i + t0 - off;
      sx[i] = (s >= 0 && s < Lin) ? x[s] : 0.f;
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      const float hv = sh[t];
#pragma unroll
      for (int r = 0; r < FO; ++r) acc[r] += hv * sx[threadIdx.x + FT * r + t];
    }
  }
#pragma unroll
  for (int r = 0; r < FO; ++r) {
    const int o = base + threadIdx.x + FT * r;
    if (o < Lout) out[(long long)b * out_stride + o] = acc[r];
  }
}

// general polyphase forward (short filters)
__global__ void fir_poly_fwd_kernel(const float* __restrict__ in, long long in_stride, const float* __restrict__ h,
                                    float* __restrict__ out, long long out_stride, int B, int Lin, int Lout, int taps, int og, int nw,
                                    int off) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Lout) return;
  const int o = (int)(idx % Lout), b = (int)(idx / Lout);
  const int j = o / nw, p = o - j * nw;
  const float* x = in + (long long)b * in_stride;
  const float* hp = h + (long long)p * taps;
  float acc = 0.f;
  const int s0 = j * og - off;
  for (int t = 0; t < taps; ++t) {
    const int s = s0 + t;
    if (s >= 0 && s < Lin) acc += hp[t] * x[s];
  }
  out[(long long)b * out_stride + o] = acc;
}
// general polyphase transpose: gather over (j, p) with t = i + off - j*og in [0, taps)
__global__ void fir_poly_bwd_kernel(const float* __restrict__ dout, long long dout_stride, const float* __restrict__ h,
                                    float* __restrict__ din, long long din_stride, int B, int Lin, int Lout, int taps, int og, int nw,
                                    int off) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Lin) return;
  const int i = (int)(idx % Lin), b = (int)(idx / Lin);
  const float* dy = dout + (long long)b * dout_stride;
  float acc = 0.f;
  int jlo = (i + off - (taps - 1) + og - 1) / og;
  if (i + off - (taps - 1) <= 0) jlo = 0;
  const int jhi = (i + off) / og;
  for (int j = jlo; j <= jhi; ++j) {
    const int t = i + off - j * og;
    for (int p = 0; p < nw; ++p) {
      const int o = j * nw + p;
      if (o < Lout) acc += h[(long long)p * taps + t] * dy[o];
    }
  }
  din[(long long)b * din_stride + i] = acc;
}

}  // namespace

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH)

int dmx_fir_fwd_impl(const float* in, long long in_stride, const float* h, float* out, long long out_stride, int B, int Lin, int Lout,
                     int taps, int og, int nw, int off, hipStream_t st) {
  if (og == 1 && nw == 1 && taps >= 128) {
    hipLaunchKernelGGL(fir_dense_kernel, dim3(cdiv(Lout, FT * FO), B), dim3(FT), 0, st, in, in_stride, h, out, out_stride, Lin, Lout, taps, off);
  } else {
    const long long n = (long long)B * Lout;
    hipLaunchKernelGGL(fir_poly_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, in_stride, h, out, out_stride, B, Lin, Lout,
                       taps, og, nw, off);
  }
  return CHECK_LAUNCH();
}
int dmx_fir_bwd_impl(const float* dout, long long dout_stride, const float* h, const float* h_rev, float* din, long long din_stride, int B,
                     int Lin, int Lout, int taps, int og, int nw, int off, hipStream_t st) {
  if (og == 1 && nw == 1 && taps >= 128 && h_rev) {
    // din[i] = sum_t' h_rev[t'] dout[i + t' - (taps-1-off)]
    hipLaunchKernelGGL(fir_dense_kernel, dim3(cdiv(Lin, FT * FO), B), dim3(FT), 0, st, dout, dout_stride, h_rev, din, din_stride, Lout, Lin, taps,
                       taps - 1 - off);
  } else {
    const long long n = (long long)B * Lin;
    hipLaunchKernelGGL(fir_poly_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dout, dout_stride, h, din, din_stride, B, Lin,
                       Lout, taps, og, nw, off);
  }
  return CHECK_LAUNCH();
}

extern "C" {
int dmx_fir_fwd(const float* in, long long in_stride, const float* h, float* out, long long out_stride, int batch, int Lin, int Lout,
                int taps, int orig, int new_, int off, void* stream) {
  return dmx_fir_fwd_impl(in, in_stride, h, out, out_stride, batch, Lin, Lout, taps, orig, new_, off, (hipStream_t)stream);
}
int dmx_fir_bwd(const float* dout, long long dout_stride, const float* h, const float* h_rev, float* din, long long din_stride, int batch,
                int Lin, int Lout, int taps, int orig, int new_, int off, void* stream) {
  return dmx_fir_bwd_impl(dout, dout_stride, h, h_rev, din, din_stride, batch, Lin, Lout, taps, orig, new_, off, (hipStream_t)stream);
}
}
