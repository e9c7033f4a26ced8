// Streaming-read bandwidth probe: what does a read-only pass reach on this device, by loads in flight per lane, cache policy and grid?
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/dev/read_bw.hip -o /tmp/read_bw && /tmp/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ x, float* __restrict__ out, long long n16, long long per_wg) {
  const long long base = (long long)blockIdx.x * per_wg;
  const long long end = base + per_wg < n16 ? base + per_wg : n16;
  float acc = 0.f;
  for (long long i = base + threadIdx.x; i < end; i += 256 * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long j = i + (long long)u * 256;
      if (j < end) v[u] = NT ? __builtin_nontemporal_load(x + j) : x[j]; else v[u] = u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
  }
  if (acc == 123.456f) out[0] = acc;
}
template <int U, bool NT>
void run(const u32x4* x, float* out, long long bytes, int wgs, const char* tag) {
  const long long n16 = bytes / 16, per = (n16 + wgs - 1) / wgs;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((rd<U, NT>), dim3(wgs), dim3(256), 0, 0, x, out, n16, per);
  hipEventRecord(a);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rd<U, NT>), dim3(wgs), dim3(256), 0, 0, x, out, n16, per);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-10s U=%d wgs=%6d  %7.1f us  %5.2f TB/s\n", tag, U, wgs, 1e3 * ms / reps, bytes / (ms / reps) / 1e9);
}
int main() {
  const long long big = 2048ll << 20;          // 2 GiB: far beyond the 256 MB Infinity Cache
  u32x4* x; float* out;
  hipMalloc(&x, big); hipMalloc(&out, 64); hipMemset(x, 1, big);
  for (int wgs : {1024, 4096, 16384}) {
    run<1, false>(x, out, big, wgs, "default"); run<4, false>(x, out, big, wgs, "default"); run<8, false>(x, out, big, wgs, "default");
    run<4, true>(x, out, big, wgs, "nt"); run<8, true>(x, out, big, wgs, "nt");
  }
  // one GroupNorm-sized tensor (131 MB), first touch: the rest of the 2 GiB buffer is rewritten before every launch
  for (int nt = 0; nt < 2; ++nt)
    for (int wgs : {1024, 4000, 16384}) {
      const long long bytes = 131072000, n16 = bytes / 16, per = (n16 + wgs - 1) / wgs;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      float tot = 0.f;
      for (int i = 0; i < 6; ++i) {
        hipMemsetAsync((char*)x + (512ll << 20), i, 1024ll << 20, 0);
        hipEventRecord(a);
        if (nt) hipLaunchKernelGGL((rd<4, true>), dim3(wgs), dim3(256), 0, 0, x, out, n16, per);
        else hipLaunchKernelGGL((rd<4, false>), dim3(wgs), dim3(256), 0, 0, x, out, n16, per);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (i) tot += ms;
      }
      printf("131 MB first touch %s wgs=%5d: %6.1f us  %5.2f TB/s\n", nt ? "nt     " : "default", wgs, 1e3 * tot / 5, bytes / (tot / 5) / 1e9);
    }
  return 0;
}
