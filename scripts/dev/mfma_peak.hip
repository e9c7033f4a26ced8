// measures the sustained rate of v_mfma_f32_16x16x32_f16 on this device (register operands, random data)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
template <int NACC>
__global__ __launch_bounds__(512) void k(const h8* __restrict__ in, float* out, int iters) {
  h8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[(threadIdx.x * 4 + i) % 4096];
  for (int i = 0; i < 2; ++i) b[i] = in[(threadIdx.x * 2 + i + 17) % 4096];
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i & 1], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  std::vector<_Float16> h(4096 * 8);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1);
  h8* din; float* dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 8 * 512 * 4);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512}) for (int rep = 0; rep < 2; ++rep) {
    const int iters = 4000;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 8 * iters * 32 * 16384.0;
    printf("blocks %d (8 waves each): %.3f ms  %.1f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
  }
  return 0;
}
