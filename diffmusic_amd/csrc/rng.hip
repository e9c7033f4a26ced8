// Device-side standard-normal noise: counter-based Philox4x32-10 (Salmon et al., SC'11) + Box-Muller.
// Optional replacement for the reference's host draw + upload (diffmusic/torch_utils.py:31-76: `torch.randn` on a CPU
// generator, then `.to(device)`), which the DSG / DiffMusic schedulers pay every step (scheduling_dsg.py:215,
// scheduling_diffmusic.py:205).  The stream differs from torch's generators by construction, so this is opt-in
// (`device_noise=True` on the schedulers); what it keeps is the property the sharded path relies on: clip b's noise is a
// pure function of (seed_b, offset, element index), independent of the batch composition and of the number of GPUs.
//   element i of clip b = normal #(i & 3) of block (offset + i / 4) under key seed_b:
//     (u0,u1,u2,u3) = philox4x32_10(counter = (lo32(blk), hi32(blk), 0, 0), key = (lo32(seed), hi32(seed)))
//     n0 = r(u0) cos(2 pi v(u1)), n1 = r(u0) sin(2 pi v(u1)), n2, n3 likewise from (u2, u3)
//     r(u) = sqrt(-2 ln((u + 1) / 2^32)),  v(u) = u / 2^32
#include "dmx_common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

struct PhiloxSeeds { unsigned long long s[64]; };

__global__ void randn_philox_kernel(float* __restrict__ out, long long n, PhiloxSeeds seeds, unsigned long long offset) {
  const long long blk = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one Philox block = 4 normals
  const int b = blockIdx.y;
  if (blk * 4 >= n) return;
  const unsigned long long ctr = offset + (unsigned long long)blk, seed = seeds.s[b];
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  float v[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u = ((float)c[2 * h] + 1.0f) * 2.3283064365386963e-10f;      // (0, 1]
    const float ang = 6.283185307179586f * ((float)c[2 * h + 1] * 2.3283064365386963e-10f);
    const float rad = sqrtf(-2.0f * logf(u));
    v[2 * h] = rad * cosf(ang);
    v[2 * h + 1] = rad * sinf(ang);
  }
  float* o = out + (long long)b * n + blk * 4;
  if (blk * 4 + 3 < n && ((((uintptr_t)o) & 15) == 0)) {
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (blk * 4 + e < n) o[e] = v[e];
  }
}

}  // namespace

extern "C" int dmx_randn_philox(float* out, int batch, long long n, const unsigned long long* seeds_host, unsigned long long offset,
                                void* stream) {
  if (!out || batch < 1 || batch > 64 || n < 1 || !seeds_host) return DMX_ERR_SHAPE;
  PhiloxSeeds s;
  for (int b = 0; b < 64; ++b) s.s[b] = b < batch ? seeds_host[b] : 0ull;
  const long long blocks = (n + 3) / 4;
  hipLaunchKernelGGL(randn_philox_kernel, dim3((unsigned)((blocks + 255) / 256), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, out, n, s,
                     offset);
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}
