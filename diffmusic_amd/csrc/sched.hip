// Scheduler update kernels: the reference-owned arithmetic of the five Scheduler.step() bodies
// (diffmusic/schedulers/scheduling_{ddim,dps,mpgd,dsg,diffmusic}.py), one fused kernel per variant.
// Latents are tiny (c*h*w = 32 000 per clip), so one workgroup per clip does the elementwise update
// and the 1-3 L2 reductions (wave64 shuffles + LDS) in a single launch; all per-step scalars
// (alpha_bar_t, alpha_bar_prev, sigma_t) are computed on the host once per step -- no device sync.
// `global_norm` reproduces the reference's whole-batch norms (it only ever ran B=1); the default
// per-clip norms make a batch equal to B independent runs.
#include "dmx_common.h"
#include "kernels.h"

namespace {

// x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t)      (DDIMScheduler.step, epsilon prediction, no clipping)
__global__ void pred_x0_kernel(const float* __restrict__ x, const float* __restrict__ eps, float* __restrict__ x0, long long n,
                               float sqrt_a, float sqrt_1ma) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x0[i] = (x[i] - sqrt_1ma * eps[i]) / sqrt_a;
}

// prediction types of the diffusers DDIM parent (DDIMScheduler.step): how pred_original_sample comes out of (x_t, model_output), optionally
// clipped to [-r, r] (clip_sample).  The guided schedulers differentiate the loss w.r.t. x_t THROUGH x0 (scheduling_dps.py:163-212), so the
// update kernels need d x0 / d x_t: 1 / sqrt(a) (epsilon), sqrt(a) (v_prediction), 0 (sample), times the clip's pass-through mask.
enum { P_EPSILON = 0, P_SAMPLE = 1, P_V = 2 };
__global__ void pred_x0_ex_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ x0, long long n, float sqrt_a,
                                  float sqrt_1ma, int ptype, float clip_r) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v;
  if (ptype == P_EPSILON) v = (x[i] - sqrt_1ma * m[i]) / sqrt_a;
  else if (ptype == P_SAMPLE) v = m[i];
  else v = sqrt_a * x[i] - sqrt_1ma * m[i];
  if (clip_r > 0.f) v = fminf(fmaxf(v, -clip_r), clip_r);
  x0[i] = v;
}

struct SchedArgs {
  const float* x;        // sample x_t
  const float* eps;      // model_output
  const float* x0;       // pred_original_sample
  const float* g0;       // dL/dx0 as returned by the VAE backward (scaled by 1/inv_scale[b])
  const float* inv_scale;  // per-clip gradient unscale (may be null)
  const float* noise;    // variance / sample noise (may be null)
  float* prev;           // out: prev_sample
  float* x0_out;         // out: updated x0 (MPGD) or null
  float* grad_out;       // out: unscaled gradient wrt the differentiated variable (optional, for tests)
  int n;                 // elements per clip
  int B;
  float sqrt_a, sqrt_1ma, sqrt_ap, dir_coef, sigma, rate, eps_small, sqrt_1map;
  int global_norm;
  int ptype;             // prediction type of the parent step (P_EPSILON: the reference's configs)
  float clip_r;          // > 0: x0 was clipped to [-clip_r, clip_r] -- no gradient flows through a clipped element
};
// dL/dx_t of one element from dL/dx0 (already unscaled): the Jacobian of x0(x_t) for the prediction type, zero where x0 sits on the clip bound
__device__ __forceinline__ float x0_jac(const SchedArgs& a, float g, float x0) {
  if (a.ptype == P_EPSILON) g = g / a.sqrt_a;
  else if (a.ptype == P_V) g = g * a.sqrt_a;
  else g = 0.f;
  if (a.clip_r > 0.f && !(fabsf(x0) < a.clip_r)) g = 0.f;
  return g;
}

enum { M_DDIM = 0, M_DPS = 1, M_MPGD = 2, M_DSG = 3, M_DIFFMUSIC = 4 };

template <int MODE>
__global__ __launch_bounds__(1024) void sched_update_kernel(SchedArgs a) {
  __shared__ float sh[16];
  const int nclip = a.global_norm ? a.B : 1;
  const long long base = a.global_norm ? 0 : (long long)blockIdx.x * a.n;
  const long long tot = (long long)a.n * nclip;
  auto gscale = [&](long long i) -> float {
    if (!a.inv_scale) return 1.f;
    const int b = a.global_norm ? (int)(i / a.n) : (int)blockIdx.x;
    return a.inv_scale[b];
  };
  if (MODE == M_DDIM) {
    for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
      const long long j = base + i;
      const float e = (a.x[j] - a.sqrt_a * a.x0[j]) / a.sqrt_1ma;
      a.prev[j] = a.sqrt_ap * a.x0[j] + a.sqrt_1map * e;
    }
  } else if (MODE == M_DPS) {
    // prev = sqrt(ap) x0 + sqrt(1-ap-s^2) eps' + s*noise - rate * dL/dx,  dL/dx = dL/dx0 / sqrt(a_t)
    for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
      const long long j = base + i;
      const float x0 = a.x0[j];
      const float e = (a.x[j] - a.sqrt_a * x0) / a.sqrt_1ma;
      const float g = x0_jac(a, a.g0[j] * gscale(i), x0);
      float p = a.sqrt_ap * x0 + a.dir_coef * e;
      if (a.noise) p += a.sigma * a.noise[j];
      a.prev[j] = p - a.rate * g;
      if (a.grad_out) a.grad_out[j] = g;
    }
  } else if (MODE == M_MPGD) {
    for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
      const long long j = base + i;
      const float g = a.g0[j] * gscale(i);
      const float x0 = a.x0[j] - a.rate * g;
      const float e = (a.x[j] - a.sqrt_a * x0) / a.sqrt_1ma;
      float p = a.sqrt_ap * x0 + a.dir_coef * e;
      if (a.noise) p += a.sigma * a.noise[j];
      a.prev[j] = p;
      if (a.x0_out) a.x0_out[j] = x0;
      if (a.grad_out) a.grad_out[j] = g;
    }
  } else if (MODE == M_DSG) {
    // grad = d(loss/1000)/dx ; r = sqrt(n)*sigma ; d* = -r grad/(|grad|+e) ; mix = s z + rate (d* - s z)
    float s = 0.f;
    for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
      const float g = a.ptype == P_EPSILON && a.clip_r <= 0.f ? a.g0[base + i] * gscale(i) / (a.sqrt_a * 1000.f) : x0_jac(a, a.g0[base + i] * gscale(i), a.x0[base + i]) / 1000.f;
      s += g * g;
    }
    const float gnorm = sqrtf(block_sum(s, sh));
    const float r = sqrtf((float)a.n) * a.sigma;
    float m = 0.f;
    for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
      const long long j = base + i;
      const float g = a.ptype == P_EPSILON && a.clip_r <= 0.f ? a.g0[j] * gscale(i) / (a.sqrt_a * 1000.f) : x0_jac(a, a.g0[j] * gscale(i), a.x0[j]) / 1000.f;
      const float ds = a.sigma * a.noise[j];
      const float mix = ds + a.rate * (-r * g / (gnorm + a.eps_small) - ds);
      m += mix * mix;
    }
    const float mnorm = sqrtf(block_sum(m, sh));
    for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
      const long long j = base + i;
      const float g = a.ptype == P_EPSILON && a.clip_r <= 0.f ? a.g0[j] * gscale(i) / (a.sqrt_a * 1000.f) : x0_jac(a, a.g0[j] * gscale(i), a.x0[j]) / 1000.f;
      const float ds = a.sigma * a.noise[j];
      const float mix = ds + a.rate * (-r * g / (gnorm + a.eps_small) - ds);
      const float mean = a.sqrt_ap * a.x0[j] + a.dir_coef * a.eps[j];
      a.prev[j] = mean + r * mix / (mnorm + a.eps_small);
      if (a.grad_out) a.grad_out[j] = g;
    }
  } else {  // M_DIFFMUSIC: slerp(z, -ghat, rate), ghat = grad/(|grad|+e) * |z|
    float s = 0.f, zz = 0.f;
    for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
      const float g = a.ptype == P_EPSILON && a.clip_r <= 0.f ? a.g0[base + i] * gscale(i) / (a.sqrt_a * 1000.f) : x0_jac(a, a.g0[base + i] * gscale(i), a.x0[base + i]) / 1000.f;
      const float z = a.noise[base + i];
      s += g * g; zz += z * z;
    }
    const float gnorm = sqrtf(block_sum(s, sh));
    const float znorm = sqrtf(block_sum(zz, sh));
    const float gh = znorm / (gnorm + a.eps_small);      // ghat = g * gh ; x1 = -ghat
    float d = 0.f, x1n = 0.f;
    for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
      const float g = a.ptype == P_EPSILON && a.clip_r <= 0.f ? a.g0[base + i] * gscale(i) / (a.sqrt_a * 1000.f) : x0_jac(a, a.g0[base + i] * gscale(i), a.x0[base + i]) / 1000.f;
      const float x1 = -g * gh;
      d += a.noise[base + i] * x1; x1n += x1 * x1;
    }
    d = block_sum(d, sh);
    x1n = sqrtf(block_sum(x1n, sh));
    const float cos_t = d / (znorm * x1n);
    float w0, w1;
    if (fabsf(cos_t) > 0.9995f) { w0 = 1.f - a.rate; w1 = a.rate; }
    else {
      const float th = acosf(cos_t), st = sinf(th);
      w0 = sinf((1.f - a.rate) * th) / st; w1 = sinf(a.rate * th) / st;
    }
    for (long long i = threadIdx.x; i < tot; i += blockDim.x) {
      const long long j = base + i;
      const float g = a.ptype == P_EPSILON && a.clip_r <= 0.f ? a.g0[j] * gscale(i) / (a.sqrt_a * 1000.f) : x0_jac(a, a.g0[j] * gscale(i), a.x0[j]) / 1000.f;
      const float mixed = w0 * a.noise[j] + w1 * (-g * gh);
      const float mean = a.sqrt_ap * a.x0[j] + a.dir_coef * a.eps[j];
      a.prev[j] = mean + a.sigma * mixed;
      if (a.grad_out) a.grad_out[j] = g;
    }
  }
}

// eps = uncond + s (text - uncond) on the two halves of the CFG batch
__global__ void cfg_combine_kernel(const float* __restrict__ eps2, float* __restrict__ out, long long n, float scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float u = eps2[i], t = eps2[n + i]; out[i] = u + scale * (t - u); }
}

}  // namespace

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH)

int dmx_pred_x0(const float* x, const float* eps, float* x0, long long n, float sqrt_a, float sqrt_1ma, hipStream_t st) {
  hipLaunchKernelGGL(pred_x0_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, eps, x0, n, sqrt_a, sqrt_1ma);
  return CHECK_LAUNCH();
}
int dmx_cfg_combine(const float* eps2, float* out, long long n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(cfg_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, eps2, out, n, scale);
  return CHECK_LAUNCH();
}
int dmx_pred_x0_ex(const float* x, const float* m, float* x0, long long n, float sqrt_a, float sqrt_1ma, int ptype, float clip_r, hipStream_t st) {
  if (ptype < P_EPSILON || ptype > P_V) return DMX_ERR_SHAPE;
  hipLaunchKernelGGL(pred_x0_ex_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, m, x0, n, sqrt_a, sqrt_1ma, ptype, clip_r);
  return CHECK_LAUNCH();
}
int dmx_sched_update(int mode, const float* x, const float* eps, const float* x0, const float* g0, const float* inv_scale,
                     const float* noise, float* prev, float* x0_out, float* grad_out, int B, int n, float alpha_t, float alpha_prev,
                     float sigma, float rate, float eps_small, int global_norm, hipStream_t st, int ptype, float clip_r) {
  if (ptype < P_EPSILON || ptype > P_V) return DMX_ERR_SHAPE;
  SchedArgs a;
  a.ptype = ptype; a.clip_r = clip_r;
  a.x = x; a.eps = eps; a.x0 = x0; a.g0 = g0; a.inv_scale = inv_scale; a.noise = noise;
  a.prev = prev; a.x0_out = x0_out; a.grad_out = grad_out; a.n = n; a.B = B;
  a.sqrt_a = sqrtf(alpha_t); a.sqrt_1ma = sqrtf(1.f - alpha_t); a.sqrt_ap = sqrtf(alpha_prev);
  a.dir_coef = sqrtf(fmaxf(1.f - alpha_prev - sigma * sigma, 0.f));
  a.sqrt_1map = sqrtf(1.f - alpha_prev);
  a.sigma = sigma; a.rate = rate; a.eps_small = eps_small; a.global_norm = global_norm;
  if ((mode == M_DSG || mode == M_DIFFMUSIC) && !noise) return DMX_ERR_SHAPE;
  if (mode != M_DDIM && !g0) return DMX_ERR_SHAPE;
  const dim3 grid(global_norm ? 1 : B), block(1024);
  switch (mode) {
    case M_DDIM: hipLaunchKernelGGL(sched_update_kernel<M_DDIM>, grid, block, 0, st, a); break;
    case M_DPS: hipLaunchKernelGGL(sched_update_kernel<M_DPS>, grid, block, 0, st, a); break;
    case M_MPGD: hipLaunchKernelGGL(sched_update_kernel<M_MPGD>, grid, block, 0, st, a); break;
    case M_DSG: hipLaunchKernelGGL(sched_update_kernel<M_DSG>, grid, block, 0, st, a); break;
    case M_DIFFMUSIC: hipLaunchKernelGGL(sched_update_kernel<M_DIFFMUSIC>, grid, block, 0, st, a); break;
    default: return DMX_ERR_SHAPE;
  }
  return CHECK_LAUNCH();
}
