// Parameter registry, weight packing and GemmDesc builders (see layers.h).
#include "layers.h"
#include "conv_pair.h"
#include <cstdarg>

static thread_local char g_err[512] = "";
void dmx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dmx_last_error() { return g_err; }

// ------------------------------------------------------------------------------ ParamStore
int ParamStore::add(const std::string& name, std::vector<int> shape) {
  Param p;
  p.name = name;
  p.shape = shape;
  p.numel = 1;
  for (int s : shape) p.numel *= (size_t)s;
  p.dev = (float*)dalloc(p.numel * sizeof(float));
  index[name] = (int)params.size();
  params.push_back(p);
  return (int)params.size() - 1;
}
void* ParamStore::dalloc(size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { dmx_set_error("hipMalloc(%zu) failed", bytes); return nullptr; }
  owned.push_back(p);
  return p;
}
int ParamStore::load(const char* name, const float* host, size_t numel) {
  auto it = index.find(name);
  if (it == index.end()) { dmx_set_error("unknown parameter '%s'", name); return DMX_ERR_PARAM; }
  Param& p = params[it->second];
  if (p.numel != numel) { dmx_set_error("parameter '%s': expected %zu elements, got %zu", name, p.numel, numel); return DMX_ERR_PARAM; }
  if (hipMemcpy(p.dev, host, numel * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    dmx_set_error("hipMemcpy failed for '%s'", name);
    return DMX_ERR_PARAM;
  }
  p.loaded = true;
  return DMX_OK;
}
bool ParamStore::all_loaded(std::string* missing) const {
  for (const Param& p : params)
    if (!p.loaded) { if (missing) *missing = p.name; return false; }
  return true;
}
void ParamStore::free_all() {
  for (void* p : owned) (void)hipFree(p);
  owned.clear();
}

// ------------------------------------------------------------------------------ packing
struct PackTaps { int idx[DMX_MAX_TAPS]; };
// packed row p of a GEGLU projection with `half` value rows followed by `half` gate rows: blocks of 32 = [16 values | their 16 gates]
__host__ __device__ inline int geglu_src_row(int p, int half) {
  const int b = p >> 5, q = p & 31;
  return q < 16 ? 16 * b + q : half + 16 * b + (q - 16);
}
__global__ void pack_weight_kernel(const float* __restrict__ src, act_t* __restrict__ dst, int Np, int Nreal, int T, int Cp,
                                   int Creal, long long sn, long long sc, long long st, PackTaps taps, int geglu_half) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Np * T * Cp;
  if (idx >= total) return;
  const int c = (int)(idx % Cp);
  const int t = (int)((idx / Cp) % T);
  int n = (int)(idx / ((long long)Cp * T));
  if (geglu_half > 0) n = geglu_src_row(n, geglu_half);
  float v = 0.f;
  if (n < Nreal && c < Creal && taps.idx[t] >= 0) v = src[n * sn + c * sc + taps.idx[t] * st];
  dst[idx] = f2a(v);
}
__global__ void pad_bias_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int np, int geglu_half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = geglu_half > 0 ? geglu_src_row(i, geglu_half) : i;
  if (i < np) dst[i] = (src && s < n) ? src[s] : 0.f;
}

// LayerNorm fold (ConvLayer::ln_g_id): Wf[n][k] = W[n][k] * gamma[k], bf[n] = bias[n] + sum_k W[n][k] * beta[k]   (fp32; one block per row)
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bias, float* __restrict__ Wf, float* __restrict__ bf, int K) {
  __shared__ float sh[16];
  const int n = blockIdx.x;
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float w = W[(long long)n * K + k];
    Wf[(long long)n * K + k] = w * gamma[k];
    acc += w * beta[k];
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) bf[n] = (bias ? bias[n] : 0.f) + acc;
}
// colsum[p] = sum_k of the PACKED 16-bit row p (what the MFMAs multiply by; fixed summation order: bit-reproducible); one wave per row
__global__ __launch_bounds__(64) void packed_rowsum_kernel(const act_t* __restrict__ Wp, float* __restrict__ colsum, int Kp) {
  const int p = blockIdx.x, lane = threadIdx.x;
  float acc = 0.f;
  for (int k = lane; k < Kp; k += 64) acc += a2f(Wp[(long long)p * Kp + k]);
  acc = wave_sum(acc);
  if (lane == 0) colsum[p] = acc;
}

static act_t* pack(ParamStore& ps, const float* src, int Np, int Nreal, int T, int Cp, int Creal, long long sn, long long sc,
                    long long st_, const std::vector<int>& tapidx, hipStream_t st, int geglu_half = 0) {
  const long long total = (long long)Np * T * Cp;
  act_t* dst = (act_t*)ps.dalloc(total * 2);
  if (!dst) return nullptr;
  PackTaps pt;
  for (int i = 0; i < DMX_MAX_TAPS; ++i) pt.idx[i] = i < (int)tapidx.size() ? tapidx[i] : -1;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, dst, Np, Nreal, T, Cp,
                     Creal, sn, sc, st_, pt, geglu_half);
  return dst;
}

// packed[n][t][c] = sum over the (<= 4) source taps listed for packed tap t of src[n][c][tap]  (fp32 sum, one rounding)
struct PackSumTaps { int idx[DMX_MAX_TAPS][4]; };
__global__ void pack_weight_sum_kernel(const float* __restrict__ src, act_t* __restrict__ dst, int Np, int Nreal, int T, int Cp,
                                       int Creal, long long sn, long long sc, long long st, PackSumTaps taps) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Np * T * Cp;
  if (idx >= total) return;
  const int c = (int)(idx % Cp);
  const int t = (int)((idx / Cp) % T);
  const int n = (int)(idx / ((long long)Cp * T));
  float v = 0.f;
  if (n < Nreal && c < Creal)
    for (int q = 0; q < 4; ++q) if (taps.idx[t][q] >= 0) v += src[n * sn + c * sc + taps.idx[t][q] * st];
  dst[idx] = f2a(v);
}

// Nearest x2 upsampling followed by a 3x3 / pad 1 convolution reads, for an output pixel of parity (py, px), only a 2x2
// neighbourhood of the low-resolution image: rows {y-1, y} (py = 0: upsampled rows 2y-1 | 2y, 2y+1) or {y, y+1} (py = 1: 2y, 2y+1 |
// 2y+2), likewise in x.  Summing the 3x3 weights that land on the same low-resolution pixel gives four 2x2-tap convolutions
// (4/9 of the multiply-adds, no upsampled tensor in HBM); zero padding of the upsampled image maps onto zero padding of the
// low-resolution one.  rows_of(p, t): the kernel rows summed into tap t of parity p.
static void up2x_rows(int p, int t, int (&k)[2]) {
  k[0] = k[1] = -1;
  if (p == 0) { if (t == 0) k[0] = 0; else { k[0] = 1; k[1] = 2; } }
  else { if (t == 0) { k[0] = 0; k[1] = 1; } else k[0] = 2; }
}
static int up2x_delta(int p, int t) { return p == 0 ? t - 1 : t; }      // low-resolution offset of tap t at parity p

int pack_layer_up2x(ParamStore& ps, ConvLayer& L, hipStream_t st) {
  if (L.transposed || L.kh != 3 || L.kw != 3 || L.stride != 1 || L.pad_h != 1 || L.pad_w != 1) return DMX_ERR_SHAPE;
  const float* w = ps.dev(L.w_id);               // W[Co][Ci][3][3]
  const int T9 = 9;
  L.wup.clear();
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      PackSumTaps pt;
      for (auto& r : pt.idx) for (int& v : r) v = -1;
      for (int ty = 0; ty < 2; ++ty)
        for (int tx = 0; tx < 2; ++tx) {
          int ky[2], kx[2], n = 0;
          up2x_rows(py, ty, ky); up2x_rows(px, tx, kx);
          for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) if (ky[a] >= 0 && kx[b] >= 0) pt.idx[ty * 2 + tx][n++] = ky[a] * 3 + kx[b];
        }
      const long long total = (long long)L.Cop * 4 * L.Cip;
      act_t* dst = (act_t*)ps.dalloc(total * 2);
      if (!dst) return DMX_ERR_PARAM;
      hipLaunchKernelGGL(pack_weight_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, dst, L.Cop, L.Co, 4, L.Cip, L.Ci,
                         (long long)L.Ci * T9, (long long)T9, 1ll, pt);
      L.wup.push_back(dst);
    }
  if (L.need_bwd) {
    // joint dgrad: din[y, x] = sum over (py, ty, px, tx) of Wsum^T . dout[2 (y - dy) + py, 2 (x - dx) + px]: 16 taps at offsets
    // py - 2 dy in {2, 0, 1, -1} of a stride-2 walk over dout; packed [Cip][16][Cop]
    PackSumTaps pt;
    for (auto& r : pt.idx) for (int& v : r) v = -1;
    for (int py = 0; py < 2; ++py) for (int ty = 0; ty < 2; ++ty)
      for (int px = 0; px < 2; ++px) for (int tx = 0; tx < 2; ++tx) {
        int ky[2], kx[2], n = 0;
        up2x_rows(py, ty, ky); up2x_rows(px, tx, kx);
        const int t = (py * 2 + ty) * 4 + (px * 2 + tx);
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) if (ky[a] >= 0 && kx[b] >= 0) pt.idx[t][n++] = ky[a] * 3 + kx[b];
      }
    const long long total = (long long)L.Cip * 16 * L.Cop;
    L.wup_b = (act_t*)ps.dalloc(total * 2);
    if (!L.wup_b) return DMX_ERR_PARAM;
    hipLaunchKernelGGL(pack_weight_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, L.wup_b, L.Cip, L.Ci, 16, L.Cop, L.Co,
                       (long long)T9, (long long)L.Ci * T9, 1ll, pt);
  }
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}

static ConvLayer base_layer(ParamStore& ps, const std::string& prefix, std::vector<int> wshape, int Co, bool bias) {
  ConvLayer L;
  L.w_id = ps.add(prefix + ".weight", wshape);
  L.has_bias = bias;
  if (bias) L.b_id = ps.add(prefix + ".bias", {Co});
  return L;
}
ConvLayer make_conv1d(ParamStore& ps, const std::string& prefix, int Ci, int Co, int k, int dil, int pad, bool need_bwd) {
  ConvLayer L = base_layer(ps, prefix, {Co, Ci, k}, Co, true);
  L.Ci = Ci; L.Co = Co; L.Cip = pad8(Ci); L.Cop = pad8(Co);
  L.kh = 1; L.kw = k; L.dil = dil; L.pad_w = pad; L.need_bwd = need_bwd;
  return L;
}
ConvLayer make_convT1d(ParamStore& ps, const std::string& prefix, int Ci, int Co, int k, int stride, int pad, bool need_bwd) {
  ConvLayer L = base_layer(ps, prefix, {Ci, Co, k}, Co, true);
  L.Ci = Ci; L.Co = Co; L.Cip = pad8(Ci); L.Cop = pad8(Co);
  L.kh = 1; L.kw = k; L.stride = stride; L.pad_w = pad; L.transposed = true; L.need_bwd = need_bwd;
  return L;
}
ConvLayer make_conv2d(ParamStore& ps, const std::string& prefix, int Ci, int Co, int k, int stride, int pad, bool need_bwd) {
  ConvLayer L = base_layer(ps, prefix, {Co, Ci, k, k}, Co, true);
  L.Ci = Ci; L.Co = Co; L.Cip = pad8(Ci); L.Cop = pad8(Co);
  L.kh = k; L.kw = k; L.stride = stride; L.pad_h = pad; L.pad_w = pad; L.need_bwd = need_bwd;
  return L;
}
ConvLayer make_linear(ParamStore& ps, const std::string& prefix, int Ci, int Co, bool bias, bool need_bwd) {
  ConvLayer L = base_layer(ps, prefix, {Co, Ci}, Co, bias);
  L.Ci = Ci; L.Co = Co; L.Cip = pad8(Ci); L.Cop = pad8(Co);
  L.need_bwd = need_bwd;
  return L;
}
GroupNormLayer make_gn(ParamStore& ps, const std::string& prefix, int C, int G, float eps) {
  GroupNormLayer g;
  g.C = C; g.G = G; g.eps = eps;
  g.g_id = ps.add(prefix + ".weight", {C});
  g.b_id = ps.add(prefix + ".bias", {C});
  return g;
}

int pack_layer(ParamStore& ps, ConvLayer& L, hipStream_t st) {
  const float* w = ps.dev(L.w_id);
  const int T = L.ntaps();
  if (T > DMX_MAX_TAPS) { dmx_set_error("too many taps"); return DMX_ERR_SHAPE; }
  std::vector<int> all(T);
  for (int i = 0; i < T; ++i) all[i] = i;
  int gh = 0;
  if (L.geglu) {
    if (L.transposed || T != 1 || L.need_bwd || (L.Co & 31) || L.Cop != L.Co) { dmx_set_error("GEGLU packing needs a forward-only linear layer with Cout %% 32 == 0"); return DMX_ERR_SHAPE; }
    gh = L.Co / 2;
  }
  float* ln_w = nullptr;       // LayerNorm fold: temporary fp32 W diag(gamma) and the folded bias
  float* ln_bias = nullptr;
  if (L.ln_g_id >= 0) {
    if (L.transposed || T != 1 || L.need_bwd) { dmx_set_error("LayerNorm fold needs a forward-only linear layer"); return DMX_ERR_SHAPE; }
    if (hipMalloc(&ln_w, (size_t)L.Co * L.Ci * sizeof(float)) != hipSuccess) return DMX_ERR_PARAM;
    ln_bias = (float*)ps.dalloc((size_t)L.Co * sizeof(float));
    if (!ln_bias) { (void)hipFree(ln_w); return DMX_ERR_PARAM; }
    hipLaunchKernelGGL(ln_fold_kernel, dim3(L.Co), dim3(256), 0, st, w, ps.dev(L.ln_g_id), ps.dev(L.ln_b_id),
                       L.has_bias ? ps.dev(L.b_id) : (const float*)nullptr, ln_w, ln_bias, L.Ci);
    w = ln_w;
  }
  if (!L.transposed) {
    // W[Co][Ci][T]
    L.wf.push_back(pack(ps, w, L.Cop, L.Co, T, L.Cip, L.Ci, (long long)L.Ci * T, T, 1, all, st, gh));
    if (L.need_bwd) L.wb = pack(ps, w, L.Cip, L.Ci, T, L.Cop, L.Co, T, (long long)L.Ci * T, 1, all, st);
  } else {
    // W[Ci][Co][k]; one packed matrix per output phase
    const int k = L.kw, s = L.stride, p = L.pad_w;
    for (int r = 0; r < s; ++r) {
      const int j0 = (r + p) % s;
      std::vector<int> taps;
      for (int j = j0; j < k; j += s) taps.push_back(j);
      L.wf_taps.push_back(taps);
      L.wf.push_back(pack(ps, w, L.Cop, L.Co, (int)taps.size(), L.Cip, L.Ci, k, (long long)L.Co * k, 1, taps, st));
    }
    if (L.need_bwd) L.wb = pack(ps, w, L.Cip, L.Ci, k, L.Cop, L.Co, (long long)L.Co * k, k, 1, all, st);
  }
  L.bias = (float*)ps.dalloc(L.Cop * sizeof(float));
  hipLaunchKernelGGL(pad_bias_kernel, dim3(cdiv(L.Cop, 256)), dim3(256), 0, st,
                     ln_bias ? (const float*)ln_bias : (L.has_bias ? ps.dev(L.b_id) : (const float*)nullptr), L.bias, L.Co, L.Cop, gh);
  for (act_t* pw : L.wf) if (!pw) { if (ln_w) (void)hipFree(ln_w); return DMX_ERR_PARAM; }
  if (ln_w) {
    L.has_bias = true;                      // W beta (zero when beta is)
    L.colsum = (float*)ps.dalloc(L.Cop * sizeof(float));
    if (!L.colsum) { (void)hipFree(ln_w); return DMX_ERR_PARAM; }
    hipLaunchKernelGGL(packed_rowsum_kernel, dim3(L.Cop), dim3(64), 0, st, L.wf[0], L.colsum, L.Cip);
    (void)hipStreamSynchronize(st);         // model-load time: the temporary is read by the kernels above
    (void)hipFree(ln_w);
  }
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}

int pack_linear_stack(ParamStore& ps, ConvLayer& dst, const ConvLayer* const* src, int n, hipStream_t st) {
  if (n < 1) return DMX_ERR_SHAPE;
  const int Ci = src[0]->Ci;
  int Co = 0;
  bool bias = false, bwd = false;
  for (int i = 0; i < n; ++i) {
    if (src[i]->Ci != Ci || src[i]->ntaps() != 1 || src[i]->transposed) { dmx_set_error("stacked projection: linear layers of equal input width"); return DMX_ERR_SHAPE; }
    Co += src[i]->Co; bias = bias || src[i]->has_bias; bwd = bwd || src[i]->need_bwd;
  }
  float *w = nullptr, *b = nullptr;
  if (hipMalloc(&w, (size_t)Co * Ci * sizeof(float)) != hipSuccess || hipMalloc(&b, (size_t)Co * sizeof(float)) != hipSuccess) {
    if (w) (void)hipFree(w);
    return DMX_ERR_PARAM;
  }
  (void)hipMemsetAsync(b, 0, (size_t)Co * sizeof(float), st);
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    (void)hipMemcpyAsync(w + off * Ci, ps.dev(src[i]->w_id), (size_t)src[i]->Co * Ci * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (src[i]->has_bias) (void)hipMemcpyAsync(b + off, ps.dev(src[i]->b_id), (size_t)src[i]->Co * sizeof(float), hipMemcpyDeviceToDevice, st);
    off += src[i]->Co;
  }
  dst = ConvLayer();
  dst.Ci = Ci; dst.Co = Co; dst.Cip = pad8(Ci); dst.Cop = pad8(Co); dst.has_bias = bias; dst.need_bwd = bwd;
  const std::vector<int> all(1, 0);
  dst.wf.push_back(pack(ps, w, dst.Cop, Co, 1, dst.Cip, Ci, Ci, 1, 1, all, st));
  if (bwd) dst.wb = pack(ps, w, dst.Cip, Ci, 1, dst.Cop, Co, 1, Ci, 1, all, st);
  dst.bias = (float*)ps.dalloc(dst.Cop * sizeof(float));
  int rc = DMX_OK;
  if (!dst.wf[0] || (bwd && !dst.wb) || !dst.bias) rc = DMX_ERR_PARAM;
  else hipLaunchKernelGGL(pad_bias_kernel, dim3(cdiv(dst.Cop, 256)), dim3(256), 0, st, (const float*)b, dst.bias, Co, dst.Cop, 0);
  (void)hipStreamSynchronize(st);           // model-load time: the temporaries are read by the kernels above
  (void)hipFree(w); (void)hipFree(b);
  if (rc == DMX_OK && hipGetLastError() != hipSuccess) rc = DMX_ERR_LAUNCH;
  return rc;
}

// ------------------------------------------------------------------------------ descriptors
static void init_desc(GemmDesc& d, const Epi& e) {
  memset(&d, 0, sizeof(d));
  d.Z = 1; d.Zi = 1;
  d.sy = d.sx = 1; d.osy = d.osx = 1;
  d.alpha = e.alpha; d.act_slope = e.act_slope; d.mask_slope = e.mask_slope; d.resid_inv_slope = e.resid_inv_slope;
  d.flags = e.flags;
  d.R = e.R; d.X = e.X; d.C2 = e.C2; d.rowbias = e.rowbias; d.ldrb = e.ldrb;
  d.XB = e.XB; d.B2 = e.B2;
  d.rowstats_out = e.rowstats_out; d.rowstats_in = e.rowstats_in; d.nslots = e.nslots;
  d.gn_part = e.gn_part;
  if (e.gn_part) d.flags |= e.gnb_x ? EPI_GNBWD : EPI_GNSTATS;
  d.gnb_x = e.gnb_x; d.gnb_scale = e.gnb_scale; d.gnb_shift = e.gnb_shift; d.gnb_silu = e.gnb_silu;
  d.gnb_stats = e.gnb_stats; d.gnb_cpg = e.gnb_cpg;
}
static void set_out(GemmDesc& d, void* C, int Ho, int Wo, int ldc) {
  d.C = C; d.Ho = Ho; d.Wo = Wo; d.ldc = d.ldr = d.ldx = d.ldc2 = ldc;
  d.ldxb = d.ldb2 = ldc >> 3;
  d.gnb_ldx = ldc;
}

int conv_out_len(const ConvLayer& L, int Ti) {
  if (L.transposed) return (Ti - 1) * L.stride - 2 * L.pad_w + L.kw;
  return (Ti + 2 * L.pad_w - L.dil * (L.kw - 1) - 1) / L.stride + 1;
}

static void fwd_1d_common(GemmDesc& d, const ConvLayer& L, const act_t* in, void* out, int Ti, int To, const Epi& e) {
  init_desc(d, e);
  if (L.has_bias) { d.bias = L.bias; d.flags |= EPI_BIAS; }
  d.A = in; d.Hi = 1; d.Wi = Ti; d.Ci = L.Cip; d.lda = L.Cip;
  d.N = L.Cop;
  set_out(d, out, 1, To, L.Cop);
}
int conv_fwd_1d_desc(const ConvLayer& L, const act_t* in, void* out, int B, int Ti, const Epi& e, GemmDesc& d) {
  if (L.transposed) return DMX_ERR_SHAPE;
  const int To = conv_out_len(L, Ti);
  fwd_1d_common(d, L, in, out, Ti, To, e);
  d.W = L.wf[0]; d.ntaps = L.kw; d.K = L.kw * L.Cip; d.ldw = d.K;
  d.Hq = 1; d.Wq = To; d.sx = L.stride; d.M = B * To;
  for (int t = 0; t < L.kw; ++t) { d.tdy[t] = 0; d.tdx[t] = (signed char)(t * L.dil - L.pad_w); }
  return DMX_OK;
}

int conv_fwd_1d(const ConvLayer& L, const act_t* in, void* out, int B, int Ti, const Epi& e, hipStream_t st) {
  const int To = conv_out_len(L, Ti);
  GemmDesc d;
  if (!L.transposed) {
    const int rc = conv_fwd_1d_desc(L, in, out, B, Ti, e, d);
    return rc != DMX_OK ? rc : dmx_gemm_launch(d, st);
  }
  fwd_1d_common(d, L, in, out, Ti, To, e);
  const int s = L.stride, p = L.pad_w;
  for (int r = 0; r < s; ++r) {
    GemmDesc q = d;
    const int nt = (int)L.wf_taps[r].size();
    const int base = (r + p) / s;
    q.W = L.wf[r]; q.ntaps = nt; q.K = nt * L.Cip; q.ldw = q.K;
    q.Hq = 1; q.Wq = cdiv(To - r, s); q.M = B * q.Wq;
    q.osx = s; q.oox = r;
    for (int i = 0; i < nt; ++i) { q.tdy[i] = 0; q.tdx[i] = (signed char)(base - i); }
    if (q.Wq <= 0) continue;
    const int rc = dmx_gemm_launch(q, st);
    if (rc != DMX_OK) return rc;
  }
  return DMX_OK;
}

// dgrad: dout (B, To, Cop) -> din (B, Ti, Cip)
int conv_bwd_1d_desc(const ConvLayer& L, const act_t* dout, void* din, int B, int Ti, const Epi& e, GemmDesc& d) {
  if (!L.wb) { dmx_set_error("layer has no dgrad weights"); return DMX_ERR_STATE; }
  const int To = conv_out_len(L, Ti);
  init_desc(d, e);
  d.A = dout; d.Hi = 1; d.Wi = To; d.Ci = L.Cop; d.lda = L.Cop;
  d.W = L.wb; d.ntaps = L.kw; d.K = L.kw * L.Cop; d.ldw = d.K;
  d.N = L.Cip;
  d.Hq = 1; d.Wq = Ti; d.M = B * Ti;
  set_out(d, din, 1, Ti, L.Cip);
  if (!L.transposed) {
    if (L.stride != 1) return DMX_ERR_SHAPE;
    for (int t = 0; t < L.kw; ++t) { d.tdy[t] = 0; d.tdx[t] = (signed char)(L.pad_w - t * L.dil); }
  } else {
    d.sx = L.stride;
    for (int t = 0; t < L.kw; ++t) { d.tdy[t] = 0; d.tdx[t] = (signed char)(t - L.pad_w); }
  }
  return DMX_OK;
}
int conv_bwd_1d(const ConvLayer& L, const act_t* dout, void* din, int B, int Ti, const Epi& e, hipStream_t st) {
  GemmDesc d;
  const int rc = conv_bwd_1d_desc(L, dout, din, B, Ti, e, d);
  return rc != DMX_OK ? rc : dmx_gemm_launch(d, st);
}

// stage `a` then stage `b` (b.A is what a produces): one fused launch when the pair kernel takes the shape, else two launches
int conv_pair_run(const GemmDesc& a, const GemmDesc& b, hipStream_t st) {
  if (dmx_conv_pair_eligible(&a, b)) return dmx_conv_pair_launch(&a, b, st);
  const int rc = dmx_gemm_launch(a, st);
  return rc != DMX_OK ? rc : dmx_gemm_launch(b, st);
}

int conv_fwd_2d(const ConvLayer& L, const act_t* in, void* out, int B, int Hi, int Wi, const Epi& e, hipStream_t st) {
  const int Ho = (Hi + 2 * L.pad_h - L.kh) / L.stride + 1, Wo = (Wi + 2 * L.pad_w - L.kw) / L.stride + 1;
  GemmDesc d;
  init_desc(d, e);
  if (L.has_bias) { d.bias = L.bias; d.flags |= EPI_BIAS; }
  d.A = in; d.Hi = Hi; d.Wi = Wi; d.Ci = L.Cip; d.lda = L.Cip;
  d.W = L.wf[0]; d.ntaps = L.kh * L.kw; d.K = d.ntaps * L.Cip; d.ldw = d.K;
  d.N = L.Cop; d.Hq = Ho; d.Wq = Wo; d.sy = d.sx = L.stride; d.M = B * Ho * Wo;
  set_out(d, out, Ho, Wo, L.Cop);
  for (int ky = 0; ky < L.kh; ++ky)
    for (int kx = 0; kx < L.kw; ++kx) {
      d.tdy[ky * L.kw + kx] = (signed char)(ky - L.pad_h);
      d.tdx[ky * L.kw + kx] = (signed char)(kx - L.pad_w);
    }
  return dmx_gemm_launch(d, st);
}

int conv_up2x_fwd(const ConvLayer& L, const act_t* in, void* out, int B, int Hi, int Wi, const Epi& e, hipStream_t st,
                  float* const* gn_buf, int* gn_tm) {
  if (L.wup.size() != 4) { dmx_set_error("layer has no upsample-folded weights"); return DMX_ERR_STATE; }
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      GemmDesc d;
      Epi ep = e;
      if (gn_buf) ep.gn_part = gn_buf[py * 2 + px];
      init_desc(d, ep);
      if (L.has_bias) { d.bias = L.bias; d.flags |= EPI_BIAS; }
      d.A = in; d.Hi = Hi; d.Wi = Wi; d.Ci = L.Cip; d.lda = L.Cip;
      d.W = L.wup[py * 2 + px]; d.ntaps = 4; d.K = 4 * L.Cip; d.ldw = d.K;
      d.N = L.Cop; d.Hq = Hi; d.Wq = Wi; d.M = B * Hi * Wi;
      set_out(d, out, 2 * Hi, 2 * Wi, L.Cop);
      d.osy = d.osx = 2; d.ooy = py; d.oox = px;
      for (int ty = 0; ty < 2; ++ty)
        for (int tx = 0; tx < 2; ++tx) {
          d.tdy[ty * 2 + tx] = (signed char)up2x_delta(py, ty);
          d.tdx[ty * 2 + tx] = (signed char)up2x_delta(px, tx);
        }
      const int rc = dmx_gemm_launch(d, st);
      if (rc != DMX_OK) return rc;
      if (gn_tm) gn_tm[py * 2 + px] = dmx_gemm_last_tile_rows();
    }
  return DMX_OK;
}

int conv_up2x_bwd(const ConvLayer& L, const act_t* dout, void* din, int B, int Hi, int Wi, const Epi& e, hipStream_t st) {
  if (!L.wup_b) { dmx_set_error("layer has no upsample-folded dgrad weights"); return DMX_ERR_STATE; }
  GemmDesc d;
  init_desc(d, e);
  d.A = dout; d.Hi = 2 * Hi; d.Wi = 2 * Wi; d.Ci = L.Cop; d.lda = L.Cop;
  d.W = L.wup_b; d.ntaps = 16; d.K = 16 * L.Cop; d.ldw = d.K;
  d.N = L.Cip; d.Hq = Hi; d.Wq = Wi; d.sy = d.sx = 2; d.M = B * Hi * Wi;
  set_out(d, din, Hi, Wi, L.Cip);
  for (int py = 0; py < 2; ++py) for (int ty = 0; ty < 2; ++ty)
    for (int px = 0; px < 2; ++px) for (int tx = 0; tx < 2; ++tx) {
      const int t = (py * 2 + ty) * 4 + (px * 2 + tx);
      d.tdy[t] = (signed char)(py - 2 * up2x_delta(py, ty));
      d.tdx[t] = (signed char)(px - 2 * up2x_delta(px, tx));
    }
  return dmx_gemm_launch(d, st);
}

int conv_bwd_2d(const ConvLayer& L, const act_t* dout, void* din, int B, int Hi, int Wi, const Epi& e, hipStream_t st) {
  if (!L.wb) { dmx_set_error("layer has no dgrad weights"); return DMX_ERR_STATE; }
  if (L.stride != 1) return DMX_ERR_SHAPE;
  const int Ho = Hi + 2 * L.pad_h - L.kh + 1, Wo = Wi + 2 * L.pad_w - L.kw + 1;
  GemmDesc d;
  init_desc(d, e);
  d.A = dout; d.Hi = Ho; d.Wi = Wo; d.Ci = L.Cop; d.lda = L.Cop;
  d.W = L.wb; d.ntaps = L.kh * L.kw; d.K = d.ntaps * L.Cop; d.ldw = d.K;
  d.N = L.Cip; d.Hq = Hi; d.Wq = Wi; d.M = B * Hi * Wi;
  set_out(d, din, Hi, Wi, L.Cip);
  for (int ky = 0; ky < L.kh; ++ky)
    for (int kx = 0; kx < L.kw; ++kx) {
      d.tdy[ky * L.kw + kx] = (signed char)(L.pad_h - ky);
      d.tdx[ky * L.kw + kx] = (signed char)(L.pad_w - kx);
    }
  return dmx_gemm_launch(d, st);
}

int gemm_nt(const act_t* A, int lda, const act_t* Bm, int ldb, void* C, int ldc, int M, int N, int K, const Epi& e,
            const GemmBatch& gb, hipStream_t st) {
  GemmDesc d;
  init_desc(d, e);
  d.A = A; d.Hi = 1; d.Wi = M; d.Ci = K; d.lda = lda;
  d.W = Bm; d.ntaps = 1; d.K = K; d.ldw = ldb;
  d.N = N; d.Hq = 1; d.Wq = M; d.M = M;
  set_out(d, C, 1, M, ldc);
  d.Z = gb.Z; d.Zi = gb.Zi;
  d.sAo = gb.sAo; d.sAi = gb.sAi; d.sWo = gb.sBo; d.sWi = gb.sBi; d.sCo = gb.sCo; d.sCi = gb.sCi;
  return dmx_gemm_launch(d, st);
}

int linear_fwd(const ConvLayer& L, const act_t* in, int lda, void* out, int ldc, long long rows, const Epi& e, hipStream_t st) {
  Epi ee = e;
  GemmDesc d;
  init_desc(d, ee);
  if (L.has_bias) { d.bias = L.bias; d.flags |= EPI_BIAS; }
  d.A = in; d.Hi = 1; d.Wi = (int)rows; d.Ci = L.Cip; d.lda = lda;
  d.W = L.wf[0]; d.ntaps = 1; d.K = L.Cip; d.ldw = L.Cip;
  d.N = L.Cop; d.Hq = 1; d.Wq = (int)rows; d.M = (int)rows;
  set_out(d, out, 1, (int)rows, ldc);
  if (L.geglu) d.flags |= EPI_GEGLU;          // out is (rows, Cop / 2): value * gelu(gate), applied in the epilogue
  if (L.colsum) {            // `in` holds the RAW rows: LayerNorm happens inside, from the row statistics the producer of `in` wrote
    if (!e.rowstats_in || e.nslots * 32 != L.Cip) { dmx_set_error("LayerNorm-folded layer needs the row statistics of its input"); return DMX_ERR_STATE; }
    d.flags |= EPI_LNFOLD; d.colsum = L.colsum; d.ln_eps = L.ln_eps;
    d.flags &= ~EPI_BIAS;          // the folded bias (d.bias stays set) is added together with the LayerNorm correction, not by the epilogue
  }
  return dmx_gemm_launch(d, st);
}
int linear_bwd(const ConvLayer& L, const act_t* dout, int lda, void* din, int ldc, long long rows, const Epi& e, hipStream_t st) {
  if (!L.wb) { dmx_set_error("layer has no dgrad weights"); return DMX_ERR_STATE; }
  GemmDesc d;
  init_desc(d, e);
  d.A = dout; d.Hi = 1; d.Wi = (int)rows; d.Ci = L.Cop; d.lda = lda;
  d.W = L.wb; d.ntaps = 1; d.K = L.Cop; d.ldw = L.Cop;
  d.N = L.Cip; d.Hq = 1; d.Wq = (int)rows; d.M = (int)rows;
  set_out(d, din, 1, (int)rows, ldc);
  return dmx_gemm_launch(d, st);
}
