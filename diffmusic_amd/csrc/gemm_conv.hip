// Implicit-GEMM convolution / batched NT-GEMM on gfx950 MFMA (fp16 in -- bf16 with -DDMX_BF16 --, fp32 accumulate).
//
// One kernel family serves every dense contraction on the hot path ("the
// kernel family is closed under dgrad"): conv1d (dilated), ConvTranspose1d (one launch per output
// phase), strided conv1d (ConvTranspose dgrad), conv2d 3x3/1x1/stride-2, linear layers and the
// attention products.  Activations are channels-last 16-bit so the K axis (tap, cin) is contiguous
// per tap; the A tile is gathered with zero fill at the borders into an XOR-swizzled LDS image (LDS-DMA
// in gemm_glds_kernel, register-staged in gemm_kernel) and consumed by v_mfma_f32_16x16x32_f16 with the
// weight as the MFMA A-operand, so each lane ends up with 4 consecutive output channels of one output pixel.  Everything pointwise around a contraction (bias, time-embedding row bias,
// residual, resblock averaging, leaky-relu of the stored activation, leaky-relu' mask for dgrad,
// tanh) is fused into the epilogue.
#include "gemm_tile.h"

namespace {
// the sign-bit tape epilogue (HiFi-GAN layers) is a separate instantiation of every tile
template <int BM, int BN, int WM, int WN, int NSTAGE>
int launch_glds(const GemmDesc& d, hipStream_t stream) {
  if (d.flags & EPI_SOFTBWD) {      // the fused softmax backward exists for the tile its one caller (VAE mid attention, N x N scores) gets
    if constexpr (BM == 256 && BN == 256) return launch_glds_t<BM, BN, WM, WN, NSTAGE, 2>(d, stream);
    else return DMX_ERR_SHAPE;
  }
  if (d.flags & EPI_GEGLU) return launch_glds_t<BM, BN, WM, WN, NSTAGE, 3>(d, stream);
  return (d.flags & (EPI_MASKBITS | EPI_BITS2)) ? launch_glds_t<BM, BN, WM, WN, NSTAGE, 1>(d, stream)
                                                : launch_glds_t<BM, BN, WM, WN, NSTAGE, 0>(d, stream);
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmDesc& d, hipStream_t stream) {
  if (d.flags & EPI_SOFTBWD) return DMX_ERR_SHAPE;
  if (d.flags & EPI_GEGLU) {
    if constexpr ((BN / WN) % 32 == 0) return launch_cfg_t<BM, BN, WM, WN, 3>(d, stream);
    else return DMX_ERR_SHAPE;
  }
  return (d.flags & (EPI_MASKBITS | EPI_BITS2)) ? launch_cfg_t<BM, BN, WM, WN, 1>(d, stream) : launch_cfg_t<BM, BN, WM, WN, 0>(d, stream);
}

}  // namespace

// ---- optional per-launch profiling (bench.py roofline leg): HIP events around every GEMM launch on its stream
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
namespace {
struct ProfRec { hipEvent_t a, b; double flops, bytes; int M, N, K, Z, taps, flags, cfg; };
int g_last_cfg = 0;   // tile configuration chosen by the most recent dispatch (profiling only)
bool g_prof = false;
std::vector<ProfRec> g_prof_recs;
// tile configurations: 1 = LDS-DMA 256x256, 2 = LDS-DMA 256x128, 3 = 128x128, 4 = 128x64, 5 = 128x32, 6 = 64x64
struct TileEntry { int M, N, K, Z, cfg; };
static const TileEntry g_tile_table[] = {
#include "tile_table.inc"
    {0, 0, 0, 0, 0}};

bool glds_ok(const GemmDesc& d) {
  // operand spans must stay below 2 GiB for the 32-bit buffer offsets of the LDS-DMA kernels
  // (input span: images x Hi x Wi pixels -- more than the M output rows when the walk over the input is strided)
  const long long imgs = d.Hq * d.Wq > 0 ? (d.M + (long long)d.Hq * d.Wq - 1) / ((long long)d.Hq * d.Wq) : 1;
  const long long in_elems = (d.Hi > 1 || d.ntaps > 1 ? imgs * d.Hi * d.Wi : (long long)d.M) * d.lda;
  static const bool strided_ok = getenv("DMX_NO_GLDS_STRIDED") == nullptr;
  return in_elems < (1ll << 29) && (long long)d.M * d.lda < (1ll << 29) && ((long long)d.N + 512) * d.ldw < (1ll << 29) &&
         (d.sy == 1 || strided_ok);
}
int launch_by_cfg(int cfg, const GemmDesc& d, hipStream_t stream) {
  g_last_cfg = cfg;
  if (d.flags & EPI_LNFOLD) return dmx_gemm_launch_ln(cfg, d, stream);       // (gemm_ln.hip: the same tiles with the LayerNorm correction ahead of the epilogue)
  if (d.flags & EPI_ROWSTATS) return dmx_gemm_launch_rowstats(cfg, d, stream);   // (... and with the row-statistics epilogue)
  if (d.flags & (EPI_GNSTATS | EPI_GNBWD)) return dmx_gemm_launch_gnstats(cfg, d, stream);     // (gemm_gn.hip: GroupNorm partial sums of the stored tile)
  switch (cfg) {
    case 1: return launch_glds<256, 256, 2, 4, 2>(d, stream);
    case 2: return launch_glds<256, 128, 4, 2, 3>(d, stream);
    case 7: return launch_glds<320, 256, 2, 4, 2>(d, stream);
    case 8: return launch_glds<192, 256, 2, 4, 2>(d, stream);
    case 9: return launch_glds<320, 128, 4, 2, 2>(d, stream);
    case 10: return launch_glds<192, 128, 4, 2, 3>(d, stream);
    case 11: return launch_glds<128, 128, 2, 2, 2>(d, stream);   // 4-wave LDS-DMA tiles for the small-M U-Net / VAE-mid GEMMs
    case 12: return launch_glds<64, 64, 2, 2, 4>(d, stream);
    case 13: return launch_glds<128, 64, 2, 2, 3>(d, stream);
    case 14: return launch_glds<64, 128, 2, 2, 3>(d, stream);
    // deep rings for the latency-bound small-M launches (one workgroup per CU, the whole K panel of a short GEMM in flight at once)
    case 15: return launch_glds<64, 64, 2, 2, 8>(d, stream);
    case 16: return launch_glds<64, 128, 2, 2, 6>(d, stream);
    case 17: return launch_glds<128, 64, 2, 2, 6>(d, stream);
    case 18: return launch_glds<128, 128, 2, 2, 4>(d, stream);
    // N = 128 layers at full resolution (VAE 128-channel 3x3 convolutions, M = 512 000): the 256 x 256 tile's wave layout (8 waves of
    // 128 x 64) on a 512 x 128 block -- twice the MFMAs per barrier and per weight fetch of the 256 x 128 tile; its ring is all 160 KiB
    case 19: return launch_glds<512, 128, 4, 2, 2>(d, stream);
    case 3: return launch_cfg<128, 128, 2, 2>(d, stream);
    case 4: return launch_cfg<128, 64, 2, 2>(d, stream);
    case 5: return launch_cfg<128, 32, 4, 1>(d, stream);
    default: return launch_cfg<64, 64, 2, 2>(d, stream);
  }
}
int launch_dispatch(const GemmDesc& d, hipStream_t stream) {
  const bool gl = glds_ok(d);
  if (d.flags & EPI_SOFTBWD) return gl ? launch_by_cfg(1, d, stream) : DMX_ERR_SHAPE;   // (its one caller checks the span up front: vae.hip)
  if (d.tile_cfg >= 1 && d.tile_cfg <= 19 && ((d.tile_cfg > 2 && d.tile_cfg < 7) || gl)) return launch_by_cfg(d.tile_cfg, d, stream);
  {  // tuning hook: DMX_CFG_OVERRIDE="N:cfg,N:cfg" forces a tile configuration for large-M launches with that N
    static int ovN[8], ovC[8], nov = -1;
    if (nov < 0) {
      nov = 0;
      if (const char* e = getenv("DMX_CFG_OVERRIDE")) {
        while (*e && nov < 8) {
          int n = 0, c = 0;
          if (sscanf(e, "%d:%d", &n, &c) == 2) { ovN[nov] = n; ovC[nov] = c; ++nov; }
          while (*e && *e != ',') ++e;
          if (*e == ',') ++e;
        }
      }
    }
    if (gl && d.M >= 40000)
      for (int i = 0; i < nov; ++i) if (ovN[i] == d.N) return launch_by_cfg(ovC[i], d, stream);
  }
  // measured best configuration for the shapes of the shipped benchmark configs (scripts/dev/tune_tiles.py)
  for (const TileEntry* e = g_tile_table; e->cfg; ++e)
    if (e->M == d.M && e->N == d.N && e->K == d.K && e->Z == d.Z) {
      const int c = e->cfg % 100;                  // (hundreds = a split-K plan, taken by splitk_plan when the launch allows it)
      if ((c > 2 && c < 7) || gl) return launch_by_cfg(c, d, stream);
    }
  if (gl && d.M >= 2048 && d.N % 128 == 0) {
    // otherwise pick the tile that minimises (rounds over the 256 CUs) x (time per block); efficiencies measured on MI355X
    auto cost = [&](int bm, int bn, int slots, double eff) {
      const double blocks = (double)cdiv(d.M, bm) * cdiv(d.N, bn) * d.Z;
      const double rounds = ceil(blocks / slots);
      return rounds * (bm / 128.0) * (bn / 128.0) * (slots / 256.0) / eff;
    };
    // candidates: LDS-DMA tiles with 192 / 256 / 320 rows (the row count is chosen to fill whole rounds of 256 CUs:
    // M = 40 008 x N = 512 is 314 tiles of 256x256 = 2 rounds at 61 %, but 252 tiles of 320x256 = 1 round at 98 %)
    struct Cand { int cfg, bm, bn, slots; double eff; };
    static const Cand cands[] = {{1, 256, 256, 256, 0.95}, {7, 320, 256, 256, 0.93}, {8, 192, 256, 256, 0.88},
                                 {2, 256, 128, 256, 0.70}, {9, 320, 128, 256, 0.70}, {10, 192, 128, 256, 0.66},
                                 // 512 x 128: same-device A/B on the VAE's M = 512 000, N = 128 convolutions: VAE forward + backward -0.25 ms
                                 // against 256 x 128; slower on the U-Net's M = 64 000 (125 tiles for 256 CUs), which the rounds term excludes
                                 {19, 512, 128, 256, 0.75},
                                 {3, 128, 128, 512, 0.66}};
    int best = 0;
    double bc = 1e30;
    for (const Cand& c : cands) {
      if (c.bn == 256 && d.N % 256) continue;
      const double v = cost(c.bm, c.bn, c.slots, c.eff);
      if (v < bc) { bc = v; best = c.cfg; }
    }
    if (best && best != 3) return launch_by_cfg(best, d, stream);
  }
  // small problems: 64x64 tiles so that at least ~1 block per CU exists (U-Net levels with 1k-4k pixels)
  if (d.N > 32 && (long long)cdiv(d.M, 128) * cdiv(d.N, 128) * d.Z < 200) {
    // shapes the measured table does not know (other batch sizes / clip lengths): what the tuner found on ~150 of them -- the 4-stage LDS-DMA
    // 64-row tiles win from K >= 512 on (64x128 when N allows it and there are rows enough), the register-staged tile below that
    if (gl && d.K >= 512 && d.Z == 1) return launch_by_cfg(d.N % 128 == 0 && d.M >= 4000 ? 14 : 12, d, stream);
    return launch_by_cfg(6, d, stream);
  }
  if (d.N > 64) return launch_by_cfg(3, d, stream);
  if (d.N > 32) return launch_by_cfg(4, d, stream);
  return launch_by_cfg(5, d, stream);
}
}  // namespace

bool dmx_prof_is_active() { return g_prof; }

extern "C" void dmx_prof_begin(void) {
  for (auto& r : g_prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof_recs.clear();
  g_prof = true;
}
// stops recording; returns the number of launches, total kernel milliseconds and algorithmic FLOPs (2*M*N*K*Z)
static double g_dma_ms = 0.0, g_dma_fl = 0.0, g_dma_by = 0.0;
static int g_dma_n = 0;
#ifdef DMX_GEMM_STAMPS
extern "C" int dmx_gemm_stamps_read(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 8192 * 6) == hipSuccess ? 0 : -1;
}
#endif

// share of the last profiled region that ran in gemm_glds_kernel (the dominant, MFMA-bound kernel)
extern "C" int dmx_prof_dominant(double* ms, double* flops, double* bytes) {
  if (ms) *ms = g_dma_ms;
  if (flops) *flops = g_dma_fl;
  if (bytes) *bytes = g_dma_by;
  return g_dma_n;
}
extern "C" int dmx_prof_end(double* total_ms, double* total_flops) {
  g_prof = false;
  g_dma_ms = g_dma_fl = g_dma_by = 0.0; g_dma_n = 0;
  (void)hipDeviceSynchronize();
  double ms = 0.0, fl = 0.0;
  FILE* csv = getenv("DMX_PROF_CSV") ? fopen(getenv("DMX_PROF_CSV"), "w") : nullptr;
  if (csv) fprintf(csv, "M,N,K,Z,taps,flags,cfg,ms,tflops\n");
  for (auto& r : g_prof_recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) ms += t;
    if (csv) fprintf(csv, "%d,%d,%d,%d,%d,%d,%d,%.4f,%.1f\n", r.M, r.N, r.K, r.Z, r.taps, r.flags, r.cfg, t, t > 0 ? r.flops / t / 1e9 : 0.0);
    const bool dma = r.cfg == 1 || r.cfg == 2 || (r.cfg >= 7 && r.cfg <= 10) || r.cfg == 19;   // gemm_glds_kernel, 8-wave tiles (192 ... 512 rows)
    if (dma) { g_dma_ms += t; g_dma_fl += r.flops; g_dma_by += r.bytes; ++g_dma_n; }
    fl += r.flops;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  if (csv) fclose(csv);
  const int n = (int)g_prof_recs.size();
  g_prof_recs.clear();
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  return n;
}

// ---- split-K for small-M / deep-K convolutions (the low-resolution U-Net levels: M = 1024, K = 5760 is 160 tiles of 90
// serial K steps): the K range is cut into `ksplit` slices that run as extra workgroups (blockIdx.z) and write fp32 partial
// tiles; this kernel adds the slices and applies the whole fused epilogue.
namespace {
float* g_splitk_ws = nullptr;
size_t g_splitk_bytes = 0;

__global__ void splitk_epilogue_kernel(const GemmDesc p, const float* __restrict__ ws, int ksplit) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 4 consecutive channels
  const int nq = p.N >> 2;
  if (q >= (long long)p.M * nq) return;
  const int m = (int)(q / nq), n = (int)(q - (long long)m * nq) * 4;
  const long long mn = (long long)p.M * p.N;
  float4 v = *reinterpret_cast<const float4*>(ws + (long long)m * p.N + n);
  for (int s = 1; s < ksplit; ++s) {
    const float4 w = *reinterpret_cast<const float4*>(ws + s * mn + (long long)m * p.N + n);
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  }
  float a[4] = {v.x, v.y, v.z, v.w};
  const int flags = p.flags;
  const long long row = m;                                                   // identity row map (checked by the dispatcher)
  if (flags & EPI_MASK) {
    const uint2 xr = *reinterpret_cast<const uint2*>(p.X + row * p.ldx + n);
    const float x[4] = {alo(xr.x), ahi(xr.x), alo(xr.y), ahi(xr.y)};
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] *= x[e] > 0.f ? 1.f : p.mask_slope;
  }
  if (flags & EPI_BIAS) {
    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
    a[0] += bb.x; a[1] += bb.y; a[2] += bb.z; a[3] += bb.w;
  }
  if (flags & EPI_ROWBIAS) {
    const float4 bb = *reinterpret_cast<const float4*>(p.rowbias + (long long)(m / (p.Hq * p.Wq)) * (p.ldrb ? p.ldrb : p.N) + n);
    a[0] += bb.x; a[1] += bb.y; a[2] += bb.z; a[3] += bb.w;
  }
  if (flags & EPI_RESID) {
    const uint2 rr = *reinterpret_cast<const uint2*>(p.R + row * p.ldr + n);
    const float is = (flags & EPI_RESID_INV) ? p.resid_inv_slope : 1.f;
    const float x[4] = {alo(rr.x), ahi(rr.x), alo(rr.y), ahi(rr.y)};
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] += fminf(x[e], x[e] * is);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) a[e] *= p.alpha;
  if (!(flags & EPI_NO_C))
    *reinterpret_cast<uint2*>(reinterpret_cast<act_t*>(p.C) + row * p.ldc + n) = make_uint2(pack2a(a[0], a[1]), pack2a(a[2], a[3]));
  if (flags & EPI_LRELU2) {
    const float sl = p.act_slope;
    *reinterpret_cast<uint2*>(p.C2 + row * p.ldc2 + n) =
        make_uint2(pack2a(fmaxf(a[0], a[0] * sl), fmaxf(a[1], a[1] * sl)), pack2a(fmaxf(a[2], a[2] * sl), fmaxf(a[3], a[3] * sl)));
  }
}

// returns the number of K slices to use for `d` (1 = no split) and the tile configuration of the slices.  A plan comes from
// (i) the tuning hook tile_cfg = 100 * slices + tile, (ii) the measured table (same encoding), (iii) the heuristic below.
int splitk_plan(const GemmDesc& d, int* tile) {
  static const bool off = getenv("DMX_NO_SPLITK") != nullptr;
  if (off || !g_splitk_ws || d.Z != 1 || !glds_ok(d)) return 1;
  if (d.flags & (EPI_ACCUM | EPI_F32OUT | EPI_TANH | EPI_MASKBITS | EPI_BITS2 | EPI_SOFTBWD | EPI_GEGLU | EPI_LNFOLD | EPI_ROWSTATS)) return 1;
  if (!(d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0 && d.Ho == d.Hq && d.Wo == d.Wq)) return 1;
  if ((d.N & 7) || (d.ldc & 3)) return 1;
  const int nk = (d.K + BK - 1) / BK;
  auto fits = [&](int ks) { return ks >= 2 && ks <= nk && (size_t)ks * d.M * d.N * sizeof(float) <= g_splitk_bytes; };
  auto dma_tile = [](int t) { return t == 1 || t == 2 || (t >= 7 && t <= 19); };      // only the LDS-DMA kernel walks a K slice
  if (d.tile_cfg >= 100) { *tile = d.tile_cfg % 100; return fits(d.tile_cfg / 100) && dma_tile(*tile) ? d.tile_cfg / 100 : 1; }
  if (d.tile_cfg) return 1;
  for (const TileEntry* e = g_tile_table; e->cfg; ++e)
    if (e->M == d.M && e->N == d.N && e->K == d.K && e->Z == d.Z) {
      if (e->cfg < 100) return 1;
      *tile = e->cfg % 100;
      return fits(e->cfg / 100) && dma_tile(*tile) ? e->cfg / 100 : 1;
    }
  if (d.M > 8192) return 1;
  const long long tiles = (long long)cdiv(d.M, 64) * cdiv(d.N, 64);
  if (nk < 24 || tiles >= 256) return 1;        // only when the tile grid cannot fill the 256 CUs (measured: 378 tiles x 2 slices is slower)
  int ks = (int)(1024 / tiles);                 // aim at ~4 workgroups per CU
  if (ks > nk / 8) ks = nk / 8;                 // at least 8 K steps per slice
  if (ks > 8) ks = 8;
  while (ks > 1 && !fits(ks)) --ks;
  *tile = d.N % 128 == 0 && d.M >= 4096 ? 14 : 12;
  return ks < 2 ? 1 : ks;
}

int launch_splitk(const GemmDesc& d, int ks, int tile, hipStream_t stream) {
  GemmDesc g = d;
  g.ksplit = ks;
  g.flags = EPI_F32OUT;
  g.C = g_splitk_ws; g.C2 = nullptr; g.ldc = d.N; g.alpha = 1.f;
  g.bias = nullptr; g.rowbias = nullptr; g.R = nullptr; g.X = nullptr;
  g.Ho = d.Hq; g.Wo = d.Wq;
  g.tile_cfg = 0;
  int rc = launch_by_cfg(tile, g, stream);
  if (rc != DMX_OK) return rc;
  const long long nthr = (long long)d.M * (d.N >> 2);
  hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, d, g_splitk_ws, ks);
  g_last_cfg = 100 * ks + tile;
  return hipGetLastError() == hipSuccess ? DMX_OK : DMX_ERR_LAUNCH;
}
}  // namespace

// fp32 scratch for split-K partial tiles (owned by the caller, e.g. a model; stream-ordered use only)
void dmx_gemm_set_splitk_workspace(float* ws, size_t bytes) { g_splitk_ws = ws; g_splitk_bytes = bytes; }
// the owner is going away: forget the scratch if it is the one installed (a dangling pointer here would be written by the
// next eligible launch of any model)
void dmx_gemm_release_splitk_workspace(const float* ws) { if (g_splitk_ws == ws) { g_splitk_ws = nullptr; g_splitk_bytes = 0; } }

int dmx_prof_open(hipStream_t st) {
  if (!g_prof) return -1;
  ProfRec r;
  memset(&r, 0, sizeof(r));
  (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
  (void)hipEventRecord(r.a, st);
  g_prof_recs.push_back(r);
  return (int)g_prof_recs.size() - 1;
}
void dmx_prof_close(int rec, hipStream_t st, double flops, double bytes, int M, int N, int K, int taps, int flags, int cfg) {
  if (rec < 0 || rec >= (int)g_prof_recs.size()) return;
  ProfRec& r = g_prof_recs[rec];
  (void)hipEventRecord(r.b, st);
  r.flops = flops; r.bytes = bytes; r.M = M; r.N = N; r.K = K; r.Z = 1; r.taps = taps; r.flags = flags; r.cfg = cfg;
}

int dmx_gemm_launch(const GemmDesc& d, hipStream_t stream) {
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return DMX_ERR_SHAPE;
  if ((d.Ci & 7) || (d.N & 3) || (d.K & 7) || (d.lda & 7) || (d.ldw & 7) || (d.ldc & 3)) return DMX_ERR_SHAPE;
  if (d.ntaps < 1 || d.ntaps > DMX_MAX_TAPS || d.K != d.ntaps * d.Ci) return DMX_ERR_SHAPE;
  if (d.Z < 1 || d.Zi < 1) return DMX_ERR_SHAPE;
  // the LDS epilogue evaluates leaky-relu as max(v, v*slope) and the inverse as min(x, x/slope)
  if ((d.flags & EPI_LRELU2) && !(d.act_slope >= 0.f && d.act_slope <= 1.f)) return DMX_ERR_SHAPE;
  if ((d.flags & EPI_RESID_INV) && !(d.resid_inv_slope >= 1.f)) return DMX_ERR_SHAPE;
  if (d.flags & EPI_SOFTBWD) {
    // fused softmax backward: 16-bit output through the LDS-staged epilogue, P tile as X, per-row delta in rowbias, Zi == 1
    if ((d.flags & ~(EPI_SOFTBWD)) || ((d.N | d.ldc | d.ldx) & 7) || !d.X || !d.rowbias || d.Zi != 1 || d.ntaps != 1) return DMX_ERR_SHAPE;
    if (!(d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0 && d.Ho == d.Hq && d.Wo == d.Wq)) return DMX_ERR_SHAPE;
  }
  if (d.flags & EPI_GEGLU) {
    // fused GEGLU: bias only, 16-bit half-width output through the LDS-staged epilogue, interleaved blocks of 32 weight rows
    if ((d.flags & ~(EPI_GEGLU | EPI_BIAS | EPI_LNFOLD)) || (d.N & 31) || (d.ldc & 7) || d.ldc < d.N / 2 || d.alpha != 1.f) return DMX_ERR_SHAPE;
    if (!(d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0 && d.Ho == d.Hq && d.Wo == d.Wq)) return DMX_ERR_SHAPE;
  }
  if (d.flags & (EPI_MASKBITS | EPI_BITS2)) {
    // sign-bit tensors are handled by the LDS-staged epilogue only: 16-bit output, 16-byte granular rows, one batch
    if ((d.flags & EPI_F32OUT) || ((d.N | d.ldc | d.ldr | d.ldx | d.ldc2) & 7) || d.Z != 1) return DMX_ERR_SHAPE;
    if ((d.flags & EPI_MASKBITS) && (!d.XB || d.ldxb * 8 < d.N)) return DMX_ERR_SHAPE;
    if ((d.flags & EPI_BITS2) && (!d.B2 || d.ldb2 * 8 < d.N)) return DMX_ERR_SHAPE;
  }
  if ((d.flags & EPI_LNFOLD) && (d.flags & EPI_BIAS)) {
    // with a folded LayerNorm the bias (d.bias, folded: b + W beta) is added together with the LayerNorm correction ahead of the epilogue
    GemmDesc q = d;
    q.flags &= ~EPI_BIAS;
    return dmx_gemm_launch(q, stream);
  }
  if (d.flags & EPI_LNFOLD) {
    // LayerNorm fold: single-tap projection over the whole normalised width, row statistics from the producer of A
    if (d.ntaps != 1 || d.K != d.Ci || !d.colsum || !d.rowstats_in || d.nslots < 1 || d.Z != 1 ||
        (d.flags & (EPI_F32OUT | EPI_MASK | EPI_MASKBITS | EPI_BITS2 | EPI_SOFTBWD | EPI_ACCUM | EPI_TANH | EPI_LRELU2 | EPI_ROWSTATS)))
      return DMX_ERR_SHAPE;
  }
  if (d.flags & EPI_ROWSTATS) {
    // row statistics: 16-bit output through the LDS-staged epilogue, identity row map, whole 32-column slots
    if (!d.rowstats_out || d.nslots * 32 != d.N || d.Z != 1 || ((d.ldc | d.ldr) & 7) ||
        (d.flags & (EPI_F32OUT | EPI_MASK | EPI_MASKBITS | EPI_BITS2 | EPI_SOFTBWD | EPI_ACCUM | EPI_TANH | EPI_LRELU2 | EPI_GEGLU | EPI_NO_C)))
      return DMX_ERR_SHAPE;
    if (!(d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0 && d.Ho == d.Hq && d.Wo == d.Wq)) return DMX_ERR_SHAPE;
  }
  dmx_gemm_reset_last_tile_rows();
  if (d.flags & (EPI_GNSTATS | EPI_GNBWD)) {
    // GroupNorm partial sums ride on the LDS-staged 16-bit epilogue of an unsplit launch; anything else launches WITHOUT them and reports
    // 0 tile rows, so that the caller's GroupNorm takes its own statistics pass
    bool can = d.gn_part && d.Z == 1 && !(d.flags & (EPI_F32OUT | EPI_NO_C | EPI_GEGLU | EPI_SOFTBWD | EPI_MASKBITS | EPI_BITS2 | EPI_LNFOLD | EPI_ROWSTATS | EPI_LRELU2)) &&
               !((d.N | d.ldc | d.ldr | d.ldx | d.ldc2) & 7) && (d.flags & (EPI_GNSTATS | EPI_GNBWD)) != (EPI_GNSTATS | EPI_GNBWD);
    if (d.flags & EPI_GNBWD) can = can && d.gnb_x && d.gnb_scale && d.gnb_shift && d.gnb_stats && !(d.gnb_ldx & 7) && d.gnb_cpg >= 4 && !(d.gnb_cpg & 3) && d.N % d.gnb_cpg == 0;
    int kt = 12;
    if (!can || splitk_plan(d, &kt) > 1) { GemmDesc q = d; q.flags &= ~(EPI_GNSTATS | EPI_GNBWD); return dmx_gemm_launch(q, stream); }
  }
  int ktile = 12;
  const int ksp = splitk_plan(d, &ktile);
  if (ksp <= 1 && d.tile_cfg >= 100) return DMX_ERR_SHAPE;          // a forced split-K plan this launch cannot take (tuning hook)
  if (!g_prof) return ksp > 1 ? launch_splitk(d, ksp, ktile, stream) : launch_dispatch(d, stream);
  ProfRec r;
  (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
  r.flops = 2.0 * d.M * (double)d.N * d.K * d.Z;
  {  // algorithmic HBM bytes: every operand touched once (input positions x Cin, packed weights, outputs + fused side tensors)
    const double mn = (double)d.M * d.N * d.Z;
    double by = 2.0 * d.M * (double)d.Ci * d.Z + 2.0 * d.N * (double)d.K * (d.sWo ? d.Z : 1);
    if (!(d.flags & EPI_NO_C)) by += mn * ((d.flags & EPI_F32OUT) ? 4.0 : 2.0);
    if (d.flags & EPI_ACCUM) by += mn * ((d.flags & EPI_F32OUT) ? 4.0 : 2.0);
    if (d.flags & (EPI_RESID | EPI_RESID_INV)) by += mn * 2.0;
    if (d.flags & (EPI_MASK | EPI_SOFTBWD)) by += mn * 2.0;
    if (d.flags & EPI_MASKBITS) by += mn / 8.0;
    if (d.flags & EPI_BITS2) by += mn / 8.0;
    if (d.C2 && (d.flags & EPI_LRELU2)) by += mn * 2.0;
    r.bytes = by;
  }
  r.M = d.M; r.N = d.N; r.K = d.K; r.Z = d.Z; r.taps = d.ntaps; r.flags = d.flags;
  (void)hipEventRecord(r.a, stream);
  const int rc = ksp > 1 ? launch_splitk(d, ksp, ktile, stream) : launch_dispatch(d, stream);
  (void)hipEventRecord(r.b, stream);
  r.cfg = g_last_cfg;
  g_prof_recs.push_back(r);
  return rc;
}
