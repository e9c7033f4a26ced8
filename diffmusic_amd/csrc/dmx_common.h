// Common device/host helpers for the DiffMusic MI355X (gfx950) hot-path library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// 16-bit activation type: raw bits of fp16 (default) or bf16 (-DDMX_BF16); all activation tensors are
// channels-last.  fp16 matches the reference's own torch_dtype=float16 (run.py:218) and has 8x less rounding
// noise than bf16, which matters for the leaky-relu' masks of near-zero units in the guidance gradient.
typedef uint16_t act_t;

#define DMX_OK 0
#define DMX_ERR_SHAPE (-1)
#define DMX_ERR_WORKSPACE (-2)
#define DMX_ERR_PARAM (-3)
#define DMX_ERR_STATE (-4)
#define DMX_ERR_LAUNCH (-5)

typedef __attribute__((ext_vector_type(4))) float f32x4;
#ifdef DMX_BF16
typedef __bf16 native16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 frag8_t;
#define DMX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define DMX_ACT_DTYPE 0
#else
typedef _Float16 native16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 frag8_t;
#define DMX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define DMX_ACT_DTYPE 1
#endif

__device__ __forceinline__ float a2f(act_t v) { return (float)__builtin_bit_cast(native16_t, v); }
__device__ __forceinline__ act_t f2a(float f) { return __builtin_bit_cast(act_t, (native16_t)f); }
// two floats -> one packed pair of 16-bit activations (round to nearest even).  Written as a 2-vector conversion so that it becomes ONE
// v_cvt_pk_f16_f32: as (f2a(lo) | f2a(hi) << 16) a 4-element pack was compiled into conversions of the (0, 2) and (1, 3) pairs plus four
// and / shift / or instructions to re-interleave them (GEMM epilogue staging loop: -0.3 ... -0.7 us per 320 x 256 tile, in-kernel stamps)
#ifdef DMX_BF16
__device__ __forceinline__ uint32_t pack2a(float lo, float hi) { return (uint32_t)f2a(lo) | ((uint32_t)f2a(hi) << 16); }
#else
__device__ __forceinline__ uint32_t pack2a(float lo, float hi) {
  typedef float dmx_f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 dmx_f16x2 __attribute__((ext_vector_type(2)));
  const dmx_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dmx_f16x2));
}
#endif
__device__ __forceinline__ float alo(uint32_t u) { return a2f((act_t)(u & 0xffffu)); }
__device__ __forceinline__ float ahi(uint32_t u) { return a2f((act_t)(u >> 16)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x multiple of 64 (<=1024); `sh` needs 16 floats
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = -3.0e38f;
  for (int i = 0; i < nw; ++i) r = fmaxf(r, sh[i]);
  return r;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM descriptor.  C[m, n] = sum_k A_gather[m, k] * W[n, k]
//   rows m enumerate (b, qy, qx) over (B, Hq, Wq); k enumerates (tap, cin) tap-major.
//   input pixel  : iy = qy*sy + tdy[tap], ix = qx*sx + tdx[tap]  (zero outside [0,Hi)x[0,Wi))
//   output pixel : oy = qy*osy + ooy,     ox = qx*osx + oox
// Plain / batched GEMMs use ntaps=1, Hq=1, Wq=M and the z strides.
#define DMX_MAX_TAPS 16
enum {
  EPI_BIAS = 1,        // + bias[n]                     (fp32)
  EPI_ROWBIAS = 2,     // + rowbias[b, n]               (fp32, ld = N) per batch image
  EPI_RESID = 4,       // + R[row, n]                   (bf16, same indexing as C, ld = ldr)
  EPI_ACCUM = 8,       // + previous C[row, n]          (after alpha)
  EPI_MASK = 16,       // * (X[row, n] > 0 ? 1 : mask_slope)   applied to acc before resid
  EPI_LRELU2 = 32,     // second output C2 = leaky_relu(v, act_slope)
  EPI_TANH = 64,       // v = tanh(v)
  EPI_F32OUT = 128,    // C is float* instead of bf16*
  EPI_NO_C = 256,      // skip the primary output (only C2)
  EPI_RESID_INV = 512, // R holds leaky_relu(x, 1/resid_inv_slope): the residual added is the reconstructed x
  EPI_MASKBITS = 1024, // like EPI_MASK, but the mask source is a SIGN-BIT tensor XB: byte [row][n / 8], bit (n & 7) set <=> x > 0
  EPI_BITS2 = 2048,    // also write the sign bits (v > 0) of the stored value to B2 (same layout): the leaky-relu' mask of the
                       // backward sweep at 1/16 of the bytes of the 16-bit tensor (HiFi-GAN tape)
  EPI_GEGLU = 8192,    // GEGLU of a feed-forward's first projection fused into its epilogue: the weight rows are packed in blocks of
                       // 32 = [16 value rows | their 16 gate rows], so accumulator fragments 2t / 2t + 1 of a wave hold value and gate of
                       // the same 16 channels: out[row, 16 t' + c] = (v + bias) * gelu_erf(g + bias), N / 2 output columns (ld = ldc)
  EPI_LNFOLD = 16384,  // LayerNorm of the input rows folded into this projection (single tap, K = the normalised width): mean / rstd of every
                       // row come from the partial sums its producer wrote (rowstats_in, EPI_ROWSTATS) and the accumulators become
                       // rstd * (acc - mean * colsum[n]) before the rest of the epilogue; W carries gamma, bias carries W beta (pack_layer)
  EPI_ROWSTATS = 32768, // also write per-row partial sums (sum v, sum v^2 per 32-column slot, fp32) of the final values to rowstats_out
                        // [row][nslots][2]: the statistics of a LayerNorm folded into the next projection (identity row map, N % 32 == 0)
  EPI_GNSTATS = 65536,  // also write GroupNorm partial sums of the stored output to gn_part: (sum v, sum v^2) per 4-channel quad, per wave tile and
                        // image -- [image][slot][N / 4][2] fp32 with slot = wave-tile index inside the image (rows per tile = the tile's TM,
                        // reported by dmx_gemm_last_tile_rows()); the consumer's GroupNorm needs no statistics pass over the tensor
  EPI_GNBWD = 131072,   // the output is dy of a GroupNorm(+SiLU): also write the BACKWARD partial sums (sum dxh, sum dxh x per quad, slot layout of
                        // EPI_GNSTATS) to gn_part, from gnb_x (the GroupNorm's input, row stride gnb_ldx), gnb_scale / gnb_shift ([image][N]) and gnb_stats
  EPI_BIASINIT = 262144, // internal (set by the LDS-DMA launchers in place of EPI_BIAS): the accumulators start at bias[n] instead of zero
  EPI_SOFTBWD = 4096   // softmax backward fused into dP = dO . V^T:  v = (acc - rowbias[z * M + m]) * X[row, n]  (then alpha), with
                       // X = the probabilities P and rowbias = delta[row] = sum_c dO * O (fp32, one value per GEMM row and batch z; Zi = 1)
};

struct GemmDesc {
  const act_t* A;
  const act_t* W;
  void* C;
  act_t* C2;
  const float* bias;
  const float* rowbias;
  const act_t* R;
  const act_t* X;
  int M, N, K;           // K = ntaps*Ci (un-padded); ldw = W row stride in elements
  int ldw;
  int Hi, Wi, Ci, lda;   // input geometry; lda = elements per input pixel
  int Hq, Wq, sy, sx;
  int ntaps;
  int Ho, Wo, ldc, osy, ooy, osx, oox;
  int ldr, ldx, ldc2;
  int Z, Zi;             // batch count and inner batch size (z = zo*Zi + zi)
  long long sAo, sAi, sWo, sWi, sCo, sCi;
  float alpha, act_slope, mask_slope;
  int flags;
  signed char tdy[DMX_MAX_TAPS], tdx[DMX_MAX_TAPS];
  float resid_inv_slope;
  int tile_cfg;          // 0 = automatic; 1..18 force a tile configuration, 100 * slices + tile a split-K plan (tuning hook)
  int ldrb;              // row stride of rowbias in floats (0 = N); > N when it is a slice of a batched projection
  int ksplit;            // internal: > 1 = this launch is one K slice per blockIdx.z writing fp32 partials (set by the dispatcher)
  const unsigned char* XB;   // EPI_MASKBITS source, row stride ldxb bytes (rows indexed like C; Z must be 1)
  unsigned char* B2;         // EPI_BITS2 destination, row stride ldb2 bytes
  int ldxb, ldb2;
  const float* colsum;       // EPI_LNFOLD: sum over k of the packed (gamma-folded, 16-bit rounded) weight row, fp32 [N]
  float ln_eps;              // EPI_LNFOLD: LayerNorm epsilon
  const float* rowstats_in;  // EPI_LNFOLD: per-row partial sums [M][nslots][2] written by the producer of A (EPI_ROWSTATS)
  float* rowstats_out;       // EPI_ROWSTATS destination [rows][nslots][2]
  int nslots;                // 32-column slots per row of rowstats_in / rowstats_out
  float* gn_part;            // EPI_GNSTATS / EPI_GNBWD destination
  const act_t* gnb_x;        // EPI_GNBWD: the GroupNorm's saved input (rows indexed like C), row stride gnb_ldx elements
  const float* gnb_scale;    // EPI_GNBWD: gamma * rstd per (image, channel), row stride N
  const float* gnb_shift;    // EPI_GNBWD: beta - mean * gamma * rstd
  int gnb_ldx, gnb_silu;
  const float* gnb_stats;    // EPI_GNBWD: (mean, rstd) per (image, group)
  int gnb_cpg;               // EPI_GNBWD: channels per group (a multiple of 4)
};

int dmx_gemm_launch(const GemmDesc& d, hipStream_t stream);
void dmx_gemm_set_splitk_workspace(float* ws, size_t bytes);
void dmx_gemm_release_splitk_workspace(const float* ws);
// EPI_LNFOLD launches (gemm_ln.hip): tile configuration `cfg` as numbered in gemm_conv.hip, mapped onto the instantiated subset
int dmx_gemm_launch_ln(int cfg, const GemmDesc& d, hipStream_t stream);
int dmx_gemm_launch_rowstats(int cfg, const GemmDesc& d, hipStream_t stream);      // EPI_ROWSTATS producers (same file)
int dmx_gemm_launch_gnstats(int cfg, const GemmDesc& d, hipStream_t stream);       // EPI_GNSTATS producers (gemm_gn.hip)
// rows per wave tile (the TM of EPI_GNSTATS slots) of the most recent dmx_gemm_launch of this thread, 0 when that launch carried no
// statistics (split-K plan, direct epilogue)
int dmx_gemm_last_tile_rows();
void dmx_gemm_reset_last_tile_rows();
bool dmx_prof_is_active();
