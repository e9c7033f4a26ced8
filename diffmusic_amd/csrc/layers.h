// Host-side layer plumbing shared by the three network executors (hifigan.cpp, vae.cpp, unet.cpp):
// parameter registry, workspace arena, packed conv layers and the GemmDesc builders.
#pragma once
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <cstring>
#include "dmx_common.h"
#include "kernels.h"

void dmx_set_error(const char* fmt, ...);

inline int pad8(int c) { return (c + 7) & ~7; }

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool overflow = false;
  void reset(void* b, size_t c) { base = (char*)b; cap = c; off = 0; overflow = false; }
  void* raw(size_t bytes) {
    bytes = align_up(bytes, 256);
    if (off + bytes > cap) { overflow = true; off += bytes; if (off > peak) peak = off; return base; }  // keep counting
    void* p = base + off;
    off += bytes;
    if (off > peak) peak = off;
    return p;
  }
  act_t* bf(size_t n) { return (act_t*)raw(n * 2); }
  float* f32(size_t n) { return (float*)raw(n * 4); }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

struct Param {
  std::string name;
  std::vector<int> shape;
  size_t numel = 0;
  float* dev = nullptr;  // fp32 copy on device
  bool loaded = false;
};

struct ParamStore {
  std::vector<Param> params;
  std::map<std::string, int> index;
  std::vector<void*> owned;  // device allocations freed on destroy
  int add(const std::string& name, std::vector<int> shape);
  float* dev(int id) const { return params[id].dev; }
  int load(const char* name, const float* host, size_t numel);
  bool all_loaded(std::string* missing) const;
  void* dalloc(size_t bytes);
  void free_all();
};

// A convolution (1-D, 2-D, transposed 1-D) or linear layer prepared for the implicit-GEMM kernel.
struct ConvLayer {
  int Ci = 0, Co = 0, Cip = 0, Cop = 0;
  int kh = 1, kw = 1, stride = 1, dil = 1, pad_h = 0, pad_w = 0;
  bool transposed = false;  // ConvTranspose1d (weight layout Cin,Cout,k)
  bool has_bias = true, need_bwd = false;
  bool geglu = false;       // linear layer whose output is [values | gates]: rows packed in blocks of 32 = [16 values | their 16 gates]
                            // (bias likewise) so that the GEMM epilogue can apply GEGLU (EPI_GEGLU); forward only
  int w_id = -1, b_id = -1;
  // LayerNorm folded into this linear layer (forward only): parameter ids of the norm's weight / bias.  pack_layer then packs W diag(gamma),
  // adds W beta to the bias and keeps the row sums of the packed matrix; linear_fwd runs it on the RAW (un-normalised) rows with EPI_LNFOLD
  int ln_g_id = -1, ln_b_id = -1;
  float ln_eps = 1e-5f;
  float* colsum = nullptr;          // fp32 [Cop]
  // packed
  std::vector<act_t*> wf;          // forward weights (one per output phase for transposed)
  std::vector<std::vector<int>> wf_taps;  // tap index list per phase (transposed)
  act_t* wb = nullptr;             // dgrad weights
  float* bias = nullptr;            // fp32 [Cop]
  // nearest x2 upsampling folded into a 3x3 convolution (pack_layer_up2x): four 2x2-tap matrices, one per output parity class,
  // with the weights of the taps that read the same low-resolution pixel summed; and their joint dgrad as one 4x4-tap stride-2 matrix
  std::vector<act_t*> wup;
  act_t* wup_b = nullptr;
  int ntaps() const { return kh * kw; }
};

struct Epi {
  int flags = 0;
  float alpha = 1.f, act_slope = 0.f, mask_slope = 0.f, resid_inv_slope = 1.f;
  const act_t* R = nullptr;
  const act_t* X = nullptr;
  act_t* C2 = nullptr;
  const float* rowbias = nullptr;
  int ldrb = 0;                 // row stride of rowbias (0 = the layer's padded Cout)
  const unsigned char* XB = nullptr;   // EPI_MASKBITS source (sign bits, N / 8 bytes per output row)
  unsigned char* B2 = nullptr;         // EPI_BITS2 destination
  float* rowstats_out = nullptr;       // EPI_ROWSTATS: per-row partial sums of the output (for a LayerNorm folded into the next projection)
  const float* rowstats_in = nullptr;  // linear_fwd of a LayerNorm-folded layer: the partial sums its input's producer wrote
  int nslots = 0;                      // 32-column slots per row of either
  float* gn_part = nullptr;            // EPI_GNSTATS: GroupNorm partial sums of the output (ask dmx_gemm_last_tile_rows() after the launch)
  // EPI_GNBWD (with gn_part): the output is dy of a GroupNorm(+SiLU) with saved input gnb_x and tape scale / shift: backward partial sums
  const act_t* gnb_x = nullptr;
  const float* gnb_scale = nullptr;
  const float* gnb_shift = nullptr;
  int gnb_silu = 0, gnb_cpg = 0;
  const float* gnb_stats = nullptr;
};

// registry helpers
ConvLayer make_conv1d(ParamStore& ps, const std::string& prefix, int Ci, int Co, int k, int dil, int pad, bool need_bwd);
ConvLayer make_convT1d(ParamStore& ps, const std::string& prefix, int Ci, int Co, int k, int stride, int pad, bool need_bwd);
ConvLayer make_conv2d(ParamStore& ps, const std::string& prefix, int Ci, int Co, int k, int stride, int pad, bool need_bwd);
ConvLayer make_linear(ParamStore& ps, const std::string& prefix, int Ci, int Co, bool bias, bool need_bwd);
int pack_layer(ParamStore& ps, ConvLayer& L, hipStream_t st);
// n linear layers of equal input width (registered, loaded) packed as ONE layer whose output is their outputs side by side
// (weights stacked along N, forward and -- when the sources ask for it -- dgrad; biases concatenated): q | k | v in one GEMM
int pack_linear_stack(ParamStore& ps, ConvLayer& dst, const ConvLayer* const* src, int n, hipStream_t st);

// launches.  Tensors are channels-last with padded channel counts (Cip / Cop).
// 1-D: in (B, Ti, Cip) -> out (B, To, Cop);  2-D: in (B, Hi, Wi, Cip) -> out (B, Ho, Wo, Cop)
int conv_out_len(const ConvLayer& L, int Ti);
int conv_fwd_1d(const ConvLayer& L, const act_t* in, void* out, int B, int Ti, const Epi& e, hipStream_t st);
int conv_bwd_1d(const ConvLayer& L, const act_t* dout, void* din, int B, int Ti, const Epi& e, hipStream_t st);
// descriptor-only builders (stride-1, non-transposed) and the fused two-stage launch used by the HiFi-GAN resblocks
int conv_fwd_1d_desc(const ConvLayer& L, const act_t* in, void* out, int B, int Ti, const Epi& e, GemmDesc& d);
int conv_bwd_1d_desc(const ConvLayer& L, const act_t* dout, void* din, int B, int Ti, const Epi& e, GemmDesc& d);
int conv_pair_run(const GemmDesc& a, const GemmDesc& b, hipStream_t st);
int conv_fwd_2d(const ConvLayer& L, const act_t* in, void* out, int B, int Hi, int Wi, const Epi& e, hipStream_t st);
// conv3x3(pad 1)(nearest_upsample_x2(in)) without the upsampled tensor: in (B, Hi, Wi, Cip) -> out (B, 2Hi, 2Wi, Cop), and its dgrad
// dout (B, 2Hi, 2Wi, Cop) -> din (B, Hi, Wi, Cip) (the gradient w.r.t. the LOW-resolution input, upsample backward included)
int pack_layer_up2x(ParamStore& ps, ConvLayer& L, hipStream_t st);
// gn_buf / gn_tm (optional, 4 entries): GroupNorm partial-sum buffers of the four output-parity launches (EPI_GNSTATS) and, on return, the
// slot rows each launch used (0 = that launch carried no statistics)
int conv_up2x_fwd(const ConvLayer& L, const act_t* in, void* out, int B, int Hi, int Wi, const Epi& e, hipStream_t st,
                  float* const* gn_buf = nullptr, int* gn_tm = nullptr);
int conv_up2x_bwd(const ConvLayer& L, const act_t* dout, void* din, int B, int Hi, int Wi, const Epi& e, hipStream_t st);
int conv_bwd_2d(const ConvLayer& L, const act_t* dout, void* din, int B, int Hi, int Wi, const Epi& e, hipStream_t st);
// plain (batched) NT GEMM: C[z] = alpha * A[z] (M,K; lda) * Bm[z]^T (N,K; ldb)  (+ epilogue)
struct GemmBatch { int Z = 1, Zi = 1; long long sAo = 0, sAi = 0, sBo = 0, sBi = 0, sCo = 0, sCi = 0; };
int gemm_nt(const act_t* A, int lda, const act_t* Bm, int ldb, void* C, int ldc, int M, int N, int K, const Epi& e,
            const GemmBatch& gb, hipStream_t st);
// linear layer on (rows, Cip) -> (rows, Cop)
int linear_fwd(const ConvLayer& L, const act_t* in, int lda, void* out, int ldc, long long rows, const Epi& e, hipStream_t st);
int linear_bwd(const ConvLayer& L, const act_t* dout, int lda, void* din, int ldc, long long rows, const Epi& e, hipStream_t st);

struct GroupNormLayer {
  int C = 0, G = 32, g_id = -1, b_id = -1;
  float eps = 1e-5f;
};
GroupNormLayer make_gn(ParamStore& ps, const std::string& prefix, int C, int G, float eps);
struct GnSave { float* stats = nullptr; float* scale = nullptr; float* shift = nullptr; };
