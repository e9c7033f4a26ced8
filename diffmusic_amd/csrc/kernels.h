// Internal launcher prototypes (stream-ordered, non-allocating, return DMX_* codes).
#pragma once
#include "dmx_common.h"

#define DMX_GN_MAX_CHUNKS 512

// ---- elementwise.hip
size_t dmx_gn_scratch_floats(int B, int C, int G);
// partial sums of a GroupNorm input left by its producers (EPI_GNSTATS): see gn_parts_kernel (elementwise.hip)
struct GnRegion {
  float* part = nullptr;        // [image][slot][nq][2] fp32
  int tm = 0, P = 0, nq = 0;    // rows per slot; GEMM rows per image of the producing launch; quads per slot row (producer N / 4)
  int qoff = 0, cq = 0;         // first quad of this source in the normalised tensor; real quads of this source (C / 4)
};
struct GnParts { int n = 0; GnRegion r[8]; };
size_t dmx_gn_part_floats(int B, int P, int N);                      // fp32 elements of one region's buffer (worst case: 32-row slots)
int dmx_groupnorm_fwd(const act_t* x, act_t* y, const float* gamma, const float* beta, float* stats, float* scale,
                      float* shift, float* partial, int B, int P, int C, int G, float eps, int silu, hipStream_t st,
                      const GnParts* parts = nullptr);
int dmx_groupnorm_bwd(const act_t* x, const act_t* dy, const act_t* add, act_t* dx, const float* stats,
                      const float* scale, const float* shift, float* k0, float* k1, float* partial, int B, int P, int C,
                      int G, int silu, hipStream_t st, const GnParts* parts = nullptr);
int dmx_layernorm_fwd(const act_t* x, act_t* y, const float* gamma, const float* beta, int rows, int C, float eps,
                      hipStream_t st);
int dmx_softmax_fwd(const float* S, act_t* P, const float* colbias, long long rows, int N, long long lds, long long ldp,
                    int rows_per_bias, hipStream_t st);
int dmx_softmax_act(const act_t* S, act_t* P, const float* colbias, long long rows, int N, long long ldp, int rows_per_bias,
                    hipStream_t st);
int dmx_rowdot(const act_t* a, const act_t* b, float* out, long long rows, int C, long long lda, long long ldb, hipStream_t st);
int dmx_geglu(const act_t* x, act_t* y, long long rows, int I, hipStream_t st);
int dmx_silu(const act_t* x, act_t* y, long long n, hipStream_t st);
int dmx_upsample_nearest(const act_t* x, act_t* y, int B, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t st);
int dmx_upsample2x_bwd(const act_t* dy, act_t* dx, int B, int Hi, int Wi, int C, hipStream_t st);
int dmx_transpose(const act_t* in, act_t* out, int R, int C, long long ldi, long long ldo, int Z, int Zi, long long sIo,
                  long long sIi, long long sOo, long long sOi, hipStream_t st);
int dmx_concat2(const act_t* a, const act_t* b, act_t* dst, long long rows, int Ca, int Cb, hipStream_t st);
int dmx_copy_channels(const act_t* src, act_t* dst, long long rows, int C, int lds, int ldd, int soff, int doff,
                      hipStream_t st);
int dmx_axpby(const act_t* x, const act_t* y0, act_t* y, float a, float b, long long n, hipStream_t st);
int dmx_nchw_f32_to_nhwc_bf16(const float* x, act_t* y, int B, int C, int HW, int Cp, float scale, hipStream_t st);
int dmx_nhwc_bf16_to_nchw_f32(const act_t* x, float* y, int B, int C, int HW, int Cp, float scale, hipStream_t st);
int dmx_f32_to_bf16(const float* x, act_t* y, long long n, float scale, hipStream_t st);
int dmx_bf16_to_f32(const act_t* x, float* y, long long n, float scale, hipStream_t st);
int dmx_extract_col(const act_t* x, float* y, long long rows, int ld, int col, hipStream_t st);
int dmx_gather_col_f32(const float* x, float* y, long long rows, int ld, int col, hipStream_t st);
int dmx_tanh_bwd_pad8(const float* dwav, const float* wav8, act_t* gz, long long rows, hipStream_t st);
int dmx_scatter_col_pad8(const float* v, act_t* y, long long rows, float scale, hipStream_t st);
int dmx_gather_col_f32_to_act(const float* x, act_t* y, long long rows, int ld, int col, hipStream_t st);
int dmx_pad_col8_act(const act_t* v, act_t* y, long long rows, hipStream_t st);
int dmx_timestep_embed(const float* t, act_t* y, int B, int dim, hipStream_t st);

// ---- flash_attn.hip (forward-only fused attention for the U-Net)
bool dmx_flash_attn_ok(int dh, int C);
int dmx_flash_attn_fwd(const act_t* q, const act_t* k, const act_t* v, act_t* o, const float* colbias, int B, int Nq, int Nk,
                       int C, int heads, float scale, hipStream_t st, int ldq = 0, int ldk = 0, int ldv = 0);

// ---- mel.hip (STFT / mel measurement path, fp32)
int dmx_stft_tables(float* table, float* tableT, int n_fft, int bins, int Npad, int Kpad, int hann, hipStream_t st);
int dmx_stft_fwd(const float* wav, long long wav_stride, const float* table, float* X, int B, int L, int T, int n_fft, int hop, int Npad,
                 hipStream_t st);
int dmx_stft_bwd_frames(const float* Y, const float* tableT, float* dframe, int M, int n_fft, int Kpad, hipStream_t st);
int dmx_mel_fwd(const float* X, const float* fb, float* mel_lin, float* mel_out, int rows, int ldx, int bins, int n_mels, int power2,
                int to_db, float lo, float hi, hipStream_t st);
int dmx_mel_bwd(const float* X, const float* fb, const float* mel_lin, const float* dmel, float* Y, int rows, int ldx, int ldy, int bins,
                int n_mels, int power2, int to_db, float lo, float hi, hipStream_t st);
int dmx_stft_mag(const float* X, float* mag, int B, int T, int bins, int ldx, hipStream_t st);
int dmx_stft_mag_bwd(const float* X, const float* dmag, float* Y, int B, int T, int bins, int ldx, int ldy, hipStream_t st);
int dmx_overlap_add(const float* dframe, float* dwav, long long out_stride, int B, int T, int L, int n_fft, int hop, int accumulate,
                    hipStream_t st);
int dmx_l2_loss_grad(const float* ref, const float* pred, float* loss, float* dpred, int B, long long n, long long ref_stride, float gscale,
                     hipStream_t st);
int dmx_mask_mul(const float* x, long long xs, const float* mask, float* y, long long ys, int B, int L, int Ly, hipStream_t st);
int dmx_melscale(const float* mag, const float* fb, float* mel, int B, int T, int bins, int n_mels, float lo, float hi, hipStream_t st);
int dmx_absmax_normalize(float* x, float* inv_scale, int B, long long n, float target, hipStream_t st);

// ---- stft_mel.hip (fused STFT -> mel -> dB [-> L2] and its backward; n_fft = 1024, 64 mel columns)
struct DmxStftMelTables {
  const float2* tw;     // exp(-2 pi i m / 1024)
  const float* win;     // analysis window
  const int *klo, *klen; const float* fbc; int kmax;     // per mel column: first bin, bin count, weights [i][64]
  const int *mlo, *mlen; const float* fbr; int mmax;     // per bin: first mel column, column count, weights [i][576]
};
int dmx_stft_mel_parts(int L, int hop);                  // workgroups per clip of the forward launch (= partial sums per clip)
int dmx_stft_mel_fwd(const DmxStftMelTables& t, const float* wav, long long wav_stride, const float* mask, const float* ref, long long ref_stride,
                     float* mel_out, float* partial, int B, int L, int hop, int power2, int to_db, float lo, float hi, hipStream_t st);
int dmx_stft_mel_bwd(const DmxStftMelTables& t, const float* wav, long long wav_stride, const float* mask, const float* ref, long long ref_stride,
                     const float* dmel, const float* partial, float gscale, float* loss, float* dwav, long long dwav_stride, int Lfull,
                     int accumulate, int B, int L, int hop, int power2, int to_db, float lo, float hi, hipStream_t st);

// ---- sched.hip
int dmx_pred_x0(const float* x, const float* eps, float* x0, long long n, float sqrt_a, float sqrt_1ma, hipStream_t st);
int dmx_pred_x0_ex(const float* x, const float* m, float* x0, long long n, float sqrt_a, float sqrt_1ma, int ptype, float clip_r, hipStream_t st);
int dmx_cfg_combine(const float* eps2, float* out, long long n, float scale, hipStream_t st);
int dmx_sched_update(int mode, const float* x, const float* eps, const float* x0, const float* g0, const float* inv_scale,
                     const float* noise, float* prev, float* x0_out, float* grad_out, int B, int n, float alpha_t, float alpha_prev,
                     float sigma, float rate, float eps_small, int global_norm, hipStream_t st, int ptype = 0, float clip_r = 0.f);
