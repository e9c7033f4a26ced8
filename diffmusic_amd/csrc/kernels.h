// Internal launcher prototypes (stream-ordered, non-allocating, return DMX_* codes).
#pragma once
#include "dmx_common.h"

#define DMX_GN_MAX_CHUNKS 512

// ---- elementwise.hip
size_t dmx_gn_scratch_floats(int B, int C, int G);
int dmx_groupnorm_fwd(const act_t* x, act_t* y, const float* gamma, const float* beta, float* stats, float* scale,
                      float* shift, float* partial, int B, int P, int C, int G, float eps, int silu, hipStream_t st);
int dmx_groupnorm_bwd(const act_t* x, const act_t* dy, const act_t* add, act_t* dx, const float* stats,
                      const float* scale, const float* shift, float* k0, float* k1, float* partial, int B, int P, int C,
                      int G, int silu, hipStream_t st);
int dmx_layernorm_fwd(const act_t* x, act_t* y, const float* gamma, const float* beta, int rows, int C, float eps,
                      hipStream_t st);
int dmx_softmax_fwd(const float* S, act_t* P, const float* colbias, long long rows, int N, long long lds, long long ldp,
                    int rows_per_bias, hipStream_t st);
int dmx_softmax_bwd(const act_t* P, const float* dP, act_t* dS, long long rows, int N, long long ld, float scale,
                    hipStream_t st);
int dmx_geglu(const act_t* x, act_t* y, long long rows, int I, hipStream_t st);
int dmx_silu(const act_t* x, act_t* y, long long n, hipStream_t st);
int dmx_upsample_nearest(const act_t* x, act_t* y, int B, int Hi, int Wi, int Ho, int Wo, int C, hipStream_t st);
int dmx_upsample2x_bwd(const act_t* dy, act_t* dx, int B, int Hi, int Wi, int C, hipStream_t st);
int dmx_transpose(const act_t* in, act_t* out, int R, int C, long long ldi, long long ldo, int Z, int Zi, long long sIo,
                  long long sIi, long long sOo, long long sOi, hipStream_t st);
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

// ---- mel.hip (measurement operators + mel transform, fp32)
int dmx_logmel_fwd(const float* wav, const float* fb, float* mel, float* power_ws, int B, int L, int n_mels, int to_db,
                   float clamp_lo, float clamp_hi, int window_hann, hipStream_t st);
int dmx_logmel_bwd(const float* wav, const float* fb, const float* dmel, float* dwav, int B, int L, int n_mels, int to_db,
                   float clamp_lo, float clamp_hi, int window_hann, hipStream_t st);

// ---- sched.hip
