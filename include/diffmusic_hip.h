/* diffmusic_hip.h -- C ABI of the MI355X-native DiffMusic hot-path library (libdiffmusic_hip.so).
 *
 * The reference (jwliao1209/DiffMusic) is pure Python and has no FFI layer; its boundary for this
 * path is three duck-typed protocols: Scheduler.step()
 * (diffmusic/schedulers/scheduling_dps.py:137-219 and siblings), Pipeline.__call__()
 * (diffmusic/pipelines/pipeline_musicldm.py:491-799) and BaseOperator
 * (diffmusic/inverse_problem/operator.py:6-14).  The Python facade in diffmusic_amd/ keeps those
 * protocols and binds the entry points below with ctypes (INTEGRATION.md shows the stub a
 * maintainer of the reference would add).  Each entry point cites the reference call site whose
 * third-party / PyTorch computation it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; tensors are contiguous;
 *  - `stream` is a hipStream_t passed as void*; calls are stream-ordered, never synchronise and
 *    never allocate (model creation / parameter loading / finalize excepted);
 *  - workspaces are caller-owned device buffers; query sizes with the *_workspace_bytes calls;
 *  - return value 0 = OK, negative = error (dmx_last_error() gives the message);
 *  - 16-bit activation tensors (`uint16_t*`) are raw fp16 bit patterns (bf16 when the library was
 *    built with -DDMX_BF16; query dmx_act_dtype()), channels-last.
 */
#ifndef DIFFMUSIC_HIP_H
#define DIFFMUSIC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMX_ABI_VERSION 4   /* 4: dmx_htsat_* / dmx_gram_* (CLAP HTS-AT audio tower of the style-guidance operator).  Earlier:  2: dmx_flash_attn_raw takes row-major V (ld = ldv) instead of per-head V^T; GemmDesc grew.  3: GemmDesc grew (EPI_LNFOLD / EPI_ROWSTATS / EPI_GNSTATS / EPI_GNBWD: colsum, ln_eps, rowstats_in, rowstats_out, nslots, gn_part, gnb_*) */
#define DMX_MAX_STAGES 8

typedef struct dmx_model dmx_model; /* opaque network handle (weights repacked for MFMA) */

/* transformers SpeechT5HifiGanConfig fields used by the vocoder (operator.py:126-130 call site) */
typedef struct dmx_hifigan_config {
  int model_in_dim;             /* 64 */
  int upsample_initial_channel; /* 1024 */
  int num_upsamples;            /* 5 */
  int upsample_rates[DMX_MAX_STAGES];
  int upsample_kernel_sizes[DMX_MAX_STAGES];
  int num_kernels;              /* 3 */
  int resblock_kernel_sizes[DMX_MAX_STAGES];
  int num_dilations;            /* 3 */
  int resblock_dilation_sizes[DMX_MAX_STAGES * DMX_MAX_STAGES]; /* [kernel][dilation] */
  float leaky_relu_slope;       /* 0.1 */
} dmx_hifigan_config;

/* diffusers AutoencoderKL decoder config (scheduling_dps.py:195-197 call site) */
typedef struct dmx_vae_config {
  int latent_channels;          /* 8 */
  int out_channels;             /* 1 */
  int num_blocks;               /* 3 */
  int block_out_channels[DMX_MAX_STAGES]; /* 128,256,512 */
  int layers_per_block;         /* 2 */
  int norm_num_groups;          /* 32 */
  float eps;                    /* 1e-6 */
} dmx_vae_config;

/* diffusers UNet2DConditionModel config as used by MusicLDM (pipeline_musicldm.py:696-703) and,
 * with cross_attention contexts, AudioLDM2 (plpeline_audioldm2.py:1147-1154) */
typedef struct dmx_unet_config {
  int in_channels, out_channels; /* 8, 8 */
  int num_blocks;                /* 4 */
  int block_out_channels[DMX_MAX_STAGES];
  int layers_per_block;          /* 2 */
  int attention_heads;           /* 8 */
  int norm_num_groups;           /* 32 */
  int down_attn[DMX_MAX_STAGES]; /* 0,1,1,1 */
  int up_attn[DMX_MAX_STAGES];   /* 1,1,1,0 */
  int class_embed_dim;           /* 512 (simple_projection, concat) ; 0 = none */
  int num_attn_per_layer;        /* MusicLDM 1 ; AudioLDM2 3 (one Transformer2DModel per cross_attention_dim entry) */
  int attn_cross_dims[4];        /* per transformer: <= 0 self-attention (context None), > 0 cross-attention width;
                                    the k-th positive entry attends context k (AudioLDM2: {0, 768, 1024}) */
} dmx_unet_config;

int dmx_abi_version(void);
int dmx_act_dtype(void); /* 1 = fp16 (default build), 0 = bf16 */
const char* dmx_last_error(void);

/* ---- model lifecycle ------------------------------------------------------------------------ */
dmx_model* dmx_hifigan_create(const dmx_hifigan_config* cfg);
dmx_model* dmx_vae_decoder_create(const dmx_vae_config* cfg);
dmx_model* dmx_unet_create(const dmx_unet_config* cfg);
void dmx_model_destroy(dmx_model* m);
int dmx_model_num_params(const dmx_model* m);
const char* dmx_model_param_name(const dmx_model* m, int i);
size_t dmx_model_param_numel(const dmx_model* m, int i);
int dmx_model_param_ndim(const dmx_model* m, int i);
int dmx_model_param_dim(const dmx_model* m, int i, int d);
/* copy one fp32 parameter (upstream naming, e.g. "resblocks.3.convs1.0.weight") from host memory */
int dmx_model_load_param(dmx_model* m, const char* name, const float* data_host, size_t numel);
/* repack all parameters into the MFMA layouts; fails if any parameter is missing */
int dmx_model_finalize(dmx_model* m, void* stream);

/* ---- HiFi-GAN vocoder: replaces `vocoder(mel)` (operator.py:126-130) and its autograd backward */
int dmx_hifigan_out_len(const dmx_model* m, int frames);
size_t dmx_hifigan_workspace_bytes(dmx_model* m, int batch, int frames);
/* mel (B, frames, model_in_dim) f16 -> wav (B, out_len) fp32; keeps the backward state in ws */
int dmx_hifigan_fwd(dmx_model* m, const uint16_t* mel, float* wav, int batch, int frames, void* ws, size_t ws_bytes,
                    void* stream);
/* dwav (B, out_len) fp32 -> dmel (B, frames, model_in_dim) f16; ws as left by the forward call */
int dmx_hifigan_bwd(dmx_model* m, const float* dwav, uint16_t* dmel, void* stream);

/* ---- VAE decoder: replaces `vae.decode(z).sample` (scheduling_dps.py:195-197) + backward ------- */
size_t dmx_vae_workspace_bytes(dmx_model* m, int batch, int h, int w);
/* z (B, latent_channels, h, w) fp32 NCHW, multiplied by z_scale -> mel (B, 4h, 4w) f16 (+ fp32 copy if mel_f32) */
int dmx_vae_decode_fwd(dmx_model* m, const float* z, float z_scale, uint16_t* mel, float* mel_f32, int batch, int h, int w,
                       int keep_state, void* ws, size_t ws_bytes, void* stream);
/* dmel (B, 4h, 4w) f16 -> dz (B, latent_channels, h, w) fp32 NCHW, multiplied by z_scale */
int dmx_vae_decode_bwd(dmx_model* m, const uint16_t* dmel, float z_scale, float* dz, void* stream);

/* ---- U-Net forward: replaces `self.unet(latent_model_input, t, ..., class_labels=...)` -------- */
size_t dmx_unet_workspace_bytes(dmx_model* m, int batch, int h, int w);
/* x (B, in_ch, h, w) fp32 NCHW, t (B) fp32 timesteps, class_labels (B, class_embed_dim) fp32 -> eps (B, out_ch, h, w) fp32 */
int dmx_unet_fwd(dmx_model* m, const float* x, const float* t, const float* class_labels, float* eps, int batch, int h,
                 int w, void* ws, size_t ws_bytes, void* stream);
/* AudioLDM2 variant (plpeline_audioldm2.py:1147-1154): ctx0 (B, n0, d0) = generated_prompt_embeds (GPT-2), ctx1 (B, n1, d1) =
 * prompt_embeds (T5), bias1 (B, n1) additive score bias = (1 - attention_mask) * -10000; all fp32; n0, n1 multiples of 4 */
int dmx_unet_fwd_ctx(dmx_model* m, const float* x, const float* t, const float* class_labels, const float* ctx0, int n0,
                     const float* ctx1, int n1, const float* bias1, float* eps, int batch, int h, int w, void* ws,
                     size_t ws_bytes, void* stream);
size_t dmx_unet_workspace_bytes_ctx(dmx_model* m, int batch, int h, int w, int n0, int n1);

/* ---- CLAP HTS-AT audio tower (transformers ClapAudioModel; `StyleGuidanceOperator.transform`, diffmusic/inverse_problem/operator.py:253-271,
 * config 5 of BASELINE.json): forward with tape and input-gradient backward, hand-written like the three networks above.  Parameter names
 * are the `ClapAudioModel.state_dict()` keys ("audio_encoder. ..."). ---------------------------------------------------------------------- */
typedef struct dmx_htsat_config {
  int spec_size;      /* 256: side of the mel "image" */
  int num_mel_bins;   /* 64 */
  int patch_size;     /* 4 (= stride) */
  int embed_dim;      /* 96 */
  int window_size;    /* 8 */
  int num_stages;     /* 4 */
  int depths[4];      /* 2, 2, 6, 2 */
  int num_heads[4];   /* 4, 8, 16, 32 (head dim 24 in every stage) */
  float ln_eps;       /* 1e-5 */
  float bn_eps;       /* 1e-5 */
} dmx_htsat_config;
dmx_model* dmx_htsat_create(const dmx_htsat_config* cfg);
/* tokens x channels of the feature map dmx_htsat_fwd returns (64 x 768 for the default configuration) */
int dmx_htsat_feature_dims(dmx_model* m, int* tokens, int* channels);
size_t dmx_htsat_workspace_bytes(dmx_model* m, int batch, int frames);      /* forward tape + backward scratch */
/* mel (B, frames, num_mel_bins) fp32 log-mel (ClapFeatureExtractor's input_features without the channel axis), 2 <= frames <= 1024 ->
 * feat (B, tokens, channels) fp32 = the tower's last_hidden_state (after the final LayerNorm), tokens in grid order.  keep_state != 0
 * keeps the tape in `ws` for dmx_htsat_bwd.  The first call with a new `frames` builds that length's bicubic tables (allocates). */
int dmx_htsat_fwd(dmx_model* m, const float* mel, int batch, int frames, float* feat, int keep_state, void* ws, size_t ws_bytes, void* stream);
/* dfeat (B, tokens, channels) fp32 -> dmel (B, frames, num_mel_bins) fp32, multiplied by scale[b] when scale != NULL (the inverse of a
 * per-clip normalisation the caller applied to dfeat: the sweep runs in 16 bits). */
int dmx_htsat_bwd(dmx_model* m, const float* dfeat, const float* scale, float* dmel, void* stream);
/* test hook: copies a tape tensor of the last dmx_htsat_fwd(keep_state = 1) into dst (16-bit activations, token-major): which = 0 block
 * input, 1 q|k|v, 2 hidden state after attention, 3 MLP pre-activation; block == the stage's depth: the stage output before patch merging.
 * Returns the tensor's element count (0: no such tensor); copies only when dst_elems is at least that. */
size_t dmx_htsat_tape_raw(dmx_model* m, int stage, int block, int which, void* dst, size_t dst_elems, void* stream);
/* Gram matrix of token features: G[b] = F[b]^T F[b] / T, F (B, T, C) fp32 -> G (B, C, C); and dF = F (dG + dG^T) / T */
int dmx_gram_fwd(const float* F, float* G, int batch, int tokens, int channels, void* stream);
int dmx_gram_bwd(const float* F, const float* dG, float* dF, int batch, int tokens, int channels, void* stream);

/* ---- STFT / mel measurement path (fp32): replaces torchaudio MelSpectrogram + AmplitudeToDB / MelScale and
 * torch.stft as used by the operators (diffmusic/inverse_problem/operator.py:23-33,143-147,162-170) and the
 * autograd sweep through them (diffmusic/schedulers/scheduling_dps.py:202-212) ------------------------------- */
typedef struct dmx_audio dmx_audio;
/* fb_host: (n_fft/2+1, n_mels) fp32 mel filterbank in host memory; window_hann: 1 = periodic hann, 0 = rectangular */
dmx_audio* dmx_audio_create(int n_fft, int hop, int n_mels, int window_hann, const float* fb_host);
void dmx_audio_destroy(dmx_audio* a);
int dmx_audio_num_frames(const dmx_audio* a, int L);
int dmx_audio_num_bins(const dmx_audio* a); /* n_fft / 2 + 1 */
size_t dmx_audio_state_bytes(const dmx_audio* a, int batch, int L);
/* wav (B, L) fp32 (row stride wav_stride) -> mel_out (B, frames, n_mels) fp32.  power2: |X|^2 (1) or |X| (0);
 * to_db: 10*log10(max(.,1e-10)); then clamp(lo, hi).  `state` keeps what the backward call needs (the spectrum on the dense-DFT path,
 * a copy of the waveform on the fused n_fft = 1024 path). */
int dmx_audio_transform_fwd(dmx_audio* a, const float* wav, long long wav_stride, float* mel_out, void* state, int batch, int L,
                            int power2, int to_db, float lo, float hi, void* stream);
/* dmel (B, frames, n_mels) -> dwav (B, L) (row stride dwav_stride), same flags as the forward call */
int dmx_audio_transform_bwd(dmx_audio* a, const float* dmel, float* dwav, long long dwav_stride, void* state, int batch, int L,
                            int power2, int to_db, float lo, float hi, int accumulate, void* stream);
/* Fused guidance pair for n_fft = 1024 (dmx_audio_is_fused): everything between the vocoder output and its gradient in one forward and
 * one backward launch -- y = wav * mask (mask NULL: y = wav; MusicInpaintingOperator.forward, operator.py:132-133), transform(y) as
 * above (operator.py:23-33 / :143-147), loss[b] = ||ref[b] - transform(y[b])||_2 (torch.linalg.norm, scheduling_dps.py:205-211) and
 * dwav = gscale * d loss / d wav (torch.autograd.grad, scheduling_dps.py:212), written for samples [0, L) and zeroed on [L, Lfull).
 * ref: (B or 1, frames, n_mels) with row stride ref_stride elements per clip (0 = one reference for all clips).  The spectrum is never
 * stored: the backward launch recomputes it from wav.  `state` carries the per-workgroup partial sums of the loss from _fwd to _bwd
 * (same stream).  mel_out may be NULL.  Returns DMX_ERR_SHAPE for handles / lengths the fused kernels do not cover. */
int dmx_audio_is_fused(const dmx_audio* a, int L);
int dmx_audio_guidance_fwd(dmx_audio* a, const float* wav, long long wav_stride, const float* mask, const float* ref, long long ref_stride,
                           float* mel_out, void* state, int batch, int L, int power2, int to_db, float lo, float hi, void* stream);
int dmx_audio_guidance_bwd(dmx_audio* a, const float* wav, long long wav_stride, const float* mask, const float* ref, long long ref_stride,
                           float gscale, float* loss, float* dwav, long long dwav_stride, int Lfull, void* state, int batch, int L,
                           int power2, int to_db, float lo, float hi, void* stream);
/* PhaseRetrievalOperator.forward: |torch.stft(wav)| as (B, n_fft/2+1, frames) fp32 */
int dmx_audio_stft_mag(dmx_audio* a, const float* wav, long long wav_stride, float* mag, void* state, int batch, int L, void* stream);
/* gradient of a loss on that magnitude (PhaseRetrievalOperator.forward, operator.py:156-163, differentiated by
 * torch.autograd in scheduling_dps.py:199-212 when supervised_space == "wav_form"): dmag (batch, bins, frames) -> dwav;
 * uses the spectrum the last dmx_audio_stft_mag left in `state` */
int dmx_audio_stft_mag_bwd(dmx_audio* a, const float* dmag, float* dwav, long long dwav_stride, void* state, int batch, int L,
                           int accumulate, void* stream);
/* PhaseRetrievalOperator.transform on a given magnitude (B, bins, frames) -> (B, frames, n_mels) */
int dmx_audio_melscale(dmx_audio* a, const float* mag, float* mel_out, int batch, int frames, float lo, float hi, void* stream);
/* MusicInpaintingOperator.forward (operator.py:132-133): y[b,t] = x[b,t]*mask[t] (t<L), 0 for L<=t<Ly; mask NULL = copy */
int dmx_mask_apply(const float* x, long long x_stride, const float* mask, float* y, long long y_stride, int batch, int L, int Ly,
                   void* stream);
/* per-clip loss[b] = ||ref_b - pred_b||_2 (torch.linalg.norm, scheduling_dps.py:211) and dpred = gscale * dloss/dpred */
int dmx_l2_loss(const float* ref, long long ref_stride, const float* pred, float* loss, float* dpred, int batch, long long n,
                float gscale, void* stream);
/* per-clip x *= target/max|x| ; inv_scale[b] = max|x|/target  (keeps the fp16 backward sweep in range) */
int dmx_grad_normalize(float* x, float* inv_scale, int batch, long long n, float target, void* stream);

/* Polyphase / dense FIR (fp32): out[j*new + p] = sum_t h[p][t] * in[j*orig + t - off], zero outside [0, Lin).
 * Replaces torchaudio Resample in SuperResolutionOperator.forward (operator.py:203-205; h = sinc-hann kernel (new, taps),
 * off = width) and F.conv1d in MusicDereverberationOperator.forward (operator.py:247-249; orig = new = 1, off = taps/2). */
int dmx_fir_fwd(const float* in, long long in_stride, const float* h, float* out, long long out_stride, int batch, int Lin, int Lout,
                int taps, int orig, int new_, int off, void* stream);
/* transpose of dmx_fir_fwd (gradient w.r.t. `in`); h_rev = time-reversed taps, required only for the dense 1:1 case */
int dmx_fir_bwd(const float* dout, long long dout_stride, const float* h, const float* h_rev, float* din, long long din_stride, int batch,
                int Lin, int Lout, int taps, int orig, int new_, int off, void* stream);

/* ---- scheduler arithmetic (diffmusic/schedulers/scheduling_{ddim,dps,mpgd,dsg,diffmusic}.py step bodies) -------------- */
#define DMX_SCHED_DDIM 0
#define DMX_SCHED_DPS 1
#define DMX_SCHED_MPGD 2
#define DMX_SCHED_DSG 3
#define DMX_SCHED_DIFFMUSIC 4
/* x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t) */
int dmx_sched_pred_x0(const float* x, const float* eps, float* x0, long long n, float alpha_t, void* stream);
/* the other prediction types of the diffusers DDIM parent every reference scheduler subclasses (scheduling_dps.py:15-61): prediction_type
 * 0 epsilon (as above), 1 sample (x0 = model_output), 2 v_prediction (x0 = sqrt(a_t) x - sqrt(1-a_t) v); clip_range > 0 clamps x0 to
 * [-clip_range, clip_range] (clip_sample) */
int dmx_sched_pred_x0_ex(const float* x, const float* model_output, float* x0, long long n, float alpha_t, int prediction_type, float clip_range,
                         void* stream);
/* classifier-free guidance combine on a (2B, ...) U-Net output (pipeline_musicldm.py:706-708) */
int dmx_sched_cfg_combine(const float* eps2, float* out, long long n, float scale, void* stream);
/* fused update: g0 = dLoss/dx0 (times 1/inv_scale[b]); see csrc/sched.hip for the per-mode formulas */
int dmx_sched_step(int mode, const float* x, const float* eps, const float* x0, const float* g0, const float* inv_scale,
                   const float* noise, float* prev, float* x0_out, float* grad_out, int batch, int n, float alpha_t, float alpha_prev,
                   float sigma, float rate, float eps_small, int global_norm, void* stream);
/* dmx_sched_step for a parent step of another prediction type / with clip_sample: the gradient w.r.t. x_t passes through x0(x_t), i.e.
 * d x0 / d x_t = 1 / sqrt(a_t), 0 or sqrt(a_t), and zero where x0 sits on the clip bound */
int dmx_sched_step_ex(int mode, const float* x, const float* eps, const float* x0, const float* g0, const float* inv_scale,
                      const float* noise, float* prev, float* x0_out, float* grad_out, int batch, int n, float alpha_t, float alpha_prev,
                      float sigma, float rate, float eps_small, int global_norm, int prediction_type, float clip_range, void* stream);

/* Device-side N(0,1) noise, Philox4x32-10 + Box-Muller (csrc/rng.hip): optional replacement for the host draw + upload of
 * randn_tensor (diffmusic/torch_utils.py:31-76) that DSG / DiffMusic pay every step (scheduling_dsg.py:215).  out (batch, n)
 * fp32; clip b uses key seeds_host[b] (HOST array of `batch` <= 64 values); element i = normal (i & 3) of Philox block
 * offset + i / 4, so a draw depends only on (seed_b, offset, i) -- not on the batch composition or the number of GPUs. */
int dmx_randn_philox(float* out, int batch, long long n, const unsigned long long* seeds_host, unsigned long long offset, void* stream);

/* ---- measurement hooks: HIP events around every implicit-GEMM launch (bench.py roofline leg) ---------- */
void dmx_prof_begin(void);
int dmx_prof_end(double* total_ms, double* total_flops); /* returns the number of launches recorded */
/* of the region closed by the last dmx_prof_end: launches / kernel ms / FLOPs / algorithmic operand bytes of gemm_glds_kernel (LDS-DMA tiles) alone */
int dmx_prof_dominant(double* ms, double* flops, double* bytes);

/* ---- low-level test hook: one implicit-GEMM launch described by the internal descriptor --------*/
int dmx_gemm_raw(const void* desc, size_t desc_bytes, void* stream);
/* test hook for the fused convolution pair (HiFi-GAN resblock step, C = 32 / 64): stage `a` (may be NULL: plain slab
 * convolution) feeds stage `b` through LDS.  Returns DMX_ERR_SHAPE when the shape is not handled by the fused kernel. */
/* test hook for the fused forward attention of the U-Net (diffusers Attention inside UNet2DConditionModel,
 * pipeline_musicldm.py:696-703): q (B,Nq,C), k (B,Nk,C), v (B,Nk,ldv) fp16 channels-last (ldv = 0: C), o (B,Nq,C); colbias optional
 * (B,Nk) fp32 additive key bias. */
int dmx_flash_attn_raw(const void* q, const void* k, const void* v, void* o, const float* colbias, int B, int Nq, int Nk, int ldv,
                       int C, int heads, float scale, void* stream);
/* test hook: fp32 scratch that lets small-M / deep-K launches run split-K (NULL disables it); the U-Net executor
 * installs its own */
int dmx_gemm_splitk_workspace(void* ws, size_t bytes);
int dmx_conv_pair_raw(const void* desc_a, const void* desc_b, size_t desc_bytes, void* stream);
/* test hook: n (<= 3) mutually independent fused pairs of one width as ONE grid, longest problem first (the k = 3 / 7 / 11 branches
 * of a HiFi-GAN resblock step, transformers HifiGanResidualBlock.forward); descs_a / descs_b: n consecutive descriptors each. */
int dmx_conv_pair_group_raw(int n, const void* descs_a, const void* descs_b, size_t desc_bytes, void* stream);
/* test hook: GroupNorm (+ SiLU) forward as the U-Net / VAE executors run it (diffusers ResnetBlock2D norm1 / norm2, Attention
 * group_norm; reached from pipeline_musicldm.py:696-703 and scheduling_dps.py:195-197).  x, y (B, P, C) fp16 channels-last;
 * stats (B, G, 2) = (mean, rstd), scale / shift (B, C) fp32 outputs; partial: fp32 scratch of dmx_groupnorm_scratch_floats(B, C, G). */
size_t dmx_groupnorm_scratch_floats(int B, int C, int G);
int dmx_groupnorm_raw(const void* x, void* y, const float* gamma, const float* beta, float* stats, float* scale, float* shift,
                      float* partial, int B, int P, int C, int G, float eps, int silu, void* stream);
/* GroupNorm whose statistics come from partial sums the PRODUCERS of x wrote in their GEMM epilogues (GemmDesc flag EPI_GNSTATS,
 * gn_part): nreg (1..8) regions, part[i] = buffer of dmx_groupnorm_part_floats(B, P_i, N_i) floats, geom[6 i ..] = {rows per slot (what
 * dmx_gemm_last_tile_rows_raw() reported after the producing launch), GEMM rows per image of that launch, N_i / 4, first 4-channel
 * quad of the source in x, real quads of the source, 0}.  Same outputs as dmx_groupnorm_raw; no statistics pass over x. */
size_t dmx_groupnorm_part_floats(int B, int P, int N);
int dmx_groupnorm_parts_raw(const void* x, void* y, const float* gamma, const float* beta, float* stats, float* scale, float* shift,
                            int B, int P, int C, int G, float eps, int silu, int nreg, float* const* part, const int* geom, void* stream);
int dmx_gemm_last_tile_rows_raw(void);
/* GroupNorm(+SiLU) backward (input gradient): the two per-group sums from EPI_GNBWD partial sums of the dgrad launch that produced dy
 * (nreg regions, as above) or, with nreg == 0, from the classic pass over x and dy.  stats / scale / shift: the forward's outputs;
 * k0, k1: (B, C) fp32 scratch; partial: dmx_groupnorm_scratch_floats(B, C, G) floats (used when nreg == 0); add: optional tensor added to dx. */
int dmx_groupnorm_bwd_raw(const void* x, const void* dy, const void* add, void* dx, const float* stats, const float* scale, const float* shift,
                          float* k0, float* k1, float* partial, int B, int P, int C, int G, int silu, int nreg, float* const* part,
                          const int* geom, void* stream);

#ifdef __cplusplus
}
#endif
#endif
