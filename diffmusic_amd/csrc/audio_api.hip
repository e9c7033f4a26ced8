// extern "C" entry points of the STFT / mel measurement path and the scheduler updates.
#include "kernels.h"
#include "../../include/diffmusic_hip.h"
#include <vector>
void dmx_set_error(const char* fmt, ...);
#define ST(s) ((hipStream_t)(s))

struct dmx_audio {
  int n_fft, hop, bins, n_mels, hann, Npad, Kpad;
  float *table, *tableT, *fb;
};

static int frames_of(const dmx_audio* a, int L) { return 1 + L / a->hop; }
struct AudioState { float *X, *mel_lin, *Y, *dframe; };
static AudioState carve(const dmx_audio* a, void* state, int B, int L) {
  const size_t M = (size_t)B * frames_of(a, L);
  AudioState s;
  float* p = (float*)state;
  s.X = p; p += align_up(M * a->Npad, 64);
  s.mel_lin = p; p += align_up(M * a->n_mels, 64);
  s.Y = p; p += align_up(M * a->Kpad, 64);
  s.dframe = p;
  return s;
}

extern "C" {

dmx_audio* dmx_audio_create(int n_fft, int hop, int n_mels, int window_hann, const float* fb_host) {
  if (n_fft % 32 || n_fft < 64 || hop < 1 || n_mels != 64 || !fb_host) { dmx_set_error("bad audio config"); return nullptr; }
  dmx_audio* a = new dmx_audio();
  a->n_fft = n_fft; a->hop = hop; a->bins = n_fft / 2 + 1; a->n_mels = n_mels; a->hann = window_hann;
  a->Npad = (int)align_up(2 * a->bins, 64); a->Kpad = (int)align_up(2 * a->bins, 32);
  if (hipMalloc(&a->table, (size_t)a->Npad * n_fft * 4) != hipSuccess || hipMalloc(&a->tableT, (size_t)n_fft * a->Kpad * 4) != hipSuccess ||
      hipMalloc(&a->fb, (size_t)a->bins * n_mels * 4) != hipSuccess) { dmx_set_error("hipMalloc failed"); delete a; return nullptr; }
  hipMemcpy(a->fb, fb_host, (size_t)a->bins * n_mels * 4, hipMemcpyHostToDevice);
  dmx_stft_tables(a->table, a->tableT, n_fft, a->bins, a->Npad, a->Kpad, window_hann, nullptr);
  hipDeviceSynchronize();
  return a;
}
void dmx_audio_destroy(dmx_audio* a) {
  if (!a) return;
  hipFree(a->table); hipFree(a->tableT); hipFree(a->fb);
  delete a;
}
int dmx_audio_num_frames(const dmx_audio* a, int L) { return frames_of(a, L); }
int dmx_audio_num_bins(const dmx_audio* a) { return a->bins; }
size_t dmx_audio_state_bytes(const dmx_audio* a, int batch, int L) {
  const size_t M = (size_t)batch * frames_of(a, L);
  return 4 * (align_up(M * a->Npad, 64) + align_up(M * a->n_mels, 64) + align_up(M * a->Kpad, 64) + align_up(M * a->n_fft, 64));
}
int dmx_audio_transform_fwd(dmx_audio* a, const float* wav, long long wav_stride, float* mel_out, void* state, int batch, int L,
                            int power2, int to_db, float lo, float hi, void* stream) {
  if (L < a->n_fft / 2 + 1) { dmx_set_error("clip shorter than n_fft/2+1 (reflect pad)"); return DMX_ERR_SHAPE; }
  const int T = frames_of(a, L);
  AudioState s = carve(a, state, batch, L);
  int rc = dmx_stft_fwd(wav, wav_stride, a->table, s.X, batch, L, T, a->n_fft, a->hop, a->Npad, ST(stream));
  if (rc) return rc;
  return dmx_mel_fwd(s.X, a->fb, s.mel_lin, mel_out, batch * T, a->Npad, a->bins, a->n_mels, power2, to_db, lo, hi, ST(stream));
}
int dmx_audio_transform_bwd(dmx_audio* a, const float* dmel, float* dwav, long long dwav_stride, void* state, int batch, int L,
                            int power2, int to_db, float lo, float hi, int accumulate, void* stream) {
  const int T = frames_of(a, L);
  AudioState s = carve(a, state, batch, L);
  int rc = dmx_mel_bwd(s.X, a->fb, s.mel_lin, dmel, s.Y, batch * T, a->Npad, a->Kpad, a->bins, a->n_mels, power2, to_db, lo, hi, ST(stream));
  if (rc) return rc;
  rc = dmx_stft_bwd_frames(s.Y, a->tableT, s.dframe, batch * T, a->n_fft, a->Kpad, ST(stream));
  if (rc) return rc;
  return dmx_overlap_add(s.dframe, dwav, dwav_stride, batch, T, L, a->n_fft, a->hop, accumulate, ST(stream));
}
int dmx_audio_stft_mag(dmx_audio* a, const float* wav, long long wav_stride, float* mag, void* state, int batch, int L, void* stream) {
  const int T = frames_of(a, L);
  AudioState s = carve(a, state, batch, L);
  int rc = dmx_stft_fwd(wav, wav_stride, a->table, s.X, batch, L, T, a->n_fft, a->hop, a->Npad, ST(stream));
  if (rc) return rc;
  return dmx_stft_mag(s.X, mag, batch, T, a->bins, a->Npad, ST(stream));
}
int dmx_audio_stft_mag_bwd(dmx_audio* a, const float* dmag, float* dwav, long long dwav_stride, void* state, int batch, int L,
                           int accumulate, void* stream) {
  const int T = frames_of(a, L);
  AudioState s = carve(a, state, batch, L);          // s.X: spectrum kept by the last dmx_audio_stft_mag on this state
  int rc = dmx_stft_mag_bwd(s.X, dmag, s.Y, batch, T, a->bins, a->Npad, a->Kpad, ST(stream));
  if (rc) return rc;
  rc = dmx_stft_bwd_frames(s.Y, a->tableT, s.dframe, batch * T, a->n_fft, a->Kpad, ST(stream));
  if (rc) return rc;
  return dmx_overlap_add(s.dframe, dwav, dwav_stride, batch, T, L, a->n_fft, a->hop, accumulate, ST(stream));
}
int dmx_audio_melscale(dmx_audio* a, const float* mag, float* mel_out, int batch, int T, float lo, float hi, void* stream) {
  return dmx_melscale(mag, a->fb, mel_out, batch, T, a->bins, a->n_mels, lo, hi, ST(stream));
}
int dmx_mask_apply(const float* x, long long x_stride, const float* mask, float* y, long long y_stride, int batch, int L, int Ly,
                   void* stream) {
  return dmx_mask_mul(x, x_stride, mask, y, y_stride, batch, L, Ly, ST(stream));
}
int dmx_l2_loss(const float* ref, long long ref_stride, const float* pred, float* loss, float* dpred, int batch, long long n,
                float gscale, void* stream) {
  return dmx_l2_loss_grad(ref, pred, loss, dpred, batch, n, ref_stride, gscale, ST(stream));
}
int dmx_grad_normalize(float* x, float* inv_scale, int batch, long long n, float target, void* stream) {
  return dmx_absmax_normalize(x, inv_scale, batch, n, target, ST(stream));
}
int dmx_sched_pred_x0(const float* x, const float* eps, float* x0, long long n, float alpha_t, void* stream) {
  return dmx_pred_x0(x, eps, x0, n, sqrtf(alpha_t), sqrtf(1.f - alpha_t), ST(stream));
}
int dmx_sched_cfg_combine(const float* eps2, float* out, long long n, float scale, void* stream) {
  return dmx_cfg_combine(eps2, out, n, scale, ST(stream));
}
int dmx_sched_step(int mode, const float* x, const float* eps, const float* x0, const float* g0, const float* inv_scale,
                   const float* noise, float* prev, float* x0_out, float* grad_out, int batch, int n, float alpha_t, float alpha_prev,
                   float sigma, float rate, float eps_small, int global_norm, void* stream) {
  return dmx_sched_update(mode, x, eps, x0, g0, inv_scale, noise, prev, x0_out, grad_out, batch, n, alpha_t, alpha_prev, sigma, rate,
                          eps_small, global_norm, ST(stream));
}

}  // extern "C"
