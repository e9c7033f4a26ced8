// extern "C" entry points of the STFT / mel measurement path and the scheduler updates.
#include "kernels.h"
#include "../../include/diffmusic_hip.h"
#include <vector>
#include <cmath>
void dmx_set_error(const char* fmt, ...);
#define ST(s) ((hipStream_t)(s))

struct dmx_audio {
  int n_fft, hop, bins, n_mels, hann, Npad, Kpad;
  float *table, *tableT, *fb;
  // fused STFT -> mel path (stft_mel.hip): n_fft = 1024 only; tables built once from the same window / filterbank
  bool fused = false;
  void* fused_mem = nullptr;
  DmxStftMelTables ft;
};

// twiddles, window and the two compacted views of the (bins x 64) filterbank for stft_mel.hip, in ONE device allocation
static bool build_fused_tables(dmx_audio* a, const float* fb) {
  const int N = a->n_fft, NBv = a->bins, NMv = a->n_mels, NBP = 576;
  if (N != 1024 || NMv != 64) return false;
  std::vector<int> klo(NMv, 0), klen(NMv, 0), mlo(NBP, 0), mlen(NBP, 0);
  int kmax = 1, mmax = 1;
  for (int m = 0; m < NMv; ++m) {
    int lo = -1, hi = -1;
    for (int k = 0; k < NBv; ++k) if (fb[(size_t)k * NMv + m] != 0.f) { if (lo < 0) lo = k; hi = k; }
    if (lo >= 0) { klo[m] = lo; klen[m] = hi - lo + 1; if (klen[m] > kmax) kmax = klen[m]; }
  }
  for (int k = 0; k < NBv; ++k) {
    int lo = -1, hi = -1;
    for (int m = 0; m < NMv; ++m) if (fb[(size_t)k * NMv + m] != 0.f) { if (lo < 0) lo = m; hi = m; }
    if (lo >= 0) { mlo[k] = lo; mlen[k] = hi - lo + 1; if (mlen[k] > mmax) mmax = mlen[k]; }
  }
  std::vector<float> fbc((size_t)kmax * NMv, 0.f), fbr((size_t)mmax * NBP, 0.f), win(N), tw(2 * (size_t)N);
  for (int m = 0; m < NMv; ++m) for (int i = 0; i < klen[m]; ++i) fbc[(size_t)i * NMv + m] = fb[(size_t)(klo[m] + i) * NMv + m];
  for (int k = 0; k < NBv; ++k) for (int i = 0; i < mlen[k]; ++i) fbr[(size_t)i * NBP + k] = fb[(size_t)k * NMv + mlo[k] + i];
  const double two_pi = 6.283185307179586476925286766559;
  for (int n = 0; n < N; ++n) {
    win[n] = a->hann ? (float)(0.5 - 0.5 * cos(two_pi * n / N)) : 1.f;       // periodic Hann (torch.hann_window default) / rectangular
    tw[2 * n] = (float)cos(two_pi * n / N); tw[2 * n + 1] = (float)(-sin(two_pi * n / N));
  }
  size_t off = 0;
  auto place = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t o_tw = place(tw.size() * 4), o_win = place(win.size() * 4), o_klo = place(NMv * 4), o_klen = place(NMv * 4),
               o_mlo = place(NBP * 4), o_mlen = place(NBP * 4), o_fbc = place(fbc.size() * 4), o_fbr = place(fbr.size() * 4);
  char* d = nullptr;
  if (hipMalloc(&d, off) != hipSuccess) return false;
  // every table upload is checked: a failed copy returns false and the dense-DFT path is used (a half-filled table must never be read)
  const struct { size_t off; const void* src; size_t bytes; } ups[] = {
      {o_tw, tw.data(), tw.size() * 4}, {o_win, win.data(), win.size() * 4}, {o_klo, klo.data(), (size_t)NMv * 4}, {o_klen, klen.data(), (size_t)NMv * 4},
      {o_mlo, mlo.data(), (size_t)NBP * 4}, {o_mlen, mlen.data(), (size_t)NBP * 4}, {o_fbc, fbc.data(), fbc.size() * 4}, {o_fbr, fbr.data(), fbr.size() * 4}};
  for (const auto& u : ups)
    if (hipMemcpy(d + u.off, u.src, u.bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return false; }
  a->fused_mem = d;
  a->ft.tw = (const float2*)(d + o_tw); a->ft.win = (const float*)(d + o_win);
  a->ft.klo = (const int*)(d + o_klo); a->ft.klen = (const int*)(d + o_klen); a->ft.fbc = (const float*)(d + o_fbc); a->ft.kmax = kmax;
  a->ft.mlo = (const int*)(d + o_mlo); a->ft.mlen = (const int*)(d + o_mlen); a->ft.fbr = (const float*)(d + o_fbr); a->ft.mmax = mmax;
  return true;
}
// state layout of the fused path: [waveform kept for transform_bwd: B x L][per-workgroup partial sums of the loss: B x parts]
static float* fused_stash(void* state) { return (float*)state; }
static float* fused_partials(const dmx_audio* a, void* state, int B, int L) { return (float*)state + align_up((size_t)B * L, 64); }

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
  a->fused = build_fused_tables(a, fb_host);
  hipDeviceSynchronize();
  return a;
}
void dmx_audio_destroy(dmx_audio* a) {
  if (!a) return;
  hipFree(a->table); hipFree(a->tableT); hipFree(a->fb);
  if (a->fused_mem) hipFree(a->fused_mem);
  delete a;
}
int dmx_audio_num_frames(const dmx_audio* a, int L) { return frames_of(a, L); }
int dmx_audio_num_bins(const dmx_audio* a) { return a->bins; }
size_t dmx_audio_state_bytes(const dmx_audio* a, int batch, int L) {
  const size_t M = (size_t)batch * frames_of(a, L);
  const size_t dense = 4 * (align_up(M * a->Npad, 64) + align_up(M * a->n_mels, 64) + align_up(M * a->Kpad, 64) + align_up(M * a->n_fft, 64));
  const size_t fused = 4 * (align_up((size_t)batch * L, 64) + align_up((size_t)batch * dmx_stft_mel_parts(L, a->hop), 64));
  return dense > fused ? dense : fused;        // (|STFT| operator calls and n_fft != 1024 use the dense-DFT layout)
}
int dmx_audio_is_fused(const dmx_audio* a, int L) { return a->fused && L >= 2 * a->n_fft ? 1 : 0; }
int dmx_audio_transform_fwd(dmx_audio* a, const float* wav, long long wav_stride, float* mel_out, void* state, int batch, int L,
                            int power2, int to_db, float lo, float hi, void* stream) {
  if (L < a->n_fft / 2 + 1) { dmx_set_error("clip shorter than n_fft/2+1 (reflect pad)"); return DMX_ERR_SHAPE; }
  const int T = frames_of(a, L);
  if (dmx_audio_is_fused(a, L)) {
    // fused path: the spectrum is never stored; transform_bwd re-derives it from a copy of the waveform kept in `state`
    if (hipMemcpy2DAsync(fused_stash(state), (size_t)L * 4, wav, (size_t)wav_stride * 4, (size_t)L * 4, batch, hipMemcpyDeviceToDevice,
                         ST(stream)) != hipSuccess) return DMX_ERR_LAUNCH;
    return dmx_stft_mel_fwd(a->ft, wav, wav_stride, nullptr, nullptr, 0, mel_out, nullptr, batch, L, a->hop, power2, to_db, lo, hi, ST(stream));
  }
  AudioState s = carve(a, state, batch, L);
  int rc = dmx_stft_fwd(wav, wav_stride, a->table, s.X, batch, L, T, a->n_fft, a->hop, a->Npad, ST(stream));
  if (rc) return rc;
  return dmx_mel_fwd(s.X, a->fb, s.mel_lin, mel_out, batch * T, a->Npad, a->bins, a->n_mels, power2, to_db, lo, hi, ST(stream));
}
int dmx_audio_guidance_fwd(dmx_audio* a, const float* wav, long long wav_stride, const float* mask, const float* ref, long long ref_stride,
                           float* mel_out, void* state, int batch, int L, int power2, int to_db, float lo, float hi, void* stream) {
  if (!dmx_audio_is_fused(a, L)) { dmx_set_error("fused guidance needs n_fft = 1024, 64 mel columns and a clip of >= 2048 samples"); return DMX_ERR_SHAPE; }
  if (!ref) { dmx_set_error("guidance_fwd needs the reference transform"); return DMX_ERR_SHAPE; }
  return dmx_stft_mel_fwd(a->ft, wav, wav_stride, mask, ref, ref_stride, mel_out, fused_partials(a, state, batch, L), batch, L, a->hop,
                          power2, to_db, lo, hi, ST(stream));
}
int dmx_audio_guidance_bwd(dmx_audio* a, const float* wav, long long wav_stride, const float* mask, const float* ref, long long ref_stride,
                           float gscale, float* loss, float* dwav, long long dwav_stride, int Lfull, void* state, int batch, int L,
                           int power2, int to_db, float lo, float hi, void* stream) {
  if (!dmx_audio_is_fused(a, L)) { dmx_set_error("fused guidance needs n_fft = 1024, 64 mel columns and a clip of >= 2048 samples"); return DMX_ERR_SHAPE; }
  return dmx_stft_mel_bwd(a->ft, wav, wav_stride, mask, ref, ref_stride, nullptr, fused_partials(a, state, batch, L), gscale, loss, dwav,
                          dwav_stride, Lfull, 0, batch, L, a->hop, power2, to_db, lo, hi, ST(stream));
}
int dmx_audio_transform_bwd(dmx_audio* a, const float* dmel, float* dwav, long long dwav_stride, void* state, int batch, int L,
                            int power2, int to_db, float lo, float hi, int accumulate, void* stream) {
  const int T = frames_of(a, L);
  if (dmx_audio_is_fused(a, L))
    return dmx_stft_mel_bwd(a->ft, fused_stash(state), L, nullptr, nullptr, 0, dmel, nullptr, 1.f, nullptr, dwav, dwav_stride, L, accumulate,
                            batch, L, a->hop, power2, to_db, lo, hi, ST(stream));
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
int dmx_sched_pred_x0_ex(const float* x, const float* model_output, float* x0, long long n, float alpha_t, int prediction_type, float clip_range,
                         void* stream) {
  const int rc = dmx_pred_x0_ex(x, model_output, x0, n, sqrtf(alpha_t), sqrtf(1.f - alpha_t), prediction_type, clip_range, ST(stream));
  if (rc == DMX_ERR_SHAPE) dmx_set_error("sched_pred_x0_ex: prediction_type 0 (epsilon), 1 (sample) or 2 (v_prediction)");
  return rc;
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
int dmx_sched_step_ex(int mode, const float* x, const float* eps, const float* x0, const float* g0, const float* inv_scale,
                      const float* noise, float* prev, float* x0_out, float* grad_out, int batch, int n, float alpha_t, float alpha_prev,
                      float sigma, float rate, float eps_small, int global_norm, int prediction_type, float clip_range, void* stream) {
  return dmx_sched_update(mode, x, eps, x0, g0, inv_scale, noise, prev, x0_out, grad_out, batch, n, alpha_t, alpha_prev, sigma, rate,
                          eps_small, global_norm, ST(stream), prediction_type, clip_range);
}

}  // extern "C"
