// TORCH_LIBRARY(diffmusic_hip, m): the PyTorch-ROCm custom-op layer over the C ABI of libdiffmusic_hip.so (SURVEY.md section
// 8b.4).  Every op is a thin wrapper: it checks device / dtype / contiguity, allocates the outputs with torch's caching
// allocator, takes torch's CURRENT HIP stream and calls the `extern "C"` launcher declared in include/diffmusic_hip.h -- the
// launchers never allocate, synchronise or throw, so an op is one stream-ordered enqueue.  A non-zero return code becomes a
// c10::Error carrying dmx_last_error().  Network / audio handles (dmx_model*, dmx_audio*) travel as int64 values obtained
// from the *_create calls of the C ABI (diffmusic_amd/engine.py owns them).
//
// Callers on the reference side are its three protocols (Scheduler.step, Pipeline.__call__, BaseOperator); the reference
// lines each op replaces are cited at the C-ABI declarations.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // torch-ROCm tensors report DeviceType::CUDA: the guard / stream types that accept it
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/diffmusic_hip.h"

namespace {

// every op makes its first tensor's device current for its own duration, so the outputs are allocated there and the launch goes to
// THAT device's current stream (a tensor on a non-current device must not be enqueued on the current device's stream)
#define DMX_DEVICE_OF(t) const c10::hip::HIPGuardMasqueradingAsCUDA dmx_guard_((t).device())
inline void* cur_stream() { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }
inline void same_numel(const at::Tensor& a, const at::Tensor& b, const char* what) {
  TORCH_CHECK(a.numel() == b.numel() && a.device() == b.device(), what, ": size / device mismatch");
}

inline void ok(int rc, const char* what) {
  TORCH_CHECK(rc == 0, "diffmusic_hip::", what, " failed with code ", rc, ": ", dmx_last_error());
}
inline void f32_cuda(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be a GPU tensor (the diffmusic_hip ops have no CPU fallback)");
  TORCH_CHECK(t.scalar_type() == at::kFloat && t.is_contiguous(), name, " must be contiguous fp32");
}
inline void act_cuda(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.is_contiguous(), name, " must be a contiguous GPU tensor");
  TORCH_CHECK(t.scalar_type() == (dmx_act_dtype() == 1 ? at::kHalf : at::kBFloat16), name, " must have the library's 16-bit activation dtype");
}
inline const float* fp(const std::optional<at::Tensor>& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }

// ---- scheduler arithmetic (scheduling_{ddim,dps,mpgd,dsg,diffmusic}.py step bodies; pipeline_musicldm.py:706-708)
at::Tensor sched_pred_x0(const at::Tensor& x, const at::Tensor& eps, double alpha_t) {
  f32_cuda(x, "x"); f32_cuda(eps, "eps");
  same_numel(x, eps, "sched_pred_x0(x, eps)");
  DMX_DEVICE_OF(x);
  at::Tensor x0 = at::empty_like(x);
  ok(dmx_sched_pred_x0(x.data_ptr<float>(), eps.data_ptr<float>(), x0.data_ptr<float>(), x.numel(), (float)alpha_t, cur_stream()), "sched_pred_x0");
  return x0;
}
at::Tensor cfg_combine(const at::Tensor& eps2, double scale) {
  f32_cuda(eps2, "eps2");
  TORCH_CHECK(eps2.dim() >= 1 && eps2.size(0) % 2 == 0, "eps2 must hold the [uncond | text] halves of the CFG batch");
  DMX_DEVICE_OF(eps2);
  auto sizes = eps2.sizes().vec();
  sizes[0] /= 2;
  at::Tensor out = at::empty(sizes, eps2.options());
  ok(dmx_sched_cfg_combine(eps2.data_ptr<float>(), out.data_ptr<float>(), out.numel(), (float)scale, cur_stream()), "cfg_combine");
  return out;
}
// returns (prev_sample, x0_updated): the second is MPGD's guided x0 (scheduling_mpgd.py:202) and None for every other mode (no output
// aliases an input: the schema is a plain functional op)
std::tuple<at::Tensor, std::optional<at::Tensor>> sched_update(int64_t mode, const at::Tensor& x, const at::Tensor& eps, const at::Tensor& x0,
                                                const std::optional<at::Tensor>& g0, const std::optional<at::Tensor>& inv_scale,
                                                const std::optional<at::Tensor>& noise, double alpha_t, double alpha_prev, double sigma,
                                                double rate, double eps_small, bool global_norm) {
  f32_cuda(x, "x"); f32_cuda(eps, "eps"); f32_cuda(x0, "x0");
  TORCH_CHECK(x.dim() >= 1 && x.size(0) >= 1, "x must have a batch dimension");
  same_numel(x, eps, "sched_update(x, eps)"); same_numel(x, x0, "sched_update(x, x0)");
  const int B = (int)x.size(0);
  const int n = (int)(x.numel() / B);
  if (g0) { f32_cuda(*g0, "g0"); same_numel(x, *g0, "sched_update(x, g0)"); }
  if (inv_scale) { f32_cuda(*inv_scale, "inv_scale"); TORCH_CHECK(inv_scale->numel() == B && inv_scale->device() == x.device(), "inv_scale must hold one value per clip"); }
  if (noise) { f32_cuda(*noise, "noise"); same_numel(x, *noise, "sched_update(x, noise)"); }
  TORCH_CHECK(mode == DMX_SCHED_DDIM || (g0 && inv_scale), "guided modes need g0 and inv_scale");
  DMX_DEVICE_OF(x);
  at::Tensor prev = at::empty_like(x);
  std::optional<at::Tensor> x0_out;
  if (mode == DMX_SCHED_MPGD) x0_out = at::empty_like(x);
  ok(dmx_sched_step((int)mode, x.data_ptr<float>(), eps.data_ptr<float>(), x0.data_ptr<float>(), fp(g0), fp(inv_scale), fp(noise),
                    prev.data_ptr<float>(), x0_out ? x0_out->data_ptr<float>() : nullptr, nullptr, B, n, (float)alpha_t,
                    (float)alpha_prev, (float)sigma, (float)rate, (float)eps_small, global_norm ? 1 : 0, cur_stream()), "sched_update");
  return {prev, x0_out};
}
at::Tensor randn_philox(at::IntArrayRef shape, at::IntArrayRef seeds, int64_t offset, at::Device device) {
  TORCH_CHECK(!shape.empty() && (int64_t)seeds.size() == shape[0] && shape[0] <= 64, "need one seed per clip (batch <= 64)");
  TORCH_CHECK(device.is_cuda(), "randn_philox draws on the GPU");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(device);
  at::Tensor out = at::empty(shape, at::TensorOptions().dtype(at::kFloat).device(device));
  std::vector<unsigned long long> s(seeds.begin(), seeds.end());
  ok(dmx_randn_philox(out.data_ptr<float>(), (int)shape[0], out.numel() / shape[0], s.data(), (unsigned long long)offset, cur_stream()), "randn_philox");
  return out;
}

// ---- measurement operators (operator.py) and the loss (scheduling_dps.py:211)
at::Tensor mask_mul(const at::Tensor& x, const std::optional<at::Tensor>& mask, int64_t L, int64_t Ly) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 2 && x.stride(1) == 1, "x must be (B, >= L) fp32 on the GPU");
  DMX_DEVICE_OF(x);
  if (mask) f32_cuda(*mask, "mask");
  at::Tensor y = at::empty({x.size(0), Ly}, x.options());
  ok(dmx_mask_apply(x.data_ptr<float>(), x.stride(0), fp(mask), y.data_ptr<float>(), Ly, (int)x.size(0), (int)L, (int)Ly, cur_stream()), "mask_mul");
  return y;
}
std::tuple<at::Tensor, at::Tensor> l2norm(const at::Tensor& ref, const at::Tensor& pred, double gscale) {
  f32_cuda(ref, "ref"); f32_cuda(pred, "pred");
  DMX_DEVICE_OF(pred);
  const int B = (int)pred.size(0);
  const long long n = pred.numel() / B;
  TORCH_CHECK((ref.numel() == n || ref.numel() == n * B) && ref.device() == pred.device(), "ref must match pred or broadcast over the batch");
  at::Tensor loss = at::empty({B}, pred.options()), dpred = at::empty_like(pred);
  ok(dmx_l2_loss(ref.data_ptr<float>(), ref.numel() == n && B > 1 ? 0 : n, pred.data_ptr<float>(), loss.data_ptr<float>(), dpred.data_ptr<float>(),
                 B, n, (float)gscale, cur_stream()), "l2norm");
  return {loss, dpred};
}
at::Tensor resample_fwd(const at::Tensor& x, const at::Tensor& h, int64_t Lin, int64_t Lout, int64_t orig, int64_t new_, int64_t off) {
  TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat && x.dim() == 2 && x.stride(1) == 1, "x must be (B, >= Lin) fp32 on the GPU");
  DMX_DEVICE_OF(x);
  f32_cuda(h, "h");
  at::Tensor y = at::empty({x.size(0), Lout}, x.options());
  ok(dmx_fir_fwd(x.data_ptr<float>(), x.stride(0), h.data_ptr<float>(), y.data_ptr<float>(), Lout, (int)x.size(0), (int)Lin, (int)Lout,
                 (int)h.size(-1), (int)orig, (int)new_, (int)off, cur_stream()), "resample_fwd");
  return y;
}
at::Tensor resample_bwd(const at::Tensor& dy, const at::Tensor& h, const std::optional<at::Tensor>& h_rev, int64_t Lin, int64_t Lfull,
                        int64_t orig, int64_t new_, int64_t off) {
  f32_cuda(dy, "dy"); f32_cuda(h, "h");
  DMX_DEVICE_OF(dy);
  at::Tensor d = at::zeros({dy.size(0), Lfull}, dy.options());
  ok(dmx_fir_bwd(dy.data_ptr<float>(), dy.size(1), h.data_ptr<float>(), fp(h_rev), d.data_ptr<float>(), Lfull, (int)dy.size(0), (int)Lin,
                 (int)dy.size(1), (int)h.size(-1), (int)orig, (int)new_, (int)off, cur_stream()), "resample_bwd");
  return d;
}
at::Tensor logmel_fwd(int64_t audio, const at::Tensor& wav, const at::Tensor& state, int64_t L, bool power2, bool to_db, double lo, double hi) {
  dmx_audio* a = reinterpret_cast<dmx_audio*>(audio);
  TORCH_CHECK(wav.is_cuda() && wav.scalar_type() == at::kFloat && wav.dim() == 2 && wav.stride(1) == 1, "wav must be (B, >= L) fp32 on the GPU");
  DMX_DEVICE_OF(wav);
  TORCH_CHECK(state.is_cuda() && (size_t)state.nbytes() >= dmx_audio_state_bytes(a, (int)wav.size(0), (int)L), "state buffer too small");
  at::Tensor mel = at::empty({wav.size(0), dmx_audio_num_frames(a, (int)L), 64}, wav.options());
  ok(dmx_audio_transform_fwd(a, wav.data_ptr<float>(), wav.stride(0), mel.data_ptr<float>(), state.data_ptr(), (int)wav.size(0), (int)L,
                             power2, to_db, (float)lo, (float)hi, cur_stream()), "logmel_fwd");
  return mel;
}
at::Tensor logmel_bwd(int64_t audio, const at::Tensor& dmel, const at::Tensor& state, int64_t L, bool power2, bool to_db, double lo, double hi) {
  dmx_audio* a = reinterpret_cast<dmx_audio*>(audio);
  f32_cuda(dmel, "dmel");
  DMX_DEVICE_OF(dmel);
  at::Tensor dwav = at::empty({dmel.size(0), L}, dmel.options());
  ok(dmx_audio_transform_bwd(a, dmel.data_ptr<float>(), dwav.data_ptr<float>(), L, state.data_ptr(), (int)dmel.size(0), (int)L, power2, to_db,
                             (float)lo, (float)hi, 0, cur_stream()), "logmel_bwd");
  return dwav;
}
// fused guidance of the mel-space operators (include/diffmusic_hip.h dmx_audio_guidance_{fwd,bwd}): (loss (B), dwav (B, Lfull))
std::tuple<at::Tensor, at::Tensor> mel_guidance(int64_t audio, const at::Tensor& wav, const std::optional<at::Tensor>& mask, const at::Tensor& ref,
                                                at::Tensor state, int64_t L, int64_t Lfull, bool power2, bool to_db, double lo, double hi, double gscale) {
  dmx_audio* a = reinterpret_cast<dmx_audio*>(audio);
  TORCH_CHECK(wav.is_cuda() && wav.scalar_type() == at::kFloat && wav.dim() == 2 && wav.stride(1) == 1 && wav.size(1) >= L, "wav must be (B, >= L) fp32 on the GPU");
  DMX_DEVICE_OF(wav);
  f32_cuda(ref, "ref");
  if (mask) { f32_cuda(*mask, "mask"); TORCH_CHECK(mask->numel() >= L && mask->device() == wav.device(), "mask must hold L samples"); }
  const int B = (int)wav.size(0), T = dmx_audio_num_frames(a, (int)L);
  TORCH_CHECK(ref.device() == wav.device() && (ref.numel() == (int64_t)T * 64 || ref.numel() == (int64_t)B * T * 64), "ref must be (B or 1, frames, 64)");
  TORCH_CHECK(state.is_cuda() && (size_t)state.nbytes() >= dmx_audio_state_bytes(a, B, (int)L), "state buffer too small");
  TORCH_CHECK(Lfull >= L, "Lfull < L");
  const long long rs = ref.numel() == (int64_t)T * 64 && B > 1 ? 0 : (long long)T * 64;
  at::Tensor loss = at::empty({B}, wav.options()), dwav = at::empty({B, Lfull}, wav.options());
  ok(dmx_audio_guidance_fwd(a, wav.data_ptr<float>(), wav.stride(0), fp(mask), ref.data_ptr<float>(), rs, nullptr, state.data_ptr(), B, (int)L,
                            power2, to_db, (float)lo, (float)hi, cur_stream()), "mel_guidance (forward)");
  ok(dmx_audio_guidance_bwd(a, wav.data_ptr<float>(), wav.stride(0), fp(mask), ref.data_ptr<float>(), rs, (float)gscale, loss.data_ptr<float>(),
                            dwav.data_ptr<float>(), Lfull, (int)Lfull, state.data_ptr(), B, (int)L, power2, to_db, (float)lo, (float)hi, cur_stream()),
     "mel_guidance (backward)");
  return {loss, dwav};
}
at::Tensor stft_mag_fwd(int64_t audio, const at::Tensor& wav, const at::Tensor& state, int64_t L) {
  dmx_audio* a = reinterpret_cast<dmx_audio*>(audio);
  TORCH_CHECK(wav.is_cuda() && wav.scalar_type() == at::kFloat && wav.dim() == 2 && wav.stride(1) == 1, "wav must be (B, >= L) fp32 on the GPU");
  DMX_DEVICE_OF(wav);
  const int T = dmx_audio_num_frames(a, (int)L);
  TORCH_CHECK((size_t)state.nbytes() >= dmx_audio_state_bytes(a, (int)wav.size(0), (int)L), "state buffer too small");
  at::Tensor mag = at::empty({wav.size(0), dmx_audio_num_bins(a), T}, wav.options());
  ok(dmx_audio_stft_mag(a, wav.data_ptr<float>(), wav.stride(0), mag.data_ptr<float>(), state.data_ptr(), (int)wav.size(0), (int)L, cur_stream()), "stft_mag_fwd");
  return mag;
}
at::Tensor stft_mag_bwd(int64_t audio, const at::Tensor& dmag, const at::Tensor& state, int64_t L, int64_t Lfull) {
  dmx_audio* a = reinterpret_cast<dmx_audio*>(audio);
  f32_cuda(dmag, "dmag");
  DMX_DEVICE_OF(dmag);
  at::Tensor dwav = at::zeros({dmag.size(0), Lfull}, dmag.options());
  ok(dmx_audio_stft_mag_bwd(a, dmag.data_ptr<float>(), dwav.data_ptr<float>(), Lfull, state.data_ptr(), (int)dmag.size(0), (int)L, 0, cur_stream()), "stft_mag_bwd");
  return dwav;
}
at::Tensor melscale_fwd(int64_t audio, const at::Tensor& mag, double lo, double hi) {
  dmx_audio* a = reinterpret_cast<dmx_audio*>(audio);
  f32_cuda(mag, "mag");
  DMX_DEVICE_OF(mag);
  at::Tensor mel = at::empty({mag.size(0), mag.size(2), 64}, mag.options());
  ok(dmx_audio_melscale(a, mag.data_ptr<float>(), mel.data_ptr<float>(), (int)mag.size(0), (int)mag.size(2), (float)lo, (float)hi, cur_stream()), "melscale_fwd");
  return mel;
}

// ---- networks (handles from dmx_*_create; workspaces are caller-owned byte tensors sized by *_workspace_bytes)
at::Tensor unet_fwd(int64_t model, const at::Tensor& x, const at::Tensor& t, const std::optional<at::Tensor>& class_labels, at::Tensor ws) {
  f32_cuda(x, "x"); f32_cuda(t, "t");
  DMX_DEVICE_OF(x);
  if (class_labels) f32_cuda(*class_labels, "class_labels");
  at::Tensor eps = at::empty_like(x);
  ok(dmx_unet_fwd(reinterpret_cast<dmx_model*>(model), x.data_ptr<float>(), t.data_ptr<float>(), fp(class_labels), eps.data_ptr<float>(),
                  (int)x.size(0), (int)x.size(2), (int)x.size(3), ws.data_ptr(), ws.nbytes(), cur_stream()), "unet_fwd");
  return eps;
}
// AudioLDM2UNet2DConditionModel call (plpeline_audioldm2.py:1147-1154): GPT-2 states c0 (B, n0, d0), T5 states c1 (B, n1, d1) + additive key bias
at::Tensor unet_fwd_ctx(int64_t model, const at::Tensor& x, const at::Tensor& t, const std::optional<at::Tensor>& class_labels,
                        const at::Tensor& c0, const at::Tensor& c1, const at::Tensor& bias1, at::Tensor ws) {
  f32_cuda(x, "x"); f32_cuda(t, "t"); f32_cuda(c0, "encoder_hidden_states"); f32_cuda(c1, "encoder_hidden_states_1"); f32_cuda(bias1, "bias1");
  if (class_labels) f32_cuda(*class_labels, "class_labels");
  TORCH_CHECK(c0.dim() == 3 && c1.dim() == 3 && c0.size(0) == x.size(0) && c1.size(0) == x.size(0) && bias1.numel() == c1.size(0) * c1.size(1),
              "context tensors must be (B, tokens, dim) with one additive bias per T5 token");
  DMX_DEVICE_OF(x);
  at::Tensor eps = at::empty_like(x);
  ok(dmx_unet_fwd_ctx(reinterpret_cast<dmx_model*>(model), x.data_ptr<float>(), t.data_ptr<float>(), fp(class_labels), c0.data_ptr<float>(),
                      (int)c0.size(1), c1.data_ptr<float>(), (int)c1.size(1), bias1.data_ptr<float>(), eps.data_ptr<float>(), (int)x.size(0),
                      (int)x.size(2), (int)x.size(3), ws.data_ptr(), ws.nbytes(), cur_stream()), "unet_fwd_ctx");
  return eps;
}
// returns (mel 16-bit, mel fp32 or None): `vae.decode(z).sample` (scheduling_dps.py:195-197)
std::tuple<at::Tensor, std::optional<at::Tensor>> vae_dec_fwd(int64_t model, const at::Tensor& z, double z_scale, bool keep_state, bool want_f32,
                                                              int64_t scale_factor, at::Tensor ws) {
  f32_cuda(z, "z");
  DMX_DEVICE_OF(z);
  const int B = (int)z.size(0), h = (int)z.size(2), w = (int)z.size(3);
  at::Tensor mel = at::empty({B, scale_factor * h, scale_factor * w}, z.options().dtype(dmx_act_dtype() == 1 ? at::kHalf : at::kBFloat16));
  std::optional<at::Tensor> mel32;
  if (want_f32) mel32 = at::empty({B, scale_factor * h, scale_factor * w}, z.options());
  ok(dmx_vae_decode_fwd(reinterpret_cast<dmx_model*>(model), z.data_ptr<float>(), (float)z_scale, (uint16_t*)mel.data_ptr(),
                        mel32 ? mel32->data_ptr<float>() : nullptr, B, h, w, keep_state, ws.data_ptr(), ws.nbytes(), cur_stream()), "vae_dec_fwd");
  return {mel, mel32};
}
// per-clip rescale of the waveform gradient before the 16-bit backward sweep (max |g| -> target), IN PLACE; returns the factors that undo it
at::Tensor grad_normalize_(at::Tensor dwav, double target) {
  f32_cuda(dwav, "dwav");
  TORCH_CHECK(dwav.dim() == 2, "dwav must be (B, samples)");
  DMX_DEVICE_OF(dwav);
  at::Tensor inv = at::empty({dwav.size(0)}, dwav.options());
  ok(dmx_grad_normalize(dwav.data_ptr<float>(), inv.data_ptr<float>(), (int)dwav.size(0), dwav.size(1), (float)target, cur_stream()), "grad_normalize");
  return inv;
}
at::Tensor vae_dec_bwd(int64_t model, const at::Tensor& dmel, double z_scale, int64_t latent_channels, int64_t scale_factor) {
  act_cuda(dmel, "dmel");
  DMX_DEVICE_OF(dmel);
  at::Tensor dz = at::empty({dmel.size(0), latent_channels, dmel.size(1) / scale_factor, dmel.size(2) / scale_factor}, dmel.options().dtype(at::kFloat));
  ok(dmx_vae_decode_bwd(reinterpret_cast<dmx_model*>(model), (const uint16_t*)dmel.data_ptr(), (float)z_scale, dz.data_ptr<float>(), cur_stream()), "vae_dec_bwd");
  return dz;
}
at::Tensor hifigan_fwd(int64_t model, const at::Tensor& mel, at::Tensor ws) {
  act_cuda(mel, "mel");
  DMX_DEVICE_OF(mel);
  dmx_model* m = reinterpret_cast<dmx_model*>(model);
  const int B = (int)mel.size(0), T = (int)mel.size(1);
  at::Tensor wav = at::empty({B, dmx_hifigan_out_len(m, T)}, mel.options().dtype(at::kFloat));
  ok(dmx_hifigan_fwd(m, (const uint16_t*)mel.data_ptr(), wav.data_ptr<float>(), B, T, ws.data_ptr(), ws.nbytes(), cur_stream()), "hifigan_fwd");
  return wav;
}
at::Tensor hifigan_bwd(int64_t model, const at::Tensor& dwav, int64_t frames, int64_t model_in_dim) {
  f32_cuda(dwav, "dwav");
  DMX_DEVICE_OF(dwav);
  at::Tensor dmel = at::empty({dwav.size(0), frames, model_in_dim}, dwav.options().dtype(dmx_act_dtype() == 1 ? at::kHalf : at::kBFloat16));
  ok(dmx_hifigan_bwd(reinterpret_cast<dmx_model*>(model), dwav.data_ptr<float>(), (uint16_t*)dmel.data_ptr(), cur_stream()), "hifigan_bwd");
  return dmel;
}

// ---- CLAP HTS-AT tower of the style-guidance operator (operator.py:253-271) and the Gram matrix of its token features
at::Tensor htsat_fwd(int64_t model, const at::Tensor& mel, bool keep_state, at::Tensor ws) {
  f32_cuda(mel, "mel");
  TORCH_CHECK(mel.dim() == 3, "mel must be (B, frames, mel bins)");
  DMX_DEVICE_OF(mel);
  dmx_model* m = reinterpret_cast<dmx_model*>(model);
  int tokens = 0, channels = 0;
  ok(dmx_htsat_feature_dims(m, &tokens, &channels), "htsat_feature_dims");
  at::Tensor feat = at::empty({mel.size(0), tokens, channels}, mel.options());
  ok(dmx_htsat_fwd(m, mel.data_ptr<float>(), (int)mel.size(0), (int)mel.size(1), feat.data_ptr<float>(), keep_state, ws.data_ptr(), ws.nbytes(),
                   cur_stream()), "htsat_fwd");
  return feat;
}
at::Tensor htsat_bwd(int64_t model, const at::Tensor& dfeat, const std::optional<at::Tensor>& scale, int64_t frames, int64_t bins) {
  f32_cuda(dfeat, "dfeat");
  DMX_DEVICE_OF(dfeat);
  if (scale) { f32_cuda(*scale, "scale"); TORCH_CHECK(scale->numel() == dfeat.size(0) && scale->device() == dfeat.device(), "scale must hold one value per clip"); }
  at::Tensor dmel = at::empty({dfeat.size(0), frames, bins}, dfeat.options());
  ok(dmx_htsat_bwd(reinterpret_cast<dmx_model*>(model), dfeat.data_ptr<float>(), fp(scale), dmel.data_ptr<float>(), cur_stream()), "htsat_bwd");
  return dmel;
}
at::Tensor gram_fwd(const at::Tensor& feat) {
  f32_cuda(feat, "feat");
  TORCH_CHECK(feat.dim() == 3, "feat must be (B, tokens, channels)");
  DMX_DEVICE_OF(feat);
  at::Tensor g = at::empty({feat.size(0), feat.size(2), feat.size(2)}, feat.options());
  ok(dmx_gram_fwd(feat.data_ptr<float>(), g.data_ptr<float>(), (int)feat.size(0), (int)feat.size(1), (int)feat.size(2), cur_stream()), "gram_fwd");
  return g;
}
at::Tensor gram_bwd(const at::Tensor& feat, const at::Tensor& dgram) {
  f32_cuda(feat, "feat"); f32_cuda(dgram, "dgram");
  TORCH_CHECK(feat.dim() == 3 && dgram.numel() == feat.size(0) * feat.size(2) * feat.size(2) && dgram.device() == feat.device(), "dgram must be (B, C, C)");
  DMX_DEVICE_OF(feat);
  at::Tensor d = at::empty_like(feat);
  ok(dmx_gram_bwd(feat.data_ptr<float>(), dgram.data_ptr<float>(), d.data_ptr<float>(), (int)feat.size(0), (int)feat.size(1), (int)feat.size(2), cur_stream()), "gram_bwd");
  return d;
}

}  // namespace

// the version of include/diffmusic_hip.h this op library was COMPILED against (the loaded libdiffmusic_hip.so reports its own through
// dmx_abi_version(); the two are compared when the library is loaded, below)
int64_t abi_version() { return DMX_ABI_VERSION; }

TORCH_LIBRARY(diffmusic_hip, m) {
  // A stale or copied op library would call entry points of another ABI version with this one's struct layouts and argument lists
  // (version 2 changed dmx_flash_attn_raw and GemmDesc): refuse to load instead -- torch.ops.load_library raises, and
  // diffmusic_amd.ops.enabled() falls back to the ctypes binding (which checks the same number) with one warning.
  TORCH_CHECK(dmx_abi_version() == DMX_ABI_VERSION, "libdiffmusic_torch_ops.so was built against C-ABI version ", DMX_ABI_VERSION,
              " but the loaded libdiffmusic_hip.so reports ", dmx_abi_version(), ": rebuild with `python -m diffmusic_amd.build`");
  m.def("abi_version() -> int", &abi_version);
  // Schemas: ops that write into a caller-owned tensor besides their outputs declare it (a!): `state` of the measurement front end
  // (written by *_fwd / mel_guidance, read by the matching *_bwd) and the network workspaces `ws` (written by *_fwd; the model handle's
  // tape points into it and *_bwd reads it).  Handles travel as ints: effects behind a handle are invisible to the schema, so under
  // torch.compile / functionalization a forward and its backward must still be kept in program order by the caller (eager use today).
  m.def("sched_pred_x0(Tensor x, Tensor eps, float alpha_t) -> Tensor", &sched_pred_x0);
  m.def("cfg_combine(Tensor eps2, float scale) -> Tensor", &cfg_combine);
  m.def("sched_update(int mode, Tensor x, Tensor eps, Tensor x0, Tensor? g0, Tensor? inv_scale, Tensor? noise, float alpha_t, float alpha_prev, "
        "float sigma, float rate, float eps_small, bool global_norm) -> (Tensor, Tensor?)", &sched_update);
  m.def("randn_philox(int[] shape, int[] seeds, int offset, Device device) -> Tensor", &randn_philox);
  m.def("mask_mul(Tensor x, Tensor? mask, int L, int Ly) -> Tensor", &mask_mul);
  m.def("l2norm(Tensor ref, Tensor pred, float gscale) -> (Tensor, Tensor)", &l2norm);
  m.def("resample_fwd(Tensor x, Tensor h, int Lin, int Lout, int orig, int new_, int off) -> Tensor", &resample_fwd);
  m.def("resample_bwd(Tensor dy, Tensor h, Tensor? h_rev, int Lin, int Lfull, int orig, int new_, int off) -> Tensor", &resample_bwd);
  m.def("logmel_fwd(int audio, Tensor wav, Tensor(a!) state, int L, bool power2, bool to_db, float lo, float hi) -> Tensor", &logmel_fwd);
  m.def("logmel_bwd(int audio, Tensor dmel, Tensor state, int L, bool power2, bool to_db, float lo, float hi) -> Tensor", &logmel_bwd);
  m.def("mel_guidance(int audio, Tensor wav, Tensor? mask, Tensor ref, Tensor(a!) state, int L, int Lfull, bool power2, bool to_db, float lo, float hi, "
        "float gscale) -> (Tensor, Tensor)", &mel_guidance);
  m.def("stft_mag_fwd(int audio, Tensor wav, Tensor(a!) state, int L) -> Tensor", &stft_mag_fwd);
  m.def("stft_mag_bwd(int audio, Tensor dmag, Tensor state, int L, int Lfull) -> Tensor", &stft_mag_bwd);
  m.def("melscale_fwd(int audio, Tensor mag, float lo, float hi) -> Tensor", &melscale_fwd);
  m.def("unet_fwd(int model, Tensor x, Tensor t, Tensor? class_labels, Tensor(a!) ws) -> Tensor", &unet_fwd);
  m.def("unet_fwd_ctx(int model, Tensor x, Tensor t, Tensor? class_labels, Tensor c0, Tensor c1, Tensor bias1, Tensor(a!) ws) -> Tensor", &unet_fwd_ctx);
  m.def("vae_dec_fwd(int model, Tensor z, float z_scale, bool keep_state, bool want_f32, int scale_factor, Tensor(a!) ws) -> (Tensor, Tensor?)", &vae_dec_fwd);
  m.def("vae_dec_bwd(int model, Tensor dmel, float z_scale, int latent_channels, int scale_factor) -> Tensor", &vae_dec_bwd);
  m.def("grad_normalize_(Tensor(a!) dwav, float target) -> Tensor", &grad_normalize_);
  m.def("hifigan_fwd(int model, Tensor mel, Tensor(a!) ws) -> Tensor", &hifigan_fwd);
  m.def("hifigan_bwd(int model, Tensor dwav, int frames, int model_in_dim) -> Tensor", &hifigan_bwd);
  m.def("htsat_fwd(int model, Tensor mel, bool keep_state, Tensor(a!) ws) -> Tensor", &htsat_fwd);
  m.def("htsat_bwd(int model, Tensor dfeat, Tensor? scale, int frames, int bins) -> Tensor", &htsat_bwd);
  m.def("gram_fwd(Tensor feat) -> Tensor", &gram_fwd);
  m.def("gram_bwd(Tensor feat, Tensor dgram) -> Tensor", &gram_bwd);
}
