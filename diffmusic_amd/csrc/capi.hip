// extern "C" surface of libdiffmusic_hip.so (declared in include/diffmusic_hip.h).
#include "conv_pair.h"
#include "models.h"

#define M_IMPL(m) ((m) ? (m)->impl : nullptr)
#define ST(s) ((hipStream_t)(s))

extern "C" {

int dmx_abi_version(void) { return DMX_ABI_VERSION; }
int dmx_act_dtype(void) { return DMX_ACT_DTYPE; }

dmx_model* dmx_hifigan_create(const dmx_hifigan_config* cfg) {
  if (!cfg || cfg->num_upsamples > DMX_MAX_STAGES || cfg->num_kernels > DMX_MAX_STAGES || cfg->num_dilations > DMX_MAX_STAGES) {
    dmx_set_error("bad hifigan config");
    return nullptr;
  }
  return new dmx_model{dmx_make_hifigan(cfg)};
}
dmx_model* dmx_vae_decoder_create(const dmx_vae_config* cfg) {
  if (!cfg || cfg->num_blocks > DMX_MAX_STAGES) { dmx_set_error("bad vae config"); return nullptr; }
  Model* m = dmx_make_vae(cfg);
  return m ? new dmx_model{m} : nullptr;
}
dmx_model* dmx_unet_create(const dmx_unet_config* cfg) {
  if (!cfg || cfg->num_blocks > DMX_MAX_STAGES) { dmx_set_error("bad unet config"); return nullptr; }
  Model* m = dmx_make_unet(cfg);
  return m ? new dmx_model{m} : nullptr;
}
void dmx_model_destroy(dmx_model* m) {
  if (!m) return;
  delete m->impl;
  delete m;
}
int dmx_model_num_params(const dmx_model* m) { return (int)m->impl->ps.params.size(); }
const char* dmx_model_param_name(const dmx_model* m, int i) { return m->impl->ps.params[i].name.c_str(); }
size_t dmx_model_param_numel(const dmx_model* m, int i) { return m->impl->ps.params[i].numel; }
int dmx_model_param_ndim(const dmx_model* m, int i) { return (int)m->impl->ps.params[i].shape.size(); }
int dmx_model_param_dim(const dmx_model* m, int i, int d) { return m->impl->ps.params[i].shape[d]; }
int dmx_model_load_param(dmx_model* m, const char* name, const float* data_host, size_t numel) {
  if (m->impl->finalized) { dmx_set_error("model already finalized"); return DMX_ERR_STATE; }
  return m->impl->ps.load(name, data_host, numel);
}
int dmx_model_finalize(dmx_model* m, void* stream) {
  std::string missing;
  if (!m->impl->ps.all_loaded(&missing)) { dmx_set_error("parameter '%s' was never loaded", missing.c_str()); return DMX_ERR_PARAM; }
  const int rc = m->impl->finalize(ST(stream));
  if (rc == DMX_OK) m->impl->finalized = true;
  return rc;
}

static int check(dmx_model* m, int kind) {
  if (!m || !m->impl || m->impl->kind != kind) { dmx_set_error("wrong model handle"); return DMX_ERR_STATE; }
  if (!m->impl->finalized) { dmx_set_error("model not finalized"); return DMX_ERR_STATE; }
  return DMX_OK;
}

int dmx_hifigan_out_len(const dmx_model* m, int frames) { return dmx_hifigan_out_len_impl(m->impl, frames); }
size_t dmx_hifigan_workspace_bytes(dmx_model* m, int batch, int frames) { return dmx_hifigan_ws_impl(m->impl, batch, frames); }
int dmx_hifigan_fwd(dmx_model* m, const uint16_t* mel, float* wav, int batch, int frames, void* ws, size_t ws_bytes, void* stream) {
  int rc = check(m, DMX_MODEL_HIFIGAN);
  if (rc) return rc;
  if (!ws) { dmx_set_error("null workspace"); return DMX_ERR_WORKSPACE; }
  return dmx_hifigan_fwd_impl(m->impl, mel, wav, batch, frames, ws, ws_bytes, ST(stream));
}
int dmx_hifigan_bwd(dmx_model* m, const float* dwav, uint16_t* dmel, void* stream) {
  int rc = check(m, DMX_MODEL_HIFIGAN);
  if (rc) return rc;
  return dmx_hifigan_bwd_impl(m->impl, dwav, dmel, ST(stream));
}

size_t dmx_vae_workspace_bytes(dmx_model* m, int batch, int h, int w) { return dmx_vae_ws_impl(m->impl, batch, h, w); }
int dmx_vae_decode_fwd(dmx_model* m, const float* z, float z_scale, uint16_t* mel, float* mel_f32, int batch, int h, int w,
                       int keep_state, void* ws, size_t ws_bytes, void* stream) {
  int rc = check(m, DMX_MODEL_VAE);
  if (rc) return rc;
  if (!ws) { dmx_set_error("null workspace"); return DMX_ERR_WORKSPACE; }
  return dmx_vae_fwd_impl(m->impl, z, z_scale, mel, mel_f32, batch, h, w, keep_state, ws, ws_bytes, ST(stream));
}
int dmx_vae_decode_bwd(dmx_model* m, const uint16_t* dmel, float z_scale, float* dz, void* stream) {
  int rc = check(m, DMX_MODEL_VAE);
  if (rc) return rc;
  return dmx_vae_bwd_impl(m->impl, dmel, z_scale, dz, ST(stream));
}

dmx_model* dmx_htsat_create(const dmx_htsat_config* cfg) {
  Model* impl = dmx_make_htsat(cfg);
  if (!impl) return nullptr;
  return new dmx_model{impl};
}
int dmx_htsat_feature_dims(dmx_model* m, int* tokens, int* channels) {
  if (!m || !m->impl || m->impl->kind != DMX_MODEL_HTSAT) { dmx_set_error("wrong model handle"); return DMX_ERR_STATE; }
  dmx_htsat_dims_impl(m->impl, tokens, channels);
  return DMX_OK;
}
size_t dmx_htsat_workspace_bytes(dmx_model* m, int batch, int frames) {
  if (!m || !m->impl || m->impl->kind != DMX_MODEL_HTSAT) return 0;
  return dmx_htsat_ws_impl(m->impl, batch, frames);
}
int dmx_htsat_fwd(dmx_model* m, const float* mel, int batch, int frames, float* feat, int keep_state, void* ws, size_t ws_bytes, void* stream) {
  int rc = check(m, DMX_MODEL_HTSAT);
  if (rc) return rc;
  if (!ws) { dmx_set_error("null workspace"); return DMX_ERR_WORKSPACE; }
  return dmx_htsat_fwd_impl(m->impl, mel, batch, frames, feat, keep_state, ws, ws_bytes, ST(stream));
}
int dmx_htsat_bwd(dmx_model* m, const float* dfeat, const float* scale, float* dmel, void* stream) {
  int rc = check(m, DMX_MODEL_HTSAT);
  if (rc) return rc;
  return dmx_htsat_bwd_impl(m->impl, dfeat, scale, dmel, ST(stream));
}
size_t dmx_htsat_tape_raw(dmx_model* m, int stage, int block, int which, void* dst, size_t dst_elems, void* stream) {
  if (!m || !m->impl || m->impl->kind != DMX_MODEL_HTSAT) return 0;
  const act_t* p = nullptr;
  const size_t n = dmx_htsat_tape_impl(m->impl, stage, block, which, &p);
  if (n && dst && dst_elems >= n) (void)hipMemcpyAsync(dst, p, n * sizeof(act_t), hipMemcpyDeviceToDevice, ST(stream));
  return n;
}
int dmx_gram_fwd(const float* F, float* G, int batch, int tokens, int channels, void* stream) {
  const int rc = dmx_gram_fwd_impl(F, G, batch, tokens, channels, ST(stream));
  if (rc == DMX_ERR_SHAPE) dmx_set_error("gram: 1 <= tokens <= 96");
  return rc;
}
int dmx_gram_bwd(const float* F, const float* dG, float* dF, int batch, int tokens, int channels, void* stream) {
  return dmx_gram_bwd_impl(F, dG, dF, batch, tokens, channels, ST(stream));
}

size_t dmx_unet_workspace_bytes(dmx_model* m, int batch, int h, int w) { return dmx_unet_ws_impl(m->impl, batch, h, w, 0, 0); }
size_t dmx_unet_workspace_bytes_ctx(dmx_model* m, int batch, int h, int w, int n0, int n1) { return dmx_unet_ws_impl(m->impl, batch, h, w, n0, n1); }
int dmx_unet_fwd(dmx_model* m, const float* x, const float* t, const float* class_labels, float* eps, int batch, int h, int w,
                 void* ws, size_t ws_bytes, void* stream) {
  int rc = check(m, DMX_MODEL_UNET);
  if (rc) return rc;
  if (!ws) { dmx_set_error("null workspace"); return DMX_ERR_WORKSPACE; }
  return dmx_unet_fwd_impl(m->impl, x, t, class_labels, eps, batch, h, w, ws, ws_bytes, ST(stream), nullptr, 0, nullptr, 0, nullptr);
}
int dmx_unet_fwd_ctx(dmx_model* m, const float* x, const float* t, const float* class_labels, const float* ctx0, int n0,
                     const float* ctx1, int n1, const float* bias1, float* eps, int batch, int h, int w, void* ws, size_t ws_bytes,
                     void* stream) {
  int rc = check(m, DMX_MODEL_UNET);
  if (rc) return rc;
  if (!ws) { dmx_set_error("null workspace"); return DMX_ERR_WORKSPACE; }
  return dmx_unet_fwd_impl(m->impl, x, t, class_labels, eps, batch, h, w, ws, ws_bytes, ST(stream), ctx0, n0, ctx1, n1, bias1);
}

int dmx_gemm_raw(const void* desc, size_t desc_bytes, void* stream) {
  if (desc_bytes != sizeof(GemmDesc)) { dmx_set_error("GemmDesc size mismatch: %zu vs %zu", desc_bytes, sizeof(GemmDesc)); return DMX_ERR_SHAPE; }
  return dmx_gemm_launch(*reinterpret_cast<const GemmDesc*>(desc), ST(stream));
}

int dmx_conv_pair_raw(const void* desc_a, const void* desc_b, size_t desc_bytes, void* stream) {
  if (desc_bytes != sizeof(GemmDesc)) { dmx_set_error("GemmDesc size mismatch: %zu vs %zu", desc_bytes, sizeof(GemmDesc)); return DMX_ERR_SHAPE; }
  const GemmDesc* a = reinterpret_cast<const GemmDesc*>(desc_a);
  const GemmDesc& b = *reinterpret_cast<const GemmDesc*>(desc_b);
  if (!dmx_conv_pair_eligible(a, b)) { dmx_set_error("shape not handled by the fused convolution-pair kernel"); return DMX_ERR_SHAPE; }
  return dmx_conv_pair_launch(a, b, ST(stream));
}

int dmx_conv_pair_group_raw(int n, const void* descs_a, const void* descs_b, size_t desc_bytes, void* stream) {
  if (desc_bytes != sizeof(GemmDesc)) { dmx_set_error("GemmDesc size mismatch: %zu vs %zu", desc_bytes, sizeof(GemmDesc)); return DMX_ERR_SHAPE; }
  if (n < 1 || n > 3) { dmx_set_error("conv pair group: 1..3 problems"); return DMX_ERR_SHAPE; }
  const GemmDesc* a = reinterpret_cast<const GemmDesc*>(descs_a);
  const GemmDesc* b = reinterpret_cast<const GemmDesc*>(descs_b);
  const GemmDesc* pa[3]; const GemmDesc* pb[3];
  for (int j = 0; j < n; ++j) { pa[j] = a + j; pb[j] = b + j; }
  const int rc = dmx_conv_pair_group_launch(n, pa, pb, ST(stream));
  if (rc == DMX_ERR_SHAPE) dmx_set_error("shape not handled by the fused convolution-pair kernel");
  return rc;
}

int dmx_flash_attn_raw(const void* q, const void* k, const void* v, void* o, const float* colbias, int B, int Nq, int Nk, int ldv, int C,
                       int heads, float scale, void* stream) {
  const int rc = dmx_flash_attn_fwd((const act_t*)q, (const act_t*)k, (const act_t*)v, (act_t*)o, colbias, B, Nq, Nk, C, heads, scale,
                                    ST(stream), 0, 0, ldv);
  if (rc == DMX_ERR_SHAPE) dmx_set_error("flash attention: unsupported head_dim / strides");
  return rc;
}

int dmx_gemm_splitk_workspace(void* ws, size_t bytes) {
  dmx_gemm_set_splitk_workspace(reinterpret_cast<float*>(ws), ws ? bytes : 0);
  return DMX_OK;
}

size_t dmx_groupnorm_scratch_floats(int B, int C, int G) { return dmx_gn_scratch_floats(B, C, G); }
int dmx_groupnorm_raw(const void* x, void* y, const float* gamma, const float* beta, float* stats, float* scale, float* shift,
                      float* partial, int B, int P, int C, int G, float eps, int silu, void* stream) {
  const int rc = dmx_groupnorm_fwd((const act_t*)x, (act_t*)y, gamma, beta, stats, scale, shift, partial, B, P, C, G, eps, silu, ST(stream));
  if (rc == DMX_ERR_SHAPE) dmx_set_error("groupnorm: unsupported channel / group counts");
  return rc;
}

// GroupNorm from producer-written partial sums (EPI_GNSTATS regions, see kernels.h GnParts / gn_parts_kernel): test hook.
// regions: nreg x 6 ints {tm, P, nq, qoff, cq, 0} and nreg buffer pointers.
size_t dmx_groupnorm_part_floats(int B, int P, int N) { return dmx_gn_part_floats(B, P, N); }
int dmx_groupnorm_parts_raw(const void* x, void* y, const float* gamma, const float* beta, float* stats, float* scale, float* shift,
                            int B, int P, int C, int G, float eps, int silu, int nreg, float* const* part, const int* geom, void* stream) {
  if (nreg < 1 || nreg > 8) { dmx_set_error("groupnorm_parts: 1..8 regions"); return DMX_ERR_SHAPE; }
  // this entry point has no scratch for the classic statistics pass: everything that would make dmx_groupnorm_fwd fall back to it is refused
  if (G <= 0 || C % G || ((C / G) & 3)) { dmx_set_error("groupnorm_parts: channels per group must be a multiple of 4 (C %d, G %d)", C, G); return DMX_ERR_SHAPE; }
  GnParts gp;
  for (int i = 0; i < nreg; ++i) {
    GnRegion r;
    r.part = part[i]; r.tm = geom[i * 6]; r.P = geom[i * 6 + 1]; r.nq = geom[i * 6 + 2]; r.qoff = geom[i * 6 + 3]; r.cq = geom[i * 6 + 4];
    if (!r.part || r.tm < 32 || r.P < r.tm || r.nq <= 0 || r.cq <= 0 || r.cq > r.nq || r.qoff < 0 || r.qoff + r.cq > C / 4) {
      dmx_set_error("groupnorm_parts: region %d: tm %d, P %d, nq %d, qoff %d, cq %d do not describe a producer launch of this tensor", i, r.tm, r.P, r.nq, r.qoff, r.cq);
      return DMX_ERR_SHAPE;
    }
    gp.r[gp.n++] = r;
  }
  const int rc = dmx_groupnorm_fwd((const act_t*)x, (act_t*)y, gamma, beta, stats, scale, shift, nullptr, B, P, C, G, eps, silu, ST(stream), &gp);
  if (rc == DMX_ERR_SHAPE) dmx_set_error("groupnorm: unsupported channel / group counts");
  return rc;
}
int dmx_gemm_last_tile_rows_raw(void) { return dmx_gemm_last_tile_rows(); }
// GroupNorm(+SiLU) backward: dx = d/dx of <dy, act(GN(x))> (+ add), with the two per-group sums taken from partial sums the dgrad launch
// that produced dy wrote (EPI_GNBWD; regions as in dmx_groupnorm_parts_raw) or, with nreg == 0, by the classic pass over x and dy.
// stats / scale / shift: the forward's tape; k0 / k1: (B, C) fp32 scratch; partial: scratch of dmx_groupnorm_scratch_floats (nreg == 0).
int dmx_groupnorm_bwd_raw(const void* x, const void* dy, const void* add, void* dx, const float* stats, const float* scale, const float* shift,
                          float* k0, float* k1, float* partial, int B, int P, int C, int G, int silu, int nreg, float* const* part,
                          const int* geom, void* stream) {
  if (nreg < 0 || nreg > 8) { dmx_set_error("groupnorm_bwd: 0..8 regions"); return DMX_ERR_SHAPE; }
  if (nreg > 0 && !partial && (G <= 0 || C % G || ((C / G) & 3))) { dmx_set_error("groupnorm_bwd: regions need channels per group % 4 == 0, or scratch for the classic pass"); return DMX_ERR_SHAPE; }
  GnParts gp;
  for (int i = 0; i < nreg; ++i) {
    GnRegion r;
    r.part = part[i]; r.tm = geom[i * 6]; r.P = geom[i * 6 + 1]; r.nq = geom[i * 6 + 2]; r.qoff = geom[i * 6 + 3]; r.cq = geom[i * 6 + 4];
    if (!r.part || r.tm < 32 || r.P < r.tm || r.nq <= 0 || r.cq <= 0 || r.cq > r.nq || r.qoff < 0 || r.qoff + r.cq > C / 4) {
      dmx_set_error("groupnorm_bwd: region %d does not describe a producer launch of this tensor", i);
      return DMX_ERR_SHAPE;
    }
    gp.r[gp.n++] = r;
  }
  return dmx_groupnorm_bwd((const act_t*)x, (const act_t*)dy, (const act_t*)add, (act_t*)dx, stats, scale, shift, k0, k1, partial, B, P, C, G,
                           silu, ST(stream), nreg ? &gp : nullptr);
}

}  // extern "C"
