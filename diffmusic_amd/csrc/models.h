// Network executors (HiFi-GAN, VAE decoder, U-Net): common base.
#pragma once
#include "layers.h"
#include "../../include/diffmusic_hip.h"

enum { DMX_MODEL_HIFIGAN = 1, DMX_MODEL_VAE = 2, DMX_MODEL_UNET = 3, DMX_MODEL_HTSAT = 4 };

struct Model {
  int kind = 0;
  ParamStore ps;
  Arena arena;
  bool finalized = false;
  bool dry = false;  // dry run: only workspace accounting, no launches
  virtual ~Model() { ps.free_all(); }
  virtual int finalize(hipStream_t st) = 0;
};
struct dmx_model { Model* impl; };

// launch unless dry-running
#define RUN(expr) do { if (!dry) { int rc_ = (expr); if (rc_ != DMX_OK) return rc_; } } while (0)
#define CHECK_WS(name) do { if (arena.overflow) { dmx_set_error(name " workspace too small: need %zu bytes", arena.peak); return DMX_ERR_WORKSPACE; } } while (0)

Model* dmx_make_hifigan(const dmx_hifigan_config* c);
int dmx_hifigan_out_len_impl(Model* m, int T);
int dmx_hifigan_fwd_impl(Model* m, const act_t* mel, float* wav, int B, int T, void* ws, size_t wsb, hipStream_t st);
int dmx_hifigan_bwd_impl(Model* m, const float* dwav, act_t* dmel, hipStream_t st);
size_t dmx_hifigan_ws_impl(Model* m, int B, int T);

Model* dmx_make_vae(const dmx_vae_config* c);
size_t dmx_vae_ws_impl(Model* m, int B, int h, int w);
int dmx_vae_fwd_impl(Model* m, const float* z, float zs, act_t* mel, float* mel32, int B, int h, int w, int keep, void* ws, size_t wsb,
                     hipStream_t st);
int dmx_vae_bwd_impl(Model* m, const act_t* dmel, float zs, float* dz, hipStream_t st);
Model* dmx_make_unet(const dmx_unet_config* c);
size_t dmx_unet_ws_impl(Model* m, int B, int h, int w, int n0, int n1);
int dmx_unet_fwd_impl(Model* m, const float* x, const float* t, const float* cls, float* eps, int B, int h, int w, void* ws, size_t wsb,
                      hipStream_t st, const float* c0, int n0, const float* c1, int n1, const float* bias1);

// CLAP HTS-AT audio tower (style-guidance operator): forward with tape + input-gradient backward, and the Gram matrix of its features
Model* dmx_make_htsat(const dmx_htsat_config* c);
size_t dmx_htsat_ws_impl(Model* m, int B, int frames);
int dmx_htsat_fwd_impl(Model* m, const float* mel, int B, int frames, float* feat, int keep, void* ws, size_t wsb, hipStream_t st);
int dmx_htsat_bwd_impl(Model* m, const float* dfeat, const float* scale, float* dmel, hipStream_t st);
void dmx_htsat_dims_impl(Model* m, int* tokens, int* channels);
size_t dmx_htsat_tape_impl(Model* m, int stage, int block, int which, const act_t** p);
int dmx_gram_fwd_impl(const float* F, float* G, int B, int T, int C, hipStream_t st);
int dmx_gram_bwd_impl(const float* F, const float* dG, float* dF, int B, int T, int C, hipStream_t st);
