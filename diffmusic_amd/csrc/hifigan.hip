// HiFi-GAN vocoder forward + input-gradient backward on the implicit-GEMM kernel.
// Semantics follow transformers SpeechT5HifiGan.forward (modeling_speecht5.py:3010-3069), which the
// reference calls inside every guided step through BaseOperator.inverse_transform
// (diffmusic/inverse_problem/operator.py:126-130; gradient path scheduling_dps.py:195-212).
//
// No autograd tape: the only state kept for the backward pass is the leaky-relu'd activation that each
// convolution consumed (its sign is the leaky-relu' mask; sign(lrelu(x)) == sign(x)) and the tanh output;
// weights never receive gradients, so dgrad needs no saved conv inputs.  In the narrow stages (C <= 128), whose
// resblock steps run in the fused pair kernel and are HBM-bound, the mask travels as SIGN BITS instead (1 byte
// per 8 channels, EPI_BITS2 -> EPI_MASKBITS): the intermediate lrelu(convs1 out) never leaves LDS as a tensor
// and the dgrad pair reads 2/16 instead of 2 mask tensors.  The wide stages (C = 256 / 512, generic tiles whose
// epilogues are VALU-bound) keep 16-bit masks: extracting the bits there cost as much as reading them saved.
#include "models.h"
#include "conv_pair.h"
#include <cstdlib>

// installs a model-owned split-K scratch for the launches of one executor call and removes it on every return path
struct SplitKScope {
  explicit SplitKScope(float* ws, size_t bytes, bool on) : on_(on && ws) { if (on_) dmx_gemm_set_splitk_workspace(ws, bytes); }
  ~SplitKScope() { if (on_) dmx_gemm_set_splitk_workspace(nullptr, 0); }
  bool on_;
};

struct HifiGan : Model {
  dmx_hifigan_config cfg;
  // fp32 partial tiles for the few small-M / deep-K launches of the vocoder (conv_pre and its dgrad: M = B * frames, K = 7 * C0)
  static constexpr size_t kSplitKBytes = 32u << 20;
  float* splitk_ws = nullptr;
  ConvLayer conv_pre, conv_post;
  std::vector<ConvLayer> ups;
  std::vector<ConvLayer> c1, c2;  // [stage][kernel][dil] flattened
  int nk = 0, nd = 0, ns = 0;
  // tape
  int B = 0, T = 0;
  std::vector<int> Ts;                 // length after each stage
  act_t* act_pre = nullptr;           // lrelu(conv_pre(mel))
  std::vector<act_t*> xs_a;           // [stage] lrelu(upsampler out)
  std::vector<act_t*> ha, xa;         // [stage][kernel][dil] lrelu(convs1 out) (unfused steps only) / input of convs1[d]
  std::vector<act_t*> act_out;        // [stage] lrelu(stage output) (slope of the consumer)
  // fused steps: sign-bit tapes (1 byte per 8 channels) of xa and of the intermediate lrelu(convs1 out)
  std::vector<unsigned char*> hb, xb;
  std::vector<char> fused;            // [stage][kernel][dil] this resblock step runs in the fused pair kernel
  float* wav8 = nullptr;               // (B, Tout, 8) fp32 tanh output, channel 0 real
  bool have_tape = false;
  // The nk resblock branches of a stage are independent until their outputs are averaged: optionally (DMX_MULTI_STREAM=1) they
  // run on their own HIP streams so that the tail of one branch's launches is filled by the others.  Off by default: with the
  // round-filling tile heights and the fused pair kernel the tails are short and one stream measures faster.
  hipStream_t bstream[DMX_MAX_STAGES] = {};
  std::vector<hipEvent_t> events;
  size_t ev_next = 0;
  bool want_multi = true;
  hipEvent_t next_event() { hipEvent_t e = events[ev_next]; ev_next = (ev_next + 1) % events.size(); return e; }
  bool multi() const { return want_multi && !dry && !dmx_prof_is_active() && nk > 1; }
  bool multi_wanted() const { return want_multi && nk > 1; }      // (the one shared split-K scratch is for single-stream schedules only)

  int idx(int s, int k, int d) const { return (s * nk + k) * nd + d; }

  explicit HifiGan(const dmx_hifigan_config& c) : cfg(c) {
    kind = DMX_MODEL_HIFIGAN;
    ns = c.num_upsamples; nk = c.num_kernels; nd = c.num_dilations;
    const int C0 = c.upsample_initial_channel;
    conv_pre = make_conv1d(ps, "conv_pre", c.model_in_dim, C0, 7, 1, 3, true);
    for (int i = 0; i < ns; ++i) {
      const int s = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
      ups.push_back(make_convT1d(ps, "upsampler." + std::to_string(i), C0 >> i, C0 >> (i + 1), k, s, (k - s) / 2, true));
    }
    int ch = C0;
    for (int i = 0; i < ns; ++i) {
      ch = C0 >> (i + 1);
      for (int j = 0; j < nk; ++j) {
        const int k = c.resblock_kernel_sizes[j];
        const std::string pre = "resblocks." + std::to_string(i * nk + j);
        for (int d = 0; d < nd; ++d) {
          const int dil = c.resblock_dilation_sizes[j * nd + d];
          c1.push_back(make_conv1d(ps, pre + ".convs1." + std::to_string(d), ch, ch, k, dil, (k * dil - dil) / 2, true));
          c2.push_back(make_conv1d(ps, pre + ".convs2." + std::to_string(d), ch, ch, k, 1, (k - 1) / 2, true));
        }
      }
    }
    conv_post = make_conv1d(ps, "conv_post", ch, 1, 7, 1, 3, true);
    // measured after the fused resblock-pair kernel: one stream is ~0.9 ms/step faster than three (the tails the extra streams
    // used to fill are gone); DMX_MULTI_STREAM=1 restores the three-stream schedule
    want_multi = getenv("DMX_MULTI_STREAM") != nullptr && getenv("DMX_SINGLE_STREAM") == nullptr;
    for (int k = 1; k < nk; ++k) (void)hipStreamCreateWithFlags(&bstream[k], hipStreamNonBlocking);
    events.resize(256);
    for (auto& e : events) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    splitk_ws = (float*)ps.dalloc(kSplitKBytes);
  }
  ~HifiGan() override {
    dmx_gemm_release_splitk_workspace(splitk_ws);
    for (int k = 1; k < nk; ++k) if (bstream[k]) (void)hipStreamDestroy(bstream[k]);
    for (auto& e : events) (void)hipEventDestroy(e);
  }

  int finalize(hipStream_t st) override {
    int rc = pack_layer(ps, conv_pre, st);
    for (auto& l : ups) if (rc == DMX_OK) rc = pack_layer(ps, l, st);
    for (auto& l : c1) if (rc == DMX_OK) rc = pack_layer(ps, l, st);
    for (auto& l : c2) if (rc == DMX_OK) rc = pack_layer(ps, l, st);
    if (rc == DMX_OK) rc = pack_layer(ps, conv_post, st);
    return rc;
  }

  // does resblock step `id` (length To per clip) go to the fused pair kernel?  Decided from the shapes alone, before any buffer
  // exists (it selects the tape format: sign bits vs 16-bit tensors), with the flag sets the forward pass will use.
  bool step_is_fused(int id, int To) const {
    GemmDesc da, db;
    act_t* dummy = reinterpret_cast<act_t*>(size_t(256));
    unsigned char* dummyb = reinterpret_cast<unsigned char*>(size_t(256));
    Epi ea; ea.flags = EPI_LRELU2 | EPI_NO_C | EPI_BITS2; ea.act_slope = cfg.leaky_relu_slope; ea.B2 = dummyb;
    if (conv_fwd_1d_desc(c1[id], dummy, nullptr, B, To, ea, da) != DMX_OK) return false;
    Epi eb; eb.flags = EPI_RESID | EPI_RESID_INV | EPI_LRELU2 | EPI_NO_C | EPI_ACCUM; eb.R = dummy; eb.C2 = dummy;
    eb.resid_inv_slope = 1.f / cfg.leaky_relu_slope; eb.act_slope = cfg.leaky_relu_slope;
    if (conv_fwd_1d_desc(c2[id], nullptr, dummy, B, To, eb, db) != DMX_OK) return false;
    return dmx_conv_pair_eligible(&da, db);
  }

  int out_len(int T_) const {
    int t = T_;
    for (int i = 0; i < ns; ++i) t = conv_out_len(ups[i], t);
    return t;
  }

  // mel: (B, T, model_in_dim) bf16 ; wav: (B, Tout) fp32
  int forward(const act_t* mel, float* wav, int B_, int T_, void* ws, size_t ws_bytes, hipStream_t st) {
    if (cfg.model_in_dim & 7) return DMX_ERR_SHAPE;
    dry = (ws == nullptr);
    SplitKScope sk_scope(splitk_ws, kSplitKBytes, !dry && !multi_wanted());
    arena.reset(ws, dry ? (size_t)-1 : ws_bytes);
    B = B_; T = T_;
    Ts.assign(ns, 0);
    xs_a.assign(ns, nullptr); act_out.assign(ns, nullptr);
    ha.assign(ns * nk * nd, nullptr); xa.assign(ns * nk * nd, nullptr);
    hb.assign(ns * nk * nd, nullptr); xb.assign(ns * nk * nd, nullptr);
    fused.assign(ns * nk * nd, 0);
    const float slope = cfg.leaky_relu_slope;
    auto bits = [&](size_t rows, int C) { return (unsigned char*)arena.raw(rows * (size_t)(C >> 3)); };
    // conv_pre -> only the activated tensor is needed downstream
    act_pre = arena.bf((size_t)B * T * conv_pre.Cop);
    {
      Epi e; e.flags = EPI_LRELU2 | EPI_NO_C; e.act_slope = slope; e.C2 = act_pre;
      RUN(conv_fwd_1d(conv_pre, mel, act_pre, B, T, e, st));
    }
    const act_t* cur_act = act_pre;
    int Tin = T;
    for (int s = 0; s < ns; ++s) {
      const ConvLayer& up = ups[s];
      const int To = conv_out_len(up, Tin), C = up.Cop;
      const size_t n = (size_t)B * To * C;
      Ts[s] = To;
      xs_a[s] = arena.bf(n);
      act_out[s] = arena.bf(n);
      unsigned char* xs_b = nullptr;        // sign bits of xs_a, wanted when a first resblock step of the stage is fused
      for (int k = 0; k < nk; ++k)
        for (int d = 0; d < nd; ++d) {
          const int id = idx(s, k, d);
          fused[id] = step_is_fused(id, To) ? 1 : 0;
          xa[id] = d == 0 ? xs_a[s] : arena.bf(n);
          if (fused[id]) {
            hb[id] = bits((size_t)B * To, C);
            if (d == 0) { if (!xs_b) xs_b = bits((size_t)B * To, C); xb[id] = xs_b; }
            else xb[id] = bits((size_t)B * To, C);
          } else {
            ha[id] = arena.bf(n);
          }
        }
      const size_t mk = arena.mark();       // transients below are released per stage
      // the raw residual stream x is never stored: a conv reads the leaky-relu'd tensor it needs anyway, and the residual
      // add reconstructs x = a > 0 ? a : a / slope in the epilogue (EPI_RESID_INV) -- one 16-bit tensor write less per conv2
      act_t* sum = arena.bf(n);
      {
        Epi e; e.flags = EPI_LRELU2 | EPI_NO_C; e.act_slope = slope; e.C2 = xs_a[s];
        if (xs_b) { e.flags |= EPI_BITS2; e.B2 = xs_b; }
        RUN(conv_fwd_1d(up, cur_act, xs_a[s], B, Tin, e, st));
      }
      const float next_slope = (s == ns - 1) ? 0.01f : slope;
      const bool mt = multi();
      bool all_fused = nk <= 3;
      for (int k = 0; k < nk; ++k) for (int d = 0; d < nd; ++d) all_fused = all_fused && fused[idx(s, k, d)] != 0;
      if (all_fused && !mt) {
        // narrow stage, every resblock step in the fused pair kernel: step d of all nk branches is ONE grouped launch (the branches
        // are independent until the averaged sum); the last step accumulates into `sum` in branch order and stays nk launches
        for (int d = 0; d < nd; ++d) {
          GemmDesc da[DMX_MAX_STAGES], db[DMX_MAX_STAGES];
          const GemmDesc* pa[DMX_MAX_STAGES]; const GemmDesc* pb[DMX_MAX_STAGES];
          for (int k = 0; k < nk; ++k) {
            const int id = idx(s, k, d);
            {
              Epi e; e.flags = EPI_LRELU2 | EPI_NO_C | EPI_BITS2; e.act_slope = slope; e.B2 = hb[id];
              RUN(conv_fwd_1d_desc(c1[id], xa[id], ha[id], B, To, e, da[k]));
            }
            if (d < nd - 1) {
              const int idn = idx(s, k, d + 1);
              act_t* xn = xa[idn];
              Epi e; e.flags = EPI_RESID | EPI_RESID_INV | EPI_LRELU2 | EPI_NO_C | EPI_BITS2; e.R = xa[id]; e.resid_inv_slope = 1.f / slope;
              e.act_slope = slope; e.C2 = xn; e.B2 = xb[idn];
              RUN(conv_fwd_1d_desc(c2[id], ha[id], xn, B, To, e, db[k]));
            } else {
              Epi e; e.flags = EPI_RESID | EPI_RESID_INV; e.R = xa[id]; e.resid_inv_slope = 1.f / slope; e.alpha = 1.f / nk;
              if (k > 0) e.flags |= EPI_ACCUM;
              if (k == nk - 1) { e.flags |= EPI_LRELU2 | EPI_NO_C; e.act_slope = next_slope; e.C2 = act_out[s]; }
              RUN(conv_fwd_1d_desc(c2[id], ha[id], sum, B, To, e, db[k]));
            }
            pa[k] = &da[k]; pb[k] = &db[k];
          }
          if (dry) continue;
          // the tape of these steps holds sign bits only (no ha tensor exists): the unfused two-launch fallback of conv_pair_run
          // would run stage a with C2 == nullptr -- refuse instead of falling back silently
          for (int k = 0; k < nk; ++k)
            if (!dmx_conv_pair_eligible(&da[k], db[k])) {
              dmx_set_error("hifigan: resblock step %d was planned as fused but the pair kernel refuses it", idx(s, k, d));
              return DMX_ERR_STATE;
            }
          if (d < nd - 1) RUN(dmx_conv_pair_group_launch(nk, pa, pb, st));
          else for (int k = 0; k < nk; ++k) RUN(conv_pair_run(da[k], db[k], st));
        }
        arena.release(mk);
        cur_act = act_out[s];
        Tin = To;
        continue;
      }
      hipEvent_t fin_prev = nullptr;
      if (mt) {
        hipEvent_t e0 = next_event();
        (void)hipEventRecord(e0, st);
        for (int k = 1; k < nk; ++k) (void)hipStreamWaitEvent(bstream[k], e0, 0);
      }
      for (int k = 0; k < nk; ++k) {
        hipStream_t sk = (mt && k > 0) ? bstream[k] : st;
        for (int d = 0; d < nd; ++d) {
          const int id = idx(s, k, d);
          const bool fz = fused[id] != 0;
          GemmDesc da, db;
          {
            // fused: the activated intermediate stays in LDS, only its sign bits are written; unfused: it is a tape tensor
            Epi e; e.flags = EPI_LRELU2 | EPI_NO_C; e.act_slope = slope;
            if (fz) { e.flags |= EPI_BITS2; e.B2 = hb[id]; } else e.C2 = ha[id];
            RUN(conv_fwd_1d_desc(c1[id], xa[id], ha[id], B, To, e, da));
          }
          if (d < nd - 1) {
            const int idn = idx(s, k, d + 1);
            act_t* xn = xa[idn];
            Epi e; e.flags = EPI_RESID | EPI_RESID_INV | EPI_LRELU2 | EPI_NO_C; e.R = xa[id]; e.resid_inv_slope = 1.f / slope;
            e.act_slope = slope; e.C2 = xn;
            if (fused[idn]) { e.flags |= EPI_BITS2; e.B2 = xb[idn]; }
            RUN(conv_fwd_1d_desc(c2[id], ha[id], xn, B, To, e, db));
          } else {
            Epi e; e.flags = EPI_RESID | EPI_RESID_INV; e.R = xa[id]; e.resid_inv_slope = 1.f / slope; e.alpha = 1.f / nk;
            if (k > 0) e.flags |= EPI_ACCUM;
            if (k == nk - 1) { e.flags |= EPI_LRELU2 | EPI_NO_C; e.act_slope = next_slope; e.C2 = act_out[s]; }
            RUN(conv_fwd_1d_desc(c2[id], ha[id], sum, B, To, e, db));
          }
          if (!dry && fz && !dmx_conv_pair_eligible(&da, db)) {       // the tape format was chosen for the fused kernel: never fall back silently
            dmx_set_error("hifigan: resblock step %d was planned as fused but the pair kernel refuses it", id);
            return DMX_ERR_STATE;
          }
          if (d == nd - 1 && mt && fin_prev) (void)hipStreamWaitEvent(sk, fin_prev, 0);      // the averaged sum is accumulated in branch order
          RUN(conv_pair_run(da, db, sk));
          if (d == nd - 1 && mt) { fin_prev = next_event(); (void)hipEventRecord(fin_prev, sk); }
        }
      }
      if (mt && fin_prev) (void)hipStreamWaitEvent(st, fin_prev, 0);               // join
      arena.release(mk);
      cur_act = act_out[s];
      Tin = To;
    }
    const int Tout = Tin;
    wav8 = arena.f32((size_t)B * Tout * 8);
    CHECK_WS("hifigan");
    {
      Epi e; e.flags = EPI_F32OUT | EPI_TANH;
      RUN(conv_fwd_1d(conv_post, cur_act, wav8, B, Tout, e, st));
    }
    RUN(dmx_gather_col_f32(wav8, wav, (long long)B * Tout, 8, 0, st));
    have_tape = true;
    return DMX_OK;
  }

  // dwav: (B, Tout) fp32 -> dmel: (B, T, model_in_dim) bf16.  Uses the tape of the last forward.
  int backward(const float* dwav, act_t* dmel, hipStream_t st) {
    if (!have_tape && !dry) { dmx_set_error("hifigan backward without forward"); return DMX_ERR_STATE; }
    SplitKScope sk_scope(splitk_ws, kSplitKBytes, !dry && !multi_wanted());
    const float slope = cfg.leaky_relu_slope;
    const int Tout = Ts[ns - 1];
    const size_t mk0 = arena.mark();
    // d tanh: gz = dwav * (1 - wav^2), padded to 8 channels
    act_t* gz = arena.bf((size_t)B * Tout * 8);
    const int Clast = ups[ns - 1].Cop;
    act_t* g = arena.bf((size_t)B * Tout * Clast);      // grad wrt each resblock output of the last stage
    CHECK_WS("hifigan");
    RUN(dmx_tanh_bwd_pad8(dwav, wav8, gz, (long long)B * Tout, st));
    {
      Epi e; e.flags = EPI_MASK; e.X = act_out[ns - 1]; e.mask_slope = 0.01f; e.alpha = 1.f / nk;
      RUN(conv_bwd_1d(conv_post, gz, g, B, Tout, e, st));
    }
    for (int s = ns - 1; s >= 0; --s) {
      const int To = Ts[s], Tin = s == 0 ? T : Ts[s - 1], C = ups[s].Cop;
      const size_t n = (size_t)B * To * C;
      act_t* gxs = arena.bf(n);
      act_t* ghk[DMX_MAX_STAGES];
      act_t* gAB[DMX_MAX_STAGES][2];
      for (int k = 0; k < nk; ++k) { ghk[k] = arena.bf(n); gAB[k][0] = arena.bf(n); gAB[k][1] = arena.bf(n); }
      const int Cin = ups[s].Cip;
      act_t* gprev = arena.bf((size_t)B * Tin * Cin);
      CHECK_WS("hifigan");
      const bool mt = multi();
      hipEvent_t fin_prev = nullptr;
      bool all_fused = nk <= 3;
      for (int k = 0; k < nk; ++k) for (int d = 0; d < nd; ++d) all_fused = all_fused && fused[idx(s, k, d)] != 0;
      if (all_fused && !mt) {
        // grouped launches as in forward(): step d of all branches at once; the last step (d = 0) accumulates into gxs in branch order
        const act_t* gcs[DMX_MAX_STAGES];
        for (int k = 0; k < nk; ++k) gcs[k] = g;
        for (int d = nd - 1; d >= 0; --d) {
          GemmDesc da[DMX_MAX_STAGES], db[DMX_MAX_STAGES];
          const GemmDesc* pa[DMX_MAX_STAGES]; const GemmDesc* pb[DMX_MAX_STAGES];
          act_t* dsts[DMX_MAX_STAGES];
          for (int k = 0; k < nk; ++k) {
            const int id = idx(s, k, d);
            {
              Epi e; e.mask_slope = slope; e.flags = EPI_MASKBITS; e.XB = hb[id];
              RUN(conv_bwd_1d_desc(c2[id], gcs[k], ghk[k], B, To, e, da[k]));
            }
            Epi e; e.flags = EPI_RESID | EPI_MASKBITS; e.mask_slope = slope; e.R = gcs[k]; e.XB = xb[id];
            if (d == 0) { dsts[k] = gxs; if (k > 0) e.flags |= EPI_ACCUM; }
            else dsts[k] = (gcs[k] == gAB[k][0]) ? gAB[k][1] : gAB[k][0];
            RUN(conv_bwd_1d_desc(c1[id], ghk[k], dsts[k], B, To, e, db[k]));
            pa[k] = &da[k]; pb[k] = &db[k];
          }
          if (!dry) {
            if (d > 0) RUN(dmx_conv_pair_group_launch(nk, pa, pb, st));
            else for (int k = 0; k < nk; ++k) RUN(conv_pair_run(da[k], db[k], st));
          }
          for (int k = 0; k < nk; ++k) gcs[k] = dsts[k];
        }
      } else {
      if (mt) {
        hipEvent_t e0 = next_event();
        (void)hipEventRecord(e0, st);
        for (int k = 1; k < nk; ++k) (void)hipStreamWaitEvent(bstream[k], e0, 0);
      }
      for (int k = 0; k < nk; ++k) {
        hipStream_t sk = (mt && k > 0) ? bstream[k] : st;
        const act_t* gc = g;
        for (int d = nd - 1; d >= 0; --d) {
          const int id = idx(s, k, d);
          GemmDesc da, db;
          {
            Epi e; e.mask_slope = slope;
            if (fused[id]) { e.flags = EPI_MASKBITS; e.XB = hb[id]; } else { e.flags = EPI_MASK; e.X = ha[id]; }
            RUN(conv_bwd_1d_desc(c2[id], gc, ghk[k], B, To, e, da));
          }
          Epi e; e.flags = EPI_RESID; e.mask_slope = slope; e.R = gc;
          if (fused[id]) { e.flags |= EPI_MASKBITS; e.XB = xb[id]; } else { e.flags |= EPI_MASK; e.X = xa[id]; }
          act_t* dst;
          if (d == 0) {
            dst = gxs;
            if (k > 0) e.flags |= EPI_ACCUM;
            if (mt && fin_prev) (void)hipStreamWaitEvent(sk, fin_prev, 0);
          } else {
            dst = (gc == gAB[k][0]) ? gAB[k][1] : gAB[k][0];
          }
          RUN(conv_bwd_1d_desc(c1[id], ghk[k], dst, B, To, e, db));
          RUN(conv_pair_run(da, db, sk));
          if (d == 0 && mt) { fin_prev = next_event(); (void)hipEventRecord(fin_prev, sk); }
          gc = dst;
        }
      }
      }
      if (mt && fin_prev) (void)hipStreamWaitEvent(st, fin_prev, 0);
      // through the upsampler (strided conv) and the leaky-relu that fed it
      {
        Epi e; e.flags = EPI_MASK; e.mask_slope = slope;
        if (s > 0) { e.X = act_out[s - 1]; e.alpha = 1.f / nk; } else { e.X = act_pre; }
        RUN(conv_bwd_1d(ups[s], gxs, gprev, B, Tin, e, st));
      }
      g = gprev;   // (buffers of this stage stay allocated until the end of backward; sizes shrink geometrically)
    }
    {
      Epi e;
      RUN(conv_bwd_1d(conv_pre, g, dmel, B, T, e, st));
    }
    arena.release(mk0);
    return DMX_OK;
  }
};

Model* dmx_make_hifigan(const dmx_hifigan_config* c) { return new HifiGan(*c); }
int dmx_hifigan_out_len_impl(Model* m, int T) { return static_cast<HifiGan*>(m)->out_len(T); }
int dmx_hifigan_fwd_impl(Model* m, const act_t* mel, float* wav, int B, int T, void* ws, size_t wsb, hipStream_t st) {
  return static_cast<HifiGan*>(m)->forward(mel, wav, B, T, ws, wsb, st);
}
int dmx_hifigan_bwd_impl(Model* m, const float* dwav, act_t* dmel, hipStream_t st) {
  return static_cast<HifiGan*>(m)->backward(dwav, dmel, st);
}
size_t dmx_hifigan_ws_impl(Model* m, int B, int T) {
  HifiGan* h = static_cast<HifiGan*>(m);
  h->arena.peak = 0;
  h->forward(nullptr, nullptr, B, T, nullptr, 0, nullptr);
  h->backward(nullptr, nullptr, nullptr);
  h->have_tape = false;
  h->dry = false;
  return h->arena.peak + 256;
}
