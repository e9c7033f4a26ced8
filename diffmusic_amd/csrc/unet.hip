// UNet2DConditionModel forward (inference only; the guidance gradient never enters the U-Net because
// every in-scope scheduler detaches `sample` and treats eps as a constant, scheduling_dps.py:165).
// diffusers 0.31.0 semantics as configured for MusicLDM:
// simple_projection class embedding concatenated to the time embedding, encoder_hidden_states=None
// so both attention layers of a transformer block are self-attention, conv proj_in/proj_out, GEGLU.
// Replaces `self.unet(latent_model_input, t, encoder_hidden_states=None, class_labels=prompt_embeds)`
// (diffmusic/pipelines/pipeline_musicldm.py:696-703).
#include "blocks.h"

struct LnParams {
  int g_id = -1, b_id = -1;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  void build(ParamStore& ps, const std::string& pre, int C) { g_id = ps.add(pre + ".weight", {C}); b_id = ps.add(pre + ".bias", {C}); }
  void bind(ParamStore& ps) { gamma = ps.dev(g_id); beta = ps.dev(b_id); }
};

struct AttnLayer {
  ConvLayer to_q, to_k, to_v, to_out;
  int heads = 1, C = 0;
  void build(ParamStore& ps, const std::string& pre, int dim, int cross_dim, int heads_) {
    heads = heads_; C = dim;
    to_q = make_linear(ps, pre + ".to_q", dim, dim, false, false);
    to_k = make_linear(ps, pre + ".to_k", cross_dim, dim, false, false);
    to_v = make_linear(ps, pre + ".to_v", cross_dim, dim, false, false);
    to_out = make_linear(ps, pre + ".to_out.0", dim, dim, true, false);
  }
  ConvLayer to_qkv;           // self-attention only: one projection with the three packed weights stacked along N
  bool fused = false;
  int pack(ParamStore& ps, hipStream_t st) {
    CTRY(pack_layer(ps, to_q, st)); CTRY(pack_layer(ps, to_k, st));
    CTRY(pack_layer(ps, to_v, st)); CTRY(pack_layer(ps, to_out, st));
    if (to_k.Ci == C && to_k.Cip == to_q.Cip && to_q.Cop == C) {
      const size_t one = (size_t)C * to_q.Cip * sizeof(act_t);
      act_t* w = (act_t*)ps.dalloc(3 * one);
      if (!w) return DMX_ERR_PARAM;
      (void)hipMemcpyAsync(w, to_q.wf[0], one, hipMemcpyDeviceToDevice, st);
      (void)hipMemcpyAsync((char*)w + one, to_k.wf[0], one, hipMemcpyDeviceToDevice, st);
      (void)hipMemcpyAsync((char*)w + 2 * one, to_v.wf[0], one, hipMemcpyDeviceToDevice, st);
      to_qkv = to_q;
      to_qkv.Co = to_qkv.Cop = 3 * C;
      to_qkv.wf.assign(1, w);
      to_qkv.has_bias = false;
      to_qkv.bias = nullptr;
      to_qkv.colsum = nullptr;
      if (to_q.colsum && to_k.colsum && to_v.colsum) {
        // LayerNorm folded into all three (self-attention): stack the row sums and the folded biases (W beta) like the weights
        float* cs = (float*)ps.dalloc(3 * (size_t)C * sizeof(float));
        float* bs = (float*)ps.dalloc(3 * (size_t)C * sizeof(float));
        if (!cs || !bs) return DMX_ERR_PARAM;
        const ConvLayer* src[3] = {&to_q, &to_k, &to_v};
        for (int i = 0; i < 3; ++i) {
          (void)hipMemcpyAsync(cs + (size_t)i * C, src[i]->colsum, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st);
          (void)hipMemcpyAsync(bs + (size_t)i * C, src[i]->bias, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st);
        }
        to_qkv.colsum = cs; to_qkv.bias = bs; to_qkv.has_bias = true;
      } else if (to_q.colsum || to_k.colsum || to_v.colsum) {
        return DMX_OK;           // (mixed: cannot happen for self-attention; leave the three projections unstacked)
      }
      fused = getenv("DMX_NO_QKV_FUSE") == nullptr;
    }
    return DMX_OK;
  }
  // hres += to_out(attn(l, ctx)) ; l (B,N,C) normalised input, ctx (B,Nc,Cc) or nullptr (self)
  // kv_pre / ldkv: this layer's [k | v] slice (2C columns, row stride ldkv) of a projection of the context computed once per forward
  // for all cross-attention layers (UNet::ctxkv); nullptr: project here
  // rs / ns: row statistics of `l` when the projections carry a folded LayerNorm (l is then the RAW hidden state); the out-projection
  // writes the statistics of the updated hidden state back into the same buffer (for the next folded LayerNorm)
  int fwd(Ctx& cx, const act_t* l, const act_t* ctx, int Nc, act_t* hres, int B, int N, const float* colbias,
          const act_t* kv_pre = nullptr, int ldkv = 0, float* rs = nullptr, int ns = 0) const {
    Arena& A = *cx.arena;
    const size_t mk = A.mark();
    const act_t* kv_in = ctx ? ctx : l;
    const int Nk = ctx ? Nc : N;
    act_t* o = A.bf((size_t)B * N * C);
    Epi e;
    Epi eq; eq.rowstats_in = rs; eq.nslots = ns;          // projections of the (LayerNorm-folded) hidden state
    if (ctx && ldkv > 0) {
      act_t* q = A.bf((size_t)B * N * C);
      CRUN(linear_fwd(to_q, l, to_q.Cip, q, C, (long long)B * N, eq, cx.st));
      CTRY(attention_core(cx, q, kv_pre, kv_pre + C, o, B, N, Nk, C, heads, nullptr, colbias, C, ldkv, ldkv));
    } else if (!ctx && fused) {
      // self-attention: q | k | v come out of one GEMM (N = 3C) and are consumed as strided slices
      act_t* qkv = A.bf((size_t)B * N * 3 * C);
      CRUN(linear_fwd(to_qkv, l, to_q.Cip, qkv, 3 * C, (long long)B * N, eq, cx.st));
      CTRY(attention_core(cx, qkv, qkv + C, qkv + 2 * C, o, B, N, N, C, heads, nullptr, colbias, 3 * C, 3 * C, 3 * C));
    } else {
      act_t* q = A.bf((size_t)B * N * C);
      act_t* k = A.bf((size_t)B * Nk * C);
      act_t* v = A.bf((size_t)B * Nk * C);
      CRUN(linear_fwd(to_q, l, to_q.Cip, q, C, (long long)B * N, eq, cx.st));
      CRUN(linear_fwd(to_k, kv_in, to_k.Cip, k, C, (long long)B * Nk, ctx ? e : eq, cx.st));
      CRUN(linear_fwd(to_v, kv_in, to_v.Cip, v, C, (long long)B * Nk, ctx ? e : eq, cx.st));
      CTRY(attention_core(cx, q, k, v, o, B, N, Nk, C, heads, nullptr, colbias));
    }
    Epi er; er.flags = EPI_RESID; er.R = hres;
    if (rs) { er.flags |= EPI_ROWSTATS; er.rowstats_out = rs; er.nslots = ns; }
    CRUN(linear_fwd(to_out, o, C, hres, C, (long long)B * N, er, cx.st));
    A.release(mk);
    return DMX_OK;
  }
};

struct Transformer2D {
  int C = 0;
  GnLayer norm;
  ConvLayer proj_in, proj_out, ff1, ff2;
  LnParams ln1, ln2, ln3;
  AttnLayer attn1, attn2;
  int cross = 0;   // > 0: attn2 attends an external context of that width
  void build(ParamStore& ps, const std::string& pre, int ch, int heads, int cross_dim, int groups) {
    C = ch; cross = cross_dim > 0 ? cross_dim : 0;
    norm.build(ps, pre + ".norm", ch, groups, 1e-6f);
    proj_in = make_conv2d(ps, pre + ".proj_in", ch, ch, 1, 1, 0, false);
    const std::string tb = pre + ".transformer_blocks.0";
    ln1.build(ps, tb + ".norm1", ch);
    attn1.build(ps, tb + ".attn1", ch, ch, heads);
    ln2.build(ps, tb + ".norm2", ch);
    attn2.build(ps, tb + ".attn2", ch, cross > 0 ? cross : ch, heads);
    ln3.build(ps, tb + ".norm3", ch);
    ff1 = make_linear(ps, tb + ".ff.net.0.proj", ch, ch * 8, true, false);
    ff1.geglu = ((ch * 8) % 32 == 0) && getenv("DMX_NO_GEGLU_FUSE") == nullptr;     // GEGLU in the projection's epilogue (no 8C-wide tensor)
    ff2 = make_linear(ps, tb + ".ff.net.2", ch * 4, ch, true, false);
    proj_out = make_conv2d(ps, pre + ".proj_out", ch, ch, 1, 1, 0, false);
    // LayerNorm folded into the projections that consume it (EPI_LNFOLD): norm1 -> attn1 q/k/v, norm2 -> attn2 q (and k/v when attn2 is
    // self-attention), norm3 -> ff1.  The three layernorm launches and their normalised tensors disappear.
    ln_fold = getenv("DMX_NO_LN_FOLD") == nullptr && (ch % 32) == 0;      // (row statistics travel in slots of 32 channels)
    if (ln_fold) {
      auto fold = [](ConvLayer& L, const LnParams& ln) { L.ln_g_id = ln.g_id; L.ln_b_id = ln.b_id; L.ln_eps = 1e-5f; };
      fold(attn1.to_q, ln1); fold(attn1.to_k, ln1); fold(attn1.to_v, ln1);
      fold(attn2.to_q, ln2);
      if (cross == 0) { fold(attn2.to_k, ln2); fold(attn2.to_v, ln2); }
      fold(ff1, ln3);
    }
  }
  bool ln_fold = false;
  int pack(ParamStore& ps, hipStream_t st) {
    norm.bind(ps); ln1.bind(ps); ln2.bind(ps); ln3.bind(ps);
    CTRY(pack_layer(ps, proj_in, st)); CTRY(pack_layer(ps, proj_out, st));
    CTRY(pack_layer(ps, ff1, st)); CTRY(pack_layer(ps, ff2, st));
    CTRY(attn1.pack(ps, st)); CTRY(attn2.pack(ps, st));
    return DMX_OK;
  }
  // x (B,H,W,C) -> out (B,H,W,C); out may not alias x
  // x_parts / out_parts: GroupNorm partial sums of x (from its producer) and for `out` (Resnet2D::fwd's protocol, blocks.h)
  int fwd(Ctx& cx, const act_t* x, act_t* out, int B, int H, int W, const act_t* ctx = nullptr, int Nc = 0,
          const float* colbias = nullptr, const act_t* kv_pre = nullptr, int ldkv = 0, const GnParts* x_parts = nullptr,
          GnParts* out_parts = nullptr) const {
    Arena& A = *cx.arena;
    const int N = H * W;
    const size_t mk = A.mark();
    GnTape gt = norm.alloc(cx, B);
    act_t* n = A.bf((size_t)B * N * C);
    act_t* hbuf = A.bf((size_t)B * N * C);
    act_t* l = A.bf((size_t)B * N * C);
    CTRY(norm.fwd(cx, x, n, B, N, 0, gt, x_parts));
    Epi e;
    // ln_fold: every GEMM that writes the hidden state also writes its per-row partial sums (EPI_ROWSTATS) -- the statistics of the
    // LayerNorm folded into the projection that reads it next
    const int ns = C / 32;
    float* rs = ln_fold ? A.f32((size_t)B * N * ns * 2) : nullptr;
    {
      Epi ei;
      if (ln_fold) { ei.flags = EPI_ROWSTATS; ei.rowstats_out = rs; ei.nslots = ns; }
      CRUN(conv_fwd_2d(proj_in, n, hbuf, B, H, W, ei, cx.st));
    }
    // (ln_fold: the projections read the raw hidden state `hbuf` and normalise inside; they finish before the out-projection's
    //  residual epilogue rewrites hbuf -- launches of one stream run in order)
    const act_t* lin = ln_fold ? hbuf : l;
    if (!ln_fold) CRUN(dmx_layernorm_fwd(hbuf, l, ln1.gamma, ln1.beta, B * N, C, 1e-5f, cx.st));
    CTRY(attn1.fwd(cx, lin, nullptr, 0, hbuf, B, N, nullptr, nullptr, 0, rs, ns));
    if (!ln_fold) CRUN(dmx_layernorm_fwd(hbuf, l, ln2.gamma, ln2.beta, B * N, C, 1e-5f, cx.st));
    if (cross > 0) CTRY(attn2.fwd(cx, lin, ctx, Nc, hbuf, B, N, colbias, kv_pre, ldkv, rs, ns));
    else CTRY(attn2.fwd(cx, lin, nullptr, 0, hbuf, B, N, nullptr, nullptr, 0, rs, ns));   // encoder_hidden_states=None -> self-attention
    if (!ln_fold) CRUN(dmx_layernorm_fwd(hbuf, l, ln3.gamma, ln3.beta, B * N, C, 1e-5f, cx.st));
    {
      const size_t mk2 = A.mark();
      act_t* gg = A.bf((size_t)B * N * C * 4);
      Epi ef; ef.rowstats_in = rs; ef.nslots = ns;
      if (ff1.geglu) {
        CRUN(linear_fwd(ff1, lin, C, gg, C * 4, (long long)B * N, ef, cx.st));       // [values | gates] -> values * gelu(gates) in the epilogue
      } else {
        act_t* f = A.bf((size_t)B * N * C * 8);
        CRUN(linear_fwd(ff1, lin, C, f, C * 8, (long long)B * N, ef, cx.st));
        CRUN(dmx_geglu(f, gg, (long long)B * N, C * 4, cx.st));
      }
      Epi er; er.flags = EPI_RESID; er.R = hbuf;
      CRUN(linear_fwd(ff2, gg, C * 4, hbuf, C, (long long)B * N, er, cx.st));
      A.release(mk2);
    }
    Epi eo; eo.flags = EPI_RESID; eo.R = x;
    eo.gn_part = out_parts ? gn_parts_buf(*out_parts) : nullptr;
    CRUN(conv_fwd_2d(proj_out, hbuf, out, B, H, W, eo, cx.st));
    if (eo.gn_part) gn_parts_push(cx, out_parts, eo.gn_part, cx.dry ? 0 : dmx_gemm_last_tile_rows(), N, proj_out.Cop, C);
    A.release(mk);
    return DMX_OK;
  }
};

struct UNet : Model {
  dmx_unet_config cfg;
  ConvLayer time1, time2, class_emb, conv_in, conv_out;
  GnLayer norm_out;
  struct Block {
    std::vector<Resnet2D> res;
    std::vector<Transformer2D> attn;
    bool has_attn = false, has_sampler = false;
    ConvLayer sampler;
    int ch = 0;
  };
  std::vector<Block> down, up;
  Resnet2D mid_r0, mid_r1;
  float* gn_partial = nullptr;
  bool up2x = true;
  int temb_ch = 0, tdim = 0, napl = 1;
  int cross_dims[4] = {-1, 0, 0, 0};
  std::vector<Transformer2D> mid_attns;
  // per-call contexts (fp16 copies) for cross-attention transformers
  const act_t* ctx_ptr[4] = {nullptr, nullptr, nullptr, nullptr};
  int ctx_n[4] = {0, 0, 0, 0};
  const float* ctx_bias[4] = {nullptr, nullptr, nullptr, nullptr};

  explicit UNet(const dmx_unet_config& c) : cfg(c) {
    kind = DMX_MODEL_UNET;
    const int nb = c.num_blocks, G = c.norm_num_groups, hd = c.attention_heads;
    const int* boc = c.block_out_channels;
    tdim = boc[0] * 4;
    napl = c.num_attn_per_layer > 0 ? c.num_attn_per_layer : 1;
    for (int q = 0; q < 4; ++q) cross_dims[q] = q < napl ? c.attn_cross_dims[q] : 0;
    temb_ch = c.class_embed_dim > 0 ? tdim * 2 : tdim;
    time1 = make_linear(ps, "time_embedding.linear_1", boc[0], tdim, true, false);
    time2 = make_linear(ps, "time_embedding.linear_2", tdim, tdim, true, false);
    if (c.class_embed_dim > 0) class_emb = make_linear(ps, "class_embedding", c.class_embed_dim, tdim, true, false);
    conv_in = make_conv2d(ps, "conv_in", c.in_channels, boc[0], 3, 1, 1, false);
    int out = boc[0];
    for (int i = 0; i < nb; ++i) {
      const int cin = out;
      out = boc[i];
      Block b;
      b.ch = out; b.has_attn = c.down_attn[i] != 0; b.has_sampler = i != nb - 1;
      const std::string pre = "down_blocks." + std::to_string(i);
      b.res.resize(c.layers_per_block);
      if (b.has_attn) b.attn.resize(c.layers_per_block * napl);
      for (int j = 0; j < c.layers_per_block; ++j) {
        b.res[j].build(ps, pre + ".resnets." + std::to_string(j), j == 0 ? cin : out, out, temb_ch, G, 1e-5f, false);
        if (b.has_attn)
          for (int q = 0; q < napl; ++q)
            b.attn[j * napl + q].build(ps, pre + ".attentions." + std::to_string(j * napl + q), out, hd, cross_dims[q], G);
      }
      if (b.has_sampler) b.sampler = make_conv2d(ps, pre + ".downsamplers.0.conv", out, out, 3, 2, 1, false);
      down.push_back(b);
    }
    const int cm = boc[nb - 1];
    mid_r0.build(ps, "mid_block.resnets.0", cm, cm, temb_ch, G, 1e-5f, false);
    mid_attns.resize(napl);
    for (int q = 0; q < napl; ++q) mid_attns[q].build(ps, "mid_block.attentions." + std::to_string(q), cm, hd, cross_dims[q], G);
    mid_r1.build(ps, "mid_block.resnets.1", cm, cm, temb_ch, G, 1e-5f, false);
    out = boc[nb - 1];
    for (int i = 0; i < nb; ++i) {
      const int prev = out;
      out = boc[nb - 1 - i];
      const int cin = boc[nb - 1 - (i + 1 < nb ? i + 1 : nb - 1)];
      Block b;
      b.ch = out; b.has_attn = c.up_attn[i] != 0; b.has_sampler = i != nb - 1;
      const std::string pre = "up_blocks." + std::to_string(i);
      const int n = c.layers_per_block + 1;
      b.res.resize(n);
      if (b.has_attn) b.attn.resize(n * napl);
      for (int j = 0; j < n; ++j) {
        const int skip = j == n - 1 ? cin : out;
        const int rin = j == 0 ? prev : out;
        b.res[j].build(ps, pre + ".resnets." + std::to_string(j), rin + skip, out, temb_ch, G, 1e-5f, false);
        if (b.has_attn)
          for (int q = 0; q < napl; ++q)
            b.attn[j * napl + q].build(ps, pre + ".attentions." + std::to_string(j * napl + q), out, hd, cross_dims[q], G);
      }
      if (b.has_sampler) b.sampler = make_conv2d(ps, pre + ".upsamplers.0.conv", out, out, 3, 1, 1, false);
      up.push_back(b);
    }
    norm_out.build(ps, "conv_norm_out", boc[0], G, 1e-5f);
    conv_out = make_conv2d(ps, "conv_out", boc[0], c.out_channels, 3, 1, 1, false);
    gn_partial = (float*)ps.dalloc(dmx_gn_scratch_floats(64, 2048, G) * sizeof(float));
    splitk_ws = (float*)ps.dalloc(kSplitKBytes);        // fp32 partial tiles of the split-K low-resolution convolutions
  }

  // the to_k / to_v projections of every cross-attention layer that attends context k, stacked along N: the context is the same for all
  // of them, so ONE (B * tokens) x (sum 2C) GEMM per context and forward replaces two M = 64 ... 256 launches per cross-attention layer
  // (AudioLDM2: 64 launches per forward); each layer takes its [k | v] slice with the stacked row stride
  struct CtxKV { ConvLayer all; std::vector<const AttnLayer*> order; std::vector<int> off; int total = 0; bool ok = false; };
  CtxKV ctxkv[2];
  const act_t* ctx_kv[2] = {nullptr, nullptr};
  bool ctx_kv_on[2] = {false, false};
  int ctxkv_slot(int k, const AttnLayer* a) const {
    for (size_t i = 0; i < ctxkv[k].order.size(); ++i) if (ctxkv[k].order[i] == a) return ctxkv[k].off[i];
    return -1;
  }
  static constexpr size_t kSplitKBytes = 64u << 20;
  float* splitk_ws = nullptr;
  // every resnet's time_emb_proj stacked along N: one (B x temb_ch) x (sum Cout) GEMM per forward instead of 22 M = 16 launches
  ConvLayer temb_all;
  std::vector<const Resnet2D*> temb_order;
  std::vector<int> temb_off;
  int temb_total = 0;
  bool temb_fused = false;
  int temb_slot(const Resnet2D* r) const { for (size_t i = 0; i < temb_order.size(); ++i) if (temb_order[i] == r) return temb_off[i]; return -1; }
  ~UNet() override { dmx_gemm_release_splitk_workspace(splitk_ws); }

  int finalize(hipStream_t st) override {
    CTRY(pack_layer(ps, time1, st)); CTRY(pack_layer(ps, time2, st));
    if (cfg.class_embed_dim > 0) CTRY(pack_layer(ps, class_emb, st));
    CTRY(pack_layer(ps, conv_in, st)); CTRY(pack_layer(ps, conv_out, st));
    norm_out.bind(ps);
    for (auto* blocks : {&down, &up})
      for (auto& b : *blocks) {
        for (auto& r : b.res) CTRY(r.pack(ps, st));
        for (auto& a : b.attn) CTRY(a.pack(ps, st));
        if (b.has_sampler) CTRY(pack_layer(ps, b.sampler, st));
      }
    for (auto& b : up) if (b.has_sampler) CTRY(pack_layer_up2x(ps, b.sampler, st));      // exact x2 upsamplers run folded (layers.hip)
    up2x = getenv("DMX_NO_UP2X") == nullptr;
    CTRY(mid_r0.pack(ps, st)); CTRY(mid_r1.pack(ps, st));
    for (auto& a : mid_attns) CTRY(a.pack(ps, st));
    // stacked context projections (cross-attention k / v)
    for (int k = 0; k < 2; ++k) { ctxkv[k].order.clear(); ctxkv[k].off.clear(); ctxkv[k].total = 0; ctxkv[k].ok = false; }
    if (getenv("DMX_NO_CTXKV_FUSE") == nullptr) {
      auto visit = [&](std::vector<Transformer2D>& tfs) {
        for (size_t i = 0; i < tfs.size(); ++i) {
          if (tfs[i].cross <= 0) continue;
          int k = 0;
          for (int q = 0; q < (int)(i % napl); ++q) if (cross_dims[q] > 0) ++k;       // which context this transformer of its layer attends
          if (k >= 2) continue;
          ctxkv[k].order.push_back(&tfs[i].attn2);
          ctxkv[k].off.push_back(ctxkv[k].total);
          ctxkv[k].total += 2 * tfs[i].attn2.C;
        }
      };
      for (auto& b : down) visit(b.attn);
      visit(mid_attns);
      for (auto& b : up) visit(b.attn);
      for (int k = 0; k < 2; ++k) {
        CtxKV& s = ctxkv[k];
        if (s.order.empty()) continue;
        const int Cip = s.order[0]->to_k.Cip;
        bool same = true;
        for (const AttnLayer* a : s.order) same = same && a->to_k.Cip == Cip && a->to_v.Cip == Cip && a->to_k.Cop == a->C && a->to_v.Cop == a->C && !a->to_k.has_bias && !a->to_v.has_bias;
        if (!same) continue;
        act_t* w = (act_t*)ps.dalloc((size_t)s.total * Cip * sizeof(act_t));
        if (!w) return DMX_ERR_PARAM;
        for (size_t i = 0; i < s.order.size(); ++i) {
          const AttnLayer* a = s.order[i];
          const size_t one = (size_t)a->C * Cip * sizeof(act_t);
          (void)hipMemcpyAsync((char*)w + (size_t)s.off[i] * Cip * sizeof(act_t), a->to_k.wf[0], one, hipMemcpyDeviceToDevice, st);
          (void)hipMemcpyAsync((char*)w + (size_t)s.off[i] * Cip * sizeof(act_t) + one, a->to_v.wf[0], one, hipMemcpyDeviceToDevice, st);
        }
        s.all = s.order[0]->to_k;
        s.all.Co = s.all.Cop = s.total;
        s.all.wf.assign(1, w);
        s.all.has_bias = false;
        s.all.bias = nullptr;
        s.ok = true;
      }
    }
    // fused time-embedding projection
    temb_order.clear(); temb_off.clear(); temb_total = 0;
    for (auto& b : down) for (auto& r : b.res) if (r.has_temb) temb_order.push_back(&r);
    if (mid_r0.has_temb) temb_order.push_back(&mid_r0);
    if (mid_r1.has_temb) temb_order.push_back(&mid_r1);
    for (auto& b : up) for (auto& r : b.res) if (r.has_temb) temb_order.push_back(&r);
    for (const Resnet2D* r : temb_order) { temb_off.push_back(temb_total); temb_total += r->temb.Cop; }
    temb_fused = false;
    if (!temb_order.empty() && getenv("DMX_NO_TEMB_FUSE") == nullptr) {
      const int Cip = temb_order[0]->temb.Cip;
      act_t* w = (act_t*)ps.dalloc((size_t)temb_total * Cip * sizeof(act_t));
      float* bsum = (float*)ps.dalloc((size_t)temb_total * sizeof(float));
      if (!w || !bsum) return DMX_ERR_PARAM;
      bool ok = true;
      for (size_t i = 0; i < temb_order.size(); ++i) {
        const ConvLayer& L = temb_order[i]->temb;
        if (L.Cip != Cip) { ok = false; break; }
        (void)hipMemcpyAsync(w + (size_t)temb_off[i] * Cip, L.wf[0], (size_t)L.Cop * Cip * sizeof(act_t), hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(bsum + temb_off[i], L.bias, (size_t)L.Cop * sizeof(float), hipMemcpyDeviceToDevice, st);
      }
      if (ok) {
        temb_all = temb_order[0]->temb;
        temb_all.Co = temb_all.Cop = temb_total;
        temb_all.wf.assign(1, w);
        temb_all.bias = bsum;
        temb_all.has_bias = true;
        temb_fused = true;
      }
    }
    return DMX_OK;
  }

  struct Skip { act_t* p; int H, W, C; GnParts gp; };

  // runs the napl transformers of one layer in sequence: in -> out (both (B,H,W,ch)); tmp is a scratch of the same size
  // in_parts: GroupNorm partial sums of `in`; out_parts: for `out` (the last transformer's proj_out fills it)
  int run_attn(Ctx& cx, const Transformer2D* tf, const act_t* in, act_t* out, act_t* tmp, int B, int H, int W,
               const GnParts* in_parts = nullptr, GnParts* out_parts = nullptr) {
    const act_t* src = in;
    int kctx = 0;
    GnParts sp = in_parts ? *in_parts : GnParts(), mid[2];
    if (!in_parts) sp.n = -1;
    for (int q = 0; q < napl; ++q) {
      act_t* dst = ((napl - 1 - q) & 1) ? tmp : out;          // last one lands in `out`
      GnParts* dp = out_parts;
      if (q + 1 < napl) { mid[q & 1] = gn_parts_new(cx, B, (size_t)H * W, pad8(tf[q].C)); dp = &mid[q & 1]; }
      const bool cross = tf[q].cross > 0;
      const act_t* kvp = nullptr;
      int ldkv = 0;
      if (cross && kctx < 2 && ctx_kv_on[kctx]) {
        const int off = ctxkv_slot(kctx, &tf[q].attn2);
        if (off >= 0) { kvp = ctx_kv[kctx] + off; ldkv = ctxkv[kctx].total; }
      }
      CTRY(tf[q].fwd(cx, src, dst, B, H, W, cross ? ctx_ptr[kctx] : nullptr, cross ? ctx_n[kctx] : 0, cross ? ctx_bias[kctx] : nullptr, kvp, ldkv,
                     &sp, dp));
      if (cross) ++kctx;
      src = dst;
      if (dp) sp = *dp; else sp.n = -1;
    }
    return DMX_OK;
  }

  int forward(const float* x, const float* t, const float* cls, float* eps, int B, int H0, int W0, void* ws, size_t wsb, hipStream_t st,
              const float* c0 = nullptr, int n0 = 0, const float* c1 = nullptr, int n1 = 0, const float* bias1 = nullptr) {
    if (B > 64) { dmx_set_error("unet: batch > 64 unsupported"); return DMX_ERR_SHAPE; }
    dry = (ws == nullptr);
    arena.reset(ws, dry ? (size_t)-1 : wsb);
    if (!dry) dmx_gemm_set_splitk_workspace(splitk_ws, splitk_ws ? kSplitKBytes : 0);
    Ctx cx{&arena, st, dry, gn_partial};
    Arena& A = arena;
    const int nb = cfg.num_blocks;
    const int* boc = cfg.block_out_channels;
    Epi e;
    // ---- cross-attention contexts -> fp16
    {
      const float* cin[2] = {c0, c1};
      const int cn[2] = {n0, n1};
      int k = 0;
      for (int q = 0; q < napl; ++q) {
        if (cross_dims[q] <= 0) continue;
        if (k >= 2 || (cn[k] & 3) || cn[k] <= 0 || (!dry && !cin[k])) { dmx_set_error("unet: context %d missing or length not a multiple of 4", k); return DMX_ERR_SHAPE; }
        act_t* c16 = A.bf((size_t)B * cn[k] * cross_dims[q]);
        CRUN(dmx_f32_to_bf16(cin[k], c16, (long long)B * cn[k] * cross_dims[q], 1.f, st));
        ctx_ptr[k] = c16; ctx_n[k] = cn[k]; ctx_bias[k] = (k == 1) ? bias1 : nullptr;
        ctx_kv[k] = nullptr; ctx_kv_on[k] = false;
        if (ctxkv[k].ok) {
          act_t* kv = A.bf((size_t)B * cn[k] * ctxkv[k].total);
          Epi ek;
          CRUN(linear_fwd(ctxkv[k].all, c16, ctxkv[k].all.Cip, kv, ctxkv[k].total, (long long)B * cn[k], ek, st));
          ctx_kv[k] = kv; ctx_kv_on[k] = true;
        }
        ++k;
      }
    }
    // ---- embeddings: silu([time_emb | class_emb])
    act_t* semb = A.bf((size_t)B * temb_ch);
    {
      const size_t mk = A.mark();
      act_t* te = A.bf((size_t)B * boc[0]);
      act_t* t1 = A.bf((size_t)B * tdim);
      act_t* emb = A.bf((size_t)B * temb_ch);
      CRUN(dmx_timestep_embed(t, te, B, boc[0], st));
      CRUN(linear_fwd(time1, te, boc[0], t1, tdim, B, e, st));
      CRUN(dmx_silu(t1, t1, (long long)B * tdim, st));
      CRUN(linear_fwd(time2, t1, tdim, emb, temb_ch, B, e, st));
      if (cfg.class_embed_dim > 0) {
        act_t* c16 = A.bf((size_t)B * cfg.class_embed_dim);
        CRUN(dmx_f32_to_bf16(cls, c16, (long long)B * cfg.class_embed_dim, 1.f, st));
        CRUN(linear_fwd(class_emb, c16, cfg.class_embed_dim, emb + tdim, temb_ch, B, e, st));
      }
      CRUN(dmx_silu(emb, semb, (long long)B * temb_ch, st));
      A.release(mk);
    }
    float* rb_all = nullptr;
    if (temb_fused) {
      rb_all = A.f32((size_t)B * temb_total);
      Epi et; et.flags = EPI_F32OUT;
      CRUN(linear_fwd(temb_all, semb, temb_all.Cip, rb_all, temb_total, B, et, st));
    }
    auto rb_of = [&](const Resnet2D& r) -> const float* { const int o = temb_fused ? temb_slot(&r) : -1; return o >= 0 ? rb_all + o : nullptr; };
    // ---- input conv
    int H = H0, W = W0;
    std::vector<Skip> skips;
    const int Cin_p = conv_in.Cip;
    act_t* cur = A.bf((size_t)B * H * W * boc[0]);
    // curp: GroupNorm partial sums of `cur`, written by the launch that produced it (EPI_GNSTATS, blocks.h): the two-launch GroupNorm of
    // the full- and half-resolution levels loses its statistics pass; skips carry theirs to the up path
    // OFF by default in the U-Net (DMX_UNET_GN_PARTS=1 switches it on): a slot's position inside the batch decides where an image
    // boundary cuts it, so two IDENTICAL images of a batch get statistics that differ in the last bits, a few 16-bit activations round
    // the other way and the cond / uncond rows of a CFG batch with equal conditioning no longer cancel exactly -- guidance_scale times
    // that jitter (7e-3 of eps at scale 3.5) for 0.04 ms per forward (profiles/r04_gn_parts.log).  The VAE decoder keeps it.
    cx.gn_parts = getenv("DMX_UNET_GN_PARTS") != nullptr;
    GnParts curp = gn_parts_new(cx, B, (size_t)H * W, pad8(boc[0]));
    {
      const size_t mk = A.mark();
      act_t* x16 = A.bf((size_t)B * H * W * Cin_p);
      CRUN(dmx_nchw_f32_to_nhwc_bf16(x, x16, B, cfg.in_channels, H * W, Cin_p, 1.f, st));
      Epi ei; ei.gn_part = gn_parts_buf(curp);
      CRUN(conv_fwd_2d(conv_in, x16, cur, B, H, W, ei, st));
      if (ei.gn_part) gn_parts_push(cx, &curp, ei.gn_part, dry ? 0 : dmx_gemm_last_tile_rows(), H * W, conv_in.Cop, boc[0]);
      A.release(mk);
    }
    skips.push_back({cur, H, W, boc[0], curp});
    // ---- down
    for (int i = 0; i < nb; ++i) {
      Block& b = down[i];
      for (size_t j = 0; j < b.res.size(); ++j) {
        act_t* y = A.bf((size_t)B * H * W * b.ch);
        GnParts yp = gn_parts_new(cx, B, (size_t)H * W, pad8(b.ch));
        if (b.has_attn) {
          act_t* y2 = A.bf((size_t)B * H * W * b.ch);   // resnet output (transient but simpler to keep)
          act_t* y3 = napl > 1 ? A.bf((size_t)B * H * W * b.ch) : nullptr;
          GnParts y2p = gn_parts_new(cx, B, (size_t)H * W, pad8(b.ch));
          CTRY(b.res[j].fwd(cx, cur, y2, B, H, W, semb, nullptr, rb_of(b.res[j]), temb_total, &curp, &y2p));
          CTRY(run_attn(cx, &b.attn[j * napl], y2, y, y3, B, H, W, &y2p, &yp));
        } else {
          CTRY(b.res[j].fwd(cx, cur, y, B, H, W, semb, nullptr, rb_of(b.res[j]), temb_total, &curp, &yp));
        }
        cur = y; curp = yp;
        skips.push_back({cur, H, W, b.ch, curp});
      }
      if (b.has_sampler) {
        const int H2 = (H + 2 - 3) / 2 + 1, W2 = (W + 2 - 3) / 2 + 1;
        act_t* y = A.bf((size_t)B * H2 * W2 * b.ch);
        GnParts yp = gn_parts_new(cx, B, (size_t)H2 * W2, b.sampler.Cop);
        Epi es; es.gn_part = gn_parts_buf(yp);
        CRUN(conv_fwd_2d(b.sampler, cur, y, B, H, W, es, st));
        if (es.gn_part) gn_parts_push(cx, &yp, es.gn_part, dry ? 0 : dmx_gemm_last_tile_rows(), H2 * W2, b.sampler.Cop, b.ch);
        cur = y; curp = yp; H = H2; W = W2;
        skips.push_back({cur, H, W, b.ch, curp});
      }
    }
    // ---- mid
    {
      const int cm = boc[nb - 1];
      act_t* y0 = A.bf((size_t)B * H * W * cm);
      act_t* y1 = A.bf((size_t)B * H * W * cm);
      act_t* y2 = A.bf((size_t)B * H * W * cm);
      act_t* y3 = napl > 1 ? A.bf((size_t)B * H * W * cm) : nullptr;
      GnParts p0 = gn_parts_new(cx, B, (size_t)H * W, pad8(cm)), p1 = gn_parts_new(cx, B, (size_t)H * W, pad8(cm)),
              p2 = gn_parts_new(cx, B, (size_t)H * W, pad8(cm));
      CTRY(mid_r0.fwd(cx, cur, y0, B, H, W, semb, nullptr, rb_of(mid_r0), temb_total, &curp, &p0));
      CTRY(run_attn(cx, mid_attns.data(), y0, y1, y3, B, H, W, &p0, &p1));
      CTRY(mid_r1.fwd(cx, y1, y2, B, H, W, semb, nullptr, rb_of(mid_r1), temb_total, &p1, &p2));
      cur = y2; curp = p2;
    }
    // ---- up
    int curC = boc[nb - 1];
    for (int i = 0; i < nb; ++i) {
      Block& b = up[i];
      for (size_t j = 0; j < b.res.size(); ++j) {
        const Skip s = skips.back();
        skips.pop_back();
        if (s.H != H || s.W != W) { dmx_set_error("unet skip shape mismatch"); return DMX_ERR_STATE; }
        const int cc = curC + s.C;
        act_t* y = A.bf((size_t)B * H * W * b.ch);
        act_t* y2 = b.has_attn ? A.bf((size_t)B * H * W * b.ch) : nullptr;
        act_t* y3 = (b.has_attn && napl > 1) ? A.bf((size_t)B * H * W * b.ch) : nullptr;
        GnParts yp = gn_parts_new(cx, B, (size_t)H * W, pad8(b.ch)), y2p;
        if (b.has_attn) y2p = gn_parts_new(cx, B, (size_t)H * W, pad8(b.ch));
        const size_t mk = A.mark();
        act_t* cat = A.bf((size_t)B * H * W * cc);
        CRUN(dmx_concat2(cur, s.p, cat, (long long)B * H * W, curC, s.C, st));               // [hidden | skip] in one launch
        const GnParts catp = gn_parts_concat(curp, s.gp, curC);      // the partial sums of both sources describe the concatenation
        if (b.has_attn) {
          CTRY(b.res[j].fwd(cx, cat, y2, B, H, W, semb, nullptr, rb_of(b.res[j]), temb_total, &catp, &y2p));
          CTRY(run_attn(cx, &b.attn[j * napl], y2, y, y3, B, H, W, &y2p, &yp));
        } else {
          CTRY(b.res[j].fwd(cx, cat, y, B, H, W, semb, nullptr, rb_of(b.res[j]), temb_total, &catp, &yp));
        }
        A.release(mk);
        cur = y; curp = yp; curC = b.ch;
      }
      if (b.has_sampler) {
        const Skip nxt = skips.back();                 // upsample to the matching skip's size (forward_upsample_size)
        const int H2 = nxt.H, W2 = nxt.W;
        act_t* y = A.bf((size_t)B * H2 * W2 * b.ch);
        const bool fold = up2x && H2 == 2 * H && W2 == 2 * W;
        GnParts upp;
        float* ubuf[4] = {nullptr, nullptr, nullptr, nullptr};
        if (gn_parts_wanted(cx, (size_t)H2 * W2)) for (int q = 0; q < (fold ? 4 : 1); ++q) ubuf[q] = gn_part_alloc(cx, B, fold ? (size_t)H * W : (size_t)H2 * W2, b.sampler.Cop);
        else upp.n = -1;
        const size_t mk = A.mark();
        if (fold) {
          // exact x2 (the other levels interpolate to the skip tensor's odd size): nearest x2 + conv3x3 as four parity convolutions
          int tms[4] = {0, 0, 0, 0};
          CRUN(conv_up2x_fwd(b.sampler, cur, y, B, H, W, e, st, ubuf[0] ? ubuf : nullptr, tms));
          if (ubuf[0]) for (int q = 0; q < 4; ++q) gn_parts_push(cx, &upp, ubuf[q], tms[q], H * W, b.sampler.Cop, b.ch);
        } else {
          act_t* u = A.bf((size_t)B * H2 * W2 * b.ch);
          CRUN(dmx_upsample_nearest(cur, u, B, H, W, H2, W2, b.ch, st));
          Epi eu; eu.gn_part = ubuf[0];
          CRUN(conv_fwd_2d(b.sampler, u, y, B, H2, W2, eu, st));
          if (ubuf[0]) gn_parts_push(cx, &upp, ubuf[0], dry ? 0 : dmx_gemm_last_tile_rows(), H2 * W2, b.sampler.Cop, b.ch);
        }
        A.release(mk);
        cur = y; curp = upp; H = H2; W = W2;
      }
    }
    // ---- out
    {
      GnTape gt = norm_out.alloc(cx, B);
      act_t* n = A.bf((size_t)B * H * W * boc[0]);
      act_t* o = A.bf((size_t)B * H * W * conv_out.Cop);
      CTRY(norm_out.fwd(cx, cur, n, B, H * W, 1, gt, &curp));
      CRUN(conv_fwd_2d(conv_out, n, o, B, H, W, e, st));
      CRUN(dmx_nhwc_bf16_to_nchw_f32(o, eps, B, cfg.out_channels, H * W, conv_out.Cop, 1.f, st));
    }
    CHECK_WS("unet");
    return DMX_OK;
  }
};

Model* dmx_make_unet(const dmx_unet_config* c) { return new UNet(*c); }
size_t dmx_unet_ws_impl(Model* m, int B, int h, int w, int n0, int n1) {
  UNet* u = static_cast<UNet*>(m);
  u->arena.peak = 0;
  u->forward(nullptr, nullptr, nullptr, nullptr, B, h, w, nullptr, 0, nullptr, nullptr, n0, nullptr, n1, nullptr);
  u->dry = false;
  return u->arena.peak + 256;
}
int dmx_unet_fwd_impl(Model* m, const float* x, const float* t, const float* cls, float* eps, int B, int h, int w, void* ws, size_t wsb,
                      hipStream_t st, const float* c0, int n0, const float* c1, int n1, const float* bias1) {
  const int rc = static_cast<UNet*>(m)->forward(x, t, cls, eps, B, h, w, ws, wsb, st, c0, n0, c1, n1, bias1);
  // the split-K scratch is installed for this executor's launches only: later VAE / HiFi-GAN launches (possibly on other
  // streams) must not pick it up, or their numerics would depend on whether a U-Net ran earlier in the process
  dmx_gemm_set_splitk_workspace(nullptr, 0);
  return rc;
}
