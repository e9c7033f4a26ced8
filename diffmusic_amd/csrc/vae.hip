// AutoencoderKL decoder forward + input-gradient backward (diffusers 0.31.0 semantics, SURVEY.md
// section 8c Appendix B6).  Replaces `vae.decode(1/sf * x0).sample` inside every guided step
// (reference: diffmusic/schedulers/scheduling_dps.py:195-197) and the autograd sweep through it
// (scheduling_dps.py:211-212).  Tape = each resnet's input + mid activation + GroupNorm statistics,
// and q/k/v/P of the single-head mid attention; conv dgrad needs only weights.
#include "blocks.h"

struct VaeDecoder : Model {
  dmx_vae_config cfg;
  ConvLayer post_quant, conv_in, conv_out;
  Resnet2D mid0, mid1;
  GnLayer attn_gn, norm_out;
  ConvLayer to_q, to_k, to_v, to_out;
  std::vector<std::vector<Resnet2D>> up_res;
  std::vector<ConvLayer> up_conv;
  std::vector<int> up_ch;
  float* gn_partial = nullptr;
  int Cmid = 0;
  bool up2x = true;            // fold the nearest x2 upsampling into the 3x3 convolution that follows it
  // tape
  int B = 0, h = 0, w = 0;
  bool have_tape = false;
  ResnetTape t_mid0, t_mid1;
  std::vector<std::vector<ResnetTape>> t_up;
  GnTape t_attn_gn, t_norm_out;
  const act_t* attn_x = nullptr;
  act_t *aq = nullptr, *ak = nullptr, *av = nullptr, *aP = nullptr, *ao = nullptr;   // q, k, v, probabilities, attention output
  const act_t* final_x = nullptr;

  explicit VaeDecoder(const dmx_vae_config& c) : cfg(c) {
    kind = DMX_MODEL_VAE;
    const int nb = c.num_blocks, G = c.norm_num_groups;
    const float eps = c.eps;
    Cmid = c.block_out_channels[nb - 1];
    post_quant = make_conv2d(ps, "post_quant_conv", c.latent_channels, c.latent_channels, 1, 1, 0, true);
    conv_in = make_conv2d(ps, "decoder.conv_in", c.latent_channels, Cmid, 3, 1, 1, true);
    mid0.build(ps, "decoder.mid_block.resnets.0", Cmid, Cmid, 0, G, eps, true);
    const std::string ap = "decoder.mid_block.attentions.0";
    attn_gn.build(ps, ap + ".group_norm", Cmid, G, eps);
    to_q = make_linear(ps, ap + ".to_q", Cmid, Cmid, true, true);
    to_k = make_linear(ps, ap + ".to_k", Cmid, Cmid, true, true);
    to_v = make_linear(ps, ap + ".to_v", Cmid, Cmid, true, true);
    to_out = make_linear(ps, ap + ".to_out.0", Cmid, Cmid, true, true);
    mid1.build(ps, "decoder.mid_block.resnets.1", Cmid, Cmid, 0, G, eps, true);
    int prev = Cmid;
    for (int i = 0; i < nb; ++i) {
      const int ch = c.block_out_channels[nb - 1 - i];
      std::vector<Resnet2D> rs(c.layers_per_block + 1);
      for (int j = 0; j <= c.layers_per_block; ++j)
        rs[j].build(ps, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? prev : ch, ch, 0, G, eps, true);
      up_res.push_back(rs);
      up_ch.push_back(ch);
      if (i != nb - 1) up_conv.push_back(make_conv2d(ps, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", ch, ch, 3, 1, 1, true));
      prev = ch;
    }
    norm_out.build(ps, "decoder.conv_norm_out", prev, G, eps);
    conv_out = make_conv2d(ps, "decoder.conv_out", prev, c.out_channels, 3, 1, 1, true);
    gn_partial = (float*)ps.dalloc(dmx_gn_scratch_floats(64, 2048, G) * sizeof(float));
  }

  int finalize(hipStream_t st) override {
    CTRY(pack_layer(ps, post_quant, st));
    CTRY(pack_layer(ps, conv_in, st));
    CTRY(pack_layer(ps, conv_out, st));
    CTRY(mid0.pack(ps, st));
    CTRY(mid1.pack(ps, st));
    attn_gn.bind(ps); norm_out.bind(ps);
    CTRY(pack_layer(ps, to_q, st));
    CTRY(pack_layer(ps, to_k, st));
    CTRY(pack_layer(ps, to_v, st));
    CTRY(pack_layer(ps, to_out, st));
    for (auto& rs : up_res) for (auto& r : rs) CTRY(r.pack(ps, st));
    for (auto& l : up_conv) { CTRY(pack_layer(ps, l, st)); CTRY(pack_layer_up2x(ps, l, st)); }
    up2x = getenv("DMX_NO_UP2X") == nullptr;
    return DMX_OK;
  }

  // z (B, latent, h, w) fp32 NCHW -> mel (B, H*W) with H = h*2^(nb-1)
  int forward(const float* z, float z_scale, act_t* mel, float* mel_f32, int B_, int h_, int w_, bool keep, void* ws, size_t wsb,
              hipStream_t st) {
    if (B_ > 64) { dmx_set_error("vae: batch > 64 unsupported"); return DMX_ERR_SHAPE; }
    dry = (ws == nullptr);
    arena.reset(ws, dry ? (size_t)-1 : wsb);
    Ctx cx{&arena, st, dry, gn_partial};
    Arena& A = arena;
    B = B_; h = h_; w = w_;
    have_tape = false;
    const int nb = cfg.num_blocks, Lp = post_quant.Cip;
    int H = h, W = w;
    size_t P = (size_t)H * W;
    t_up.assign(nb, std::vector<ResnetTape>(cfg.layers_per_block + 1));
    ResnetTape* nt = nullptr;
    act_t* z16 = A.bf(B * P * Lp);
    act_t* a0 = A.bf(B * P * Lp);
    act_t* x = A.bf(B * P * Cmid);
    CRUN(dmx_nchw_f32_to_nhwc_bf16(z, z16, B, cfg.latent_channels, (int)P, Lp, z_scale, st));
    Epi e0;
    CRUN(conv_fwd_2d(post_quant, z16, a0, B, H, W, e0, st));
    // xp: GroupNorm partial sums of the current tensor x, written by the launch that produced it (EPI_GNSTATS): every GroupNorm of the
    // decoder whose input comes straight out of a GEMM epilogue runs without its statistics pass
    GnParts xp = gn_parts_new(cx, B, P, conv_in.Cop);
    {
      Epi ei; ei.gn_part = gn_parts_buf(xp);
      CRUN(conv_fwd_2d(conv_in, a0, x, B, H, W, ei, st));
      if (ei.gn_part) gn_parts_push(cx, &xp, ei.gn_part, dry ? 0 : dmx_gemm_last_tile_rows(), (int)P, conv_in.Cop, Cmid);
    }
    act_t* y = A.bf(B * P * Cmid);
    GnParts yp = gn_parts_new(cx, B, P, pad8(Cmid));
    CTRY(mid0.fwd(cx, x, y, B, H, W, nullptr, keep ? &t_mid0 : nt, nullptr, 0, &xp, &yp));
    x = y; xp = yp;
    {  // mid attention (one head of dim Cmid)
      const int N = (int)P, C = Cmid;
      attn_x = x;
      t_attn_gn = attn_gn.alloc(cx, B);
      aq = A.bf((size_t)B * N * C); ak = A.bf((size_t)B * N * C); av = A.bf((size_t)B * N * C);
      aP = keep ? A.bf((size_t)B * N * pad8(N)) : nullptr;
      ao = keep ? A.bf((size_t)B * N * C) : nullptr;      // O = P V stays on the tape: delta = rowsum(dO * O) in the backward pass
      act_t* out = A.bf((size_t)B * N * C);
      GnParts op = gn_parts_new(cx, B, P, pad8(C));
      const size_t mk = A.mark();
      act_t* xn = A.bf((size_t)B * N * C);
      act_t* o = keep ? ao : A.bf((size_t)B * N * C);
      CTRY(attn_gn.fwd(cx, x, xn, B, N, 0, t_attn_gn, &xp));
      Epi e;
      CRUN(linear_fwd(to_q, xn, C, aq, C, (long long)B * N, e, st));
      CRUN(linear_fwd(to_k, xn, C, ak, C, (long long)B * N, e, st));
      CRUN(linear_fwd(to_v, xn, C, av, C, (long long)B * N, e, st));
      CTRY(attention_core(cx, aq, ak, av, o, B, N, N, C, 1, aP, nullptr));
      Epi er; er.flags = EPI_RESID; er.R = x;
      er.gn_part = gn_parts_buf(op);
      if (er.gn_part) {       // as a 1x1 convolution over the (H, W) image: the same GEMM, with rows-per-image known to the statistics epilogue
        CRUN(conv_fwd_2d(to_out, o, out, B, H, W, er, st));
        gn_parts_push(cx, &op, er.gn_part, dry ? 0 : dmx_gemm_last_tile_rows(), (int)P, to_out.Cop, C);
      } else {
        CRUN(linear_fwd(to_out, o, C, out, C, (long long)B * N, er, st));
      }
      A.release(mk);
      x = out; xp = op;
    }
    y = A.bf(B * P * Cmid);
    yp = gn_parts_new(cx, B, P, pad8(Cmid));
    CTRY(mid1.fwd(cx, x, y, B, H, W, nullptr, keep ? &t_mid1 : nt, nullptr, 0, &xp, &yp));
    x = y; xp = yp;
    for (int i = 0; i < nb; ++i) {
      const int ch = up_ch[i];
      for (int j = 0; j <= cfg.layers_per_block; ++j) {
        y = A.bf(B * P * ch);
        yp = gn_parts_new(cx, B, P, pad8(ch));
        CTRY(up_res[i][j].fwd(cx, x, y, B, H, W, nullptr, keep ? &t_up[i][j] : nt, nullptr, 0, &xp, &yp));
        x = y; xp = yp;
      }
      if (i != nb - 1) {
        const int H2 = H * 2, W2 = W * 2;
        const size_t P2 = (size_t)H2 * W2;
        y = A.bf(B * P2 * ch);
        // partial sums of the upsampler's output: one region per output-parity launch (each covers P low-resolution positions per image)
        GnParts up;
        float* ubuf[4] = {nullptr, nullptr, nullptr, nullptr};
        if (gn_parts_wanted(cx, P2)) for (int q = 0; q < (up2x ? 4 : 1); ++q) ubuf[q] = gn_part_alloc(cx, B, up2x ? P : P2, up_conv[i].Cop);
        else up.n = -1;
        const size_t mk = A.mark();
        Epi e;
        if (up2x) {
          // nearest x2 + conv3x3 as four 2x2-tap convolutions of the low-resolution tensor (one per output parity): 4/9 of the
          // multiply-adds and no upsampled tensor (Upsample2D, diffusers 0.31.0 semantics)
          int tms[4] = {0, 0, 0, 0};
          CRUN(conv_up2x_fwd(up_conv[i], x, y, B, H, W, e, st, ubuf[0] ? ubuf : nullptr, tms));
          if (ubuf[0]) for (int q = 0; q < 4; ++q) gn_parts_push(cx, &up, ubuf[q], tms[q], (int)P, up_conv[i].Cop, ch);
        } else {
          act_t* u = A.bf(B * P2 * ch);
          CRUN(dmx_upsample_nearest(x, u, B, H, W, H2, W2, ch, st));
          e.gn_part = ubuf[0];
          CRUN(conv_fwd_2d(up_conv[i], u, y, B, H2, W2, e, st));
          if (ubuf[0]) gn_parts_push(cx, &up, ubuf[0], dry ? 0 : dmx_gemm_last_tile_rows(), (int)P2, up_conv[i].Cop, ch);
        }
        A.release(mk);
        x = y; xp = up; H = H2; W = W2; P = P2;
      }
    }
    final_x = x;
    t_norm_out = norm_out.alloc(cx, B);
    {
      const size_t mk = A.mark();
      act_t* n = A.bf(B * P * norm_out.g.C);
      float* m8 = A.f32(B * P * 8);
      CTRY(norm_out.fwd(cx, x, n, B, (int)P, 1, t_norm_out, &xp));
      Epi e; e.flags = EPI_F32OUT;
      CRUN(conv_fwd_2d(conv_out, n, m8, B, H, W, e, st));
      if (mel_f32) CRUN(dmx_gather_col_f32(m8, mel_f32, (long long)B * P, 8, 0, st));
      if (mel) CRUN(dmx_gather_col_f32_to_act(m8, mel, (long long)B * P, 8, 0, st));
      A.release(mk);
    }
    CHECK_WS("vae");
    have_tape = keep;
    return DMX_OK;
  }

  // dmel (B, H*W) fp16 -> dz (B, latent, h, w) fp32 NCHW (times z_scale)
  int backward(const act_t* dmel, float z_scale, float* dz, hipStream_t st) {
    if (!have_tape && !dry) { dmx_set_error("vae backward without a kept forward"); return DMX_ERR_STATE; }
    Ctx cx{&arena, st, dry, gn_partial};
    Arena& A = arena;
    const size_t mk0 = A.mark();
    const int nb = cfg.num_blocks;
    int H = h << (nb - 1), W = w << (nb - 1);
    size_t P = (size_t)H * W;
    Epi e;
    act_t* g8 = A.bf(B * P * 8);
    act_t* gn = A.bf(B * P * norm_out.g.C);
    act_t* g = A.bf(B * P * norm_out.g.C);
    CRUN(dmx_pad_col8_act(dmel, g8, (long long)B * P, st));
    {
      Epi eo;
      float* pbo = norm_out.bwd_epi(cx, eo, final_x, B, P, 1, t_norm_out);
      CRUN(conv_bwd_2d(conv_out, g8, gn, B, H, W, eo, st));
      const GnParts bpo = norm_out.bwd_parts(cx, pbo, P);
      CTRY(norm_out.bwd(cx, final_x, gn, nullptr, g, B, (int)P, 1, t_norm_out, &bpo));
    }
    for (int i = nb - 1; i >= 0; --i) {
      const int ch = up_ch[i];
      if (i != nb - 1) {   // upsampler of block i sits after its resnets: undo it first
        const int Hl = H / 2, Wl = W / 2;
        act_t* gl = A.bf((size_t)B * Hl * Wl * ch);
        const size_t mk = A.mark();
        if (up2x) {
          CRUN(conv_up2x_bwd(up_conv[i], g, gl, B, Hl, Wl, e, st));      // dgrad of the folded convolution: one 4x4-tap stride-2 launch
        } else {
          act_t* gu = A.bf(B * P * ch);
          CRUN(conv_bwd_2d(up_conv[i], g, gu, B, H, W, e, st));
          CRUN(dmx_upsample2x_bwd(gu, gl, B, Hl, Wl, ch, st));
        }
        A.release(mk);
        g = gl; H = Hl; W = Wl; P = (size_t)H * W;
      }
      for (int j = cfg.layers_per_block; j >= 0; --j) {
        act_t* gx = A.bf(B * P * up_res[i][j].Cin);
        CTRY(up_res[i][j].bwd(cx, g, gx, B, H, W, t_up[i][j]));
        g = gx;
      }
    }
    {
      act_t* gx = A.bf(B * P * Cmid);
      CTRY(mid1.bwd(cx, g, gx, B, H, W, t_mid1));
      g = gx;
    }
    {  // attention backward
      const int N = (int)P, C = Cmid;
      if (N & 7) { dmx_set_error("vae attention backward needs h*w %% 8 == 0"); return DMX_ERR_SHAPE; }
      // the softmax backward exists only as the fused epilogue of the dP GEMM on the LDS-DMA 256x256 tile (32-bit buffer offsets):
      // say so here instead of failing inside the launch (B <= 64 keeps every supported shape below the limit)
      if ((long long)B * N * C >= (1ll << 29)) { dmx_set_error("vae attention backward: B*h*w*C >= 2^29 elements unsupported"); return DMX_ERR_SHAPE; }
      const float scale = 1.0f / sqrtf((float)C);
      act_t* gx = A.bf((size_t)B * N * C);
      const size_t mk = A.mark();
      act_t* go = A.bf((size_t)B * N * C);
      act_t* goT = A.bf((size_t)B * N * C);
      act_t* PT = A.bf((size_t)B * N * N);
      float* delta = A.f32((size_t)B * N);
      act_t* dS = A.bf((size_t)B * N * N);
      act_t* T1 = A.bf((size_t)B * N * C);
      act_t* gq = A.bf((size_t)B * N * C);
      act_t* gk = A.bf((size_t)B * N * C);
      act_t* gv = A.bf((size_t)B * N * C);
      act_t* gxn = A.bf((size_t)B * N * C);
      CRUN(linear_bwd(to_out, g, C, go, C, (long long)B * N, e, st));
      GemmBatch gb; gb.Z = B; gb.Zi = 1;
      // dS = P * (go . v^T - delta) * scale with delta = rowsum(go * O) (= rowsum(dP * P)): the softmax backward runs in the
      // epilogue of the dP GEMM, so neither the fp32 dP (512 MB at B = 8) nor a separate softmax-backward pass exists
      CRUN(dmx_rowdot(go, ao, delta, (long long)B * N, C, C, C, st));
      gb.sAo = (long long)N * C; gb.sBo = (long long)N * C; gb.sCo = (long long)N * N;
      {
        Epi es; es.flags = EPI_SOFTBWD; es.X = aP; es.rowbias = delta; es.alpha = scale;
        CRUN(gemm_nt(go, C, av, C, dS, N, N, N, C, es, gb, st));
      }
      // dv = P^T . go = gemm_nt(PT (Nk,Nq), goT (C,Nq))
      CRUN(dmx_transpose(aP, PT, N, N, N, N, B, 1, (long long)N * N, 0, (long long)N * N, 0, st));
      CRUN(dmx_transpose(go, goT, N, C, C, N, B, 1, (long long)N * C, 0, (long long)N * C, 0, st));
      gb.sAo = (long long)N * N; gb.sBo = (long long)N * C; gb.sCo = (long long)N * C;
      CRUN(gemm_nt(PT, N, goT, N, gv, C, N, C, N, e, gb, st));
      // dq = dS . k = gemm_nt(dS (Nq,Nk), kT (C,Nk))
      CRUN(dmx_transpose(ak, T1, N, C, C, N, B, 1, (long long)N * C, 0, (long long)N * C, 0, st));
      CRUN(gemm_nt(dS, N, T1, N, gq, C, N, C, N, e, gb, st));
      // dk = dS^T . q = gemm_nt(dST (Nk,Nq), qT (C,Nq))   (PT buffer reused for dS^T)
      CRUN(dmx_transpose(dS, PT, N, N, N, N, B, 1, (long long)N * N, 0, (long long)N * N, 0, st));
      CRUN(dmx_transpose(aq, T1, N, C, C, N, B, 1, (long long)N * C, 0, (long long)N * C, 0, st));
      CRUN(gemm_nt(PT, N, T1, N, gk, C, N, C, N, e, gb, st));
      // d xn = gq Wq + gk Wk + gv Wv
      Epi ea; ea.flags = EPI_ACCUM;
      CRUN(linear_bwd(to_q, gq, C, gxn, C, (long long)B * N, e, st));
      CRUN(linear_bwd(to_k, gk, C, gxn, C, (long long)B * N, ea, st));
      CRUN(linear_bwd(to_v, gv, C, gxn, C, (long long)B * N, ea, st));
      CTRY(attn_gn.bwd(cx, attn_x, gxn, g, gx, B, N, 0, t_attn_gn));
      A.release(mk);
      g = gx;
    }
    {
      act_t* gx = A.bf(B * P * Cmid);
      CTRY(mid0.bwd(cx, g, gx, B, H, W, t_mid0));
      g = gx;
    }
    const int Lp = post_quant.Cip;
    act_t* g1 = A.bf(B * P * Lp);
    act_t* g2 = A.bf(B * P * Lp);
    CRUN(conv_bwd_2d(conv_in, g, g1, B, H, W, e, st));
    CRUN(conv_bwd_2d(post_quant, g1, g2, B, H, W, e, st));
    CRUN(dmx_nhwc_bf16_to_nchw_f32(g2, dz, B, cfg.latent_channels, (int)P, Lp, z_scale, st));
    CHECK_WS("vae");
    A.release(mk0);
    return DMX_OK;
  }
};

Model* dmx_make_vae(const dmx_vae_config* c) { return new VaeDecoder(*c); }
size_t dmx_vae_ws_impl(Model* m, int B, int h, int w) {
  VaeDecoder* v = static_cast<VaeDecoder*>(m);
  v->arena.peak = 0;
  v->forward(nullptr, 1.f, nullptr, nullptr, B, h, w, true, nullptr, 0, nullptr);
  v->backward(nullptr, 1.f, nullptr, nullptr);
  v->have_tape = false;
  v->dry = false;
  return v->arena.peak + 256;
}
int dmx_vae_fwd_impl(Model* m, const float* z, float zs, act_t* mel, float* mel32, int B, int h, int w, int keep, void* ws, size_t wsb,
                     hipStream_t st) {
  return static_cast<VaeDecoder*>(m)->forward(z, zs, mel, mel32, B, h, w, keep != 0, ws, wsb, st);
}
int dmx_vae_bwd_impl(Model* m, const act_t* dmel, float zs, float* dz, hipStream_t st) {
  return static_cast<VaeDecoder*>(m)->backward(dmel, zs, dz, st);
}
