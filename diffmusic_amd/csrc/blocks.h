// 2-D building blocks shared by the VAE decoder and the U-Net executors: GroupNorm, ResnetBlock2D,
// attention through batched GEMMs.  Semantics follow diffusers 0.31.0 (SURVEY.md section 8c Appendix
// B2/B3/B6); tensors are channels-last (B, P, C) fp16.
//
// Arena discipline: a block first takes its persistent (tape) buffers, then marks the arena, takes
// temporaries and releases back to the mark.  Launches are stream-ordered, so a released temporary
// may be handed out again to a later launch without a hazard.
#pragma once
#include <cstdlib>
#include "models.h"

struct Ctx {           // per-call execution context
  Arena* arena;
  hipStream_t st;
  bool dry;
  float* gn_partial;   // scratch for GroupNorm partial sums (model-owned)
  bool gn_parts = true; // producers write GroupNorm partial sums from their epilogues (EPI_GNSTATS / EPI_GNBWD); the U-Net switches it off
};
#define CRUN(expr) do { if (!cx.dry) { int rc_ = (expr); if (rc_ != DMX_OK) return rc_; } } while (0)
#define CTRY(expr) do { int rc__ = (expr); if (rc__ != DMX_OK) return rc__; } while (0)

struct GnTape { float* stats = nullptr; float* scale = nullptr; float* shift = nullptr; };

// ---- GroupNorm partial sums written by the producers of a tensor (EPI_GNSTATS; kernels.h GnParts): a tensor that will be normalised
// travels with a GnParts: n > 0 = valid regions, 0 = none requested, -1 = a producer could not provide them (the GroupNorm then makes its
// own statistics pass).  Buffers live in the arena next to the tensor they describe.
inline bool gn_parts_wanted(const Ctx& cx, size_t P) {
  const bool on = cx.gn_parts && getenv("DMX_NO_GN_PARTS") == nullptr;      // (read per call: tests compare both paths in one process)
  return on && P > 512;                   // (<= 512 pixels: the single-launch gn_small plan, which needs no partial sums)
}
inline float* gn_part_alloc(Ctx& cx, int B, size_t P, int Np) { return cx.arena->f32(dmx_gn_part_floats(B, (int)P, Np)); }
// after the launch that carried EPI_GNSTATS into `buf` (`tm` = dmx_gemm_last_tile_rows() of it; dry runs record a placeholder)
inline void gn_parts_push(Ctx& cx, GnParts* gp, float* buf, int tm, int P, int Np, int Creal, int qoff = 0) {
  if (!gp || gp->n < 0) return;
  if (cx.dry) tm = 32;
  if (tm > 0 && gp->n < 8) gp->r[gp->n++] = GnRegion{buf, tm, P, Np / 4, qoff, Creal / 4};
  else gp->n = -1;
}
inline GnParts gn_parts_concat(const GnParts& a, const GnParts& b, int Ca) {       // [a | b] along the channels (a has Ca channels)
  GnParts o;
  if (a.n <= 0 || b.n <= 0 || a.n + b.n > 8) { o.n = -1; return o; }
  for (int i = 0; i < a.n; ++i) o.r[o.n++] = a.r[i];
  for (int i = 0; i < b.n; ++i) { o.r[o.n] = b.r[i]; o.r[o.n++].qoff += Ca / 4; }
  return o;
}
inline const GnParts* gn_parts_valid(const GnParts* p) { return p && p->n > 0 ? p : nullptr; }
// a fresh GnParts for a tensor of B x P pixels x Np (padded) channels about to be produced: buffer in r[0].part, n = 0; n = -1 when
// the consumer's GroupNorm would not use partial sums anyway
inline GnParts gn_parts_new(Ctx& cx, int B, size_t P, int Np) {
  GnParts g;
  if (gn_parts_wanted(cx, P)) g.r[0].part = gn_part_alloc(cx, B, P, Np); else g.n = -1;
  return g;
}
inline float* gn_parts_buf(const GnParts& g) { return g.n == 0 ? g.r[0].part : nullptr; }

struct GnLayer {
  GroupNormLayer g;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  void build(ParamStore& ps, const std::string& pre, int C, int G, float eps) { g = make_gn(ps, pre, C, G, eps); }
  void bind(ParamStore& ps) { gamma = ps.dev(g.g_id); beta = ps.dev(g.b_id); }
  GnTape alloc(Ctx& cx, int B) const {
    GnTape t;
    t.stats = cx.arena->f32((size_t)B * g.G * 2);
    t.scale = cx.arena->f32((size_t)B * g.C);
    t.shift = cx.arena->f32((size_t)B * g.C);
    return t;
  }
  int fwd(Ctx& cx, const act_t* x, act_t* y, int B, int P, int silu, const GnTape& t, const GnParts* parts = nullptr) const {
    CRUN(dmx_groupnorm_fwd(x, y, gamma, beta, t.stats, t.scale, t.shift, cx.gn_partial, B, P, g.C, g.G, g.eps, silu, cx.st,
                           gn_parts_valid(parts)));
    return DMX_OK;
  }
  // bparts: backward partial sums written by the dgrad launch that produced dy (bwd_epi below)
  // arm the Epi of the dgrad launch that produces dy of THIS GroupNorm (input x, tape t): the launch writes the backward partial sums into
  // a fresh buffer (EPI_GNBWD); call bwd_parts() after the launch.  Returns the buffer (nullptr: the classic statistics pass will run).
  float* bwd_epi(Ctx& cx, Epi& e, const act_t* x, int B, size_t P, int silu, const GnTape& t) const {
    if (!gn_parts_wanted(cx, P) || ((g.C / g.G) & 3) || (g.C & 7) || getenv("DMX_NO_GN_BWD_PARTS")) return nullptr;
    float* buf = gn_part_alloc(cx, B, P, g.C);
    e.gn_part = buf; e.gnb_x = x; e.gnb_scale = t.scale; e.gnb_shift = t.shift; e.gnb_silu = silu;
    e.gnb_stats = t.stats; e.gnb_cpg = g.C / g.G;
    return buf;
  }
  GnParts bwd_parts(Ctx& cx, float* buf, size_t P) const {
    GnParts gp;
    if (buf) gn_parts_push(cx, &gp, buf, cx.dry ? 0 : dmx_gemm_last_tile_rows(), (int)P, g.C, g.C);
    else gp.n = -1;
    return gp;
  }
  int bwd(Ctx& cx, const act_t* x, const act_t* dy, const act_t* add, act_t* dx, int B, int P, int silu, const GnTape& t,
          const GnParts* bparts = nullptr) const {
    const size_t mk = cx.arena->mark();
    float* k0 = cx.arena->f32((size_t)B * g.C);
    float* k1 = cx.arena->f32((size_t)B * g.C);
    CRUN(dmx_groupnorm_bwd(x, dy, add, dx, t.stats, t.scale, t.shift, k0, k1, cx.gn_partial, B, P, g.C, g.G, silu, cx.st,
                           gn_parts_valid(bparts)));
    cx.arena->release(mk);
    return DMX_OK;
  }
};

struct ResnetTape { const act_t* x = nullptr; act_t* h1 = nullptr; GnTape g1, g2; };

struct Resnet2D {
  int Cin = 0, Cout = 0;
  GnLayer norm1, norm2;
  ConvLayer conv1, conv2, shortcut, temb;
  bool has_shortcut = false, has_temb = false;

  void build(ParamStore& ps, const std::string& pre, int cin, int cout, int temb_ch, int groups, float eps, bool need_bwd) {
    Cin = cin; Cout = cout;
    norm1.build(ps, pre + ".norm1", cin, groups, eps);
    conv1 = make_conv2d(ps, pre + ".conv1", cin, cout, 3, 1, 1, need_bwd);
    if (temb_ch > 0) { temb = make_linear(ps, pre + ".time_emb_proj", temb_ch, cout, true, false); has_temb = true; }
    norm2.build(ps, pre + ".norm2", cout, groups, eps);
    conv2 = make_conv2d(ps, pre + ".conv2", cout, cout, 3, 1, 1, need_bwd);
    if (cin != cout) { shortcut = make_conv2d(ps, pre + ".conv_shortcut", cin, cout, 1, 1, 0, need_bwd); has_shortcut = true; }
  }
  int pack(ParamStore& ps, hipStream_t st) {
    norm1.bind(ps); norm2.bind(ps);
    CTRY(pack_layer(ps, conv1, st));
    CTRY(pack_layer(ps, conv2, st));
    if (has_shortcut) CTRY(pack_layer(ps, shortcut, st));
    if (has_temb) CTRY(pack_layer(ps, temb, st));
    return DMX_OK;
  }
  // x (B,H,W,Cin) -> out (B,H,W,Cout) (caller-allocated).  silu_emb (B, temb_ch) fp16, already SiLU'd.
  // tape != nullptr keeps what backward() needs (persistent arena allocations).
  // rb_pre / ldrb: this block's slice of a time-embedding projection computed for all blocks in one GEMM (U-Net)
  // x_parts: partial sums of x from its producers (norm1 then skips its statistics pass); out_parts: the caller's GnParts for `out` with
  // r[0].part pointing at a buffer of dmx_gn_part_floats(B, P, pad8(Cout)) floats -- conv2 fills it (n = 1) or marks it invalid (n = -1)
  int fwd(Ctx& cx, const act_t* x, act_t* out, int B, int H, int W, const act_t* silu_emb, ResnetTape* tape,
          const float* rb_pre = nullptr, int ldrb = 0, const GnParts* x_parts = nullptr, GnParts* out_parts = nullptr) const {
    Arena& A = *cx.arena;
    const size_t P = (size_t)H * W;
    ResnetTape t;
    t.x = x;
    if (tape) { t.h1 = A.bf(B * P * Cout); t.g1 = norm1.alloc(cx, B); t.g2 = norm2.alloc(cx, B); }
    const size_t mk = A.mark();
    if (!tape) { t.h1 = A.bf(B * P * Cout); t.g1 = norm1.alloc(cx, B); t.g2 = norm2.alloc(cx, B); }
    act_t* n = A.bf(B * P * (Cin > Cout ? Cin : Cout));
    CTRY(norm1.fwd(cx, x, n, B, (int)P, 1, t.g1, x_parts));
    Epi e1;
    GnParts h1p;
    float* h1buf = gn_parts_wanted(cx, P) ? gn_part_alloc(cx, B, P, conv1.Cop) : nullptr;      // conv1 -> norm2
    if (has_temb && rb_pre) {
      e1.flags = EPI_ROWBIAS; e1.rowbias = rb_pre; e1.ldrb = ldrb;
    } else if (has_temb) {
      float* rb = A.f32((size_t)B * Cout);
      Epi et; et.flags = EPI_F32OUT;
      CRUN(linear_fwd(temb, silu_emb, temb.Cip, rb, Cout, B, et, cx.st));
      e1.flags = EPI_ROWBIAS; e1.rowbias = rb;
    }
    e1.gn_part = h1buf;
    CRUN(conv_fwd_2d(conv1, n, t.h1, B, H, W, e1, cx.st));
    if (h1buf) gn_parts_push(cx, &h1p, h1buf, cx.dry ? 0 : dmx_gemm_last_tile_rows(), (int)P, conv1.Cop, Cout);
    CTRY(norm2.fwd(cx, t.h1, n, B, (int)P, 1, t.g2, &h1p));
    Epi e2; e2.flags = EPI_RESID; e2.R = x;
    if (has_shortcut) {
      act_t* sc = A.bf(B * P * Cout);
      Epi es;
      CRUN(conv_fwd_2d(shortcut, x, sc, B, H, W, es, cx.st));
      e2.R = sc;
    }
    float* obuf = out_parts ? gn_parts_buf(*out_parts) : nullptr;
    e2.gn_part = obuf;
    CRUN(conv_fwd_2d(conv2, n, out, B, H, W, e2, cx.st));
    if (obuf) gn_parts_push(cx, out_parts, obuf, cx.dry ? 0 : dmx_gemm_last_tile_rows(), (int)P, conv2.Cop, Cout);
    A.release(mk);
    if (tape) *tape = t;
    return DMX_OK;
  }
  // dout (B,H,W,Cout) -> dx (B,H,W,Cin) (caller-allocated; may not alias dout)
  int bwd(Ctx& cx, const act_t* dout, act_t* dx, int B, int H, int W, const ResnetTape& t) const {
    Arena& A = *cx.arena;
    const size_t P = (size_t)H * W;
    const size_t mk = A.mark();
    act_t* a = A.bf(B * P * Cout);
    act_t* b = A.bf(B * P * (Cin > Cout ? Cin : Cout));
    Epi e;
    // the dgrad launches also write the backward sums of the GroupNorm they feed (EPI_GNBWD): norm2.bwd / norm1.bwd skip their pass over x, dy
    Epi e2;
    float* pb2 = norm2.bwd_epi(cx, e2, t.h1, B, P, 1, t.g2);
    CRUN(conv_bwd_2d(conv2, dout, a, B, H, W, e2, cx.st));                // d n2
    const GnParts bp2 = norm2.bwd_parts(cx, pb2, P);
    CTRY(norm2.bwd(cx, t.h1, a, nullptr, b, B, (int)P, 1, t.g2, &bp2));   // d h1 (in b, Cout channels)
    act_t* c = A.bf(B * P * Cin);
    Epi e1;
    float* pb1 = norm1.bwd_epi(cx, e1, t.x, B, P, 1, t.g1);
    CRUN(conv_bwd_2d(conv1, b, c, B, H, W, e1, cx.st));                   // d n1
    const GnParts bp1 = norm1.bwd_parts(cx, pb1, P);
    const act_t* add = dout;
    if (has_shortcut) {
      CRUN(conv_bwd_2d(shortcut, dout, b, B, H, W, e, cx.st));            // reuse b (Cin channels)
      add = b;
    }
    CTRY(norm1.bwd(cx, t.x, c, add, dx, B, (int)P, 1, t.g1, &bp1));
    A.release(mk);
    return DMX_OK;
  }
};

// Multi-head attention core on projected q (B,Nq,C), k/v (B,Nk,C) with `heads` heads of dim C/heads:
// o = softmax(q k^T * scale [+ colbias]) v, all through batched NT GEMMs with materialised scores.
// P_keep (B*heads, Nq, Nk) fp16 is written to caller memory when backward needs it, else a temp.
inline int attention_core(Ctx& cx, const act_t* q, const act_t* k, const act_t* v, act_t* o, int B, int Nq, int Nk, int C,
                          int heads, act_t* P_keep, const float* colbias, int ldq = 0, int ldk = 0, int ldv = 0) {
  if (ldq <= 0) ldq = C;     // row strides of q / k / v (3C when they are slices of one fused QKV projection)
  if (ldk <= 0) ldk = C;
  if (ldv <= 0) ldv = C;
  Arena& A = *cx.arena;
  const int dh = C / heads, Z = B * heads;
  const int Nkp = pad8(Nk);           // P / vT rows are padded to a multiple of 8 keys (zero columns)
  if (dh & 7) { dmx_set_error("attention needs head_dim %% 8 == 0 (dh=%d)", dh); return DMX_ERR_SHAPE; }
  const float scale = 1.0f / sqrtf((float)dh);
  const size_t mk = A.mark();
  // scores are written by the GEMM epilogue as fp16 straight into the P buffer (row pitch Nkp) and soft-maxed in place:
  // half the HBM traffic of an fp32 score matrix (the N = 1000 U-Net levels are write-bound on it)
  static const bool fused_ok = getenv("DMX_NO_FLASH") == nullptr;
  const bool flash = !P_keep && fused_ok && dmx_flash_attn_ok(dh, C);
  // (the flash kernel masks its key tail per key: any Nk -- an 8 s clip's deepest U-Net level has 25 x 2 = 50 tokens; the materialised
  //  path's softmax walks 4 keys per lane)
  if (!flash && (Nk & 3)) { dmx_set_error("attention through materialised scores needs Nk %% 4 == 0 (Nk=%d)", Nk); return DMX_ERR_SHAPE; }
  if (flash) {
    // forward-only callers (the U-Net): no score matrix at all -- flash_attn.hip walks the keys with an online softmax and takes
    // q, k, v as they come out of the projections (V is transposed inside the kernel on its way into LDS)
    CRUN(dmx_flash_attn_fwd(q, k, v, o, colbias, B, Nq, Nk, C, heads, scale, cx.st, ldq, ldk, ldv));
    A.release(mk);
    return DMX_OK;
  }
  act_t* Pm = P_keep ? P_keep : A.bf((size_t)Z * Nq * Nkp);
  act_t* vT = A.bf((size_t)Z * dh * Nkp);
  GemmBatch gb;
  gb.Z = Z; gb.Zi = heads;
  gb.sAo = (long long)Nq * ldq; gb.sAi = dh;
  gb.sBo = (long long)Nk * ldk; gb.sBi = dh;
  gb.sCo = (long long)heads * Nq * Nkp; gb.sCi = (long long)Nq * Nkp;
  Epi e; e.alpha = scale;
  CRUN(gemm_nt(q, ldq, k, ldk, Pm, Nkp, Nq, Nk, dh, e, gb, cx.st));
  CRUN(dmx_softmax_act(Pm, Pm, colbias, (long long)Z * Nq, Nk, Nkp, heads * Nq, cx.st));
  if (Nkp != Nk && !cx.dry) (void)hipMemsetAsync(vT, 0, (size_t)Z * dh * Nkp * sizeof(act_t), cx.st);
  // vT[z] (dh, Nkp) = v[b, :, h*dh:(h+1)*dh]^T
  CRUN(dmx_transpose(v, vT, Nk, dh, ldv, Nkp, Z, heads, (long long)Nk * ldv, dh, (long long)heads * dh * Nkp, (long long)dh * Nkp, cx.st));
  GemmBatch g2;
  g2.Z = Z; g2.Zi = heads;
  g2.sAo = (long long)heads * Nq * Nkp; g2.sAi = (long long)Nq * Nkp;
  g2.sBo = (long long)heads * dh * Nkp; g2.sBi = (long long)dh * Nkp;
  g2.sCo = (long long)Nq * C; g2.sCi = dh;
  Epi e2;
  CRUN(gemm_nt(Pm, Nkp, vT, Nkp, o, C, Nq, dh, Nkp, e2, g2, cx.st));
  A.release(mk);
  return DMX_OK;
}
