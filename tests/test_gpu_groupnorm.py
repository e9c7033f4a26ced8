"""-m gpu: GroupNorm (+SiLU) forward through the C ABI test hook against torch fp32 GroupNorm on the same fp16-rounded input.
The three launch plans (one workgroup per (group, image) for tiny images; row-coalesced statistics + fused finalize/apply for
the mid-size U-Net levels; statistics / finalize / apply for the big VAE tensors) must agree with the reference and write the
same (mean, rstd) / scale / shift tape.  Semantics: diffusers ResnetBlock2D norm1/norm2 (torch.nn.GroupNorm)."""
import ctypes as C
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,P,Cc,G,silu", [
    (16, 64, 640, 32, 1),        # tiny: single-launch plan
    (16, 252, 384, 32, 1),
    (16, 1000, 256, 32, 1),      # mid: statistics + finalize/apply (U-Net level 1)
    (16, 4000, 128, 32, 1),      # mid, 4 channels per group (U-Net level 0)
    (16, 4000, 384, 32, 0),      # mid, concatenated skip (12 channels per group, 240-thread workgroups)
    (3, 1003, 640, 32, 1),       # mid, ragged chunking, 20 channels per group
    (2, 4000, 512, 32, 1),       # VAE mid block
    (2, 16000, 256, 32, 1),      # big: three-launch plan
    (1, 64000, 128, 32, 0),
])
def test_groupnorm_forward_plans(B, P, Cc, G, silu):
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(P + Cc)
    x = (torch.randn(B, P, Cc, generator=g) * 1.7 + 0.6).to(L.act_dtype()).cuda()
    gamma = (torch.randn(Cc, generator=g) * 0.3 + 1.0).cuda()
    beta = (torch.randn(Cc, generator=g) * 0.2).cuda()
    y = torch.empty_like(x)
    stats = torch.empty(B, G, 2, device="cuda")
    scale = torch.empty(B, Cc, device="cuda")
    shift = torch.empty(B, Cc, device="cuda")
    partial = torch.empty(L.lib().dmx_groupnorm_scratch_floats(B, Cc, G), device="cuda")
    eps = 1e-5
    L.check(L.lib().dmx_groupnorm_raw(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(gamma.data_ptr()),
                                      C.c_void_p(beta.data_ptr()), C.c_void_p(stats.data_ptr()), C.c_void_p(scale.data_ptr()),
                                      C.c_void_p(shift.data_ptr()), C.c_void_p(partial.data_ptr()), B, P, Cc, G, eps, silu,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "groupnorm")
    torch.cuda.synchronize()
    xf = x.float()
    ref = F.group_norm(xf.transpose(1, 2), G, gamma, beta, eps).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    assert (y.float() - ref).abs().max().item() < 2e-2          # fp16 output rounding of values up to ~8
    assert ((y.float() - ref).norm() / ref.norm()).item() < 1e-3
    xg = xf.reshape(B, P, G, Cc // G)
    mean = xg.mean(dim=(1, 3))
    rstd = (xg.var(dim=(1, 3), unbiased=False) + eps).rsqrt()
    assert torch.allclose(stats[..., 0], mean, atol=2e-5, rtol=1e-5)
    assert torch.allclose(stats[..., 1], rstd, atol=0, rtol=2e-5)
    sc_ref = rstd.repeat_interleave(Cc // G, dim=1) * gamma
    sh_ref = beta - mean.repeat_interleave(Cc // G, dim=1) * sc_ref
    assert torch.allclose(scale, sc_ref, atol=1e-6, rtol=3e-5)
    assert torch.allclose(shift, sh_ref, atol=3e-5, rtol=3e-5)


def _conv1x1_with_gnstats(L, x, w, bias, res, P, cfg):
    """(B*P, K) @ w^T + bias + res through dmx_gemm_raw as a 1x1 convolution over images of P pixels, with EPI_GNSTATS.
    Returns (out fp16 (B*P, N), part buffer, rows per slot)."""
    M, K = x.shape
    N = w.shape[0]
    B = M // P
    out = torch.empty(M, N, dtype=L.act_dtype(), device="cuda")
    part = torch.full((L.lib().dmx_groupnorm_part_floats(B, P, N),), float("nan"), device="cuda")
    d = L.GemmDesc()
    d.Z = d.Zi = 1
    d.sy = d.sx = d.osy = d.osx = 1
    d.alpha = 1.0
    for k, v in dict(A=x, W=w, C=out, bias=bias, R=res, gn_part=part).items():
        setattr(d, k, v.data_ptr())
    for k, v in dict(M=M, N=N, K=K, ldw=K, Hi=1, Wi=P, Ci=K, lda=K, Hq=1, Wq=P, ntaps=1, Ho=1, Wo=P, ldc=N, ldr=N, ldx=N, ldc2=N,
                     flags=L.EPI_BIAS | L.EPI_RESID | L.EPI_GNSTATS, tile_cfg=cfg).items():
        setattr(d, k, v)
    L.check(L.lib().dmx_gemm_raw(C.byref(d), C.sizeof(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
    return out, part, L.lib().dmx_gemm_last_tile_rows_raw()


@pytest.mark.parametrize("cfg", [0, 1, 2, 7, 8, 9, 10, 11, 12, 13, 14, 19, 3, 4, 6])
@pytest.mark.parametrize("B,P,K,Cc,silu", [(3, 1003, 64, 256, 1), (2, 4000, 72, 128, 0), (5, 700, 128, 640, 1)])
def test_groupnorm_from_producer_partial_sums(cfg, B, P, K, Cc, silu):
    """EPI_GNSTATS -> gn_parts_kernel: the GEMM that produces a GroupNorm input writes (sum, sum of squares) per wave tile, image and
    4-channel quad from its epilogue; the GroupNorm combines them (Chan, two sweeps) and normalises WITHOUT a statistics pass.  Every
    instantiated tile (P = 1003 / 700 are no multiples of their wave tiles of 32 ... 160 rows: the launch tiles M per image, each
    image's last tile partial), 4 / 8 / 20 channels per group, and the classic path on the same tensor as the second reference."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(cfg * 7 + P)
    G, eps = 32, 1e-5
    x = torch.randn(B * P, K, generator=g).to(L.act_dtype()).cuda()
    w = (torch.randn(Cc, K, generator=g) / K ** 0.5).to(L.act_dtype()).cuda()
    bias = torch.randn(Cc, generator=g).cuda()
    res = (1.5 * torch.randn(B * P, Cc, generator=g) + 0.7).to(L.act_dtype()).cuda()
    out, part, tm = _conv1x1_with_gnstats(L, x, w, bias, res, P, cfg)
    torch.cuda.synchronize()
    assert tm in (32, 48, 64, 80, 96, 128, 160), tm
    gamma = (torch.randn(Cc, generator=g) * 0.3 + 1.0).cuda()
    beta = (torch.randn(Cc, generator=g) * 0.2).cuda()
    y = torch.empty_like(out)
    stats, scale, shift = torch.empty(B, G, 2, device="cuda"), torch.empty(B, Cc, device="cuda"), torch.empty(B, Cc, device="cuda")
    geom = (C.c_int * 6)(tm, P, Cc // 4, 0, Cc // 4, 0)
    parts = (C.c_void_p * 1)(part.data_ptr())
    L.check(L.lib().dmx_groupnorm_parts_raw(C.c_void_p(out.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(gamma.data_ptr()),
                                            C.c_void_p(beta.data_ptr()), C.c_void_p(stats.data_ptr()), C.c_void_p(scale.data_ptr()),
                                            C.c_void_p(shift.data_ptr()), B, P, Cc, G, eps, silu, 1, parts, geom,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "groupnorm_parts")
    torch.cuda.synchronize()
    xf = out.float().reshape(B, P, Cc)
    ref = F.group_norm(xf.transpose(1, 2), G, gamma, beta, eps).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    got = y.float().reshape(B, P, Cc)
    assert ((got - ref).norm() / ref.norm()).item() < 1e-3
    xg = xf.reshape(B, P, G, Cc // G)
    mean, rstd = xg.mean(dim=(1, 3)), (xg.var(dim=(1, 3), unbiased=False) + eps).rsqrt()
    assert torch.allclose(stats[..., 0], mean, atol=3e-5, rtol=1e-5)
    assert torch.allclose(stats[..., 1], rstd, atol=0, rtol=3e-5)


def test_groupnorm_parts_over_a_concatenation_and_parity_regions():
    """Two sources concatenated along the channels (384 + 256 channels, 20 per group: group 19 straddles the seam) and a source that
    arrives as four regions (the output-parity launches of an upsample-folded convolution: each region covers a quarter of the pixels)."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    B, P, G, eps = 3, 1000, 32, 1e-5
    Ca, Cb = 384, 256
    def produce(Cc, Pq, nimg_rows):
        x = torch.randn(B * Pq, 64, generator=g).to(L.act_dtype()).cuda()
        w = (torch.randn(Cc, 64, generator=g) / 8.0).to(L.act_dtype()).cuda()
        bias = torch.randn(Cc, generator=g).cuda()
        res = (torch.randn(B * Pq, Cc, generator=g) + 0.3).to(L.act_dtype()).cuda()
        return _conv1x1_with_gnstats(L, x, w, bias, res, Pq, 0)
    a, pa, tma = produce(Ca, P, P)
    # source b: four quarter-resolution launches whose rows interleave into the P pixels of an image (pixel = 4 q + launch)
    quarters = [produce(Cb, P // 4, P // 4) for _ in range(4)]
    torch.cuda.synchronize()
    bfull = torch.stack([q[0].reshape(B, P // 4, Cb) for q in quarters], dim=2).reshape(B, P, Cb)
    cat = torch.cat([a.reshape(B, P, Ca), bfull], dim=2).contiguous()
    Cc = Ca + Cb
    gamma = (torch.randn(Cc, generator=g) * 0.3 + 1.0).cuda()
    beta = (torch.randn(Cc, generator=g) * 0.2).cuda()
    y = torch.empty_like(cat)
    stats, scale, shift = torch.empty(B, G, 2, device="cuda"), torch.empty(B, Cc, device="cuda"), torch.empty(B, Cc, device="cuda")
    geom = [tma, P, Ca // 4, 0, Ca // 4, 0]
    ptrs = [pa.data_ptr()]
    for q in quarters:
        geom += [q[2], P // 4, Cb // 4, Ca // 4, Cb // 4, 0]
        ptrs.append(q[1].data_ptr())
    L.check(L.lib().dmx_groupnorm_parts_raw(C.c_void_p(cat.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(gamma.data_ptr()),
                                            C.c_void_p(beta.data_ptr()), C.c_void_p(stats.data_ptr()), C.c_void_p(scale.data_ptr()),
                                            C.c_void_p(shift.data_ptr()), B, P, Cc, G, eps, 1, 5, (C.c_void_p * 5)(*ptrs), (C.c_int * 30)(*geom),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "groupnorm_parts")
    torch.cuda.synchronize()
    xf = cat.float()
    ref = F.silu(F.group_norm(xf.transpose(1, 2), G, gamma, beta, eps).transpose(1, 2))
    assert ((y.float() - ref).norm() / ref.norm()).item() < 1e-3
    xg = xf.reshape(B, P, G, Cc // G)
    assert torch.allclose(stats[..., 0], xg.mean(dim=(1, 3)), atol=3e-5, rtol=1e-5)
    assert torch.allclose(stats[..., 1], (xg.var(dim=(1, 3), unbiased=False) + eps).rsqrt(), atol=0, rtol=3e-5)


@pytest.mark.parametrize("cfg", [0, 1, 2, 8, 10, 11, 12, 14, 19, 3, 6])
@pytest.mark.parametrize("B,P,K,Cc,silu", [(3, 1003, 64, 256, 1), (2, 4000, 72, 128, 1), (5, 700, 128, 640, 0)])
def test_groupnorm_backward_from_producer_partial_sums(cfg, B, P, K, Cc, silu):
    """EPI_GNBWD -> gn_bwd_parts_finalize_kernel: the dgrad launch that produces dy of a GroupNorm(+SiLU) writes the two backward sums
    per wave tile / image / quad from its epilogue (sum dxh, sum dxh (x - mean)); the backward then needs no pass over x and dy for them.
    Checked against torch autograd of group_norm(+silu) and against the classic path on the same tensors, with images that are no
    multiple of the wave tiles (P = 1003 / 700: image-aligned M tiling) and 4 / 8 / 20 channels per group."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(cfg * 11 + P)
    G, eps = 32, 1e-5
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = (torch.randn(B, P, Cc, generator=g) * 1.3 + 2.0 * torch.randn(B, 1, Cc, generator=g)).to(L.act_dtype()).cuda()   # per-channel offsets: large group means
    gamma = (torch.randn(Cc, generator=g) * 0.3 + 1.0).cuda()
    beta = (torch.randn(Cc, generator=g) * 0.2).cuda()
    y = torch.empty_like(x)
    stats, scale, shift = torch.empty(B, G, 2, device="cuda"), torch.empty(B, Cc, device="cuda"), torch.empty(B, Cc, device="cuda")
    partial = torch.empty(L.lib().dmx_groupnorm_scratch_floats(B, Cc, G), device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    L.check(L.lib().dmx_groupnorm_raw(p(x), p(y), p(gamma), p(beta), p(stats), p(scale), p(shift), p(partial), B, P, Cc, G, eps, silu, st), "gn")
    # dy = a 1x1 "dgrad" GEMM over images of P pixels, with EPI_GNBWD
    a = torch.randn(B * P, K, generator=g).to(L.act_dtype()).cuda()
    w = (torch.randn(Cc, K, generator=g) / K ** 0.5).to(L.act_dtype()).cuda()
    dy = torch.empty(B * P, Cc, dtype=L.act_dtype(), device="cuda")
    part = torch.full((L.lib().dmx_groupnorm_part_floats(B, P, Cc),), float("nan"), device="cuda")
    d = L.GemmDesc()
    d.Z = d.Zi = 1
    d.sy = d.sx = d.osy = d.osx = 1
    d.alpha = 1.0
    for k, v in dict(A=a, W=w, C=dy, gn_part=part, gnb_x=x, gnb_scale=scale, gnb_shift=shift, gnb_stats=stats).items():
        setattr(d, k, v.data_ptr())
    for k, v in dict(M=B * P, N=Cc, K=K, ldw=K, Hi=1, Wi=P, Ci=K, lda=K, Hq=1, Wq=P, ntaps=1, Ho=1, Wo=P, ldc=Cc, ldr=Cc, ldx=Cc, ldc2=Cc,
                     gnb_ldx=Cc, gnb_silu=silu, gnb_cpg=Cc // G, flags=L.EPI_GNBWD, tile_cfg=cfg).items():
        setattr(d, k, v)
    L.check(L.lib().dmx_gemm_raw(C.byref(d), C.sizeof(d), st), "gemm")
    tm = L.lib().dmx_gemm_last_tile_rows_raw()
    assert tm in (32, 48, 64, 96, 128), tm
    add = torch.randn(B, P, Cc, generator=g).to(L.act_dtype()).cuda()
    k0, k1 = torch.empty(B, Cc, device="cuda"), torch.empty(B, Cc, device="cuda")
    dx_parts, dx_classic = torch.empty_like(x), torch.empty_like(x)
    geom = (C.c_int * 6)(tm, P, Cc // 4, 0, Cc // 4, 0)
    L.check(L.lib().dmx_groupnorm_bwd_raw(p(x), p(dy), p(add), p(dx_parts), p(stats), p(scale), p(shift), p(k0), p(k1), p(partial), B, P, Cc, G,
                                          silu, 1, (C.c_void_p * 1)(part.data_ptr()), geom, st), "gn_bwd parts")
    L.check(L.lib().dmx_groupnorm_bwd_raw(p(x), p(dy), p(add), p(dx_classic), p(stats), p(scale), p(shift), p(k0), p(k1), p(partial), B, P, Cc, G,
                                          silu, 0, None, None, st), "gn_bwd classic")
    torch.cuda.synchronize()
    xr = x.float().requires_grad_(True)
    yr = F.group_norm(xr.transpose(1, 2), G, gamma, beta, eps).transpose(1, 2)
    if silu:
        yr = F.silu(yr)
    (gref,) = torch.autograd.grad((yr * dy.float().reshape(B, P, Cc)).sum(), xr)
    gref = gref + add.float()
    rel = lambda u, v: ((u.float() - v.float()).norm() / v.float().norm()).item()
    assert rel(dx_classic, gref) < 2e-3
    assert rel(dx_parts, gref) < 2e-3
    assert rel(dx_parts, dx_classic) < 1e-3


@pytest.mark.parametrize("cfg", [1, 2, 19, 11, 12, 3])
def test_producer_partial_sums_do_not_depend_on_the_batch_position(cfg):
    """The same image at batch positions 0, 2 and 4 of a launch whose images (P = 1003 pixels) are no multiple of the wave tile's rows:
    its slots -- and the statistics combined from them -- are bit-identical (image-aligned M tiling of the GroupNorm-statistics launches)."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(cfg)
    B, P, K, Cc, G, eps = 5, 1003, 64, 256, 32, 1e-5
    xi = torch.randn(3, P, K, generator=g)
    x = torch.stack([xi[0], xi[1], xi[0], xi[2], xi[0]]).reshape(B * P, K).to(L.act_dtype()).cuda()
    w = (torch.randn(Cc, K, generator=g) / K ** 0.5).to(L.act_dtype()).cuda()
    bias = torch.randn(Cc, generator=g).cuda()
    ri = (1.5 * torch.randn(3, P, Cc, generator=g) + 0.7)
    res = torch.stack([ri[0], ri[1], ri[0], ri[2], ri[0]]).reshape(B * P, Cc).to(L.act_dtype()).cuda()
    out, part, tm = _conv1x1_with_gnstats(L, x, w, bias, res, P, cfg)
    torch.cuda.synchronize()
    assert tm > 0 and P % tm != 0
    o = out.reshape(B, P, Cc)
    assert torch.equal(o[0], o[2]) and torch.equal(o[0], o[4])
    slots = (P + tm - 1) // tm + 1
    pr = part[: B * slots * (Cc // 4) * 2].reshape(B, slots, Cc // 4, 2)[:, : (P + tm - 1) // tm]
    assert bool(torch.isfinite(pr).all())
    assert torch.equal(pr[0], pr[2]) and torch.equal(pr[0], pr[4]) and not torch.equal(pr[0], pr[1])
    gamma, beta = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    y = torch.empty_like(out)
    stats, scale, shift = torch.empty(B, G, 2, device="cuda"), torch.empty(B, Cc, device="cuda"), torch.empty(B, Cc, device="cuda")
    L.check(L.lib().dmx_groupnorm_parts_raw(C.c_void_p(out.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(gamma.data_ptr()),
                                            C.c_void_p(beta.data_ptr()), C.c_void_p(stats.data_ptr()), C.c_void_p(scale.data_ptr()),
                                            C.c_void_p(shift.data_ptr()), B, P, Cc, G, eps, 1, 1, (C.c_void_p * 1)(part.data_ptr()),
                                            (C.c_int * 6)(tm, P, Cc // 4, 0, Cc // 4, 0), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gn parts")
    torch.cuda.synchronize()
    assert torch.equal(stats[0], stats[2]) and torch.equal(stats[0], stats[4])
    yy = y.reshape(B, P, Cc)
    assert torch.equal(yy[0], yy[2]) and torch.equal(yy[0], yy[4])


def test_groupnorm_parts_raw_refuses_what_it_cannot_run():
    """No scratch for the classic pass behind this entry point: channel counts per group that are no multiple of 4 and regions that
    describe no producer launch are DMX_ERR_SHAPE, not a fault on the device."""
    from diffmusic_amd import _lib as L
    B, P, Cc, G = 2, 256, 96, 32                      # 3 channels per group
    x = torch.zeros(B, P, Cc, dtype=L.act_dtype(), device="cuda")
    y = torch.empty_like(x)
    f = lambda *s: torch.zeros(*s, device="cuda")
    part = f(L.lib().dmx_groupnorm_part_floats(B, P, Cc))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    args = lambda Cc_, G_, geom: (p(x), p(y), p(f(Cc_)), p(f(Cc_)), p(f(B, G_, 2)), p(f(B, Cc_)), p(f(B, Cc_)), B, P, Cc_, G_, 1e-5, 0, 1,
                                  (C.c_void_p * 1)(part.data_ptr()), (C.c_int * 6)(*geom), st)
    assert L.lib().dmx_groupnorm_parts_raw(*args(96, 32, (64, P, 24, 0, 24, 0))) != 0
    assert L.lib().dmx_groupnorm_parts_raw(*args(128, 32, (0, P, 32, 0, 32, 0))) != 0        # tm = 0
    assert L.lib().dmx_groupnorm_parts_raw(*args(128, 32, (64, 32, 32, 0, 32, 0))) != 0       # P < tm
    assert L.lib().dmx_groupnorm_parts_raw(*args(128, 32, (64, P, 32, 8, 32, 0))) != 0        # quads past C / 4
    torch.cuda.synchronize()


def test_producer_partial_sums_are_the_same_bits_from_every_statistics_tile():
    """A slot is a 64-row chunk of a 64-column wave tile, summed in one order by every tile that carries statistics (256 x 256, 256 x 128,
    512 x 128, 128 x 128 LDS-DMA tiles, the register-staged 128 x 128 tile; other choices are mapped onto these): the partial sums and the
    statistics combined from them do not depend on the tile the cost model picks for a batch size."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(99)
    B, P, K, Cc = 3, 1003, 128, 256
    x = torch.randn(B * P, K, generator=g).to(L.act_dtype()).cuda()
    w = (torch.randn(Cc, K, generator=g) / K ** 0.5).to(L.act_dtype()).cuda()
    bias = torch.randn(Cc, generator=g).cuda()
    res = (1.5 * torch.randn(B * P, Cc, generator=g) + 0.7).to(L.act_dtype()).cuda()
    ref = None
    for cfg in (1, 2, 19, 11, 18, 3, 7, 10, 12, 6):
        out, part, tm = _conv1x1_with_gnstats(L, x, w, bias, res, P, cfg)
        torch.cuda.synchronize()
        assert tm == 64
        slots = (P + 63) // 64 + 1
        pr = part[: B * slots * (Cc // 4) * 2].reshape(B, slots, Cc // 4, 2)[:, : (P + 63) // 64].clone()
        assert bool(torch.isfinite(pr).all())
        if ref is None:
            ref = (out.clone(), pr)
        else:
            assert torch.equal(out, ref[0]), f"cfg {cfg}: output differs"
            assert torch.equal(pr, ref[1]), f"cfg {cfg}: partial sums differ from cfg 1"
