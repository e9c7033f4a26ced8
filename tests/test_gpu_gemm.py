"""-m gpu: the implicit-GEMM kernel (through the C ABI test hook dmx_gemm_raw) against torch fp32
convolutions on the same bf16-rounded operands."""
import ctypes as C
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _adt():
    from diffmusic_amd import _lib as L
    return L.act_dtype()


def _desc(L, **kw):
    d = L.GemmDesc()
    d.Z = d.Zi = 1
    d.sy = d.sx = d.osy = d.osx = 1
    d.alpha = 1.0
    for k, v in kw.items():
        if k in ("tdy", "tdx"):
            for i, t in enumerate(v):
                getattr(d, k)[i] = t
        elif isinstance(v, torch.Tensor):
            setattr(d, k, v.data_ptr())
        else:
            setattr(d, k, v)
    return d


def _run(L, d):
    L.check(L.lib().dmx_gemm_raw(C.byref(d), C.sizeof(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
    torch.cuda.synchronize()


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("B,T,Ci,Co,k,dil", [(2, 333, 64, 128, 3, 1), (1, 1000, 32, 32, 11, 5), (3, 257, 128, 64, 7, 3),
                                             (1, 130, 8, 16, 7, 1), (2, 77, 512, 512, 3, 1),
                                             # M >= 4096: LDS-DMA large-tile kernels (256x256 and 256x128)
                                             (2, 2500, 128, 256, 7, 3), (2, 3001, 64, 128, 11, 5), (1, 5001, 512, 512, 3, 1),
                                             (3, 1777, 40, 384, 3, 1)])
def test_conv1d_dilated(B, T, Ci, Co, k, dil):
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, Ci, generator=g).to(_adt()).cuda()
    w = (torch.randn(Co, Ci, k, generator=g) / (Ci * k) ** 0.5).to(_adt()).cuda()
    bias = torch.randn(Co, generator=g).cuda()
    res = torch.randn(B, T, Co, generator=g).to(_adt()).cuda()
    pad = (k * dil - dil) // 2
    wp = w.permute(0, 2, 1).reshape(Co, k * Ci).contiguous()
    out = torch.empty(B, T, Co, dtype=_adt(), device="cuda")
    out2 = torch.empty_like(out)
    d = _desc(L, A=x, W=wp, C=out, C2=out2, bias=bias, R=res, M=B * T, N=Co, K=k * Ci, ldw=k * Ci, Hi=1, Wi=T, Ci=Ci, lda=Ci,
              Hq=1, Wq=T, ntaps=k, Ho=1, Wo=T, ldc=Co, ldr=Co, ldx=Co, ldc2=Co,
              flags=L.EPI_BIAS | L.EPI_RESID | L.EPI_LRELU2, act_slope=0.1,
              tdy=[0] * k, tdx=[t * dil - pad for t in range(k)])
    _run(L, d)
    ref = F.conv1d(x.float().transpose(1, 2), w.float(), bias, padding=pad, dilation=dil).transpose(1, 2) + res.float()
    assert _rel(out, ref) < 6e-3
    assert _rel(out2, F.leaky_relu(ref, 0.1)) < 6e-3


def test_conv2d_3x3_and_stride2():
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    B, H, W, Ci, Co = 2, 37, 16, 64, 96
    x = torch.randn(B, H, W, Ci, generator=g).to(_adt()).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).to(_adt()).cuda()
    wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
    for stride in (1, 2):
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        out = torch.empty(B, Ho, Wo, Co, dtype=_adt(), device="cuda")
        d = _desc(L, A=x, W=wp, C=out, M=B * Ho * Wo, N=Co, K=9 * Ci, ldw=9 * Ci, Hi=H, Wi=W, Ci=Ci, lda=Ci, Hq=Ho, Wq=Wo,
                  sy=stride, sx=stride, ntaps=9, Ho=Ho, Wo=Wo, ldc=Co, tdy=[t // 3 - 1 for t in range(9)],
                  tdx=[t % 3 - 1 for t in range(9)])
        _run(L, d)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, stride=stride, padding=1).permute(0, 2, 3, 1)
        assert _rel(out, ref) < 6e-3, stride


def test_batched_nt_gemm_f32_out_and_mask():
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    Z, M, N, K = 6, 250, 252, 48
    a = torch.randn(Z, M, K, generator=g).to(_adt()).cuda()
    b = torch.randn(Z, N, K, generator=g).to(_adt()).cuda()
    out = torch.empty(Z, M, N, dtype=torch.float32, device="cuda")
    d = _desc(L, A=a, W=b, C=out, M=M, N=N, K=K, ldw=K, Hi=1, Wi=M, Ci=K, lda=K, Hq=1, Wq=M, ntaps=1, Ho=1, Wo=M, ldc=N,
              Z=Z, Zi=3, sAo=3 * M * K, sAi=M * K, sWo=3 * N * K, sWi=N * K, sCo=3 * M * N, sCi=M * N, alpha=0.25,
              flags=L.EPI_F32OUT, tdy=[0], tdx=[0])
    _run(L, d)
    ref = 0.25 * a.float() @ b.float().transpose(1, 2)
    assert _rel(out, ref) < 1e-5
    # mask epilogue
    xm = torch.randn(Z, M, N, generator=g).to(_adt()).cuda()
    outb = torch.empty(Z, M, N, dtype=_adt(), device="cuda")
    d.C = outb.data_ptr(); d.X = xm.data_ptr(); d.ldx = N; d.flags = L.EPI_MASK; d.mask_slope = 0.1; d.alpha = 1.0
    _run(L, d)
    ref = (a.float() @ b.float().transpose(1, 2)) * torch.where(xm.float() > 0, 1.0, 0.1)
    assert _rel(outb, ref) < 6e-3


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19])
def test_forced_tile_configs_agree(cfg):
    """every tile configuration (incl. the 320x256 and 512x128 LDS-DMA tiles and the register-staged 128x32 tile) gives the same conv result"""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    B, T, Ci, Co, k, dil = 2, 2400, 128, 256, 7, 3
    x = torch.randn(B, T, Ci, generator=g).to(_adt()).cuda()
    w = (torch.randn(Co, Ci, k, generator=g) / (Ci * k) ** 0.5).to(_adt()).cuda()
    res = torch.randn(B, T, Co, generator=g).to(_adt()).cuda()
    pad = (k * dil - dil) // 2
    wp = w.permute(0, 2, 1).reshape(Co, k * Ci).contiguous()
    out = torch.empty(B, T, Co, dtype=_adt(), device="cuda")
    d = _desc(L, A=x, W=wp, C=out, R=res, M=B * T, N=Co, K=k * Ci, ldw=k * Ci, Hi=1, Wi=T, Ci=Ci, lda=Ci, Hq=1, Wq=T, ntaps=k, Ho=1, Wo=T,
              ldc=Co, ldr=Co, ldx=Co, ldc2=Co, flags=L.EPI_RESID, tdy=[0] * k, tdx=[t * dil - pad for t in range(k)], tile_cfg=cfg)
    _run(L, d)
    ref = F.conv1d(x.float().transpose(1, 2), w.float(), None, padding=pad, dilation=dil).transpose(1, 2) + res.float()
    assert _rel(out, ref) < 3e-3


def _conv_desc(L, x, wp, k, dil, Cc, B, T, flip=False, **kw):
    pad = (k * dil - dil) // 2
    tdx = [(pad - t * dil) if flip else (t * dil - pad) for t in range(k)]
    return _desc(L, A=x, W=wp, M=B * T, N=Cc, K=k * Cc, ldw=k * Cc, Hi=1, Wi=T, Ci=Cc, lda=Cc, Hq=1, Wq=T, ntaps=k, Ho=1, Wo=T,
                 ldc=Cc, ldr=Cc, ldx=Cc, ldc2=Cc, tdy=[0] * k, tdx=tdx, **kw)


def _pair_run(L, da, db):
    L.check(L.lib().dmx_conv_pair_raw(C.byref(da) if da is not None else None, C.byref(db), C.sizeof(db),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "conv_pair")
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,T,Cc,k,dil", [(2, 1000, 64, 3, 1), (1, 2049, 64, 11, 5), (3, 700, 32, 7, 3), (2, 5000, 32, 11, 5),
                                          (1, 100, 32, 3, 1), (2, 256, 64, 7, 1), (1, 4097, 64, 3, 5),
                                          (2, 1000, 128, 3, 1), (1, 2049, 128, 11, 5), (2, 777, 128, 7, 3)])
def test_fused_resblock_pair_forward(B, T, Cc, k, dil):
    """conv1(dilated) -> leaky-relu -> conv2 -> + reconstructed residual, the HiFi-GAN resblock step
    (modeling_speecht5.py HifiGanResidualBlock.forward), fused in one launch; checked against torch fp32 and
    against the same two stages run as separate implicit-GEMM launches."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    slope = 0.1
    xa = F.leaky_relu(torch.randn(B, T, Cc, generator=g), slope).to(_adt()).cuda()       # the stored (activated) residual stream
    w1 = (torch.randn(Cc, Cc, k, generator=g) / (Cc * k) ** 0.5).to(_adt()).cuda()
    w2 = (torch.randn(Cc, Cc, k, generator=g) / (Cc * k) ** 0.5).to(_adt()).cuda()
    b1, b2 = torch.randn(Cc, generator=g).cuda() * 0.1, torch.randn(Cc, generator=g).cuda() * 0.1
    w1p = w1.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous()
    w2p = w2.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous()
    outs = {}
    for mode in ("fused", "separate"):
        ha = torch.zeros(B, T, Cc, dtype=_adt(), device="cuda")
        xn = torch.zeros_like(ha)
        raw = torch.zeros_like(ha)
        da = _conv_desc(L, xa, w1p, k, dil, Cc, B, T, C=ha, C2=ha, bias=b1, flags=L.EPI_BIAS | L.EPI_LRELU2 | L.EPI_NO_C, act_slope=slope)
        db = _conv_desc(L, ha, w2p, k, 1, Cc, B, T, C=raw, C2=xn, bias=b2, R=xa, resid_inv_slope=1.0 / slope, act_slope=slope,
                        flags=L.EPI_BIAS | L.EPI_RESID | L.EPI_RESID_INV | L.EPI_LRELU2)
        if mode == "fused":
            _pair_run(L, da, db)
        else:
            _run(L, da)
            _run(L, db)
        outs[mode] = (ha, raw, xn)
    x = torch.where(xa.float() > 0, xa.float(), xa.float() / slope)
    h = F.leaky_relu(F.conv1d(xa.float().transpose(1, 2), w1.float(), b1, padding=(k * dil - dil) // 2, dilation=dil), slope)
    hq = h.to(_adt()).float()
    y = F.conv1d(hq, w2.float(), b2, padding=(k - 1) // 2).transpose(1, 2) + x
    ha, raw, xn = outs["fused"]
    assert _rel(ha, h.transpose(1, 2)) < 4e-3
    assert _rel(raw, y) < 6e-3
    assert _rel(xn, F.leaky_relu(y, slope)) < 6e-3
    for a, b in zip(outs["fused"], outs["separate"]):
        assert _rel(a, b) < 2e-3


@pytest.mark.parametrize("B,T,Cc,k,dil", [(2, 1000, 64, 3, 1), (1, 2049, 64, 11, 5), (3, 700, 32, 7, 3), (2, 3000, 32, 11, 5),
                                          (1, 100, 64, 7, 5), (2, 1000, 128, 3, 1), (1, 2049, 128, 11, 5), (1, 300, 128, 7, 3)])
def test_fused_resblock_pair_backward(B, T, Cc, k, dil):
    """dgrad(conv2) -> leaky-relu' -> dgrad(conv1) -> leaky-relu' + residual, accumulated into an existing gradient."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(4)
    slope = 0.1
    gc = torch.randn(B, T, Cc, generator=g).to(_adt()).cuda()
    ha = torch.randn(B, T, Cc, generator=g).to(_adt()).cuda()
    xa = torch.randn(B, T, Cc, generator=g).to(_adt()).cuda()
    prev = torch.randn(B, T, Cc, generator=g).to(_adt()).cuda()
    w1 = (torch.randn(Cc, Cc, k, generator=g) / (Cc * k) ** 0.5).to(_adt()).cuda()
    w2 = (torch.randn(Cc, Cc, k, generator=g) / (Cc * k) ** 0.5).to(_adt()).cuda()
    w1b = w1.permute(1, 2, 0).reshape(Cc, k * Cc).contiguous()          # dgrad packing: [Cin][tap][Cout]
    w2b = w2.permute(1, 2, 0).reshape(Cc, k * Cc).contiguous()
    outs = {}
    for mode in ("fused", "separate"):
        ghk = torch.zeros(B, T, Cc, dtype=_adt(), device="cuda")
        dst = prev.clone()
        da = _conv_desc(L, gc, w2b, k, 1, Cc, B, T, flip=True, C=ghk, X=ha, flags=L.EPI_MASK, mask_slope=slope)
        db = _conv_desc(L, ghk, w1b, k, dil, Cc, B, T, flip=True, C=dst, X=xa, R=gc, mask_slope=slope,
                        flags=L.EPI_MASK | L.EPI_RESID | L.EPI_ACCUM)
        if mode == "fused":
            _pair_run(L, da, db)
        else:
            _run(L, da)
            _run(L, db)
        outs[mode] = dst
    gcf = gc.float().transpose(1, 2)
    m_h = torch.where(ha.float() > 0, 1.0, slope).transpose(1, 2)
    m_x = torch.where(xa.float() > 0, 1.0, slope).transpose(1, 2)
    g1 = F.conv_transpose1d(gcf, w2.float(), padding=(k - 1) // 2) * m_h
    g1 = g1.to(_adt()).float()
    g0 = F.conv_transpose1d(g1, w1.float(), padding=(k * dil - dil) // 2, dilation=dil) * m_x + gcf
    ref = g0.transpose(1, 2) + prev.float()
    assert _rel(outs["fused"], ref) < 6e-3
    assert _rel(outs["fused"], outs["separate"]) < 3e-3


def test_slab_conv_single_stage():
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    B, T, Cc, k, dil = 2, 1500, 64, 7, 3
    x = torch.randn(B, T, Cc, generator=g).to(_adt()).cuda()
    w = (torch.randn(Cc, Cc, k, generator=g) / (Cc * k) ** 0.5).to(_adt()).cuda()
    bias = torch.randn(Cc, generator=g).cuda()
    out = torch.zeros(B, T, Cc, dtype=_adt(), device="cuda")
    d = _conv_desc(L, x, w.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous(), k, dil, Cc, B, T, C=out, bias=bias, flags=L.EPI_BIAS)
    _pair_run(L, None, d)
    ref = F.conv1d(x.float().transpose(1, 2), w.float(), bias, padding=(k * dil - dil) // 2, dilation=dil).transpose(1, 2)
    assert _rel(out, ref) < 4e-3


@pytest.mark.parametrize("B,H,W,Ci,Co", [(16, 8, 8, 640, 640), (4, 16, 16, 384, 384), (16, 8, 8, 1280, 640)])
def test_split_k_small_m_conv(B, H, W, Ci, Co):
    """3x3 convolution of a low-resolution U-Net level (M = B*H*W <= 4096, K = 9*Ci): with the split-K scratch installed the
    launch runs as K slices + a reduce/epilogue kernel; same result as the single-pass kernel and as torch fp32."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, H, W, Ci, generator=g).to(_adt()).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).to(_adt()).cuda()
    wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
    bias = torch.randn(Co, generator=g).cuda()
    rowb = torch.randn(B, Co, generator=g).cuda()
    res = torch.randn(B, H, W, Co, generator=g).to(_adt()).cuda()
    outs = []
    scratch = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    for ws in (None, scratch):
        L.check(L.lib().dmx_gemm_splitk_workspace(C.c_void_p(ws.data_ptr()) if ws is not None else None, ws.numel() * 4 if ws is not None else 0), "ws")
        out = torch.zeros(B, H, W, Co, dtype=_adt(), device="cuda")
        d = _desc(L, A=x, W=wp, C=out, bias=bias, rowbias=rowb, R=res, M=B * H * W, N=Co, K=9 * Ci, ldw=9 * Ci, Hi=H, Wi=W, Ci=Ci, lda=Ci,
                  Hq=H, Wq=W, ntaps=9, Ho=H, Wo=W, ldc=Co, ldr=Co, ldx=Co, ldc2=Co, flags=L.EPI_BIAS | L.EPI_ROWBIAS | L.EPI_RESID,
                  tdy=[t // 3 - 1 for t in range(9)], tdx=[t % 3 - 1 for t in range(9)])
        _run(L, d)
        outs.append(out)
    L.check(L.lib().dmx_gemm_splitk_workspace(None, 0), "ws")
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1) + rowb[:, None, None, :] + res.float()
    assert _rel(outs[1], ref) < 3e-3
    assert _rel(outs[0], ref) < 3e-3
    assert _rel(outs[1], outs[0]) < 2e-3


@pytest.mark.parametrize("B,T,Cc", [(2, 3000, 32), (2, 2500, 64), (1, 3001, 128)])
def test_grouped_pair_launch_is_bitwise_the_separate_launches(B, T, Cc):
    """The k = 3 / 7 / 11 branches of one resblock step as ONE grid (dmx_conv_pair_group_raw) give bit for bit what three fused
    launches give: the grouping only changes which workgroup runs where."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(21)
    slope = 0.1
    ks, dil = (3, 7, 11), 3
    xa = F.leaky_relu(torch.randn(B, T, Cc, generator=g), slope).to(_adt()).cuda()
    res = {}
    for mode in ("group", "single"):
        das, dbs, keep = (L.GemmDesc * 3)(), (L.GemmDesc * 3)(), []
        outs = []
        for j, k in enumerate(ks):
            gj = torch.Generator().manual_seed(100 + k)
            w1 = (torch.randn(Cc, Cc, k, generator=gj) / (Cc * k) ** 0.5).to(_adt()).cuda()
            w2 = (torch.randn(Cc, Cc, k, generator=gj) / (Cc * k) ** 0.5).to(_adt()).cuda()
            b1, b2 = torch.randn(Cc, generator=gj).cuda() * 0.1, torch.randn(Cc, generator=gj).cuda() * 0.1
            w1p = w1.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous()
            w2p = w2.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous()
            ha = torch.zeros(B, T, Cc, dtype=_adt(), device="cuda")
            xn = torch.zeros_like(ha)
            raw = torch.zeros_like(ha)
            das[j] = _conv_desc(L, xa, w1p, k, dil, Cc, B, T, C=ha, C2=ha, bias=b1, flags=L.EPI_BIAS | L.EPI_LRELU2 | L.EPI_NO_C, act_slope=slope)
            dbs[j] = _conv_desc(L, ha, w2p, k, 1, Cc, B, T, C=raw, C2=xn, bias=b2, R=xa, resid_inv_slope=1.0 / slope, act_slope=slope,
                                flags=L.EPI_BIAS | L.EPI_RESID | L.EPI_RESID_INV | L.EPI_LRELU2)
            keep += [w1p, w2p, b1, b2]
            outs += [ha, raw, xn]
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if mode == "group":
            L.check(L.lib().dmx_conv_pair_group_raw(3, C.byref(das), C.byref(dbs), C.sizeof(L.GemmDesc), st), "pair group")
        else:
            for j in range(3):
                _pair_run(L, das[j], dbs[j])
        torch.cuda.synchronize()
        res[mode] = outs
    for a, b in zip(res["group"], res["single"]):
        assert torch.equal(a, b)
    assert res["group"][1].float().abs().max().item() > 0.1


@pytest.mark.parametrize("Z,N,Cc", [(2, 520, 64), (1, 1000, 512), (3, 264, 128)])
def test_softmax_backward_fused_into_the_dp_gemm(Z, N, Cc):
    """EPI_SOFTBWD: dS = P * (dO V^T - delta) * scale with delta = rowsum(dO * O), the attention backward of the VAE mid block
    (diffusers Attention, single head) without a materialised dP; against torch fp32 on the same fp16-rounded operands."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(31)
    scale = Cc ** -0.5
    q = torch.randn(Z, N, Cc, generator=g); kk = torch.randn(Z, N, Cc, generator=g)
    v = torch.randn(Z, N, Cc, generator=g).to(_adt()).cuda()
    go = torch.randn(Z, N, Cc, generator=g).to(_adt()).cuda()
    Pm = torch.softmax(q @ kk.transpose(1, 2) * scale, dim=-1).to(_adt()).cuda().contiguous()
    O = (Pm.float() @ v.float())
    delta = (go.float() * O).sum(-1).contiguous()                        # (Z, N) fp32
    dS = torch.empty(Z, N, N, dtype=_adt(), device="cuda")
    d = _desc(L, A=go, W=v, C=dS, X=Pm, rowbias=delta, M=N, N=N, K=Cc, ldw=Cc, Hi=1, Wi=N, Ci=Cc, lda=Cc, Hq=1, Wq=N, ntaps=1, Ho=1, Wo=N,
              ldc=N, ldr=N, ldx=N, ldc2=N, Z=Z, Zi=1, sAo=N * Cc, sWo=N * Cc, sCo=N * N, tdy=[0], tdx=[0], flags=L.EPI_SOFTBWD, alpha=scale)
    _run(L, d)
    dP = go.float() @ v.float().transpose(1, 2)
    ref = Pm.float() * (dP - delta[..., None]) * scale
    assert _rel(dS, ref) < 4e-3


@pytest.mark.parametrize("plan", [212, 313, 414, 611, 815, 318, 202])
def test_forced_split_k_plans_agree(plan):
    """tile_cfg = 100 * slices + tile forces a split-K plan (what the measured table stores): every plan gives the single-pass result"""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    B, H, W, Ci, Co = 8, 16, 16, 384, 384
    x = torch.randn(B, H, W, Ci, generator=g).to(_adt()).cuda()
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).to(_adt()).cuda()
    wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
    bias = torch.randn(Co, generator=g).cuda()
    res = torch.randn(B, H, W, Co, generator=g).to(_adt()).cuda()
    scratch = torch.empty(32 << 20, dtype=torch.float32, device="cuda")
    L.check(L.lib().dmx_gemm_splitk_workspace(C.c_void_p(scratch.data_ptr()), scratch.numel() * 4), "ws")
    outs = []
    for cfg in (6, plan):
        out = torch.zeros(B, H, W, Co, dtype=_adt(), device="cuda")
        d = _desc(L, A=x, W=wp, C=out, bias=bias, R=res, M=B * H * W, N=Co, K=9 * Ci, ldw=9 * Ci, Hi=H, Wi=W, Ci=Ci, lda=Ci,
                  Hq=H, Wq=W, ntaps=9, Ho=H, Wo=W, ldc=Co, ldr=Co, ldx=Co, ldc2=Co, flags=L.EPI_BIAS | L.EPI_RESID,
                  tdy=[t // 3 - 1 for t in range(9)], tdx=[t % 3 - 1 for t in range(9)], tile_cfg=cfg)
        _run(L, d)
        outs.append(out)
    L.check(L.lib().dmx_gemm_splitk_workspace(None, 0), "ws")
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1) + res.float()
    assert _rel(outs[1], ref) < 3e-3
    assert _rel(outs[1], outs[0]) < 2e-3


def _packbits(t):
    """(rows..., C) 16-bit tensor -> sign-bit tensor (rows..., C / 8) uint8, bit e of byte k <=> channel 8k + e > 0."""
    b = (t.float() > 0).to(torch.uint8)
    b = b.reshape(*t.shape[:-1], t.shape[-1] // 8, 8)
    w = (2 ** torch.arange(8, device=t.device, dtype=torch.int32)).to(torch.uint8)
    return (b * w).sum(-1).to(torch.uint8).contiguous()


@pytest.mark.parametrize("B,T,Cc,k,dil", [(2, 1000, 64, 3, 1), (3, 700, 32, 7, 3), (2, 1000, 128, 3, 1), (1, 2049, 128, 11, 5),
                                          (2, 3000, 256, 3, 1), (1, 2500, 512, 7, 3), (2, 1500, 16, 3, 1), (2, 1200, 8, 7, 1)])
def test_sign_bit_tape_forward_and_backward(B, T, Cc, k, dil):
    """HiFi-GAN tape as sign bits: EPI_BITS2 writes (v > 0) of the stored activation, 1 byte per 8 channels (fused pair: also the
    bits-only intermediate), and EPI_MASKBITS in the dgrad epilogues equals EPI_MASK on the 16-bit tensors bit for bit --
    fused pair kernel (C <= 128) and generic tiles (C = 256 / 512) alike."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    slope = 0.1
    xa = F.leaky_relu(torch.randn(B, T, Cc, generator=g), slope).to(_adt()).cuda()
    w1 = (torch.randn(Cc, Cc, k, generator=g) / (Cc * k) ** 0.5).to(_adt()).cuda()
    w2 = (torch.randn(Cc, Cc, k, generator=g) / (Cc * k) ** 0.5).to(_adt()).cuda()
    b1 = torch.randn(Cc, generator=g).cuda() * 0.1
    w1p = w1.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous()
    w2p = w2.permute(0, 2, 1).reshape(Cc, k * Cc).contiguous()
    # ---- forward, reference run: full 16-bit tensors
    ha = torch.zeros(B, T, Cc, dtype=_adt(), device="cuda")
    xn = torch.zeros_like(ha)
    da = _conv_desc(L, xa, w1p, k, dil, Cc, B, T, C=ha, C2=ha, bias=b1, flags=L.EPI_BIAS | L.EPI_LRELU2 | L.EPI_NO_C, act_slope=slope)
    db = _conv_desc(L, ha, w2p, k, 1, Cc, B, T, C=xn, C2=xn, R=xa, resid_inv_slope=1.0 / slope, act_slope=slope,
                    flags=L.EPI_RESID | L.EPI_RESID_INV | L.EPI_LRELU2 | L.EPI_NO_C)
    _run(L, da)
    _run(L, db)
    # ---- forward with the sign-bit outputs (pair kernel when it takes the shape, else two launches)
    hb = torch.full((B, T, Cc // 8), 0xAA, dtype=torch.uint8, device="cuda")
    xb = torch.full((B, T, Cc // 8), 0xAA, dtype=torch.uint8, device="cuda")
    ha2, xn2 = torch.zeros_like(ha), torch.zeros_like(ha)
    fused = Cc in (32, 64, 128)          # (C = 8 / 16: the narrow last HiFi-GAN stages of small test nets, N tail inside a wave tile)
    da2 = _conv_desc(L, xa, w1p, k, dil, Cc, B, T, C=ha2, C2=(None if fused else ha2), B2=hb, ldb2=Cc // 8, bias=b1,
                     flags=L.EPI_BIAS | L.EPI_LRELU2 | L.EPI_NO_C | L.EPI_BITS2, act_slope=slope)
    db2 = _conv_desc(L, ha2, w2p, k, 1, Cc, B, T, C=xn2, C2=xn2, B2=xb, ldb2=Cc // 8, R=xa, resid_inv_slope=1.0 / slope, act_slope=slope,
                     flags=L.EPI_RESID | L.EPI_RESID_INV | L.EPI_LRELU2 | L.EPI_NO_C | L.EPI_BITS2)
    if fused:
        da2.C2 = None
        _pair_run(L, da2, db2)
        assert float(ha2.float().abs().max()) == 0.0                    # bits-only: the intermediate never reached HBM
    else:
        _run(L, da2)
        _run(L, db2)
        assert torch.equal(ha2, ha)
    assert _rel(xn2, xn) < 2e-3
    # the bits are those of the tensor this run stored (fused and unfused differ by rounding of a few near-zero elements only)
    assert torch.equal(xb, _packbits(xn2))
    if not fused:
        assert torch.equal(hb, _packbits(ha))
    else:
        assert float((hb != _packbits(ha)).float().mean()) < 2e-3
    # ---- backward: EPI_MASKBITS == EPI_MASK on the 16-bit tensors
    gc = torch.randn(B, T, Cc, generator=g).to(_adt()).cuda()
    w1b = w1.permute(1, 2, 0).reshape(Cc, k * Cc).contiguous()
    w2b = w2.permute(1, 2, 0).reshape(Cc, k * Cc).contiguous()
    hbits, xbits = _packbits(ha), _packbits(xa)
    res = {}
    for mode in ("full", "bits"):
        ghk = torch.zeros(B, T, Cc, dtype=_adt(), device="cuda")
        dst = torch.zeros_like(ghk)
        if mode == "full":
            ea = _conv_desc(L, gc, w2b, k, 1, Cc, B, T, flip=True, C=ghk, X=ha, flags=L.EPI_MASK, mask_slope=slope)
            eb = _conv_desc(L, ghk, w1b, k, dil, Cc, B, T, flip=True, C=dst, X=xa, R=gc, mask_slope=slope, flags=L.EPI_MASK | L.EPI_RESID)
        else:
            ea = _conv_desc(L, gc, w2b, k, 1, Cc, B, T, flip=True, C=ghk, XB=hbits, ldxb=Cc // 8, flags=L.EPI_MASKBITS, mask_slope=slope)
            eb = _conv_desc(L, ghk, w1b, k, dil, Cc, B, T, flip=True, C=dst, XB=xbits, ldxb=Cc // 8, R=gc, mask_slope=slope,
                            flags=L.EPI_MASKBITS | L.EPI_RESID)
        if fused:
            _pair_run(L, ea, eb)
        else:
            _run(L, ea)
            _run(L, eb)
        res[mode] = dst
    assert torch.equal(res["full"], res["bits"])


def _row_slots(v):
    """(sum, sum of squares) per 32-column slot of every row: the EPI_ROWSTATS layout [row][slot][2]"""
    M, C = v.shape
    v = v.float().reshape(M, C // 32, 32)
    return torch.stack([v.sum(-1), (v * v).sum(-1)], dim=-1).contiguous()


@pytest.mark.parametrize("cfg", [0, 1, 2, 10, 11, 12, 13, 14, 18, 3, 4, 6])
@pytest.mark.parametrize("M,K,N", [(1000, 320, 384), (4032, 384, 1152), (333, 224, 72)])
def test_layernorm_fold_gemm(cfg, M, K, N):
    """EPI_LNFOLD: out = LayerNorm(x) W^T + b computed by ONE GEMM on the raw rows -- gamma folded into the packed weights, beta into the
    bias, the accumulators corrected with the packed row sums and the rows' mean / rstd taken from the 32-column partial sums their
    producer wrote (csrc/gemm_tile.h ln_apply; the U-Net's LN -> QKV / FF1 pairs).  Every instantiated tile, full grids (several
    workgroups per CU), K with a partial last 64-channel group, M / N tails; rows with a large common offset (mean >> std) stress the
    mean * colsum cancellation."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(100 + cfg)
    x = (torch.randn(M, K, generator=g) * (0.2 + torch.rand(M, 1, generator=g)) + 3.0 * torch.randn(M, 1, generator=g)).to(_adt())
    w = torch.randn(N, K, generator=g) / K ** 0.5
    gamma, beta = 0.5 + torch.rand(K, generator=g), 0.3 * torch.randn(K, generator=g)
    bias = 0.1 * torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(_adt())
    wf = (w * gamma).to(_adt())                                   # what pack_layer packs
    colsum = wf.float().sum(1).contiguous()
    bf = (bias + w @ beta).contiguous()
    out = torch.empty(M, N, dtype=_adt(), device="cuda")
    xd, wd, rd, cd, bd, sd = x.cuda(), wf.cuda(), res.cuda(), colsum.cuda(), bf.cuda(), _row_slots(x).cuda()
    d = _desc(L, A=xd, W=wd, C=out, bias=bd, R=rd, colsum=cd, ln_eps=1e-5, rowstats_in=sd, nslots=K // 32, M=M, N=N, K=K, ldw=K, Hi=1, Wi=M,
              Ci=K, lda=K, Hq=1, Wq=M, ntaps=1, Ho=1, Wo=M, ldc=N, ldr=N, ldx=N, ldc2=N, flags=L.EPI_BIAS | L.EPI_RESID | L.EPI_LNFOLD,
              tdy=[0], tdx=[0], tile_cfg=cfg)
    _run(L, d)
    ref = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ w.t() + bias + res.float()
    err = _rel(out.cpu(), ref)
    assert err < 4e-3, err
    # a split-K plan cannot carry the fold: the forced plan is refused, not silently mis-normalised
    d.tile_cfg = 212
    assert L.lib().dmx_gemm_raw(C.byref(d), C.sizeof(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)) != 0


@pytest.mark.parametrize("cfg", [0, 1, 2, 10, 11, 12, 13, 14, 18, 3, 4, 6])
@pytest.mark.parametrize("M,K,N", [(1000, 320, 384), (4032, 384, 640), (333, 200, 96)])
def test_rowstats_epilogue(cfg, M, K, N):
    """EPI_ROWSTATS: the producer of a LayerNorm input writes, next to its 16-bit output, (sum v, sum v^2) of every row per 32-column slot
    from the final fp32 values (bias and residual included) -- csrc/gemm_epilogue.h.  Checked against the sums of the fp32 reference rows
    and, chained, as the statistics input of an EPI_LNFOLD consumer."""
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(300 + cfg)
    x = torch.randn(M, K, generator=g).to(_adt())
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(_adt())
    bias = torch.randn(N, generator=g)
    res = (2.0 * torch.randn(M, N, generator=g) + torch.randn(M, 1, generator=g)).to(_adt())
    out = torch.empty(M, N, dtype=_adt(), device="cuda")
    st = torch.full((M, N // 32, 2), float("nan"), device="cuda")
    xd, wd, rd, bd = x.cuda(), w.cuda(), res.cuda(), bias.cuda()
    d = _desc(L, A=xd, W=wd, C=out, bias=bd, R=rd, rowstats_out=st, nslots=N // 32, M=M, N=N, K=K, ldw=K, Hi=1, Wi=M, Ci=K, lda=K, Hq=1, Wq=M,
              ntaps=1, Ho=1, Wo=M, ldc=N, ldr=N, ldx=N, ldc2=N, flags=L.EPI_BIAS | L.EPI_RESID | L.EPI_ROWSTATS, tdy=[0], tdx=[0], tile_cfg=cfg)
    _run(L, d)
    ref = x.float() @ w.float().t() + bias + res.float()
    assert _rel(out.cpu(), ref) < 3e-3
    want = _row_slots(ref)
    got = st.cpu()
    assert torch.isfinite(got).all()
    assert _rel(got[..., 0], want[..., 0]) < 2e-3 and _rel(got[..., 1], want[..., 1]) < 2e-3
    # chained: LayerNorm over the N columns folded into a second projection that reads the statistics just written
    N2 = 64
    w2 = torch.randn(N2, N, generator=g) / N ** 0.5
    gamma, beta = 0.5 + torch.rand(N, generator=g), 0.3 * torch.randn(N, generator=g)
    wf = (w2 * gamma).to(_adt())
    out2 = torch.empty(M, N2, dtype=_adt(), device="cuda")
    cd, bd2, wfd = wf.float().sum(1).contiguous().cuda(), (w2 @ beta).contiguous().cuda(), wf.cuda()
    d2 = _desc(L, A=out, W=wfd, C=out2, bias=bd2, colsum=cd, ln_eps=1e-5, rowstats_in=st, nslots=N // 32, M=M, N=N2, K=N, ldw=N, Hi=1, Wi=M,
               Ci=N, lda=N, Hq=1, Wq=M, ntaps=1, Ho=1, Wo=M, ldc=N2, ldr=N2, ldx=N2, ldc2=N2, flags=L.EPI_BIAS | L.EPI_LNFOLD, tdy=[0], tdx=[0])
    _run(L, d2)
    ref2 = F.layer_norm(out.cpu().float(), (N,), gamma, beta, 1e-5) @ w2.t()
    assert _rel(out2.cpu(), ref2) < 5e-3
