"""-m gpu: the torch.ops.diffmusic_hip.* layer (csrc_torch/torch_ops.cpp) against the ctypes binding of the same C-ABI entry
points: same launchers, same stream => bit-identical results, for the scheduler ops, the measurement-path ops and the three
networks driven through their handles."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def test_scheduler_ops_equal_ctypes_path():
    from diffmusic_amd import _lib as L, ops
    lib, h = L.lib(), ops.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    B, shp = 3, (3, 8, 25, 16)
    x, e, g0, z = (torch.randn(shp, generator=g).cuda() for _ in range(4))
    inv = torch.rand(B, generator=g).cuda() + 0.5
    x0 = torch.empty_like(x)
    L.check(lib.dmx_sched_pred_x0(_p(x), _p(e), _p(x0), x.numel(), 0.37, st), "x0")
    assert torch.equal(h.sched_pred_x0(x, e, 0.37), x0)
    e2 = torch.cat([e, g0])
    out = torch.empty_like(x)
    L.check(lib.dmx_sched_cfg_combine(_p(e2), _p(out), out.numel(), 2.0, st), "cfg")
    assert torch.equal(h.cfg_combine(e2, 2.0), out)
    for mode, noise, gn in ((0, None, False), (1, None, False), (1, z, False), (2, None, False), (3, z, False), (4, z, False), (3, z, True)):
        prev = torch.empty_like(x)
        x0o = torch.empty_like(x) if mode == 2 else None
        gg = g0 if mode else None
        L.check(lib.dmx_sched_step(mode, _p(x), _p(e), _p(x0), _p(gg), _p(inv if mode else None), _p(noise), _p(prev), _p(x0o), None, B,
                                   x[0].numel(), 0.37, 0.41, 0.2, 0.08, 1e-8, int(gn), st), "step")
        p2, x02 = h.sched_update(mode, x, e, x0, gg, inv if mode else None, noise, 0.37, 0.41, 0.2, 0.08, 1e-8, gn)
        assert torch.equal(p2, prev), mode
        if mode == 2:
            assert torch.equal(x02, x0o)
    from diffmusic_amd.torch_utils import randn_philox
    assert torch.equal(h.randn_philox(list(shp), [1, 2, 3], 5, torch.device("cuda")), randn_philox(shp, [1, 2, 3], 5, "cuda"))


def test_measurement_ops_equal_facade(monkeypatch):
    from diffmusic_amd import ops, inverse_problem as P
    from diffmusic_amd.inverse_problem.operator import l2_loss
    h = ops.load()
    monkeypatch.setattr(ops, "USE_TORCH_OPS", False)          # the facade legs below: ctypes binding (their default is the op layer too)
    assert not ops.enabled()
    g = torch.Generator().manual_seed(1)
    L_ = 16000
    wav = (0.3 * torch.randn(2, L_ + 32, generator=g)).cuda()
    op = P.IdentityOperator(16000)
    fe = op.frontend
    mel = fe.transform_fwd(wav, L_, True, True, -80.0, 80.0).clone()
    state = torch.empty_like(fe._state)
    mel2 = h.logmel_fwd(fe._h.value, wav, state, L_, True, True, -80.0, 80.0)
    assert torch.equal(mel, mel2)
    d = torch.randn(mel.shape, generator=g).cuda()
    assert torch.equal(fe.transform_bwd(d.contiguous()), h.logmel_bwd(fe._h.value, d.contiguous(), state, L_, True, True, -80.0, 80.0))
    pr = P.PhaseRetrievalOperator(noiser=None)
    mag = pr.forward(wav[:, :L_].contiguous())
    st2 = torch.empty_like(pr.frontend._state)
    mag2 = h.stft_mag_fwd(pr.frontend._h.value, wav, st2, L_)
    assert torch.equal(mag, mag2) and mag2.shape == (2, 513, 101)
    assert torch.equal(pr.frontend.melscale(mag, -80.0, 80.0), h.melscale_fwd(pr.frontend._h.value, mag, -80.0, 80.0))
    loss, dp = l2_loss(mel, mel2 * 0.9)
    loss2, dp2 = h.l2norm(mel, (mel2 * 0.9).contiguous(), 1.0)
    assert torch.equal(loss, loss2) and torch.equal(dp, dp2)
    sr = P.SuperResolutionOperator(16000, 4, noiser=None)
    y = sr._a_fwd(wav, L_)
    y2 = h.resample_fwd(wav, sr._k(wav.device), L_, y.shape[1], sr.orig, sr.new, sr.width)
    assert torch.equal(y, y2)
    assert torch.equal(sr._a_bwd(y.contiguous(), wav.shape[1]), h.resample_bwd(y.contiguous(), sr._k(wav.device), None, L_, wav.shape[1], sr.orig, sr.new, sr.width))
    inp = P.MusicInpaintingOperator(1, L_, "box", 0.25, 0.5, 0.3, 0.1, 0.2, noiser=None)
    assert torch.equal(inp._a_fwd(wav, L_), h.mask_mul(wav, inp._mask_on(wav.device), L_, L_))


def test_network_ops_through_handles_equal_engines(monkeypatch):
    """The engines' default binding is torch.ops.diffmusic_hip.* (ops.enabled()); DMX_TORCH_OPS=0 is the ctypes binding of the same
    launchers: bit-identical, for the three networks (MusicLDM and AudioLDM2 U-Nets) and the in-place gradient rescale."""
    from diffmusic_amd import _lib as L, ops
    from diffmusic_amd.engine import HifiGanEngine, VaeDecoderEngine, UNetEngine
    from tests.test_gpu_step import HIFI, VAE
    from tests.test_gpu_unet import SMALL
    h = ops.load()
    assert ops.enabled()                                  # the product default goes through the custom ops
    g = torch.Generator().manual_seed(2)
    voc = HifiGanEngine(HIFI); voc.load_state_dict(voc.synth_state_dict(1))
    vae = VaeDecoderEngine(VAE); vae.load_state_dict(vae.synth_state_dict(2))
    un = UNetEngine(SMALL); un.load_state_dict(un.synth_state_dict(9))
    a2 = UNetEngine(dict(SMALL, class_embed_dim=0, attn_cross_dims=[0, 48, 64])); a2.load_state_dict(a2.synth_state_dict(3))
    mel = torch.randn(2, 40, 64, generator=g).to(L.act_dtype()).cuda()
    z = torch.randn(2, 8, 10, 16, generator=g).cuda()
    x = torch.randn(2, 8, 26, 16, generator=g).cuda()
    t = torch.full((2,), 501.0).cuda()
    cls = torch.randn(2, 512, generator=g).cuda()
    c0, c1 = torch.randn(2, 8, 48, generator=g).cuda(), torch.randn(2, 12, 64, generator=g).cuda()
    m1 = torch.ones(2, 12).cuda(); m1[1, 9:] = 0

    def run():
        wav = voc.forward(mel).clone()
        d = torch.randn(wav.shape, generator=torch.Generator().manual_seed(5)).cuda()
        dmel = voc.backward(d).clone()
        m16, m32 = vae.decode_hip(z, 1.1, keep_state=True, want_f32=True)
        m16, m32 = m16.clone(), m32.clone()
        dm = torch.randn(m16.shape, generator=torch.Generator().manual_seed(6)).to(L.act_dtype()).cuda()
        dz = vae.backward(dm, 1.1).clone()
        eps = un.forward(x, t, cls).clone()
        eps2 = a2.forward(x, t, None, c0, c1, m1).clone()
        return wav, dmel, m16, m32, dz, eps, eps2

    via_ops = run()
    monkeypatch.setattr(ops, "USE_TORCH_OPS", False)
    assert not ops.enabled()
    via_ctypes = run()
    monkeypatch.setattr(ops, "USE_TORCH_OPS", True)
    for a, b in zip(via_ops, via_ctypes):
        assert a.dtype == b.dtype and torch.equal(a, b)
    # direct calls through the handles
    ws = next(iter(voc._ws.values()))
    assert torch.equal(h.hifigan_fwd(voc._h.value, mel, ws), via_ctypes[0])
    wsu = next(iter(un._ws.values()))
    assert torch.equal(h.unet_fwd(un._h.value, x, t, cls, wsu), via_ctypes[5])
    # in-place gradient rescale
    dw = (torch.randn(3, 4000, generator=g) * torch.tensor([1e-3, 1.0, 50.0])[:, None]).cuda()
    dw2 = dw.clone()
    inv = h.grad_normalize_(dw, 64.0)
    inv2 = torch.empty(3, device="cuda")
    L.check(L.lib().dmx_grad_normalize(_p(dw2), _p(inv2), 3, 4000, 64.0, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gn")
    assert torch.equal(dw, dw2) and torch.equal(inv, inv2)
    assert torch.allclose(dw.abs().amax(dim=1), torch.full((3,), 64.0, device="cuda"), rtol=1e-5)


def test_style_operator_and_operator_helpers_equal_between_bindings(monkeypatch):
    """The style-guidance pair (HIP resampler, CLAP log-mel, HTS-AT tower forward / backward, Gram, L2) and the super-resolution operator's
    guidance through torch.ops.diffmusic_hip.* (default) and through ctypes: bit-identical loss and waveform gradient."""
    import bench
    from diffmusic_amd import ops, inverse_problem as P
    assert ops.enabled()
    L_ = 32000
    y = torch.stack([bench.synth_clip(1, L_), bench.synth_clip(2, L_)]).cuda()
    wav = torch.cat([0.5 * y + 0.05 * torch.randn(2, L_, generator=torch.Generator().manual_seed(0)).cuda(), torch.zeros(2, 32, device="cuda")], dim=1).contiguous()
    style = P.StyleGuidanceOperator(16000, noiser=None, device="cuda", seed=3)
    sr = P.SuperResolutionOperator(16000, 4, noiser=None)
    res = {}
    for tag, on in (("ops", True), ("ctypes", False)):
        monkeypatch.setattr(ops, "USE_TORCH_OPS", on)
        assert ops.enabled() == on
        out = []
        for op in (style, sr):
            op.reset_cache()
            meas = op.forward(y)
            loss, dwav = op.guidance(wav, L_, meas, "mel_spectrogram")
            out += [loss.clone(), dwav.clone()]
        res[tag] = out
    monkeypatch.setattr(ops, "USE_TORCH_OPS", True)
    for a, b in zip(res["ops"], res["ctypes"]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    assert float(res["ops"][1].abs().max()) > 0.0


def test_ops_fall_back_to_ctypes_with_a_warning_when_the_op_library_is_missing(monkeypatch, tmp_path):
    from diffmusic_amd import ops
    monkeypatch.setattr(ops, "_loaded", False)
    monkeypatch.setattr(ops, "_usable", None)
    monkeypatch.setattr(ops, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.warns(RuntimeWarning, match="ctypes binding"):
        assert not ops.enabled()
    assert not ops.enabled()                              # warned once, stays on ctypes
