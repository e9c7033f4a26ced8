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


def test_measurement_ops_equal_facade():
    from diffmusic_amd import ops, inverse_problem as P
    from diffmusic_amd.inverse_problem.operator import l2_loss
    h = ops.load()
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


def test_network_ops_through_handles_equal_engines():
    from diffmusic_amd import _lib as L, ops
    from diffmusic_amd.engine import HifiGanEngine, VaeDecoderEngine, UNetEngine
    from tests.test_gpu_step import HIFI, VAE
    from tests.test_gpu_unet import SMALL
    h = ops.load()
    g = torch.Generator().manual_seed(2)
    voc = HifiGanEngine(HIFI); voc.load_state_dict(voc.synth_state_dict(1))
    mel = torch.randn(2, 40, 64, generator=g).to(L.act_dtype()).cuda()
    wav = voc.forward(mel).clone()
    ws = next(iter(voc._ws.values()))
    assert torch.equal(h.hifigan_fwd(voc._h.value, mel, ws), wav)
    d = torch.randn(wav.shape, generator=g).cuda()
    assert torch.equal(h.hifigan_bwd(voc._h.value, d, 40, 64), voc.backward(d))
    vae = VaeDecoderEngine(VAE); vae.load_state_dict(vae.synth_state_dict(2))
    z = torch.randn(2, 8, 10, 16, generator=g).cuda()
    m = vae.decode_hip(z, 1.1, keep_state=True).clone()
    wsv = next(iter(vae._ws.values()))
    assert torch.equal(h.vae_dec_fwd(vae._h.value, z, 1.1, True, wsv), m)
    dm = torch.randn(m.shape, generator=g).to(L.act_dtype()).cuda()
    assert torch.equal(h.vae_dec_bwd(vae._h.value, dm, 1.1, 8), vae.backward(dm, 1.1))
    un = UNetEngine(SMALL); un.load_state_dict(un.synth_state_dict(9))
    x = torch.randn(2, 8, 26, 16, generator=g).cuda()
    t = torch.full((2,), 501.0).cuda()
    cls = torch.randn(2, 512, generator=g).cuda()
    eps = un.forward(x, t, cls).clone()
    wsu = next(iter(un._ws.values()))
    assert torch.equal(h.unet_fwd(un._h.value, x, t, cls, wsu), eps)
