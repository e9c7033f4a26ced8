"""GPU leg of the data + evaluation adjacency: the loader's device path (HIP FIR resampling) against the host
path, and the device legs of LSD (HIP STFT kernel) / MSE against the numpy definitions the CPU tests pin."""
import numpy as np
import pytest
import torch

from diffmusic_amd.data import WAVDataset
from diffmusic_amd.metrics import LogSpectralDistance, MeanSquaredError
from tests.test_dataloader import _write

pytestmark = pytest.mark.gpu


def test_wav_dataset_device_leg_matches_host(tmp_path):
    sr = 16000
    t44 = np.arange(3 * 44100) / 44100
    rng = np.random.default_rng(0)
    _write(str(tmp_path / "a_44k_stereo.wav"), np.stack([0.4 * np.sin(2 * np.pi * 440 * t44), 0.1 * rng.standard_normal(t44.size)]), 44100)
    _write(str(tmp_path / "b_16k.wav"), 0.3 * np.sin(2 * np.pi * 300 * np.arange(3 * sr) / sr), sr)
    host = WAVDataset(str(tmp_path), sr, 1, start_s=0.25, end_s=2.5)
    dev = WAVDataset(str(tmp_path), sr, 1, start_s=0.25, end_s=2.5, device="cuda")
    for i in range(2):
        (wh, nh), (wd, nd) = host[i], dev[i]
        assert nh == nd and wd.is_cuda and wd.shape == wh.shape == (int(2.25 * sr),)
        assert float((wd.cpu() - wh).abs().max()) < 2e-6, i       # fp32 polyphase FIR on both sides, different summation order


def test_lsd_and_mse_device_legs_match_numpy():
    g = torch.Generator().manual_seed(3)
    B, L = 3, 16000 * 2 + 37
    ref = 0.3 * torch.randn(B, L, generator=g)
    est = ref + 0.05 * torch.randn(B, L, generator=g)
    est[1, 100] = float("nan"); est[2, 5] = float("inf")
    m = LogSpectralDistance()
    want = m.score(ref.numpy(), est.numpy(), output_mean=False)
    got = m.score(ref.cuda(), est.cuda(), output_mean=False)
    assert got.is_cuda and got.shape == (B,)
    assert np.allclose(got.cpu().numpy(), want, rtol=1e-8), (got, want)         # float64 on both sides
    assert np.isclose(float(m.score(ref.cuda(), est.cuda())), want.mean(), rtol=1e-8)
    assert float(m.score(ref.cuda(), ref.cuda())) == 0.0
    fast = LogSpectralDistance(device_precision="fp32")                          # HIP STFT kernel: fp32 spectrum vs float64
    assert np.allclose(fast.score(ref.cuda(), est.cuda(), output_mean=False).cpu().numpy(), want, rtol=2e-4)
    # a clip with a silent gap (masked inpainting region) and a band-limited reference: bins at or below 1e-9, where the definition's
    # log10(|X| + 1e-10) is decided by the floor -- the float64 device leg still equals the host leg, the fp32 kernel leg does not
    n = torch.arange(L, dtype=torch.float64)
    tone = (0.4 * torch.sin(2 * np.pi * 440.0 * n / 16000) + 0.2 * torch.sin(2 * np.pi * 1250.0 * n / 16000)).float()[None].repeat(B, 1)
    gap = tone.clone()
    gap[:, 8000:20000] = 0.0
    est2 = gap + 1e-4 * torch.randn(B, L, generator=g)
    want2 = m.score(gap.numpy(), est2.numpy(), output_mean=False)
    got2 = m.score(gap.cuda(), est2.cuda(), output_mean=False).cpu().numpy()
    assert np.allclose(got2, want2, rtol=1e-6), (got2, want2)
    off = fast.score(gap.cuda(), est2.cuda(), output_mean=False).cpu().numpy()
    print("LSD with a silent gap: float64 legs", want2, "fp32 kernel leg", off)
    assert not np.allclose(off, want2, rtol=1e-3)                                # the documented limitation of device_precision="fp32"
    short = LogSpectralDistance(n_fft=1024, hop_length=160)
    x = torch.randn(2, 1000, generator=g)                                        # clip shorter than one window
    y = torch.randn(2, 1000, generator=g)
    assert np.allclose(short.score(x.cuda(), y.cuda(), output_mean=False).cpu().numpy(), short.score(x.numpy(), y.numpy(), output_mean=False),
                       rtol=2e-4)
    for red in ("mean", "sum"):
        e = MeanSquaredError(red)
        want = e.score(list(ref.numpy()), list(est.numpy()[:, :L - 5]))
        got = e.score(ref.cuda(), est.cuda()[:, :L - 5])
        assert got.is_cuda and np.isclose(float(got), want, rtol=1e-5)
