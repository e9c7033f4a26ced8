"""CPU: independent cross-checks of the third-party arithmetic the oracle restates (parity with
diffusers/torchaudio themselves is unpinned -- they are not installed; see oracle/__init__.py)."""
import math
import numpy as np
import torch

from oracle import audio, models
from oracle.ddim import DDIMParent


def test_mel_filterbank_matches_transformers_and_product_table():
    from transformers.audio_utils import mel_filter_bank
    fb = audio.melscale_fbanks(513, 0.0, 8000.0, 64, 16000).numpy()
    ref = mel_filter_bank(513, 64, 0.0, 8000.0, 16000, norm=None, mel_scale="htk")
    assert fb.shape == ref.shape == (513, 64)
    assert np.abs(fb - ref).max() < 1e-5
    from diffmusic_amd.inverse_problem import dsp
    assert np.abs(dsp.melscale_fbanks(513, 0.0, 8000.0, 64, 16000) - fb).max() < 2e-5   # numpy vs torch fp32 rounding


def test_power_spectrogram_is_a_hann_dft():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 2000, generator=g)
    p = audio.power_spectrogram(x)                       # (1, 513, 13)
    assert p.shape == (1, 513, 1 + 2000 // 160)
    xp = torch.nn.functional.pad(x[None], (512, 512), mode="reflect")[0, 0].double()
    n = torch.arange(1024, dtype=torch.float64)
    w = 0.5 - 0.5 * torch.cos(2 * math.pi * n / 1024)
    for f, k in ((0, 0), (3, 17), (12, 512), (7, 300)):
        seg = xp[f * 160: f * 160 + 1024] * w
        X = (seg * torch.exp(-2j * math.pi * k * n / 1024)).sum()
        assert abs(p[0, k, f].item() - abs(X) ** 2) < 1e-3 * max(1.0, abs(X) ** 2)


def test_log_mel_of_tone_peaks_at_analytic_band():
    t = torch.arange(16000) / 16000.0
    mel = audio.Wav2Mel(16000)(torch.sin(2 * math.pi * 1000.0 * t)[None])
    band = int(mel[0, :, 50].argmax())
    fb = audio.melscale_fbanks()
    assert band == int(fb[round(1000 / (8000 / 512))].argmax())


def test_resample_dc_gain_lengths_and_sine():
    for scale, taps in ((2, 28), (4, 54)):
        kern, width, orig, new = audio.sinc_resample_kernel(16000, 16000 // scale)
        assert kern.shape == (1, 1, taps) and orig == scale and new == 1
        y = audio.resample(torch.ones(1, 16000), 16000, 16000 // scale)
        assert y.shape == (1, 16000 // scale) and abs(y[0, 2000].item() - 1.0) < 2e-3
        t = torch.arange(16000) / 16000.0
        ys = audio.resample(torch.sin(2 * math.pi * 440 * t)[None], 16000, 16000 // scale)
        tt = torch.arange(16000 // scale) / (16000.0 / scale)
        assert (ys[0, 200:-200] - torch.sin(2 * math.pi * 440 * tt)[200:-200]).abs().max() < 5e-3
    from diffmusic_amd.inverse_problem import dsp
    k2 = dsp.sinc_resample_kernel(16000, 4000)[0]
    assert np.abs(k2 - audio.sinc_resample_kernel(16000, 4000)[0].numpy().reshape(1, -1)).max() < 1e-7


def test_hifigan_matches_transformers():
    from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig
    kw = dict(model_in_dim=64, upsample_initial_channel=64, upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
              resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3)
    ref = SpeechT5HifiGan(SpeechT5HifiGanConfig(sampling_rate=16000, normalize_before=False, **kw)).eval()
    mine = models.kaiming_init_(models.HifiGan(**kw)).eval()
    ref.load_state_dict(mine.state_dict(), strict=True)
    x = torch.randn(2, 20, 64)
    with torch.no_grad():
        assert (ref(x) - mine(x)).abs().max() < 1e-6
    assert mine(x).shape == (2, 3232)
    assert models.HifiGan().eval()(torch.zeros(1, 4, 64)).shape[1] == 4 * 160 + 32      # 160*T + 32


def test_ddim_tables_and_known_answers():
    s = DDIMParent(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear",
                   clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    assert abs(s.alphas_cumprod[0].item() - 0.99850) < 1e-5 and abs(s.alphas_cumprod[996].item() - 1.5095e-4) < 1e-7
    s.set_timesteps(200)
    assert s.timesteps[0] == 996 and s.timesteps[1] == 991 and s.timesteps[-1] == 1 and len(s.timesteps) == 200
    s.set_timesteps(50)
    assert s.timesteps[0] == 981
    s.set_timesteps(500)
    assert s.timesteps[0] == 999
    # eta = 0: eps' == eps, so the DDIM update is the closed form
    s.set_timesteps(200)
    x, e = torch.randn(1, 8, 5, 4), torch.randn(1, 8, 5, 4)
    prev, x0 = s.parent_step(e, 501, x, eta=0.0)
    a_t, a_p = s.alphas_cumprod[501], s.alphas_cumprod[496]
    assert torch.allclose(prev, a_p.sqrt() * x0 + (1 - a_p).sqrt() * e, atol=1e-6)
    assert torch.allclose(x0, (x - (1 - a_t).sqrt() * e) / a_t.sqrt(), atol=1e-6)


def test_unet_and_vae_shapes_and_param_counts():
    assert abs(sum(p.numel() for p in models.HifiGan().parameters()) / 1e6 - 55.26) < 0.01
    vae = models.VaeDecoder(block_out_channels=(32, 64, 64))
    assert vae.decode(torch.randn(1, 8, 10, 4)).sample.shape == (1, 1, 40, 16)
    un = models.UNetMusicLDM(block_out_channels=(32, 64, 96, 160), attention_heads=4)
    assert un(torch.randn(2, 8, 26, 16), 981, class_labels=torch.randn(2, 512))[0].shape == (2, 8, 26, 16)
