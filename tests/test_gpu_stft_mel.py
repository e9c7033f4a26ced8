"""-m gpu: the fused STFT -> mel -> dB -> L2 kernels and their backward (csrc/stft_mel.hip; reference: torchaudio MelSpectrogram +
AmplitudeToDB / MelScale on |torch.stft| in diffmusic/inverse_problem/operator.py:23-33,143-147,162-170, the loss and
torch.autograd.grad in diffmusic/schedulers/scheduling_dps.py:202-212).

Truth = the same chain written with torch.stft + autograd in FLOAT64 (torch.stft is the reference's own operator; SURVEY.md
section 8c lists it as a direct oracle).  Covered: the three transform variants of the operators (power / dB without clamp,
power / dB clamped, magnitude / linear clamped with the rectangular window), the inpainting mask fused on load and on store,
a shared (batch-1) reference, hop 480 with CLAP's slaney bank and an explicit d(loss)/d(mel), clip lengths that are not a multiple
of the hop, the reflect-padded edges, the zero tail past the clip, and bitwise reproducibility."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _truth(wav, mask, fb, ref, L, hop, hann, power2, to_db, lo, hi, dmel=None):
    """float64 torch: (mel (B, T, 64), loss (B), dwav (B, full)); ref None + dmel given: the VJP of mel with cotangent dmel."""
    w = wav.double().clone().requires_grad_(True)
    y = w[:, :L] * (mask.double() if mask is not None else 1.0)
    win = torch.hann_window(1024, periodic=True, dtype=torch.float64, device=wav.device) if hann else torch.ones(1024, dtype=torch.float64, device=wav.device)
    spec = torch.stft(y, 1024, hop, 1024, window=win, center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    p = spec.real ** 2 + spec.imag ** 2
    if not power2:
        p = torch.sqrt(p)
    mel_lin = torch.einsum("bkt,km->btm", p, fb.double())
    mel = 10.0 * torch.log10(torch.clamp(mel_lin, min=1e-10)) if to_db else mel_lin
    mel = torch.clamp(mel, lo, hi)
    if dmel is not None:
        (g,) = torch.autograd.grad((mel * dmel.double()).sum(), w)
        return mel.detach(), None, g
    loss = torch.linalg.vector_norm((ref.double() - mel).flatten(1), dim=1)
    (g,) = torch.autograd.grad(loss.sum(), w)
    return mel.detach(), loss.detach(), g


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def _clips(B, full, seed):
    import bench
    g = torch.Generator().manual_seed(seed)
    x = torch.stack([bench.synth_clip(10 * seed + i, full) for i in range(B)])        # other clips per seed: the loss must not vanish
    x[-1] = 0.2 * torch.randn(full, generator=g)
    return x.cuda().contiguous()


CASES = [  # name, L, full, hop, hann, power2, to_db, lo, hi, masked, shared_ref
    ("inpainting_db_noclamp", 160000, 160032, 160, True, True, True, -3.0e38, 3.0e38, True, False),
    ("identity_db_clamped", 48000, 48000, 160, True, True, True, -80.0, 80.0, False, True),
    ("phase_mag_rect", 32000, 32032, 160, False, False, False, -80.0, 80.0, False, False),
    ("ragged_length", 20037, 20100, 160, True, True, True, -80.0, 80.0, True, False),
    ("short_two_chunks", 2600, 2600, 160, True, True, True, -3.0e38, 3.0e38, False, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fused_guidance_matches_float64_torch(case):
    from diffmusic_amd.inverse_problem.operator import SpectralFrontend
    from diffmusic_amd.inverse_problem import dsp
    name, L, full, hop, hann, power2, to_db, lo, hi, masked, shared = case
    B = 3
    fe = SpectralFrontend(16000, 1024, hop, 64, "hann" if hann else "rect")
    assert fe.fused(L)
    fb = torch.from_numpy(dsp.melscale_fbanks(513, 0.0, 8000.0, 64, 16000)).cuda()
    wav = _clips(B, full, 1)
    mask = None
    if masked:
        mask = torch.ones(L)
        mask[L // 5: L // 5 + L // 10] = 0.0
        mask[:300] = 0.0                                   # a masked stretch inside the left reflection zone
        mask = mask.cuda()
    target = _clips(1 if shared else B, L, 2)
    ref = fe.transform_fwd(target, L, power2, to_db, lo, hi).clone()            # (B or 1, T, 64) through the fused forward itself
    mel_t, loss_t, g_t = _truth(wav, mask, fb, ref, L, hop, hann, power2, to_db, max(lo, -1e300), min(hi, 1e300))
    # forward alone (transform): mel vs float64
    y = wav[:, :L] * mask if mask is not None else wav[:, :L]
    mel = fe.transform_fwd(y.contiguous(), L, power2, to_db, lo, hi)
    strong = mel_t > (mel_t.amax(dim=(1, 2), keepdim=True) - 40.0) if to_db else mel_t > 1e-3 * mel_t.amax(dim=(1, 2), keepdim=True)
    err = (mel.double() - mel_t).abs()
    tol = 1e-4 if to_db else 2e-5 * float(mel_t.abs().max())
    assert float(err[strong].max()) <= tol, (name, float(err[strong].max()))
    # fused guidance: loss and gradient
    loss, dwav = fe.guidance(wav, L, ref, mask, power2, to_db, lo, hi)
    assert loss.shape == (B,) and dwav.shape == (B, full)
    assert float(((loss.double() - loss_t).abs() / loss_t).max()) < 2e-5, (name, loss, loss_t)
    assert _rel(dwav[:, :L], g_t[:, :L]) < 2e-4, (name, _rel(dwav[:, :L], g_t[:, :L]))
    # the edges on their own (reflect padding folds the gradient of the padded samples back): first and last 600 samples
    assert _rel(dwav[:, :600], g_t[:, :600]) < 5e-4 and _rel(dwav[:, L - 600:L], g_t[:, L - 600:L]) < 5e-4
    if full > L:
        assert float(dwav[:, L:].abs().max()) == 0.0                            # no gradient past the clip
    if mask is not None:
        assert float(dwav[:, :L][:, mask == 0].abs().max()) == 0.0
    # bit-reproducible (no atomics: per-wave accumulators summed in a fixed order)
    loss2, dwav2 = fe.guidance(wav, L, ref, mask, power2, to_db, lo, hi)
    assert torch.equal(loss, loss2) and torch.equal(dwav, dwav2)
    # gscale scales the gradient only
    loss3, dwav3 = fe.guidance(wav, L, ref, mask, power2, to_db, lo, hi, gscale=0.25)
    assert torch.equal(loss3, loss) and _rel(dwav3, 0.25 * dwav) < 1e-6


def test_fused_transform_vjp_with_clap_frontend():
    """hop 480 / slaney bank (the CLAP log-mel of the style-guidance operator) with an explicit cotangent: transform_fwd keeps the
    waveform in its state, transform_bwd differentiates through it."""
    from transformers.audio_utils import mel_filter_bank
    from diffmusic_amd.inverse_problem.operator import SpectralFrontend
    fbn = mel_filter_bank(num_frequency_bins=513, num_mel_filters=64, min_frequency=0.0, max_frequency=14000.0, sampling_rate=48000,
                          norm="slaney", mel_scale="slaney")
    fe = SpectralFrontend(48000, 1024, 480, 64, "hann", fb=fbn)
    L = 96000 + 211
    assert fe.fused(L)
    wav = _clips(2, L, 3)
    mel = fe.transform_fwd(wav, L, True, True).clone()
    assert mel.shape == (2, 1 + L // 480, 64)
    g = torch.Generator().manual_seed(4)
    d = torch.randn(mel.shape, generator=g).cuda()
    dw = fe.transform_bwd(d.contiguous())
    fb = torch.from_numpy(np.asarray(fbn, dtype=np.float32)).cuda()
    mel_t, _, g_t = _truth(wav, None, fb, None, L, 480, True, True, True, -1e300, 1e300, dmel=d)
    strong = mel_t > mel_t.amax(dim=(1, 2), keepdim=True) - 40.0
    assert float((mel.double() - mel_t).abs()[strong].max()) <= 1e-4
    assert _rel(dw, g_t) < 2e-4, _rel(dw, g_t)
    # linear in the cotangent, and the second backward call does not disturb the kept waveform
    dw2 = fe.transform_bwd((2.0 * d).contiguous())
    assert _rel(dw2, 2.0 * dw) < 1e-6


def test_operators_take_the_fused_route_and_match_their_composed_path():
    """MusicInpaintingOperator / IdentityOperator / SuperResolutionOperator / PhaseRetrievalOperator.guidance through the fused
    kernels equals the composed route (transform_fwd + L2 + transform_bwd around A / A^T) the operators took before."""
    from diffmusic_amd import inverse_problem as P
    from diffmusic_amd.inverse_problem.operator import l2_loss
    L, full = 32000, 32032
    wav = _clips(2, full, 5)
    clean = _clips(2, L, 6)
    n = P.get_noiser("gaussian", 0.0)
    for op in (P.MusicInpaintingOperator(2, 16000, "box", 0.5, 0.9, 0.3, 0.1, 0.2, noiser=n), P.IdentityOperator(16000),
               P.SuperResolutionOperator(16000, 2, noiser=n), P.PhaseRetrievalOperator(noiser=n)):
        meas = op.forward(clean)
        loss, dwav = op.guidance(wav, L, meas, "mel_spectrogram")
        # composed: explicit A, transform, loss, transposes
        fe = op.frontend
        if isinstance(op, P.PhaseRetrievalOperator):
            ref = fe.melscale(meas, -80.0, 80.0)
            pred = fe.transform_fwd(wav, L, False, False, -80.0, 80.0)
            l2, dmel = l2_loss(ref, pred)
            d2 = torch.zeros_like(wav)
            fe.transform_bwd(dmel, d2)
        else:
            y = op._a_fwd(wav, L)
            ref = op._mel(meas).clone()
            pred = op._mel(y)
            l2, dmel = l2_loss(ref, pred)
            d2 = op._a_bwd(fe.transform_bwd(dmel), full)
        assert float(((loss - l2).abs() / l2).max()) < 1e-5, type(op).__name__
        assert _rel(dwav, d2) < 1e-5, (type(op).__name__, _rel(dwav, d2))
        assert math.isfinite(float(dwav.abs().max()))
