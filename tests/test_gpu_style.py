"""-m gpu: style-guidance operator (BASELINE config 5; semantics defined by the build) against the
CPU restatement in oracle/operators.py with the SAME HTS-AT weights: CLAP log-mel features (HIP resampler + STFT / mel),
Gram transform, loss and the gradient w.r.t. the waveform (HIP transposes around torch autograd through the wrapped tower)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ops():
    from diffmusic_amd import inverse_problem as P
    from oracle import operators as OO
    op = P.StyleGuidanceOperator(16000, noiser=P.get_noiser("gaussian", 0.0), device="cuda", seed=3)
    rop = OO.StyleGuidanceOperator(16000, clap_model=copy.deepcopy(op.clap).cpu().float().eval(), noiser=OO.get_noiser("gaussian", 0.0))
    return op, rop


@pytest.mark.parametrize("length", [160000, 32000])
def test_style_features_and_gram(ops, length):
    import bench
    op, rop = ops
    wav = torch.stack([bench.synth_clip(0, length), bench.synth_clip(5, length)])
    feats, n48 = op._features(wav.cuda().contiguous(), length)
    ref = rop.features(wav)
    assert n48 == 3 * length and feats.shape == ref.shape == (2, 1, 1 + n48 // 480, 64)
    away = ref > -60.0
    err = (feats.cpu() - ref).abs()
    print("CLAP log-mel max |err| dB away from the clamp: %.2e" % float(err[away].max()))
    assert float(err[away].max()) < 2e-3                      # two resampler + STFT + mel stages in fp32
    g, gr = op.transform(wav.cuda()), rop.transform(wav)
    assert g.shape == gr.shape == (2, 768, 768)
    print("Gram rel-L2: %.2e" % _rel(g, gr))
    assert _rel(g, gr) < 2e-3


def test_style_guidance_loss_and_gradient(ops):
    import bench
    op, rop = ops
    L = 32000
    y = torch.stack([bench.synth_clip(1, L), bench.synth_clip(2, L)])
    g = torch.Generator().manual_seed(0)
    wav = (0.5 * y + 0.1 * torch.randn(2, L, generator=g)).contiguous()
    wav_full = torch.cat([wav, torch.zeros(2, 32)], dim=1)                 # vocoder output is longer than the clip
    meas = op.forward(y.cuda())
    loss, dwav = op.guidance(wav_full.cuda().contiguous(), L, meas, "mel_spectrogram")
    wr = wav.clone().requires_grad_(True)
    lr = torch.linalg.vector_norm((rop.transform(y) - rop.transform(wr)).flatten(1), dim=1)
    (gr,) = torch.autograd.grad(lr.sum(), wr)
    print("style loss rel %.2e, grad rel-L2 %.2e" % (_rel(loss, lr), _rel(dwav[:, :L], gr)))
    assert _rel(loss, lr) < 2e-3
    assert float(dwav[:, L:].abs().max()) == 0.0
    assert _rel(dwav[:, :L], gr) < 2e-2
    # the cached reference Gram is keyed on the measurement tensor: a new measurement is not served the old one
    meas2 = op.forward(torch.stack([bench.synth_clip(3, L), bench.synth_clip(4, L)]).cuda())
    loss2, _ = op.guidance(wav_full.cuda().contiguous(), L, meas2, "mel_spectrogram")
    assert _rel(loss2, loss) > 1e-3
