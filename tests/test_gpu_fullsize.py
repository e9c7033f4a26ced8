"""-m gpu: BASELINE.json's full sizes (10 s @ 16 kHz, T_mel = 1000, latent 250 x 16), where the fp32 oracle is too slow
to run: size-independent properties of the path instead -- adjointness <A x, y> = <x, A^T y> of the linear operators and
their hand-written transposes, linearity of the hand-written backward passes in the incoming gradient, independence of the
clips of a batch in a guided step, idempotence / symmetry of the measurement operators."""
import math
import sys
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

L_FULL, SR = 160000, 16000


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("task", ["super_resolution_x4", "super_resolution_x2", "dereverberation_5000"])
def test_fullsize_operator_adjoint(task):
    """The FIR / polyphase-resampler kernels and their hand-written transposes are exact adjoints at 160 000 samples."""
    from diffmusic_amd import inverse_problem as P
    g = torch.Generator().manual_seed(0)
    B = 2
    x = torch.randn(B, L_FULL, generator=g).cuda()
    if task.startswith("super"):
        op = P.SuperResolutionOperator(SR, int(task[-1]), noiser=None)
        y = op._a_fwd(x, L_FULL)
    else:
        op = P.MusicDereverberationOperator(ir_length=5000, decay_factor=0.99, noiser=None, fixed_ir=True)
        op._h, op._hrev = op._get_ir(x.device)
        y = op._a_fwd(x, L_FULL)
    w = torch.randn(y.shape, generator=g).cuda()
    xt = op._a_bwd(w.contiguous(), L_FULL)
    lhs, rhs = _dot(y, w), _dot(x, xt)
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs), math.sqrt(y.numel())), (lhs, rhs)


def test_fullsize_logmel_vjp_linear_and_directional():
    """log-mel front end at full length: the VJP is linear in the incoming gradient and matches a central difference."""
    from diffmusic_amd import inverse_problem as P
    g = torch.Generator().manual_seed(1)
    B = 2
    x = (0.3 * torch.randn(B, L_FULL, generator=g)).cuda()
    fe = P.IdentityOperator(SR).frontend
    mel = fe.transform_fwd(x, L_FULL, True, True, -80.0, 80.0).clone()
    assert mel.shape == (B, 1001, 64)
    d1, d2 = torch.randn(mel.shape, generator=g).cuda(), torch.randn(mel.shape, generator=g).cuda()
    g1, g2 = fe.transform_bwd(d1.contiguous()).clone(), fe.transform_bwd(d2.contiguous()).clone()
    g12 = fe.transform_bwd((0.7 * d1 - 1.3 * d2).contiguous()).clone()
    assert _rel(g12, 0.7 * g1 - 1.3 * g2) < 1e-4
    v = torch.randn(B, L_FULL, generator=g).cuda()
    eps = 1e-3
    mp = fe.transform_fwd((x + eps * v).contiguous(), L_FULL, True, True, -80.0, 80.0).clone()
    mm = fe.transform_fwd((x - eps * v).contiguous(), L_FULL, True, True, -80.0, 80.0).clone()
    fd = _dot((mp - mm) / (2 * eps), d1)
    an = _dot(g1, v)
    assert abs(fd - an) <= 2e-2 * max(abs(fd), abs(an)), (fd, an)


def test_fullsize_measurement_symmetries():
    from diffmusic_amd import inverse_problem as P
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, L_FULL, generator=g).cuda()
    inp = P.MusicInpaintingOperator(10, SR, "box", 2, 3, 0.3, 0.1, 1.0, noiser=None)
    y = inp.forward(x)
    assert torch.equal(inp.forward(y), y)                                  # masking is idempotent
    assert float(y[:, 2 * SR:3 * SR].abs().max()) == 0.0 and torch.equal(y[:, :2 * SR], x[:, :2 * SR])
    pr = P.PhaseRetrievalOperator(noiser=None)
    m1, m2 = pr.forward(x), pr.forward(-x)
    assert m1.shape == (2, 513, 1001)
    assert _rel(m1, m2) < 1e-6                                             # |STFT| ignores the global sign
    e_t = float((x.double() ** 2).sum())
    # rectangular window, hop 160, n_fft 1024: every sample sits in 6.4 frames on average -> Parseval up to edge effects
    e_f = float(((m1.double() ** 2)[:, 1:-1].sum() * 2 + (m1.double() ** 2)[:, 0].sum() + (m1.double() ** 2)[:, -1].sum()) / 1024)
    assert abs(e_f / (6.4 * e_t) - 1.0) < 0.01


@pytest.mark.parametrize("net", ["hifigan", "vae"])
def test_fullsize_backward_is_linear(net):
    """With the tape of one forward pass fixed, the hand-written input-gradient sweep is a linear map of the incoming
    gradient (every leaky-relu' / SiLU' / softmax' factor is a constant of the tape): checked at the benchmark shapes."""
    from diffmusic_amd.engine import HifiGanEngine, VaeDecoderEngine
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    B = 2
    if net == "hifigan":
        eng = HifiGanEngine(); eng.load_state_dict(eng.synth_state_dict(2))
        mel = torch.randn(B, 1000, 64, generator=g).to(L.act_dtype()).cuda()
        wav = eng.forward(mel)
        assert wav.shape == (B, 160032)
        d1, d2 = torch.randn(wav.shape, generator=g).cuda(), torch.randn(wav.shape, generator=g).cuda()
        run = lambda d: eng.backward(d.contiguous()).float().clone()
    else:
        eng = VaeDecoderEngine(); eng.load_state_dict(eng.synth_state_dict(1))
        z = torch.randn(B, 8, 250, 16, generator=g).cuda()
        mel = eng.decode_hip(z, z_scale=1.0, keep_state=True)
        assert mel.shape == (B, 1000, 64)
        d1, d2 = torch.randn(mel.shape, generator=g).cuda(), torch.randn(mel.shape, generator=g).cuda()
        run = lambda d: eng.backward(d.to(L.act_dtype()).contiguous()).float().clone()
    g1, g2, g12 = run(d1), run(d2), run(0.5 * d1 + 0.25 * d2)
    assert torch.isfinite(g12).all()
    assert _rel(g12, 0.5 * g1 + 0.25 * g2) < 2e-2          # 16-bit activations along ~40 layers


def test_fullsize_guided_step_is_per_clip():
    """One DPS inpainting step at the benchmark shape: a batch of 3 clips equals the same clips stepped one by one."""
    import bench
    dev = torch.device("cuda")
    pipe, op, meas, lat, cond, Lw = bench.build_problem(3, 0, dev)
    t = pipe.scheduler._timesteps_host[60]
    prev, loss = bench.one_step(pipe, lat, t, cond, meas, Lw)
    prev, loss = prev.clone(), loss.clone()
    assert torch.isfinite(prev).all() and loss.numel() in (1, 3)
    pe = cond["class_labels"]
    gens = pipe._bench["gens"]
    for i in (0, 2):
        pipe._bench["gens"] = gens[i:i + 1]
        ci = dict(class_labels=torch.cat([pe[i:i + 1], pe[3 + i:4 + i]], dim=0))
        pi, li = bench.one_step(pipe, lat[i:i + 1].contiguous(), t, ci, meas[i:i + 1].contiguous(), Lw)
        assert _rel(pi, prev[i:i + 1]) < 2e-3
        if loss.numel() == 3:
            assert abs(float(li.reshape(-1)[0]) - float(loss.reshape(-1)[i])) <= 2e-3 * abs(float(loss.reshape(-1)[i]))


@pytest.mark.parametrize("seconds,task", [(5, "music_inpainting"), (2.56, "super_resolution")])
def test_other_clip_lengths_through_the_pipeline(seconds, task):
    """The shipped model config runs 5 s clips (configs/model/musicldm.yaml) and the geometry code accepts any length: full-size
    networks, two guided steps, shapes / finiteness / determinism (tile selection falls back to the cost model for these M)."""
    from diffmusic_amd.pipelines import get_pipeline
    from diffmusic_amd.schedulers import get_scheduler
    from diffmusic_amd import inverse_problem as P
    from tests.test_gpu_step import SCHED
    dev = torch.device("cuda")
    L = int(seconds * SR)
    g = torch.Generator().manual_seed(4)
    clips = (0.2 * torch.randn(2, L, generator=g)).to(dev)
    if task == "music_inpainting":
        op = P.MusicInpaintingOperator(seconds, SR, "periodic", 1, 2, 0.3, 0.1, 1.0, noiser=P.get_noiser("gaussian", 0.0))
        name = "dps"
    else:
        op = P.SuperResolutionOperator(SR, 2, noiser=P.get_noiser("gaussian", 0.0))
        name = "mpgd"
    meas = op.forward(clips)
    pipe = get_pipeline("musicldm").from_pretrained("synthetic", seed=0).to(dev)
    pipe.scheduler = get_scheduler(name)(operator=op, **SCHED)
    pe = torch.nn.functional.normalize(torch.randn(2, 512, generator=g), dim=-1)
    outs = []
    for _ in range(2):
        gens = [torch.Generator().manual_seed(10), torch.Generator().manual_seed(11)]
        a = pipe(prompt_embeds=pe, measurement=meas, num_inference_steps=2, audio_length_in_s=seconds, generator=gens,
                 show_progress=False, eta=0.0, ip_guidance_rate=5e-4).audios
        outs.append(a)
    assert outs[0].shape == (2, L)
    assert np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0], outs[1])              # same seeds, same kernels: bit-identical reruns


def test_every_stage_is_bitwise_reproducible():
    """No atomics and no data races anywhere on the path: two runs on the same inputs are bit-identical for every network stage
    (this caught a write-after-read race on the weight ring of the fused resblock-pair kernel that tolerance tests missed)."""
    from diffmusic_amd.engine import HifiGanEngine, VaeDecoderEngine, UNetEngine
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    B, T = 2, 500
    voc = HifiGanEngine(); voc.load_state_dict(voc.synth_state_dict(2))
    mel = torch.randn(B, T, 64, generator=g).to(L.act_dtype()).cuda()
    d = torch.randn(B, voc.forward(mel).shape[1], generator=g).cuda()
    runs = []
    for _ in range(3):
        w = voc.forward(mel).clone()
        runs.append((w, voc.backward(d.clone()).clone()))
    for w, gm in runs[1:]:
        assert torch.equal(w, runs[0][0]) and torch.equal(gm, runs[0][1])
    vae = VaeDecoderEngine(); vae.load_state_dict(vae.synth_state_dict(1))
    z = torch.randn(B, 8, T // 4, 16, generator=g).cuda()
    dm = torch.randn(B, T, 64, generator=g).to(L.act_dtype()).cuda()
    runs = []
    for _ in range(2):
        m = vae.decode_hip(z, z_scale=1.0, keep_state=True).clone()
        runs.append((m, vae.backward(dm.clone()).clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    un = UNetEngine(); un.load_state_dict(un.synth_state_dict(0))
    x = torch.randn(2 * B, 8, T // 4, 16, generator=g).cuda()
    t = torch.full((2 * B,), 501.0).cuda()
    c = torch.randn(2 * B, 512, generator=g).cuda()
    assert torch.equal(un.forward(x, t, c).clone(), un.forward(x, t, c).clone())
