"""-m gpu: teacher-forced single guided step (same x_t, eps, noise) -- HIP engine vs the fp32 CPU
oracle with torch.autograd, for every scheduler x operator pair built so far (SURVEY.md section 8d
'Parity tolerance': rel-L2(prev_sample) <= 1e-2, loss rel err <= 1e-2 on the fp16 MFMA path)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

HIFI = dict(model_in_dim=64, upsample_initial_channel=128, upsample_rates=[5, 4, 2, 2, 2],
            upsample_kernel_sizes=[16, 16, 8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
            resblock_dilation_sizes=[[1, 3, 5]] * 3, leaky_relu_slope=0.1)
VAE = dict(latent_channels=8, out_channels=1, block_out_channels=[32, 64, 64], layers_per_block=2,
           norm_num_groups=32, scaling_factor=0.9227914214134216, eps=1e-6)
SCHED = dict(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear",
             trained_betas=None, clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon",
             thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
             timestep_spacing="leading", rescale_betas_zero_snr=False)
H, W = 10, 16           # latent (B,8,10,16) -> mel (B,40,64) -> wav 40*160+32 = 6432 samples, L = 6400
LEN = 6400


def _rel(a, b):
    return ((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm().clamp_min(1e-20)).item()


@pytest.fixture(scope="module")
def nets():
    from diffmusic_amd.engine import HifiGanEngine, VaeDecoderEngine
    from oracle.models import HifiGan, VaeDecoder
    voc, vae = HifiGanEngine(HIFI), VaeDecoderEngine(VAE)
    sv, sa = voc.synth_state_dict(seed=1), vae.synth_state_dict(seed=2)
    voc.load_state_dict(sv)
    vae.load_state_dict(sa)
    rvoc, rvae = HifiGan(**HIFI), VaeDecoder(**VAE)
    rvoc.load_state_dict(sv, strict=False)
    rvae.load_state_dict(sa, strict=True)
    return voc, vae, rvoc.eval(), rvae.eval()


def _ops(task):
    from diffmusic_amd import inverse_problem as P
    from oracle import operators as O
    if task == "music_inpainting":
        args = (1, LEN, "box", 0.25, 0.5, 0.3, 0.1, 0.2)
        return (P.MusicInpaintingOperator(*args, noiser=P.get_noiser("gaussian", 0.0)),
                O.MusicInpaintingOperator(*args, noiser=O.get_noiser("gaussian", 0.0)))
    if task == "phase_retrieval":
        return (P.PhaseRetrievalOperator(noiser=P.get_noiser("gaussian", 0.0)),
                O.PhaseRetrievalOperator(noiser=O.get_noiser("gaussian", 0.0)))
    if task == "super_resolution":
        return (P.SuperResolutionOperator(16000, 2, noiser=P.get_noiser("gaussian", 0.0)),
                O.SuperResolutionOperator(16000, 2, noiser=O.get_noiser("gaussian", 0.0)))
    if task == "super_resolution4":
        return (P.SuperResolutionOperator(16000, 4, noiser=P.get_noiser("gaussian", 0.0)),
                O.SuperResolutionOperator(16000, 4, noiser=O.get_noiser("gaussian", 0.0)))
    if task == "music_dereverberation":
        return (P.MusicDereverberationOperator(500, 0.99, noiser=P.get_noiser("gaussian", 0.0)),
                O.MusicDereverberationOperator(500, 0.99, noiser=O.get_noiser("gaussian", 0.0)))
    return P.IdentityOperator(16000), O.IdentityOperator(16000)


CASES = [("dps", "music_inpainting", 0.0, 5e-4, "mel_spectrogram", 501), ("dps", "music_inpainting", 0.0, 5e-4, "wav_form", 996),
         ("mpgd", "music_inpainting", 0.0, 5e-3, "mel_spectrogram", 251), ("dsg", "phase_retrieval", 1.0, 0.08, "mel_spectrogram", 501),
         ("diffmusic", "music_inpainting", 1.0, 0.08, "mel_spectrogram", 501), ("dps", "identity", 0.5, 5e-4, "mel_spectrogram", 101),
         ("ddim", "identity", 0.0, 0.0, "mel_spectrogram", 501), ("mpgd", "super_resolution4", 0.0, 5e-3, "mel_spectrogram", 501),
         ("dps", "super_resolution", 0.0, 5e-4, "wav_form", 251), ("dps", "music_dereverberation", 0.0, 5e-4, "mel_spectrogram", 501),
         ("dps", "phase_retrieval", 0.0, 5e-4, "wav_form", 501)]


@pytest.mark.parametrize("name,task,eta,rate,space,t", CASES)
def test_teacher_forced_step(nets, name, task, eta, rate, space, t):
    _teacher_forced(nets, name, task, eta, rate, space, t, True)


@pytest.mark.parametrize("name,task,eta,rate,space,t", [("dps", "music_inpainting", 0.0, 5e-4, "mel_spectrogram", 501),
                                                        ("dsg", "music_inpainting", 1.0, 0.08, "mel_spectrogram", 501),
                                                        ("diffmusic", "phase_retrieval", 1.0, 0.08, "mel_spectrogram", 251)])
def test_teacher_forced_step_whole_batch_norms(nets, name, task, eta, rate, space, t):
    """per_clip_norm=False: the reference's literal torch.linalg.norm over the whole batch tensor (loss, DSG / DiffMusic norms)."""
    _teacher_forced(nets, name, task, eta, rate, space, t, False)


@pytest.mark.parametrize("name,eta,rate,variant", [
    ("dps", 0.0, 5e-4, dict(prediction_type="v_prediction")),
    ("dps", 0.5, 5e-4, dict(prediction_type="epsilon", clip_sample=True, clip_sample_range=1.0)),
    ("dsg", 1.0, 0.08, dict(prediction_type="v_prediction", clip_sample=True, clip_sample_range=1.5)),
    ("diffmusic", 1.0, 0.08, dict(prediction_type="epsilon", clip_sample=True, clip_sample_range=2.0)),
    ("mpgd", 0.0, 5e-3, dict(prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")),
    ("mpgd", 0.0, 5e-3, dict(prediction_type="sample")),
    ("ddim", 0.0, 0.0, dict(prediction_type="sample")),
    ("ddim", 0.0, 0.0, dict(prediction_type="v_prediction", clip_sample=True, clip_sample_range=1.0)),
])
def test_teacher_forced_step_parent_variants(nets, name, eta, rate, variant):
    """The other branches of the diffusers DDIM parent every reference scheduler subclasses (scheduling_dps.py:15-61, :165-174): sample and
    v prediction, clip_sample, zero-terminal-SNR betas.  The guidance gradient passes through x0(x_t): Jacobian 1 / sqrt(a), 0 or sqrt(a),
    and nothing through a clipped element -- against the oracle's autograd."""
    t = 500 if variant.get("timestep_spacing") == "trailing" else 501
    _teacher_forced(nets, name, "music_inpainting", eta, rate, "mel_spectrogram", t, True, sched_kw=variant)


def _teacher_forced(nets, name, task, eta, rate, space, t, per_clip, sched_kw=None):
    from diffmusic_amd.schedulers import get_scheduler
    from oracle import schedulers as OS
    SCHED = dict(globals()["SCHED"], **(sched_kw or {}))
    voc, vae, rvoc, rvae = nets
    op, rop = _ops(task)
    B = 2
    g = torch.Generator().manual_seed(77)
    clean = 0.3 * torch.sin(torch.arange(LEN) * 0.05)[None] * torch.tensor([[1.0], [0.6]]) + 0.05 * torch.randn(B, LEN, generator=g)
    opk = {}
    if task == "music_dereverberation":       # pin the impulse response (the reference redraws it on every call)
        ir = rop.generate_impulse_response(500, 0.99)
        opk = dict(ir=ir)
    y_ref = rop.forward(clean, **opk)
    y = op.forward(clean.cuda(), **opk)
    assert _rel(y, y_ref) < 1e-4, "operator.forward"
    x = torch.randn(B, 8, H, W, generator=g)
    e = torch.randn(B, 8, H, W, generator=g)
    z = torch.randn(B, 8, H, W, generator=g)
    sched = get_scheduler(name)(operator=op, per_clip_norm=per_clip, **SCHED)
    sched.set_timesteps(200)
    sched.debug_keep_grad = True
    rs = OS.get_scheduler(name)(operator=rop, per_clip_norm=per_clip, **SCHED)
    rs.set_timesteps(200)
    kw = dict(eta=eta, ip_guidance_rate=rate, original_waveform_length=LEN, supervised_space=space)
    noise_kw = dict(sample_noise=z.cuda()) if name in ("dsg", "diffmusic") else dict(variance_noise=z.cuda() if eta > 0 else None)
    out = sched.step(e.cuda(), t, x.cuda(), measurement=y, vae=vae, vocoder=voc, op_kwargs=opk, **kw, **noise_kw)
    torch.cuda.synchronize()
    rnoise = dict(sample_noise=z) if name in ("dsg", "diffmusic") else dict(variance_noise=z if eta > 0 else None)
    ro = rs.step(e, t, x, measurement=y_ref, vae=rvae, vocoder=rvoc, op_kwargs=opk, **kw, **rnoise)
    rp, rl = _rel(out.prev_sample, ro.prev_sample), None
    assert _rel(out.pred_original_sample, ro.pred_original_sample) < 1e-4 or name == "mpgd"
    msg = f"{name}/{task}/{space}: prev {rp:.2e}"
    if name != "ddim":
        rl = _rel(out.loss.reshape(-1), ro.loss.reshape(-1))
        rg = _rel(sched.last_grad, ro.sample)
        cos = torch.nn.functional.cosine_similarity(sched.last_grad.cpu().flatten(), ro.sample.flatten(), dim=0).item()
        msg += f" loss {rl:.2e} grad {rg:.2e} cos {cos:.4f}"
        if (sched_kw or {}).get("clip_sample") and name != "mpgd":
            frac = float((sched.last_grad == 0).float().mean())
            msg += f" clipped {frac:.2f}"
            assert 0.02 < frac < 0.98, msg                      # the clip bound really cuts some elements and leaves others
        assert rl < 1e-2, msg
        assert cos > 0.98, msg
    print(msg)
    assert rp < 1e-2, msg


def test_sample_prediction_has_no_path_from_x_t_to_the_loss(nets):
    """prediction_type="sample": x0 = model_output does not depend on x_t, and the reference's torch.autograd.grad(rec_loss, sample)
    (scheduling_dps.py:212) raises for it; so do the schedulers that differentiate w.r.t. the sample."""
    from diffmusic_amd.schedulers import get_scheduler
    voc, vae, _, _ = nets
    op, _ = _ops("music_inpainting")
    sched = get_scheduler("dps")(operator=op, **dict(SCHED, prediction_type="sample"))
    sched.set_timesteps(200)
    x = torch.randn(1, 8, H, W).cuda()
    with pytest.raises(RuntimeError, match="does not depend on"):
        sched.step(x, 501, x, measurement=op.forward(torch.zeros(1, LEN).cuda()), vae=vae, vocoder=voc, original_waveform_length=LEN)
