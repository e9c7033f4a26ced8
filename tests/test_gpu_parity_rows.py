"""-m gpu: the small parity rows.
* a2-a6: `dmx_sched_step` (csrc/sched.hip) driven DIRECTLY with the golden fixtures produced by the reference's own scheduler
  sources (tests/golden/scheduler_steps.npz): x, eps, the reference's noise draw and a gradient computed by autograd through the
  fixtures' toy vae / vocoder -- one hop HIP == reference instead of HIP ~ oracle ~ reference.
* a20: `dmx_audio_transform_fwd` (log-mel) against oracle.audio.Wav2Mel at <= 1e-4 dB away from the 1e-10 clamp.
* a22: NaN-retry with the real engines and a fault-injecting scheduler wrapper."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden.cases import CASES, SCHED_CFG, L, SR            # noqa: E402
from tests.golden.toy import ToyVae, ToyVocoder                   # noqa: E402

_MODE = dict(ddim=0, dps=1, mpgd=2, dsg=3, diffmusic=4)


def _oracle_op(task):
    from oracle import operators as O
    n = O.get_noiser("gaussian", 0.0)
    if task == "music_inpainting":
        return O.MusicInpaintingOperator(1, L, "box", 0.25, 0.5, 0.3, 0.1, 0.2, noiser=n)
    if task == "phase_retrieval":
        return O.PhaseRetrievalOperator(noiser=n)
    return O.SuperResolutionOperator(SR, 2, noiser=n)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_sched_step_kernel_matches_reference_golden(golden_dir, ci):
    from diffmusic_amd import _lib as Lb
    from oracle.ddim import DDIMParent
    steps = np.load(os.path.join(golden_dir, "scheduler_steps.npz"))
    name, task, eta, rate, n_steps, space = CASES[ci]
    tab = DDIMParent(**SCHED_CFG)
    tab.set_timesteps(n_steps)
    op, vae, voc = _oracle_op(task), ToyVae(), ToyVocoder()
    lib = Lb.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    keys = sorted({k.rsplit("/", 1)[0] for k in steps.files if k.startswith(f"c{ci}_")})
    assert len(keys) == 3
    for key in keys:
        t, n, eta_, rate_, seed = steps[key + "/meta"]
        t = int(t)
        x, eps, y = (torch.from_numpy(steps[key + "/" + s]) for s in ("x", "eps", "y"))
        prev_t = t - 1000 // n_steps
        a_t = float(tab.alphas_cumprod[t])
        a_p = float(tab.alphas_cumprod[prev_t]) if prev_t >= 0 else float(tab.final_alpha_cumprod)
        sigma = eta * float(tab._get_variance(t, prev_t)) ** 0.5
        # gradient of the loss w.r.t. x0 through the toy decoder chain (what the HIP VAE / vocoder backward would deliver)
        g0 = None
        if name != "ddim":
            x0r = ((x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5).requires_grad_(True)
            wav = op.forward(op.inverse_transform(vae.decode(x0r / vae.config.scaling_factor).sample, voc)[:, :L])
            diff = (y - wav) if space == "wav_form" else (op.transform(y) - op.transform(wav))
            g0 = torch.autograd.grad(torch.linalg.norm(diff), x0r)[0]
        # the reference's noise: one randn_tensor draw from the step's generator (DSG / DiffMusic; scheduling_dsg.py:215)
        noise = None
        if name in ("dsg", "diffmusic"):
            noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(int(seed)))
        elif name in ("dps", "mpgd") and eta > 0:
            # eta > 0: the parent DDIM step takes the generator's FIRST draw (and throws its prev_sample away); the scheduler's own
            # variance noise is the SECOND (scheduling_dps.py:166-193, scheduling_mpgd.py:164-173,206-217)
            gen = torch.Generator().manual_seed(int(seed))
            torch.randn(x.shape, generator=gen)
            noise = torch.randn(x.shape, generator=gen)
        xd, ed = x.cuda().contiguous(), eps.cuda().contiguous()
        x0d, prev = torch.empty_like(xd), torch.empty_like(xd)
        x0o = torch.empty_like(xd) if name == "mpgd" else None
        Lb.check(lib.dmx_sched_pred_x0(_p(xd), _p(ed), _p(x0d), xd.numel(), a_t, st), "pred_x0")
        g0d = g0.cuda().contiguous() if g0 is not None else None
        nd = noise.cuda().contiguous() if noise is not None else None
        Lb.check(lib.dmx_sched_step(_MODE[name], _p(xd), _p(ed), _p(x0d), _p(g0d), None, _p(nd), _p(prev), _p(x0o), None,
                                    1, xd[0].numel(), a_t, a_p, sigma, float(rate), 1e-8, 1, st), "sched_step")
        torch.cuda.synchronize()
        ref = steps[key + "/prev_sample"]
        got = prev.cpu().numpy()
        assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), (key, float(np.abs(got - ref).max()))
        ref0 = steps[key + "/pred_original_sample"]
        got0 = (x0o if x0o is not None else x0d).cpu().numpy()
        assert np.abs(got0 - ref0).max() <= 5e-5 * max(1.0, np.abs(ref0).max()), key


@pytest.mark.parametrize("ci", [i for i, c in enumerate(CASES) if c[0] in ("dps", "mpgd") and c[2] > 0])
def test_eta_double_draw_through_product_step(golden_dir, ci, monkeypatch):
    """DPS / MPGD at eta > 0 through the PRODUCT's `Scheduler.step` (host RNG handling + HIP update kernel): with `generator=` the
    product must throw one draw away (`passes_eta_to_parent`, the parent DDIM step's, scheduling_dps.py:166-175) and use the second as
    variance noise, or it does not land on the reference's `prev_sample`.  The guidance sweep is replaced by the fixtures' toy decoder
    chain (autograd), so everything else in `step` is the shipped code."""
    from diffmusic_amd.schedulers import get_scheduler
    steps = np.load(os.path.join(golden_dir, "scheduler_steps.npz"))
    name, task, eta, rate, n_steps, space = CASES[ci]
    op, vae, voc = _oracle_op(task), ToyVae(), ToyVocoder()
    sched = get_scheduler(name)(operator=None, **SCHED_CFG)
    assert sched.passes_eta_to_parent
    sched.set_timesteps(n_steps)
    keys = sorted({k.rsplit("/", 1)[0] for k in steps.files if k.startswith(f"c{ci}_")})
    assert len(keys) == 3
    for key in keys:
        t, n, eta_, rate_, seed = steps[key + "/meta"]
        x, eps, y = (torch.from_numpy(steps[key + "/" + s]) for s in ("x", "eps", "y"))

        def toy_guidance(x0, measurement, vae_, vocoder_, length, supervised_space, op_kwargs=None):
            x0r = x0.detach().cpu().requires_grad_(True)
            wav = op.forward(op.inverse_transform(vae.decode(x0r / vae.config.scaling_factor).sample, voc)[:, :L])
            diff = (y - wav) if supervised_space == "wav_form" else (op.transform(y) - op.transform(wav))
            loss = torch.linalg.norm(diff)
            g0 = torch.autograd.grad(loss, x0r)[0]
            return loss.detach().reshape(1).cuda(), g0.cuda().contiguous(), torch.ones(x0.shape[0], device="cuda")

        monkeypatch.setattr(sched, "_guidance", toy_guidance)
        o = sched.step(eps.cuda(), int(t), x.cuda(), eta=eta, generator=torch.Generator().manual_seed(int(seed)), measurement=y,
                       vae=vae, vocoder=voc, original_waveform_length=L, ip_guidance_rate=rate, supervised_space=space)
        torch.cuda.synchronize()
        for f in ("prev_sample", "pred_original_sample"):
            ref, got = steps[key + "/" + f], getattr(o, f).cpu().numpy()
            assert np.abs(got - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), (key, f, float(np.abs(got - ref).max()))
        assert abs(float(o.loss) - float(steps[key + "/loss"].reshape(-1)[0])) <= 1e-4 * max(1.0, float(steps[key + "/loss"].reshape(-1)[0]))
        # a single draw (what a scheduler that ignores the parent's draw would use) does NOT reproduce the reference
        one = torch.randn(x.shape, generator=torch.Generator().manual_seed(int(seed)))
        o1 = sched.step(eps.cuda(), int(t), x.cuda(), eta=eta, variance_noise=one.cuda(), measurement=y, vae=vae, vocoder=voc,
                        original_waveform_length=L, ip_guidance_rate=rate, supervised_space=space)
        assert np.abs(o1.prev_sample.cpu().numpy() - steps[key + "/prev_sample"]).max() > 1e-3


def test_operator_fixtures_one_hop_hip(golden_dir):
    """The reference-generated operator fixtures fed straight to the HIP operators (one hop HIP == reference): rectangular-window
    |STFT| of `PhaseRetrievalOperator.forward` (operator.py:162-171) and the reverb convolution of
    `MusicDereverberationOperator.forward` with the fixture's impulse response (operator.py:238-250)."""
    from diffmusic_amd import inverse_problem as P
    fx = np.load(os.path.join(golden_dir, "operators.npz"))
    wav = torch.from_numpy(fx["wav"]).cuda()
    mag = P.PhaseRetrievalOperator(noiser=None).forward(wav)
    ref = fx["phase_retrieval/forward"]
    assert tuple(mag.shape) == ref.shape == (2, 513, 26)
    err = float(np.abs(mag.cpu().numpy() - ref).max())
    assert err <= 2e-4 * max(1.0, float(np.abs(ref).max())), err
    dr = P.MusicDereverberationOperator(ir_length=500, decay_factor=0.99, noiser=None)
    out = dr.forward(wav, ir=torch.from_numpy(fx["dereverb_seed77/ir"]))
    ref = fx["dereverb_seed77/forward"]
    assert tuple(out.shape) == ref.shape == (2, 4001)
    err = float(np.abs(out.cpu().numpy() - ref).max())
    assert err <= 1e-4 * max(1.0, float(np.abs(ref).max())), err
    # the product draws its own response like the reference does (global RNG, operator.py:244-246): same seed, same response
    torch.manual_seed(77)
    out2 = dr.forward(wav)
    assert float(np.abs(out2.cpu().numpy() - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("length", [160000, 6400])
def test_logmel_kernel_vs_oracle_wav2mel(length):
    """a20: MelSpectrogram(16000,1024,160,1024,n_mels=64,power=2) + AmplitudeToDB('power').  Truth = the oracle evaluated in
    float64; SURVEY.md section 8d asks <= 1e-4 dB away from the 1e-10 clamp.  Held on every bin within 40 dB of the clip's
    strongest bin; weaker bins sit below the fp32 rounding floor of a 1024-term dot product (any fp32 STFT, torch.stft's
    included, is off by more there), so they are bounded at 2e-3 dB and compared with the fp32 oracle's own error."""
    import bench
    from diffmusic_amd import inverse_problem as P
    from oracle.audio import Wav2Mel
    g = torch.Generator().manual_seed(5)
    wav = torch.stack([bench.synth_clip(0, length), 0.3 * torch.randn(length, generator=g), torch.zeros(length)])
    wav[2, length // 3: length // 2] = bench.synth_clip(1, length)[length // 3: length // 2]      # silence + a burst: exercises the clamp
    truth = Wav2Mel(16000)(wav.double())                                 # (B, 64, T) float64
    ref32 = Wav2Mel(16000)(wav).double()
    fe = P.IdentityOperator(16000).frontend
    got = fe.transform_fwd(wav.cuda().contiguous(), length, True, True).transpose(1, 2).cpu().double()
    assert got.shape == truth.shape == (3, 64, 1 + length // 160)
    away = truth > -60.0                                                 # mel power > 1e-6: 4 decades above the clamp
    strong = truth > truth.amax(dim=(1, 2), keepdim=True) - 40.0
    err, err32 = (got - truth).abs(), (ref32 - truth).abs()
    print("log-mel |err| dB vs float64: strong bins HIP %.2e (fp32 oracle %.2e); all bins away from the clamp HIP %.2e (fp32 oracle %.2e); "
          "at the clamp %.2e" % (float(err[strong].max()), float(err32[strong].max()), float(err[away].max()), float(err32[away].max()),
                                 float(err[~away].max()) if (~away).any() else 0.0))
    assert float(err[strong].max()) <= 1e-4
    assert float(err[away].max()) <= 2e-3
    assert float(err.max()) <= 1e-2
    # clamp variant used by the other operators (operator.py:35-36): identical after clamping
    got_c = fe.transform_fwd(wav.cuda().contiguous(), length, True, True, -80.0, 80.0).transpose(1, 2).cpu().double()
    assert float((got_c - truth.clamp(-80, 80)).abs()[away].max()) <= 2e-3


def test_nan_retry_with_real_engines():
    """a22 on the GPU: the scheduler's loss is NaN once at step 2 => latents redrawn, trajectory restarted, run completes."""
    from tests.test_gpu_pipeline import _build, UNET
    from diffmusic_amd import inverse_problem as P
    L_, B, N = 6400, 2, 5
    op = P.MusicInpaintingOperator(1, L_, "box", 0.25, 0.5, 0.3, 0.1, 0.2, noiser=P.get_noiser("gaussian", 0.0))
    pipe = _build("musicldm", UNET, "dps", op)
    pipe.assume_uncond_equals_cond = True
    real_step = pipe.scheduler.step
    state = dict(calls=0, first=[])

    def step(model_output, timestep, sample, **kw):
        out = real_step(model_output, timestep, sample, **kw)
        if timestep == pipe.scheduler._timesteps_host[0]:
            state["first"].append(sample.clone())
        if state["calls"] == 2:
            out.loss = out.loss * float("nan")
        state["calls"] += 1
        return out
    pipe.scheduler.step = step
    g = torch.Generator().manual_seed(3)
    y = op.forward((0.2 * torch.randn(B, L_, generator=g)).cuda())
    pe = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    out = pipe(prompt_embeds=pe, audio_length_in_s=0.4, num_inference_steps=N, measurement=y, show_progress=False,
               generator=[torch.Generator().manual_seed(k) for k in range(B)])
    assert pipe.nan_restarts == 1 and state["calls"] == 3 + N
    assert len(state["first"]) == 2 and not torch.equal(state["first"][0], state["first"][1])
    assert out.audios.shape == (B, L_) and np.isfinite(out.audios).all()


def test_device_philox_noise_matches_numpy_restatement():
    """csrc/rng.hip vs oracle/rng.py (Philox4x32-10 known-answer vectors are checked on the CPU side): values to 2e-5 (fp32
    log / sin / cos), per-clip keys, offset continuation, and the DSG scheduler's device_noise option."""
    from diffmusic_amd.torch_utils import randn_philox
    from oracle.rng import randn_philox as ref_philox
    shape = (3, 8, 25, 16)
    n = 8 * 25 * 16
    seeds = [0, 1234, 2 ** 40 + 7]
    a = randn_philox(shape, seeds, 0, "cuda").cpu().reshape(3, -1).double().numpy()
    for b, sd in enumerate(seeds):
        assert np.abs(a[b] - ref_philox(n, sd, 0)).max() < 2e-5
    off = (n + 3) // 4
    b2 = randn_philox(shape, seeds, off, "cuda").cpu().reshape(3, -1).double().numpy()
    both = ref_philox(2 * n, seeds[1], 0)
    assert np.abs(b2[1] - both[n:]).max() < 2e-5                       # the second draw continues the clip's stream
    big = randn_philox((2, 1, 1, 400001), [5, 6], 0, "cuda")            # odd length: scalar tail stores
    assert abs(float(big.mean())) < 5e-3 and abs(float(big.std()) - 1.0) < 5e-3 and abs(float((big ** 4).mean()) - 3.0) < 0.05
    one = randn_philox((1, 8, 25, 16), [1234], 0, "cuda").cpu().reshape(-1).double().numpy()
    assert np.array_equal(one, a[1])                                   # a clip's noise does not depend on its batch


def test_scheduler_device_noise_option():
    """DSG with device_noise=True: the per-step noise comes from the Philox kernel keyed by the generators' seeds -- the step equals
    the same step fed that noise explicitly, repeats exactly after set_timesteps, and advances its offset from step to step."""
    from tests.test_gpu_step import nets as _nets_fixture, _ops, SCHED, LEN, H, W   # noqa: F401
    from diffmusic_amd.engine import HifiGanEngine, VaeDecoderEngine
    from tests.test_gpu_step import HIFI, VAE
    from diffmusic_amd.schedulers import get_scheduler
    from diffmusic_amd.torch_utils import randn_philox
    voc, vae = HifiGanEngine(HIFI), VaeDecoderEngine(VAE)
    voc.load_state_dict(voc.synth_state_dict(seed=1))
    vae.load_state_dict(vae.synth_state_dict(seed=2))
    op, _ = _ops("music_inpainting")
    g = torch.Generator().manual_seed(11)
    B = 2
    y = op.forward((0.2 * torch.randn(B, LEN, generator=g)).cuda())
    x, e = torch.randn(B, 8, H, W, generator=g).cuda(), torch.randn(B, 8, H, W, generator=g).cuda()
    kw = dict(eta=1.0, ip_guidance_rate=0.08, measurement=y, vae=vae, vocoder=voc, original_waveform_length=LEN)
    gens = lambda: [torch.Generator().manual_seed(100 + k) for k in range(B)]       # noqa: E731
    s = get_scheduler("dsg")(operator=op, device_noise=True, **SCHED)
    s.set_timesteps(200)
    a1 = s.step(e, 501, x, generator=gens(), **kw).prev_sample.clone()
    a2 = s.step(e, 496, x, generator=gens(), **kw).prev_sample.clone()              # second call: next Philox blocks
    s.set_timesteps(200)
    b1 = s.step(e, 501, x, generator=gens(), **kw).prev_sample.clone()
    assert torch.equal(a1, b1) and not torch.equal(a1, a2)
    n = 8 * H * W
    z1 = randn_philox(x.shape, [100, 101], 0, "cuda")
    z2 = randn_philox(x.shape, [100, 101], (n + 3) // 4, "cuda")
    ref = get_scheduler("dsg")(operator=op, **SCHED)
    ref.set_timesteps(200)
    assert torch.equal(ref.step(e, 501, x, sample_noise=z1, **kw).prev_sample, a1)
    assert torch.equal(ref.step(e, 496, x, sample_noise=z2, **kw).prev_sample, a2)
    host = ref.step(e, 501, x, generator=gens(), **kw).prev_sample
    assert not torch.equal(host, a1)                                                # the default stays torch's generator stream


@pytest.mark.parametrize("B,n", [(8, 160000), (3, 160004), (2, 1000), (4, 40001)])
def test_grad_normalize_scales_each_clip_to_the_target(B, n):
    """dmx_grad_normalize (the per-clip rescale that keeps fp16 gradients in range before the vocoder backward): max|x| == target per clip,
    inv_scale undoes it exactly, long clips (parallel chunk maxima) and short / unaligned ones (one workgroup per clip) agree with torch."""
    import ctypes as C
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(B, n, generator=g) * torch.logspace(-6, 2, B).unsqueeze(1)).cuda().contiguous()
    x[0, n // 3] = -7.5e3                                   # the maximum may be negative and sit anywhere
    ref = x.clone()
    inv = torch.empty(B, device="cuda")
    L.check(L.lib().dmx_grad_normalize(C.c_void_p(x.data_ptr()), C.c_void_p(inv.data_ptr()), B, n, 64.0,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "grad_normalize")
    torch.cuda.synchronize()
    m = ref.abs().amax(dim=1)
    s = 64.0 / m
    assert torch.allclose(x, ref * s.unsqueeze(1), rtol=1e-6, atol=0)
    assert torch.allclose(inv, 1.0 / s, rtol=1e-6, atol=0)
    assert torch.allclose(x.abs().amax(dim=1), torch.full((B,), 64.0, device="cuda"), rtol=1e-6)
