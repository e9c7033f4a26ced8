"""The five in-scope schedulers behind the reference's Scheduler protocol:
same constructor keys (configs/model/*.yaml `scheduler:`), `set_timesteps`, `timesteps`,
`scale_model_input`, `init_noise_sigma`, `order`, `config`, and `step(...)` with the reference's
signature (diffmusic/schedulers/scheduling_dps.py:137-156 and siblings).

Host side = DDIM tables and per-step scalars (fp32, computed once; no device sync per step).
Device side = HIP: x0 prediction, VAE decode, HiFi-GAN, measurement operator, mel, L2 loss, the
hand-written backward sweep and the fused update kernel (csrc/sched.hip).  There is no autograd and
no CPU fallback."""
import ctypes as C
from types import SimpleNamespace
import numpy as np
import torch

from .. import _lib as L
from .. import ops
from ..torch_utils import randn_tensor, randn_philox
from ..profiling import stage
from .utils import InverseProblemSchedulerOutput

_MODE = dict(ddim=0, dps=1, mpgd=2, dsg=3, diffmusic=4)
_PTYPE = dict(epsilon=0, sample=1, v_prediction=2)          # include/diffmusic_hip.h dmx_sched_pred_x0_ex


def rescale_zero_terminal_snr(betas):
    """diffusers `rescale_zero_terminal_snr` (Lin et al. 2023, Algorithm 1): shift / scale sqrt(alpha_bar) so that the last timestep has zero SNR."""
    alphas_bar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0, aT = alphas_bar_sqrt[0].clone(), alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt = (alphas_bar_sqrt - aT) * (a0 / (a0 - aT))
    alphas_bar = alphas_bar_sqrt ** 2
    alphas = torch.cat([alphas_bar[0:1], alphas_bar[1:] / alphas_bar[:-1]])
    return 1.0 - alphas


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class GuidedDDIMScheduler:
    """Common base: diffusers-0.31.0-compatible DDIM tables + the HIP guided step."""
    order = 1
    init_noise_sigma = 1.0
    mode = "ddim"
    default_eta = 0.0
    default_rate = 0.0
    passes_eta_to_parent = False      # DPS/MPGD: parent DDIM step consumes one randn draw when eta > 0

    def __init__(self, operator=None, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                 timestep_spacing="leading", rescale_betas_zero_snr=False, grad_target=64.0, per_clip_norm=True, device_noise=False,
                 *args, **kwargs):
        if prediction_type not in _PTYPE:
            raise ValueError(f"prediction_type given as {prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
        if thresholding:
            # (dynamic thresholding rescales x0 by a per-sample quantile of |x0| -- Imagen's trick for pixel-space models; no config of the
            #  reference's latent pipelines selects it: configs/model/*.yaml)
            raise NotImplementedError("thresholding=True (dynamic thresholding of x0) is not built")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
                                      set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                      prediction_type=prediction_type, thresholding=thresholding,
                                      timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr,
                                      clip_sample_range=clip_sample_range)
        self._ptype = _PTYPE[prediction_type]
        self._clip_r = float(clip_sample_range) if clip_sample else 0.0
        if trained_betas is not None:
            betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented")
        if rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)            # fp32, host
        if rescale_betas_zero_snr:
            self.alphas_cumprod[-1] = 2 ** -24                             # diffusers: "close to 0 without being 0 so first sigma is not inf"
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self._ac = self.alphas_cumprod.numpy()
        self.operator = operator
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.grad_target = grad_target
        # True (default): every clip of a batch is guided by its own loss / gradient norms, i.e. a batch equals B independent
        # B = 1 runs (the only case the reference ever ran).  False: the reference's literal whole-tensor norms
        # (torch.linalg.norm over the batch tensor, scheduling_dps.py:205-206; DSG / DiffMusic norms likewise).
        self.per_clip_norm = per_clip_norm
        self.last_grad = None
        self.debug_keep_grad = False
        # False (default): per-step noise comes from `randn_tensor` (host generators, uploaded), bit-identical to the reference's
        # stream.  True: drawn on the device by the Philox kernel (csrc/rng.hip), keyed per clip by the generators' initial seeds
        # -- a different (but equally per-clip, GPU-count-independent) stream, without the host draw and the H2D copy.
        self.device_noise = device_noise
        self._noise_offset = 0

    # ---- protocol pieces the pipelines touch
    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "leading":
            ratio = n_train // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        elif sp == "trailing":
            ts = np.round(np.arange(n_train, 0, -n_train / num_inference_steps)).astype(np.int64) - 1
        elif sp == "linspace":
            ts = np.linspace(0, n_train - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(f"{sp} is not supported")
        self._timesteps_host = [int(t) for t in ts]
        self._noise_offset = 0
        if self.operator is not None and hasattr(self.operator, "reset_cache"):
            self.operator.reset_cache()           # a new trajectory: forget the cached transform(measurement)
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def _get_variance(self, timestep, prev_timestep):
        a_t = float(self._ac[timestep])
        a_p = float(self._ac[prev_timestep]) if prev_timestep >= 0 else float(self.final_alpha_cumprod)
        return (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)

    def _scalars(self, timestep, eta):
        t = int(timestep)                       # a device tensor here costs one sync; the pipeline passes host ints
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self._ac[t])
        a_p = float(self._ac[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        sigma = eta * self._get_variance(t, prev_t) ** 0.5
        return t, a_t, a_p, sigma

    # ---- HIP guidance sweep: x0 -> vae -> vocoder -> A -> loss ; and back
    def _guidance(self, x0, measurement, vae, vocoder, length, supervised_space, op_kwargs=None):
        zs = 1.0 / vae.config.scaling_factor
        with stage("vae_fwd"):
            mel = vae.decode_hip(x0, z_scale=zs, keep_state=True)              # (B, H, W) fp16
        with stage("hifigan_fwd"):
            wav = vocoder.forward(mel)                                         # (B, Lfull) fp32
        with stage("operator_mel_loss_fwd_bwd"):
            loss, dwav = self.operator.guidance(wav, length, measurement, supervised_space, **(op_kwargs or {}))
            if not self.per_clip_norm and loss.numel() > 1:
                # whole-batch norm: L = sqrt(sum_b L_b^2), dL/dwav_b = (L_b / L) * dL_b/dwav_b  (device-side, B scalars)
                total = torch.linalg.vector_norm(loss)
                dwav.mul_((loss / total.clamp_min(1e-30))[:, None])
                loss = total.reshape(1)
            if ops.enabled():
                inv_scale = ops.hip.grad_normalize_(dwav, float(self.grad_target))       # in place on dwav
            else:
                inv_scale = torch.empty(x0.shape[0], dtype=torch.float32, device=x0.device)
                L.check(L.lib().dmx_grad_normalize(_p(dwav), _p(inv_scale), dwav.shape[0], dwav.shape[1], self.grad_target, _stream()),
                        "grad_normalize")
        with stage("hifigan_bwd"):
            dmel = vocoder.backward(dwav)
        with stage("vae_bwd"):
            g0 = vae.backward(dmel, z_scale=zs)                                # dLoss/dx0 * (1/inv_scale)
        return loss, g0, inv_scale

    def step(self, model_output, timestep, sample, eta=None, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True, measurement=None, ip_guidance_rate=None, vae=None, vocoder=None,
             original_waveform_length=0, supervised_space="mel_spectrogram", eps=1e-8, *args, **kwargs):
        eta = self.default_eta if eta is None else eta
        rate = self.default_rate if ip_guidance_rate is None else ip_guidance_rate
        if supervised_space not in ("wav_form", "mel_spectrogram"):
            raise ValueError("supervised_space should be either 'wav_form' or 'mel_spectrogram")
        if variance_noise is not None and generator is not None and eta > 0:
            raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either `generator` or"
                             " `variance_noise` stays `None`.")
        if not sample.is_cuda:
            raise RuntimeError("diffmusic_amd schedulers run on the GPU only (no CPU fallback)")
        if self._ptype == 1 and self.mode in ("dps", "dsg", "diffmusic"):
            raise RuntimeError("prediction_type='sample': pred_original_sample = model_output does not depend on the sample, so the gradient of "
                               "the loss w.r.t. the sample these schedulers take does not exist (the reference's torch.autograd.grad(rec_loss, "
                               "sample) raises for it, scheduling_dps.py:212)")
        t, a_t, a_p, sigma = self._scalars(timestep, eta)
        x = sample.detach().to(torch.float32).contiguous()
        e = model_output.detach().to(torch.float32).contiguous()
        B, n = x.shape[0], x[0].numel()
        lib = L.lib()
        plain = self._ptype == 0 and self._clip_r == 0.0          # epsilon prediction, no clipping: the reference's configs
        if not plain:
            # the DDIM parent's other branches (sample / v_prediction, clip_sample); C-ABI entry points with the type and the clip range
            x0 = torch.empty_like(x)
            L.check(lib.dmx_sched_pred_x0_ex(_p(x), _p(e), _p(x0), x.numel(), a_t, self._ptype, self._clip_r, _stream()), "pred_x0_ex")
        elif ops.enabled():
            x0 = ops.hip.sched_pred_x0(x, e, a_t)                  # torch.ops.diffmusic_hip.* (csrc_torch/torch_ops.cpp)
        else:
            x0 = torch.empty_like(x)
            L.check(lib.dmx_sched_pred_x0(_p(x), _p(e), _p(x0), x.numel(), a_t, _stream()), "pred_x0")
        mode = _MODE[self.mode]
        noise = None
        if self.mode in ("dps", "mpgd") and eta > 0:
            if variance_noise is None:
                if self.passes_eta_to_parent:          # keep the reference's RNG stream
                    randn_tensor(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
                variance_noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device,
                                              dtype=model_output.dtype)
            noise = variance_noise.to(torch.float32).contiguous()
        loss = g0 = inv_scale = None
        if self.mode == "ddim":
            loss = torch.tensor([t])
        else:
            loss, g0, inv_scale = self._guidance(x0, measurement, vae, vocoder, original_waveform_length, supervised_space,
                                                 kwargs.get("op_kwargs"))
            if self.mode in ("dsg", "diffmusic"):
                sn = kwargs.get("sample_noise")
                if sn is None and self.device_noise:
                    sn = self._philox_noise(model_output.shape, generator, model_output.device)
                elif sn is None:
                    sn = randn_tensor(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
                noise = sn.to(torch.float32).contiguous()
        grad_out = torch.empty_like(x) if self.debug_keep_grad and self.mode != "ddim" else None
        if not plain:
            prev = torch.empty_like(x)
            x0_out = torch.empty_like(x) if self.mode == "mpgd" else None
            with stage("sched_update"):
                L.check(lib.dmx_sched_step_ex(mode, _p(x), _p(e), _p(x0), _p(g0), _p(inv_scale), _p(noise), _p(prev), _p(x0_out),
                                              _p(grad_out), B, n, a_t, a_p, sigma, float(rate), float(eps), 0 if self.per_clip_norm else 1,
                                              self._ptype, self._clip_r, _stream()), "sched_step_ex")
        elif ops.enabled() and grad_out is None:
            with stage("sched_update"):
                prev, x0_u = ops.hip.sched_update(mode, x, e, x0, g0, inv_scale, noise, a_t, a_p, sigma, float(rate), float(eps),
                                                  not self.per_clip_norm)
            x0_out = x0_u if self.mode == "mpgd" else None
        else:
            prev = torch.empty_like(x)
            x0_out = torch.empty_like(x) if self.mode == "mpgd" else None
            with stage("sched_update"):
                L.check(lib.dmx_sched_step(mode, _p(x), _p(e), _p(x0), _p(g0), _p(inv_scale), _p(noise), _p(prev), _p(x0_out),
                                           _p(grad_out), B, n, a_t, a_p, sigma, float(rate), float(eps), 0 if self.per_clip_norm else 1,
                                           _stream()), "sched_step")
        self.last_grad = grad_out
        if loss.numel() == 1 and self.mode != "ddim":
            loss = loss.reshape(())
        return InverseProblemSchedulerOutput(prev_sample=prev.to(sample.dtype),
                                             pred_original_sample=(x0_out if x0_out is not None else x0).to(sample.dtype),
                                             loss=loss)

    def _philox_noise(self, shape, generator, device):
        gens = generator if isinstance(generator, (list, tuple)) else [generator] * shape[0]
        if any(g is None for g in gens):
            raise ValueError("device_noise=True needs generator(s): their initial seeds key the per-clip Philox streams")
        seeds = [int(g.initial_seed()) + (0 if isinstance(generator, (list, tuple)) else i) for i, g in enumerate(gens)]
        n = 1
        for d in shape[1:]:
            n *= int(d)
        out = randn_philox(shape, seeds, self._noise_offset, device)
        self._noise_offset += (n + 3) // 4
        return out

    def optim_prompt(self, model_output, timestep, sample, encoder_hidden_states=None, encoder_hidden_states_1=None, eta=0.0,
                     use_clipped_model_output=False, generator=None, variance_noise=None, return_dict=True, measurement=None,
                     vae=None, vocoder=None, original_waveform_length=0, optim_prompt_learning_rate=1e-4,
                     supervised_space="mel_spectrogram", *args, **kwargs):
        """`optim_prompt` of the reference (scheduling_dps.py:63-135 and siblings): meant to be one SGD step on the prompt
        embeddings, but the `requires_grad_` clones it makes (:93-96) are discarded and `model_output` was computed from the
        originals, so no gradient ever reaches the embeddings -- the call returns them unchanged.  Reproduced as that no-op
        (same signature, same output fields, same argument validation); the pipeline calls it every `t % 30 == 1` when
        `optim_prompt=True` (pipeline_musicldm.py:710-723)."""
        if supervised_space not in ("wav_form", "mel_spectrogram"):
            raise ValueError("supervised_space should be either 'wav_form' or 'mel_spectrogram")
        d = lambda v: v.detach() if v is not None else None          # noqa: E731  (the reference crashes on None here, :132)
        return InverseProblemSchedulerOutput(encoder_hidden_states=d(encoder_hidden_states),
                                             encoder_hidden_states_1=d(encoder_hidden_states_1))


class DDIMScheduler(GuidedDDIMScheduler):        # scheduling_ddim.py:58-104 (the formula; the reference body crashes)
    mode = "ddim"


class DPSScheduler(GuidedDDIMScheduler):         # scheduling_dps.py:137-219
    mode, default_eta, default_rate, passes_eta_to_parent = "dps", 0.0, 5e-4, True


class MPGDScheduler(GuidedDDIMScheduler):        # scheduling_mpgd.py:137-224
    mode, default_eta, default_rate, passes_eta_to_parent = "mpgd", 0.0, 1.0, True


class DSGScheduler(GuidedDDIMScheduler):         # scheduling_dsg.py:148-230
    mode, default_eta, default_rate = "dsg", 1.0, 0.08


class DiffMusicScheduler(GuidedDDIMScheduler):   # scheduling_diffmusic.py:148-229
    mode, default_eta, default_rate = "diffmusic", 0.0, 0.08
