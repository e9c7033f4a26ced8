"""The five reference Scheduler.step() bodies restated on CPU fp32 with torch.autograd
(reference: diffmusic/schedulers/scheduling_{ddim,dps,mpgd,dsg,diffmusic}.py).
`per_clip_norm=True` generalises the reference's whole-tensor norms (which only ever ran at
B=1, run.py:249) to per-clip norms; False is the literal formula."""
from dataclasses import dataclass
from typing import Optional
import torch
from .ddim import DDIMParent
from .rng import randn_tensor


@dataclass
class InverseProblemSchedulerOutput:                 # schedulers/utils.py:8-16
    sample: Optional[torch.Tensor] = None
    prev_sample: torch.Tensor = None
    pred_original_sample: Optional[torch.Tensor] = None
    loss: Optional[torch.Tensor] = None
    encoder_hidden_states: Optional[torch.Tensor] = None
    encoder_hidden_states_1: Optional[torch.Tensor] = None
    init_latents: Optional[torch.Tensor] = None


def _norm(x, per_clip):
    if per_clip:
        return torch.linalg.norm(x.reshape(x.shape[0], -1), dim=1).reshape(-1, *([1] * (x.dim() - 1)))
    return torch.linalg.norm(x)


class _Guided(DDIMParent):
    def __init__(self, operator=None, per_clip_norm=True, **kw):
        super().__init__(**kw)
        self.operator = operator
        self.per_clip_norm = per_clip_norm

    def _scalars(self, timestep, eta):
        t = int(timestep)
        prev_t = t - self.cfg["num_train_timesteps"] // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        var = self._get_variance(t, prev_t)
        return t, a_t, 1 - a_t, a_p, eta * var ** 0.5

    def _loss(self, x0, measurement, vae, vocoder, L, supervised_space, op_kwargs=None):
        """scheduling_dps.py:195-211: decode -> vocoder -> A(.) -> (mel) -> L2."""
        mel = vae.decode(1 / vae.config.scaling_factor * x0).sample
        wav = self.operator.inverse_transform(mel, vocoder)
        wav = wav[:, :L]
        wav = self.operator.forward(wav, **(op_kwargs or {}))
        if supervised_space == "wav_form":
            diff = measurement - wav
        elif supervised_space == "mel_spectrogram":
            diff = self.operator.transform(measurement) - self.operator.transform(wav)
        else:
            raise ValueError("supervised_space should be either 'wav_form' or 'mel_spectrogram")
        if self.per_clip_norm:
            per = torch.linalg.norm(diff.reshape(diff.shape[0], -1), dim=1)
            return per.sum(), per          # d(sum of per-clip norms)/dx == per-clip gradients
        n = torch.linalg.norm(diff)
        return n, n


class DDIMScheduler(_Guided):                        # scheduling_ddim.py:58-104
    def step(self, model_output, timestep, sample, eta=0.0, generator=None, variance_noise=None, **kw):
        t, a_t, b_t, a_p, _ = self._scalars(timestep, eta)
        _, x0 = self.parent_step(model_output, t, sample, eta, generator, variance_noise)
        eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        prev = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps
        return InverseProblemSchedulerOutput(prev_sample=prev.detach(), pred_original_sample=x0,
                                             loss=torch.tensor([t]))


class DPSScheduler(_Guided):                         # scheduling_dps.py:137-219
    def step(self, model_output, timestep, sample, eta=0.0, generator=None, variance_noise=None,
             measurement=None, ip_guidance_rate=5e-4, vae=None, vocoder=None,
             original_waveform_length=0, supervised_space="mel_spectrogram", op_kwargs=None, **kw):
        t, a_t, b_t, a_p, std = self._scalars(timestep, eta)
        with torch.enable_grad():
            sample = sample.clone().detach().requires_grad_(True)
            _, x0 = self.parent_step(model_output, t, sample, eta, generator, variance_noise)
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
            prev = a_p ** 0.5 * x0 + (1 - a_p - std ** 2) ** 0.5 * eps
            if eta > 0:
                if variance_noise is None:
                    variance_noise = randn_tensor(model_output.shape, generator=generator,
                                                  device=model_output.device, dtype=model_output.dtype)
                prev = prev + std * variance_noise
            loss_sum, loss = self._loss(x0, measurement, vae, vocoder, original_waveform_length,
                                        supervised_space, op_kwargs)
            g = torch.autograd.grad(loss_sum, sample)[0]
            prev = prev - ip_guidance_rate * g
        return InverseProblemSchedulerOutput(prev_sample=prev.detach(), pred_original_sample=x0.detach(),
                                             loss=loss.detach(), sample=g.detach())


class MPGDScheduler(_Guided):                        # scheduling_mpgd.py:137-224
    def step(self, model_output, timestep, sample, eta=0.0, generator=None, variance_noise=None,
             measurement=None, ip_guidance_rate=1.0, vae=None, vocoder=None,
             original_waveform_length=0, supervised_space="mel_spectrogram", op_kwargs=None, **kw):
        t, a_t, b_t, a_p, std = self._scalars(timestep, eta)
        _, x0 = self.parent_step(model_output, t, sample, eta, generator, variance_noise)
        with torch.enable_grad():
            x0 = x0.clone().detach().requires_grad_(True)
            loss_sum, loss = self._loss(x0, measurement, vae, vocoder, original_waveform_length,
                                        supervised_space, op_kwargs)
            g = torch.autograd.grad(loss_sum, x0)[0]
            x0 = x0.detach() - ip_guidance_rate * g
        eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        prev = a_p ** 0.5 * x0 + (1 - a_p - std ** 2) ** 0.5 * eps
        if eta > 0:
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator,
                                              device=model_output.device, dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return InverseProblemSchedulerOutput(prev_sample=prev.detach(), pred_original_sample=x0,
                                             loss=loss.detach(), sample=g.detach())


class DSGScheduler(_Guided):                         # scheduling_dsg.py:148-230
    def step(self, model_output, timestep, sample, eta=1.0, generator=None, variance_noise=None,
             measurement=None, vae=None, vocoder=None, original_waveform_length=0,
             ip_guidance_rate=0.08, eps=1e-8, supervised_space="mel_spectrogram", op_kwargs=None,
             sample_noise=None, **kw):
        t, a_t, b_t, a_p, std = self._scalars(timestep, eta)
        with torch.enable_grad():
            sample = sample.clone().detach().requires_grad_(True)
            _, x0 = self.parent_step(model_output, t, sample, 0.0, generator, variance_noise)  # :178-186 no eta
            mean = a_p ** 0.5 * x0 + (1 - a_p - std ** 2) ** 0.5 * model_output
            loss_sum, loss = self._loss(x0, measurement, vae, vocoder, original_waveform_length,
                                        supervised_space, op_kwargs)
            grad = torch.autograd.grad(loss_sum / 1000, sample)[0]
            grad_norm = _norm(grad, self.per_clip_norm)
            _, c, h, w = sample.shape
            r = torch.sqrt(torch.tensor(c * h * w)) * std
            d_star = -r * grad / (grad_norm + eps)
            if sample_noise is None:
                sample_noise = randn_tensor(model_output.shape, generator=generator,
                                            device=model_output.device, dtype=model_output.dtype)
            d_sample = std * sample_noise
            mix = d_sample + ip_guidance_rate * (d_star - d_sample)
            prev = mean + r * mix / (_norm(mix, self.per_clip_norm) + eps)
        return InverseProblemSchedulerOutput(prev_sample=prev.detach(), pred_original_sample=x0.detach(),
                                             loss=loss.detach(), sample=grad.detach())


def slerp(x0, x1, gamma=0.008, threshold=0.9995, per_clip=False):   # scheduling_diffmusic.py:59-68
    if not per_clip:
        cos_theta = ((x0 / torch.norm(x0)) * (x1 / torch.norm(x1))).sum()
        if cos_theta.abs() > threshold:
            return x0 + gamma * (x1 - x0)
        theta = torch.acos(cos_theta)
        s = torch.sin(theta)
        return torch.sin((1 - gamma) * theta) / s * x0 + torch.sin(gamma * theta) / s * x1
    return torch.cat([slerp(x0[i:i + 1], x1[i:i + 1], gamma, threshold) for i in range(x0.shape[0])], 0)


class DiffMusicScheduler(_Guided):                   # scheduling_diffmusic.py:148-229
    def step(self, model_output, timestep, sample, eta=0.0, generator=None, variance_noise=None,
             measurement=None, vae=None, vocoder=None, original_waveform_length=0,
             ip_guidance_rate=0.08, eps=1e-8, supervised_space="mel_spectrogram", op_kwargs=None,
             sample_noise=None, **kw):
        t, a_t, b_t, a_p, std = self._scalars(timestep, eta)
        with torch.enable_grad():
            sample = sample.clone().detach().requires_grad_(True)
            _, x0 = self.parent_step(model_output, t, sample, 0.0, generator, variance_noise)
            mean = a_p ** 0.5 * x0 + (1 - a_p - std ** 2) ** 0.5 * model_output
            loss_sum, loss = self._loss(x0, measurement, vae, vocoder, original_waveform_length,
                                        supervised_space, op_kwargs)
            grad = torch.autograd.grad(loss_sum / 1000, sample)[0]
            grad_norm = _norm(grad, self.per_clip_norm)
            if sample_noise is None:
                sample_noise = randn_tensor(model_output.shape, generator=generator,
                                            device=model_output.device, dtype=model_output.dtype)
            ngrad = grad / (grad_norm + eps) * _norm(sample_noise, self.per_clip_norm)
            mixed = slerp(sample_noise, -ngrad, ip_guidance_rate, per_clip=self.per_clip_norm)
            prev = mean + std * mixed
        return InverseProblemSchedulerOutput(prev_sample=prev.detach(), pred_original_sample=x0.detach(),
                                             loss=loss.detach(), sample=grad.detach())


def get_scheduler(name):                             # schedulers/__init__.py:9-24
    table = dict(ddim=DDIMScheduler, dps=DPSScheduler, mpgd=MPGDScheduler, dsg=DSGScheduler,
                 diffmusic=DiffMusicScheduler)
    if name not in table:
        raise ValueError(f"Unknown scheduler: {name}")
    return table[name]
