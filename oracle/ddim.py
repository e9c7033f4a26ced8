"""Restatement of the diffusers 0.31.0 `DDIMScheduler` parent used by every reference scheduler
(reference: diffmusic/schedulers/scheduling_dps.py:15-61 subclasses it; requirements.txt:6 pins
diffusers==0.31.0).  Third-party semantics restated from the public source; parity unpinned
(diffusers is not installed here)."""
import math
import numpy as np
import torch


class DDIMParent:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02,
                 beta_schedule="linear", trained_betas=None, clip_sample=True,
                 set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0,
                 sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, **kwargs):
        self.cfg = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                        beta_end=beta_end, beta_schedule=beta_schedule, clip_sample=clip_sample,
                        set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                        prediction_type=prediction_type, timestep_spacing=timestep_spacing,
                        clip_sample_range=clip_sample_range)
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            # diffusers rescale_zero_terminal_snr: shift / scale sqrt(alpha_bar) to zero terminal SNR, back to betas
            abs_ = torch.cumprod(1.0 - self.betas, dim=0).sqrt()
            a0, aT = abs_[0].clone(), abs_[-1].clone()
            abs_ = (abs_ - aT) * (a0 / (a0 - aT))
            ab = abs_ ** 2
            self.betas = 1.0 - torch.cat([ab[0:1], ab[1:] / ab[:-1]])
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        if rescale_betas_zero_snr:
            self.alphas_cumprod[-1] = 2 ** -24
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        n_train = self.cfg["num_train_timesteps"]
        self.num_inference_steps = num_inference_steps
        sp = self.cfg["timestep_spacing"]
        if sp == "leading":
            step_ratio = n_train // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            ts += self.cfg["steps_offset"]
        elif sp == "trailing":
            step_ratio = n_train / num_inference_steps
            ts = np.round(np.arange(n_train, 0, -step_ratio)).astype(np.int64) - 1
        elif sp == "linspace":
            ts = np.linspace(0, n_train - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts)

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)

    def parent_step(self, model_output, timestep, sample, eta=0.0, generator=None,
                    variance_noise=None):
        """Returns (prev_sample, pred_original_sample) exactly as DDIMScheduler.step (diffusers 0.31.0) for the three
        prediction types and clip_sample; thresholding=False.  With eta>0 and
        variance_noise None it draws one randn from `generator` (the 'throw-away' draw noted in
        SURVEY.md section 7)."""
        from .rng import randn_tensor
        prev_t = timestep - self.cfg["num_train_timesteps"] // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        pt = self.cfg["prediction_type"]
        if pt == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif pt == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif pt == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample`, or `v_prediction`")
        if self.cfg["clip_sample"]:
            r = self.cfg["clip_sample_range"]
            x0 = x0.clamp(-r, r)
        var = self._get_variance(timestep, prev_t)
        std = eta * var ** 0.5
        direction = (1 - a_p - std ** 2) ** 0.5 * eps
        prev = a_p ** 0.5 * x0 + direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator,
                                              device=model_output.device, dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return prev, x0
