"""MusicLDM pipeline facade with the reference's `__call__` surface
(diffmusic/pipelines/pipeline_musicldm.py:491-799): geometry, prepare_latents, the denoising loop
(U-Net on the CFG batch -> CFG combine -> scheduler.step), NaN-retry, final decode.

In scope: everything inside and right around the loop.
Out of scope: the CLAP text encoder -- pass `prompt_embeds` (and optionally
`negative_prompt_embeds`) directly, exactly as the reference signature allows.
Extension: all B clips are returned (the reference returns clip 0 only, :781)."""
import contextlib
import inspect
import os
import warnings
from types import SimpleNamespace

import numpy as np
import torch

from .. import _lib as L
from .. import ops
from ..engine import HifiGanEngine, UNetEngine, VaeDecoderEngine
from ..torch_utils import randn_tensor
from ..profiling import stage
from .. import parallel

import ctypes as C


class AudioPipelineOutput(SimpleNamespace):
    pass


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class MusicLDMPipeline:
    def __init__(self, vae, unet, vocoder, scheduler=None):
        self.vae, self.unet, self.vocoder, self.scheduler = vae, unet, vocoder, scheduler
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self.device = torch.device("cuda")
        self.nan_check_every = 1          # host check of the loss every k steps (reference: every step, :742)
        self.dedupe_cfg = False           # opt-in: run one U-Net pass when cond == uncond
        self.lanes = 1                    # > 1: clip lanes, the loop software-pipelined over clip groups (pipelines/lanes.py)
        self._lane_runner = None

    # ---- construction ---------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, repo_id, torch_dtype=None, seed=0, unet_config=None, vae_config=None, vocoder_config=None, **kw):
        """`repo_id`: a local checkpoint directory in the upstream layout -- {unet,vae,vocoder}/config.json + *.safetensors -- whose
        config files decide the architecture exactly as the reference's `from_pretrained(repo)` does (run.py:218; translated and checked
        key by key in diffmusic_amd/checkpoint.py, explicit `*_config` arguments win), or "synthetic" (seeded variance-preserving weights of
        the benchmark architecture; no checkpoint exists offline)."""
        if os.path.isdir(str(repo_id)):
            from ..checkpoint import ConfigError, read_configs
            given = dict(unet=unet_config, vae=vae_config, vocoder=vocoder_config)
            cfgs = read_configs(repo_id) if any(v is None for v in given.values()) else {}
            cfgs.update({k: v for k, v in given.items() if v is not None})
            n_ctx = sum(1 for d in cfgs["unet"].get("attn_cross_dims", [0]) if d)
            want = sum(1 for d in (cls.unet_default_config or {}).get("attn_cross_dims", [0]) if d)
            if n_ctx != want:
                raise ConfigError(f"{repo_id}/unet: a U-Net with {n_ctx} cross-attention context(s), but {cls.__name__} conditions on {want}")
            unet, vae, voc = UNetEngine(cfgs["unet"]), VaeDecoderEngine(cfgs["vae"]), HifiGanEngine(cfgs["vocoder"])
            from safetensors.torch import load_file
            for eng, sub in ((unet, "unet"), (vae, "vae"), (voc, "vocoder")):
                files = [f for f in os.listdir(os.path.join(repo_id, sub)) if f.endswith(".safetensors")]
                sd = {}
                for f in files:
                    sd.update(load_file(os.path.join(repo_id, sub, f)))
                eng.load_state_dict(sd, strict=True)          # fails with the full list of missing / mis-shaped / unknown tensors
        else:
            unet = UNetEngine(unet_config if unet_config is not None else cls.unet_default_config)
            vae, voc = VaeDecoderEngine(vae_config), HifiGanEngine(vocoder_config)
            for i, eng in enumerate((unet, vae, voc)):
                eng.load_state_dict(eng.synth_state_dict(seed=seed + i))
        return cls(vae, unet, voc)

    def to(self, device):
        self.device = torch.device(device)
        return self

    # ---- helpers --------------------------------------------------------------------------
    def prepare_extra_step_kwargs(self, generator, eta):          # pipeline_musicldm.py:335-343
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        extra = {}
        if "eta" in params:
            extra["eta"] = eta
        if "generator" in params:
            extra["generator"] = generator
        return extra

    def prepare_latents(self, batch_size, num_channels_latents, height, dtype, device, generator, latents=None):   # :406-426
        shape = (batch_size, num_channels_latents, int(height) // self.vae_scale_factor,
                 int(self.vocoder.config.model_in_dim) // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def mel_spectrogram_to_waveform(self, mel_spectrogram):       # :428-435
        if mel_spectrogram.dim() == 4:
            mel_spectrogram = mel_spectrogram.squeeze(1)
        return self.vocoder(mel_spectrogram).cpu().float()

    def save_mel_spectrogram(self, mel_spectrogram, path, sample_rate=16000, hop_length=160, gt_mel_spectrogram=None,
                             gt_sample_rate=16000, title="Mel-Spectrogram"):     # :462-489
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        m = torch.as_tensor(mel_spectrogram).detach().float().cpu().squeeze().numpy()
        n = 2 if gt_mel_spectrogram is not None else 1
        fig, axes = plt.subplots(n, 1, figsize=(10, 4 * n), squeeze=False)
        axes[0][0].imshow(m, origin="lower", aspect="auto")
        axes[0][0].set_title(title)
        if gt_mel_spectrogram is not None:
            axes[1][0].imshow(torch.as_tensor(gt_mel_spectrogram).float().cpu().squeeze().numpy(), origin="lower", aspect="auto")
            axes[1][0].set_title("ground truth")
        fig.savefig(path)
        plt.close(fig)

    default_guidance_scale = 2.0
    unet_default_config = None

    def _prepare_cond(self, prompt_embeds, negative_prompt_embeds, n_per, do_cfg, device, **extra):
        """MusicLDM: CLAP text embedding (B, 512) as class_labels; [uncond | text] on the CFG batch (:243-248)."""
        pe = prompt_embeds.to(device=device, dtype=torch.float32).repeat_interleave(n_per, dim=0)
        if do_cfg:
            # the reference encodes negative_prompt (default "") with CLAP as the unconditional branch (:208-238); __call__ does
            # that when a text front end is attached.  Without one the caller states cond == uncond (prompt="" everywhere in
            # run.py:122) by leaving negative_prompt_embeds None -- __call__ warns about it once.
            ne = negative_prompt_embeds if negative_prompt_embeds is not None else prompt_embeds
            ne = ne.to(device=device, dtype=torch.float32).repeat_interleave(n_per, dim=0)
            pe = torch.cat([ne, pe], dim=0)
        return dict(class_labels=pe)

    def _cond_is_symmetric(self, cond, B):
        return all(torch.equal(v[:B], v[B:]) for v in cond.values() if v is not None)

    def _unet_eps(self, latents, t_host, cond, guidance_scale, do_cfg):
        """U-Net on the (2B) CFG batch + combine (pipeline_musicldm.py:692-708), all HIP."""
        if not isinstance(cond, dict):
            cond = dict(class_labels=cond)
        x = latents.to(torch.float32).contiguous()
        B, dev = x.shape[0], x.device
        with stage("unet_cfg"):
            if not do_cfg:
                return self.unet.forward(x, torch.full((B,), float(t_host), device=dev), **cond)
            if self.dedupe_cfg and self._cond_is_symmetric(cond, B):
                half = {k: (v[:B] if v is not None else None) for k, v in cond.items()}
                return self.unet.forward(x, torch.full((B,), float(t_host), device=dev), **half)   # uncond + s*(text-uncond) == text
            x2 = torch.cat([x, x], dim=0)
            eps2 = self.unet.forward(x2, torch.full((2 * B,), float(t_host), device=dev), **cond)
            if ops.enabled():
                return ops.hip.cfg_combine(eps2, float(guidance_scale))          # torch.ops.diffmusic_hip.cfg_combine
            out = torch.empty_like(x)
            L.check(L.lib().dmx_sched_cfg_combine(C.c_void_p(eps2.data_ptr()), C.c_void_p(out.data_ptr()), out.numel(),
                                                  float(guidance_scale), _stream()), "cfg_combine")
            return out

    # ---- the call ---------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt=None, audio_length_in_s=None, num_inference_steps=200, guidance_scale=2.0, negative_prompt=None,
                 num_waveforms_per_prompt=1, eta=0.0, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, return_dict=True, callback=None, callback_steps=1, cross_attention_kwargs=None,
                 output_type="np", measurement=None, optim_prompt=False, ip_guidance_rate=0.0005,
                 optim_prompt_learning_rate=0.0001, optim_outer_loop=1, show_progress=True, prompt_type=None,
                 supervised_space="mel_spectrogram", shard=False, group=None, lanes=None):
        """Reference signature (pipeline_musicldm.py:493-519) plus `shard` / `group` (extension): with
        torch.distributed initialised, `shard=True` (or a process `group`) makes every rank run clips k = rank, rank + G, ... of
        the batch and all-gathers the finished waveforms once at the end (RCCL over xGMI); every rank returns all B clips.
        `lanes` (default `self.lanes`): > 1 runs this rank's clips as that many clip lanes whose U-Net forwards are hidden under
        each other's guidance sweeps (pipelines/lanes.py); a lane's clips see exactly a call on those clips alone."""
        front = getattr(self, "text_frontend", None)
        do_cfg = guidance_scale > 1.0
        if prompt_embeds is None:
            if front is None or prompt is None:
                raise NotImplementedError("no text front end attached: pass prompt_embeds (B, 512), or set pipe.text_frontend = "
                                          "ClapTextFrontEnd(...) (diffmusic_amd/pipelines/prompt.py) to use `prompt=`")
            prompt_embeds, ne = front.encode(prompt, negative_prompt, do_cfg)                  # pipeline_musicldm.py:119-250
            if negative_prompt_embeds is None:
                negative_prompt_embeds = ne
        elif do_cfg and negative_prompt_embeds is None and self._needs_negative_embeds():
            if front is not None:                                   # the reference's unconditional branch: CLAP("") (:208-238)
                negative_prompt_embeds = front.encode([negative_prompt or ""] * prompt_embeds.shape[0], None, False)[0]
            elif not self.assume_uncond_equals_cond:
                warnings.warn("prompt_embeds given without negative_prompt_embeds and no text front end attached: the unconditional "
                              "CFG branch reuses prompt_embeds, which equals the reference only for prompt=\"\" (its default, "
                              "run.py:122); pass negative_prompt_embeds or set pipe.assume_uncond_equals_cond = True", stacklevel=2)
        vcfg = self.vocoder.config
        vocoder_upsample_factor = np.prod(vcfg.upsample_rates) / vcfg.sampling_rate               # :602
        if audio_length_in_s is None:
            audio_length_in_s = 10.24
        height = int(audio_length_in_s / vocoder_upsample_factor)
        original_waveform_length = int(audio_length_in_s * vcfg.sampling_rate)
        if height % self.vae_scale_factor != 0:
            height = int(np.ceil(height / self.vae_scale_factor)) * self.vae_scale_factor
        device = self.device
        batch_size = prompt_embeds.shape[0]
        pe = self._prepare_cond(prompt_embeds, negative_prompt_embeds, num_waveforms_per_prompt, do_cfg, device,
                                **getattr(self, "_extra_cond", {}))
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = list(self.scheduler._timesteps_host)
        nlat = self.unet.cfg["in_channels"]
        B_all = B = batch_size * num_waveforms_per_prompt
        if measurement is not None:
            measurement = measurement.to(device=device, dtype=torch.float32)      # once per call (transform(y) is cached per tensor)
        # ---- clip sharding (no per-step collective; clips are independent because norms and RNG are per clip)
        sel = None
        if shard or group is not None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("shard=True needs an initialised torch.distributed process group")
            if isinstance(generator, torch.Generator):
                raise ValueError("clip sharding needs one generator per clip (a list of length B) so that the noise does not "
                                 "depend on the number of ranks")
            sel = parallel.shard_indices(B_all, dist.get_rank(group), dist.get_world_size(group))
            rows = sel + [B_all + k for k in sel] if do_cfg else sel
            pe = {k: (v[rows] if v is not None else None) for k, v in pe.items()}
            if isinstance(generator, (list, tuple)):
                if len(generator) != B_all:
                    raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                                     f"batch size of {B_all}.")
                generator = [generator[k] for k in sel]
            if latents is not None:
                latents = latents[sel]
            if measurement is not None and measurement.shape[0] == B_all and B_all > 1:
                measurement = measurement[sel].contiguous()
            B = len(sel)
        extra = self.prepare_extra_step_kwargs(generator, eta)
        self.last_losses = []
        self.nan_restarts = 0
        n_lanes = int(self.lanes if lanes is None else lanes)
        if B > 0 and n_lanes > 1 and B > 1:
            if callback is not None:
                raise ValueError("lanes > 1 cannot serve `callback(i, t, latents)`: the clip groups are at different steps at any one time")
            if isinstance(generator, torch.Generator) and eta > 0 or (isinstance(generator, torch.Generator) and
                                                                      self.scheduler.mode in ("dsg", "diffmusic")):
                raise ValueError("lanes > 1 with per-step noise needs one generator per clip (a list of length B): one shared "
                                 "generator would be drawn from in another order than by the whole batch")
            latents = self.prepare_latents(B, nlat, height, torch.float32, device, generator, latents)
            for _ in range(optim_outer_loop):
                latents = self._denoise_lanes(n_lanes, latents, pe, do_cfg, guidance_scale, measurement, timesteps, generator, eta,
                                              original_waveform_length, ip_guidance_rate, supervised_space, optim_prompt,
                                              optim_prompt_learning_rate, nlat, height, show_progress)
        elif B > 0:
            latents = self.prepare_latents(B, nlat, height, torch.float32, device, generator, latents)
            init_latents = latents
            init_pe = pe
            for _ in range(optim_outer_loop):
                retry = 10
                latents = init_latents
                while True:
                    is_done = True
                    pending = []
                    bar = None
                    if show_progress:
                        from tqdm import tqdm
                        bar = tqdm(total=num_inference_steps)
                    with bar if bar is not None else contextlib.nullcontext():
                        for i, t in enumerate(timesteps):
                            noise_pred = self._unet_eps(self.scheduler.scale_model_input(latents, t), t, pe, guidance_scale, do_cfg)
                            if optim_prompt and t % 30 == 1:                                        # :710-723
                                pe = self._optim_prompt_step(noise_pred, t, latents, pe, measurement, original_waveform_length,
                                                             optim_prompt_learning_rate, supervised_space, extra)
                            out = self.scheduler.step(noise_pred, t, latents, measurement=measurement,
                                                      original_waveform_length=original_waveform_length, vae=self.vae,
                                                      vocoder=self.vocoder, ip_guidance_rate=ip_guidance_rate,
                                                      ditto_optimizer=None, init_latents=init_latents,
                                                      supervised_space=supervised_space, **extra)
                            pending.append(out.loss)
                            last = i == len(timesteps) - 1
                            if (len(pending) >= self.nan_check_every or last) and retry >= 0:
                                bad = any(bool(torch.isnan(l.float()).any()) for l in pending)      # one host sync per check
                                self.last_losses.extend(pending)
                                pending = []
                                if bad:                                                             # NaN-retry (:741-756)
                                    retry -= 1
                                    self.nan_restarts += 1
                                    latents = self.prepare_latents(B, nlat, height, torch.float32, device, generator, None)
                                    is_done = False
                                    pe = init_pe
                                    break
                            latents = out.prev_sample.detach()
                            if bar is not None:
                                bar.update()
                            if callback is not None and i % callback_steps == 0:
                                callback(i, t, latents)
                    if is_done:
                        break
        else:
            latents = torch.zeros(0, nlat, int(height) // self.vae_scale_factor, int(vcfg.model_in_dim) // self.vae_scale_factor,
                                  dtype=torch.float32, device=device)
        if output_type == "latent":
            if sel is not None:
                shp = latents.shape[1:]
                latents = parallel.gather_waveforms(latents.reshape(B, int(np.prod(shp))).contiguous(), B_all, group).reshape(B_all, *shp)
            return AudioPipelineOutput(audios=latents)
        if B > 0:
            with stage("final_decode"):
                mel = self.vae.decode(latents / self.vae.config.scaling_factor).sample                 # (B,1,H,W) fp32
                audio = self.vocoder(mel.squeeze(1))[:, :original_waveform_length].float()         # :428-435, on the device
        else:
            audio = torch.zeros(0, original_waveform_length, dtype=torch.float32, device=device)
        if sel is not None:
            with stage("gather_waveforms"):
                audio = parallel.gather_waveforms(audio.contiguous(), B_all, group)                    # the only collective
        audio = audio.cpu()
        if output_type == "np":
            audio = audio.numpy()
        if not return_dict:
            return (audio,)
        return AudioPipelineOutput(audios=audio)

    def _denoise_lanes(self, n_lanes, latents, pe, do_cfg, guidance_scale, measurement, timesteps, generator, eta, length, rate,
                       supervised_space, optim_prompt, optim_lr, nlat, height, show_progress):
        """The loop of `__call__` (NaN-retry included) over clip lanes.  Lane k holds a contiguous group of this call's clips with
        their conditioning rows (both halves of the CFG batch), measurement rows and generators; every lane-step is the same
        `_unet_eps` + `scheduler.step` the plain loop makes, on the lane's stream (pipeline_musicldm.py:690-758)."""
        from .lanes import Lane, LaneRunner, split_sizes
        B, device = latents.shape[0], latents.device
        if self._lane_runner is None or self._lane_runner.device != device:
            self._lane_runner = LaneRunner(device)
        runner = self._lane_runner
        sizes = split_sizes(B, n_lanes)
        groups, o = [], 0
        for n in sizes:
            groups.append(list(range(o, o + n)))
            o += n

        def make_lanes(lat):
            out = []
            for ids in groups:
                rows = ids + [B + k for k in ids] if do_cfg else ids
                cond = {k: (v[rows].contiguous() if v is not None else None) for k, v in pe.items()}
                meas = measurement
                if measurement is not None and measurement.shape[0] == B and B > 1:
                    meas = measurement[ids].contiguous()
                gen = [generator[k] for k in ids] if isinstance(generator, (list, tuple)) else generator
                out.append(Lane(ids, lat[ids].contiguous(), cond, meas, gen))
            return out

        def unet_fn(ln, i):
            t = timesteps[i]
            return self._unet_eps(self.scheduler.scale_model_input(ln.latents, t), t, ln.cond, guidance_scale, do_cfg)

        def step_fn(ln, i, eps):
            t = timesteps[i]
            extra = self.prepare_extra_step_kwargs(ln.generator, eta)
            if optim_prompt and t % 30 == 1:
                ln.cond = self._optim_prompt_step(eps, t, ln.latents, ln.cond, ln.measurement, length, optim_lr, supervised_space, extra)
            out = self.scheduler.step(eps, t, ln.latents, measurement=ln.measurement, original_waveform_length=length, vae=self.vae,
                                      vocoder=self.vocoder, ip_guidance_rate=rate, ditto_optimizer=None, init_latents=None,
                                      supervised_space=supervised_space, **extra)
            return out.prev_sample.detach(), out.loss

        retry = 10
        while True:
            lanes = make_lanes(latents)
            bar = None
            if show_progress:
                from tqdm import tqdm
                bar = tqdm(total=len(timesteps))
            with bar if bar is not None else contextlib.nullcontext():
                bad = runner.run(lanes, len(timesteps), unet_fn, step_fn, self.nan_check_every if retry >= 0 else 10 ** 9,
                                 (lambda i: bar.update()) if bar is not None else None)
            n_done = min(len(ln.losses) for ln in lanes)
            self.last_losses.extend(torch.cat([ln.losses[i].reshape(-1) for ln in lanes]) for i in range(n_done))
            if bad is None or retry < 0:
                return torch.cat([ln.latents for ln in lanes], dim=0)
            retry -= 1                                                                  # NaN-retry (:741-756): all clips restart
            self.nan_restarts += 1
            latents = self.prepare_latents(B, nlat, height, torch.float32, device, generator, None)

    assume_uncond_equals_cond = False      # True: prompt_embeds without negative_prompt_embeds means prompt == "" (no warning)

    def _needs_negative_embeds(self):
        return True

    def _optim_prompt_step(self, noise_pred, t, latents, cond, measurement, length, lr, supervised_space, extra):
        """The reference's prompt-optimisation hook (pipeline_musicldm.py:710-723): calls scheduler.optim_prompt with the
        conditioning as encoder_hidden_states_1 and takes the returned embeddings (unchanged: see scheduler.optim_prompt)."""
        out = self.scheduler.optim_prompt(noise_pred, t, latents, measurement=measurement, original_waveform_length=length,
                                          vae=self.vae, vocoder=self.vocoder, encoder_hidden_states_1=cond.get("class_labels"),
                                          optim_prompt_learning_rate=lr, supervised_space=supervised_space, **extra)
        return dict(cond, class_labels=out.encoder_hidden_states_1)
