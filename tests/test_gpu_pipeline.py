"""-m gpu: the pipeline facades end to end on small nets: MusicLDM + DDIM (config 1 plumbing) against the oracle
loop, MusicLDM + DPS (deterministic sampler, short trajectory SNR), AudioLDM2 + DSG (runs, finite, all clips)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from tests.test_gpu_step import HIFI, VAE, SCHED                                   # noqa: E402
UNET = dict(in_channels=8, out_channels=8, block_out_channels=[32, 64, 96, 160], layers_per_block=2, attention_heads=4,
            norm_num_groups=32, down_attn=[0, 1, 1, 1], up_attn=[1, 1, 1, 0], class_embed_dim=512)


def _build(name, unet_cfg, sched_name, op):
    from diffmusic_amd.pipelines import get_pipeline
    from diffmusic_amd.schedulers import get_scheduler
    pipe = get_pipeline(name).from_pretrained("synthetic", seed=0, unet_config=unet_cfg, vae_config=VAE, vocoder_config=HIFI)
    pipe.scheduler = get_scheduler(sched_name)(operator=op, **SCHED)
    return pipe.to("cuda")


def _oracle_nets(pipe, unet_kw):
    from oracle import models as OM
    ru, rv, rh = OM.UNetMusicLDM(**unet_kw).eval(), OM.VaeDecoder(**VAE).eval(), OM.HifiGan(**HIFI).eval()
    ru.load_state_dict(pipe.unet.synth_state_dict(0))
    rv.load_state_dict(pipe.vae.synth_state_dict(1))
    rh.load_state_dict(pipe.vocoder.synth_state_dict(2), strict=False)
    return ru, rv, rh


def test_musicldm_dps_short_trajectory_matches_oracle_loop():
    from diffmusic_amd import inverse_problem as P
    from oracle import operators as OO, schedulers as OS
    L, B, N = 6400, 2, 6
    args = (1, L, "box", 0.25, 0.5, 0.3, 0.1, 0.2)
    op = P.MusicInpaintingOperator(*args, noiser=P.get_noiser("gaussian", 0.0))
    pipe = _build("musicldm", UNET, "dps", op)
    g = torch.Generator().manual_seed(5)
    clean = 0.3 * torch.sin(torch.arange(L) * 0.05)[None].repeat(B, 1) + 0.05 * torch.randn(B, L, generator=g)
    y = op.forward(clean.cuda())
    pe = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    lat0 = torch.randn(B, 8, 10, 16, generator=g)
    out = pipe(prompt_embeds=pe, audio_length_in_s=0.4, num_inference_steps=N, guidance_scale=2.0, latents=lat0.clone(),
               measurement=y, ip_guidance_rate=5e-4, eta=0.0, show_progress=False, output_type="np")
    assert out.audios.shape == (B, L)
    # oracle loop (pipeline_musicldm.py:690-766 restated): CFG with cond == uncond
    ru, rv, rh = _oracle_nets(pipe, UNET)
    rop = OO.MusicInpaintingOperator(*args, noiser=OO.get_noiser("gaussian", 0.0))
    rs = OS.DPSScheduler(operator=rop, **SCHED)
    rs.set_timesteps(N)
    x, yr = lat0.clone(), rop.forward(clean)
    for t in [int(v) for v in rs.timesteps]:
        with torch.no_grad():
            e2 = ru(torch.cat([x, x]), t, class_labels=torch.cat([pe, pe]))[0]
        e = e2[:B] + 2.0 * (e2[B:] - e2[:B])
        x = rs.step(e, t, x, eta=0.0, measurement=yr, vae=rv, vocoder=rh, original_waveform_length=L, ip_guidance_rate=5e-4).prev_sample
    with torch.no_grad():
        wav = rh(rv.decode(x / VAE["scaling_factor"]).sample.squeeze(1))[:, :L]
    got = torch.from_numpy(out.audios)
    snr = 10 * torch.log10(wav.pow(2).sum() / (wav - got).pow(2).sum())
    print("waveform SNR vs oracle loop: %.1f dB" % snr.item())
    assert snr > 30.0                                      # deterministic samplers, short run >= 30 dB


def test_musicldm_ddim_generation_runs_config1_plumbing():
    from diffmusic_amd import inverse_problem as P
    pipe = _build("musicldm", UNET, "ddim", P.IdentityOperator(16000))
    pe = torch.nn.functional.normalize(torch.randn(1, 512, generator=torch.Generator().manual_seed(1)), dim=-1)
    out = pipe(prompt_embeds=pe, audio_length_in_s=0.4, num_inference_steps=5, generator=torch.Generator().manual_seed(0),
               show_progress=False)
    assert out.audios.shape == (1, 6400) and bool((abs(out.audios) <= 1.0).all())
    lat = pipe(prompt_embeds=pe, audio_length_in_s=0.4, num_inference_steps=5, generator=torch.Generator().manual_seed(0),
               show_progress=False, output_type="latent").audios
    assert lat.shape == (1, 8, 10, 16) and bool(torch.isfinite(lat).all())


def test_audioldm2_dsg_phase_retrieval_runs():
    from diffmusic_amd import inverse_problem as P
    op = P.PhaseRetrievalOperator(noiser=P.get_noiser("gaussian", 0.0))
    a2 = dict(UNET, class_embed_dim=0, attn_cross_dims=[0, 48, 64])
    pipe = _build("audioldm2", a2, "dsg", op)
    g = torch.Generator().manual_seed(2)
    B, L = 2, 6400
    y = op.forward((0.2 * torch.randn(B, L, generator=g)).cuda())
    gens = [torch.Generator().manual_seed(k) for k in range(B)]
    out = pipe(prompt_embeds=torch.randn(B, 10, 64, generator=g), attention_mask=torch.ones(B, 10),
               generated_prompt_embeds=torch.randn(B, 8, 48, generator=g), audio_length_in_s=0.4, num_inference_steps=4,
               generator=gens, measurement=y, eta=1.0, ip_guidance_rate=0.08, show_progress=False)
    assert out.audios.shape == (B, L) and bool(torch.isfinite(torch.from_numpy(out.audios)).all())
    assert len(pipe.last_losses) == 4 and all(bool(torch.isfinite(l).all()) for l in pipe.last_losses)


def test_example_driver_end_to_end(tmp_path):
    """examples/run_inverse_problem.py: run.py-style wiring, three guided steps, wav / mel outputs and metrics."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("run_inverse_problem", os.path.join(os.path.dirname(__file__), "..", "examples",
                                                                                      "run_inverse_problem.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(["-c", "dps", "-t", "music_inpainting", "--batch", "2", "--num_inference_steps", "3", "--output_dir", str(tmp_path)])
    out = tmp_path / "musicldm" / "moises" / "dps" / "music_inpainting"
    for d in ("wav_input", "wav_recon", "wav_label", "mel_recon"):
        assert len(list((out / d).iterdir())) == 2


def test_pipeline_accepts_prompt_through_text_frontend():
    """`pipe(prompt=...)` with a text front end attached (diffmusic_amd/pipelines/prompt.py) equals passing its embeddings."""
    import torch
    from types import SimpleNamespace
    from diffmusic_amd.pipelines import get_pipeline
    from diffmusic_amd.pipelines.prompt import ClapTextFrontEnd
    from diffmusic_amd.schedulers import get_scheduler
    from diffmusic_amd import inverse_problem as P

    class Tok:
        model_max_length = 8

        def __call__(self, texts, padding=None, max_length=None, truncation=False, return_tensors=None):
            n = max_length if padding == "max_length" else max(1, max(len(t) for t in texts))
            ids = torch.zeros(len(texts), n, dtype=torch.long)
            mask = torch.zeros(len(texts), n, dtype=torch.long)
            for i, t in enumerate(texts):
                r = [1 + (ord(c) % 50) for c in t][:n]
                ids[i, :len(r)] = torch.tensor(r, dtype=torch.long)
                mask[i, :len(r)] = 1
            return SimpleNamespace(input_ids=ids, attention_mask=mask)

        def batch_decode(self, ids):
            return [""] * len(ids)

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.emb = torch.nn.Embedding(64, 512)

        def get_text_features(self, ids, attention_mask=None):
            return torch.nn.functional.normalize((self.emb(ids) * attention_mask[..., None]).sum(1) + 1e-3, dim=-1)

    pipe = get_pipeline("musicldm").from_pretrained("synthetic", seed=0).to("cuda")
    pipe.scheduler = get_scheduler("ddim")(operator=P.IdentityOperator(16000), **SCHED)
    front = ClapTextFrontEnd(Enc(), Tok())
    pe, ne = front.encode(["soft piano"], None, True)
    kw = dict(num_inference_steps=2, audio_length_in_s=0.64, guidance_scale=2.0, show_progress=False, output_type="latent")
    a = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, generator=torch.Generator().manual_seed(5), **kw).audios
    pipe.text_frontend = front
    b = pipe(prompt=["soft piano"], generator=torch.Generator().manual_seed(5), **kw).audios
    assert torch.equal(a, b)


def test_from_pretrained_directory_with_non_default_architecture(tmp_path):
    """`from_pretrained(<dir>)` takes the architecture from the checkpoint's config.json files (run.py:218; diffmusic_amd/checkpoint.py): a
    directory in upstream layout with non-default widths loads and runs, and gives the bits of a pipeline configured by hand."""
    import json
    import os
    from safetensors.torch import save_file
    from diffmusic_amd import checkpoint as ck, inverse_problem as P
    from diffmusic_amd.engine import HifiGanEngine, UNetEngine, VaeDecoderEngine
    from diffmusic_amd.pipelines import get_pipeline
    from diffmusic_amd.schedulers import get_scheduler
    from tests.test_checkpoint_config import UNET_MUSICLDM, VAE as VAE_JSON, VOCODER
    ucfg = dict(UNET_MUSICLDM, block_out_channels=[32, 64, 96, 160])
    ucfg["cross_attention_dim"] = [32, 64, 96, 160]
    vcfg = dict(VAE_JSON, block_out_channels=[32, 64, 64])
    hcfg = dict(VOCODER, upsample_initial_channel=128)
    engines = {}
    for sub, cfg, fn, Eng in (("unet", ucfg, ck.unet_config, UNetEngine), ("vae", vcfg, ck.vae_config, VaeDecoderEngine),
                              ("vocoder", hcfg, ck.vocoder_config, HifiGanEngine)):
        os.makedirs(tmp_path / sub)
        with open(tmp_path / sub / "config.json", "w") as fh:
            json.dump(cfg, fh)
        eng = Eng(fn(cfg))
        sd = eng.synth_state_dict(seed=len(engines))
        save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / sub / "diffusion_pytorch_model.safetensors"))
        engines[sub] = eng.load_state_dict(sd)
    pipe = get_pipeline("musicldm").from_pretrained(str(tmp_path)).to("cuda")
    assert pipe.unet.cfg["block_out_channels"] == [32, 64, 96, 160] and pipe.vocoder.cfg["upsample_initial_channel"] == 128
    ref = get_pipeline("musicldm")(engines["vae"], engines["unet"], engines["vocoder"]).to("cuda")
    pe = torch.nn.functional.normalize(torch.randn(1, 512, generator=torch.Generator().manual_seed(1)), dim=-1)
    outs = []
    for p in (pipe, ref):
        p.scheduler = get_scheduler("ddim")(operator=P.IdentityOperator(16000), **SCHED)
        p.assume_uncond_equals_cond = True
        outs.append(p(prompt_embeds=pe, audio_length_in_s=0.4, num_inference_steps=3, generator=torch.Generator().manual_seed(0),
                      show_progress=False).audios)
    assert outs[0].shape == (1, 6400) and bool((abs(outs[0]) <= 1.0).all()) and float(abs(outs[0]).max()) > 1e-4
    assert (outs[0] == outs[1]).all()


def test_pipeline_runs_at_a_length_whose_deepest_level_has_an_odd_row_count():
    """0.96 s -> latent 24 x 16 -> U-Net levels of 24, 12, 6, 3 rows: 3 x 2 = 6 tokens in the deepest attention (an 8 s clip has 25 x 2 = 50).
    The attention kernels take any key count (they used to require a multiple of 4: 8 s clips failed)."""
    from diffmusic_amd import inverse_problem as P
    pipe = _build("musicldm", UNET, "dps", P.MusicInpaintingOperator(1, 15360, "box", 0.25, 0.5, 0.3, 0.1, 0.2, noiser=P.get_noiser("gaussian", 0.0)))
    pipe.assume_uncond_equals_cond = True
    g = torch.Generator().manual_seed(3)
    L = 15360
    y = pipe.scheduler.operator.forward((0.2 * torch.randn(1, L, generator=g)).cuda())
    pe = torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)
    out = pipe(prompt_embeds=pe, audio_length_in_s=0.96, num_inference_steps=3, generator=[torch.Generator().manual_seed(0)], measurement=y,
               show_progress=False)
    assert out.audios.shape == (1, L) and bool(torch.isfinite(torch.from_numpy(out.audios)).all()) and pipe.nan_restarts == 0
