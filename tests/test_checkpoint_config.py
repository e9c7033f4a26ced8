"""The architecture comes from the checkpoint's config.json files (diffmusic_amd/checkpoint.py), as in the reference's
`from_pretrained(repo)` (run.py:218, configs/model/*.yaml:2): upstream key names are translated into the executors' configs, anything
the HIP executors do not implement is refused by key, and a directory with NON-default widths yields executors whose parameter
manifests are exactly the tensors of that architecture.  (CPU: creating an executor and reading its manifest needs no GPU; the
upload itself is covered by tests/test_gpu_pipeline.py::test_from_pretrained_directory_with_non_default_architecture.)"""
import json
import os

import pytest
import torch

UNET_MUSICLDM = dict(_class_name="UNet2DConditionModel", _diffusers_version="0.31.0", act_fn="silu", attention_head_dim=4,
                     block_out_channels=[64, 128, 192, 320], center_input_sample=False, class_embed_type="simple_projection",
                     class_embeddings_concat=True, cross_attention_dim=[64, 128, 192, 320],
                     down_block_types=["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
                     downsample_padding=1, dual_cross_attention=False, flip_sin_to_cos=True, freq_shift=0, in_channels=8, layers_per_block=2,
                     mid_block_scale_factor=1, mid_block_type="UNetMidBlock2DCrossAttn", norm_eps=1e-05, norm_num_groups=32,
                     num_class_embeds=None, only_cross_attention=False, out_channels=8, projection_class_embeddings_input_dim=512,
                     resnet_time_scale_shift="default", sample_size=128, time_embedding_type="positional", timestep_post_act=None,
                     up_block_types=["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"],
                     upcast_attention=False, use_linear_projection=False)
UNET_AUDIOLDM2 = dict(UNET_MUSICLDM, _class_name="AudioLDM2UNet2DConditionModel", class_embed_type=None, class_embeddings_concat=False,
                      projection_class_embeddings_input_dim=None, cross_attention_dim=[[None, 96, 160]] * 4)
VAE = dict(_class_name="AutoencoderKL", act_fn="silu", block_out_channels=[64, 128, 256], down_block_types=["DownEncoderBlock2D"] * 3,
           in_channels=1, latent_channels=8, layers_per_block=2, norm_num_groups=32, out_channels=1, sample_size=512,
           scaling_factor=0.9227914214134216, up_block_types=["UpDecoderBlock2D"] * 3)
VOCODER = dict(architectures=["SpeechT5HifiGan"], initializer_range=0.01, leaky_relu_slope=0.1, model_in_dim=64, model_type="hifigan",
               normalize_before=False, resblock_dilation_sizes=[[1, 3, 5]] * 3, resblock_kernel_sizes=[3, 7, 11], sampling_rate=16000,
               upsample_initial_channel=256, upsample_kernel_sizes=[16, 16, 8, 4, 4], upsample_rates=[5, 4, 2, 2, 2])


def test_upstream_keys_translate_into_executor_configs():
    from diffmusic_amd import checkpoint as ck
    u = ck.unet_config(UNET_MUSICLDM)
    assert u == dict(in_channels=8, out_channels=8, block_out_channels=[64, 128, 192, 320], layers_per_block=2, norm_num_groups=32,
                     down_attn=[0, 1, 1, 1], up_attn=[1, 1, 1, 0], attention_heads=4, class_embed_dim=512, attn_cross_dims=[0])
    a = ck.unet_config(UNET_AUDIOLDM2)
    assert a["class_embed_dim"] == 0 and a["attn_cross_dims"] == [0, 96, 160]
    assert ck.unet_config(dict(UNET_MUSICLDM, num_attention_heads=8, attention_head_dim=[5, 10, 20, 20]))["attention_heads"] == 8
    v = ck.vae_config(VAE)
    assert v["block_out_channels"] == [64, 128, 256] and v["latent_channels"] == 8 and abs(v["scaling_factor"] - 0.9227914214134216) < 1e-12
    h = ck.vocoder_config(VOCODER)
    assert h["upsample_initial_channel"] == 256 and h["upsample_rates"] == [5, 4, 2, 2, 2] and h["model_in_dim"] == 64


@pytest.mark.parametrize("fn,base,patch,needle", [
    ("unet_config", UNET_MUSICLDM, dict(class_embed_type="timestep"), "class_embed_type"),
    ("unet_config", UNET_MUSICLDM, dict(class_embeddings_concat=False), "class_embeddings_concat"),
    ("unet_config", UNET_MUSICLDM, dict(use_linear_projection=True), "use_linear_projection"),
    ("unet_config", UNET_MUSICLDM, dict(norm_eps=1e-6), "norm_eps"),
    ("unet_config", UNET_MUSICLDM, dict(resnet_time_scale_shift="scale_shift"), "resnet_time_scale_shift"),
    ("unet_config", UNET_MUSICLDM, dict(cross_attention_dim=768), "cross_attention_dim"),
    ("unet_config", UNET_MUSICLDM, dict(attention_head_dim=[4, 4, 8, 8]), "attention_head_dim"),
    ("unet_config", UNET_MUSICLDM, dict(down_block_types=["DownBlock2D"] * 3 + ["AttnDownBlock2D"]), "down_block_types"),
    ("unet_config", UNET_MUSICLDM, dict(addition_embed_type="text"), "addition_embed_type"),
    ("unet_config", UNET_MUSICLDM, dict(some_future_switch=True), "some_future_switch"),
    ("unet_config", UNET_AUDIOLDM2, dict(cross_attention_dim=[[None, 96, 160]] * 3 + [[None, 96, 200]]), "cross_attention_dim"),
    ("vae_config", VAE, dict(up_block_types=["UpDecoderBlock2D", "AttnUpDecoderBlock2D", "UpDecoderBlock2D"]), "up_block_types"),
    ("vae_config", VAE, dict(mid_block_add_attention=False), "mid_block_add_attention"),
    ("vae_config", VAE, dict(norm_num_groups=48), "norm_num_groups"),
    ("vocoder_config", VOCODER, dict(normalize_before=True), "normalize_before"),
    ("vocoder_config", VOCODER, dict(resblock_kernel_sizes=[3, 8, 11]), "resblock_kernel_sizes"),
    ("vocoder_config", VOCODER, dict(upsample_rates=[5, 4, 2, 2]), "upsample_rates"),
])
def test_what_the_executors_do_not_implement_is_refused_by_key(fn, base, patch, needle):
    from diffmusic_amd import checkpoint as ck
    with pytest.raises(ck.ConfigError) as ei:
        getattr(ck, fn)(dict(base, **patch))
    assert needle in str(ei.value)


def test_every_problem_of_a_file_is_reported_at_once():
    from diffmusic_amd import checkpoint as ck
    with pytest.raises(ck.ConfigError) as ei:
        ck.unet_config(dict(UNET_MUSICLDM, act_fn="gelu", freq_shift=1, upcast_attention=True))
    msg = str(ei.value)
    assert "act_fn" in msg and "freq_shift" in msg and "upcast_attention" in msg


def _write_repo(tmp_path, unet=UNET_MUSICLDM):
    for sub, cfg in (("unet", unet), ("vae", VAE), ("vocoder", VOCODER)):
        os.makedirs(tmp_path / sub, exist_ok=True)
        with open(tmp_path / sub / "config.json", "w") as fh:
            json.dump(cfg, fh)
    return str(tmp_path)


def test_directory_with_non_default_widths_configures_the_executors(tmp_path):
    """A checkpoint directory in upstream layout whose widths are NOT the benchmark defaults: the executors built from its config files
    expect exactly that architecture's tensors (upstream names, the non-default shapes) -- and the benchmark defaults would not fit."""
    from diffmusic_amd import checkpoint as ck
    from diffmusic_amd.engine import HifiGanEngine, UNetEngine, VaeDecoderEngine
    from diffmusic_amd.weights import check_manifest
    repo = _write_repo(tmp_path)
    cfgs = ck.read_configs(repo)
    unet, vae, voc = UNetEngine(cfgs["unet"], device="cpu"), VaeDecoderEngine(cfgs["vae"], device="cpu"), HifiGanEngine(cfgs["vocoder"], device="cpu")
    su, sv, sh = dict(unet.param_specs()), dict(vae.param_specs()), dict(voc.param_specs())
    assert su["conv_in.weight"] == (64, 8, 3, 3) and su["class_embedding.weight"] == (256, 512)
    assert su["time_embedding.linear_1.weight"] == (256, 64)
    assert su["down_blocks.0.resnets.0.time_emb_proj.weight"] == (64, 512)            # [time 256 | class 256] concatenated
    assert su["mid_block.attentions.0.transformer_blocks.0.attn1.to_q.weight"] == (320, 320)
    assert su["up_blocks.0.resnets.0.conv1.weight"] == (320, 640, 3, 3)
    assert sv["decoder.conv_in.weight"] == (256, 8, 3, 3) and sv["decoder.conv_out.weight"] == (1, 64, 3, 3)
    assert sh["conv_pre.weight"] == (256, 64, 7) and sh["upsampler.0.weight"] == (256, 128, 16)
    # a checkpoint of this architecture passes the manifest check; the benchmark-default executor refuses it tensor by tensor
    sd = unet.synth_state_dict(seed=0)
    check_manifest(unet.param_specs(), sd, "unet")
    with pytest.raises(ValueError, match="of another shape"):
        check_manifest(UNetEngine(device="cpu").param_specs(), sd, "unet (benchmark defaults)")


def test_missing_config_file_and_wrong_pipeline_are_errors(tmp_path):
    from diffmusic_amd import checkpoint as ck
    from diffmusic_amd.pipelines import get_pipeline
    repo = _write_repo(tmp_path, unet=UNET_AUDIOLDM2)
    assert ck.read_configs(repo)["unet"]["attn_cross_dims"] == [0, 96, 160]
    with pytest.raises(ck.ConfigError, match="cross-attention context"):
        get_pipeline("musicldm").from_pretrained(repo)                      # an AudioLDM2 U-Net under the MusicLDM pipeline
    os.remove(os.path.join(repo, "vae", "config.json"))
    with pytest.raises(ck.ConfigError, match="vae/config.json"):
        ck.read_configs(repo)
