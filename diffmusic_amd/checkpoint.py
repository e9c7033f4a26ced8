"""Architecture of the three network executors from a checkpoint directory's own `config.json` files.

The reference builds its pipelines with `from_pretrained(repo)` (run.py:218, configs/model/*.yaml:2): diffusers /
transformers read `unet/config.json`, `vae/config.json` and `vocoder/config.json` of the checkpoint and construct the networks
from them.  `from_pretrained(<dir>)` here does the same: the keys below are translated into the executors' configs
(`dmx_unet_config` / `dmx_vae_config` / `dmx_hifigan_config`, include/diffmusic_hip.h), and every key whose value asks for
something the hand-written executors do not implement is an error that names the key -- a checkpoint of another architecture
must not load into the recalled defaults and produce garbage.  `weights.check_manifest` stays as the
second line of defence (tensor names and shapes of the configured architecture against the checkpoint).

Key names: diffusers 0.31 `UNet2DConditionModel` / `AudioLDM2UNet2DConditionModel` / `AutoencoderKL`, transformers
`SpeechT5HifiGanConfig`."""
import json
import os

_UNSET = object()


class ConfigError(ValueError):
    pass


def _read(path):
    with open(path) as fh:
        cfg = json.load(fh)
    if not isinstance(cfg, dict):
        raise ConfigError(f"{path}: not a JSON object")
    return cfg


class _Checker:
    """Collects every unsupported key of one config file so that the error lists them all at once."""

    def __init__(self, cfg, what):
        self.cfg, self.what, self.errors, self.seen = cfg, what, [], set()

    def get(self, key, default=_UNSET):
        self.seen.add(key)
        if key in self.cfg:
            return self.cfg[key]
        if default is _UNSET:
            self.errors.append(f"missing key `{key}`")
            return None
        return default

    def require(self, key, allowed, default=_UNSET):
        """The key (or its documented default when absent) must be one of `allowed`."""
        v = self.get(key, default)
        ok = any((v == a and type(v) is type(a)) or (isinstance(a, float) and isinstance(v, (int, float)) and not isinstance(v, bool)
                                                     and abs(float(v) - a) <= 1e-12 * max(1.0, abs(a))) for a in allowed)
        if not ok and not (key not in self.cfg and default is _UNSET):
            self.errors.append(f"`{key}` = {v!r}: only {', '.join(repr(a) for a in allowed)} is implemented")
        return v

    def fail(self, msg):
        self.errors.append(msg)

    def done(self, ignorable=()):
        unknown = [k for k in self.cfg if k not in self.seen and not k.startswith("_") and k not in ignorable]
        if unknown:
            self.errors.append("keys this build does not know (cannot tell whether they change the architecture): " + ", ".join(sorted(unknown)))
        if self.errors:
            raise ConfigError(f"{self.what}: the checkpoint's architecture is not one the HIP executors implement:\n  " + "\n  ".join(self.errors))


def _uniform_int(ck, key, v, n):
    """An int, or a per-block list of n equal ints (diffusers accepts both)."""
    if isinstance(v, (list, tuple)):
        if len(v) != n or len(set(v)) != 1:
            ck.fail(f"`{key}` = {v!r}: per-block values must all be equal (one value for the whole network is implemented)")
            return int(v[0]) if v else 0
        v = v[0]
    if not isinstance(v, int) or isinstance(v, bool) or v <= 0:
        ck.fail(f"`{key}` = {v!r}: a positive integer is needed")
        return 0
    return v


def unet_config(cfg, what="unet/config.json"):
    """diffusers UNet2DConditionModel (MusicLDM) or AudioLDM2UNet2DConditionModel config dict -> UNetEngine config."""
    ck = _Checker(cfg, what)
    cls = cfg.get("_class_name", "UNet2DConditionModel")
    if cls not in ("UNet2DConditionModel", "AudioLDM2UNet2DConditionModel"):
        ck.fail(f"`_class_name` = {cls!r}: UNet2DConditionModel or AudioLDM2UNet2DConditionModel expected")
    boc = ck.get("block_out_channels")
    out = dict(in_channels=ck.get("in_channels", 4), out_channels=ck.get("out_channels", 4))
    if not (isinstance(boc, (list, tuple)) and 2 <= len(boc) <= 8 and all(isinstance(c, int) and c > 0 and c % 8 == 0 for c in boc)):
        ck.fail(f"`block_out_channels` = {boc!r}: 2..8 positive multiples of 8")
        boc = [128, 256]
    n = len(boc)
    out["block_out_channels"] = list(boc)
    out["layers_per_block"] = _uniform_int(ck, "layers_per_block", ck.get("layers_per_block", 2), n)
    out["norm_num_groups"] = ck.get("norm_num_groups", 32)
    if not isinstance(out["norm_num_groups"], int) or any(c % out["norm_num_groups"] for c in boc):
        ck.fail(f"`norm_num_groups` = {out['norm_num_groups']!r} does not divide every entry of block_out_channels")
    down = ck.get("down_block_types", ["CrossAttnDownBlock2D"] * (n - 1) + ["DownBlock2D"])
    up = ck.get("up_block_types", ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * (n - 1))
    kinds = {"DownBlock2D": 0, "CrossAttnDownBlock2D": 1, "UpBlock2D": 0, "CrossAttnUpBlock2D": 1}
    for key, types, names in (("down_block_types", down, ("DownBlock2D", "CrossAttnDownBlock2D")),
                              ("up_block_types", up, ("UpBlock2D", "CrossAttnUpBlock2D"))):
        if not isinstance(types, (list, tuple)) or len(types) != n or any(t not in names for t in types):
            ck.fail(f"`{key}` = {types!r}: {n} entries out of {names}")
    out["down_attn"] = [kinds.get(t, 0) for t in down][:n]
    out["up_attn"] = [kinds.get(t, 0) for t in up][:n]
    ck.require("mid_block_type", ("UNetMidBlock2DCrossAttn",), "UNetMidBlock2DCrossAttn")
    # head count: diffusers reads `attention_head_dim` as the NUMBER of heads unless `num_attention_heads` is given (its documented quirk)
    nah = ck.get("num_attention_heads", None)
    ahd = ck.get("attention_head_dim", 8)
    out["attention_heads"] = _uniform_int(ck, "num_attention_heads" if nah is not None else "attention_head_dim", nah if nah is not None else ahd, n)
    attn_widths = [c for i, c in enumerate(boc) if i == n - 1 or out["down_attn"][i] or out["up_attn"][n - 1 - i]]      # (+ the mid block)
    if out["attention_heads"] and any(c % out["attention_heads"] or (c // out["attention_heads"]) % 8 for c in attn_widths):
        ck.fail(f"{out['attention_heads']} heads: every block width with attention must split into heads of a multiple of 8 channels ({attn_widths})")
    # conditioning
    cet = ck.get("class_embed_type", None)
    if cet is None:
        out["class_embed_dim"] = 0
        ck.get("projection_class_embeddings_input_dim", None)
        ck.get("class_embeddings_concat", False)
    elif cet == "simple_projection":
        d = ck.get("projection_class_embeddings_input_dim", None)
        if not isinstance(d, int) or d <= 0:
            ck.fail(f"`projection_class_embeddings_input_dim` = {d!r}: needed with class_embed_type simple_projection")
            d = 0
        out["class_embed_dim"] = d
        ck.require("class_embeddings_concat", (True,), False)
    else:
        ck.fail(f"`class_embed_type` = {cet!r}: only null or \"simple_projection\" is implemented")
        out["class_embed_dim"] = 0
    cad = ck.get("cross_attention_dim", 1280)
    if cls == "AudioLDM2UNet2DConditionModel":
        # one Transformer2DModel per entry and layer; the same tuple for every block (None = self-attention)
        rows = cad if isinstance(cad, (list, tuple)) and cad and isinstance(cad[0], (list, tuple)) else [cad] * n
        rows = [list(r) if isinstance(r, (list, tuple)) else [r] for r in rows]
        if len(rows) != n or any(r != rows[0] for r in rows) or not (1 <= len(rows[0]) <= 4):
            ck.fail(f"`cross_attention_dim` = {cad!r}: one tuple of 1..4 context widths, equal for every block")
            rows = [[None]]
        dims = [int(d) if d else 0 for d in rows[0]]
        if any(d % 8 for d in dims):
            ck.fail(f"`cross_attention_dim` = {cad!r}: context widths must be multiples of 8")
        out["attn_cross_dims"] = dims
    else:
        # MusicLDM / AudioLDM: the pipeline passes encoder_hidden_states=None (pipeline_musicldm.py:699), so attn2 attends the hidden states
        # themselves and its to_k / to_v take the block's own width: cross_attention_dim must say exactly that (or be null)
        per = list(cad) if isinstance(cad, (list, tuple)) else [cad] * n
        if len(per) != n or any(d is not None and d != c for d, c in zip(per, boc)):
            ck.fail(f"`cross_attention_dim` = {cad!r}: with encoder_hidden_states=None every block's value must equal its width {list(boc)} (or be null)")
        out["attn_cross_dims"] = [0]
    # everything else must be the plain variant the executor hard-codes
    ck.require("act_fn", ("silu",), "silu")
    ck.require("norm_eps", (1e-5,), 1e-5)
    ck.require("flip_sin_to_cos", (True,), True)
    ck.require("freq_shift", (0,), 0)
    ck.require("time_embedding_type", ("positional",), "positional")
    ck.require("use_linear_projection", (False,), False)
    ck.require("only_cross_attention", (False,), False)
    ck.require("dual_cross_attention", (False,), False)
    ck.require("upcast_attention", (False,), False)
    ck.require("resnet_time_scale_shift", ("default",), "default")
    ck.require("downsample_padding", (1,), 1)
    ck.require("mid_block_scale_factor", (1, 1.0), 1)
    ck.require("conv_in_kernel", (3,), 3)
    ck.require("conv_out_kernel", (3,), 3)
    ck.require("center_input_sample", (False,), False)
    ck.require("dropout", (0, 0.0), 0.0)
    tl = ck.get("transformer_layers_per_block", 1)
    if _uniform_int(ck, "transformer_layers_per_block", tl, n) not in (0, 1):
        ck.fail(f"`transformer_layers_per_block` = {tl!r}: 1 is implemented")
    for key in ("num_class_embeds", "addition_embed_type", "addition_time_embed_dim", "time_cond_proj_dim", "encoder_hid_dim",
                "encoder_hid_dim_type", "timestep_post_act", "time_embedding_act_fn", "time_embedding_dim", "class_embed_type_act",
                "mid_block_only_cross_attention", "cross_attention_norm", "resnet_skip_time_act", "resnet_out_scale_factor",
                "attention_type", "reverse_transformer_layers_per_block", "addition_embed_type_num_heads", "conv_in_kernel_size"):
        if key in cfg:
            v = ck.get(key)
            if v not in (None, False, "default", 1.0, 1):
                ck.fail(f"`{key}` = {v!r}: not implemented (null / default only)")
    ck.get("sample_size", None)                       # geometry comes from the call (audio_length_in_s), not from the config
    ck.done()
    return out


def vae_config(cfg, what="vae/config.json"):
    """diffusers AutoencoderKL config dict -> VaeDecoderEngine config (decoder side; the hot path never encodes)."""
    ck = _Checker(cfg, what)
    cls = cfg.get("_class_name", "AutoencoderKL")
    if cls != "AutoencoderKL":
        ck.fail(f"`_class_name` = {cls!r}: AutoencoderKL expected")
    boc = ck.get("block_out_channels")
    if not (isinstance(boc, (list, tuple)) and 1 <= len(boc) <= 8 and all(isinstance(c, int) and c > 0 and c % 8 == 0 for c in boc)):
        ck.fail(f"`block_out_channels` = {boc!r}: 1..8 positive multiples of 8")
        boc = [128]
    n = len(boc)
    g = ck.get("norm_num_groups", 32)
    if not isinstance(g, int) or any(c % g for c in boc):
        ck.fail(f"`norm_num_groups` = {g!r} does not divide every entry of block_out_channels")
    up = ck.get("up_block_types", ["UpDecoderBlock2D"] * n)
    if not isinstance(up, (list, tuple)) or len(up) != n or any(t != "UpDecoderBlock2D" for t in up):
        ck.fail(f"`up_block_types` = {up!r}: {n} x UpDecoderBlock2D")
    ck.get("down_block_types", None)                  # encoder side: not built (allow_unexpected tensors of the checkpoint)
    ck.get("in_channels", None)
    out = dict(latent_channels=ck.get("latent_channels", 4), out_channels=ck.get("out_channels", 3), block_out_channels=list(boc),
               layers_per_block=ck.get("layers_per_block", 1), norm_num_groups=g, scaling_factor=float(ck.get("scaling_factor", 0.18215)),
               eps=1e-6)
    ck.require("act_fn", ("silu",), "silu")
    ck.require("mid_block_add_attention", (True,), True)
    ck.require("use_quant_conv", (True,), True)
    ck.require("use_post_quant_conv", (True,), True)
    for key in ("shift_factor", "latents_mean", "latents_std"):
        if cfg.get(key) is not None:
            ck.fail(f"`{key}` = {cfg[key]!r}: not implemented (null only)")
        ck.get(key, None)
    ck.get("sample_size", None)
    ck.get("force_upcast", None)                      # a dtype policy of diffusers' decode(), not architecture
    ck.done()
    return out


def vocoder_config(cfg, what="vocoder/config.json"):
    """transformers SpeechT5HifiGanConfig dict -> HifiGanEngine config."""
    ck = _Checker(cfg, what)
    rates, ks = ck.get("upsample_rates", [4, 4, 4, 4]), ck.get("upsample_kernel_sizes", [8, 8, 8, 8])
    rk, rd = ck.get("resblock_kernel_sizes", [3, 7, 11]), ck.get("resblock_dilation_sizes", [[1, 3, 5]] * 3)
    if not (isinstance(rates, list) and isinstance(ks, list) and len(rates) == len(ks) and 1 <= len(rates) <= 8):
        ck.fail(f"`upsample_rates` {rates!r} / `upsample_kernel_sizes` {ks!r}: two lists of equal length 1..8")
    if not (isinstance(rk, list) and isinstance(rd, list) and len(rk) == len(rd) and 1 <= len(rk) <= 8 and
            all(isinstance(r, list) and len(r) == len(rd[0]) for r in rd)):
        ck.fail(f"`resblock_kernel_sizes` {rk!r} / `resblock_dilation_sizes` {rd!r}: one dilation list (of equal lengths) per kernel size")
    if isinstance(rk, list) and any(isinstance(k, int) and k % 2 == 0 for k in rk):
        ck.fail(f"`resblock_kernel_sizes` = {rk!r}: odd kernel sizes (same-length convolutions)")
    out = dict(model_in_dim=ck.get("model_in_dim", 80), sampling_rate=ck.get("sampling_rate", 16000),
               upsample_initial_channel=ck.get("upsample_initial_channel", 512), upsample_rates=rates, upsample_kernel_sizes=ks,
               resblock_kernel_sizes=rk, resblock_dilation_sizes=rd, leaky_relu_slope=float(ck.get("leaky_relu_slope", 0.1)))
    uic = out["upsample_initial_channel"]
    if not isinstance(uic, int) or uic <= 0 or (isinstance(rates, list) and uic % (2 ** len(rates))):
        ck.fail(f"`upsample_initial_channel` = {uic!r}: must halve {len(rates) if isinstance(rates, list) else '?'} times")
    # the mel normalisation in front of the network ((x - mean) / scale): the reference's checkpoints switch it off
    ck.require("normalize_before", (False,), True)
    ck.get("initializer_range", None)
    for key in ("model_type", "architectures", "torch_dtype", "transformers_version"):
        ck.get(key, None)
    ck.done()
    return out


def read_configs(repo_dir):
    """{"unet": ..., "vae": ..., "vocoder": ...} executor configs of a checkpoint directory (every sub-folder needs its config.json)."""
    out, errs = {}, []
    for sub, fn in (("unet", unet_config), ("vae", vae_config), ("vocoder", vocoder_config)):
        path = os.path.join(repo_dir, sub, "config.json")
        if not os.path.isfile(path):
            errs.append(f"{path}: missing (the architecture is read from the checkpoint, as the reference's from_pretrained does)")
            continue
        try:
            out[sub] = fn(_read(path), what=path)
        except ConfigError as e:
            errs.append(str(e))
    if errs:
        raise ConfigError("\n".join(errs))
    return out
