"""Python handles over the native network executors (HiFi-GAN, VAE decoder, U-Net).

Only plumbing lives here: device buffers come from torch (caching allocator), launches go to the
C ABI on torch's current HIP stream -- through the PyTorch custom ops `torch.ops.diffmusic_hip.*`
(csrc_torch/torch_ops.cpp; default) or through ctypes (DMX_TORCH_OPS=0), the same `extern "C"` launchers either way."""
import ctypes as C
import torch
from . import _lib as L
from . import ops
from .weights import synth_state_dict

HIFIGAN_DEFAULT = dict(model_in_dim=64, upsample_initial_channel=1024, upsample_rates=[5, 4, 2, 2, 2],
                       upsample_kernel_sizes=[16, 16, 8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
                       resblock_dilation_sizes=[[1, 3, 5]] * 3, leaky_relu_slope=0.1, sampling_rate=16000)
VAE_DEFAULT = dict(latent_channels=8, out_channels=1, block_out_channels=[128, 256, 512], layers_per_block=2,
                   norm_num_groups=32, scaling_factor=0.9227914214134216, eps=1e-6)
UNET_MUSICLDM_DEFAULT = dict(in_channels=8, out_channels=8, block_out_channels=[128, 256, 384, 640],
                             layers_per_block=2, attention_heads=8, norm_num_groups=32,
                             down_attn=[0, 1, 1, 1], up_attn=[1, 1, 1, 0], class_embed_dim=512, attn_cross_dims=[0])
# AudioLDM2UNet2DConditionModel: three transformers per layer (self, GPT-2 states 768, T5 states 1024), no class embedding
UNET_AUDIOLDM2_DEFAULT = dict(UNET_MUSICLDM_DEFAULT, class_embed_dim=0, attn_cross_dims=[0, 768, 1024])


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _fill(arr, vals):
    for i, v in enumerate(vals):
        arr[i] = int(v)


class _Engine:
    kind = "generic"

    def __init__(self, handle, config, device):
        if not handle:
            L.check(-1, "model create")
        self._h = C.c_void_p(handle)
        self.cfg = dict(config)
        self.device = torch.device(device)
        self._ws = {}
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().dmx_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def param_specs(self):
        lib = L.lib()
        out = []
        for i in range(lib.dmx_model_num_params(self._h)):
            nd = lib.dmx_model_param_ndim(self._h, i)
            out.append((lib.dmx_model_param_name(self._h, i).decode(),
                        tuple(lib.dmx_model_param_dim(self._h, i, d) for d in range(nd))))
        return out

    def _stride_of(self, name):
        return 1

    def synth_state_dict(self, seed=0):
        return synth_state_dict(self.param_specs(), seed, self.kind, self._stride_of)

    # tensors a real checkpoint of the same model may carry that this executor does not use
    allow_unexpected = ()

    def load_state_dict(self, sd, strict=False):
        """strict=True (checkpoint directories): every tensor of the checkpoint must be one the configured architecture expects
        (besides `allow_unexpected` prefixes); either way all missing / mis-shaped tensors are reported together (weights.check_manifest)."""
        lib = L.lib()
        from .weights import check_manifest
        specs = self.param_specs()
        check_manifest(specs, sd if strict else {n: sd[n] for n, _ in specs if n in sd}, f"{self.kind} ({type(self).__name__})",
                       self.allow_unexpected)
        for name, shape in specs:
            t = sd[name].detach().to(torch.float32).contiguous().cpu()
            L.check(lib.dmx_model_load_param(self._h, name.encode(), C.c_void_p(t.data_ptr()), t.numel()), name)
        with torch.cuda.device(self.device):
            L.check(lib.dmx_model_finalize(self._h, _stream()), "finalize")
            torch.cuda.synchronize()
        self._finalized = True
        return self

    def _workspace(self, key, nbytes):
        """One byte buffer per executor, grown to the largest request so far (calls of different batch sizes alternate when the
        clip lanes of a call are of unequal size: no reallocation per call).  The launches that use it are stream-ordered."""
        ws = self._ws.get("buf")
        if ws is None or ws.numel() < nbytes:
            self._ws.clear()
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self._ws["buf"] = ws
        return ws


class HifiGanEngine(_Engine):
    """`vocoder(mel)` of the reference (operator.py:126-130) + its input-gradient backward."""
    kind = "hifigan"
    allow_unexpected = ("mean", "scale")                   # SpeechT5HifiGan buffers (normalize_before = False in the configs used)

    def __init__(self, config=None, device="cuda"):
        cfg = dict(HIFIGAN_DEFAULT)
        cfg.update(config or {})
        c = L.HifiGanConfig()
        c.model_in_dim, c.upsample_initial_channel = cfg["model_in_dim"], cfg["upsample_initial_channel"]
        c.num_upsamples = len(cfg["upsample_rates"])
        _fill(c.upsample_rates, cfg["upsample_rates"])
        _fill(c.upsample_kernel_sizes, cfg["upsample_kernel_sizes"])
        c.num_kernels = len(cfg["resblock_kernel_sizes"])
        _fill(c.resblock_kernel_sizes, cfg["resblock_kernel_sizes"])
        c.num_dilations = len(cfg["resblock_dilation_sizes"][0])
        _fill(c.resblock_dilation_sizes, [d for row in cfg["resblock_dilation_sizes"] for d in row])
        c.leaky_relu_slope = cfg["leaky_relu_slope"]
        super().__init__(L.lib().dmx_hifigan_create(C.byref(c)), cfg, device)

    def _stride_of(self, name):
        return self.cfg["upsample_rates"][int(name.split(".")[1])]

    @property
    def config(self):
        from types import SimpleNamespace
        return SimpleNamespace(**self.cfg)

    def out_len(self, frames):
        return L.lib().dmx_hifigan_out_len(self._h, frames)

    def __call__(self, mel):
        """`vocoder(mel_spectrogram)` of the reference: (B, T, model_in_dim) any float dtype -> (B, samples) fp32."""
        return self.forward(mel.to(device=self.device, dtype=L.act_dtype()).contiguous())

    def forward(self, mel):
        """mel (B, T, model_in_dim) act-dtype cuda -> wav (B, out_len) fp32."""
        assert mel.dtype == L.act_dtype() and mel.is_cuda and mel.is_contiguous()
        B, T, _ = mel.shape
        lib = L.lib()
        ws = self._workspace(("h", B, T), lib.dmx_hifigan_workspace_bytes(self._h, B, T))
        self._shape = (B, T)
        if ops.enabled():
            return ops.hip.hifigan_fwd(self._h.value, mel, ws)
        wav = torch.empty(B, self.out_len(T), dtype=torch.float32, device=mel.device)
        L.check(lib.dmx_hifigan_fwd(self._h, _ptr(mel), _ptr(wav), B, T, _ptr(ws), ws.numel(), _stream()), "hifigan_fwd")
        return wav

    def backward(self, dwav):
        B, T = self._shape
        assert dwav.dtype == torch.float32 and dwav.is_contiguous()
        if ops.enabled():
            return ops.hip.hifigan_bwd(self._h.value, dwav, T, self.cfg["model_in_dim"])
        dmel = torch.empty(B, T, self.cfg["model_in_dim"], dtype=L.act_dtype(), device=dwav.device)
        L.check(L.lib().dmx_hifigan_bwd(self._h, _ptr(dwav), _ptr(dmel), _stream()), "hifigan_bwd")
        return dmel


class VaeDecoderEngine(_Engine):
    """`vae.decode(z).sample` of the reference (scheduling_dps.py:195-197) + input-gradient backward."""
    kind = "vae"
    allow_unexpected = ("encoder.", "quant_conv.")          # AutoencoderKL checkpoints carry the encoder too; the hot path decodes only

    def __init__(self, config=None, device="cuda"):
        cfg = dict(VAE_DEFAULT)
        cfg.update(config or {})
        c = L.VaeConfig()
        c.latent_channels, c.out_channels = cfg["latent_channels"], cfg["out_channels"]
        c.num_blocks = len(cfg["block_out_channels"])
        _fill(c.block_out_channels, cfg["block_out_channels"])
        c.layers_per_block, c.norm_num_groups, c.eps = cfg["layers_per_block"], cfg["norm_num_groups"], cfg["eps"]
        super().__init__(L.lib().dmx_vae_decoder_create(C.byref(c)), cfg, device)
        self.scale_factor = 2 ** (c.num_blocks - 1)
        self.scaling_factor = cfg["scaling_factor"]

    @property
    def config(self):
        from types import SimpleNamespace
        return SimpleNamespace(**self.cfg)

    def decode(self, z):
        """diffusers-shaped `vae.decode(z).sample`: (B, latent, h, w) -> (B, 1, 4h, 4w) fp32 (no backward state)."""
        from types import SimpleNamespace
        _, m32 = self.decode_hip(z.to(device=self.device, dtype=torch.float32).contiguous(), 1.0, keep_state=False, want_f32=True)
        return SimpleNamespace(sample=m32.unsqueeze(1))

    def decode_hip(self, z, z_scale=1.0, keep_state=True, want_f32=False):
        """z (B, latent, h, w) fp32 NCHW -> mel (B, H, W) act dtype [, fp32 copy]."""
        assert z.dtype == torch.float32 and z.is_cuda and z.is_contiguous()
        B, _, h, w = z.shape
        lib = L.lib()
        ws = self._workspace(("v", B, h, w), lib.dmx_vae_workspace_bytes(self._h, B, h, w))
        s = self.scale_factor
        self._shape = (B, h, w)
        if ops.enabled():
            mel, mel32 = ops.hip.vae_dec_fwd(self._h.value, z, float(z_scale), bool(keep_state), bool(want_f32), s, ws)
            return (mel, mel32) if want_f32 else mel
        mel = torch.empty(B, h * s, w * s, dtype=L.act_dtype(), device=z.device)
        mel32 = torch.empty(B, h * s, w * s, dtype=torch.float32, device=z.device) if want_f32 else None
        L.check(lib.dmx_vae_decode_fwd(self._h, _ptr(z), float(z_scale), _ptr(mel), _ptr(mel32), B, h, w, int(keep_state),
                                       _ptr(ws), ws.numel(), _stream()), "vae_decode_fwd")
        return (mel, mel32) if want_f32 else mel

    def backward(self, dmel, z_scale=1.0):
        B, h, w = self._shape
        assert dmel.dtype == L.act_dtype() and dmel.is_contiguous()
        if ops.enabled():
            return ops.hip.vae_dec_bwd(self._h.value, dmel, float(z_scale), self.cfg["latent_channels"], self.scale_factor)
        dz = torch.empty(B, self.cfg["latent_channels"], h, w, dtype=torch.float32, device=dmel.device)
        L.check(L.lib().dmx_vae_decode_bwd(self._h, _ptr(dmel), float(z_scale), _ptr(dz), _stream()), "vae_decode_bwd")
        return dz


class UNetEngine(_Engine):
    """`self.unet(latent_model_input, t, encoder_hidden_states=None, class_labels=prompt_embeds)[0]`
    of the reference (pipeline_musicldm.py:696-703), forward only."""
    kind = "unet"

    def __init__(self, config=None, device="cuda"):
        cfg = dict(UNET_MUSICLDM_DEFAULT)
        cfg.update(config or {})
        c = L.UNetConfig()
        c.in_channels, c.out_channels = cfg["in_channels"], cfg["out_channels"]
        c.num_blocks = len(cfg["block_out_channels"])
        _fill(c.block_out_channels, cfg["block_out_channels"])
        c.layers_per_block, c.attention_heads = cfg["layers_per_block"], cfg["attention_heads"]
        c.norm_num_groups = cfg["norm_num_groups"]
        _fill(c.down_attn, cfg["down_attn"])
        _fill(c.up_attn, cfg["up_attn"])
        c.class_embed_dim = cfg["class_embed_dim"]
        acd = [int(d or 0) for d in cfg.get("attn_cross_dims", [0])]
        c.num_attn_per_layer = len(acd)
        _fill(c.attn_cross_dims, acd)
        self._n_ctx = sum(1 for d in acd if d > 0)
        super().__init__(L.lib().dmx_unet_create(C.byref(c)), cfg, device)

    def forward(self, x, t, class_labels=None, encoder_hidden_states=None, encoder_hidden_states_1=None,
                encoder_attention_mask_1=None):
        """x (B, C, h, w) fp32, t (B,) fp32, class_labels (B, class_embed_dim) fp32 -> eps fp32 like x.
        AudioLDM2: encoder_hidden_states (B, 8, 768), encoder_hidden_states_1 (B, L, 1024) + mask (B, L)."""
        assert x.dtype == torch.float32 and x.is_cuda and x.is_contiguous()
        B, _, h, w = x.shape
        dev = x.device
        t = t.to(device=dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        if class_labels is not None:
            class_labels = class_labels.to(device=dev, dtype=torch.float32).contiguous()
        lib = L.lib()
        use_ops = ops.enabled() and self.cfg["out_channels"] == self.cfg["in_channels"]
        eps = None if use_ops else torch.empty(B, self.cfg["out_channels"], h, w, dtype=torch.float32, device=dev)
        if self._n_ctx == 0:
            ws = self._workspace(("u", B, h, w), lib.dmx_unet_workspace_bytes(self._h, B, h, w))
            if use_ops:
                return ops.hip.unet_fwd(self._h.value, x, t, class_labels, ws)
            L.check(lib.dmx_unet_fwd(self._h, _ptr(x), _ptr(t), _ptr(class_labels), _ptr(eps), B, h, w, _ptr(ws), ws.numel(),
                                     _stream()), "unet_fwd")
            return eps
        c0 = encoder_hidden_states.to(device=dev, dtype=torch.float32).contiguous()
        c1 = encoder_hidden_states_1.to(device=dev, dtype=torch.float32)
        m1 = encoder_attention_mask_1
        m1 = torch.ones(c1.shape[:2], device=dev) if m1 is None else m1.to(device=dev, dtype=torch.float32)
        pad = (-c1.shape[1]) % 4                     # key count must be a multiple of 4: pad with masked-out tokens
        if pad:
            c1 = torch.nn.functional.pad(c1, (0, 0, 0, pad))
            m1 = torch.nn.functional.pad(m1, (0, pad))
        c1 = c1.contiguous()
        bias1 = ((1.0 - m1) * -10000.0).contiguous()
        n0, n1 = c0.shape[1], c1.shape[1]
        ws = self._workspace(("u", B, h, w, n0, n1), lib.dmx_unet_workspace_bytes_ctx(self._h, B, h, w, n0, n1))
        if use_ops:
            return ops.hip.unet_fwd_ctx(self._h.value, x, t, class_labels, c0, c1, bias1, ws)
        L.check(lib.dmx_unet_fwd_ctx(self._h, _ptr(x), _ptr(t), _ptr(class_labels), _ptr(c0), n0, _ptr(c1), n1, _ptr(bias1), _ptr(eps),
                                     B, h, w, _ptr(ws), ws.numel(), _stream()), "unet_fwd_ctx")
        return eps


HTSAT_DEFAULT = dict(spec_size=256, num_mel_bins=64, patch_size=4, patch_embeds_hidden_size=96, window_size=8, depths=[2, 2, 6, 2],
                     num_attention_heads=[4, 8, 16, 32], layer_norm_eps=1e-5, batch_norm_eps=1e-5)


class HtsatEngine(_Engine):
    """CLAP HTS-AT audio tower (transformers `ClapAudioModel`) as the style-guidance operator uses it
    (`StyleGuidanceOperator.transform`, diffmusic/inverse_problem/operator.py:253-271): log-mel features -> token features, and the
    gradient of a scalar w.r.t. the log-mel features.  `config`: a `ClapAudioConfig` (or a dict of its fields)."""
    kind = "htsat"
    allow_unexpected = ("audio_encoder.batch_norm.num_batches_tracked", "audio_encoder.layers.")      # (relative_position_index buffers)

    def __init__(self, config=None, device="cuda"):
        cfg = dict(HTSAT_DEFAULT)
        src = config if isinstance(config, dict) or config is None else {k: getattr(config, k) for k in HTSAT_DEFAULT if hasattr(config, k)}
        cfg.update(src or {})
        if config is not None and not isinstance(config, dict):
            unsupported = dict(enable_fusion=False, hidden_act="gelu", qkv_bias=True, enable_patch_layer_norm=True, flatten_patch_embeds=True,
                               patch_embed_input_channels=1, mlp_ratio=4.0, patch_stride=[4, 4])
            for k, want in unsupported.items():
                got = getattr(config, k, want)
                got = list(got) if isinstance(got, (tuple, list)) else got
                if got != want:
                    raise ValueError(f"ClapAudioConfig.{k} = {got!r}: the HIP tower implements {want!r} only")
        c = L.HtsatConfig()
        c.spec_size, c.num_mel_bins, c.patch_size = cfg["spec_size"], cfg["num_mel_bins"], cfg["patch_size"]
        c.embed_dim, c.window_size, c.num_stages = cfg["patch_embeds_hidden_size"], cfg["window_size"], len(cfg["depths"])
        _fill(c.depths, cfg["depths"])
        _fill(c.num_heads, cfg["num_attention_heads"])
        c.ln_eps, c.bn_eps = cfg["layer_norm_eps"], cfg.get("batch_norm_eps", 1e-5)
        super().__init__(L.lib().dmx_htsat_create(C.byref(c)), cfg, device)
        t, ch = C.c_int(), C.c_int()
        L.check(L.lib().dmx_htsat_feature_dims(self._h, C.byref(t), C.byref(ch)), "htsat dims")
        self.tokens, self.channels = t.value, ch.value

    def load_state_dict(self, sd, strict=False):
        # buffers of the torch module that carry no parameters of the tower
        sd = {k: v for k, v in sd.items() if not k.endswith("relative_position_index") and not k.endswith("num_batches_tracked")}
        return super().load_state_dict(sd, strict=strict)

    def forward(self, mel, keep_state=True):
        """mel (B, frames, num_mel_bins) fp32 cuda -> token features (B, tokens, channels) fp32 (last_hidden_state, grid order)."""
        assert mel.dtype == torch.float32 and mel.is_cuda and mel.is_contiguous() and mel.dim() == 3
        B, frames, _ = mel.shape
        lib = L.lib()
        nbytes = lib.dmx_htsat_workspace_bytes(self._h, B, frames)
        if nbytes == 0:
            L.check(-1, "htsat workspace")
        ws = self._workspace(("t", B, frames), nbytes)
        self._shape = (B, frames, mel.shape[2])
        self._mel = mel                          # the backward pass re-reads the input (its stage is recomputed, not taped)
        if ops.enabled():
            return ops.hip.htsat_fwd(self._h.value, mel, bool(keep_state), ws)
        feat = torch.empty(B, self.tokens, self.channels, dtype=torch.float32, device=mel.device)
        L.check(lib.dmx_htsat_fwd(self._h, _ptr(mel), B, frames, _ptr(feat), int(keep_state), _ptr(ws), ws.numel(), _stream()), "htsat_fwd")
        return feat

    def backward(self, dfeat, scale=None):
        """dfeat (B, tokens, channels) fp32 -> d mel (B, frames, bins) fp32, times scale[b] when given."""
        B, frames, bins = self._shape
        assert dfeat.dtype == torch.float32 and dfeat.is_contiguous() and dfeat.shape == (B, self.tokens, self.channels)
        if ops.enabled():
            return ops.hip.htsat_bwd(self._h.value, dfeat, scale, frames, bins)
        dmel = torch.empty(B, frames, bins, dtype=torch.float32, device=dfeat.device)
        L.check(L.lib().dmx_htsat_bwd(self._h, _ptr(dfeat), _ptr(scale), _ptr(dmel), _stream()), "htsat_bwd")
        return dmel


def gram(feat):
    """G[b] = F[b]^T F[b] / T for token features F (B, T, C) fp32 cuda -> (B, C, C)."""
    assert feat.dtype == torch.float32 and feat.is_cuda and feat.is_contiguous()
    B, T, Cc = feat.shape
    if ops.enabled():
        return ops.hip.gram_fwd(feat)
    g = torch.empty(B, Cc, Cc, dtype=torch.float32, device=feat.device)
    L.check(L.lib().dmx_gram_fwd(_ptr(feat), _ptr(g), B, T, Cc, _stream()), "gram_fwd")
    return g


def gram_backward(feat, dgram):
    """dF = F (dG + dG^T) / T."""
    assert dgram.dtype == torch.float32 and dgram.is_contiguous() and feat.is_contiguous()
    B, T, Cc = feat.shape
    if ops.enabled():
        return ops.hip.gram_bwd(feat, dgram)
    d = torch.empty_like(feat)
    L.check(L.lib().dmx_gram_bwd(_ptr(feat), _ptr(dgram), _ptr(d), B, T, Cc, _stream()), "gram_bwd")
    return d
