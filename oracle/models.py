"""fp32 eager PyTorch restatements of the three third-party networks on the hot path, with the
public parameter naming of their upstream implementations so real checkpoints map 1:1:

* HifiGan      == transformers SpeechT5HifiGan (modeling_speecht5.py; reference call site
                  diffmusic/inverse_problem/operator.py:126-130)
* VaeDecoder   == diffusers 0.31.0 AutoencoderKL.decode (reference call site
                  diffmusic/schedulers/scheduling_dps.py:195-197)
* UNetMusicLDM == diffusers 0.31.0 UNet2DConditionModel as configured for MusicLDM (reference
                  call site diffmusic/pipelines/pipeline_musicldm.py:696-703)

diffusers is absent here: VaeDecoder/UNet follow SURVEY.md section 8c Appendix A/B and are parity-unpinned
against diffusers; HifiGan is cross-checked against the installed transformers class in
tests/test_oracle_models.py.  All hyper-parameters are constructor-config driven.
"""
import math
from types import SimpleNamespace
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- HiFi-GAN
class HifiGanResBlock(nn.Module):
    def __init__(self, ch, k, dil, slope):
        super().__init__()
        self.slope = slope
        self.convs1 = nn.ModuleList([nn.Conv1d(ch, ch, k, 1, dilation=d, padding=(k * d - d) // 2) for d in dil])
        self.convs2 = nn.ModuleList([nn.Conv1d(ch, ch, k, 1, dilation=1, padding=(k - 1) // 2) for _ in dil])

    def forward(self, x):
        for c1, c2 in zip(self.convs1, self.convs2):
            r = x
            x = c1(F.leaky_relu(x, self.slope))
            x = c2(F.leaky_relu(x, self.slope))
            x = x + r
        return x


class HifiGan(nn.Module):
    def __init__(self, model_in_dim=64, upsample_initial_channel=1024, upsample_rates=(5, 4, 2, 2, 2),
                 upsample_kernel_sizes=(16, 16, 8, 4, 4), resblock_kernel_sizes=(3, 7, 11),
                 resblock_dilation_sizes=((1, 3, 5),) * 3, leaky_relu_slope=0.1, sampling_rate=16000,
                 normalize_before=False):
        super().__init__()
        self.config = SimpleNamespace(model_in_dim=model_in_dim, upsample_rates=list(upsample_rates),
                                      sampling_rate=sampling_rate, leaky_relu_slope=leaky_relu_slope,
                                      upsample_initial_channel=upsample_initial_channel,
                                      upsample_kernel_sizes=list(upsample_kernel_sizes),
                                      resblock_kernel_sizes=list(resblock_kernel_sizes),
                                      resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes],
                                      normalize_before=normalize_before)
        c0 = upsample_initial_channel
        self.num_kernels = len(resblock_kernel_sizes)
        self.conv_pre = nn.Conv1d(model_in_dim, c0, 7, 1, padding=3)
        self.upsampler = nn.ModuleList([
            nn.ConvTranspose1d(c0 // 2 ** i, c0 // 2 ** (i + 1), k, s, padding=(k - s) // 2)
            for i, (s, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes))])
        self.resblocks = nn.ModuleList()
        for i in range(len(upsample_rates)):
            ch = c0 // 2 ** (i + 1)
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(HifiGanResBlock(ch, k, d, leaky_relu_slope))
        self.conv_post = nn.Conv1d(ch, 1, 7, 1, padding=3)
        self.register_buffer("mean", torch.zeros(model_in_dim))
        self.register_buffer("scale", torch.ones(model_in_dim))

    def forward(self, spectrogram):                  # (B, T, model_in_dim) -> (B, T*prod(rates)+..)
        if self.config.normalize_before:
            spectrogram = (spectrogram - self.mean) / self.scale
        h = self.conv_pre(spectrogram.transpose(2, 1))
        for i, up in enumerate(self.upsampler):
            h = up(F.leaky_relu(h, self.config.leaky_relu_slope))
            rs = self.resblocks[i * self.num_kernels](h)
            for j in range(1, self.num_kernels):
                rs = rs + self.resblocks[i * self.num_kernels + j](h)
            h = rs / self.num_kernels
        h = F.leaky_relu(h)                           # default slope 0.01 (modeling_speecht5.py forward)
        return torch.tanh(self.conv_post(h)).squeeze(1)


# ----------------------------------------------------------------------------- shared 2-D blocks
class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_ch=None, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout) if temb_ch else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Attention(nn.Module):
    """diffusers Attention (AttnProcessor): softmax(q k^T / sqrt(d)) v, to_out[0] with bias."""

    def __init__(self, query_dim, heads, dim_head, cross_dim=None, bias=False, norm_groups=None,
                 eps=1e-5, residual=False):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.residual = heads, residual
        self.group_norm = nn.GroupNorm(norm_groups, query_dim, eps=eps) if norm_groups else None
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross_dim or query_dim, inner, bias=bias)
        self.to_v = nn.Linear(cross_dim or query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Identity()])

    def forward(self, x, ctx=None, mask=None):
        res = x
        shp = None
        if x.dim() == 4:                              # VAE mid-block path (B,C,H,W)
            shp = x.shape
            x = x.view(shp[0], shp[1], -1).transpose(1, 2)
        if self.group_norm is not None:
            x = self.group_norm(x.transpose(1, 2)).transpose(1, 2)
        ctx = x if ctx is None else ctx
        B, N, _ = x.shape
        q, k, v = self.to_q(x), self.to_k(ctx), self.to_v(ctx)
        h = self.heads
        q, k, v = [t.view(B, -1, h, t.shape[-1] // h).transpose(1, 2) for t in (q, k, v)]
        s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
        if mask is not None:
            s = s + mask
        o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, -1)
        o = self.to_out[0](o)
        if shp is not None:
            o = o.transpose(1, 2).reshape(shp)
        return o + res if self.residual else o


# ----------------------------------------------------------------------------- VAE decoder
class _MidBlock(nn.Module):
    def __init__(self, ch, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, groups, eps), ResnetBlock2D(ch, ch, None, groups, eps)])
        self.attentions = nn.ModuleList([Attention(ch, 1, ch, bias=True, norm_groups=groups, eps=eps, residual=True)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, n, add_up, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, eps) for i in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class _Decoder(nn.Module):
    def __init__(self, latent, out_ch, boc, lpb, groups, eps):
        super().__init__()
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = _MidBlock(rev[0], groups, eps)
        self.up_blocks = nn.ModuleList()
        prev = rev[0]
        for i, c in enumerate(rev):
            self.up_blocks.append(_UpDecoderBlock(prev, c, lpb + 1, i != len(rev) - 1, groups, eps))
            prev = c
        self.conv_norm_out = nn.GroupNorm(groups, rev[-1], eps=eps)
        self.conv_out = nn.Conv2d(rev[-1], out_ch, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class VaeDecoder(nn.Module):
    """AutoencoderKL.decode: post_quant_conv -> Decoder.  `.decode(z).sample` like diffusers."""

    def __init__(self, latent_channels=8, out_channels=1, block_out_channels=(128, 256, 512),
                 layers_per_block=2, norm_num_groups=32, scaling_factor=0.9227914214134216, eps=1e-6):
        super().__init__()
        self.config = SimpleNamespace(scaling_factor=scaling_factor, latent_channels=latent_channels,
                                      block_out_channels=list(block_out_channels), out_channels=out_channels,
                                      layers_per_block=layers_per_block, norm_num_groups=norm_num_groups)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = _Decoder(latent_channels, out_channels, block_out_channels, layers_per_block,
                                norm_num_groups, eps)

    def decode(self, z):
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))


# ----------------------------------------------------------------------------- U-Net
class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim=cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx=None, mask=None):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx, mask)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    def __init__(self, ch, heads, dim_head, cross_dim, groups):
        super().__init__()
        self.external_ctx = cross_dim is not None and cross_dim > 0
        if not cross_dim or cross_dim <= 0:
            cross_dim = ch
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, dim_head, cross_dim)])
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x, ctx=None, mask=None):
        B, C, H, W = x.shape
        r = x
        h = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, ctx, mask)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return self.proj_out(h) + r


def _run_attn(attentions, i, napl, x, ctxs):
    """AudioLDM2-style: napl transformers per resnet layer; a transformer whose cross dim is an external width consumes the
    next (context, additive_mask) pair of `ctxs`; the others are (double) self-attention (context None)."""
    k = 0
    for q in range(napl):
        t = attentions[i * napl + q]
        if t.external_ctx:
            x = t(x, ctxs[k][0], ctxs[k][1])
            k += 1
        else:
            x = t(x, None)
    return x


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, n, temb, groups, heads, cross_dims, add_down, attn):
        super().__init__()
        self.napl = len(cross_dims)
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups) for i in range(n)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cout // heads, cd, groups)
                                         for _ in range(n) for cd in cross_dims]) if attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, ctxs=None):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = _run_attn(self.attentions, i, self.napl, x, ctxs)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, n, temb, groups, heads, cross_dims, add_up, attn):
        super().__init__()
        self.napl = len(cross_dims)
        self.resnets = nn.ModuleList()
        for i in range(n):
            skip = cin if i == n - 1 else cout
            rin = prev if i == 0 else cout
            self.resnets.append(ResnetBlock2D(rin + skip, cout, temb, groups))
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cout // heads, cd, groups)
                                         for _ in range(n) for cd in cross_dims]) if attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips, temb, ctxs=None, up_size=None):
        for i, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], dim=1), temb)
            if self.attentions is not None:
                x = _run_attn(self.attentions, i, self.napl, x, ctxs)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, up_size)
        return x


class _UNetMid(nn.Module):
    def __init__(self, ch, temb, groups, heads, cross_dims):
        super().__init__()
        self.napl = len(cross_dims)
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, groups), ResnetBlock2D(ch, ch, temb, groups)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, ch // heads, cd, groups) for cd in cross_dims])

    def forward(self, x, temb, ctxs=None):
        return self.resnets[1](_run_attn(self.attentions, 0, self.napl, self.resnets[0](x, temb), ctxs), temb)


def timestep_embedding(t, dim=128, flip_sin_to_cos=True, shift=0.0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - shift)
    e = t[:, None].float() * torch.exp(exponent)[None]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=-1)
    if flip_sin_to_cos:
        e = torch.cat([e[:, half:], e[:, :half]], dim=-1)
    return e


class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class UNetMusicLDM(nn.Module):
    # also serves as the AudioLDM2 U-Net: attn_cross_dims=(None, 768, 1024), class_embed_dim=0
    """UNet2DConditionModel with class_embed_type='simple_projection', class_embeddings_concat,
    encoder_hidden_states=None (so attn2 is self-attention), conv proj_in/out."""

    def __init__(self, in_channels=8, out_channels=8, block_out_channels=(128, 256, 384, 640),
                 layers_per_block=2, attention_heads=8, norm_num_groups=32,
                 down_attn=(False, True, True, True), up_attn=(True, True, True, False),
                 class_embed_dim=512, attn_cross_dims=(None,)):
        super().__init__()
        boc = list(block_out_channels)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=boc, layers_per_block=layers_per_block,
                                      attention_heads=attention_heads, norm_num_groups=norm_num_groups,
                                      down_attn=list(down_attn), up_attn=list(up_attn),
                                      class_embed_dim=class_embed_dim, sample_size=128)
        tdim = boc[0] * 4
        self.time_embedding = _TimestepEmbedding(boc[0], tdim)
        self.class_embedding = nn.Linear(class_embed_dim, tdim) if class_embed_dim else None
        temb = tdim * 2 if class_embed_dim else tdim
        g, hd = norm_num_groups, attention_heads
        acd = [c if c and c > 0 else None for c in attn_cross_dims]
        cd = [acd] * len(boc)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, c in enumerate(boc):
            cin, out = out, c
            self.down_blocks.append(_DownBlock(cin, out, layers_per_block, temb, g, hd, cd[i],
                                               i != len(boc) - 1, down_attn[i]))
        self.mid_block = _UNetMid(boc[-1], temb, g, hd, cd[-1])
        rev, rcd = list(reversed(boc)), list(reversed(cd))
        self.up_blocks = nn.ModuleList()
        out = rev[0]
        for i, c in enumerate(rev):
            prev, out = out, c
            cin = rev[min(i + 1, len(rev) - 1)]
            self.up_blocks.append(_UpBlock(cin, out, prev, layers_per_block + 1, temb, g, hd, rcd[i],
                                           i != len(rev) - 1, up_attn[i]))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states=None, class_labels=None, encoder_hidden_states_1=None,
                encoder_attention_mask_1=None, **kw):
        """MusicLDM: class_labels only.  AudioLDM2 (plpeline_audioldm2.py:1147-1154): encoder_hidden_states = GPT-2 states
        (B,8,768), encoder_hidden_states_1 = T5 states (B,L,1024) with encoder_attention_mask_1 (B,L)."""
        B = sample.shape[0]
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(B)
        emb = self.time_embedding(timestep_embedding(t, self.config.block_out_channels[0]).to(sample.dtype))
        if self.class_embedding is not None:
            emb = torch.cat([emb, self.class_embedding(class_labels.to(sample.dtype))], dim=-1)
        ctxs = []
        if encoder_hidden_states is not None:
            ctxs.append((encoder_hidden_states, None))
        if encoder_hidden_states_1 is not None:
            bias = None
            if encoder_attention_mask_1 is not None:
                bias = ((1 - encoder_attention_mask_1.to(sample.dtype)) * -10000.0)[:, None, None, :]
            ctxs.append((encoder_hidden_states_1, bias))
        encoder_hidden_states = ctxs
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states)
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            mine, skips = skips[-n:], skips[:-n]
            up_size = skips[-1].shape[2:] if blk.upsamplers is not None else None
            x = blk(x, mine, emb, encoder_hidden_states, up_size)
        return (self.conv_out(F.silu(self.conv_norm_out(x))),)


# ----------------------------------------------------------------------------- init
def kaiming_init_(module, seed=0, gain=1.0):
    """Variance-preserving fan-in init for every conv/linear (SURVEY.md section 8d 'weights'): keeps
    activations O(1) through the ~60-layer VAE+vocoder chain so parity checks are not vacuous."""
    g = torch.Generator().manual_seed(seed)
    for name, p in sorted(module.named_parameters()):
        with torch.no_grad():
            if p.dim() >= 2:
                if "upsampler" in name and p.dim() == 3:       # ConvTranspose1d (Cin, Cout, k): fan_in = Cin*k/stride
                    fan_in = p.shape[0] * p.shape[2]
                else:
                    fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (gain / math.sqrt(fan_in)))
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            else:                                               # norm weights
                p.copy_(1.0 + 0.05 * torch.randn(p.shape, generator=g))
    return module
