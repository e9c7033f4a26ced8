"""-m gpu: MusicLDM-style U-Net forward (HIP) against the fp32 oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
SMALL = dict(in_channels=8, out_channels=8, block_out_channels=[32, 64, 96, 160], layers_per_block=2,
             attention_heads=4, norm_num_groups=32, down_attn=[0, 1, 1, 1], up_attn=[1, 1, 1, 0], class_embed_dim=512)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("B,h,w,t", [(2, 26, 16, 981.0), (1, 30, 16, 1.0), (2, 24, 16, 501.0)])     # (h = 24: 3 x 2 = 6 tokens at the deepest level)
def test_unet_fwd_vs_oracle(B, h, w, t):
    from diffmusic_amd.engine import UNetEngine
    from oracle.models import UNetMusicLDM
    eng = UNetEngine(SMALL)
    sd = eng.synth_state_dict(seed=9)
    eng.load_state_dict(sd)
    ref = UNetMusicLDM(**SMALL)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, 8, h, w, generator=g)
    cls = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    out = eng.forward(x.cuda(), torch.full((B,), t), cls.cuda())
    torch.cuda.synchronize()
    with torch.no_grad():
        oref = ref(x, t, class_labels=cls)[0]
    print("rel eps", _rel(out.cpu(), oref), float(oref.std()))
    assert oref.std() > 0.05
    assert _rel(out.cpu(), oref) < 1e-2


@pytest.mark.parametrize("cfg", [
    dict(SMALL, down_attn=[1, 1, 1, 0], up_attn=[0, 1, 1, 1]),                          # diffusers' default block order: attention at the first levels
    dict(SMALL, block_out_channels=[32, 64, 128], down_attn=[1, 1, 0], up_attn=[0, 1, 1], layers_per_block=1, attention_heads=2),
    dict(SMALL, down_attn=[0, 0, 0, 0], up_attn=[0, 0, 0, 0], class_embed_dim=0),       # no attention blocks, no class embedding (mid block only)
])
def test_unet_layouts_a_checkpoint_config_can_ask_for(cfg):
    """`from_pretrained(<dir>)` configures the U-Net from the checkpoint's `down_block_types` / `up_block_types` / `layers_per_block` /
    head count / `class_embed_type` (diffmusic_amd/checkpoint.py): layouts other than the benchmark's against the oracle."""
    from diffmusic_amd.engine import UNetEngine
    from oracle.models import UNetMusicLDM
    eng = UNetEngine(cfg)
    sd = eng.synth_state_dict(seed=13)
    eng.load_state_dict(sd)
    ref = UNetMusicLDM(**cfg)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(5)
    B = 2
    x = torch.randn(B, 8, 32, 16, generator=g)               # (every attention level keeps a multiple of 4 tokens: the kernels' key granularity)
    cls = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1) if cfg["class_embed_dim"] else None
    out = eng.forward(x.cuda(), torch.full((B,), 301.0), cls.cuda() if cls is not None else None)
    torch.cuda.synchronize()
    with torch.no_grad():
        oref = ref(x, 301, class_labels=cls)[0]
    print("rel eps", _rel(out.cpu(), oref))
    assert oref.std() > 0.05 and _rel(out.cpu(), oref) < 1e-2


A2 = dict(SMALL, class_embed_dim=0, attn_cross_dims=[0, 48, 64])


def test_audioldm2_unet_fwd_vs_oracle():
    """Three transformers per layer: self, cross (8 GPT-2 tokens), cross with key mask (T5 tokens, L=10 -> padded to 12)."""
    from diffmusic_amd.engine import UNetEngine
    from oracle.models import UNetMusicLDM
    eng = UNetEngine(A2)
    sd = eng.synth_state_dict(seed=4)
    eng.load_state_dict(sd)
    ref = UNetMusicLDM(**{k: v for k, v in A2.items() if k != "attn_cross_dims"}, attn_cross_dims=(None, 48, 64))
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(8)
    B = 2
    x = torch.randn(B, 8, 26, 16, generator=g)
    c0 = torch.randn(B, 8, 48, generator=g)
    c1 = torch.randn(B, 10, 64, generator=g)
    mask = torch.ones(B, 10)
    mask[1, 7:] = 0
    out = eng.forward(x.cuda(), torch.full((B,), 501.0), None, c0.cuda(), c1.cuda(), mask.cuda())
    torch.cuda.synchronize()
    with torch.no_grad():
        oref = ref(x, 501, encoder_hidden_states=c0, encoder_hidden_states_1=c1, encoder_attention_mask_1=mask)[0]
    print("rel eps (audioldm2)", _rel(out.cpu(), oref))
    assert _rel(out.cpu(), oref) < 1e-2


@pytest.mark.parametrize("cfg_name", ["musicldm", "audioldm2"])
def test_layernorm_fold_matches_oracle_and_unfused_path(cfg_name, monkeypatch):
    """EPI_LNFOLD (LayerNorm folded into the QKV / Q / FF1 projections, csrc/gemm_tile.h): with strongly non-trivial LayerNorm weights
    and biases the folded U-Net stays at the oracle's level and agrees with the same engine built with the fold switched off
    (separate layernorm launches) to 16-bit rounding."""
    from diffmusic_amd.engine import UNetEngine
    from oracle.models import UNetMusicLDM
    cfg = SMALL if cfg_name == "musicldm" else A2
    eng = UNetEngine(cfg)
    sd = eng.synth_state_dict(seed=13)
    g = torch.Generator().manual_seed(77)
    n_ln = 0
    for k in sd:
        if ".transformer_blocks.0.norm" in k:
            n_ln += 1
            sd[k] = (0.5 + torch.rand(sd[k].shape, generator=g)) if k.endswith("weight") else 0.3 * torch.randn(sd[k].shape, generator=g)
    assert n_ln >= 6
    eng.load_state_dict(sd)
    monkeypatch.setenv("DMX_NO_LN_FOLD", "1")
    eng_plain = UNetEngine(cfg)
    eng_plain.load_state_dict(sd)
    monkeypatch.delenv("DMX_NO_LN_FOLD")
    B = 2
    x = torch.randn(B, 8, 26, 16, generator=g)
    if cfg_name == "musicldm":
        cls = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
        args = (x.cuda(), torch.full((B,), 501.0), cls.cuda())
        ref = UNetMusicLDM(**SMALL)
        okw = dict(class_labels=cls)
    else:
        c0, c1 = torch.randn(B, 8, 48, generator=g), torch.randn(B, 12, 64, generator=g)
        args = (x.cuda(), torch.full((B,), 501.0), None, c0.cuda(), c1.cuda(), torch.ones(B, 12).cuda())
        ref = UNetMusicLDM(**{k: v for k, v in A2.items() if k != "attn_cross_dims"}, attn_cross_dims=(None, 48, 64))
        okw = dict(encoder_hidden_states=c0, encoder_hidden_states_1=c1, encoder_attention_mask_1=torch.ones(B, 12))
    ref.load_state_dict(sd, strict=True)
    out, out_plain = eng.forward(*args), eng_plain.forward(*args)
    torch.cuda.synchronize()
    with torch.no_grad():
        oref = ref(x, 501, **okw)[0]
    print(f"LN fold ({cfg_name}): folded vs oracle {_rel(out.cpu(), oref):.2e}, unfused vs oracle {_rel(out_plain.cpu(), oref):.2e}, "
          f"folded vs unfused {_rel(out, out_plain):.2e}")
    assert _rel(out.cpu(), oref) < 1e-2 and _rel(out_plain.cpu(), oref) < 1e-2
    assert _rel(out, out_plain) < 5e-3
    assert not torch.equal(out, out_plain)          # the two engines really take different paths


@pytest.mark.parametrize("cfg_name", ["musicldm", "audioldm2"])
def test_unet_groupnorm_from_producer_partial_sums(cfg_name, monkeypatch):
    """Levels above 512 pixels (1024 at level 0 here): GroupNorm statistics from the producers' epilogues (EPI_GNSTATS), including the
    skip concatenations of the up path (two sources, group boundaries across the seam) and the folded x2 upsampler (four parity
    regions).  Equal to the classic path to 16-bit rounding, both at the oracle's level."""
    from diffmusic_amd.engine import UNetEngine
    from oracle.models import UNetMusicLDM
    # (partial sums travel per 4-channel quad: groups of 4 / 8 channels here, like the production widths 128 ... 640)
    wide = dict(block_out_channels=[128, 128, 256, 256])
    cfg = dict(SMALL, **wide) if cfg_name == "musicldm" else dict(A2, **wide)
    eng = UNetEngine(cfg)
    sd = eng.synth_state_dict(seed=17)
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    B, h, w = 2, 64, 16
    x = torch.randn(B, 8, h, w, generator=g)
    if cfg_name == "musicldm":
        cls = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
        args = (x.cuda(), torch.full((B,), 501.0), cls.cuda())
        ref = UNetMusicLDM(**cfg)
        okw = dict(class_labels=cls)
    else:
        c0, c1 = torch.randn(B, 8, 48, generator=g), torch.randn(B, 12, 64, generator=g)
        args = (x.cuda(), torch.full((B,), 501.0), None, c0.cuda(), c1.cuda(), torch.ones(B, 12).cuda())
        ref = UNetMusicLDM(**{k: v for k, v in cfg.items() if k != "attn_cross_dims"}, attn_cross_dims=(None, 48, 64))
        okw = dict(encoder_hidden_states=c0, encoder_hidden_states_1=c1, encoder_attention_mask_1=torch.ones(B, 12))
    ref.load_state_dict(sd, strict=True)
    monkeypatch.setenv("DMX_UNET_GN_PARTS", "1")            # (off by default in the U-Net: CFG rows of equal conditioning)
    out = eng.forward(*args).clone()
    monkeypatch.setenv("DMX_NO_GN_PARTS", "1")
    out_classic = eng.forward(*args).clone()
    monkeypatch.delenv("DMX_NO_GN_PARTS")
    torch.cuda.synchronize()
    with torch.no_grad():
        oref = ref(x, 501, **okw)[0]
    print(f"GN parts ({cfg_name}): vs oracle {_rel(out.cpu(), oref):.2e}, classic vs oracle {_rel(out_classic.cpu(), oref):.2e}, "
          f"parts vs classic {_rel(out, out_classic):.2e}")
    assert _rel(out.cpu(), oref) < 1e-2 and _rel(out_classic.cpu(), oref) < 1e-2
    assert _rel(out, out_classic) < 5e-3
    assert not torch.equal(out, out_classic)
