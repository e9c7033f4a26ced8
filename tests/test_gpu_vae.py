"""-m gpu: VAE decoder forward + input-gradient backward (HIP) against the fp32 oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
SMALL = dict(latent_channels=8, out_channels=1, block_out_channels=[32, 64, 64], layers_per_block=2,
             norm_num_groups=32, scaling_factor=0.9227914214134216, eps=1e-6)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("B,h,w", [(1, 10, 4), (2, 12, 4)])
def test_vae_decode_fwd_bwd_vs_oracle(B, h, w):
    from diffmusic_amd.engine import VaeDecoderEngine
    from diffmusic_amd import _lib as L
    from oracle.models import VaeDecoder
    eng = VaeDecoderEngine(SMALL)
    sd = eng.synth_state_dict(seed=5)
    eng.load_state_dict(sd)
    ref = VaeDecoder(**SMALL)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(21)
    z = torch.randn(B, 8, h, w, generator=g)
    dmel = torch.randn(B, 4 * h, 4 * w, generator=g)
    zs = 1.0 / SMALL["scaling_factor"]
    mel, mel32 = eng.decode_hip(z.cuda(), z_scale=zs, want_f32=True)
    dz = eng.backward(dmel.to(L.act_dtype()).cuda(), z_scale=zs)
    torch.cuda.synchronize()
    zr = z.clone().requires_grad_(True)
    mref = ref.decode(zs * zr).sample[:, 0]
    (gref,) = torch.autograd.grad((mref * dmel.to(L.act_dtype()).float()).sum(), zr)
    print("rel mel", _rel(mel32.cpu(), mref), "rel grad", _rel(dz.cpu(), gref))
    assert mref.std() > 0.1
    assert _rel(mel32.cpu(), mref) < 1e-2
    assert _rel(mel.cpu(), mref) < 1e-2
    assert _rel(dz.cpu(), gref) < 3e-2


def test_vae_groupnorm_from_producer_partial_sums(monkeypatch):
    """A latent large enough for the two- and three-launch GroupNorm plans (768 and 3072 pixels): with EPI_GNSTATS the convolutions
    that produce a GroupNorm input write its partial sums and the GroupNorm skips its statistics pass (csrc/gemm_epilogue.h,
    gn_parts_kernel).  Same result as the classic path (DMX_NO_GN_PARTS) to 16-bit rounding, both at the oracle's level, forward and
    backward (the tape's mean / rstd / scale / shift come from the combined partial sums)."""
    from diffmusic_amd.engine import VaeDecoderEngine
    from diffmusic_amd import _lib as L
    from oracle.models import VaeDecoder
    cfg = dict(SMALL, block_out_channels=[128, 128, 256])      # (partial sums travel per 4-channel quad: groups of 4 / 8 channels)
    eng = VaeDecoderEngine(cfg)
    sd = eng.synth_state_dict(seed=6)
    eng.load_state_dict(sd)
    ref = VaeDecoder(**cfg)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(23)
    B, h, w = 3, 24, 8
    z = torch.randn(B, 8, h, w, generator=g)
    dmel = torch.randn(B, 4 * h, 4 * w, generator=g).to(L.act_dtype())
    zs = 1.0 / SMALL["scaling_factor"]
    res = {}
    for tag in ("parts", "classic"):
        if tag == "classic":
            monkeypatch.setenv("DMX_NO_GN_PARTS", "1")
        mel, mel32 = eng.decode_hip(z.cuda(), z_scale=zs, want_f32=True)
        dz = eng.backward(dmel.cuda(), z_scale=zs)
        torch.cuda.synchronize()
        res[tag] = (mel32.clone(), dz.clone())
    monkeypatch.delenv("DMX_NO_GN_PARTS")
    zr = z.clone().requires_grad_(True)
    mref = ref.decode(zs * zr).sample[:, 0]
    (gref,) = torch.autograd.grad((mref * dmel.float()).sum(), zr)
    for tag, (m, d) in res.items():
        print(tag, "rel mel", _rel(m.cpu(), mref), "rel grad", _rel(d.cpu(), gref))
        assert _rel(m.cpu(), mref) < 1e-2 and _rel(d.cpu(), gref) < 3e-2
    assert _rel(res["parts"][0], res["classic"][0]) < 3e-3 and _rel(res["parts"][1], res["classic"][1]) < 1e-2
    assert not torch.equal(res["parts"][0], res["classic"][0])          # two different statistics paths really ran


def test_vae_is_invariant_to_batch_position_and_composition():
    """Production-size decoder (default config, 250 x 16 latents: images of 4000 / 16000 / 64000 pixels; 4000 is no multiple of any wave
    tile's rows).  The GroupNorm partial sums are kept per image-aligned wave tile (gemm_glds_kernel), so what a clip gets -- mel and
    input-gradient -- must not depend on where it sits in the batch or on its neighbours: the same clip at positions 0, 3 and 7 of
    batches with different other clips comes back BIT-identical, forward and backward."""
    from diffmusic_amd.engine import VaeDecoderEngine
    from diffmusic_amd import _lib as L
    eng = VaeDecoderEngine()
    eng.load_state_dict(eng.synth_state_dict(seed=1))
    g = torch.Generator().manual_seed(77)
    B, h, w = 8, 250, 16
    z = torch.randn(B, 8, h, w, generator=g).cuda()
    dmel = torch.randn(B, 4 * h, 4 * w, generator=g).to(L.act_dtype()).cuda()
    zs = 1.0 / eng.config.scaling_factor

    def run(order):
        mel = eng.decode_hip(z[order].contiguous(), z_scale=zs).clone()
        dz = eng.backward(dmel[order].contiguous(), z_scale=zs).clone()
        return mel, dz
    base_mel, base_dz = run(list(range(B)))
    for order in ([3, 1, 2, 0, 4, 5, 6, 7], [7, 6, 5, 4, 3, 2, 1, 0], [5, 5, 0, 0, 5, 1, 0, 5]):
        mel, dz = run(order)
        torch.cuda.synchronize()
        for pos, k in enumerate(order):
            assert torch.equal(mel[pos], base_mel[k]), f"mel of clip {k} at position {pos} of {order} differs from position {k}"
            assert torch.equal(dz[pos], base_dz[k]), f"input-gradient of clip {k} at position {pos} of {order} differs"
    assert float(base_mel.float().std()) > 0.05 and bool(torch.isfinite(base_dz).all())


def test_vae_layout_a_checkpoint_config_can_ask_for():
    """Two blocks, one layer per block, 4 latent channels, 16 groups (vae/config.json keys block_out_channels / layers_per_block /
    latent_channels / norm_num_groups): forward and input-gradient against the oracle."""
    from diffmusic_amd.engine import VaeDecoderEngine
    from diffmusic_amd import _lib as L
    from oracle.models import VaeDecoder
    cfg = dict(latent_channels=4, out_channels=1, block_out_channels=[32, 64], layers_per_block=1, norm_num_groups=16, scaling_factor=0.5, eps=1e-6)
    eng = VaeDecoderEngine(cfg)
    sd = eng.synth_state_dict(seed=3)
    eng.load_state_dict(sd)
    ref = VaeDecoder(**cfg)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(2)
    z = torch.randn(2, 4, 12, 8, generator=g)
    dmel = torch.randn(2, 24, 16, generator=g).to(L.act_dtype())
    mel, mel32 = eng.decode_hip(z.cuda(), z_scale=2.0, want_f32=True)
    dz = eng.backward(dmel.cuda(), z_scale=2.0)
    torch.cuda.synchronize()
    zr = z.clone().requires_grad_(True)
    mref = ref.decode(2.0 * zr).sample[:, 0]
    (gref,) = torch.autograd.grad((mref * dmel.float()).sum(), zr)
    print("rel mel", _rel(mel32.cpu(), mref), "rel grad", _rel(dz.cpu(), gref))
    assert mref.std() > 0.05 and _rel(mel32.cpu(), mref) < 1e-2 and _rel(dz.cpu(), gref) < 3e-2
