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
