"""-m gpu: MusicLDM-style U-Net forward (HIP) against the fp32 oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
SMALL = dict(in_channels=8, out_channels=8, block_out_channels=[32, 64, 96, 160], layers_per_block=2,
             attention_heads=4, norm_num_groups=32, down_attn=[0, 1, 1, 1], up_attn=[1, 1, 1, 0], class_embed_dim=512)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("B,h,w,t", [(2, 26, 16, 981.0), (1, 30, 16, 1.0)])
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
