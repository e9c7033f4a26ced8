"""-m gpu: GroupNorm (+SiLU) forward through the C ABI test hook against torch fp32 GroupNorm on the same fp16-rounded input.
The three launch plans (one workgroup per (group, image) for tiny images; row-coalesced statistics + fused finalize/apply for
the mid-size U-Net levels; statistics / finalize / apply for the big VAE tensors) must agree with the reference and write the
same (mean, rstd) / scale / shift tape.  Semantics: diffusers ResnetBlock2D norm1/norm2 (torch.nn.GroupNorm)."""
import ctypes as C
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,P,Cc,G,silu", [
    (16, 64, 640, 32, 1),        # tiny: single-launch plan
    (16, 252, 384, 32, 1),
    (16, 1000, 256, 32, 1),      # mid: statistics + finalize/apply (U-Net level 1)
    (16, 4000, 128, 32, 1),      # mid, 4 channels per group (U-Net level 0)
    (16, 4000, 384, 32, 0),      # mid, concatenated skip (12 channels per group, 240-thread workgroups)
    (3, 1003, 640, 32, 1),       # mid, ragged chunking, 20 channels per group
    (2, 4000, 512, 32, 1),       # VAE mid block
    (2, 16000, 256, 32, 1),      # big: three-launch plan
    (1, 64000, 128, 32, 0),
])
def test_groupnorm_forward_plans(B, P, Cc, G, silu):
    from diffmusic_amd import _lib as L
    g = torch.Generator().manual_seed(P + Cc)
    x = (torch.randn(B, P, Cc, generator=g) * 1.7 + 0.6).to(L.act_dtype()).cuda()
    gamma = (torch.randn(Cc, generator=g) * 0.3 + 1.0).cuda()
    beta = (torch.randn(Cc, generator=g) * 0.2).cuda()
    y = torch.empty_like(x)
    stats = torch.empty(B, G, 2, device="cuda")
    scale = torch.empty(B, Cc, device="cuda")
    shift = torch.empty(B, Cc, device="cuda")
    partial = torch.empty(L.lib().dmx_groupnorm_scratch_floats(B, Cc, G), device="cuda")
    eps = 1e-5
    L.check(L.lib().dmx_groupnorm_raw(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(gamma.data_ptr()),
                                      C.c_void_p(beta.data_ptr()), C.c_void_p(stats.data_ptr()), C.c_void_p(scale.data_ptr()),
                                      C.c_void_p(shift.data_ptr()), C.c_void_p(partial.data_ptr()), B, P, Cc, G, eps, silu,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "groupnorm")
    torch.cuda.synchronize()
    xf = x.float()
    ref = F.group_norm(xf.transpose(1, 2), G, gamma, beta, eps).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    assert (y.float() - ref).abs().max().item() < 2e-2          # fp16 output rounding of values up to ~8
    assert ((y.float() - ref).norm() / ref.norm()).item() < 1e-3
    xg = xf.reshape(B, P, G, Cc // G)
    mean = xg.mean(dim=(1, 3))
    rstd = (xg.var(dim=(1, 3), unbiased=False) + eps).rsqrt()
    assert torch.allclose(stats[..., 0], mean, atol=2e-5, rtol=1e-5)
    assert torch.allclose(stats[..., 1], rstd, atol=0, rtol=2e-5)
    sc_ref = rstd.repeat_interleave(Cc // G, dim=1) * gamma
    sh_ref = beta - mean.repeat_interleave(Cc // G, dim=1) * sc_ref
    assert torch.allclose(scale, sc_ref, atol=1e-6, rtol=3e-5)
    assert torch.allclose(shift, sh_ref, atol=3e-5, rtol=3e-5)
