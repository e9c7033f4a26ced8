"""-m gpu: HiFi-GAN forward + input-gradient backward (HIP, fp16 MFMA) against the fp32 oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _adt():
    from diffmusic_amd import _lib as L
    return L.act_dtype()

SMALL = dict(model_in_dim=64, upsample_initial_channel=128, upsample_rates=[5, 4, 2, 2, 2],
             upsample_kernel_sizes=[16, 16, 8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
             resblock_dilation_sizes=[[1, 3, 5]] * 3, leaky_relu_slope=0.1)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("B,T,slope,gtol", [(1, 40, 1.0, 6e-2), (3, 57, 1.0, 6e-2), (2, 40, 0.1, 0.12)])
def test_hifigan_fwd_bwd_vs_oracle(B, T, slope, gtol):
    """slope=1.0 makes the net linear (pins every conv/dgrad/epilogue path at bf16 rounding level);
    slope=0.1 is the real net, where bf16 activations flip leaky-relu' masks of near-zero units, so the
    input gradient is compared at a looser tolerance plus a cosine check (DESIGN.md 'Numerics')."""
    from diffmusic_amd.engine import HifiGanEngine
    from oracle.models import HifiGan
    cfg = dict(SMALL, leaky_relu_slope=slope)
    eng = HifiGanEngine(cfg)
    sd = eng.synth_state_dict(seed=3)
    eng.load_state_dict(sd)
    ref = HifiGan(**cfg)
    ref.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(11)
    mel = torch.randn(B, T, 64, generator=g).to(_adt())
    dwav_cpu = torch.randn(B, eng.out_len(T), generator=g)
    wav = eng.forward(mel.cuda())
    dmel = eng.backward(dwav_cpu.cuda())
    torch.cuda.synchronize()
    x = mel.float().requires_grad_(True)
    wref = ref(x)
    assert wav.shape == wref.shape
    (gref,) = torch.autograd.grad((wref * dwav_cpu).sum(), x)
    assert wref.abs().mean() > 0.02                       # not vacuous
    print("rel err wav", _rel(wav.cpu(), wref), "grad", _rel(dmel.cpu(), gref))
    assert _rel(wav.cpu(), wref) < 3e-2, "waveform"
    assert _rel(dmel.cpu(), gref) < gtol, "input gradient"
    cos = torch.nn.functional.cosine_similarity(dmel.cpu().float().flatten(), gref.flatten(), dim=0).item()
    assert cos > 0.97


def test_fp16_torch_oracle_shows_the_same_mask_flip_gap():
    """DESIGN.md section 5 explains the 5e-2 rel-L2 of the vocoder's input gradient by leaky-relu' mask flips of 16-bit activations
    (37 kinks in series).  Evidence instead of explanation: the ORACLE network itself, run by torch in fp16 on the same device
    (`oracle.models.HifiGan.half().cuda()`: checker code, MIOpen / rocBLAS kernels, nothing of the product), deviates from its own fp32
    run by the same amount on the same production-size input -- and the HIP path is not further from fp32 than that fp16 reference
    (x 1.5 + margin).  Removing the resblocks' kinks (slope 1) lowers both gaps together."""
    from diffmusic_amd.engine import HifiGanEngine
    from oracle.models import HifiGan
    out = {}
    for slope in (0.1, 1.0):
        eng = HifiGanEngine(dict(leaky_relu_slope=slope))
        sd = eng.synth_state_dict(seed=2)
        eng.load_state_dict(sd)
        cfg = dict(leaky_relu_slope=slope)
        ref32 = HifiGan(**cfg).eval()
        ref32.load_state_dict(sd, strict=False)
        ref32 = ref32.cuda()
        import copy
        ref16 = copy.deepcopy(ref32).half()
        g = torch.Generator().manual_seed(21)
        T = 1000                                             # one 10 s clip: (1, 1000, 64) -> 160 032 samples
        mel = (0.5 * torch.randn(1, T, 64, generator=g)).to(torch.float16)
        dwav = torch.randn(1, eng.out_len(T), generator=g)
        dwav = dwav * (64.0 / float(dwav.abs().max()))       # the product's own backward scaling (dmx_grad_normalize target)
        x32 = mel.float().cuda().requires_grad_(True)
        w32 = ref32(x32)
        (g32,) = torch.autograd.grad((w32 * dwav.cuda()).sum(), x32)
        x16 = mel.cuda().requires_grad_(True)
        w16 = ref16(x16)
        (g16,) = torch.autograd.grad((w16.float() * dwav.cuda()).sum(), x16)
        wav = eng.forward(mel.to(_adt()).cuda())
        dmel = eng.backward(dwav.cuda())
        torch.cuda.synchronize()
        out[slope] = dict(wav_fp16_torch=_rel(w16, w32), wav_hip=_rel(wav, w32), grad_fp16_torch=_rel(g16, g32), grad_hip=_rel(dmel, g32))
        del eng
    print("\nvocoder fwd / input-gradient rel-L2 against the fp32 oracle (torch fp16 oracle | HIP):")
    for slope, r in out.items():
        print(f"  slope {slope}: wav {r['wav_fp16_torch']:.2e} | {r['wav_hip']:.2e}   grad {r['grad_fp16_torch']:.2e} | {r['grad_hip']:.2e}")
    try:
        import json, os
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump({str(k): v for k, v in out.items()}, open("gpurun_out/vocoder_fp16_gap.json", "w"), indent=1)
    except OSError:
        pass
    real, lin = out[0.1], out[1.0]
    # measured (MI355X, round 4): slope 0.1: torch fp16 6.8e-2, HIP 5.6e-2; slope 1.0: torch fp16 4.1e-2, HIP 3.1e-2 -- at slope 1 the
    # resblocks are linear but the network's last leaky_relu keeps its hard-coded slope 0.01 (modeling_speecht5.py forward), a x100 kink
    # in front of conv_post that every 16-bit run flips for near-zero units; the HIP path sits below torch's own fp16 run in both nets
    assert 1e-2 < real["grad_fp16_torch"] < 0.15, real       # the fp16 torch run of the oracle shows the gap ...
    assert real["grad_hip"] < 1.5 * real["grad_fp16_torch"] + 1e-2, real     # ... and the HIP path is not worse than it
    assert lin["grad_hip"] < 1.5 * lin["grad_fp16_torch"] + 1e-2, lin
    assert real["grad_fp16_torch"] > lin["grad_fp16_torch"], out             # more kinks, more flips
    assert real["wav_hip"] < 1e-2 and real["wav_fp16_torch"] < 1e-2, real    # forward: rounding level on both


def test_hifigan_layout_a_checkpoint_config_can_ask_for():
    """Three upsamplers (4, 4, 2), two resblock kernel sizes with two dilations each, 40 mel bins, slope 0.2 (vocoder/config.json keys
    upsample_rates / upsample_kernel_sizes / resblock_kernel_sizes / resblock_dilation_sizes / model_in_dim / leaky_relu_slope)."""
    from diffmusic_amd.engine import HifiGanEngine
    from oracle.models import HifiGan
    cfg = dict(model_in_dim=40, upsample_initial_channel=64, upsample_rates=[4, 4, 2], upsample_kernel_sizes=[8, 8, 4],
               resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [1, 3]], leaky_relu_slope=0.2)
    eng = HifiGanEngine(cfg)
    sd = eng.synth_state_dict(seed=4)
    eng.load_state_dict(sd)
    ref = HifiGan(**cfg)
    ref.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(12)
    B, T = 2, 48
    mel = torch.randn(B, T, 40, generator=g).to(_adt())
    dw = torch.randn(B, eng.out_len(T), generator=g)
    wav = eng.forward(mel.cuda())
    dmel = eng.backward(dw.cuda())
    torch.cuda.synchronize()
    x = mel.float().requires_grad_(True)
    wref = ref(x)
    (gref,) = torch.autograd.grad((wref * dw).sum(), x)
    assert wav.shape == wref.shape == (B, 32 * T)
    cos = torch.nn.functional.cosine_similarity(dmel.cpu().float().flatten(), gref.flatten(), dim=0).item()
    print("rel err wav", _rel(wav.cpu(), wref), "grad", _rel(dmel.cpu(), gref), "cos", cos)
    assert wref.abs().mean() > 0.02 and _rel(wav.cpu(), wref) < 3e-2 and cos > 0.97
