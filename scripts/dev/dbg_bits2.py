import sys, os, torch
sys.path.insert(0, '.')
from diffmusic_amd.engine import HifiGanEngine
from diffmusic_amd import _lib as L
from oracle.models import HifiGan
def rel(a, b): return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()
def run(name, **cfgkw):
    cfg = dict(model_in_dim=64, upsample_initial_channel=128, upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
               resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, leaky_relu_slope=1.0)
    cfg.update(cfgkw)
    eng = HifiGanEngine(cfg); sd = eng.synth_state_dict(seed=3); eng.load_state_dict(sd)
    ref = HifiGan(**cfg); ref.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(11)
    B, T = 2, 300
    mel = torch.randn(B, T, 64, generator=g).to(L.act_dtype())
    dw = torch.randn(B, eng.out_len(T), generator=g)
    wav = eng.forward(mel.cuda()); dmel = eng.backward(dw.cuda()); torch.cuda.synchronize()
    x = mel.float().requires_grad_(True); wref = ref(x)
    (gref,) = torch.autograd.grad((wref * dw).sum(), x)
    print(f"{name:40s} wav {rel(wav.cpu(), wref):.2e} grad {rel(dmel.cpu(), gref):.2e}", flush=True)
K1 = dict(resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1]])
run("3 stages (64,32,16) k=3 d=1", upsample_rates=[5, 4, 2], upsample_kernel_sizes=[16, 16, 8], **K1)
run("4 stages (..8) k=3 d=1", upsample_rates=[5, 4, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4], **K1)
run("5 stages (..4) k=3 d=1", **K1)
run("3 stages C0=32 (16,8,4)", upsample_initial_channel=32, upsample_rates=[5, 4, 2], upsample_kernel_sizes=[16, 16, 8], **K1)
run("1 stage C0=32 (16)", upsample_initial_channel=32, upsample_rates=[5], upsample_kernel_sizes=[16], **K1)
run("1 stage C0=16 (8)", upsample_initial_channel=16, upsample_rates=[5], upsample_kernel_sizes=[16], **K1)
run("1 stage C0=8 (4)", upsample_initial_channel=8, upsample_rates=[5], upsample_kernel_sizes=[16], **K1)
run("2 stages C0=64 (32,16)", upsample_initial_channel=64, upsample_rates=[5, 4], upsample_kernel_sizes=[16, 16], **K1)
