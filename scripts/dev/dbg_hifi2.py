import torch, sys
sys.path.insert(0, '.')
from diffmusic_amd.engine import HifiGanEngine
from oracle.models import HifiGan
from diffmusic_amd import _lib as L
ADT = L.act_dtype()
def rel(a,b): return ((a.float()-b.float()).norm()/b.float().norm()).item()
base = dict(model_in_dim=64, upsample_initial_channel=128, leaky_relu_slope=1.0)
cases = {
 "A r2k4 rb3 d1": dict(upsample_rates=[2], upsample_kernel_sizes=[4], resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1]]),
 "B r5k16 rb3 d1": dict(upsample_rates=[5], upsample_kernel_sizes=[16], resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1]]),
 "B2 r4k16 rb3 d1": dict(upsample_rates=[4], upsample_kernel_sizes=[16], resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1]]),
 "C r2k4 rb11 d5": dict(upsample_rates=[2], upsample_kernel_sizes=[4], resblock_kernel_sizes=[11], resblock_dilation_sizes=[[5]]),
 "D r2k4 rb3,7,11 d1": dict(upsample_rates=[2], upsample_kernel_sizes=[4], resblock_kernel_sizes=[3,7,11], resblock_dilation_sizes=[[1]]*3),
 "E r2k4 rb3 d1,3,5": dict(upsample_rates=[2], upsample_kernel_sizes=[4], resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1,3,5]]),
 "F two stages": dict(upsample_rates=[2,2], upsample_kernel_sizes=[4,4], resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1]]),
}
for name, c in cases.items():
    cfg = dict(base, **c)
    eng = HifiGanEngine(cfg)
    sd = eng.synth_state_dict(seed=3)
    sd["conv_post.weight"] = sd["conv_post.weight"] * 0.1
    eng.load_state_dict(sd)
    ref = HifiGan(**cfg); ref.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(11)
    B, T = 2, 40
    mel = torch.randn(B, T, 64, generator=g).to(ADT)
    dw = torch.randn(B, eng.out_len(T), generator=g)
    wav = eng.forward(mel.cuda()); dmel = eng.backward(dw.cuda()); torch.cuda.synchronize()
    x = mel.float().requires_grad_(True); wref = ref(x)
    (gref,) = torch.autograd.grad((wref*dw).sum(), x)
    print(f"{name}: |wav| {wref.abs().mean():.3f} rel wav {rel(wav.cpu(), wref):.4f} rel grad {rel(dmel.cpu(), gref):.4f}")
