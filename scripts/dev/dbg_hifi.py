import torch, sys
sys.path.insert(0, '.')
from diffmusic_amd.engine import HifiGanEngine
from oracle.models import HifiGan
from diffmusic_amd import _lib as L
ADT = L.act_dtype()
SMALL = dict(model_in_dim=64, upsample_initial_channel=128, upsample_rates=[5, 4, 2, 2, 2],
             upsample_kernel_sizes=[16, 16, 8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
             resblock_dilation_sizes=[[1, 3, 5]] * 3, leaky_relu_slope=0.1)
def rel(a,b): return ((a.float()-b.float()).norm()/b.float().norm()).item()
for slope, post in [(1.0, 0.02), (0.1, 1.0), (0.1, 0.3)]:
    cfg = dict(SMALL, leaky_relu_slope=slope)
    eng = HifiGanEngine(cfg)
    sd = eng.synth_state_dict(seed=3)
    sd["conv_post.weight"] = sd["conv_post.weight"] * post
    eng.load_state_dict(sd)
    ref = HifiGan(**cfg); ref.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(11)
    B, T = 2, 40
    mel = torch.randn(B, T, 64, generator=g).to(ADT)
    dw = torch.randn(B, eng.out_len(T), generator=g)
    wav = eng.forward(mel.cuda()); dmel = eng.backward(dw.cuda()); torch.cuda.synchronize()
    x = mel.float().requires_grad_(True); wref = ref(x)
    (gref,) = torch.autograd.grad((wref*dw).sum(), x)
    print(f"slope {slope} post {post}: |wav| {wref.abs().mean():.3f} max {wref.abs().max():.3f} rel wav {rel(wav.cpu(), wref):.4f} rel grad {rel(dmel.cpu(), gref):.4f}")
