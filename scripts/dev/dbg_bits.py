import sys, os, torch
sys.path.insert(0, '.')
from diffmusic_amd.engine import HifiGanEngine
from diffmusic_amd import _lib as L
from oracle.models import HifiGan
SMALL = dict(model_in_dim=64, upsample_initial_channel=int(os.environ.get("C0", "128")), upsample_rates=[5, 4, 2, 2, 2],
             upsample_kernel_sizes=[16, 16, 8, 4, 4], resblock_kernel_sizes=[3, 7, 11],
             resblock_dilation_sizes=[[1, 3, 5]] * 3, leaky_relu_slope=float(os.environ.get("SLOPE", "1.0")))
def rel(a, b): return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()
for B, T in ((1, 40), (3, 57)):
    eng = HifiGanEngine(SMALL); sd = eng.synth_state_dict(seed=3); eng.load_state_dict(sd)
    ref = HifiGan(**SMALL); ref.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(11)
    mel = torch.randn(B, T, 64, generator=g).to(L.act_dtype())
    dw = torch.randn(B, eng.out_len(T), generator=g)
    wav = eng.forward(mel.cuda()); dmel = eng.backward(dw.cuda()); torch.cuda.synchronize()
    x = mel.float().requires_grad_(True); wref = ref(x)
    (gref,) = torch.autograd.grad((wref * dw).sum(), x)
    print(os.environ.get("DMX_NO_PAIR"), SMALL["upsample_initial_channel"], SMALL["leaky_relu_slope"], B, T, "wav", rel(wav.cpu(), wref), "grad", rel(dmel.cpu(), gref), flush=True)
