"""Run every stage twice on the same inputs and report bitwise differences."""
import sys, torch
sys.path.insert(0, '.')
from diffmusic_amd.engine import HifiGanEngine, VaeDecoderEngine, UNetEngine
from diffmusic_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
g = torch.Generator().manual_seed(0)
def cmp(name, a, b):
    d = (a.float() - b.float()).abs().max().item()
    print(f"{name:28s} max|diff| {d:.3e}  {'IDENTICAL' if torch.equal(a, b) else 'DIFFERENT'}", flush=True)
voc = HifiGanEngine(); voc.load_state_dict(voc.synth_state_dict(2))
mel = torch.randn(B, T, 64, generator=g).to(L.act_dtype()).cuda()
w1 = voc.forward(mel).clone(); w2 = voc.forward(mel).clone(); cmp("hifigan fwd", w1, w2)
d = torch.randn(w1.shape, generator=g).cuda()
voc.forward(mel); g1 = voc.backward(d.clone()).clone(); voc.forward(mel); g2 = voc.backward(d.clone()).clone(); cmp("hifigan bwd", g1, g2)
vae = VaeDecoderEngine(); vae.load_state_dict(vae.synth_state_dict(1))
z = torch.randn(B, 8, T // 4, 16, generator=g).cuda()
m1 = vae.decode_hip(z, z_scale=1.0, keep_state=True).clone(); m2 = vae.decode_hip(z, z_scale=1.0, keep_state=True).clone(); cmp("vae fwd", m1, m2)
dm = torch.randn(m1.shape, generator=g).to(L.act_dtype()).cuda()
vae.decode_hip(z, z_scale=1.0, keep_state=True); a1 = vae.backward(dm.clone()).clone(); vae.decode_hip(z, z_scale=1.0, keep_state=True); a2 = vae.backward(dm.clone()).clone(); cmp("vae bwd", a1, a2)
un = UNetEngine(); un.load_state_dict(un.synth_state_dict(0))
x = torch.randn(2 * B, 8, T // 4, 16, generator=g).cuda(); t = torch.full((2 * B,), 501.0).cuda(); c = torch.randn(2 * B, 512, generator=g).cuda()
e1 = un.forward(x, t, c).clone(); e2 = un.forward(x, t, c).clone(); cmp("unet fwd", e1, e2)
