import sys, torch
sys.path.insert(0, '.')
from diffmusic_amd.engine import VaeDecoderEngine
eng = VaeDecoderEngine(); eng.load_state_dict(eng.synth_state_dict(1))
B = 8
x0 = torch.randn(B, 8, 250, 16, device="cuda")
for _ in range(10):
    mel = eng.decode_hip(x0, z_scale=1.0, keep_state=True)
    dmel = torch.randn(mel.shape, device='cuda').to(mel.dtype).contiguous()
    g = eng.backward(dmel, z_scale=1.0)
torch.cuda.synchronize()
