import sys, torch
sys.path.insert(0, '.')
from diffmusic_amd.engine import UNetEngine
eng = UNetEngine(); eng.load_state_dict(eng.synth_state_dict(0))
B = 16
x = torch.randn(B, 8, 250, 16, device="cuda"); t = torch.full((B,), 501.0, device="cuda"); cls = torch.randn(B, 512, device="cuda")
for _ in range(10): out = eng.forward(x, t, cls)
torch.cuda.synchronize()
