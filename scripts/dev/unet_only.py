"""U-Net forwards only (profiling target): `python scripts/dev/unet_only.py [musicldm|audioldm2] [2B] [n_forwards]`.
Prints the HIP-event time per forward (median of the timed ones)."""
import sys, torch
sys.path.insert(0, '.')
from diffmusic_amd.engine import UNetEngine, UNET_AUDIOLDM2_DEFAULT
kind = sys.argv[1] if len(sys.argv) > 1 else "musicldm"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if kind == "musicldm" else 8)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eng = UNetEngine(UNET_AUDIOLDM2_DEFAULT if kind == "audioldm2" else None)
eng.load_state_dict(eng.synth_state_dict(0))
x = torch.randn(B, 8, 250, 16, device="cuda"); t = torch.full((B,), 501.0, device="cuda")
if kind == "audioldm2":
    kw = dict(encoder_hidden_states=torch.randn(B, 8, 768, device="cuda"), encoder_hidden_states_1=torch.randn(B, 16, 1024, device="cuda"),
              encoder_attention_mask_1=torch.ones(B, 16, device="cuda"))
else:
    kw = dict(class_labels=torch.randn(B, 512, device="cuda"))
for _ in range(3): out = eng.forward(x, t, **kw)
torch.cuda.synchronize()
ms = []
for _ in range(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = eng.forward(x, t, **kw); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
ms.sort()
print(f"unet {kind} 2B={B}: {ms[len(ms)//2]:.3f} ms per forward (median of {n}; min {ms[0]:.3f})", flush=True)
