import sys, torch
sys.path.insert(0, '.')
from diffmusic_amd.engine import HifiGanEngine
from diffmusic_amd import _lib as L
B, T = 2, 500
g = torch.Generator().manual_seed(0)
voc = HifiGanEngine(); voc.load_state_dict(voc.synth_state_dict(2))
mel = torch.randn(B, T, 64, generator=g).to(L.act_dtype()).cuda()
outs = [voc.forward(mel).clone() for _ in range(4)]
print("fwd diffs:", [f"{(outs[0]-o).abs().max().item():.2e}" for o in outs[1:]], "nonzero frac", [(outs[0]!=o).float().mean().item() for o in outs[1:]])
