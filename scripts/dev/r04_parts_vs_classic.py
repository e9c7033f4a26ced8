"""GPU box: production U-Nets, GroupNorm statistics from producer partial sums vs the classic statistics pass on the SAME batch (same tiles)."""
import os, sys, torch
sys.path.insert(0, '.')
from diffmusic_amd.engine import UNetEngine, UNET_AUDIOLDM2_DEFAULT
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for kind in ("musicldm", "audioldm2"):
    eng = UNetEngine(UNET_AUDIOLDM2_DEFAULT if kind == "audioldm2" else None)
    eng.load_state_dict(eng.synth_state_dict(0))
    for B in (2, 8, 16):
        g = torch.Generator().manual_seed(B)
        x = torch.randn(B, 8, 250, 16, generator=g).cuda(); t = torch.full((B,), 501.0, device="cuda")
        if kind == "audioldm2":
            kw = dict(encoder_hidden_states=torch.randn(B, 8, 768, generator=g).cuda(), encoder_hidden_states_1=torch.randn(B, 16, 1024, generator=g).cuda(),
                      encoder_attention_mask_1=torch.ones(B, 16, device="cuda"))
        else:
            kw = dict(class_labels=torch.randn(B, 512, generator=g).cuda())
        os.environ.pop("DMX_NO_GN_PARTS", None)
        a = eng.forward(x, t, **kw).clone()
        a2 = eng.forward(x, t, **kw).clone()
        os.environ["DMX_NO_GN_PARTS"] = "1"
        c = eng.forward(x, t, **kw).clone()
        os.environ.pop("DMX_NO_GN_PARTS", None)
        # per-clip: clip 0 alone (different tiles) against its row of the batch
        kw1 = {k: v[:1].contiguous() for k, v in kw.items()}
        s = eng.forward(x[:1].contiguous(), t[:1], **kw1)
        os.environ["DMX_NO_GN_PARTS"] = "1"
        sc = eng.forward(x[:1].contiguous(), t[:1], **kw1)
        os.environ.pop("DMX_NO_GN_PARTS", None)
        print(f"{kind} B={B}: parts vs classic {rel(a, c):.2e} (rerun identical: {torch.equal(a, a2)}); clip 0 alone vs in batch: parts {rel(a[:1], s):.2e}, classic {rel(c[:1], sc):.2e}; "
              f"alone parts vs alone classic {rel(s, sc):.2e}", flush=True)
