"""GPU box: the production-size pipeline at clip lengths other than the benchmark's 10 s (2 guided DPS steps, batch 2): finite outputs."""
import sys, torch
sys.path.insert(0, '.')
import bench
from diffmusic_amd.pipelines import get_pipeline
from diffmusic_amd.schedulers import get_scheduler
from diffmusic_amd import inverse_problem as P
pipe = get_pipeline("musicldm").from_pretrained("synthetic", seed=0).to("cuda")
pipe.assume_uncond_equals_cond = True
for secs in (5.0, 8.0, 10.24, 3.3, 7.8, 2.56):
    L = int(secs * 16000)
    op = P.MusicInpaintingOperator(secs, 16000, "box", 1, 2, 0.3, 0.1, 1.0, noiser=P.get_noiser("gaussian", 0.0))
    pipe.scheduler = get_scheduler("dps")(operator=op, **bench.SCHED_CFG)
    clips = torch.stack([bench.synth_clip(k, L) for k in range(2)]).cuda()
    y = op.forward(clips)
    pe = torch.nn.functional.normalize(torch.randn(2, 512, generator=torch.Generator().manual_seed(1)), dim=-1)
    try:
        out = pipe(prompt_embeds=pe, audio_length_in_s=secs, num_inference_steps=2, generator=[torch.Generator().manual_seed(k) for k in range(2)],
                   measurement=y, show_progress=False)
    except Exception as e:           # noqa: BLE001
        print(f"{secs} s: FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
        continue
    a = torch.from_numpy(out.audios)
    print(f"{secs} s: audio {tuple(a.shape)} finite {bool(torch.isfinite(a).all())} nan_restarts {pipe.nan_restarts} loss {[float(l.mean()) for l in pipe.last_losses]}", flush=True)
