"""-m gpu: clip lanes (diffmusic_amd/pipelines/lanes.py).  A lane is exactly a `Pipeline.__call__` on its clips, so the lane run must
equal the lanes' clip groups run one after the other through the plain loop -- bit for bit, for a deterministic sampler (DPS) and a
stochastic one (DSG, per-clip generators), on equal and unequal lane sizes; NaN-retry restarts all lanes; reruns are bit-identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from tests.test_gpu_step import SCHED                                              # noqa: E402
from tests.test_gpu_pipeline import UNET, _build                                    # noqa: E402


def _problem(sched_name, B, seed=3):
    from diffmusic_amd import inverse_problem as P
    L = 6400
    if sched_name == "dsg":
        op = P.PhaseRetrievalOperator(noiser=P.get_noiser("gaussian", 0.0))
    else:
        op = P.MusicInpaintingOperator(1, L, "box", 0.25, 0.5, 0.3, 0.1, 0.2, noiser=P.get_noiser("gaussian", 0.0))
    pipe = _build("musicldm", UNET, sched_name, op)
    g = torch.Generator().manual_seed(seed)
    clean = 0.3 * torch.sin(torch.arange(L) * 0.05)[None].repeat(B, 1) + 0.05 * torch.randn(B, L, generator=g)
    y = op.forward(clean.cuda())
    pe = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    ne = torch.nn.functional.normalize(torch.randn(B, 512, generator=g), dim=-1)
    lat0 = torch.randn(B, 8, 10, 16, generator=g)
    return pipe, y, pe, ne, lat0, L


def _call(pipe, y, pe, ne, lat0, ids, sched_name, N, lanes):
    gens = [torch.Generator().manual_seed(100 + k) for k in ids]
    eta, rate = (1.0, 0.08) if sched_name == "dsg" else (0.0, 5e-4)
    out = pipe(prompt_embeds=pe[ids], negative_prompt_embeds=ne[ids], audio_length_in_s=0.4, num_inference_steps=N, guidance_scale=2.0,
               latents=lat0[ids].clone(), measurement=y[ids].contiguous(), ip_guidance_rate=rate, eta=eta, generator=gens,
               show_progress=False, output_type="latent", lanes=lanes)
    return out.audios, [l.reshape(-1).clone() for l in pipe.last_losses]


@pytest.mark.parametrize("sched_name,B,n_lanes", [("dps", 4, 2), ("dsg", 5, 2), ("dps", 6, 3)])
def test_lanes_equal_the_clip_groups_run_alone(sched_name, B, n_lanes):
    from diffmusic_amd.pipelines.lanes import split_sizes
    N = 5
    pipe, y, pe, ne, lat0, L = _problem(sched_name, B)
    got, losses = _call(pipe, y, pe, ne, lat0, list(range(B)), sched_name, N, n_lanes)
    assert got.shape == (B, 8, 10, 16) and len(losses) == N and all(l.numel() == B for l in losses)
    o = 0
    for n in split_sizes(B, n_lanes):
        ids = list(range(o, o + n))
        ref, ref_losses = _call(pipe, y, pe, ne, lat0, ids, sched_name, N, 1)
        assert torch.equal(got[ids], ref), f"lane {ids}: latents differ from the plain loop on those clips"
        for i in range(N):
            assert torch.equal(losses[i][ids], ref_losses[i]), f"lane {ids}: loss of step {i} differs"
        o += n
    again, _ = _call(pipe, y, pe, ne, lat0, list(range(B)), sched_name, N, n_lanes)
    assert torch.equal(again, got)                               # run-to-run bitwise


def test_lanes_nan_retry_restarts_every_lane():
    pipe, y, pe, ne, lat0, L = _problem("dps", 4)
    real_step, calls = pipe.scheduler.step, {"n": 0}

    def step(*a, **kw):                                          # poison the loss of the 3rd lane-step of the first attempt
        out = real_step(*a, **kw)
        calls["n"] += 1
        if calls["n"] == 3:
            out.loss = out.loss * float("nan")
        return out
    pipe.scheduler.step = step
    gens = [torch.Generator().manual_seed(k) for k in range(4)]
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, audio_length_in_s=0.4, num_inference_steps=4, measurement=y, generator=gens,
               show_progress=False, output_type="latent", lanes=2)
    assert pipe.nan_restarts == 1 and bool(torch.isfinite(out.audios).all())
    assert calls["n"] >= 3 + 2 * 4                                # the poisoned attempt + a whole clean trajectory of both lanes


def test_lanes_refuse_a_shared_generator_for_per_step_noise():
    pipe, y, pe, ne, lat0, L = _problem("dsg", 4)
    with pytest.raises(ValueError, match="one generator per clip"):
        pipe(prompt_embeds=pe, negative_prompt_embeds=ne, audio_length_in_s=0.4, num_inference_steps=2, measurement=y, eta=1.0,
             generator=torch.Generator().manual_seed(0), show_progress=False, lanes=2)
