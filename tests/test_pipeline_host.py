"""CPU: host logic of the product pipeline (diffmusic_amd/pipelines/pipeline_musicldm.py) with stub engines:
NaN-retry (reference pipeline_musicldm.py:682,741-756), the optim_prompt hook (:710-723), all-B output, and the clip-sharded
call under a world-2 gloo group with the single final gather."""
import os
import warnings

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.stubs import make_pipeline

N, B, SECONDS = 12, 3, 0.64            # 0.64 s -> mel height 64 -> latent (B, 8, 16, 4)


def _call(pipe, seeds=None, **kw):
    g = torch.Generator().manual_seed(99)
    pe = torch.randn(B, 512, generator=g)
    gens = [torch.Generator().manual_seed(s) for s in (seeds or range(B))]
    args = dict(prompt_embeds=pe, audio_length_in_s=SECONDS, num_inference_steps=N, generator=gens, show_progress=False, eta=0.5)
    args.update(kw)
    return pipe(**args)


def test_returns_all_clips_and_is_reproducible():
    a = _call(make_pipeline()).audios
    b = _call(make_pipeline()).audios
    assert a.shape == (B, int(SECONDS * 16000)) and a.dtype == np.float32
    assert np.array_equal(a, b)
    assert not np.array_equal(a[0], a[1])


def test_nan_retry_redraws_latents_and_restarts():
    """Loss NaN once at step 4 of the first trajectory => fresh latents from the same generators, loop restarted from step 0,
    one restart counted, the run completes with N more steps (pipeline_musicldm.py:741-756)."""
    pipe = make_pipeline(nan_at={4})
    out = _call(pipe)
    s = pipe.scheduler
    assert pipe.nan_restarts == 1
    assert s.calls == 5 + N
    assert len(s.first_samples) == 2 and not torch.equal(s.first_samples[0], s.first_samples[1])   # redrawn, not reused
    assert np.isfinite(out.audios).all()
    # the restarted trajectory equals a clean run that starts from the redrawn latents
    pipe0 = make_pipeline(nan_at={4})
    got = _call(pipe0, eta=0.0).audios
    clean = make_pipeline()
    ref = clean(prompt_embeds=torch.randn(B, 512, generator=torch.Generator().manual_seed(99)), audio_length_in_s=SECONDS,
                num_inference_steps=N, latents=pipe0.scheduler.first_samples[1].clone(), show_progress=False, eta=0.0).audios
    assert np.allclose(got, ref, atol=1e-6)


def test_nan_retry_gives_up_after_eleven_restarts():
    """`retry = 10` and the test is `retry >= 0` (:682,:742): NaN at the first step of every trajectory restarts 11 times,
    then the check is disabled and the loop runs to the end."""
    pipe = make_pipeline(nan_at=set(range(0, 40)))
    _call(pipe)
    assert pipe.nan_restarts == 11
    assert pipe.scheduler.calls == 11 + N


def test_nan_check_every_k_steps_still_catches_it():
    pipe = make_pipeline(nan_at={2})
    pipe.nan_check_every = 4
    _call(pipe)
    assert pipe.nan_restarts == 1 and pipe.scheduler.calls == 4 + N


def test_optim_prompt_is_a_no_op_call():
    """optim_prompt=True calls scheduler.optim_prompt at every t % 30 == 1 and leaves the result unchanged (a8)."""
    a = _call(make_pipeline()).audios
    pipe = make_pipeline()
    b = _call(pipe, optim_prompt=True).audios
    ts = pipe.scheduler._timesteps_host
    assert pipe.scheduler.optim_calls == [t for t in ts if t % 30 == 1] and len(pipe.scheduler.optim_calls) > 0
    assert np.array_equal(a, b)


def test_missing_negative_embeds_warns_unless_declared():
    pipe = make_pipeline()
    pipe.assume_uncond_equals_cond = False
    with pytest.warns(UserWarning, match="negative_prompt_embeds"):
        _call(pipe)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _call(pipe, guidance_scale=1.0)                     # no CFG: nothing to warn about
        _call(pipe, negative_prompt_embeds=torch.zeros(B, 512))


def test_single_generator_is_refused_when_sharding():
    pipe = make_pipeline()
    with pytest.raises(RuntimeError):
        _call(pipe, shard=True)                             # no process group


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = make_pipeline()
        g = torch.Generator().manual_seed(99)
        pe = torch.randn(n_clips, 512, generator=g)
        meas = torch.randn(n_clips, 100, generator=g)
        gens = [torch.Generator().manual_seed(s) for s in range(n_clips)]
        out = pipe(prompt_embeds=pe, audio_length_in_s=SECONDS, num_inference_steps=N, generator=gens, show_progress=False, eta=0.5,
                   measurement=meas, shard=True)
        lat = pipe(prompt_embeds=pe, audio_length_in_s=SECONDS, num_inference_steps=3,
                   generator=[torch.Generator().manual_seed(s) for s in range(n_clips)], show_progress=False, eta=0.5,
                   measurement=meas, group=dist.group.WORLD, output_type="latent").audios
        q.put((rank, out.audios, lat.numpy(), pipe.scheduler.calls))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [5, 1])
def test_pipeline_shards_clips_and_gathers_world2(n_clips):
    """Pipeline.__call__(shard=True) under gloo, world 2: every rank steps only its clips (rank 0: 0,2,4; rank 1: 1,3), the
    result on every rank is all clips in order and equals the unsharded call (per-clip generators => independent of G);
    n_clips = 1 leaves rank 1 without work (it still takes part in the gather)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, 29650 + n_clips
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    for _ in range(world):
        r, a, lat, calls = q.get(timeout=180)
        got[r] = (a, lat, calls)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    pipe = make_pipeline()
    g = torch.Generator().manual_seed(99)
    pe = torch.randn(n_clips, 512, generator=g)
    meas = torch.randn(n_clips, 100, generator=g)
    ref = pipe(prompt_embeds=pe, audio_length_in_s=SECONDS, num_inference_steps=N,
               generator=[torch.Generator().manual_seed(s) for s in range(n_clips)], show_progress=False, eta=0.5, measurement=meas).audios
    ref_lat = pipe(prompt_embeds=pe, audio_length_in_s=SECONDS, num_inference_steps=3,
                   generator=[torch.Generator().manual_seed(s) for s in range(n_clips)], show_progress=False, eta=0.5, measurement=meas,
                   output_type="latent").audios.numpy()
    for r in range(world):
        assert got[r][0].shape == (n_clips, int(SECONDS * 16000))
        assert np.allclose(got[r][0], ref, atol=1e-6), r
        assert np.allclose(got[r][1], ref_lat, atol=1e-6), r
    assert got[0][2] == (N + 3 if n_clips else 0) and got[1][2] == (N + 3 if n_clips > 1 else 0)


def test_clip_lanes_host_logic_equals_per_group_calls_and_restarts_all_lanes():
    """`lanes=2` on the CPU stand-ins (the lane runner's enqueue order and NaN protocol without streams): every lane is the pipeline on
    its clip group -- same generators, conditioning rows and losses -- and a NaN in one lane restarts all of them (pipeline_musicldm.py:741-756)."""
    from diffmusic_amd.pipelines.lanes import split_sizes
    assert split_sizes(8, 2) == [4, 4] and split_sizes(5, 2) == [3, 2] and split_sizes(2, 3) == [1, 1] and split_sizes(7, 3) == [3, 2, 2]
    pipe = make_pipeline()
    whole = _call(pipe, lanes=2, eta=0.0)
    assert whole.audios.shape[0] == B and len(pipe.last_losses) == N and all(l.numel() == B for l in pipe.last_losses)
    g = torch.Generator().manual_seed(99)
    pe = torch.randn(B, 512, generator=g)
    o = 0
    for n in split_sizes(B, 2):
        ids = list(range(o, o + n))
        part = make_pipeline()(prompt_embeds=pe[ids], audio_length_in_s=SECONDS, num_inference_steps=N, show_progress=False, eta=0.0,
                               generator=[torch.Generator().manual_seed(s) for s in ids]).audios
        assert np.array_equal(whole.audios[ids], part)
        o += n
    bad = make_pipeline(nan_at={3})                       # the 4th lane-step of the first attempt
    out = _call(bad, lanes=2, eta=0.0)
    assert bad.nan_restarts == 1 and np.isfinite(out.audios).all()
    assert bad.scheduler.calls >= 4 + 2 * N
    with pytest.raises(ValueError, match="callback"):
        _call(make_pipeline(), lanes=2, callback=lambda i, t, x: None)
    with pytest.raises(ValueError, match="one generator per clip"):
        make_pipeline()(prompt_embeds=pe, audio_length_in_s=SECONDS, num_inference_steps=2, show_progress=False, eta=0.5,
                        generator=torch.Generator().manual_seed(0), lanes=2)
