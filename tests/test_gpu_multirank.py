"""-m gpu: the multi-rank launch rehearsed on a 1-GPU box: two rank processes, gloo rendezvous on 127.0.0.1,
both on cuda:0.  (1) `bench.py --gpus 2` -- the command the driver's SCALE run uses -- starts its own ranks, times the loop, decodes
and all-gathers; (2) the product `Pipeline.__call__(shard=True)` gives, on every rank, all clips in order and BIT-equal to the
single-rank call (per-clip generators and conditioning rows keyed by the global clip number: nothing a clip sees depends on the
number of ranks).  The rank processes are fresh children (`subprocess`): each one brings up its own HIP runtime."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_rank_rehearsal_on_one_gpu():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--global-batch", "8",
           "--steps", "2", "--warmup", "1", "--settle", "2", "--no-cpu-baseline", "--no-stage-times", "--no-full-trajectory"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 8 and d["config"]["clips_per_gpu"] == 4
    assert d["after_loop"]["gather_world_size"] == 2 and d["after_loop"]["gather_ms"] > 0.0
    assert d["config"]["finite"] and d["value"] is not None and d["value"] > 0.0
    assert "gloo REHEARSAL" in d["config"]["parallelism"]


@pytest.mark.parametrize("n_clips", [5, 1])
def test_pipeline_shard_two_ranks_bit_equal_to_single_rank(tmp_path, n_clips):
    port = _free_port()
    out = str(tmp_path / "gathered.npy")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), str(n_clips), out],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=600)
            logs.append(o)
    finally:
        for p in procs:                      # exactly the processes started above
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-1500:] for l in logs)
    from tests.multirank_worker import problem, gens
    pipe, kw = problem(n_clips)
    # (a) bit-equal to the same clips run WITHOUT torch.distributed in the per-rank batch compositions (rank r: clips r, r + 2, ...):
    # sharding changes nothing a clip sees -- generator, conditioning row, measurement row are keyed by the global clip number -- and
    # the gather puts every clip back at its global index
    ref = np.zeros((n_clips, 6400), dtype=np.float32)
    for rank in range(2):
        sel = list(range(rank, n_clips, 2))
        if not sel:
            continue
        kws = dict(kw, prompt_embeds=kw["prompt_embeds"][sel], measurement=kw["measurement"][sel].contiguous())
        ref[sel] = np.asarray(pipe(generator=[gens(n_clips)[k] for k in sel], **kws).audios)
    assert np.isfinite(ref).all() and float(np.abs(ref).max()) > 1e-3
    got = [np.load(out.replace(".npy", f"_rank{rank}.npy")) for rank in range(2)]
    for rank in range(2):
        assert got[rank].shape == ref.shape
        assert np.array_equal(got[rank], ref), (rank, float(np.abs(got[rank] - ref).max()))
    # (b) against ONE call on all clips: the same result up to the batch-size dependence of the tile / split-K choice inside the U-Net
    # (different fp32 summation order) -- no dependence on the number of ranks beyond that
    full = np.asarray(pipe(generator=gens(n_clips), **kw).audios)
    snr = 10 * np.log10((full.astype(np.float64) ** 2).sum() / max(((full - got[0]).astype(np.float64) ** 2).sum(), 1e-30))
    print(f"sharded over 2 ranks vs one batch of {n_clips}: waveform SNR {snr:.1f} dB")
    assert snr > 30.0


def test_bench_rccl_path_at_world_size_one():
    """The RCCL ("nccl") code path the driver's multi-GPU run takes -- init_process_group(device_id=...), device barrier, max-over-ranks
    all_reduce, device all_gather of the waveforms -- executed on the one GPU of this box at world size 1 (`--force-dist`)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl", "--steps", "2", "--warmup", "1",
           "--settle", "2", "--no-cpu-baseline", "--no-stage-times", "--no-full-trajectory"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["finite"] and d["value"] > 0.0
    assert "RCCL" in d["after_loop"]["collective"] and d["after_loop"]["gather_world_size"] == 1
    assert d["after_loop"]["gathered_equals_local"] is True and d["after_loop"]["gather_ms"] > 0.0
    assert d["config"]["launched_by"] == "bench.py spawn"


def test_pipeline_shard_through_rccl_at_world_size_one(tmp_path):
    """`Pipeline.__call__(shard=True)` under an RCCL process group of one rank: the gathered clips equal the call without torch.distributed."""
    out = str(tmp_path / "gathered.npy")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), "3", out, "nccl"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    from tests.multirank_worker import problem, gens
    pipe, kw = problem(3)
    ref = np.asarray(pipe(generator=gens(3), **kw).audios)
    got = np.load(out.replace(".npy", "_rank0.npy"))
    assert np.array_equal(got, ref)
