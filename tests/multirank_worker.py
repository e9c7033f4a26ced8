"""Rank process of tests/test_gpu_multirank.py (not a test module): joins a gloo group of WORLD_SIZE ranks that all use cuda:0 of a
1-GPU box, runs the product `Pipeline.__call__(shard=True)` on small nets and lets rank 0 save what every rank must hold after the
final all_gather (diffmusic_amd/parallel.py): all clips in global order.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/multirank_worker.py <n_clips> <out.npy> [gloo|nccl]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem(n_clips):
    """Pipeline, conditioning and measurement of the sharding rehearsal (the same objects in every rank and in the test's
    single-rank reference call)."""
    from diffmusic_amd import inverse_problem as P
    from tests.test_gpu_pipeline import _build, UNET
    L = 6400
    op = P.MusicInpaintingOperator(1, L, "box", 0.25, 0.5, 0.3, 0.1, 0.2, noiser=P.get_noiser("gaussian", 0.0))
    pipe = _build("musicldm", UNET, "dsg", op)            # DSG: a stochastic sampler, so the per-clip RNG streams are part of the contract
    g = torch.Generator().manual_seed(31)
    clean = 0.3 * torch.sin(torch.arange(L) * 0.05)[None].repeat(n_clips, 1) + 0.05 * torch.randn(n_clips, L, generator=g)
    y = op.forward(clean.cuda())
    pe = torch.nn.functional.normalize(torch.randn(n_clips, 512, generator=g), dim=-1)
    kw = dict(prompt_embeds=pe, audio_length_in_s=0.4, num_inference_steps=4, guidance_scale=2.0, measurement=y, ip_guidance_rate=0.08,
              eta=1.0, show_progress=False, output_type="np")
    return pipe, kw


def gens(n_clips):
    return [torch.Generator().manual_seed(100 + k) for k in range(n_clips)]


def main():
    n_clips, out_path = int(sys.argv[1]), sys.argv[2]
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    if backend == "nccl":                                  # RCCL: one device per rank (the world-size-1 rehearsal on the 1-GPU box)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        pipe, kw = problem(n_clips)
        out = pipe(generator=gens(n_clips), shard=True, **kw)
        a = np.asarray(out.audios)
        assert a.shape[0] == n_clips, a.shape
        np.save(out_path.replace(".npy", f"_rank{rank}.npy"), a)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    print(f"[worker] rank {rank}/{world} done", flush=True)


if __name__ == "__main__":
    main()
