"""Clip-level data parallelism: clips are independent, weights are replicated,
clip k runs on rank k mod G with its own RNG generator, so results do not depend on G.  There is no
per-step collective; the only exchange is one all_gather of the finished waveforms (RCCL over xGMI
when the backend is "nccl", gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_indices(n_clips, rank, world):
    return list(range(rank, n_clips, world))


def gather_waveforms(local, n_clips, group=None):
    """local: (n_local, L) tensor holding clips shard_indices(n_clips, rank, world) in order.
    Returns (n_clips, L) on every rank, in global clip order."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (n_clips + world - 1) // world
    L = local.shape[1]
    dev = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":     # gloo has no device all_gather: stage through the host
        local = local.cpu()
    pad = torch.zeros(per, L, dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.empty(n_clips, L, dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_clips, r, world)
        out[idx] = parts[r][: len(idx)]
    return out.to(dev)
