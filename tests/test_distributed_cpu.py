"""CPU, world_size 2, gloo: clip sharding and the final waveform gather (the only collective of the path)."""
import os
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffmusic_amd.parallel import shard_indices, gather_waveforms
    from diffmusic_amd.torch_utils import randn_tensor
    idx = shard_indices(n_clips, rank, world)
    # per-clip generators: the noise (hence the result) does not depend on the number of ranks
    local = torch.cat([randn_tensor((1, 16), generator=[torch.Generator().manual_seed(k)], device=torch.device("cpu"),
                                    dtype=torch.float32) for k in idx]) if idx else torch.zeros(0, 16)
    full = gather_waveforms(local, n_clips)
    q.put((rank, full))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_clips, world, port = 5, 2, 29611
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    ref = torch.cat([torch.randn((1, 16), generator=torch.Generator().manual_seed(k)) for k in range(n_clips)])
    for r in range(world):
        assert torch.equal(got[r], ref)


def test_shard_indices_partition():
    from diffmusic_amd.parallel import shard_indices
    for n, w in ((32, 8), (16, 4), (5, 2), (3, 8)):
        allidx = sorted(i for r in range(w) for i in shard_indices(n, r, w))
        assert allidx == list(range(n))
