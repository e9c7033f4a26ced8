"""randn_tensor semantics (reference: diffmusic/torch_utils.py:31-76)."""
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None):
    rand_device = device
    batch = shape[0]
    device = torch.device(device) if device is not None else torch.device("cpu")
    if generator is not None:
        g0 = generator[0] if isinstance(generator, list) else generator
        if g0.device.type != device.type and g0.device.type == "cpu":
            rand_device = "cpu"                                   # torch_utils.py:49-51
        elif g0.device.type != device.type and g0.device.type == "cuda":
            raise ValueError(f"Cannot generate a {device} tensor from a generator of type cuda.")
    if isinstance(generator, list) and len(generator) == 1:      # torch_utils.py:62-63
        generator = generator[0]
    if isinstance(generator, list):                               # torch_utils.py:65-72
        shp = (1,) + tuple(shape[1:])
        lat = [torch.randn(shp, generator=generator[i], device=rand_device, dtype=dtype)
               for i in range(batch)]
        return torch.cat(lat, dim=0).to(device)
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype).to(device)


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) on numpy uint32 arrays:
    counter (n, 4), key (2,) -> (n, 4).  Known-answer vectors of the Random123 distribution are checked in the tests."""
    import numpy as np
    c = np.array(counter, dtype=np.uint64) & 0xFFFFFFFF
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = 0xD2511F53 * c[:, 0]
        p1 = 0xCD9E8D57 * c[:, 2]
        n0 = ((p1 >> 32) ^ c[:, 1] ^ k0) & 0xFFFFFFFF
        n1 = p1 & 0xFFFFFFFF
        n2 = ((p0 >> 32) ^ c[:, 3] ^ k1) & 0xFFFFFFFF
        n3 = p0 & 0xFFFFFFFF
        c = np.stack([n0, n1, n2, n3], axis=1)
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c.astype(np.uint32)


def randn_philox(n, seed, offset=0):
    """CPU restatement of csrc/rng.hip (float64 Box-Muller): n normals of one clip keyed by `seed`."""
    import numpy as np
    blocks = (n + 3) // 4
    ctr = np.arange(blocks, dtype=np.uint64) + np.uint64(offset)
    counter = np.stack([ctr & np.uint64(0xFFFFFFFF), ctr >> np.uint64(32), np.zeros_like(ctr), np.zeros_like(ctr)], axis=1)
    u = philox4x32_10(counter, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.float64)
    out = np.empty((blocks, 4))
    for h in range(2):
        rad = np.sqrt(-2.0 * np.log((u[:, 2 * h] + 1.0) / 4294967296.0))
        ang = 2.0 * np.pi * (u[:, 2 * h + 1] / 4294967296.0)
        out[:, 2 * h], out[:, 2 * h + 1] = rad * np.cos(ang), rad * np.sin(ang)
    return out.reshape(-1)[:n]
