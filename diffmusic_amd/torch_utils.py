"""randn_tensor with the reference's semantics (diffmusic/torch_utils.py:31-76): a list of generators
seeds each clip individually; CPU generators draw on the CPU and the result is moved to `device`, so
the noise is identical whatever the device (and whatever the number of GPUs the clips are sharded on)."""
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    rand_device = device
    batch_size = shape[0]
    device = torch.device(device) if device is not None else torch.device("cpu")
    if generator is not None:
        gtype = generator[0].device.type if isinstance(generator, list) else generator.device.type
        if gtype != device.type and gtype == "cpu":
            rand_device = "cpu"
        elif gtype != device.type and gtype == "cuda":
            raise ValueError(f"Cannot generate a {device} tensor from a generator of type {gtype}.")
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        shp = (1,) + tuple(shape[1:])
        parts = [torch.randn(shp, generator=generator[i], device=rand_device, dtype=dtype) for i in range(batch_size)]
        return torch.cat(parts, dim=0).to(device)
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype).to(device)


def randn_philox(shape, seeds, offset=0, device="cuda"):
    """Device-side N(0,1) draw (csrc/rng.hip, Philox4x32-10 + Box-Muller): shape (B, ...) fp32, clip b keyed by seeds[b];
    `offset` counts Philox blocks (4 normals each) already consumed -- advance it by ceil(n / 4) per draw.  Opt-in
    replacement for `randn_tensor` on the per-step noise of DSG / DiffMusic (no host draw, no H2D copy); the values do not
    match torch's generators, but they depend only on (seed_b, offset, element), never on the batch or the GPU count."""
    import ctypes as C
    from . import _lib as L
    B = int(shape[0])
    n = 1
    for d in shape[1:]:
        n *= int(d)
    if len(seeds) != B:
        raise ValueError(f"need one seed per clip: got {len(seeds)} for batch {B}")
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    arr = (C.c_ulonglong * B)(*[int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds])
    L.check(L.lib().dmx_randn_philox(C.c_void_p(out.data_ptr()), B, n, arr, int(offset) & 0xFFFFFFFFFFFFFFFF,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "randn_philox")
    return out
