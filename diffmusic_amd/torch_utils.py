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
