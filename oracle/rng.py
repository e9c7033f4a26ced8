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
