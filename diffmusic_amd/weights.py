"""Seeded synthetic weights for the benchmark configs (no pretrained checkpoints exist in either
box).  Parameter names/shapes come from the native library's own registry
(`dmx_model_param_*`), which uses the upstream diffusers / transformers naming, so a real
checkpoint's state_dict can be loaded through the same `load_state_dict` path."""
import math
import torch


def synth_state_dict(specs, seed=0, kind="generic", stride_of=None):
    """specs: list of (name, shape).  Variance-preserving fan-in init; norm gamma ~ 1, beta ~ 0."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in specs:
        shape = tuple(shape)
        if len(shape) >= 2:
            if kind == "hifigan" and name.startswith("upsampler"):
                s = stride_of(name)
                fan_in = shape[0] * shape[2] / s          # ConvTranspose1d (Cin, Cout, k)
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
            gain = 1.0
            if kind == "hifigan":
                gain = 1.0 if name.startswith("conv_pre") else 1.3   # inputs pass a leaky-relu(0.1)
                if ".convs2." in name:
                    gain = 0.6                                        # residual branch: keep x + f(x) bounded
            w = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        elif name.endswith("bias"):
            w = torch.randn(shape, generator=g) * 0.02
        else:
            w = 1.0 + 0.05 * torch.randn(shape, generator=g)
        sd[name] = w
    return sd
