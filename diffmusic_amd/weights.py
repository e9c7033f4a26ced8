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


def check_manifest(specs, state_dict, what="model", allow_unexpected=()):
    """Compares a checkpoint's tensors with the parameter manifest of a native executor (`dmx_model_param_*`: upstream diffusers /
    transformers names and shapes for the configured architecture) and raises ONE ValueError that lists every missing tensor, every
    tensor of another shape and every tensor the architecture does not know -- the defence against an architecture config that does
    not match the checkpoint (block_out_channels, head counts, class_embeddings_concat ...: SURVEY.md Appendix A is recalled, not
    verified).  `allow_unexpected`: name prefixes of tensors a checkpoint may carry that the hot path does not use (e.g. the VAE
    encoder of an AutoencoderKL checkpoint)."""
    want = {name: tuple(shape) for name, shape in specs}
    missing = [n for n in want if n not in state_dict]
    wrong = [(n, want[n], tuple(state_dict[n].shape)) for n in want if n in state_dict and tuple(state_dict[n].shape) != want[n]]
    extra = [n for n in state_dict if n not in want and not any(n.startswith(p) for p in allow_unexpected)]
    if not (missing or wrong or extra):
        return
    lines = [f"{what}: the checkpoint does not match the configured architecture "
             f"({len(missing)} missing, {len(wrong)} of another shape, {len(extra)} unexpected of {len(want)} expected tensors)"]
    lines += [f"  missing     {n} {want[n]}" for n in missing]
    lines += [f"  shape       {n}: expected {w}, checkpoint has {g}" for n, w, g in wrong]
    lines += [f"  unexpected  {n} {tuple(state_dict[n].shape)}" for n in extra]
    raise ValueError("\n".join(lines))
