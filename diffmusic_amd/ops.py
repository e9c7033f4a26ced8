"""PyTorch-ROCm custom ops `torch.ops.diffmusic_hip.*`: TORCH_LIBRARY wrappers over the C ABI,
compiled from csrc_torch/torch_ops.cpp into lib/libdiffmusic_torch_ops.so and loaded with `torch.ops.load_library`.

    from diffmusic_amd import ops
    prev, x0 = ops.hip.sched_update(1, x, eps, x0, g0, inv_scale, None, a_t, a_p, sigma, rate, 1e-8, False)

The facades (`Scheduler.step`, `Pipeline.__call__`, the operators) may call either this layer or the ctypes binding of the
same entry points (`_lib.py`); both end in the same `extern "C"` launchers on torch's current HIP stream.  By default the
facade's per-step device work goes through this layer: the network stages (U-Net, VAE decode forward / backward, HiFi-GAN
forward / backward: diffmusic_amd/engine.py), the gradient rescale and the scheduler arithmetic (x0 prediction, CFG combine,
fused update).  DMX_TORCH_OPS=0 switches those calls to the ctypes binding; so does an op library that cannot be loaded
(missing, or built against another torch): one warning, then ctypes -- the same HIP kernels either way.  Neither binding has a
CPU fallback: without libdiffmusic_hip.so everything raises."""
import os
import warnings

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdiffmusic_torch_ops.so")
USE_TORCH_OPS = os.environ.get("DMX_TORCH_OPS", "1") not in ("", "0")
_loaded = False

OP_NAMES = ("sched_pred_x0", "cfg_combine", "sched_update", "randn_philox", "mask_mul", "l2norm", "resample_fwd", "resample_bwd",
            "logmel_fwd", "logmel_bwd", "stft_mag_fwd", "stft_mag_bwd", "melscale_fwd", "unet_fwd", "unet_fwd_ctx", "vae_dec_fwd",
            "vae_dec_bwd", "hifigan_fwd", "hifigan_bwd", "grad_normalize_", "mel_guidance", "abi_version")
_usable = None


def load():
    """Loads the op library once (raises if it is missing: build it with `python -m diffmusic_amd.build`)."""
    global _loaded
    if not _loaded:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m diffmusic_amd.build` (no CPU fallback)")
        from . import _lib
        _lib.lib()                                   # libdiffmusic_hip.so first (the op library links it by rpath $ORIGIN)
        torch.ops.load_library(LIB_PATH)             # (its TORCH_LIBRARY init refuses a libdiffmusic_hip.so of another C-ABI version)
        if int(torch.ops.diffmusic_hip.abi_version()) != _lib.ABI_VERSION:
            raise RuntimeError(f"libdiffmusic_torch_ops.so was compiled against C-ABI version {int(torch.ops.diffmusic_hip.abi_version())}, "
                               f"this package expects {_lib.ABI_VERSION}: rebuild it with `python -m diffmusic_amd.build`")
        _loaded = True
    return torch.ops.diffmusic_hip


def enabled():
    """True when the facade should call torch.ops.diffmusic_hip.* (default), False for the ctypes binding: DMX_TORCH_OPS=0, or the
    op library failed to load (warned about once; the HIP library itself is still required)."""
    global _usable
    if not USE_TORCH_OPS:
        return False
    if os.environ.get("DMX_LIB_PATH"):
        # a dev A/B build of libdiffmusic_hip.so is selected for the ctypes binding; the op library is linked (rpath) against the
        # in-tree one, so going through it would silently measure the other library's kernels
        return False
    if _usable is None:
        try:
            load()
            _usable = True
        except (RuntimeError, OSError) as e:
            warnings.warn(f"torch.ops.diffmusic_hip is unavailable ({e}); using the ctypes binding of the same HIP entry points. "
                          "Rebuild with `python -m diffmusic_amd.build`.", RuntimeWarning, stacklevel=2)
            _usable = False
    return _usable


class _Hip:
    def __getattr__(self, name):
        return getattr(load(), name)


hip = _Hip()
