"""Evaluation metrics adjacent to the hot path: log-spectral distance and MSE.
Host-side numpy like the reference's (diffmusic/metrics/lsd.py, diffmusic/metrics/mse.py) for lists / arrays, and a device leg for
CUDA tensors (LSD through the HIP STFT kernel); FAD / KL need downloaded embedding models and stay out of scope."""
from .lsd import LogSpectralDistance
from .mse import MeanSquaredError

__all__ = ["LogSpectralDistance", "MeanSquaredError"]
