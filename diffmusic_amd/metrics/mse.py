"""Per-clip mean squared error between reference and generated waveforms (reference: diffmusic/metrics/mse.py:4-29)."""
import numpy as np


def _sanitize(a):
    return np.nan_to_num(np.asarray(a, dtype=np.float32), nan=0.0, posinf=1.0, neginf=-1.0)


class MeanSquaredError:
    def __init__(self, reduction="mean"):
        if reduction not in ("mean", "sum"):
            raise AssertionError("reduction must be 'mean' or 'sum'")
        self.reduction = reduction

    def score(self, audio_background, audio_eval):
        """Both arguments: sequences of 1-D clips (lengths may differ; each pair is compared on its common prefix).
        Two (B, L) CUDA tensors are scored on the GPU (CUDA scalar out)."""
        if getattr(audio_background, "is_cuda", False) and getattr(audio_eval, "is_cuda", False):
            import torch
            n = min(audio_background.shape[-1], audio_eval.shape[-1])
            ref = torch.nan_to_num(audio_background.float()[..., :n], nan=0.0, posinf=1.0, neginf=-1.0)
            est = torch.nan_to_num(audio_eval.float()[..., :n], nan=0.0, posinf=1.0, neginf=-1.0)
            per_clip = ((ref - est) ** 2).reshape(-1, n).mean(dim=1)
            return per_clip.mean() if self.reduction == "mean" else per_clip.sum()
        per_clip = []
        for ref, est in zip(audio_background, audio_eval):
            ref, est = _sanitize(ref), _sanitize(est)
            n = min(len(ref), len(est))
            per_clip.append(np.mean((ref[:n] - est[:n]) ** 2))
        per_clip = np.asarray(per_clip)
        return per_clip.mean() if self.reduction == "mean" else per_clip.sum()
