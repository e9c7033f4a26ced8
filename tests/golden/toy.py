"""Tiny deterministic stand-ins for vae / vocoder used by the golden scheduler fixtures (both the
generator, which drives the REFERENCE step() with them, and the tests, which drive the oracle
with them).  Weights are closed-form so nothing has to be shipped."""
from types import SimpleNamespace
import math
import torch


class ToyVae:
    """(B,8,h,w) -> (B,1,4h,4w): nearest x4 of a fixed channel mix, through tanh."""
    config = SimpleNamespace(scaling_factor=0.9227914214134216)

    def __init__(self):
        self.mix = torch.tensor([math.cos(0.7 * i + 0.3) for i in range(8)], dtype=torch.float32)

    def decode(self, z):
        m = torch.tanh((z * self.mix.view(1, 8, 1, 1).to(z)).sum(1, keepdim=True))
        m = m.repeat_interleave(4, dim=2).repeat_interleave(4, dim=3)
        return SimpleNamespace(sample=m)


class ToyVocoder:
    """(B,T,M) mel -> (B, T*hop) waveform: per-frame sinusoid bank, smooth in the input."""

    def __init__(self, hop=160, n_mels=16):
        n = torch.arange(hop, dtype=torch.float32)
        k = torch.arange(n_mels, dtype=torch.float32)
        self.basis = torch.sin(2 * math.pi * (k[:, None] + 1.0) * (n[None] + 0.5) / hop) / n_mels  # (M, hop)

    def __call__(self, mel):
        B, T, M = mel.shape
        return torch.tanh(mel @ self.basis.to(mel)).reshape(B, T * self.basis.shape[1])
