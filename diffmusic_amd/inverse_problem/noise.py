"""Measurement noise (reference: diffmusic/inverse_problem/noise.py)."""
import torch


class BaseNoise:
    def __call__(self, data):
        return self.forward(data)

    def forward(self, data):
        raise NotImplementedError


class GaussianNoise(BaseNoise):               # noise.py:13-18 (global RNG, like the reference)
    def __init__(self, sigma):
        self.sigma = sigma

    def forward(self, data):
        if self.sigma == 0:
            return data
        return data + torch.randn_like(data) * self.sigma


class PoissonNoise(BaseNoise):                # noise.py:21-39 (not selected by any config)
    def __init__(self, rate):
        self.rate = rate

    def forward(self, data):
        d = ((data + 1.0) / 2.0).clamp(0, 1)
        d = torch.poisson(d * 255.0 * self.rate) / 255.0 / self.rate
        return (d * 2.0 - 1.0).clamp(-1, 1)
