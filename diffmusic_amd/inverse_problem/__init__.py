from .noise import GaussianNoise, PoissonNoise
from .operator import (BaseOperator, IdentityOperator, MusicInpaintingOperator, PhaseRetrievalOperator,
                       SuperResolutionOperator, MusicDereverberationOperator, StyleGuidanceOperator)


def get_noiser(name, sigma):                  # reference: inverse_problem/__init__.py:4-11
    if name == "gaussian":
        return GaussianNoise(sigma)
    if name == "poisson":
        return PoissonNoise(sigma)
    raise ValueError(f"Unknown noise: {name}")
