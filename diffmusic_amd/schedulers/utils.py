from dataclasses import dataclass
from typing import Optional
import torch


@dataclass
class InverseProblemSchedulerOutput:          # same fields as the reference (schedulers/utils.py:8-16)
    sample: Optional[torch.Tensor] = None
    prev_sample: torch.Tensor = None
    pred_original_sample: Optional[torch.Tensor] = None
    loss: Optional[torch.Tensor] = None
    encoder_hidden_states: Optional[torch.Tensor] = None
    encoder_hidden_states_1: Optional[torch.Tensor] = None
    init_latents: Optional[torch.Tensor] = None
