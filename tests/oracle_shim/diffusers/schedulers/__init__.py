from types import SimpleNamespace
from oracle.ddim import DDIMParent


class DDIMScheduler(DDIMParent):
    """diffusers-shaped facade over oracle.ddim.DDIMParent."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.config = SimpleNamespace(**self.cfg)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False,
             generator=None, variance_noise=None, return_dict=True):
        prev, x0 = self.parent_step(model_output, int(timestep), sample, eta, generator, variance_noise)
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)
