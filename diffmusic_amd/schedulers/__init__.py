from .scheduling_guided import DDIMScheduler, DPSScheduler, MPGDScheduler, DSGScheduler, DiffMusicScheduler
from .utils import InverseProblemSchedulerOutput


def get_scheduler(scheduler_name):            # reference: diffmusic/schedulers/__init__.py:9-24
    table = {"ddim": DDIMScheduler, "dps": DPSScheduler, "mpgd": MPGDScheduler, "dsg": DSGScheduler,
             "diffmusic": DiffMusicScheduler}
    if scheduler_name == "ditto":
        raise NotImplementedError("DITTO back-propagates through the whole trajectory; out of scope of the per-step engine")
    if scheduler_name not in table:
        raise ValueError(f"Unknown scheduler: {scheduler_name}")
    return table[scheduler_name]
