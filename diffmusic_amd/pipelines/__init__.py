from .pipeline_musicldm import MusicLDMPipeline, AudioPipelineOutput
from .pipeline_audioldm2 import AudioLDM2Pipeline


def get_pipeline(pipeline_name):              # reference: diffmusic/pipelines/__init__.py:5-15
    if pipeline_name == "musicldm":
        return MusicLDMPipeline
    if pipeline_name == "audioldm2":
        return AudioLDM2Pipeline
    raise ValueError(f"Unknown pipeline: {pipeline_name}")
