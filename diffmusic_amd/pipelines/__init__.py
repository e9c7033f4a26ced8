from .pipeline_musicldm import MusicLDMPipeline, AudioPipelineOutput


def get_pipeline(pipeline_name):              # reference: diffmusic/pipelines/__init__.py:5-15
    if pipeline_name == "musicldm":
        return MusicLDMPipeline
    if pipeline_name == "audioldm2":
        raise NotImplementedError("AudioLDM2 (dual cross-attention U-Net) is the next row of the scope table")
    raise ValueError(f"Unknown pipeline: {pipeline_name}")
