"""diffmusic_amd: MI355X-native guided-diffusion sampling engine behind the DiffMusic surface
(get_scheduler / get_pipeline / BaseOperator).  The hot path is hand-written HIP (csrc/) reached
through the C ABI in include/diffmusic_hip.h; there is no CPU fallback."""
__version__ = "0.1.0"
