"""Counterparts of the callers either side of the hot path: checkpoint I/O, the evaluation harness'
post-processing and interpretation capture, the on-device input pipeline.  Same names, arguments and error behaviour
as the reference's `human_motion_prediction.environment` for the functions mirrored here."""
from .checkpoint import load_params_from_model_path, make_checkpoint, save_ckpt  # noqa: F401
from .evaluation import capture_interpretation, mpjpe_ms_table, save_interpretation  # noqa: F401
from .input_pipeline import DeviceAugmentation, DevicePrefetcher  # noqa: F401
