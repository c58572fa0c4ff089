from . import losses, gate_crf_loss, ramps  # noqa: F401
