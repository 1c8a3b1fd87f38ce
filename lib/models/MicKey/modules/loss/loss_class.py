"""reference lib/models/MicKey/modules/loss/loss_class.py:9-560 (MetricPoseLoss) -> mickey_amd (GPU, HIP-backed)."""
from mickey_amd.train_ransac import MetricPoseLoss  # noqa: F401
