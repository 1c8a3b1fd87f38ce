"""reference lib/models/MicKey/compute_pose.py:6-60 -> mickey_amd."""
from mickey_amd.model import MickeyRelativePose  # noqa: F401
