"""reference lib/models/builder.py:5-18 -> mickey_amd."""
from mickey_amd.model import build_model  # noqa: F401
