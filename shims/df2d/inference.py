"""`from df2d.inference import inference_folder` (reference df3d/core.py:11, call site :177-185)."""
from deepfly3d_amd.inference import inference_folder  # noqa: F401
