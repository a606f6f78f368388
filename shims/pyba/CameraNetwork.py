"""`from pyba.CameraNetwork import CameraNetwork` (reference df3d/core.py:12; call sites :120-126, :246-250, :355-360)."""
from deepfly3d_amd.camera_network import Camera, CameraNetwork  # noqa: F401
