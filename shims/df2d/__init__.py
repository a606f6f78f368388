"""Shim: `df2d` as the reference imports it (df3d/core.py:11), backed by deepfly3d_amd."""
