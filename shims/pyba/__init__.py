"""Shim: `pyba` as the reference imports it (df3d/core.py:12,110,311), backed by deepfly3d_amd."""
