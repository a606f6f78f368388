"""deepfly3d_amd: MI355X-native back-end of DeepFly3D's per-frame hot path (7-view stacked-hourglass 2-D
heat-maps -> arg-max/confidence -> multi-view DLT / bundle adjustment -> df3d_result.pkl).

The arithmetic lives in libdf3d_hip.so (hand-written HIP for gfx950, C ABI in include/df3d_hip.h); this
package is the Python host mirroring the reference's `df3d.core.Core` / `df3d-cli` surface.
"""
__version__ = "0.1.0"
