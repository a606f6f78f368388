"""ctypes binding of libdf3d_hip.so (the C ABI declared in include/df3d_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, this module raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DF3D_LIB") or os.path.join(_HERE, "libdf3d_hip.so")  # DF3D_LIB: developer override (kernel A/B builds)

ABI_VERSION = 610
LSMR_AUTO, LSMR_BARRIERS, LSMR_LAUNCHES, LSMR_LOCAL, LSMR_ELEVEN = 0, 1, 2, 3, 11   # DF3D_LSMR_* of include/df3d_hip.h  # DF3D_ABI_VERSION of include/df3d_hip.h: the revision these prototypes were written against
DF3D_EINVAL = -1  # include/df3d_hip.h
DF3D_ENOSPC = -5
DF3D_EIO = -6
DF3D_DTYPE_F32 = 0
DF3D_DTYPE_BF16 = 1
DF3D_DTYPE_F16 = 2
DF3D_DTYPE_F32S = 3   # float32 storage, products as two-way IEEE-half splits on the 16-bit matrix pipe (include/df3d_hip.h)
# df3d_preprocess_u8 / df3d_hg_forward_u8: the resize rule (DF3D_RESIZE_* of include/df3d_hip.h)
RESIZE_MODES = {"bilinear": 0, "bilinear_align_corners": 1, "area": 2}


class NativeLibraryError(RuntimeError):
    pass


class BAProblem(Structure):
    _fields_ = [
        ("ncam", c_int),
        ("nobs", c_int),
        ("npts", c_int),
        ("intr4", c_void_p),
        ("obs_xy", c_void_p),
        ("cam_idx", c_void_p),
        ("pt_idx", c_void_p),
        ("pt_start", c_void_p),
        ("cam_perm", c_void_p),
        ("cam_start", c_void_p),
    ]


class HGParam(Structure):
    _fields_ = [
        ("name", c_char * 64),
        ("kind", c_int),
        ("taps", c_int),
        ("cin", c_int),
        ("cout", c_int),
        ("cin_pad", c_int),
        ("cout_pad", c_int),
        ("offset", c_size_t),
        ("count", c_size_t),
        ("kperm", c_int),
        ("reserved", c_int),
    ]


# name -> (restype, argtypes); mirrors include/df3d_hip.h one to one
PROTOTYPES = {
    "df3d_last_error": (c_char_p, []),
    "df3d_version": (c_int, []),
    "df3d_device_count": (c_int, []),
    "df3d_device_name": (c_int, [c_int, c_char_p, c_int]),
    "df3d_preprocess_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, POINTER(c_float), POINTER(c_float), c_int, c_void_p]),
    "df3d_read_files": (c_int, [POINTER(c_char_p), c_int, c_void_p, c_size_t, c_void_p, c_void_p, POINTER(c_size_t), c_int]),
    "df3d_jpeg_work_bytes": (c_size_t, [c_int, c_int, c_int, c_size_t]),
    "df3d_jpeg_decode_luma": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_uint, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "df3d_heatmap_argmax": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_heatmap_argmax_checked": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_relayout_19_to_38": (c_int, [c_void_p, POINTER(c_int), c_int, c_void_p, c_void_p]),
    "df3d_triangulate": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_triangulate_scaled": (c_int, [c_void_p, c_void_p, c_double, c_double, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_column_median": (c_int, [c_void_p, c_int, c_longlong, c_longlong, c_void_p, c_void_p]),
    "df3d_procrustes_work_doubles": (c_longlong, [c_longlong]),
    "df3d_procrustes": (c_int, [c_void_p, c_longlong, POINTER(c_double), POINTER(c_double), c_void_p, c_void_p, c_longlong, c_void_p]),
    "df3d_pose_normalize": (c_int, [c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_longlong, c_void_p]),
    "df3d_oneeuro_filter": (c_int, [c_void_p, c_longlong, c_int, c_double, c_double, c_double, c_double, c_longlong, c_double, c_void_p, c_void_p]),
    "df3d_ba_eval": (c_int, [POINTER(BAProblem), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_ba_colsq": (c_int, [POINTER(BAProblem), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_ba_matvec": (c_int, [POINTER(BAProblem), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_ba_rmatvec": (c_int, [POINTER(BAProblem), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_ba_lsmr_work_doubles": (c_size_t, [POINTER(BAProblem)]),
    "df3d_ba_lsmr": (
        c_int,
        [POINTER(BAProblem), c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_int,
         c_void_p, c_void_p, POINTER(c_double), c_void_p],
    ),
    "df3d_ba_lsmr_form": (
        c_int,
        [POINTER(BAProblem), c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_int,
         c_void_p, c_void_p, POINTER(c_double), c_void_p, c_int],
    ),
    "df3d_vec_dot": (c_int, [c_void_p, c_void_p, c_size_t, POINTER(c_double), c_void_p, c_void_p]),
    "df3d_vec_dots": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_size_t), POINTER(c_double), c_void_p, c_void_p]),
    "df3d_vec_pairnorm_sum": (c_int, [c_void_p, c_size_t, POINTER(c_double), c_void_p, c_void_p]),
    "df3d_vec_axpby": (c_int, [c_double, c_void_p, c_double, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_vec_mul": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_vec_absmax": (c_int, [c_void_p, c_size_t, POINTER(c_double), c_void_p, c_void_p]),
    "df3d_ba_update_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "df3d_ba_trf_subspace": (
        c_int,
        [POINTER(BAProblem), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, POINTER(c_double), c_void_p, c_int],
    ),
    "df3d_ba_trf_trial": (
        c_int,
        [POINTER(BAProblem), c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, POINTER(c_double), c_void_p],
    ),
    "df3d_ba_trf_linearize": (
        c_int,
        [POINTER(BAProblem), c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    ),
    "df3d_hg_create": (c_int, [c_int, c_int, POINTER(c_void_p)]),
    "df3d_hg_destroy": (None, [c_void_p]),
    "df3d_hg_set_input": (c_int, [c_void_p, c_int, c_int]),
    "df3d_hg_num_params": (c_int, [c_void_p]),
    "df3d_hg_param_desc": (c_int, [c_void_p, c_int, POINTER(HGParam)]),
    "df3d_hg_blob_floats": (c_size_t, [c_void_p]),
    "df3d_hg_lowp_bytes": (c_size_t, [c_void_p]),
    "df3d_hg_set_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_hg_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "df3d_hg_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "df3d_hg_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_hg_forward_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_hg_work": (c_int, [c_void_p, c_int, POINTER(c_double), POINTER(c_double)]),
    "df3d_hg_profile": (c_int, [c_void_p, c_int]),
    "df3d_hg_profile_count": (c_int, [c_void_p]),
    "df3d_hg_profile_read": (c_int, [c_void_p, c_int, c_char_p, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_int)]),
    "df3d_hg_profile_executed_flops": (c_int, [c_void_p, c_int, POINTER(c_double)]),
    "df3d_hg_step_m1_bytes": (c_double, [c_void_p, c_int, c_int]),
    "df3d_hg_num_steps": (c_int, [c_void_p]),
    "df3d_render_pose2d_grid": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, POINTER(c_int), c_int, POINTER(ctypes.c_ubyte), c_double, c_double, c_void_p, c_void_p]),
    "df3d_render_pose3d_panels": (c_int, [c_void_p, c_int, POINTER(c_int), c_int, POINTER(ctypes.c_ubyte), POINTER(c_double), c_double, c_double, c_int, c_double, c_void_p, c_void_p]),
    "df3d_resize_rgb": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "df3d_hg_step_desc": (c_int, [c_void_p, c_int, c_char_p, c_int, POINTER(c_int)]),
    "df3d_hg_forward_upto": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
}

_lib = None


def load():
    """Load the shared library once; raise NativeLibraryError if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m deepfly3d_amd.build` "
            "(hipcc --offload-arch=gfx950).  deepfly3d_amd has no CPU fallback."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    missing = []
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise NativeLibraryError(f"{LIB_PATH} lacks symbols {missing}; rebuild with `python -m deepfly3d_amd.build --force`")
    got = lib.df3d_version()
    if got != ABI_VERSION:   # a stale build (or a DF3D_LIB variant of another revision) would take arguments in the wrong slots
        raise NativeLibraryError(f"{LIB_PATH} has ABI revision {got}, these bindings were written for {ABI_VERSION}; "
                                 "rebuild with `python -m deepfly3d_amd.build --force`")
    _lib = lib
    return lib


def library_path():
    """Path of the shared library in use (DF3D_LIB or the in-tree build)."""
    return LIB_PATH


def check(rc, what=""):
    if rc != 0:
        msg = load().df3d_last_error()
        raise NativeLibraryError(f"{what or 'libdf3d_hip'} failed (rc={rc}): {msg.decode() if msg else ''}")


def require_gpu():
    """Fail loudly when no HIP device is visible (the product path never runs on the CPU)."""
    lib = load()
    n = lib.df3d_device_count()
    if n <= 0:
        raise NativeLibraryError("no HIP device visible: deepfly3d_amd computes only on MI355X (gfx950); there is no CPU path")
    return n
