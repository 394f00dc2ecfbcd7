"""ctypes binding of liborbx_hip.so (the C ABI declared in include/orbx.h).

The library is the product: there is no CPU fallback.  Loading fails loudly
when the shared object is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbx_hip.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

ORBX_OK = 0
ORBX_ERR_EMPTY_IMAGE = -1
ORBX_ERR_TOO_SMALL = -2
ORBX_ERR_CAPACITY = -3
ORBX_ERR_BAD_ARG = -4
ORBX_ERR_NO_DEVICE = -5
ORBX_ERR_HIP = -6
ORBX_ERR_UNSUPPORTED = -7
ORBX_ERR_BATCH = -8
ORBX_ERR_SCRATCH = -9
ORBX_ERR_COMM = -10
NUM_STAGES = 6
STAGE_NAMES = ("pyramid", "fast", "octree", "assemble", "blur", "describe")


class OrbxParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("device_id", C.c_int32),
                ("max_batch", C.c_int32)]


class OrbxPinhole(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


class OrbxError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().orbx_strerror(status).decode()
        super().__init__("orbx: %s (status %d)%s" % (msg, status, (" in " + what) if what else ""))


_P = C.c_void_p
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)

# name -> (restype, argtypes); this table is also what the symbol test checks
# against include/orbx.h.
SIGNATURES = {
    "orbx_strerror": (C.c_char_p, [C.c_int]),
    "orbx_create": (C.c_int, [C.POINTER(OrbxParams), C.POINTER(_P)]),
    "orbx_destroy": (None, [_P]),
    "orbx_get_tables": (C.c_int, [_P, _f32p, _f32p, _f32p, _f32p, _i32p]),
    "orbx_max_keypoints": (C.c_int, [_P]),
    "orbx_extract": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int,
                               _i32p, _i32p]),
    "orbx_pyramid": (C.c_int, [_P, C.c_int, C.POINTER(_P), _i32p, _i32p, _i32p]),
    "orbx_extract_batch": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, _P, _P, C.c_int, _i32p, _i32p]),
    "orbx_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "orbx_host_free": (C.c_int, [_P]),
    "orbx_extract_batch_device": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "orbx_sync": (C.c_int, [_P]),
    "orbx_set_pipelined": (C.c_int, [_P, C.c_int]),
    "orbx_batch_results_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                            C.POINTER(_P), _i32p]),
    "orbx_batch_download": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _i32p, _i32p]),
    "orbx_stream": (_P, [_P]),
    "orbx_level_dims": (C.c_int, [_P, C.c_int, _i32p, _i32p, _i32p]),
    "orbx_debug_level_padded": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "orbx_debug_level_candidates": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _i32p]),
    "orbx_debug_level_selected": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _i32p]),
    "orbx_debug_level_blurred": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "orbx_debug_sort_records": (C.c_int, [C.c_int, _P, C.c_int]),
    "orbx_debug_sort_records_team": (C.c_int, [C.c_int, _P, C.c_int]),
    "orbx_set_profiling": (C.c_int, [_P, C.c_int]),
    "orbx_stage_times": (C.c_int, [_P, _f32p, C.c_int]),
    "orbx_detect_launch_times": (C.c_int, [_P, _f32p, _i32p, C.c_int, _i32p]),
    "orbx_detect_launch_history": (C.c_int, [_P, C.c_int, _f32p, _i32p, C.c_int, _i32p]),
    "orbx_hamming": (C.c_int, [_P, _P]),
    "orbx_knn2": (C.c_int, [C.c_int, _P, C.c_int, _P, C.c_int64, _P, _P]),
    "orbx_knn2_device": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_uint32, _P, _P, _P]),
    "orbx_knn2_device_noalloc": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_uint32, _P, _P, _P]),
    "orbx_knn2_noalloc_scratch_bytes": (C.c_size_t, []),
    "orbx_knn2_sharded": (C.c_int, [_P, _P, C.c_int, _P, C.c_int64, C.c_uint32, _P, _P, _P, C.c_size_t, _P]),
    "orbx_knn2_sharded_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "orbx_rccl_unique_id": (C.c_int, [_P]),
    "orbx_rccl_comm_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "orbx_rccl_comm_destroy": (C.c_int, [_P]),
    "orbx_top2_merge_device": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "orbx_stereo_band_match": (C.c_int, [C.c_int, _P, _P, C.c_int, _P, _P, C.c_int, _f32p, C.c_int,
                                         C.c_int, C.c_float, C.c_float, _P, _P]),
    "orbx_stereo_match_batch_device": (C.c_int, [_P, _P, C.c_int, C.c_float, C.c_float]),
    "orbx_stereo_results_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), _i32p]),
    "orbx_stereo_download": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _i32p]),
    "orbx_stereo_match": (C.c_int, [_P, _P, C.c_float, C.c_float, _P, _P, C.c_int, _i32p]),
    "orbx_undistort_points": (C.c_int, [C.POINTER(OrbxPinhole), _f32p, C.c_int, _f32p]),
    "orbx_image_bounds": (C.c_int, [C.POINTER(OrbxPinhole), C.c_int, C.c_int, _f32p]),
    "orbx_frame_grid_batch_device": (C.c_int, [_P, C.c_int, C.POINTER(OrbxPinhole), _f32p]),
    "orbx_frame_grid_results_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _i32p]),
    "orbx_frame_grid_download": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _i32p]),
    "orbx_features_in_area": (C.c_int, [_P, C.c_int, _f32p, C.c_int, _P, C.c_int, _P]),
    "orbx_search_by_projection_last": (C.c_int, [_P, C.c_int, C.c_int, _P, _f32p, _f32p, _i32p, _f32p, _P, _P, _P, _P,
                                                 C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, _P, C.c_int,
                                                 _i32p, _i32p, _i32p]),
    "orbx_search_by_projection_last_rig": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _f32p, _f32p, _f32p, _i32p, _f32p, _P, _P,
                                                     _P, _P, C.c_float, C.c_int, C.c_int, C.c_int, _P, C.c_int, _i32p,
                                                     _P, C.c_int, _i32p, _i32p, _i32p]),
    "orbx_search_by_projection_points_rig": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _f32p, _i32p, _f32p, _P, _f32p, _i32p,
                                                       _f32p, _P, _P, _i32p, _i32p, _P, _P, C.c_float, C.c_float, _P,
                                                       C.c_int, _i32p, _P, C.c_int, _i32p, _i32p, _i32p]),
    "orbx_search_by_projection_points": (C.c_int, [_P, C.c_int, C.c_int, _P, _f32p, _f32p, _i32p, _f32p, _P, _P, _P, _P,
                                                   C.c_float, C.c_float, _P, C.c_int, _i32p, _i32p, _i32p]),
    "orbx_lapping_knn2": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _i32p, _i32p, _i32p, _i32p]),
    "orbx_lapping_knn2_batch_device": (C.c_int, [_P, _P, C.c_int]),
    "orbx_lapping_knn2_download": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _i32p, _i32p, _i32p, _i32p]),
    "orbx_ratio_test_device": (C.c_int, [_P, _P, C.c_int, C.c_double, _P, _P]),
    "orbx_lapping_ratio_test_batch_device": (C.c_int, [_P, C.c_double]),
    "orbx_lapping_matches_device": (C.c_int, [_P, C.POINTER(_P), _i32p]),
    "orbx_lapping_matches_download": (C.c_int, [_P, C.c_int, _P, C.c_int, _i32p, _i32p]),
    "orbx_stereo_from_rgbd_batch_device": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_float]),
    "orbx_search_by_projection_keyframe": (C.c_int, [_P, C.c_int, C.c_int, _P, _f32p, _i32p, _f32p, _P, _P, C.c_float,
                                                     C.c_int, C.c_int, _P, C.c_int, _i32p, _i32p, _i32p]),
    "orbx_search_by_bow": (C.c_int, [_P, C.c_int, C.c_int, _P, _f32p, _P, C.c_int, _P, _i32p, _P, C.c_float, C.c_int,
                                     _P, C.c_int, _i32p, _i32p]),
    "orbx_search_for_initialization": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _f32p, C.c_int, C.c_float,
                                                 C.c_int, _P, _i32p, _i32p]),
    "orbx_vocabulary_load_text": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "orbx_vocabulary_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _P, _P, _P, C.c_int,
                                         C.POINTER(_P)]),
    "orbx_vocabulary_destroy": (None, [_P]),
    "orbx_vocabulary_info": (C.c_int, [_P, _i32p, _i32p, _i32p, _i32p]),
    "orbx_bow_transform_batch_device": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "orbx_bow_download": (C.c_int, [_P, C.c_int, _P, _P, _i32p, _P, _P, _P, _i32p, C.c_int]),
    "orbx_synth_frame": (C.c_int, [C.c_uint32, C.c_int, C.c_int, C.c_int, _P]),
}

_lib = None


def lib():
    """Load liborbx_hip.so once; raise if it is not there (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "orb_slam3_amd: %s is missing - build it with "
                "`make -C orb_slam3_amd/csrc` (or __graft_entry__.build()); there is no CPU fallback"
                % LIB_PATH)
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status, what=""):
    if status != ORBX_OK:
        raise OrbxError(status, what)


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
