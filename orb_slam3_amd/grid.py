"""Host mirror of the Frame members that consume the keypoints right after extraction:
UndistortKeyPoints, ComputeImageBounds, AssignFeaturesToGrid, GetFeaturesInArea
(core/src/Frame.cc:771-829, 432-459, 682-762).  All device work runs in liborbx_hip.so."""
import ctypes as C

import numpy as np

from ._lib import KEYPOINT_DTYPE, OrbxPinhole, check, lib, ptr

GRID_COLS, GRID_ROWS = 64, 48
_f32p = C.POINTER(C.c_float)


def pinhole(fx, fy, cx, cy, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0):
    return OrbxPinhole(fx, fy, cx, cy, k1, k2, p1, p2, k3)


def undistort_points(cam, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.empty_like(xy)
    check(lib().orbx_undistort_points(C.byref(cam), xy.ctypes.data_as(_f32p), len(xy),
                                      out.ctypes.data_as(_f32p)), "orbx_undistort_points")
    return out


def image_bounds(cam, width, height):
    """(mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    b = np.zeros(4, np.float32)
    check(lib().orbx_image_bounds(C.byref(cam) if cam is not None else None, int(width), int(height),
                                  b.ctypes.data_as(_f32p)), "orbx_image_bounds")
    return b


def frame_grid_batch(ex, n_frames, cam, bounds):
    """Undistort + grid assignment for frames [0, n_frames) of the last batched extraction."""
    b = np.ascontiguousarray(bounds, np.float32)
    check(lib().orbx_frame_grid_batch_device(ex._h, int(n_frames), C.byref(cam) if cam is not None else None,
                                             b.ctypes.data_as(_f32p)), "orbx_frame_grid_batch_device")


def frame_grid_download(ex, frame):
    """(mvKeysUn, cell_start[64*48+1], cell_entries) of one frame; cell c = posX*48 + posY."""
    cap = ex.max_keypoints()
    k = np.zeros(cap, KEYPOINT_DTYPE)
    cs = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
    ent = np.zeros(cap, np.int32)
    n = C.c_int32()
    check(lib().orbx_frame_grid_download(ex._h, int(frame), ptr(k), ptr(cs), ptr(ent), cap, C.byref(n)),
          "orbx_frame_grid_download")
    return k[:n.value].copy(), cs, ent[:cs[-1]].copy()


def features_in_area(ex, frame, queries, per_query_cap=64):
    """queries: (Q, 5) float32 rows x, y, r, minLevel, maxLevel -> (counts[Q], indices[Q, cap])."""
    q = np.ascontiguousarray(queries, np.float32).reshape(-1, 5)
    idx = np.full((len(q), per_query_cap), -1, np.int32)
    cnt = np.zeros(len(q), np.int32)
    check(lib().orbx_features_in_area(ex._h, int(frame), q.ctypes.data_as(_f32p), len(q), ptr(idx),
                                      int(per_query_cap), ptr(cnt)), "orbx_features_in_area")
    return cnt, idx
