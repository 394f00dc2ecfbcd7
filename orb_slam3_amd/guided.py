"""Host mirror of the guided matchers that walk the frame grid:
ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (core/lib/ORB/matcher.cc:
1591-1808), SearchByProjection(F, vpMapPoints, th) (:41-142) and SearchForInitialization (:629-733).  The current frame is a frame of
an ORBextractor's last batch for which grid.frame_grid_batch() has run; all device work runs
in liborbx_hip.so."""
import ctypes as C

import numpy as np

from ._lib import KEYPOINT_DTYPE, check, lib, ptr

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


def search_by_projection_last(ex, frame, last_valid, uv, invz, last_octave, last_angle, last_desc, last_obs,
                              u_right=None, state=None, th=15.0, forward=False, backward=False, mbf=0.0,
                              check_orientation=True):
    """-> (nmatches, assigned[n], rounds).  assigned[i2] = index of the last-frame keypoint whose
    map point current keypoint i2 received, -1 untouched, -2 cleared by the rotation check."""
    lv = np.ascontiguousarray(last_valid, np.uint8)
    n_last = len(lv)
    uvf = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    iz = np.ascontiguousarray(invz, np.float32)
    lo = np.ascontiguousarray(last_octave, np.int32)
    la = np.ascontiguousarray(last_angle, np.float32)
    ld = np.ascontiguousarray(last_desc, np.uint8).reshape(-1, 32)
    lobs = np.ascontiguousarray(last_obs, np.uint8)
    assert len(uvf) == len(iz) == len(lo) == len(la) == len(ld) == len(lobs) == n_last
    cap = ex.max_keypoints()
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    st = None if state is None else np.ascontiguousarray(state, np.uint8)
    out = np.zeros(cap, np.int32)
    n, nm, rounds = C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().orbx_search_by_projection_last(
        ex._h, int(frame), n_last, ptr(lv), uvf.ctypes.data_as(_f32p), iz.ctypes.data_as(_f32p),
        lo.ctypes.data_as(_i32p), la.ctypes.data_as(_f32p), ptr(ld), ptr(lobs), ptr(ur), ptr(st), float(th),
        int(bool(forward)), int(bool(backward)), float(mbf), int(bool(check_orientation)), ptr(out), cap,
        C.byref(n), C.byref(nm), C.byref(rounds)), "orbx_search_by_projection_last")
    if ur is not None:
        assert len(ur) >= n.value
    if st is not None:
        assert len(st) >= n.value
    return nm.value, out[:n.value].copy(), rounds.value


def search_by_projection_last_rig(left, right, frame, last_valid, uv_left, invz, uv_right, last_octave, last_angle,
                                  last_desc, last_obs, state_left=None, state_right=None, th=15.0, forward=False,
                                  backward=False, check_orientation=True):
    """The same matcher when the current frame has two cameras (CurrentFrame.Nleft != -1,
    matcher.cc:1591-1808 with the right camera's pass of :1712-1786) -> (nmatches,
    assigned_left[nL], assigned_right[nR], rounds).  left / right: the ORBextractors holding the
    two cameras' keypoints and grids for `frame`; the reference's index i2 >= Nleft is right
    keypoint i2 - Nleft."""
    lv = np.ascontiguousarray(last_valid, np.uint8)
    n_last = len(lv)
    ul = np.ascontiguousarray(uv_left, np.float32).reshape(-1, 2)
    ur = np.ascontiguousarray(uv_right, np.float32).reshape(-1, 2)
    iz = np.ascontiguousarray(invz, np.float32)
    lo = np.ascontiguousarray(last_octave, np.int32)
    la = np.ascontiguousarray(last_angle, np.float32)
    ld = np.ascontiguousarray(last_desc, np.uint8).reshape(-1, 32)
    lobs = np.ascontiguousarray(last_obs, np.uint8)
    assert len(ul) == len(ur) == len(iz) == len(lo) == len(la) == len(ld) == len(lobs) == n_last
    cap_l, cap_r = left.max_keypoints(), right.max_keypoints()
    sl = None if state_left is None else np.ascontiguousarray(state_left, np.uint8)
    sr = None if state_right is None else np.ascontiguousarray(state_right, np.uint8)
    out_l, out_r = np.zeros(cap_l, np.int32), np.zeros(cap_r, np.int32)
    nl, nr, nm, rounds = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().orbx_search_by_projection_last_rig(
        left._h, right._h, int(frame), n_last, ptr(lv), ul.ctypes.data_as(_f32p), iz.ctypes.data_as(_f32p),
        ur.ctypes.data_as(_f32p), lo.ctypes.data_as(_i32p), la.ctypes.data_as(_f32p), ptr(ld), ptr(lobs), ptr(sl),
        ptr(sr), float(th), int(bool(forward)), int(bool(backward)), int(bool(check_orientation)), ptr(out_l), cap_l,
        C.byref(nl), ptr(out_r), cap_r, C.byref(nr), C.byref(nm), C.byref(rounds)),
        "orbx_search_by_projection_last_rig")
    for st, n in ((sl, nl.value), (sr, nr.value)):
        if st is not None:
            assert len(st) >= n
    return nm.value, out_l[:nl.value].copy(), out_r[:nr.value].copy(), rounds.value


def search_by_projection_points_rig(left, right, frame, in_view, proj, level, view_cos, in_view_right, proj_right,
                                    level_right, view_cos_right, desc, obs, left_to_right, right_to_left,
                                    state_left=None, state_right=None, th=1.0, nnratio=0.8):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) for a frame with two cameras (F.Nleft != -1,
    matcher.cc:41-213) -> (nmatches, assigned_left[nL], assigned_right[nR], rounds).  left_to_right /
    right_to_left: mvLeftToRightMatch / mvRightToLeftMatch in each camera's own numbering."""
    iv = np.ascontiguousarray(in_view, np.uint8)
    n_pts = len(iv)
    pj = np.ascontiguousarray(proj, np.float32).reshape(-1, 2)
    lv = np.ascontiguousarray(level, np.int32)
    vc = np.ascontiguousarray(view_cos, np.float32)
    ivr = np.ascontiguousarray(in_view_right, np.uint8)
    pjr = np.ascontiguousarray(proj_right, np.float32).reshape(-1, 2)
    lvr = np.ascontiguousarray(level_right, np.int32)
    vcr = np.ascontiguousarray(view_cos_right, np.float32)
    pd = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    po = np.ascontiguousarray(obs, np.uint8)
    assert len(pj) == len(lv) == len(vc) == len(ivr) == len(pjr) == len(lvr) == len(vcr) == len(pd) == len(po) == n_pts
    l2r = np.ascontiguousarray(left_to_right, np.int32)
    r2l = np.ascontiguousarray(right_to_left, np.int32)
    cap_l, cap_r = left.max_keypoints(), right.max_keypoints()
    sl = None if state_left is None else np.ascontiguousarray(state_left, np.uint8)
    sr = None if state_right is None else np.ascontiguousarray(state_right, np.uint8)
    out_l, out_r = np.zeros(cap_l, np.int32), np.zeros(cap_r, np.int32)
    nl, nr, nm, rounds = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().orbx_search_by_projection_points_rig(
        left._h, right._h, int(frame), n_pts, ptr(iv), pj.ctypes.data_as(_f32p), lv.ctypes.data_as(_i32p),
        vc.ctypes.data_as(_f32p), ptr(ivr), pjr.ctypes.data_as(_f32p), lvr.ctypes.data_as(_i32p),
        vcr.ctypes.data_as(_f32p), ptr(pd), ptr(po), l2r.ctypes.data_as(_i32p) if len(l2r) else None,
        r2l.ctypes.data_as(_i32p) if len(r2l) else None, ptr(sl), ptr(sr), float(th), float(nnratio), ptr(out_l),
        cap_l, C.byref(nl), ptr(out_r), cap_r, C.byref(nr), C.byref(nm), C.byref(rounds)),
        "orbx_search_by_projection_points_rig")
    assert len(l2r) >= nl.value and len(r2l) >= nr.value
    return nm.value, out_l[:nl.value].copy(), out_r[:nr.value].copy(), rounds.value


def search_by_projection_points(ex, frame, in_view, proj, proj_xr, level, view_cos, desc, obs, u_right=None,
                                state=None, th=1.0, nnratio=0.8):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) (matcher.cc:41-142) -> (nmatches,
    assigned[n], rounds).  assigned[i2] = index of the map point keypoint i2 received, -1 = none."""
    iv = np.ascontiguousarray(in_view, np.uint8)
    n_pts = len(iv)
    pj = np.ascontiguousarray(proj, np.float32).reshape(-1, 2)
    xr = np.ascontiguousarray(proj_xr, np.float32)
    lv = np.ascontiguousarray(level, np.int32)
    vc = np.ascontiguousarray(view_cos, np.float32)
    pd = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    po = np.ascontiguousarray(obs, np.uint8)
    assert len(pj) == len(xr) == len(lv) == len(vc) == len(pd) == len(po) == n_pts
    cap = ex.max_keypoints()
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    st = None if state is None else np.ascontiguousarray(state, np.uint8)
    out = np.zeros(cap, np.int32)
    n, nm, rounds = C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().orbx_search_by_projection_points(
        ex._h, int(frame), n_pts, ptr(iv), pj.ctypes.data_as(_f32p), xr.ctypes.data_as(_f32p),
        lv.ctypes.data_as(_i32p), vc.ctypes.data_as(_f32p), ptr(pd), ptr(po), ptr(ur), ptr(st), float(th),
        float(nnratio), ptr(out), cap, C.byref(n), C.byref(nm), C.byref(rounds)), "orbx_search_by_projection_points")
    if ur is not None:
        assert len(ur) >= n.value
    if st is not None:
        assert len(st) >= n.value
    return nm.value, out[:n.value].copy(), rounds.value


def search_by_projection_keyframe(ex, frame, kf_valid, uv, level, kf_angle, kf_desc, has_point=None, th=10.0,
                                  orb_dist=100, check_orientation=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (matcher.cc:
    1810-1923) -> (nmatches, assigned[n], rounds)."""
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    uvf = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    lv = np.ascontiguousarray(level, np.int32)
    ka = np.ascontiguousarray(kf_angle, np.float32)
    kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
    assert len(uvf) == len(lv) == len(ka) == len(kd) == len(kv)
    hp = None if has_point is None else np.ascontiguousarray((np.asarray(has_point) != 0).astype(np.uint8))
    cap = ex.max_keypoints()
    out = np.zeros(cap, np.int32)
    n, nm, rounds = C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().orbx_search_by_projection_keyframe(
        ex._h, int(frame), len(kv), ptr(kv), uvf.ctypes.data_as(_f32p), lv.ctypes.data_as(_i32p),
        ka.ctypes.data_as(_f32p), ptr(kd), ptr(hp), float(th), int(orb_dist), int(bool(check_orientation)), ptr(out),
        cap, C.byref(n), C.byref(nm), C.byref(rounds)), "orbx_search_by_projection_keyframe")
    return nm.value, out[:n.value].copy(), rounds.value


def search_by_bow(ex, frame, kf_valid, kf_angle, kf_desc, kf_node_ids, kf_node_start, kf_features, nnratio=0.7,
                  check_orientation=True):
    """ORBmatcher::SearchByBoW(pKF, F, ...) (matcher.cc:222-405) -> (nmatches, matches[n]).  The
    frame's bag of words must have been computed (bow.transform_batch); the keyframe's feature
    vector is the CSR bow.download returns.  matches[i2] = keyframe keypoint, -1 none, -2 cleared."""
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    ka = np.ascontiguousarray(kf_angle, np.float32)
    kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
    assert len(kv) == len(ka) == len(kd)
    kn = np.ascontiguousarray(kf_node_ids, np.uint32)
    ks = np.ascontiguousarray(kf_node_start, np.int32)
    kf = np.ascontiguousarray(kf_features, np.uint32)
    assert len(ks) == len(kn) + 1 or len(kn) == 0
    cap = ex.max_keypoints()
    out = np.zeros(cap, np.int32)
    n, nm = C.c_int32(), C.c_int32()
    check(lib().orbx_search_by_bow(ex._h, int(frame), len(kv), ptr(kv), ka.ctypes.data_as(_f32p), ptr(kd), len(kn),
                                   ptr(kn), ks.ctypes.data_as(_i32p), ptr(kf), float(nnratio),
                                   int(bool(check_orientation)), ptr(out), cap, C.byref(n), C.byref(nm)),
          "orbx_search_by_bow")
    return nm.value, out[:n.value].copy()


def search_for_initialization(ex2, frame2, kps1_un, desc1, prev_matched, window_size=100, nnratio=0.9,
                              check_orientation=True):
    """-> (nmatches, matches12[n1], prev_matched[n1, 2] updated, rounds)."""
    k1 = np.ascontiguousarray(kps1_un, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    pm = np.ascontiguousarray(prev_matched, np.float32).reshape(-1, 2).copy()
    assert len(k1) == len(d1) == len(pm)
    m12 = np.full(max(len(k1), 1), -1, np.int32)
    nm, rounds = C.c_int32(), C.c_int32()
    check(lib().orbx_search_for_initialization(
        ex2._h, int(frame2), len(k1), ptr(k1), ptr(d1), pm.ctypes.data_as(_f32p), int(window_size),
        float(nnratio), int(bool(check_orientation)), ptr(m12), C.byref(nm), C.byref(rounds)),
        "orbx_search_for_initialization")
    return nm.value, m12[:len(k1)].copy(), pm, rounds.value
