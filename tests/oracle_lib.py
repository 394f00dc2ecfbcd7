"""ctypes wrapper of the CPU oracle (oracle/liborb_oracle.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
               ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "liborb_oracle.so"))
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_level_dims.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_level_padded.restype = C.c_void_p
        L.orc_level_padded.argtypes = [C.c_void_p, C.c_int]
        L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_level_keys.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_level_blurred.restype = C.c_void_p
        L.orc_level_blurred.argtypes = [C.c_void_p, C.c_int]
        L.orc_cv_round.argtypes = [C.c_float]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int]
        L.orc_gauss7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_int]
        L.orc_sort_sized_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_hamming.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_stereo_band_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                            C.c_void_p, C.c_void_p]
        L.orc_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.orc_undistort_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_image_bounds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_frame_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
        L.orc_features_in_area.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                           C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_three_maxima.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_search_by_projection_last.argtypes = [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] + \
            [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_search_by_projection_points.argtypes = [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] + \
            [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_void_p]
        L.orc_search_by_projection_last_rig.argtypes = [C.c_void_p] * 8 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + \
            [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                             C.c_void_p]
        L.orc_search_by_projection_points_rig.argtypes = [C.c_void_p] * 10 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + \
            [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_stereo_from_rgbd.restype = None
        L.orc_stereo_from_rgbd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                           C.c_void_p]
        L.orc_search_by_projection_keyframe.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] + \
            [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.orc_search_by_bow.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int] + \
            [C.c_void_p] * 3 + [C.c_float, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


class Extractor:
    def __init__(self, nfeatures, scale, nlevels, ini, mn):
        self.h = lib().orc_create(nfeatures, scale, nlevels, ini, mn)
        assert self.h
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def tables(self):
        nl = self.nlevels
        sc, inv, s2, is2 = (np.zeros(nl, np.float32) for _ in range(4))
        q = np.zeros(nl, np.int32)
        um = np.zeros(16, np.int32)
        lib().orc_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(q), _p(um))
        return dict(scale=sc, inv=inv, sigma2=s2, inv_sigma2=is2, quota=q, umax=um)

    def extract(self, img, lap=(0, 0), cap=None):
        img = np.ascontiguousarray(img)
        h, w = img.shape
        cap = cap or (self.nfeatures * 2 + 256)
        kps = np.zeros(cap, KP)
        desc = np.zeros((cap, 32), np.uint8)
        n, mono = C.c_int(), C.c_int()
        st = lib().orc_extract(self.h, _p(img), w, h, img.strides[0], lap[0], lap[1], _p(kps), _p(desc),
                               cap, C.byref(n), C.byref(mono))
        if st == -3:
            return self.extract(img, lap, n.value)
        return st, mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    def level_dims(self, l):
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        assert lib().orc_level_dims(self.h, l, C.byref(w), C.byref(h), C.byref(s)) == 0
        return w.value, h.value, s.value

    def padded(self, l):
        w, h, s = self.level_dims(l)
        p = lib().orc_level_padded(self.h, l)
        buf = (C.c_uint8 * ((h + 38) * s)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(h + 38, s).copy()

    def candidates(self, l):
        n = lib().orc_level_candidates(self.h, l, None, 0)
        out = np.zeros((max(n, 1), 3), np.int32)
        lib().orc_level_candidates(self.h, l, _p(out), n)
        return out[:n]

    def keys(self, l):
        n = lib().orc_level_keys(self.h, l, None, None, 0)
        k = np.zeros(max(n, 1), KP)
        m = np.zeros((max(n, 1), 2), np.int32)
        lib().orc_level_keys(self.h, l, _p(k), _p(m), n)
        return k[:n], m[:n]

    def blurred(self, l):
        w, h, _ = self.level_dims(l)
        p = lib().orc_level_blurred(self.h, l)
        if not p:
            return None
        buf = (C.c_uint8 * (w * h)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(h, w).copy()


def cv_round(v):
    return lib().orc_cv_round(float(v))


def fast_atan2(y, x):
    return lib().orc_fast_atan2(float(y), float(x))


def resize(src, dw, dh):
    src = np.ascontiguousarray(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gauss7(src):
    src = np.ascontiguousarray(src)
    dst = np.zeros_like(src)
    lib().orc_gauss7(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast(img, thr):
    img = np.ascontiguousarray(img)
    cap = img.size // 4 + 16
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast(_p(img), img.shape[1], img.shape[0], img.strides[0], thr, _p(out), cap)
    return out[:n]


def octree(xys, minX, maxX, minY, maxY, N):
    xys = np.ascontiguousarray(xys, np.int32)
    sel = np.zeros(len(xys) + 16, np.int32)
    n = lib().orc_octree(_p(xys), len(xys), minX, maxX, minY, maxY, N, _p(sel), len(sel))
    return sel[:n]


def sort_sized_nodes(size, ulx):
    size = np.ascontiguousarray(size, np.int32)
    ulx = np.ascontiguousarray(ulx, np.int32)
    perm = np.zeros(len(size), np.int32)
    lib().orc_sort_sized_nodes(_p(size), _p(ulx), len(size), _p(perm))
    return perm


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_hamming(_p(a), _p(b))


def knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(q), 2), np.uint32)
    dist = np.zeros((len(q), 2), np.uint16)
    lib().orc_knn2(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx, dist


def stereo_band_match(kl, dl, kr, dr, scale_factors, n_rows, min_d, max_d):
    kl = np.ascontiguousarray(kl, KP)
    kr = np.ascontiguousarray(kr, KP)
    dl = np.ascontiguousarray(dl, np.uint8)
    dr = np.ascontiguousarray(dr, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    bi = np.zeros(len(kl), np.int32)
    bd = np.zeros(len(kl), np.int32)
    lib().orc_stereo_band_match(_p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), _p(sf), n_rows,
                                min_d, max_d, _p(bi), _p(bd))
    return bi, bd


def stereo_match(ex_l, ex_r, kl, dl, kr, dr, mbf, mb):
    """Frame::ComputeStereoMatches on the pyramids of the last extract calls of ex_l / ex_r."""
    kl = np.ascontiguousarray(kl, KP)
    kr = np.ascontiguousarray(kr, KP)
    dl = np.ascontiguousarray(dl, np.uint8)
    dr = np.ascontiguousarray(dr, np.uint8)
    n = len(kl)
    ur = np.zeros(max(n, 1), np.float32)
    dp = np.zeros(max(n, 1), np.float32)
    sad = np.zeros(max(n, 1), np.int32)
    st = lib().orc_stereo_match(ex_l.h, ex_r.h, _p(kl), _p(dl), n, _p(kr), _p(dr), len(kr), mbf, mb,
                                _p(ur), _p(dp), _p(sad))
    assert st == 0
    return ur[:n], dp[:n], sad[:n]


def _cam(cam):
    return None if cam is None else np.ascontiguousarray(cam, np.float32)


def undistort_points(cam, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.empty_like(xy)
    lib().orc_undistort_points(_p(_cam(cam)), _p(xy), len(xy), _p(out))
    return out


def image_bounds(cam, cols, rows):
    b = np.zeros(4, np.float32)
    c = _cam(cam)
    lib().orc_image_bounds(_p(c) if c is not None else None, cols, rows, _p(b))
    return b


def frame_grid(cam, bounds, kps):
    kps = np.ascontiguousarray(kps, KP)
    un = np.zeros(max(len(kps), 1), KP)
    cs = np.zeros(64 * 48 + 1, np.int32)
    ent = np.zeros(max(len(kps), 1), np.int32)
    c = _cam(cam)
    b = np.ascontiguousarray(bounds, np.float32)
    lib().orc_frame_grid(_p(c) if c is not None else None, _p(b), _p(kps), len(kps), _p(un), _p(cs), _p(ent))
    return un[:len(kps)], cs, ent[:cs[-1]]


def features_in_area(bounds, kps_un, cs, ent, x, y, r, min_level, max_level, cap=64):
    kps_un = np.ascontiguousarray(kps_un, KP)
    ent = np.ascontiguousarray(ent, np.int32)
    if len(ent) == 0:
        ent = np.zeros(1, np.int32)
    b = np.ascontiguousarray(bounds, np.float32)
    out = np.full(cap, -1, np.int32)
    n = lib().orc_features_in_area(_p(b), _p(kps_un), _p(cs), _p(ent), x, y, r, min_level, max_level, _p(out), cap)
    return n, out


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    ind = np.zeros(3, np.int32)
    lib().orc_three_maxima(_p(sizes), len(sizes), _p(ind))
    return tuple(int(v) for v in ind)


def search_for_initialization(k1, d1, k2, d2, bounds2, cs2, ent2, prev_matched, window, nnratio=0.9,
                              check_orientation=True):
    k1 = np.ascontiguousarray(k1, KP)
    k2 = np.ascontiguousarray(k2, KP)
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    ent2 = np.ascontiguousarray(ent2 if len(ent2) else np.zeros(1), np.int32)
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(max(len(k1), 1), np.int32)
    b = np.ascontiguousarray(bounds2, np.float32)
    n = lib().orc_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), _p(b), _p(cs2),
                                            _p(ent2), _p(pm), int(window), nnratio, int(check_orientation), _p(m12))
    return n, m12[:len(k1)], pm


def search_by_projection_last(last_valid, uv, invz, last_octave, last_angle, last_desc, last_obs, k, d, u_right,
                              state, bounds, cs, ent, scale_factors, th, forward, backward, mbf,
                              check_orientation=True):
    lv = np.ascontiguousarray(last_valid, np.uint8)
    uv = np.ascontiguousarray(uv, np.float32)
    iz = np.ascontiguousarray(invz, np.float32)
    lo = np.ascontiguousarray(last_octave, np.int32)
    la = np.ascontiguousarray(last_angle, np.float32)
    ld = np.ascontiguousarray(last_desc, np.uint8)
    lobs = np.ascontiguousarray(last_obs, np.uint8)
    k = np.ascontiguousarray(k, KP)
    d = np.ascontiguousarray(d, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    st = np.ascontiguousarray(state, np.uint8)
    ent = np.ascontiguousarray(ent if len(ent) else np.zeros(1), np.int32)
    b = np.ascontiguousarray(bounds, np.float32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.zeros(max(len(k), 1), np.int32)
    n = lib().orc_search_by_projection_last(_p(lv), _p(uv), _p(iz), _p(lo), _p(la), _p(ld), _p(lobs), len(lv),
                                            _p(k), _p(d), _p(ur) if ur is not None else None, _p(st), len(k),
                                            _p(b), _p(cs), _p(ent), _p(sf), th, int(forward), int(backward), mbf,
                                            int(check_orientation), _p(out))
    return n, out[:len(k)]


def search_by_projection_points(in_view, proj, proj_xr, level, view_cos, pt_desc, pt_obs, k, d, u_right, state,
                                bounds, cs, ent, scale_factors, th, nnratio):
    iv = np.ascontiguousarray(in_view, np.uint8)
    pj = np.ascontiguousarray(proj, np.float32)
    xr = np.ascontiguousarray(proj_xr, np.float32)
    lv = np.ascontiguousarray(level, np.int32)
    vc = np.ascontiguousarray(view_cos, np.float32)
    pd = np.ascontiguousarray(pt_desc, np.uint8)
    po = np.ascontiguousarray(pt_obs, np.uint8)
    k = np.ascontiguousarray(k, KP)
    d = np.ascontiguousarray(d, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    st = np.ascontiguousarray(state, np.uint8)
    ent = np.ascontiguousarray(ent if len(ent) else np.zeros(1), np.int32)
    b = np.ascontiguousarray(bounds, np.float32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.zeros(max(len(k), 1), np.int32)
    n = lib().orc_search_by_projection_points(_p(iv), _p(pj), _p(xr), _p(lv), _p(vc), _p(pd), _p(po), len(iv),
                                              _p(k), _p(d), _p(ur) if ur is not None else None, _p(st), len(k),
                                              _p(b), _p(cs), _p(ent), _p(sf), th, nnratio, _p(out))
    return n, out[:len(k)]


def _camera(k, d, state, cs, ent):
    k = np.ascontiguousarray(k, KP)
    d = np.ascontiguousarray(d if len(d) else np.zeros((1, 32)), np.uint8)
    st = np.ascontiguousarray(state if len(state) else np.zeros(1), np.uint8)
    cs = np.ascontiguousarray(cs, np.int32)
    ent = np.ascontiguousarray(ent if len(ent) else np.zeros(1), np.int32)
    return k, d, st, cs, ent


def search_by_projection_last_rig(last_valid, uv_left, invz, uv_right, last_octave, last_angle, last_desc, last_obs,
                                  left, right, bounds, scale_factors, th, forward, backward,
                                  check_orientation=True):
    """left / right: (keypoints, descriptors, state, cellStart, entries) of the two cameras."""
    lv = np.ascontiguousarray(last_valid, np.uint8)
    ul = np.ascontiguousarray(uv_left, np.float32)
    ur = np.ascontiguousarray(uv_right, np.float32)
    iz = np.ascontiguousarray(invz, np.float32)
    lo = np.ascontiguousarray(last_octave, np.int32)
    la = np.ascontiguousarray(last_angle, np.float32)
    ld = np.ascontiguousarray(last_desc, np.uint8)
    lobs = np.ascontiguousarray(last_obs, np.uint8)
    kl, dl, sl, csl, el = _camera(*left)
    kr, dr, sr, csr, er = _camera(*right)
    b = np.ascontiguousarray(bounds, np.float32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out_l = np.zeros(max(len(kl), 1), np.int32)
    out_r = np.zeros(max(len(kr), 1), np.int32)
    n = lib().orc_search_by_projection_last_rig(_p(lv), _p(ul), _p(iz), _p(ur), _p(lo), _p(la), _p(ld), _p(lobs),
                                                len(lv), _p(kl), _p(dl), _p(sl), len(kl), _p(csl), _p(el), _p(kr),
                                                _p(dr), _p(sr), len(kr), _p(csr), _p(er), _p(b), _p(sf), th,
                                                int(forward), int(backward), int(check_orientation), _p(out_l),
                                                _p(out_r))
    return n, out_l[:len(kl)], out_r[:len(kr)]


def search_by_projection_points_rig(in_view, proj, level, view_cos, in_view_r, proj_r, level_r, view_cos_r, pt_desc,
                                    pt_obs, left, right, left_to_right, right_to_left, bounds, scale_factors, th,
                                    nnratio):
    """left / right: (keypoints, descriptors, state, cellStart, entries) of the two cameras."""
    iv = np.ascontiguousarray(in_view, np.uint8)
    pj = np.ascontiguousarray(proj, np.float32)
    lv = np.ascontiguousarray(level, np.int32)
    vc = np.ascontiguousarray(view_cos, np.float32)
    ivr = np.ascontiguousarray(in_view_r, np.uint8)
    pjr = np.ascontiguousarray(proj_r, np.float32)
    lvr = np.ascontiguousarray(level_r, np.int32)
    vcr = np.ascontiguousarray(view_cos_r, np.float32)
    pd = np.ascontiguousarray(pt_desc, np.uint8)
    po = np.ascontiguousarray(pt_obs, np.uint8)
    kl, dl, sl, csl, el = _camera(*left)
    kr, dr, sr, csr, er = _camera(*right)
    l2r = np.ascontiguousarray(left_to_right if len(left_to_right) else -np.ones(1), np.int32)
    r2l = np.ascontiguousarray(right_to_left if len(right_to_left) else -np.ones(1), np.int32)
    b = np.ascontiguousarray(bounds, np.float32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out_l = np.zeros(max(len(kl), 1), np.int32)
    out_r = np.zeros(max(len(kr), 1), np.int32)
    n = lib().orc_search_by_projection_points_rig(_p(iv), _p(pj), _p(lv), _p(vc), _p(ivr), _p(pjr), _p(lvr), _p(vcr),
                                                  _p(pd), _p(po), len(iv), _p(kl), _p(dl), _p(sl), len(kl), _p(csl),
                                                  _p(el), _p(kr), _p(dr), _p(sr), len(kr), _p(csr), _p(er), _p(l2r),
                                                  _p(r2l), _p(b), _p(sf), th, nnratio, _p(out_l), _p(out_r))
    return n, out_l[:len(kl)], out_r[:len(kr)]


def stereo_from_rgbd(kps, kps_un, depth_img, mbf):
    k = np.ascontiguousarray(kps, KP)
    u = np.ascontiguousarray(kps_un, KP)
    dimg = np.ascontiguousarray(depth_img, np.float32)
    ur = np.zeros(max(len(k), 1), np.float32)
    dp = np.zeros(max(len(k), 1), np.float32)
    lib().orc_stereo_from_rgbd(_p(k), _p(u), len(k), _p(dimg), dimg.shape[1], mbf, _p(ur), _p(dp))
    return ur[:len(k)], dp[:len(k)]


def search_by_projection_keyframe(kf_valid, uv, level, kf_angle, kf_desc, k, d, has_point, bounds, cs, ent,
                                  scale_factors, th, orb_dist, check_orientation=True):
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    uv = np.ascontiguousarray(uv, np.float32)
    lv = np.ascontiguousarray(level, np.int32)
    ka = np.ascontiguousarray(kf_angle, np.float32)
    kd = np.ascontiguousarray(kf_desc, np.uint8)
    k = np.ascontiguousarray(k, KP)
    d = np.ascontiguousarray(d, np.uint8)
    hp = np.ascontiguousarray(has_point, np.uint8)
    ent = np.ascontiguousarray(ent if len(ent) else np.zeros(1), np.int32)
    b = np.ascontiguousarray(bounds, np.float32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.zeros(max(len(k), 1), np.int32)
    n = lib().orc_search_by_projection_keyframe(_p(kv), _p(uv), _p(lv), _p(ka), _p(kd), len(kv), _p(k), _p(d), _p(hp),
                                                len(k), _p(b), _p(cs), _p(ent), _p(sf), th, int(orb_dist),
                                                int(check_orientation), _p(out))
    return n, out[:len(k)]


def search_by_bow(kf_valid, kf_angle, kf_desc, kf_nodes, kf_start, kf_feats, f_angle, f_desc, f_nodes, f_start,
                  f_feats, nnratio=0.7, check_orientation=True):
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    ka = np.ascontiguousarray(kf_angle, np.float32)
    kd = np.ascontiguousarray(kf_desc, np.uint8)
    kn = np.ascontiguousarray(kf_nodes if len(kf_nodes) else np.zeros(1), np.uint32)
    ks = np.ascontiguousarray(kf_start, np.int32)
    kf = np.ascontiguousarray(kf_feats if len(kf_feats) else np.zeros(1), np.uint32)
    fa = np.ascontiguousarray(f_angle, np.float32)
    fd = np.ascontiguousarray(f_desc, np.uint8)
    fn = np.ascontiguousarray(f_nodes if len(f_nodes) else np.zeros(1), np.uint32)
    fs = np.ascontiguousarray(f_start, np.int32)
    ff = np.ascontiguousarray(f_feats if len(f_feats) else np.zeros(1), np.uint32)
    out = np.zeros(max(len(fa), 1), np.int32)
    n = lib().orc_search_by_bow(_p(kv), _p(ka), _p(kd), len(kf_nodes), _p(kn), _p(ks), _p(kf), _p(fa), _p(fd),
                                len(fa), len(f_nodes), _p(fn), _p(fs), _p(ff), nnratio, int(check_orientation),
                                _p(out))
    return n, out[:len(fa)]


class Vocabulary:
    """The oracle's vocabulary (text file or arrays) and its transform()."""

    def __init__(self, path=None, arrays=None):
        L = lib()
        L.orc_voc_create.restype = C.c_void_p
        L.orc_voc_load_text.restype = C.c_void_p
        if path is not None:
            self.h = L.orc_voc_load_text(str(path).encode())
        else:
            k, depth, parent, leaf, desc, weight, weighting = arrays
            self._keep = (np.ascontiguousarray(parent, np.int32), np.ascontiguousarray(leaf, np.uint8),
                          np.ascontiguousarray(desc, np.uint8), np.ascontiguousarray(weight, np.float64))
            self.h = L.orc_voc_create(int(k), int(depth), 0, int(weighting), len(self._keep[0]), _p(self._keep[0]),
                                      _p(self._keep[1]), _p(self._keep[2]), _p(self._keep[3]))
        if not self.h:
            raise ValueError("vocabulary rejected")
        self.h = C.c_void_p(self.h)

    def __del__(self):
        try:
            lib().orc_voc_destroy(self.h)
        except Exception:
            pass

    def info(self):
        v = [C.c_int() for _ in range(4)]
        lib().orc_voc_info(self.h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def transform(self, desc, levelsup=4):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = max(len(d), 1)
        wi, wv = np.zeros(n, np.uint32), np.zeros(n, np.float64)
        ni, ns, ft = np.zeros(n, np.uint32), np.zeros(n + 1, np.int32), np.zeros(n, np.uint32)
        nw, nn = C.c_int(), C.c_int()
        st = lib().orc_bow_transform(self.h, _p(d), len(d), int(levelsup), _p(wi), _p(wv), C.byref(nw), _p(ni),
                                     _p(ns), _p(ft), C.byref(nn))
        assert st == 0
        W, N = nw.value, nn.value
        return wi[:W], wv[:W], ni[:N], ns[:N + 1], ft[:ns[N]]
