"""A second, independent restatement (numpy, written from the published algorithms, not from
oracle/orb_oracle.cpp) of the pixel-level stages of the ORB front-end.  Test infrastructure
only: tests/test_spec_crosscheck.py runs it beside the C++ oracle on small images so that a
slip in either restatement shows up as a difference.  Slow on purpose (clarity first).

Stage -> reference call site (paths relative to the ORB-SLAM3 tree):
  resize_linear_u8    cv::resize(INTER_LINEAR), core/lib/ORB/extractor.cc:1116
  reflect101_pad      cv::copyMakeBorder(BORDER_REFLECT_101), :1119-1124
  gaussian7           cv::GaussianBlur(7x7, sigma 2), :1064
  fast_cell           cv::FAST(roi, kps, threshold, true), :789, :806
  level_candidates    the cell loop of ComputeKeyPointsOctTree, :747-831
  ic_angle            IC_Angle + cv::fastAtan2, :75-99
  brief_descriptor    computeOrbDescriptor, :102-145
"""
import os

import numpy as np

EDGE = 19
F32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cv_round(v):
    """OpenCV's cvRound: round half to even."""
    return int(np.rint(v))


def level_tables(nlevels, scale_factor):
    scale = [F32(1.0)]
    sf = float(F32(scale_factor))          # the reference keeps the float argument in a double
    for _ in range(1, nlevels):
        scale.append(F32(float(scale[-1]) * sf))
    inv = [F32(1.0) / s for s in scale]
    return scale, inv


def level_size(W, H, inv_scale):
    return cv_round(F32(W) * inv_scale), cv_round(F32(H) * inv_scale)


# ---- cv::resize, INTER_LINEAR, 8-bit single channel ---------------------------------------------
def _axis_taps(sn, dn, horizontal):
    scale = 1.0 / (float(dn) / sn)
    out = []
    for d in range(dn):
        fx = F32((d + 0.5) * scale - 0.5)
        s = int(np.floor(fx))
        fx = F32(fx - F32(s))
        if horizontal:
            if s < 0:
                s, fx = 0, F32(0)
            if s >= sn - 1:
                s, fx = sn - 1, F32(0)
            s0, s1 = s, min(s + 1, sn - 1)
        else:
            s0, s1 = min(max(s, 0), sn - 1), min(max(s + 1, 0), sn - 1)
        w0 = int(np.clip(cv_round(F32(F32(1.0) - fx) * F32(2048)), -32768, 32767))
        w1 = int(np.clip(cv_round(fx * F32(2048)), -32768, 32767))
        out.append((s0, s1, w0, w1))
    return out


def resize_linear_u8(src, dw, dh):
    sh, sw = src.shape
    xt = _axis_taps(sw, dw, True)
    yt = _axis_taps(sh, dh, False)
    s = src.astype(np.int64)
    hrows = np.empty((sh, dw), np.int64)
    for dx, (s0, s1, a0, a1) in enumerate(xt):
        hrows[:, dx] = s[:, s0] * a0 + s[:, s1] * a1
    dst = np.empty((dh, dw), np.uint8)
    for dy, (r0, r1, b0, b1) in enumerate(yt):
        v = (((b0 * (hrows[r0] >> 4)) >> 16) + ((b1 * (hrows[r1] >> 4)) >> 16) + 2) >> 2
        dst[dy] = np.clip(v, 0, 255).astype(np.uint8)
    return dst


def reflect101_pad(img, border=EDGE):
    return np.pad(img, border, mode="reflect")      # numpy's "reflect" is gfedcb|abcdefgh|gfedcba


def build_pyramid(image, nlevels, scale_factor):
    """List of padded levels (19-pixel border), as ComputePyramid leaves them."""
    H, W = image.shape
    _, inv = level_tables(nlevels, scale_factor)
    levels, padded = [], []
    for l in range(nlevels):
        w, h = level_size(W, H, inv[l])
        lv = image.copy() if l == 0 else resize_linear_u8(levels[l - 1], w, h)
        levels.append(lv)
        padded.append(reflect101_pad(lv))
    return levels, padded


# ---- cv::GaussianBlur 7x7, sigma 2: OpenCV's fixed-point path -------------------------------------
GAUSS7 = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)   # 8.8 fixed point, sum 256


def gaussian7(img):
    p = np.pad(img, 3, mode="reflect").astype(np.int64)
    h, w = img.shape
    rows = sum(GAUSS7[i] * p[:, i:i + w] for i in range(7))            # (h+6, w), < 2^16
    cols = sum(GAUSS7[j] * rows[j:j + h, :] for j in range(7))
    return ((cols + 32768) >> 16).astype(np.uint8)


# ---- cv::FAST, TYPE_9_16, with non-max suppression -----------------------------------------------
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2),
        (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_strength(img):
    """m(x, y) = max over the 16 arcs of 9 contiguous ring pixels, and both signs, of the
    smallest |I_k - v| with all nine differences of that sign; 0 where no such arc exists.
    A pixel is a corner at threshold t iff m > t and its OpenCV score is m - 1.
    Defined for 3 <= x < cols-3, 3 <= y < rows-3; 0 elsewhere."""
    h, w = img.shape
    m = np.zeros((h, w), np.int32)
    v = img[3:h - 3, 3:w - 3].astype(np.int32)
    d = np.stack([v - img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int32) for dx, dy in RING])
    best = np.zeros_like(v)
    for k in range(16):
        idx = [(k + i) % 16 for i in range(9)]
        arc = d[idx]
        best = np.maximum(best, arc.min(axis=0))        # darker ring: all v - I_k > 0
        best = np.maximum(best, (-arc).min(axis=0))     # brighter ring
    m[3:h - 3, 3:w - 3] = np.maximum(best, 0)
    return m


def fast_cell(roi, threshold):
    """cv::FAST(roi, keypoints, threshold, nonmaxSuppression=true): [(x, y, score)], rows then x."""
    m = fast_strength(roi)
    score = np.where(m > threshold, m - 1, 0)            # non-corners count as 0 in the suppression
    out = []
    h, w = roi.shape
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if m[y, x] <= threshold:
                continue
            s = score[y, x]
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, int(s)))
    return out


def level_candidates(padded_level, w, h, ini_th, min_th):
    """The 35-pixel cell loop: candidates (x, y, score) relative to (16, 16), in emission order."""
    # minBorder = EDGE_THRESHOLD - 3, maxBorder = cols/rows - EDGE_THRESHOLD + 3, level coordinates
    min_bx, min_by = EDGE - 3, EDGE - 3
    max_bx, max_by = w - EDGE + 3, h - EDGE + 3
    width, height = F32(max_bx - min_bx), F32(max_by - min_by)
    n_cols, n_rows = int(width / F32(35)), int(height / F32(35))
    w_cell = int(np.ceil(width / F32(n_cols)))
    h_cell = int(np.ceil(height / F32(n_rows)))

    def region(x0, x1, y0, y1):
        return padded_level[EDGE + y0:EDGE + y1, EDGE + x0:EDGE + x1]

    out = []
    for i in range(n_rows):
        ini_y = min_by + i * h_cell
        max_y = ini_y + h_cell + 6
        if ini_y >= max_by - 3:
            continue
        max_y = min(max_y, max_by)
        for j in range(n_cols):
            ini_x = min_bx + j * w_cell
            max_x = ini_x + w_cell + 6
            if ini_x >= max_bx - 6:
                continue
            max_x = min(max_x, max_bx)
            roi = region(ini_x, max_x, ini_y, max_y)
            keys = fast_cell(roi, ini_th)
            if not keys:
                keys = fast_cell(roi, min_th)
            for x, y, s in keys:
                out.append((x + j * w_cell, y + i * h_cell, s))
    return out


# ---- IC_Angle + fastAtan2 ----------------------------------------------------------------------------
def umax_table():
    half = 15
    vmax = int(np.floor(half * np.sqrt(F32(2.0)) / 2 + 1))
    vmin = int(np.ceil(half * np.sqrt(F32(2.0)) / 2))
    um = [0] * (half + 1)
    for v in range(vmax + 1):
        um[v] = cv_round(np.sqrt(float(half * half - v * v)))
    v0 = 0
    for v in range(half, vmin - 1, -1):
        while um[v0] == um[v0 + 1]:
            v0 += 1
        um[v] = v0
        v0 += 1
    return um


def fast_atan2(y, x):
    """cv::fastAtan2 (degrees, [0, 360)), float arithmetic without fused multiply-add."""
    y, x = F32(y), F32(x)
    scale = F32(180.0 / np.pi)
    p1 = F32(0.9997878412794807) * scale
    p3 = F32(-0.3258083974640975) * scale
    p5 = F32(0.1555786518463281) * scale
    p7 = F32(-0.04432655554792128) * scale
    ax, ay = np.abs(x), np.abs(y)
    eps = F32(2.220446049250313e-16)
    if ax >= ay:
        c = ay / (ax + eps)
        c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps)
        c2 = c * c
        a = F32(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = F32(180.0) - a
    if y < 0:
        a = F32(360.0) - a
    return F32(a)


def ic_moments(padded_level, x, y):
    um = umax_table()
    cx, cy = EDGE + int(x), EDGE + int(y)
    m10 = m01 = 0
    for v in range(-15, 16):
        for u in range(-um[abs(v)], um[abs(v)] + 1):
            val = int(padded_level[cy + v, cx + u])
            m10 += u * val
            m01 += v * val
    return m10, m01


# ---- rotated BRIEF ---------------------------------------------------------------------------------------
def load_pattern():
    txt = open(os.path.join(ROOT, "oracle", "orb_pattern_31.inc")).read()
    txt = txt[txt.index("*/") + 2:]
    vals = [int(t) for t in txt.replace("\n", " ").split(",") if t.strip()]
    assert len(vals) == 1024
    return np.array(vals, np.int32).reshape(256, 4)


def brief_descriptor(blurred_level, x, y, angle_deg, pattern):
    """256 tests on the blurred level image around (x, y); sample offsets are rotated by the
    keypoint angle and rounded half-to-even; bit j of byte i is test 8i + j."""
    factor_pi = F32(np.pi / 180.0)
    rad = F32(angle_deg) * factor_pi
    a = F32(np.cos(float(rad)))
    b = F32(np.sin(float(rad)))
    cx, cy = int(x), int(y)
    desc = np.zeros(32, np.uint8)

    def sample(px, py):
        px, py = F32(px), F32(py)
        row = cv_round(px * b + py * a)
        col = cv_round(px * a - py * b)
        return int(blurred_level[cy + row, cx + col])

    for t in range(256):
        x0, y0, x1, y1 = pattern[t]
        if sample(x0, y0) < sample(x1, y1):
            desc[t >> 3] |= 1 << (t & 7)
    return desc


# ---- Frame::ComputeStereoMatches, core/src/Frame.cc:831-996 -----------------------------------------
def hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def c_round(v):
    """C round(): half away from zero, on a float."""
    v = float(v)
    return np.floor(v + 0.5) if v >= 0 else np.ceil(v - 0.5)


def stereo_matches(levels_l, levels_r, scale, inv_scale, kl, dl, kr, dr, mbf, mb):
    """levels_*: un-padded level images of the two cameras; kl/kr: keypoint records with fields
    x, y, octave; returns (uRight, depth) float32 arrays."""
    mbf, mb = F32(mbf), F32(mb)
    n = len(kl)
    u_right = np.full(n, -1, np.float32)
    depth = np.full(n, -1, np.float32)
    n_rows = levels_l[0].shape[0]
    rows = [[] for _ in range(n_rows)]
    for ir in range(len(kr)):
        r = F32(2.0) * scale[kr["octave"][ir]]
        hi = int(np.ceil(kr["y"][ir] + r))
        lo = int(np.floor(kr["y"][ir] - r))
        for yi in range(lo, hi + 1):
            if 0 <= yi < n_rows:
                rows[yi].append(ir)
    min_d, max_d = F32(0), mbf / mb
    dist_idx = []
    for il in range(n):
        lvl = int(kl["octave"][il])
        ul, vl = kl["x"][il], kl["y"][il]
        cand = rows[int(vl)] if 0 <= int(vl) < n_rows else []
        if not cand:
            continue
        min_u, max_u = ul - max_d, ul - min_d
        if max_u < 0:
            continue
        best, best_r = 100, 0
        for ir in cand:
            if kr["octave"][ir] < lvl - 1 or kr["octave"][ir] > lvl + 1:
                continue
            if min_u <= kr["x"][ir] <= max_u:
                d = hamming(dl[il], dr[ir])
                if d < best:
                    best, best_r = d, ir
        if best >= (100 + 50) // 2:
            continue
        sf = inv_scale[lvl]
        su = int(c_round(ul * sf))
        sv = int(c_round(vl * sf))
        sr = int(c_round(kr["x"][best_r] * sf))
        il_img, ir_img = levels_l[lvl], levels_r[lvl]
        h, w = il_img.shape
        if sv - 5 < 0 or sv + 6 > h or su - 5 < 0 or su + 6 > w:
            continue                     # cv::Mat::rowRange / colRange would throw
        if sr < 0 or sr + 11 >= w or sr - 10 < 0:
            continue
        win_l = il_img[sv - 5:sv + 6, su - 5:su + 6].astype(np.int32)
        dists = []
        for inc in range(-5, 6):
            win_r = ir_img[sv - 5:sv + 6, sr + inc - 5:sr + inc + 6].astype(np.int32)
            dists.append(F32(np.abs(win_l - win_r).sum()))
        best_inc = int(np.argmin(dists)) - 5          # first minimum
        best_sad = int(min(dists))
        if best_inc in (-5, 5):
            continue
        d1, d2, d3 = dists[5 + best_inc - 1], dists[5 + best_inc], dists[5 + best_inc + 1]
        delta = (d1 - d3) / (F32(2.0) * (d1 + d3 - F32(2.0) * d2))
        if delta < -1 or delta > 1:
            continue
        best_ur = scale[lvl] * (F32(sr) + F32(best_inc) + delta)
        disparity = ul - best_ur
        if min_d <= disparity < max_d:
            if disparity <= 0:
                disparity = F32(0.01)
                best_ur = F32(float(ul) - 0.01)
            depth[il] = mbf / disparity
            u_right[il] = best_ur
            dist_idx.append((best_sad, il))
    if dist_idx:
        dist_idx.sort()
        median = F32(dist_idx[len(dist_idx) // 2][0])
        th = F32(1.5) * F32(1.4) * median
        for sad, il in reversed(dist_idx):
            if F32(sad) < th:
                break
            u_right[il] = -1
            depth[il] = -1
    return u_right, depth


# ---- Frame::UndistortKeyPoints / AssignFeaturesToGrid / GetFeaturesInArea ---------------------------
def undistort_points(cam, xy):
    """cv::undistortPoints(src, dst, K, D, noArray(), K): five fixed-point iterations, double."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = [float(F32(c)) for c in cam]
    out = np.empty((len(xy), 2), np.float32)
    for i, (u, v) in enumerate(np.asarray(xy, np.float32)):
        u, v = float(u), float(v)
        x = x0 = (u - cx) * (1.0 / fx)
        y = y0 = (v - cy) * (1.0 / fy)
        for _ in range(5):
            r2 = x * x + y * y
            icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
            if icdist < 0:
                x, y = x0, y0
                break
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
        out[i] = (F32(fx * x + cx), F32(fy * y + cy))
    return out


def frame_grid(kps_un, bounds):
    """64 x 48 grid as a dict cell -> ascending keypoint indices, cell = posX * 48 + posY."""
    min_x, max_x, min_y, max_y = [F32(b) for b in bounds]
    inv_w, inv_h = F32(64) / (max_x - min_x), F32(48) / (max_y - min_y)
    grid = {}
    for i in range(len(kps_un)):
        px = int(c_round((kps_un["x"][i] - min_x) * inv_w))
        py = int(c_round((kps_un["y"][i] - min_y) * inv_h))
        if 0 <= px < 64 and 0 <= py < 48:
            grid.setdefault(px * 48 + py, []).append(i)
    return grid


def features_in_area(kps_un, grid, bounds, x, y, r, min_level, max_level):
    min_x, max_x, min_y, max_y = [F32(b) for b in bounds]
    inv_w, inv_h = F32(64) / (max_x - min_x), F32(48) / (max_y - min_y)
    x, y, r = F32(x), F32(y), F32(r)
    x_lo = max(0, int(np.floor((x - min_x - r) * inv_w)))
    x_hi = min(63, int(np.ceil((x - min_x + r) * inv_w)))
    y_lo = max(0, int(np.floor((y - min_y - r) * inv_h)))
    y_hi = min(47, int(np.ceil((y - min_y + r) * inv_h)))
    if x_lo >= 64 or x_hi < 0 or y_lo >= 48 or y_hi < 0:
        return []
    check = min_level > 0 or max_level >= 0
    out = []
    for ix in range(x_lo, x_hi + 1):
        for iy in range(y_lo, y_hi + 1):
            for i in grid.get(ix * 48 + iy, []):
                o = kps_un["octave"][i]
                if check and (o < min_level or (max_level >= 0 and o > max_level)):
                    continue
                if abs(kps_un["x"][i] - x) < r and abs(kps_un["y"][i] - y) < r:
                    out.append(i)
    return out


# ---- guided matching: ORBmatcher::SearchForInitialization / SearchByProjection(Frame, LastFrame) ----
HISTO_LENGTH = 30


def three_maxima(sizes):
    max1 = max2 = max3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(sizes):
        if s > max1:
            max3, max2, max1 = max2, max1, s
            i3, i2, i1 = i2, i1, i
        elif s > max2:
            max3, max2 = max2, s
            i3, i2 = i2, i
        elif s > max3:
            max3, i3 = s, i
    if max2 < F32(0.1) * F32(max1):
        i2 = i3 = -1
    elif max3 < F32(0.1) * F32(max1):
        i3 = -1
    return i1, i2, i3


def rotation_bin(a1, a2):
    rot = F32(a1) - F32(a2)
    if rot < 0:
        rot = rot + F32(360.0)
    b = int(c_round(rot * (F32(1.0) / F32(HISTO_LENGTH))))
    return 0 if b == HISTO_LENGTH else b


def search_for_initialization(k1, d1, k2, d2, grid2, bounds2, prev_matched, window, nnratio, check_orientation):
    big = 2 ** 31 - 1
    m12 = [-1] * len(k1)
    m21 = [-1] * len(k2)
    matched_dist = [big] * len(k2)
    hist = [[] for _ in range(HISTO_LENGTH)]
    nm = 0
    pm = np.array(prev_matched, np.float32).copy()
    for i1 in range(len(k1)):
        if k1["octave"][i1] > 0:
            continue
        cand = features_in_area(k2, grid2, bounds2, pm[i1, 0], pm[i1, 1], window, 0, 0)
        if not cand:
            continue
        best = best2 = big
        bi = -1
        for i2 in cand:
            dist = hamming(d1[i1], d2[i2])
            if matched_dist[i2] <= dist:
                continue
            if dist < best:
                best2, best, bi = best, dist, i2
            elif dist < best2:
                best2 = dist
        if best <= 50 and F32(best) < F32(best2) * F32(nnratio):
            if m21[bi] >= 0:
                m12[m21[bi]] = -1
                nm -= 1
            m12[i1], m21[bi], matched_dist[bi] = bi, i1, best
            nm += 1
            if check_orientation:
                hist[rotation_bin(k1["angle"][i1], k2["angle"][bi])].append(i1)
    if check_orientation:
        keep = three_maxima([len(h) for h in hist])
        for b in range(HISTO_LENGTH):
            if b in keep:
                continue
            for i1 in hist[b]:
                if m12[i1] >= 0:
                    m12[i1] = -1
                    nm -= 1
    for i1 in range(len(k1)):
        if m12[i1] >= 0:
            pm[i1] = (k2["x"][m12[i1]], k2["y"][m12[i1]])
    return nm, m12, pm


def search_by_projection_last(last_valid, uv, invz, last_octave, last_angle, last_desc, last_obs, k, d, u_right,
                              state, grid, bounds, scale, th, forward, backward, mbf, check_orientation):
    assigned = [-1] * len(k)
    blocked = [s == 1 for s in state]
    hist = [[] for _ in range(HISTO_LENGTH)]
    nm = 0
    for i in range(len(last_valid)):
        if not last_valid[i] or F32(invz[i]) < 0:
            continue
        u, v = F32(uv[i][0]), F32(uv[i][1])
        if u < bounds[0] or u > bounds[1] or v < bounds[2] or v > bounds[3]:
            continue
        o = int(last_octave[i])
        radius = F32(th) * scale[o]
        if forward:
            cand = features_in_area(k, grid, bounds, u, v, radius, o, -1)
        elif backward:
            cand = features_in_area(k, grid, bounds, u, v, radius, 0, o)
        else:
            cand = features_in_area(k, grid, bounds, u, v, radius, o - 1, o + 1)
        best, bi = 256, -1
        for i2 in cand:
            if blocked[i2]:
                continue
            if u_right is not None and u_right[i2] > 0:
                ur = u - F32(mbf) * F32(invz[i])
                if abs(ur - u_right[i2]) > radius:
                    continue
            dist = hamming(last_desc[i], d[i2])
            if dist < best:
                best, bi = dist, i2
        if bi >= 0 and best <= 100:
            assigned[bi] = i
            blocked[bi] = bool(last_obs[i])
            nm += 1
            if check_orientation:
                hist[rotation_bin(last_angle[i], k["angle"][bi])].append(bi)
    if check_orientation:
        keep = three_maxima([len(h) for h in hist])
        for b in range(HISTO_LENGTH):
            if b in keep:
                continue
            for i2 in hist[b]:
                assigned[i2] = -2
                nm -= 1
    return nm, assigned


def _point_candidates(in_view, proj, proj_xr, level, view_cos, pt_desc, k, d, u_right, grid, bounds, scale, th):
    """Per map point the keypoints of its window with their distances (matcher.cc:60-101), before
    anything that depends on other map points."""
    cand = []
    for i in range(len(in_view)):
        lst = []
        if in_view[i]:
            r = F32(2.5) if float(F32(view_cos[i])) > 0.998 else F32(4.0)
            if float(F32(th)) != 1.0:
                r = r * F32(th)
            o = int(level[i])
            radius = r * scale[o]
            for i2 in features_in_area(k, grid, bounds, F32(proj[i][0]), F32(proj[i][1]), radius, o - 1, o):
                if u_right is not None and u_right[i2] > 0:
                    if abs(F32(proj_xr[i]) - u_right[i2]) > radius:
                        continue
                lst.append((i2, hamming(pt_desc[i], d[i2]), int(k["octave"][i2])))
        cand.append(lst)
    return cand


def _point_choice(lst, is_blocked, nnratio):
    best = best2 = 256
    lvl = lvl2 = bi = -1
    for i2, dist, octave in lst:
        if is_blocked(i2):
            continue
        if dist < best:
            best2, best, lvl2, lvl, bi = best, dist, lvl, octave, i2
        elif dist < best2:
            lvl2, best2 = octave, dist
    if best <= 100 and not (lvl == lvl2 and F32(best) > F32(nnratio) * F32(best2)):
        return bi
    return -1


def search_by_projection_points(in_view, proj, proj_xr, level, view_cos, pt_desc, pt_obs, k, d, u_right, state,
                                grid, bounds, scale, th, nnratio):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...), matcher.cc:41-142, Nleft == -1."""
    cand = _point_candidates(in_view, proj, proj_xr, level, view_cos, pt_desc, k, d, u_right, grid, bounds, scale, th)
    assigned = [-1] * len(k)
    blocked = [s == 1 for s in state]
    nm = 0
    for i in range(len(in_view)):
        bi = _point_choice(cand[i], lambda i2: blocked[i2], nnratio)
        if bi >= 0:
            assigned[bi] = i
            blocked[bi] = bool(pt_obs[i])
            nm += 1
    return nm, assigned


def rounds_search_by_projection_points(in_view, proj, proj_xr, level, view_cos, pt_desc, pt_obs, k, d, u_right,
                                       state, grid, bounds, scale, th, nnratio):
    cand = _point_candidates(in_view, proj, proj_xr, level, view_cos, pt_desc, k, d, u_right, grid, bounds, scale, th)
    n_pts, n = len(in_view), len(k)
    inf = n_pts + 1
    block_time = [-1 if s == 1 else inf for s in state]
    choice = [-1] * n_pts
    rounds = 0
    while True:
        rounds += 1
        new_choice = [_point_choice(cand[i], lambda i2, i=i: block_time[i2] < i, nnratio) for i in range(n_pts)]
        new_block = [-1 if s == 1 else inf for s in state]
        for i, c in enumerate(new_choice):
            if c >= 0 and pt_obs[i]:
                new_block[c] = min(new_block[c], i)
        if new_choice == choice and new_block == block_time:
            break
        choice, block_time = new_choice, new_block
    assigned = [-1] * n
    nm = 0
    for i, c in enumerate(choice):
        if c >= 0:
            assigned[c] = i
            nm += 1
    return nm, assigned, rounds


def search_by_projection_last_rig(last_valid, uv_l, invz, uv_r, last_octave, last_angle, last_desc, last_obs, cam_l,
                                  cam_r, bounds, scale, th, forward, backward, check_orientation):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) for a current frame with two
    cameras (matcher.cc:1591-1808, Nleft != -1).  cam = (keypoints, descriptors, state, grid)."""
    (kl, dl, sl, gl), (kr, dr, sr, gr) = cam_l, cam_r
    asg = [[-1] * len(kl), [-1] * len(kr)]
    blocked = [[x == 1 for x in sl], [x == 1 for x in sr]]
    hist = [[] for _ in range(HISTO_LENGTH)]
    nm = 0
    for i in range(len(last_valid)):
        if not last_valid[i] or F32(invz[i]) < 0:
            continue
        u, v = F32(uv_l[i][0]), F32(uv_l[i][1])
        if u < bounds[0] or u > bounds[1] or v < bounds[2] or v > bounds[3]:
            continue
        o = int(last_octave[i])
        radius = F32(th) * scale[o]
        lo, hi = (o, -1) if forward else ((0, o) if backward else (o - 1, o + 1))
        cand = features_in_area(kl, gl, bounds, u, v, radius, lo, hi)
        if not cand:
            continue                     # :1652 leaves the loop body: no right camera either
        for cam, (k, d, c) in enumerate(((kl, dl, cand), (kr, dr, None))):
            if cam == 1:                 # :1712-1786; the right projection is not checked against the image
                c = features_in_area(kr, gr, bounds, F32(uv_r[i][0]), F32(uv_r[i][1]), radius, lo, hi)
            best, bi = 256, -1
            for i2 in c:
                if blocked[cam][i2]:
                    continue
                dist = hamming(last_desc[i], d[i2])
                if dist < best:
                    best, bi = dist, i2
            if bi >= 0 and best <= 100:
                asg[cam][bi] = i
                blocked[cam][bi] = bool(last_obs[i])
                nm += 1
                if check_orientation:
                    hist[rotation_bin(last_angle[i], k["angle"][bi])].append((cam, bi))
    if check_orientation:
        keep = three_maxima([len(h) for h in hist])
        for b in range(HISTO_LENGTH):
            if b in keep:
                continue
            for cam, i2 in hist[b]:
                asg[cam][i2] = -2
                nm -= 1
    return nm, asg[0], asg[1]


def _point_choice3(lst, is_blocked, nnratio):
    """_point_choice with the two ways of failing told apart: -1 nothing within TH_HIGH, -2 the
    ratio test failed (matcher.cc:123-124 leaves the map point's loop body)."""
    best = best2 = 256
    lvl = lvl2 = bi = -1
    for i2, dist, octave in lst:
        if is_blocked(i2):
            continue
        if dist < best:
            best2, best, lvl2, lvl, bi = best, dist, lvl, octave, i2
        elif dist < best2:
            lvl2, best2 = octave, dist
    if best > 100:
        return -1
    if lvl == lvl2 and F32(best) > F32(nnratio) * F32(best2):
        return -2
    return bi


def _rig_candidates(pts, cam_l, cam_r, bounds, scale, th):
    in_view, proj, level, view_cos, in_view_r, proj_r, level_r, view_cos_r, pt_desc = pts
    (kl, dl, _, gl), (kr, dr, _, gr) = cam_l, cam_r
    zero = np.zeros(len(in_view), np.float32)
    cl = _point_candidates(in_view, proj, zero, level, view_cos, pt_desc, kl, dl, None, gl, bounds, scale, th)
    view_r = [bool(in_view_r[i]) and int(level_r[i]) != -1 for i in range(len(in_view))]     # :144-146
    cr = _point_candidates(view_r, proj_r, zero, np.maximum(level_r, 0), view_cos_r, pt_desc, kr, dr, None, gr,
                           bounds, scale, 1.0)                                               # :147: no th
    return cl, cr


def search_by_projection_points_rig(pts, pt_obs, cam_l, cam_r, l2r, r2l, bounds, scale, th, nnratio):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...) for a frame with two cameras
    (matcher.cc:41-213).  pts = (in_view, proj, level, view_cos, in_view_r, proj_r, level_r,
    view_cos_r, desc); cam = (keypoints, descriptors, state, grid)."""
    cl, cr = _rig_candidates(pts, cam_l, cam_r, bounds, scale, th)
    asg = [[-1] * len(cam_l[0]), [-1] * len(cam_r[0])]
    blocked = [[x == 1 for x in cam_l[2]], [x == 1 for x in cam_r[2]]]
    partner = [l2r, r2l]
    nm = 0
    for i in range(len(pt_obs)):
        for cam, lst in ((0, cl[i]), (1, cr[i])):
            c = _point_choice3(lst, lambda i2: blocked[cam][i2], nnratio)
            if c == -2 and cam == 0:
                break
            if c >= 0:
                asg[cam][c] = i
                blocked[cam][c] = bool(pt_obs[i])
                nm += 1
                p = int(partner[cam][c])
                if p != -1:                                   # :129-135, :197-203: whatever it held
                    asg[1 - cam][p] = i
                    blocked[1 - cam][p] = bool(pt_obs[i])
                    nm += 1
    return nm, asg[0], asg[1]


def rounds_search_by_projection_points_rig(pts, pt_obs, cam_l, cam_r, l2r, r2l, bounds, scale, th, nnratio):
    """The same result from rounds of independent decisions.  Map point i decides for the left
    camera at time 2i and for the right one at 2i + 1; a decision writes the point into its
    keypoint and into the keypoint's partner in the other camera.  A keypoint is closed at time t
    when the latest write before t carried a point with observations (none yet: its initial
    state).  Every round decides from the writes of the previous round's decisions; a decision
    only depends on earlier times, so the rounds settle on the sequential loop's result."""
    cl, cr = _rig_candidates(pts, cam_l, cam_r, bounds, scale, th)
    cand = [cl, cr]
    n_pts = len(pt_obs)
    sizes = [len(cam_l[0]), len(cam_r[0])]
    init = [[x == 1 for x in cam_l[2]], [x == 1 for x in cam_r[2]]]
    partner = [l2r, r2l]
    choice = [[-1] * n_pts, [-1] * n_pts]
    rounds = 0

    def writes_of(choice):
        w = [[[] for _ in range(sizes[0])], [[] for _ in range(sizes[1])]]     # per keypoint: (time, point)
        for i in range(n_pts):
            for cam in (0, 1):
                c = choice[cam][i]
                if c >= 0:
                    w[cam][c].append((2 * i + cam, i))
                    p = int(partner[cam][c])
                    if p != -1:
                        w[1 - cam][p].append((2 * i + cam, i))
        return w

    while True:
        rounds += 1
        w = writes_of(choice)

        def closed(cam, i2, t):
            before = [e for e in w[cam][i2] if e[0] < t]
            return bool(pt_obs[max(before)[1]]) if before else init[cam][i2]

        new = [[-1] * n_pts, [-1] * n_pts]
        for i in range(n_pts):
            c0 = _point_choice3(cand[0][i], lambda i2: closed(0, i2, 2 * i), nnratio)
            new[0][i] = c0
            new[1][i] = -1 if c0 == -2 else _point_choice3(cand[1][i], lambda i2: closed(1, i2, 2 * i + 1), nnratio)
        if new == choice:
            break
        choice = new
    w = writes_of(choice)
    asg = [[max(e)[1] if e else -1 for e in w[cam]] for cam in (0, 1)]
    nm = sum(len(e) for cam in (0, 1) for e in w[cam])
    return nm, asg[0], asg[1], rounds


def search_by_bow(kf_valid, kf_angle, kf_desc, kf_fv, f_angle, f_desc, f_fv, nnratio, check_orientation):
    """ORBmatcher::SearchByBoW(pKF, F, ...), matcher.cc:222-405, Nleft == -1.  kf_fv / f_fv: dicts
    node id -> list of feature indices (the FeatureVector maps)."""
    matches = [-1] * len(f_angle)
    hist = [[] for _ in range(HISTO_LENGTH)]
    nm = 0
    for node in sorted(set(kf_fv) & set(f_fv)):
        for ikf in kf_fv[node]:
            if not kf_valid[ikf]:
                continue
            best = best2 = 256
            bi = -1
            for i2 in f_fv[node]:
                if matches[i2] >= 0:
                    continue
                dist = hamming(kf_desc[ikf], f_desc[i2])
                if dist < best:
                    best2, best, bi = best, dist, i2
                elif dist < best2:
                    best2 = dist
            if best <= 50 and F32(best) < F32(nnratio) * F32(best2):
                matches[bi] = ikf
                nm += 1
                if check_orientation:
                    hist[rotation_bin(kf_angle[ikf], f_angle[bi])].append(bi)
    if check_orientation:
        keep = three_maxima([len(h) for h in hist])
        for b in range(HISTO_LENGTH):
            if b not in keep:
                for i2 in hist[b]:
                    matches[i2] = -2
                    nm -= 1
    return nm, matches


# ---- the guided matchers as rounds of independent decisions (the form a GPU can run) ---------------
# Both reference loops are greedy over the query index.  What a query may take depends only on
# what queries with a LOWER index took, so the loop equals the unique fixed point of "every query
# decides from the claims of the lower-indexed queries of the previous round"; iterating from "no
# claims" reaches it after at most (longest dependency chain + 1) rounds.
def rounds_search_by_projection_last(last_valid, uv, invz, last_octave, last_angle, last_desc, last_obs, k, d,
                                     u_right, state, grid, bounds, scale, th, forward, backward, mbf,
                                     check_orientation):
    n_last, n = len(last_valid), len(k)
    inf = n_last + 1
    cand = []
    for i in range(n_last):                       # window walk + distances: independent per query
        lst = []
        u, v = F32(uv[i][0]), F32(uv[i][1])
        ok = last_valid[i] and not F32(invz[i]) < 0 and bounds[0] <= u <= bounds[1] and bounds[2] <= v <= bounds[3]
        if ok:
            o = int(last_octave[i])
            radius = F32(th) * scale[o]
            lo, hi = (o, -1) if forward else ((0, o) if backward else (o - 1, o + 1))
            for i2 in features_in_area(k, grid, bounds, u, v, radius, lo, hi):
                if u_right is not None and u_right[i2] > 0:
                    if abs((u - F32(mbf) * F32(invz[i])) - u_right[i2]) > radius:
                        continue
                lst.append((i2, hamming(last_desc[i], d[i2])))
        cand.append(lst)
    block_time = [0 if s == 1 else inf for s in state]   # index of the first blocking claim
    choice = [-1] * n_last
    rounds = 0
    while True:
        rounds += 1
        new_choice = []
        for i in range(n_last):
            best, bi = 256, -1
            for i2, dist in cand[i]:
                if block_time[i2] < i or (block_time[i2] == 0 and state[i2] == 1):
                    continue
                if dist < best:
                    best, bi = dist, i2
            new_choice.append(bi if best <= 100 else -1)
        new_block = [0 if s == 1 else inf for s in state]
        for i, c in enumerate(new_choice):
            if c >= 0 and last_obs[i]:
                new_block[c] = min(new_block[c], i)
        if new_choice == choice and new_block == block_time:
            break
        choice, block_time = new_choice, new_block
    assigned = [-1] * n
    nm = 0
    hist = [[] for _ in range(HISTO_LENGTH)]
    for i, c in enumerate(choice):
        if c >= 0:
            assigned[c] = i                      # ascending i: the last claim stays
            nm += 1
            if check_orientation:
                hist[rotation_bin(last_angle[i], k["angle"][c])].append(c)
    if check_orientation:
        keep = three_maxima([len(h) for h in hist])
        for b in range(HISTO_LENGTH):
            if b not in keep:
                for c in hist[b]:
                    assigned[c] = -2
                    nm -= 1
    return nm, assigned, rounds


def rounds_search_for_initialization(k1, d1, k2, d2, grid2, bounds2, prev_matched, window, nnratio,
                                     check_orientation):
    n1, n2 = len(k1), len(k2)
    big = 2 ** 31 - 1
    pm = np.array(prev_matched, np.float32)
    cand = []
    for i1 in range(n1):
        lst = []
        if k1["octave"][i1] <= 0:
            for i2 in features_in_area(k2, grid2, bounds2, pm[i1, 0], pm[i1, 1], window, 0, 0):
                lst.append((i2, hamming(d1[i1], d2[i2])))
        cand.append(lst)
    claims = [None] * n1                          # (keypoint, distance) accepted by query i1
    rounds = 0
    while True:
        rounds += 1
        # matched distance of a keypoint as query i1 sees it = smallest distance among the claims of
        # lower-indexed queries (every accepted claim undercuts the one before it)
        per_kp = [[] for _ in range(n2)]
        for j, c in enumerate(claims):
            if c is not None:
                per_kp[c[0]].append((j, c[1]))
        new_claims = []
        for i1 in range(n1):
            best = best2 = big
            bi = -1
            for i2, dist in cand[i1]:
                seen = min([dd for j, dd in per_kp[i2] if j < i1], default=big)
                if seen <= dist:
                    continue
                if dist < best:
                    best2, best, bi = best, dist, i2
                elif dist < best2:
                    best2 = dist
            ok = bi >= 0 and best <= 50 and F32(best) < F32(best2) * F32(nnratio)
            new_claims.append((bi, best) if ok else None)
        if new_claims == claims:
            break
        claims = new_claims
    m12 = [-1] * n1
    owner = {}
    hist = [[] for _ in range(HISTO_LENGTH)]
    for i1, c in enumerate(claims):
        if c is not None:
            owner[c[0]] = i1                      # ascending i1: the last claimant owns the keypoint
            if check_orientation:
                hist[rotation_bin(k1["angle"][i1], k2["angle"][c[0]])].append(i1)
    for kp, i1 in owner.items():
        m12[i1] = kp
    nm = len(owner)
    if check_orientation:
        keep = three_maxima([len(h) for h in hist])
        for b in range(HISTO_LENGTH):
            if b not in keep:
                for i1 in hist[b]:
                    if m12[i1] >= 0:
                        m12[i1] = -1
                        nm -= 1
    return nm, m12, rounds
