"""The oracle's restatements of the OpenCV primitives against independent floating-point
implementations that are present in this image (torch, scipy, numpy).  These cannot pin bits -
OpenCV's 8-bit paths are fixed point - but they pin the semantics a restatement could get wrong
without any test here noticing: half-pixel sample positions and edge clamping of INTER_LINEAR,
the sigma / normalisation / REFLECT_101 border of the 7x7 Gaussian, the quadrant handling and the
accuracy class of fastAtan2.  Tolerances: one gray level for the resize (fixed-point rounding
against float); for the blur the 8-bit kernel's quantisation on top of that (its weights differ
from the float Gaussian by up to 0.0032, i.e. up to 3 gray levels on adversarial input, a fraction
of one on average); 0.3 degrees for the angle (OpenCV's documented accuracy)."""
import numpy as np
import pytest


def test_resize_is_half_pixel_bilinear(oracle):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(0)
    for sw, sh, dw, dh in [(752, 480, 627, 400), (627, 400, 522, 333), (120, 90, 100, 75), (100, 80, 50, 40),
                           (67, 67, 56, 56)]:
        smooth = rng.integers(0, 256, (sh // 8 + 2, sw // 8 + 2)).astype(np.float32)
        src = torch.nn.functional.interpolate(torch.from_numpy(smooth)[None, None], size=(sh, sw), mode="bicubic",
                                              align_corners=True)[0, 0].clamp(0, 255).round().numpy().astype(np.uint8)
        for img in (src, rng.integers(0, 256, (sh, sw), dtype=np.uint8)):
            got = oracle.resize(img, dw, dh).astype(np.int32)
            ref = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float64))[None, None], size=(dh, dw),
                                                  mode="bilinear", align_corners=False)[0, 0].numpy()
            assert np.abs(got - ref).max() <= 1.0, (sw, sh, dw, dh, np.abs(got - ref).max())


def test_blur_is_gaussian_sigma2_reflect101(oracle):
    ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(1)
    x = np.arange(-3, 4, dtype=np.float64)
    w = np.exp(-x * x / (2 * 2.0 * 2.0))
    w /= w.sum()
    # the 8.8 fixed-point kernel OpenCV derives for 8-bit images: weights * 256 rounded from the
    # outside in with the rounding error carried along, the centre taking what is left of 256
    fixed, err = [], 0.0
    for i in range(3):
        v = w[i] * 256 + err
        r = int(np.floor(v + 0.5))
        err = v - r
        fixed.append(r)
    fixed = fixed + [256 - 2 * sum(fixed)] + fixed[::-1]
    assert fixed == [18, 34, 48, 56, 48, 34, 18]
    impulse = np.zeros((15, 15), np.uint8)
    impulse[7, 7] = 255
    want = (np.outer(fixed, fixed) * 255 + 32768) >> 16
    assert np.array_equal(oracle.gauss7(impulse)[4:11, 4:11], want)
    for h, wd in [(60, 80), (33, 47), (120, 67)]:
        img = rng.integers(0, 256, (h, wd), dtype=np.uint8)
        got = oracle.gauss7(img).astype(np.float64)
        ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), w, axis=1, mode="mirror"), w, axis=0,
                                  mode="mirror")
        assert np.abs(got - ref).max() <= 3.0, np.abs(got - ref).max()
        assert np.abs(got - ref).mean() <= 0.5, np.abs(got - ref).mean()
        smooth = ndimage.uniform_filter(img.astype(np.float64), 9).astype(np.uint8)     # natural-image-like
        got = oracle.gauss7(smooth).astype(np.float64)
        ref = ndimage.correlate1d(ndimage.correlate1d(smooth.astype(np.float64), w, axis=1, mode="mirror"), w, axis=0,
                                  mode="mirror")
        assert np.abs(got - ref).max() <= 1.0, np.abs(got - ref).max()


def test_fast_atan2_is_atan2_in_degrees(oracle):
    rng = np.random.default_rng(2)
    ys = np.concatenate([rng.uniform(-3000, 3000, 4000), [0, 0, 1, -1, 5, -5, 0]]).astype(np.float32)
    xs = np.concatenate([rng.uniform(-3000, 3000, 4000), [1, -1, 0, 0, 5, -5, 0]]).astype(np.float32)
    worst = 0.0
    for y, x in zip(ys, xs):
        got = oracle.fast_atan2(float(y), float(x))
        assert 0.0 <= got <= 360.0
        ref = np.degrees(np.arctan2(float(y), float(x))) % 360.0 if (x != 0 or y != 0) else 0.0
        d = abs(got - ref)
        worst = max(worst, min(d, 360.0 - d))
    assert worst <= 0.3, worst


def test_fast_corners_satisfy_the_segment_test(oracle):
    """Every reported corner has 9 contiguous ring pixels all brighter than centre + t or all darker
    than centre - t, its score is the largest threshold for which that still holds, and no pixel
    that passes the test with a larger score than all its 8 neighbours is missing."""
    rng = np.random.default_rng(3)
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
            (-3, 0), (-3, 1), (-2, 2), (-1, 3)]

    def score(img, x, y):
        v = int(img[y, x])
        d = [v - int(img[y + dy, x + dx]) for dx, dy in ring]
        best = 0
        for k in range(16):
            arc = [d[(k + j) % 16] for j in range(9)]
            best = max(best, min(arc), -max(arc))
        return best            # corner at threshold t  <=>  best > t

    img = (rng.integers(0, 256, (40, 46)) // 24 * 24).astype(np.uint8)      # plateaus: plenty of ties
    for t in (7, 20, 40):
        got = {(int(x), int(y)): int(s) for x, y, s in oracle.fast(img, t).tolist()}
        sc = np.zeros(img.shape, np.int32)
        for y in range(3, img.shape[0] - 3):
            for x in range(3, img.shape[1] - 3):
                sc[y, x] = score(img, x, y)
        want = {}
        for y in range(3, img.shape[0] - 3):
            for x in range(3, img.shape[1] - 3):
                s = sc[y, x]
                if s > t and all(s > sc[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy):
                    want[(x, y)] = s - 1
        assert got == want, t
        assert len(got) > 5


def test_undistortion_inverts_the_radial_tangential_model(oracle):
    """Applying the forward distortion model to what UndistortKeyPoints returns gives back the
    observed pixel, as far as cv::undistortPoints' five fixed-point iterations converge: to 1e-3
    pixels in the inner part of the image, a few tenths of a pixel towards the corners of a camera
    as distorted as the reference's EuRoC example (the contraction factor there is about 0.5)."""
    cam = (458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = cam
    rng = np.random.default_rng(4)
    xy = np.stack([rng.uniform(0, 752, 2000), rng.uniform(0, 480, 2000)], 1).astype(np.float32)
    un = oracle.undistort_points(cam, xy).astype(np.float64)
    x, y = (un[:, 0] - cx) / fx, (un[:, 1] - cy) / fy
    r2 = x * x + y * y
    radial = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    back = np.stack([xd * fx + cx, yd * fy + cy], 1)
    err = np.abs(back - xy).max(axis=1)
    rd = np.hypot((xy[:, 0] - cx) / fx, (xy[:, 1] - cy) / fy)
    assert err[rd < 0.3].max() < 1e-3 and err[rd < 0.5].max() < 1e-2 and err.max() < 0.5, err.max()
    assert np.abs(un - xy).max() > 5.0            # and it did move points near the corners
