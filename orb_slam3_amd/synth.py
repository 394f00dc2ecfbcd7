"""Deterministic synthetic test frames (textured frames: host code in liborbx_hip.so;
right views of rectified stereo pairs: numpy)."""
import numpy as np

from . import _lib


def textured_frame(seed, width, height):
    """400 random gray rectangles + uniform noise + 3x3 box blur, PCG32-seeded."""
    img = np.empty((height, width), dtype=np.uint8)
    _lib.check(_lib.lib().orbx_synth_frame(int(seed) & 0xFFFFFFFF, width, height, width, _lib.ptr(img)),
               "orbx_synth_frame")
    return img


def right_view(left, seed=0, dmin=2.0, dmax=38.0, bands=6):
    """Right image of a rectified pair: row bands of `left` shifted left by a band-wise
    disparity that also ramps slowly along x (sub-pixel, linear interpolation), so matches see
    integer and fractional disparities; columns uncovered at the right edge repeat the border."""
    rng = np.random.default_rng(seed)
    h, w = left.shape
    out = np.empty_like(left)
    edges = np.linspace(0, h, bands + 1).astype(int)
    xs = np.arange(w, dtype=np.float64)
    for b in range(bands):
        d0 = rng.uniform(dmin, dmax)
        slope = rng.uniform(-0.004, 0.004)
        src = xs + d0 + slope * xs            # right(x) = left(x + d)
        src = np.clip(src, 0, w - 1)
        i0 = np.floor(src).astype(int)
        i1 = np.minimum(i0 + 1, w - 1)
        fr = src - i0
        rows = left[edges[b]:edges[b + 1]].astype(np.float64)
        out[edges[b]:edges[b + 1]] = np.rint(rows[:, i0] * (1 - fr) + rows[:, i1] * fr).astype(np.uint8)
    return out
