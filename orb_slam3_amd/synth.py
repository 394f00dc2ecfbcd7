"""Deterministic synthetic test frames (host code in liborbx_hip.so)."""
import numpy as np

from . import _lib


def textured_frame(seed, width, height):
    """400 random gray rectangles + uniform noise + 3x3 box blur, PCG32-seeded."""
    img = np.empty((height, width), dtype=np.uint8)
    _lib.check(_lib.lib().orbx_synth_frame(int(seed) & 0xFFFFFFFF, width, height, width, _lib.ptr(img)),
               "orbx_synth_frame")
    return img
