"""ORBextractor: same constructor arguments, call arguments and results as the
reference class (core/lib/ORB/extractor.h:45-102), computed on the GPU."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KEYPOINT_DTYPE, OrbxParams, check, lib, ptr


class ORBextractor:
    """ORB_SLAM3::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST).

    `extractor(image, mask, vLappingArea)` returns (monoIndex, keypoints,
    descriptors): keypoints is a structured array with cv::KeyPoint's fields,
    descriptors an (n, 32) uint8 array (None when n == 0, as the reference
    releases the output Mat), monoIndex the reference's return value
    (extractor.cc:1100).  An empty image returns (-1, [], None) like
    extractor.cc:1023-1024.  The mask is ignored (extractor.h:56).
    """

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0, max_batch=1):
        self._h = C.c_void_p()
        self.nfeatures, self.nlevels = int(nfeatures), int(nlevels)
        self.iniThFAST, self.minThFAST = int(iniThFAST), int(minThFAST)
        self.max_batch = int(max_batch)
        prm = OrbxParams(self.nfeatures, float(scaleFactor), self.nlevels, self.iniThFAST,
                         self.minThFAST, int(device), self.max_batch)
        check(lib().orbx_create(C.byref(prm), C.byref(self._h)), "orbx_create")
        nl = self.nlevels
        self._scale = np.zeros(nl, np.float32)
        self._inv = np.zeros(nl, np.float32)
        self._sig2 = np.zeros(nl, np.float32)
        self._invsig2 = np.zeros(nl, np.float32)
        self._quota = np.zeros(nl, np.int32)
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        check(lib().orbx_get_tables(self._h, self._scale.ctypes.data_as(f32p), self._inv.ctypes.data_as(f32p),
                                    self._sig2.ctypes.data_as(f32p), self._invsig2.ctypes.data_as(f32p),
                                    self._quota.ctypes.data_as(i32p)))
        self._scaleFactor = float(np.float32(scaleFactor))
        self.mvImagePyramid = _PyramidView(self)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().orbx_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    # --- getters, extractor.h:61-73 -------------------------------------------------
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self._scaleFactor

    def GetScaleFactors(self):
        return self._scale.copy()

    def GetInverseScaleFactors(self):
        return self._inv.copy()

    def GetScaleSigmaSquares(self):
        return self._sig2.copy()

    def GetInverseScaleSigmaSquares(self):
        return self._invsig2.copy()

    @property
    def features_per_level(self):
        return self._quota.copy()

    def max_keypoints(self):
        return int(lib().orbx_max_keypoints(self._h))

    # --- operator() -----------------------------------------------------------------
    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        if image is None or image.size == 0:
            return -1, np.zeros(0, KEYPOINT_DTYPE), None
        assert image.dtype == np.uint8 and image.ndim == 2, "image must be CV_8UC1"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints()
        for _ in range(2):
            kps = np.zeros(cap, KEYPOINT_DTYPE)
            desc = np.zeros((cap, 32), np.uint8)
            n, mono = C.c_int32(0), C.c_int32(0)
            st = lib().orbx_extract(self._h, C.c_void_p(image.ctypes.data), w, h, image.strides[0],
                                    int(vLappingArea[0]), int(vLappingArea[1]), ptr(kps), ptr(desc), cap,
                                    C.byref(n), C.byref(mono))
            if st == _lib.ORBX_ERR_CAPACITY:
                cap = n.value
                continue
            break
        check(st, "orbx_extract")
        n = n.value
        return mono.value, kps[:n].copy(), (desc[:n].copy() if n > 0 else None)

    # --- batched entry points (no counterpart in the reference; same per-frame results)
    def extract_batch(self, images, vLappingArea=(0, 0)):
        """images: (B, H, W) uint8 host array -> list of (monoIndex, keypoints, descriptors)."""
        images = np.ascontiguousarray(images)
        assert images.dtype == np.uint8 and images.ndim == 3
        B, h, w = images.shape
        cap = self.max_keypoints()
        ptrs = (C.c_void_p * B)(*[images[i].ctypes.data for i in range(B)])
        kps = np.zeros((B, cap), KEYPOINT_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        mono = np.zeros(B, np.int32)
        i32p = C.POINTER(C.c_int32)
        st = lib().orbx_extract_batch(self._h, B, ptrs, w, h, w, int(vLappingArea[0]), int(vLappingArea[1]),
                                      ptr(kps), ptr(desc), cap, n.ctypes.data_as(i32p),
                                      mono.ctypes.data_as(i32p))
        if st == _lib.ORBX_ERR_CAPACITY:
            cap = int(n.max())
            kps = np.zeros((B, cap), KEYPOINT_DTYPE)
            desc = np.zeros((B, cap, 32), np.uint8)
            st = lib().orbx_extract_batch(self._h, B, ptrs, w, h, w, int(vLappingArea[0]),
                                          int(vLappingArea[1]), ptr(kps), ptr(desc), cap,
                                          n.ctypes.data_as(i32p), mono.ctypes.data_as(i32p))
        check(st, "orbx_extract_batch")
        return [(int(mono[f]), kps[f, :n[f]].copy(), desc[f, :n[f]].copy() if n[f] else None)
                for f in range(B)]

    def extract_batch_device(self, d_ptr, n_frames, width, height, stride=None, vLappingArea=(0, 0)):
        """Enqueue extraction of n_frames images resident in device memory at d_ptr
        (an integer device address, e.g. torch_tensor.data_ptr()).  Asynchronous."""
        check(lib().orbx_extract_batch_device(self._h, int(n_frames), C.c_void_p(int(d_ptr)), int(width),
                                              int(height), int(stride or width), int(vLappingArea[0]),
                                              int(vLappingArea[1])), "orbx_extract_batch_device")

    def sync(self):
        check(lib().orbx_sync(self._h), "orbx_sync")

    def set_pipelined(self, on):
        """Back-to-back equally sized batches as two independent pipelines (orbx_set_pipelined)."""
        check(lib().orbx_set_pipelined(self._h, 1 if on else 0), "orbx_set_pipelined")

    def stream(self):
        """The handle's hipStream_t as an integer."""
        return int(lib().orbx_stream(self._h) or 0)

    def download(self, frame):
        cap = self.max_keypoints()
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n, mono = C.c_int32(0), C.c_int32(0)
        check(lib().orbx_batch_download(self._h, int(frame), ptr(kps), ptr(desc), cap, C.byref(n),
                                        C.byref(mono)), "orbx_batch_download")
        n = n.value
        return mono.value, kps[:n].copy(), (desc[:n].copy() if n else None)

    def results_device(self):
        """(d_kps, d_desc, d_n, d_mono, cap): device addresses of the last batch's outputs."""
        a, b, c, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int32()
        check(lib().orbx_batch_results_device(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d),
                                              C.byref(cap)))
        return a.value, b.value, c.value, d.value, cap.value

    def set_profiling(self, on):
        check(lib().orbx_set_profiling(self._h, 1 if on else 0))

    def stage_times_ms(self):
        ms = np.zeros(_lib.NUM_STAGES, np.float32)
        check(lib().orbx_stage_times(self._h, ms.ctypes.data_as(C.POINTER(C.c_float)), _lib.NUM_STAGES),
              "orbx_stage_times")
        return dict(zip(_lib.STAGE_NAMES, ms.tolist()))

    # --- stage dumps ------------------------------------------------------------------
    def detect_launch_times(self, last_calls=1):
        """[(milliseconds, frames)] of the detection-kernel launches of the last `last_calls`
        batched calls (at most 32 are kept) in the normal schedule, oldest first."""
        cap = 4 * 32
        ms = (C.c_float * cap)()
        fr = (C.c_int32 * cap)()
        n = C.c_int32()
        check(lib().orbx_detect_launch_history(self._h, int(last_calls), ms, fr, cap, C.byref(n)),
              "orbx_detect_launch_history")
        return [(float(ms[i]), int(fr[i])) for i in range(n.value)]

    def level_dims(self, level):
        w, h, s = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().orbx_level_dims(self._h, level, C.byref(w), C.byref(h), C.byref(s)))
        return w.value, h.value, s.value

    def debug_padded(self, frame, level):
        w, h, s = self.level_dims(level)
        buf = np.zeros((h + 38, s), np.uint8)
        check(lib().orbx_debug_level_padded(self._h, frame, level, ptr(buf)))
        return buf[:, :w + 38]

    def debug_blurred(self, frame, level):
        w, h, _ = self.level_dims(level)
        buf = np.zeros((h, w), np.uint8)
        check(lib().orbx_debug_level_blurred(self._h, frame, level, ptr(buf)))
        return buf

    def _debug_list(self, fn, frame, level, cap):
        xys = np.zeros((cap, 3), np.int32)
        n = C.c_int32()
        st = fn(self._h, frame, level, ptr(xys), cap, C.byref(n))
        if st == _lib.ORBX_ERR_CAPACITY:
            return self._debug_list(fn, frame, level, n.value)
        check(st)
        return xys[:n.value].copy()

    def debug_candidates(self, frame, level):
        return self._debug_list(lib().orbx_debug_level_candidates, frame, level, 1 << 16)

    def debug_selected(self, frame, level):
        return self._debug_list(lib().orbx_debug_level_selected, frame, level, 1 << 14)


class _PyramidView:
    """mvImagePyramid (extractor.h:75): level images of the last single-frame call,
    each a view into its padded plane (19-pixel border), downloaded on first use."""

    def __init__(self, ex):
        self._ex = ex

    def __len__(self):
        return self._ex.nlevels

    def padded(self, level):
        p = C.c_void_p()
        w, h, s = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().orbx_pyramid(self._ex._h, level, C.byref(p), C.byref(w), C.byref(h), C.byref(s)),
              "orbx_pyramid")
        rows = h.value + 38
        buf = (C.c_uint8 * (rows * s.value)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(rows, s.value)[:, :w.value + 38]

    def __getitem__(self, level):
        pad = self.padded(level)
        return pad[19:-19, 19:-19]
