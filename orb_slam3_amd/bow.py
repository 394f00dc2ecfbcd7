"""Host mirror of the reference's vocabulary and of Frame::ComputeBoW
(core/lib/ORB/vocabulary.h:31-62, core/src/Frame.cc:764-769): the vocabulary tree lives in
device memory, the transform runs in liborbx_hip.so on the descriptors of an ORBextractor's
last batch."""
import ctypes as C

import numpy as np

from ._lib import check, lib, ptr

_i32p = C.POINTER(C.c_int32)
TF_IDF, TF = 0, 1      # DBoW2::WeightingType
L1_NORM = 0            # DBoW2::ScoringType


class ORBVocabulary:
    """ORBVocabulary(filename) loads the text format of ORBVocabulary::loadFromText
    (core/lib/ORB/vocabulary.cc:96-160); ORBVocabulary.from_arrays(...) takes the same
    content as arrays (nodes 1..n in file order: parent id, leaf flag, descriptor, weight)."""

    def __init__(self, filename=None, device=0, _handle=None):
        self._h = _handle
        if filename is not None:
            h = C.c_void_p()
            check(lib().orbx_vocabulary_load_text(str(filename).encode(), int(device), C.byref(h)),
                  "orbx_vocabulary_load_text")
            self._h = h

    @classmethod
    def from_arrays(cls, k, L, parent, is_leaf, descriptors, weights, weighting=TF_IDF, scoring=L1_NORM, device=0):
        par = np.ascontiguousarray(parent, np.int32)
        leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        w = np.ascontiguousarray(weights, np.float64)
        assert len(par) == len(leaf) == len(desc) == len(w)
        h = C.c_void_p()
        check(lib().orbx_vocabulary_create(int(k), int(L), int(scoring), int(weighting), len(par),
                                           par.ctypes.data_as(_i32p), ptr(leaf), ptr(desc), ptr(w), int(device),
                                           C.byref(h)), "orbx_vocabulary_create")
        return cls(_handle=h)

    def __del__(self):
        try:
            if self._h:
                lib().orbx_vocabulary_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def info(self):
        """(k, L, number of nodes including the root, number of words)."""
        v = [C.c_int32() for _ in range(4)]
        check(lib().orbx_vocabulary_info(self._h, *[C.byref(x) for x in v]), "orbx_vocabulary_info")
        return tuple(x.value for x in v)


def transform_batch(ex, voc, n_frames, levelsup=4):
    """vocabulary.transform(descriptors, mBowVec, mFeatVec, levelsup) for frames [0, n_frames)
    of the last batched extraction on ex.  Asynchronous; fetch with download()."""
    check(lib().orbx_bow_transform_batch_device(ex._h, voc._h, int(n_frames), int(levelsup)),
          "orbx_bow_transform_batch_device")


def download(ex, frame):
    """-> (word_ids uint32[W], word_values float64[W], node_ids uint32[N], node_start int32[N+1],
    features uint32[M]): mBowVec and mFeatVec of one frame as sorted arrays."""
    cap = ex.max_keypoints()
    wi = np.zeros(cap, np.uint32)
    wv = np.zeros(cap, np.float64)
    ni = np.zeros(cap, np.uint32)
    ns = np.zeros(cap + 1, np.int32)
    ft = np.zeros(cap, np.uint32)
    nw, nn = C.c_int32(), C.c_int32()
    check(lib().orbx_bow_download(ex._h, int(frame), ptr(wi), ptr(wv), C.byref(nw), ptr(ni), ptr(ns), ptr(ft),
                                  C.byref(nn), cap), "orbx_bow_download")
    W, N = nw.value, nn.value
    return wi[:W].copy(), wv[:W].copy(), ni[:N].copy(), ns[:N + 1].copy(), ft[:ns[N]].copy()
