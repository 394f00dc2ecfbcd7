"""orb_slam3_amd - MI355X-native ORB feature front-end (extractor + Hamming matching).

Host-side mirror of the reference's interface for this path:
  ORBextractor              core/lib/ORB/extractor.h:45-102
  descriptor_distance       ORBmatcher::DescriptorDistance, core/lib/ORB/matcher.h:42
  knn_match2 / stereo_band_match   the matching loops of core/src/Frame.cc:1177, 840-908
  compute_stereo_matches    Frame::ComputeStereoMatches, core/src/Frame.cc:831-996
  grid.*                    UndistortKeyPoints / AssignFeaturesToGrid / GetFeaturesInArea, Frame.cc:432-459, 682-829
  bow.*                     ORBVocabulary (text format) / Frame::ComputeBoW, vocabulary.cc:96-160, Frame.cc:764-769
  guided.*                  ORBmatcher::SearchByProjection(Frame, LastFrame) / SearchForInitialization, matcher.cc:1591-1808, 629-733
All compute runs in liborbx_hip.so (hand-written gfx950 kernels behind the C
ABI of include/orbx.h); this package only marshals arguments.
"""
from ._lib import KEYPOINT_DTYPE, OrbxError, lib  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import (descriptor_distance, knn_match2, ratio_test, stereo_band_match,  # noqa: F401
                      knn_match2_sharded, shard_bounds, compute_stereo_matches,
                      compute_stereo_matches_batch, compute_stereo_from_rgbd_batch, lapping_knn_match2,
                      lapping_knn_match2_batch, lapping_knn_download, lapping_ratio_test_batch,
                      lapping_matches_download, ratio_test_device,
                      stereo_download)
from .synth import textured_frame, right_view  # noqa: F401
from . import grid  # noqa: F401
from . import guided  # noqa: F401
from . import bow  # noqa: F401
