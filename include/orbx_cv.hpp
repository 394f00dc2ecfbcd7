/*
 * orbx_cv.hpp - OpenCV-typed C++ adaptor over the C ABI of orbx.h.
 *
 * Header-only.  It gives the hot path back the interface the rest of ORB-SLAM3
 * compiles against:
 *   orbx_cv::ORBextractor         same constructor, operator(), getters and public
 *                                 mvImagePyramid as ORB_SLAM3::ORBextractor
 *                                 (core/lib/ORB/extractor.h:45-102)
 *   orbx_cv::DescriptorDistance   ORBmatcher::DescriptorDistance
 *                                 (core/lib/ORB/matcher.h:42, matcher.cc:1961-1975)
 *   orbx_cv::ComputeStereoMatches the body of Frame::ComputeStereoMatches
 *                                 (core/src/Frame.cc:831-996) on two extractors
 *   orbx_cv::KnnMatch2            BFMatcher(NORM_HAMMING).knnMatch(q, t, m, 2)
 *                                 (core/src/Frame.cc:1177)
 *   orbx_cv::ORBVocabulary        the vocabulary in device memory, loaded from the text format
 *                                 of ORBVocabulary::loadFromText (core/lib/ORB/vocabulary.cc:96-160)
 *   orbx_cv::ComputeBoW           the body of Frame::ComputeBoW (core/src/Frame.cc:764-769) on an
 *                                 extractor's last frame, into DBoW2-shaped maps
 * A source tree that wants the drop-in adds
 *   namespace ORB_SLAM3 { using ORBextractor = orbx_cv::ORBextractor; }
 * in place of its extractor.h.  Needs <opencv2/core.hpp>; link with -lorbx_hip.
 */
#ifndef ORBX_CV_HPP_
#define ORBX_CV_HPP_

#if !defined(__has_include) || !__has_include(<opencv2/core.hpp>)
#error "orbx_cv.hpp needs OpenCV's <opencv2/core.hpp> on the include path"
#endif

#include <opencv2/core.hpp>

#include <cassert>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbx.h"

namespace orbx_cv {

static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint),
              "cv::KeyPoint must have the 28-byte layout orbx_keypoint mirrors");

inline void throw_on(int status, const char* what) {
  if (status != ORBX_OK)
    throw std::runtime_error(std::string("orbx: ") + orbx_strerror(status) + " in " + what);
}

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  // hostPyramid: fill mvImagePyramid inside operator() as the reference does (extractor.cc:1030,
  // 1111-1112), so that unchanged callers such as Frame::ComputeStereoMatches (core/src/Frame.cc:
  // 837, 921-941) can read it right after the call.  Callers that keep the stereo association on
  // the device (ComputeStereoMatches below) pass false and save the 1.3 MB transfer per call;
  // they can still fetch the pyramid later with SyncImagePyramid().
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
               int device = 0, bool hostPyramid = true)
      : nfeatures_(nfeatures), scaleFactor_(scaleFactor), nlevels_(nlevels), hostPyramid_(hostPyramid) {
    orbx_params p;
    p.nfeatures = nfeatures;
    p.scale_factor = scaleFactor;
    p.nlevels = nlevels;
    p.ini_th_fast = iniThFAST;
    p.min_th_fast = minThFAST;
    p.device_id = device;
    p.max_batch = 1;
    throw_on(orbx_create(&p, &handle_), "orbx_create");
    mvScaleFactor_.resize(nlevels);
    mvInvScaleFactor_.resize(nlevels);
    mvLevelSigma2_.resize(nlevels);
    mvInvLevelSigma2_.resize(nlevels);
    mnFeaturesPerLevel_.resize(nlevels);
    throw_on(orbx_get_tables(handle_, mvScaleFactor_.data(), mvInvScaleFactor_.data(),
                             mvLevelSigma2_.data(), mvInvLevelSigma2_.data(),
                             mnFeaturesPerLevel_.data()),
             "orbx_get_tables");
    mvImagePyramid.resize(nlevels);
  }
  ~ORBextractor() { orbx_destroy(handle_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Mask is ignored, as in the reference (extractor.h:54-59).  Returns monoIndex
  // (extractor.cc:1100), or -1 for an empty image (extractor.cc:1023-1024).
  int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                 cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    int cap = orbx_max_keypoints(handle_), n = 0, mono = -1;
    _keypoints.resize(cap);
    scratchDesc_.resize((size_t)cap * 32);
    int st = orbx_extract(handle_, image.data, image.cols, image.rows, (int)image.step, vLappingArea[0],
                          vLappingArea[1], reinterpret_cast<orbx_keypoint*>(_keypoints.data()),
                          scratchDesc_.data(), cap, &n, &mono);
    throw_on(st, "orbx_extract");
    _keypoints.resize(n);
    if (n == 0) {
      _descriptors.release();                         // extractor.cc:1041-1042
    } else {
      _descriptors.create(n, 32, CV_8U);              // extractor.cc:1044
      cv::Mat d = _descriptors.getMat();
      for (int i = 0; i < n; ++i)
        std::copy(scratchDesc_.begin() + (size_t)i * 32, scratchDesc_.begin() + (size_t)(i + 1) * 32, d.ptr(i));
    }
    pyramidFresh_ = false;
    if (hostPyramid_) SyncImagePyramid();
    else for (cv::Mat& m : mvImagePyramid) m = cv::Mat();   // no stale views of the previous frame
    return mono;
  }

  int GetLevels() { return nlevels_; }
  float GetScaleFactor() { return (float)scaleFactor_; }
  std::vector<float> GetScaleFactors() { return mvScaleFactor_; }
  std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor_; }
  std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2_; }
  std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2_; }

  // Level images as ROI views (19-pixel border around each, extractor.cc:1103-1127) into a
  // page-locked host mirror that the next operator() overwrites, as the reference overwrites its
  // pyramid.  Filled by operator() unless the extractor was built with hostPyramid = false; then
  // the planes stay on the device until SyncImagePyramid() is called.
  std::vector<cv::Mat> mvImagePyramid;
  void SyncImagePyramid() {
    if (pyramidFresh_) return;
    for (int l = 0; l < nlevels_; ++l) {
      const uint8_t* padded = nullptr;
      int w = 0, h = 0, stride = 0;
      throw_on(orbx_pyramid(handle_, l, &padded, &w, &h, &stride), "orbx_pyramid");
      cv::Mat whole(h + 38, w + 38, CV_8UC1, const_cast<uint8_t*>(padded), (size_t)stride);
      mvImagePyramid[l] = whole(cv::Rect(19, 19, w, h));
    }
    pyramidFresh_ = true;
  }

  orbx_extractor* handle() { return handle_; }

 private:
  orbx_extractor* handle_ = nullptr;
  int nfeatures_;
  double scaleFactor_;   // the reference keeps the float argument in a double (extractor.h:89)
  int nlevels_;
  bool hostPyramid_ = true;
  bool pyramidFresh_ = false;
  std::vector<uint8_t> scratchDesc_;
  std::vector<int> mnFeaturesPerLevel_;
  std::vector<float> mvScaleFactor_, mvInvScaleFactor_, mvLevelSigma2_, mvInvLevelSigma2_;
};

inline int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
  return orbx_hamming(a.ptr<uint8_t>(), b.ptr<uint8_t>());
}

// After left(imLeft, ...) and right(imRight, ...): mvuRight / mvDepth per left keypoint.
inline void ComputeStereoMatches(ORBextractor& left, ORBextractor& right, float mbf, float mb,
                                 std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
  const int cap = orbx_max_keypoints(left.handle());
  int n = 0;
  mvuRight.assign(cap, -1.0f);
  mvDepth.assign(cap, -1.0f);
  throw_on(orbx_stereo_match(left.handle(), right.handle(), mbf, mb, mvuRight.data(), mvDepth.data(), cap, &n),
           "orbx_stereo_match");
  mvuRight.resize(n);
  mvDepth.resize(n);
}

// knnMatch(k = 2): matches[i] holds up to two cv::DMatch(queryIdx, trainIdx, distance).
inline void KnnMatch2(const cv::Mat& query, const cv::Mat& train,
                      std::vector<std::vector<cv::DMatch>>& matches, int device = 0) {
  const int nq = query.rows, nt = train.rows;
  matches.assign(nq, std::vector<cv::DMatch>());
  if (nq == 0 || nt == 0) return;
  assert(query.isContinuous() && train.isContinuous() && query.cols == 32 && train.cols == 32);
  std::vector<uint32_t> idx((size_t)nq * 2);
  std::vector<uint16_t> dist((size_t)nq * 2);
  throw_on(orbx_knn2(device, query.ptr<uint8_t>(), nq, train.ptr<uint8_t>(), nt, idx.data(), dist.data()),
           "orbx_knn2");
  for (int i = 0; i < nq; ++i)
    for (int e = 0; e < 2; ++e)
      if (idx[2 * (size_t)i + e] != 0xFFFFFFFFu)
        matches[i].push_back(cv::DMatch(i, (int)idx[2 * (size_t)i + e], (float)dist[2 * (size_t)i + e]));
}

// The vocabulary tree in device memory.  load() takes the text format ("k L scoring weighting",
// then one node per line: parent, leaf flag, 32 descriptor bytes, weight).
class ORBVocabulary {
 public:
  ORBVocabulary() {}
  explicit ORBVocabulary(const std::string& filename, int device = 0) { load(filename, device); }
  ~ORBVocabulary() { orbx_vocabulary_destroy(voc_); }
  ORBVocabulary(const ORBVocabulary&) = delete;
  ORBVocabulary& operator=(const ORBVocabulary&) = delete;
  bool load(const std::string& filename, int device = 0) {
    orbx_vocabulary_destroy(voc_);
    voc_ = nullptr;
    return orbx_vocabulary_load_text(filename.c_str(), device, &voc_) == ORBX_OK;
  }
  bool empty() const { return voc_ == nullptr; }
  const orbx_vocabulary* handle() const { return voc_; }

 private:
  orbx_vocabulary* voc_ = nullptr;
};

// After extractor(image, ...): mBowVec / mFeatVec of that frame.  BowMap is a
// std::map<unsigned, double> (DBoW2::BowVector derives from it), FeatMap a
// std::map<unsigned, std::vector<unsigned>> (DBoW2::FeatureVector); entries are inserted in key
// order, the order the reference's maps iterate in.
template <class BowMap, class FeatMap>
inline void ComputeBoW(ORBextractor& extractor, const ORBVocabulary& voc, BowMap& mBowVec, FeatMap& mFeatVec,
                       int levelsup = 4) {
  mBowVec.clear();
  mFeatVec.clear();
  if (voc.empty()) return;
  throw_on(orbx_bow_transform_batch_device(extractor.handle(), voc.handle(), 1, levelsup),
           "orbx_bow_transform_batch_device");
  const int cap = orbx_max_keypoints(extractor.handle());
  std::vector<uint32_t> words(cap), nodes(cap), feats(cap);
  std::vector<double> values(cap);
  std::vector<int32_t> start(cap + 1);
  int nw = 0, nn = 0;
  throw_on(orbx_bow_download(extractor.handle(), 0, words.data(), values.data(), &nw, nodes.data(), start.data(),
                             feats.data(), &nn, cap), "orbx_bow_download");
  for (int i = 0; i < nw; ++i) mBowVec.insert(mBowVec.end(), typename BowMap::value_type(words[i], values[i]));
  for (int r = 0; r < nn; ++r) {
    typename FeatMap::iterator it =
        mFeatVec.insert(mFeatVec.end(), typename FeatMap::value_type(nodes[r], typename FeatMap::mapped_type()));
    it->second.assign(feats.begin() + start[r], feats.begin() + start[r + 1]);
  }
}

}  // namespace orbx_cv

#endif  // ORBX_CV_HPP_
