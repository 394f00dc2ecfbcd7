// TEST-ONLY stand-in for the handful of OpenCV core types include/orbx_cv.hpp touches, so that
// the adaptor can be compiled and exercised in an image without OpenCV.  It implements just
// enough of cv::Mat (reference-counted 8-bit single-channel matrices, ROI views), cv::KeyPoint,
// cv::DMatch, cv::Rect and the Input/OutputArray proxies for tests/cpp/test_adaptor.cpp.
// Nothing here is used by the library or by a build against the real OpenCV.
#ifndef ORBX_TEST_CVSTUB_CORE_HPP_
#define ORBX_TEST_CVSTUB_CORE_HPP_
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {

struct Point2f { float x = 0, y = 0; };
struct Rect { int x, y, width, height; Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {} };

struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};

struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 0;
  DMatch() {}
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), distance(d) {}
};

class Mat {
 public:
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  size_t step = 0;
  Mat() {}
  Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
  Mat(int r, int c, int /*type*/, void* p, size_t s) : rows(r), cols(c), data((uint8_t*)p), step(s) {}
  void create(int r, int c, int /*type*/) {
    if (r == rows && c == cols && own_) return;
    own_ = std::shared_ptr<std::vector<uint8_t>>(new std::vector<uint8_t>((size_t)r * c));
    rows = r; cols = c; step = (size_t)c; data = own_->data();
  }
  void release() { own_.reset(); rows = cols = 0; data = nullptr; step = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return CV_8UC1; }
  bool isContinuous() const { return step == (size_t)cols; }
  Mat operator()(const Rect& r) const {
    Mat m(r.height, r.width, CV_8U, data + (size_t)r.y * step + r.x, step);
    m.own_ = own_;
    return m;
  }
  template <typename T = uint8_t> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
  template <typename T = uint8_t> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
 private:
  std::shared_ptr<std::vector<uint8_t>> own_;
};

class _InputArray {
 public:
  _InputArray() {}
  _InputArray(const Mat& m) : m_(&m) {}
  bool empty() const { return !m_ || m_->empty(); }
  Mat getMat() const { return m_ ? *m_ : Mat(); }
 private:
  const Mat* m_ = nullptr;
};
class _OutputArray {
 public:
  _OutputArray(Mat& m) : m_(&m) {}
  void create(int r, int c, int t) const { m_->create(r, c, t); }
  void release() const { m_->release(); }
  Mat getMat() const { return *m_; }
 private:
  Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
#endif
