// Exercises include/orbx_cv.hpp (the OpenCV-typed adaptor) against the C ABI it wraps, built
// with the test-only OpenCV stand-in of tests/cpp/cvstub.  Needs a GPU.  Prints "ok" and
// returns 0 when the adaptor's outputs equal those of direct C calls.
#include <orbx_cv.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
  const int W = 752, H = 480;
  cv::Mat left(H, W, CV_8UC1), wide(H, W + 20, CV_8UC1), right;
  orbx_synth_frame(9, W + 20, H, W + 20, wide.data);
  for (int y = 0; y < H; ++y) std::memcpy(left.ptr(y), wide.ptr(y), W);
  right = wide(cv::Rect(20, 0, W, H));       // a strided ROI view: right(x) = left(x + 20)

  orbx_cv::ORBextractor exL(1200, 1.2f, 8, 20, 7), exR(1200, 1.2f, 8, 20, 7);
  CHECK(exL.GetLevels() == 8 && exL.GetScaleFactors().size() == 8);
  CHECK(exL.GetScaleFactor() == 1.2f);
  std::vector<cv::KeyPoint> kL, kR;
  cv::Mat dL, dR;
  std::vector<int> lap = {0, 0};
  int monoL = exL(left, cv::Mat(), kL, dL, lap);
  int monoR = exR(right, cv::Mat(), kR, dR, lap);
  CHECK(monoL == (int)kL.size() && monoR == (int)kR.size());   // lap (0,0): everything is "mono"
  CHECK(dL.rows == (int)kL.size() && dL.cols == 32 && kL.size() > 1000);
  CHECK(exL(cv::Mat(), cv::Mat(), kR, dR, lap) == -1);          // empty image
  exR(right, cv::Mat(), kR, dR, lap);

  // the same through the C ABI
  orbx_params p = {1200, 1.2f, 8, 20, 7, 0, 1};
  orbx_extractor* h = nullptr;
  CHECK(orbx_create(&p, &h) == ORBX_OK);
  const int cap = orbx_max_keypoints(h);
  std::vector<orbx_keypoint> ck(cap);
  std::vector<uint8_t> cd((size_t)cap * 32);
  int n = 0, mono = 0;
  CHECK(orbx_extract(h, left.data, W, H, (int)left.step, 0, 0, ck.data(), cd.data(), cap, &n, &mono) == ORBX_OK);
  CHECK(n == (int)kL.size() && mono == monoL);
  CHECK(std::memcmp(ck.data(), kL.data(), (size_t)n * 28) == 0);
  for (int i = 0; i < n; ++i) CHECK(std::memcmp(cd.data() + (size_t)i * 32, dL.ptr(i), 32) == 0);
  CHECK(orbx_cv::DescriptorDistance(dL(cv::Rect(0, 0, 32, 1)), dL(cv::Rect(0, 0, 32, 1))) == 0);

  // mvImagePyramid is usable right after operator(), with no further call, as unchanged
  // Frame::ComputeStereoMatches needs it (core/src/Frame.cc:837, 921-941): level 0 equals the
  // input, level sizes follow the tables, and every level - border included - equals the padded
  // plane the C ABI dumps for the same image
  CHECK(exL.mvImagePyramid.size() == 8);
  CHECK(exL.mvImagePyramid[0].rows == H && exL.mvImagePyramid[0].cols == W);
  for (int y = 0; y < H; y += 37) CHECK(std::memcmp(exL.mvImagePyramid[0].ptr(y), left.ptr(y), W) == 0);
  CHECK(exL.mvImagePyramid[1].cols == 627 && exL.mvImagePyramid[1].rows == 400);
  for (int l = 0; l < 8; ++l) {
    int lw = 0, lh = 0, ls = 0;
    CHECK(orbx_level_dims(h, l, &lw, &lh, &ls) == ORBX_OK);
    std::vector<uint8_t> plane((size_t)(lh + 38) * ls);
    CHECK(orbx_debug_level_padded(h, 0, l, plane.data()) == ORBX_OK);
    const cv::Mat& m = exL.mvImagePyramid[l];
    CHECK(m.rows == lh && m.cols == lw);
    for (int y = -19; y < lh + 19; ++y)   // the ROI view sits inside its padded plane
      CHECK(std::memcmp(m.ptr(0) + (long)y * (long)m.step - 19, plane.data() + (size_t)(y + 19) * ls, lw + 38) == 0);
  }
  {
    // a device-side stereo consumer opts out of the transfer: empty Mats until asked
    orbx_cv::ORBextractor exD(1200, 1.2f, 8, 20, 7, 0, /*hostPyramid=*/false);
    std::vector<cv::KeyPoint> kD;
    cv::Mat dD;
    CHECK(exD(left, cv::Mat(), kD, dD, lap) == monoL && kD.size() == kL.size());
    CHECK(exD.mvImagePyramid[0].empty());
    exD.SyncImagePyramid();
    CHECK(exD.mvImagePyramid[0].rows == H && std::memcmp(exD.mvImagePyramid[0].ptr(5), left.ptr(5), W) == 0);
  }

  // stereo association and 2-NN
  std::vector<float> uR, depth;
  orbx_cv::ComputeStereoMatches(exL, exR, 435.2f * 0.11f, 0.11f, uR, depth);
  CHECK(uR.size() == kL.size() && depth.size() == kL.size());
  int good = 0;
  for (size_t i = 0; i < uR.size(); ++i)
    if (depth[i] > 0) { ++good; CHECK(std::abs((kL[i].pt.x - uR[i]) - 20.f) < 4.f); }
  CHECK(good > 200);
  std::vector<std::vector<cv::DMatch>> m;
  orbx_cv::KnnMatch2(dL, dR, m);
  CHECK(m.size() == kL.size() && m[0].size() == 2 && m[0][0].distance <= m[0][1].distance);
  orbx_destroy(h);
  std::printf("ok %d keypoints, %d stereo matches\n", n, good);
  return 0;
}
