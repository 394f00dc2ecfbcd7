// Pins the CPU oracle's restatements of the OpenCV primitives against a real OpenCV (>= 4.4).
//
// The build image of this repository has no OpenCV, so this program cannot be compiled there;
// it is meant for any machine that has it:
//
//   make -C oracle
//   g++ -O2 -std=c++17 tools/ref_orb_opencv.cc -o ref_orb_opencv \
//       $(pkg-config --cflags --libs opencv4) -Loracle -lorb_oracle -Wl,-rpath,$PWD/oracle
//   ./ref_orb_opencv
//
// For every primitive the ORB front-end calls (core/lib/ORB/extractor.cc:789,806,1064,1116-1124,
// :98; core/src/Frame.cc:787,1177) it runs OpenCV and the oracle on the same deterministic
// inputs and reports the number of differing outputs.  "0 differences" on every line means the
// oracle (and therefore the GPU path, which is tested bit-exact against the oracle) reproduces
// this OpenCV build.
#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgproc.hpp>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

extern "C" {
struct orc_extractor;
orc_extractor* orc_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh);
void orc_destroy(orc_extractor* h);
int orc_extract(orc_extractor* h, const uint8_t* img, int W, int H, int stride, int lap0, int lap1,
                void* kps28, uint8_t* desc, int cap, int* n, int* mono);
int orc_level_dims(orc_extractor* h, int level, int* w, int* hh, int* stride);
const uint8_t* orc_level_padded(orc_extractor* h, int level);
const uint8_t* orc_level_blurred(orc_extractor* h, int level);
int orc_cv_round(float v);
float orc_fast_atan2(float y, float x);
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                          int dstride);
void orc_gauss7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
int orc_fast(const uint8_t* img, int cols, int rows, int stride, int threshold, int* xys, int cap);
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int64_t nt, uint32_t* idx, uint16_t* dist);
void orc_undistort_points(const float* cam, const float* xy, int n, float* out);
}

namespace {

// Textured test image: random rectangles + noise, lightly smoothed (any content works; it only
// has to exercise corners, flat areas and the image borders).
cv::Mat textured(int w, int h, uint64_t seed) {
  cv::RNG rng(seed);
  cv::Mat img(h, w, CV_8UC1, cv::Scalar(128));
  for (int i = 0; i < 400; ++i) {
    const int x = rng.uniform(0, w), y = rng.uniform(0, h);
    const int rw = rng.uniform(6, 97), rh = rng.uniform(6, 97);
    cv::rectangle(img, cv::Rect(x, y, rw, rh) & cv::Rect(0, 0, w, h), cv::Scalar(rng.uniform(0, 256)), cv::FILLED);
  }
  cv::Mat noise(h, w, CV_8SC1);
  rng.fill(noise, cv::RNG::UNIFORM, -6, 7);
  cv::Mat tmp;
  img.convertTo(tmp, CV_16S);
  cv::Mat n16;
  noise.convertTo(n16, CV_16S);
  tmp += n16;
  tmp.convertTo(img, CV_8U);
  cv::blur(img, img, cv::Size(3, 3));
  return img;
}

long diffBytes(const cv::Mat& a, const uint8_t* b, int bstride) {
  long d = 0;
  for (int y = 0; y < a.rows; ++y)
    for (int x = 0; x < a.cols; ++x) d += a.at<uint8_t>(y, x) != b[(size_t)y * bstride + x];
  return d;
}

int report(const char* what, long differences, long total) {
  std::printf("%-58s %ld differences of %ld%s\n", what, differences, total, differences ? "   <-- MISMATCH" : "");
  return differences != 0;
}

}  // namespace

int main() {
  int bad = 0;
  const int W = 752, H = 480, nlevels = 8;
  const float scaleFactor = 1.2f;
  cv::Mat image = textured(W, H, 1);

  // ---- cvRound ------------------------------------------------------------------------------
  {
    long d = 0, n = 0;
    for (int i = -4000; i <= 4000; ++i, ++n) d += cvRound(i * 0.25f) != orc_cv_round(i * 0.25f);
    bad += report("cvRound(float), quarters in [-1000, 1000]", d, n);
  }

  // ---- ComputePyramid: resize chain + REFLECT_101 borders (extractor.cc:1103-1127) ------------
  std::vector<cv::Mat> levels(nlevels);
  {
    orc_extractor* ex = orc_create(1000, scaleFactor, nlevels, 20, 7);
    std::vector<uint8_t> kps(28 * 4000), desc(32 * 4000);
    int n = 0, mono = 0;
    orc_extract(ex, image.data, W, H, (int)image.step, 0, 1000, kps.data(), desc.data(), 4000, &n, &mono);
    std::vector<float> scale(nlevels, 1.0f), inv(nlevels, 1.0f);
    for (int l = 1; l < nlevels; ++l) scale[l] = (float)(scale[l - 1] * (double)scaleFactor);
    for (int l = 0; l < nlevels; ++l) inv[l] = 1.0f / scale[l];
    const int E = 19;
    std::vector<cv::Mat> pyr(nlevels);
    long dPyr = 0, nPyr = 0, dBlur = 0, nBlur = 0;
    for (int l = 0; l < nlevels; ++l) {
      const cv::Size sz(cvRound((float)W * inv[l]), cvRound((float)H * inv[l]));
      cv::Mat temp(sz.height + 2 * E, sz.width + 2 * E, CV_8UC1);
      pyr[l] = temp(cv::Rect(E, E, sz.width, sz.height));
      if (l != 0) {
        cv::resize(pyr[l - 1], pyr[l], sz, 0, 0, cv::INTER_LINEAR);
        cv::copyMakeBorder(pyr[l], temp, E, E, E, E, cv::BORDER_REFLECT_101 + cv::BORDER_ISOLATED);
      } else {
        cv::copyMakeBorder(image, temp, E, E, E, E, cv::BORDER_REFLECT_101);
      }
      int w = 0, h = 0, stride = 0;
      orc_level_dims(ex, l, &w, &h, &stride);
      if (w != sz.width || h != sz.height) { std::printf("level %d size differs\n", l); ++bad; continue; }
      dPyr += diffBytes(temp, orc_level_padded(ex, l), stride);
      nPyr += (long)temp.rows * temp.cols;
      // per-level blur exactly as operator() does it (extractor.cc:1063-1064)
      cv::Mat working = pyr[l].clone();
      cv::GaussianBlur(working, working, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
      if (const uint8_t* b = orc_level_blurred(ex, l)) {
        dBlur += diffBytes(working, b, w);
        nBlur += (long)w * h;
      }
      levels[l] = pyr[l].clone();
    }
    bad += report("pyramid: resize(INTER_LINEAR) chain + REFLECT_101 border", dPyr, nPyr);
    bad += report("GaussianBlur(7x7, sigma 2, REFLECT_101) per level", dBlur, nBlur);
    orc_destroy(ex);
  }

  // ---- stand-alone resize and blur on odd sizes -------------------------------------------------
  {
    long d = 0, n = 0, db = 0;
    const int sizes[][2] = {{67, 67}, {101, 77}, {320, 240}, {641, 479}};
    for (const auto& s : sizes) {
      cv::Mat src = textured(s[0], s[1], 7 + s[0]);
      const cv::Size dsz(cvRound(s[0] / 1.2f), cvRound(s[1] / 1.2f));
      cv::Mat dst, mine(dsz, CV_8UC1);
      cv::resize(src, dst, dsz, 0, 0, cv::INTER_LINEAR);
      orc_resize_linear_u8(src.data, s[0], s[1], (int)src.step, mine.data, dsz.width, dsz.height, (int)mine.step);
      d += diffBytes(dst, mine.data, (int)mine.step);
      n += (long)dsz.area();
      cv::Mat blurred, mineB(src.size(), CV_8UC1);
      cv::GaussianBlur(src, blurred, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
      orc_gauss7(src.data, s[0], s[1], (int)src.step, mineB.data, (int)mineB.step);
      db += diffBytes(blurred, mineB.data, (int)mineB.step);
    }
    bad += report("resize on odd sizes", d, n);
    bad += report("GaussianBlur on odd sizes", db, n);
  }

  // ---- FAST(roi, keypoints, threshold, true) on cell-sized regions (extractor.cc:789,806) -------
  {
    long d = 0, n = 0;
    cv::RNG rng(5);
    for (int it = 0; it < 400; ++it) {
      const cv::Mat& lv = levels[it % nlevels];
      const int cw = rng.uniform(41, 58), ch = rng.uniform(41, 58);
      if (lv.cols <= cw || lv.rows <= ch) continue;
      const int x0 = rng.uniform(0, lv.cols - cw), y0 = rng.uniform(0, lv.rows - ch);
      const int th = (it & 1) ? 20 : 7;
      cv::Mat roi = lv(cv::Rect(x0, y0, cw, ch));
      std::vector<cv::KeyPoint> k;
      cv::FAST(roi, k, th, true);
      std::vector<int> xys(3 * 4096);
      const int m = orc_fast(roi.data, cw, ch, (int)roi.step, th, xys.data(), 4096);
      ++n;
      bool same = m == (int)k.size();
      for (int i = 0; same && i < m; ++i)
        same = xys[3 * i] == (int)k[i].pt.x && xys[3 * i + 1] == (int)k[i].pt.y && xys[3 * i + 2] == (int)k[i].response;
      d += !same;
    }
    bad += report("FAST-9/16 with non-max suppression: cells (positions, order, response)", d, n);
  }

  // ---- fastAtan2 (extractor.cc:98) ---------------------------------------------------------------
  {
    long d = 0, n = 0;
    double worst = 0;
    for (int y = -300; y <= 300; y += 7)
      for (int x = -300; x <= 300; x += 5, ++n) {
        const float a = cv::fastAtan2((float)y * 13.f, (float)x * 11.f), b = orc_fast_atan2((float)y * 13.f, (float)x * 11.f);
        d += a != b;
        worst = std::max(worst, (double)std::fabs(a - b));
      }
    bad += report("fastAtan2 (bit-identical floats)", d, n);
    std::printf("    largest |difference| %.3g degrees (1e-4 rad = 5.7e-3 degrees is the tolerance of the path)\n", worst);
  }

  // ---- undistortPoints (Frame.cc:787, 814) -------------------------------------------------------
  {
    const float cam[9] = {458.654f, 457.296f, 367.215f, 248.375f, -0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f, 0.f};
    cv::Mat K = (cv::Mat_<float>(3, 3) << cam[0], 0, cam[2], 0, cam[1], cam[3], 0, 0, 1);
    cv::Mat D = (cv::Mat_<float>(4, 1) << cam[4], cam[5], cam[6], cam[7]);
    const int n = 2000;
    cv::Mat pts(n, 2, CV_32F);
    cv::RNG rng(9);
    for (int i = 0; i < n; ++i) { pts.at<float>(i, 0) = rng.uniform(0.f, 752.f); pts.at<float>(i, 1) = rng.uniform(0.f, 480.f); }
    std::vector<float> mine(2 * n);
    orc_undistort_points(cam, pts.ptr<float>(), n, mine.data());
    cv::Mat m2 = pts.clone().reshape(2);
    cv::undistortPoints(m2, m2, K, D, cv::Mat(), K);
    m2 = m2.reshape(1);
    long d = 0;
    double worst = 0;
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < 2; ++c) {
        d += m2.at<float>(i, c) != mine[2 * i + c];
        worst = std::max(worst, (double)std::fabs(m2.at<float>(i, c) - mine[2 * i + c]));
      }
    bad += report("undistortPoints(K, D, noArray(), K) (bit-identical floats)", d, 2L * n);
    std::printf("    largest |difference| %.3g pixels\n", worst);
  }

  // ---- BFMatcher(NORM_HAMMING).knnMatch(k = 2) (Frame.cc:1177) ------------------------------------
  {
    cv::RNG rng(11);
    cv::Mat t(3000, 32, CV_8U), q(200, 32, CV_8U);
    rng.fill(t, cv::RNG::UNIFORM, 0, 256);
    for (int i = 0; i < 100; ++i) t.row(i).copyTo(t.row(1500 + i));          // exact distance ties
    for (int i = 0; i < 200; ++i) t.row(rng.uniform(0, 3000)).copyTo(q.row(i));
    for (int i = 0; i < 200; i += 2) q.at<uint8_t>(i, rng.uniform(0, 32)) ^= 0x21;
    std::vector<std::vector<cv::DMatch>> m;
    cv::BFMatcher(cv::NORM_HAMMING).knnMatch(q, t, m, 2);
    std::vector<uint32_t> idx(400);
    std::vector<uint16_t> dist(400);
    orc_knn2(q.data, 200, t.data, 3000, idx.data(), dist.data());
    long d = 0;
    for (int i = 0; i < 200; ++i)
      for (int e = 0; e < 2; ++e)
        d += m[i].size() != 2 || (uint32_t)m[i][e].trainIdx != idx[2 * i + e] || (int)m[i][e].distance != dist[2 * i + e];
    bad += report("BFMatcher(NORM_HAMMING).knnMatch(k=2): indices and distances", d, 400);
  }

  std::printf(bad ? "\nSOME PRIMITIVES DIFFER from this OpenCV build (%s)\n" : "\nall primitives reproduce this OpenCV build (%s)\n", CV_VERSION);
  return bad ? 1 : 0;
}
