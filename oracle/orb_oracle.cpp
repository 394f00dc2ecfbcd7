// CPU ORACLE for the ORB front-end.  TEST INFRASTRUCTURE ONLY.
//
// This file restates, on the CPU and with no dependencies, the algorithm of
// the reference's ORB extractor and Hamming matchers.  It exists to check the
// HIP implementation and to serve as the timed CPU baseline; the product
// (liborbx_hip.so and the orb_slam3_amd package) never calls it.
//
// PARITY STATUS: "parity unpinned".  The reference ships no tests, golden
// vectors or fixtures for this path, and its arithmetic lives in OpenCV
// (>= 4.4, unpinned; core/CMakeLists.txt:14), which is absent here, so the
// reference itself cannot be built.  The OpenCV primitives it calls
// (resize INTER_LINEAR 8U, copyMakeBorder REFLECT_101, FAST 9/16 with NMS,
// GaussianBlur 7x7 sigma 2 fixed-point, fastAtan2, cvRound) are restated from
// OpenCV 4.x's published algorithms.  What pins this file: the in-tree data
// (BRIEF pattern, constants), the derived known answers (per-level quotas,
// umax, level sizes, blur kernel), self-consistency tests under tests/, and a
// second restatement written separately in numpy (tests/numpy_spec.py) that
// must agree with this one bit for bit.  tools/ref_orb_opencv.cc compares the
// exported primitives with a real OpenCV wherever one is installed.
//
// Floating-point canon (build with -ffp-contract=off):
//   * no fused multiply-add anywhere;
//   * cos/sin of the keypoint angle = float(cos(double(angle_rad))).
//
// Each function cites the reference lines (relative to /root/reference) it
// follows.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <list>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace {

const int kPatch = 31;       // core/lib/ORB/extractor.cc:71
const int kHalfPatch = 15;   // :72
const int kEdge = 19;        // :73
const int kCellW = 35;       // :751

const int kPattern[256 * 4] = {
#include "orb_pattern_31.inc"
};

struct KeyPoint {  // cv::KeyPoint layout, 28 bytes
  float x, y, size, angle, response;
  int octave, class_id;
};

// ---- OpenCV scalar helpers -------------------------------------------------
// cvRound: round half to even (SSE cvtss2si / lrint under the default mode).
inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvRoundD(double v) { return (int)lrint(v); }
inline int cvFloorF(float v) {
  int i = (int)v;
  return i - (i > v);
}
inline int cvCeilF(float v) {
  int i = (int)v;
  return i + (i < v);
}
inline short satShort(float v) {
  int r = cvRoundF(v);
  return (short)std::min(std::max(r, -32768), 32767);
}
// BORDER_REFLECT_101 index map: gfedcb|abcdefgh|gfedcba
inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    else i = 2 * n - 2 - i;
  }
  return i;
}

// cv::fastAtan2 (degrees, polynomial approximation).
float fastAtan2(float y, float x) {
  const float k = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * k;
  const float p3 = -0.3258083974640975f * k;
  const float p5 = 0.1555786518463281f * k;
  const float p7 = -0.04432655554792128f * k;
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---- image container -------------------------------------------------------
struct Plane {  // padded level: data holds (h+2*kEdge) rows of stride bytes
  int w = 0, h = 0, stride = 0;
  std::vector<uint8_t> data;
  uint8_t* roi() { return data.data() + kEdge * stride + kEdge; }
  const uint8_t* roi() const { return data.data() + kEdge * stride + kEdge; }
};

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1: fixed-point
// bilinear with 11-bit coefficients, horizontal pass to int32, vertical pass
// with the 8-bit specialisation's double shift.
void resizeLinearU8(const uint8_t* src, int sw, int sh, int sstride,
                    uint8_t* dst, int dw, int dh, int dstride) {
  const double scale_x = 1.0 / ((double)dw / sw);
  const double scale_y = 1.0 / ((double)dh / sh);
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> alpha(2 * dw), beta(2 * dh);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloorF(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    alpha[2 * dx] = satShort((1.f - fx) * 2048);
    alpha[2 * dx + 1] = satShort(fx * 2048);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloorF(fy);
    fy -= sy;
    yofs[dy] = sy;
    beta[2 * dy] = satShort((1.f - fy) * 2048);
    beta[2 * dy + 1] = satShort(fy * 2048);
  }
  std::vector<int> row0(dw), row1(dw);
  auto hpass = [&](int sy, std::vector<int>& out) {
    sy = std::min(std::max(sy, 0), sh - 1);
    const uint8_t* S = src + (size_t)sy * sstride;
    for (int dx = 0; dx < dw; ++dx) {
      int sx = xofs[dx];
      if (sx + 1 >= sw) out[dx] = S[sx] * 2048;  // single tap past xmax
      else out[dx] = S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
    }
  };
  for (int dy = 0; dy < dh; ++dy) {
    hpass(yofs[dy], row0);
    hpass(yofs[dy] + 1, row1);
    int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int dx = 0; dx < dw; ++dx)
      D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) +
                         ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
  }
}

// In-place border fill of a padded plane from its own interior
// (copyMakeBorder with BORDER_REFLECT_101 [+ BORDER_ISOLATED]).
void fillBorder101(Plane& p) {
  uint8_t* base = p.data.data();
  for (int py = 0; py < p.h + 2 * kEdge; ++py) {
    int sy = reflect101(py - kEdge, p.h);
    uint8_t* drow = base + (size_t)py * p.stride;
    const uint8_t* srow = base + (size_t)(sy + kEdge) * p.stride + kEdge;
    if (py < kEdge || py >= p.h + kEdge)
      for (int x = 0; x < p.w; ++x) drow[kEdge + x] = srow[x];
    for (int px = 0; px < kEdge; ++px) {
      drow[px] = srow[reflect101(px - kEdge, p.w)];
      drow[kEdge + p.w + px] = srow[reflect101(p.w + px, p.w)];
    }
  }
}

// cv::GaussianBlur(Size(7,7), 2, 2, BORDER_REFLECT_101) on a contiguous 8-bit
// image: 8.8 fixed-point separable kernel {18,34,48,56,48,34,18}, 16-bit row
// sums, (sum + 2^15) >> 16 after the column pass.
const int kGauss[7] = {18, 34, 48, 56, 48, 34, 18};
void gaussianBlur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst,
                   int dstride) {
  std::vector<uint16_t> tmp((size_t)w * h);
  for (int y = 0; y < h; ++y) {
    const uint8_t* S = src + (size_t)y * sstride;
    for (int x = 0; x < w; ++x) {
      unsigned acc = 0;
      for (int k = 0; k < 7; ++k) acc += kGauss[k] * S[reflect101(x + k - 3, w)];
      tmp[(size_t)y * w + x] = (uint16_t)acc;
    }
  }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      uint32_t acc = 0;
      for (int k = 0; k < 7; ++k)
        acc += (uint32_t)kGauss[k] * tmp[(size_t)reflect101(y + k - 3, h) * w + x];
      dst[(size_t)y * dstride + x] = (uint8_t)((acc + 32768u) >> 16);
    }
}

// ---- FAST 9/16 with non-max suppression (cv::FAST(roi, kps, thr, true)) ----
const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1},
                            {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                            {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};

struct Cand { int x, y, score; };

// OpenCV cornerScore<16>: largest threshold for which the pixel is still a
// corner, minus one (but never below threshold-1).
int cornerScore16(const uint8_t* p, const int* off, int threshold) {
  int d[25];
  int v = p[0];
  for (int k = 0; k < 25; ++k) d[k] = v - p[off[k & 15]];
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min(d[k + 1], d[k + 2]);
    a = std::min(a, d[k + 3]);
    if (a <= a0) continue;
    for (int j = 4; j <= 8; ++j) a = std::min(a, d[k + j]);
    a0 = std::max(a0, std::min(a, d[k]));
    a0 = std::max(a0, std::min(a, d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max(d[k + 1], d[k + 2]);
    for (int j = 3; j <= 5; ++j) b = std::max(b, d[k + j]);
    if (b >= b0) continue;
    for (int j = 6; j <= 8; ++j) b = std::max(b, d[k + j]);
    b0 = std::min(b0, std::max(b, d[k]));
    b0 = std::min(b0, std::max(b, d[k + 9]));
  }
  return -b0 - 1;
}

// Segment test: 9 contiguous ring pixels all darker than v - t or all
// brighter than v + t.  tab classifies a difference (index value - v + 255)
// as 1 = darker, 2 = brighter, 0 = similar; the opposite-pair tests are the
// usual quick rejection (any 9-arc holds one pixel of every opposite pair).
bool isCorner16(const uint8_t* p, const int* off, int threshold, const uint8_t* tab0) {
  const int v = p[0];
  const uint8_t* tab = tab0 - v + 255;
  int d = tab[p[off[0]]] | tab[p[off[8]]];
  if (d == 0) return false;
  d &= tab[p[off[2]]] | tab[p[off[10]]];
  d &= tab[p[off[4]]] | tab[p[off[12]]];
  d &= tab[p[off[6]]] | tab[p[off[14]]];
  if (d == 0) return false;
  d &= tab[p[off[1]]] | tab[p[off[9]]];
  d &= tab[p[off[3]]] | tab[p[off[11]]];
  d &= tab[p[off[5]]] | tab[p[off[13]]];
  d &= tab[p[off[7]]] | tab[p[off[15]]];
  if (d & 1) {
    const int lo = v - threshold;
    int run = 0;
    for (int k = 0; k < 25; ++k) {
      if (p[off[k & 15]] < lo) { if (++run > 8) return true; } else run = 0;
    }
  }
  if (d & 2) {
    const int hi = v + threshold;
    int run = 0;
    for (int k = 0; k < 25; ++k) {
      if (p[off[k & 15]] > hi) { if (++run > 8) return true; } else run = 0;
    }
  }
  return false;
}

void fast9_16(const uint8_t* img, int cols, int rows, int stride, int threshold,
              std::vector<Cand>& out) {
  out.clear();
  threshold = std::min(std::max(threshold, 0), 255);
  if (cols < 7 || rows < 7) return;
  int off[16];
  for (int k = 0; k < 16; ++k) off[k] = kCircle[k][0] + kCircle[k][1] * stride;
  uint8_t tab[512];
  for (int i = -255; i <= 255; ++i) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  std::vector<uint8_t> score((size_t)cols * rows, 0);
  for (int y = 3; y < rows - 3; ++y)
    for (int x = 3; x < cols - 3; ++x) {
      const uint8_t* p = img + (size_t)y * stride + x;
      if (isCorner16(p, off, threshold, tab))
        score[(size_t)y * cols + x] = (uint8_t)cornerScore16(p, off, threshold);
    }
  for (int y = 3; y < rows - 3; ++y)
    for (int x = 3; x < cols - 3; ++x) {
      const uint8_t* s = &score[(size_t)y * cols + x];
      // Only detected corners are visited; a corner's score is >= threshold,
      // and a score of 0 can never beat its (non-negative) neighbours.
      int sc = *s;
      if (sc == 0) continue;
      if (sc > s[-1] && sc > s[1] && sc > s[-cols - 1] && sc > s[-cols] &&
          sc > s[-cols + 1] && sc > s[cols - 1] && sc > s[cols] && sc > s[cols + 1])
        out.push_back({x, y, sc});
    }
}

// ---- quadtree distribution -------------------------------------------------
struct QNode {
  int ulx, uly, urx, ury, blx, bly, brx, bry;
  std::vector<int> keys;  // indices into the candidate array, parent order kept
  std::list<QNode>::iterator self;
  bool noMore = false;
};

// core/lib/ORB/extractor.cc:474-526
void splitNode(const QNode& p, const std::vector<Cand>& c, QNode ch[4]) {
  const int halfX = (int)std::ceil(static_cast<float>(p.urx - p.ulx) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(p.bry - p.uly) / 2);
  const int mx = p.ulx + halfX, my = p.uly + halfY;
  ch[0].ulx = p.ulx; ch[0].uly = p.uly; ch[0].urx = mx; ch[0].ury = p.uly;
  ch[0].blx = p.ulx; ch[0].bly = my; ch[0].brx = mx; ch[0].bry = my;
  ch[1].ulx = mx; ch[1].uly = p.uly; ch[1].urx = p.urx; ch[1].ury = p.ury;
  ch[1].blx = mx; ch[1].bly = my; ch[1].brx = p.urx; ch[1].bry = my;
  ch[2].ulx = p.ulx; ch[2].uly = my; ch[2].urx = mx; ch[2].ury = my;
  ch[2].blx = p.blx; ch[2].bly = p.bly; ch[2].brx = mx; ch[2].bry = p.bly;
  ch[3].ulx = mx; ch[3].uly = my; ch[3].urx = p.urx; ch[3].ury = my;
  ch[3].blx = mx; ch[3].bly = p.bly; ch[3].brx = p.brx; ch[3].bry = p.bry;
  for (int k : p.keys) {
    const float kx = (float)c[k].x, ky = (float)c[k].y;
    int q = (kx < (float)mx) ? ((ky < (float)my) ? 0 : 2) : ((ky < (float)my) ? 1 : 3);
    ch[q].keys.push_back(k);
  }
  for (int q = 0; q < 4; ++q) ch[q].noMore = ch[q].keys.size() == 1;
}

typedef std::pair<int, QNode*> SizedNode;
bool sizedNodeLess(const SizedNode& a, const SizedNode& b) {  // :528-541
  if (a.first < b.first) return true;
  if (a.first > b.first) return false;
  return a.second->ulx < b.second->ulx;
}

// core/lib/ORB/extractor.cc:543-745.  cand coordinates are relative to
// (minX, minY).  Returns indices into cand, in final list order.
std::vector<int> distributeQuadTree(const std::vector<Cand>& cand, int minX,
                                    int maxX, int minY, int maxY, int N) {
  std::vector<int> result;
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<QNode> nodes;
  std::vector<QNode*> roots(nIni);
  for (int i = 0; i < nIni; ++i) {
    QNode n;
    n.ulx = (int)(hX * static_cast<float>(i)); n.uly = 0;
    n.urx = (int)(hX * static_cast<float>(i + 1)); n.ury = 0;
    n.blx = n.ulx; n.bly = maxY - minY;
    n.brx = n.urx; n.bry = maxY - minY;
    nodes.push_back(n);
    roots[i] = &nodes.back();
  }
  for (size_t i = 0; i < cand.size(); ++i)
    roots[(int)((float)cand[i].x / hX)]->keys.push_back((int)i);
  for (auto it = nodes.begin(); it != nodes.end();) {
    if (it->keys.size() == 1) { it->noMore = true; ++it; }
    else if (it->keys.empty()) it = nodes.erase(it);
    else ++it;
  }
  bool finish = false;
  std::vector<SizedNode> rec;
  auto pushChildren = [&](QNode ch[4], int* nToExpand) {
    for (int q = 0; q < 4; ++q) {
      if (ch[q].keys.empty()) continue;
      nodes.push_front(ch[q]);
      if (ch[q].keys.size() > 1) {
        if (nToExpand) ++*nToExpand;
        rec.push_back(std::make_pair((int)ch[q].keys.size(), &nodes.front()));
        nodes.front().self = nodes.begin();
      }
    }
  };
  while (!finish) {
    int prevSize = (int)nodes.size();
    int nToExpand = 0;
    rec.clear();
    for (auto it = nodes.begin(); it != nodes.end();) {
      if (it->noMore) { ++it; continue; }
      QNode ch[4];
      splitNode(*it, cand, ch);
      pushChildren(ch, &nToExpand);
      it = nodes.erase(it);
    }
    if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
      finish = true;
    } else if ((int)nodes.size() + nToExpand * 3 > N) {
      while (!finish) {
        prevSize = (int)nodes.size();
        std::vector<SizedNode> prev = rec;
        rec.clear();
        std::sort(prev.begin(), prev.end(), sizedNodeLess);
        for (int j = (int)prev.size() - 1; j >= 0; --j) {
          QNode ch[4];
          splitNode(*prev[j].second, cand, ch);
          pushChildren(ch, nullptr);
          nodes.erase(prev[j].second->self);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
      }
    }
  }
  for (auto& n : nodes) {
    int best = n.keys[0];
    float bestResp = (float)cand[best].score;
    for (size_t k = 1; k < n.keys.size(); ++k)
      if ((float)cand[n.keys[k]].score > bestResp) {
        best = n.keys[k];
        bestResp = (float)cand[best].score;
      }
    result.push_back(best);
  }
  return result;
}

// ---- the extractor ---------------------------------------------------------
struct Extractor {
  int nfeatures, nlevels, iniTh, minTh;
  double scaleFactor;  // double member holding the float argument (extractor.h:89)
  std::vector<float> scale, invScale, sigma2, invSigma2;
  std::vector<int> quota;
  int umax[kHalfPatch + 1];
  // per-call state kept for stage-by-stage comparison
  std::vector<Plane> pyr;
  std::vector<std::vector<Cand>> cands;      // coordinates relative to (16,16)
  std::vector<std::vector<KeyPoint>> keys;   // level coords, angle filled
  std::vector<std::vector<uint8_t>> blurred; // w*h contiguous
  std::vector<std::vector<int>> moments;     // (m10, m01) per level key

  Extractor(int nf, float sf, int nl, int ini, int mn)
      : nfeatures(nf), nlevels(nl), iniTh(ini), minTh(mn), scaleFactor(sf) {
    // core/lib/ORB/extractor.cc:406-463
    scale.resize(nl); invScale.resize(nl); sigma2.resize(nl); invSigma2.resize(nl);
    scale[0] = 1.0f; sigma2[0] = 1.0f;
    for (int i = 1; i < nl; ++i) {
      scale[i] = (float)(scale[i - 1] * scaleFactor);
      sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < nl; ++i) {
      invScale[i] = 1.0f / scale[i];
      invSigma2[i] = 1.0f / sigma2[i];
    }
    quota.resize(nl);
    float factor = (float)(1.0f / scaleFactor);
    float desired = nfeatures * (1 - factor) /
                    (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; ++l) {
      quota[l] = cvRoundF(desired);
      sum += quota[l];
      desired *= factor;
    }
    quota[nl - 1] = std::max(nfeatures - sum, 0);
    int vmax = cvFloorF(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = cvCeilF(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= kHalfPatch; ++v) umax[v] = 0;
    for (int v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  void levelSize(int W, int H, int l, int* w, int* h) const {  // :1105-1107
    *w = cvRoundF((float)W * invScale[l]);
    *h = cvRoundF((float)H * invScale[l]);
  }

  // :1103-1127
  void computePyramid(const uint8_t* img, int W, int H, int stride) {
    pyr.assign(nlevels, Plane());
    for (int l = 0; l < nlevels; ++l) {
      Plane& p = pyr[l];
      levelSize(W, H, l, &p.w, &p.h);
      p.stride = p.w + 2 * kEdge;
      p.data.assign((size_t)p.stride * (p.h + 2 * kEdge), 0);
      if (l == 0) {
        for (int y = 0; y < H; ++y)
          std::memcpy(p.roi() + (size_t)y * p.stride, img + (size_t)y * stride, W);
      } else {
        const Plane& s = pyr[l - 1];
        resizeLinearU8(s.roi(), s.w, s.h, s.stride, p.roi(), p.w, p.h, p.stride);
      }
      fillBorder101(p);
    }
  }

  // :747-831 (grid of 35-px cells, FAST with retry at the lower threshold)
  void detectLevel(int l) {
    const Plane& p = pyr[l];
    std::vector<Cand>& out = cands[l];
    out.clear();
    const int minBX = kEdge - 3, minBY = minBX;
    const int maxBX = p.w - kEdge + 3, maxBY = p.h - kEdge + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / kCellW), nRows = (int)(height / kCellW);
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    std::vector<Cand> cell;
    for (int i = 0; i < nRows; ++i) {
      const float iniY = (float)(minBY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBY - 3) continue;
      if (maxY > maxBY) maxY = (float)maxBY;
      for (int j = 0; j < nCols; ++j) {
        const float iniX = (float)(minBX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBX - 6) continue;
        if (maxX > maxBX) maxX = (float)maxBX;
        const int x0 = (int)iniX, y0 = (int)iniY, x1 = (int)maxX, y1 = (int)maxY;
        const uint8_t* roi = p.roi() + (size_t)y0 * p.stride + x0;
        fast9_16(roi, x1 - x0, y1 - y0, p.stride, iniTh, cell);
        if (cell.empty()) fast9_16(roi, x1 - x0, y1 - y0, p.stride, minTh, cell);
        for (const Cand& c : cell)
          out.push_back({c.x + j * wCell, c.y + i * hCell, c.score});
      }
    }
  }

  // :75-99
  float icAngle(const Plane& p, int cx, int cy, int* m10o, int* m01o) const {
    const uint8_t* c = p.roi() + (size_t)cy * p.stride + cx;
    int m01 = 0, m10 = 0;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * c[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
      int vsum = 0, d = umax[v];
      for (int u = -d; u <= d; ++u) {
        int a = c[u + v * p.stride], b = c[u - v * p.stride];
        vsum += a - b;
        m10 += u * (a + b);
      }
      m01 += v * vsum;
    }
    *m10o = m10; *m01o = m01;
    return fastAtan2((float)m01, (float)m10);
  }

  // :102-145
  void describe(const KeyPoint& kp, const uint8_t* img, int step, uint8_t* desc) const {
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = kp.angle * factorPI;
    float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
    const uint8_t* c = img + (size_t)cvRoundF(kp.y) * step + cvRoundF(kp.x);
    const int* pat = kPattern;
    for (int i = 0; i < 32; ++i) {
      int val = 0;
      for (int j = 0; j < 8; ++j, pat += 4) {
        float x0 = (float)pat[0], y0 = (float)pat[1], x1 = (float)pat[2], y1 = (float)pat[3];
        int t0 = c[cvRoundF(x0 * b + y0 * a) * step + cvRoundF(x0 * a - y0 * b)];
        int t1 = c[cvRoundF(x1 * b + y1 * a) * step + cvRoundF(x1 * a - y1 * b)];
        val |= (t0 < t1) << j;
      }
      desc[i] = (uint8_t)val;
    }
  }

  // :1018-1101.  Returns status (0 ok, -1 empty image, -2 too small, -3 capacity).
  int extract(const uint8_t* img, int W, int H, int stride, int lap0, int lap1,
              KeyPoint* kps, uint8_t* desc, int cap, int* nOut, int* monoOut) {
    *nOut = 0; *monoOut = -1;
    if (!img || W <= 0 || H <= 0) return -1;
    {
      int w, h;
      levelSize(W, H, nlevels - 1, &w, &h);
      if (w - 2 * (kEdge - 3) < kCellW || h - 2 * (kEdge - 3) < kCellW) return -2;
      // nIni = round(width/height) must be >= 1 (extractor.cc:549 divides by it)
      for (int l = 0; l < nlevels; ++l) {
        levelSize(W, H, l, &w, &h);
        if ((int)std::round(static_cast<float>(w - 32) / (h - 32)) < 1) return -2;
      }
    }
    computePyramid(img, W, H, stride);
    cands.assign(nlevels, {}); keys.assign(nlevels, {});
    blurred.assign(nlevels, {}); moments.assign(nlevels, {});
    for (int l = 0; l < nlevels; ++l) {
      detectLevel(l);
      const Plane& p = pyr[l];
      const int minB = kEdge - 3;
      std::vector<int> sel = distributeQuadTree(cands[l], minB, p.w - kEdge + 3, minB,
                                                p.h - kEdge + 3, quota[l]);
      const int patchSize = (int)(kPatch * scale[l]);  // :840
      for (int idx : sel) {
        KeyPoint k;
        k.x = (float)cands[l][idx].x + minB;
        k.y = (float)cands[l][idx].y + minB;
        k.size = (float)patchSize;
        k.angle = -1.f;
        k.response = (float)cands[l][idx].score;
        k.octave = l;
        k.class_id = -1;
        keys[l].push_back(k);
      }
    }
    for (int l = 0; l < nlevels; ++l) {
      moments[l].resize(2 * keys[l].size());
      for (size_t i = 0; i < keys[l].size(); ++i)
        keys[l][i].angle = icAngle(pyr[l], cvRoundF(keys[l][i].x), cvRoundF(keys[l][i].y),
                                   &moments[l][2 * i], &moments[l][2 * i + 1]);
    }
    int n = 0;
    for (int l = 0; l < nlevels; ++l) n += (int)keys[l].size();
    *nOut = n;
    if (n > cap) return -3;
    int mono = 0, stereo = n - 1;
    for (int l = 0; l < nlevels; ++l) {
      if (keys[l].empty()) continue;
      const Plane& p = pyr[l];
      blurred[l].resize((size_t)p.w * p.h);
      gaussianBlur7(p.roi(), p.w, p.h, p.stride, blurred[l].data(), p.w);
      const float s = scale[l];
      for (size_t i = 0; i < keys[l].size(); ++i) {
        uint8_t d[32];
        describe(keys[l][i], blurred[l].data(), p.w, d);
        KeyPoint k = keys[l][i];
        if (l != 0) { k.x *= s; k.y *= s; }
        int dstIdx;
        if (k.x >= (float)lap0 && k.x <= (float)lap1) dstIdx = stereo--;
        else dstIdx = mono++;
        kps[dstIdx] = k;
        std::memcpy(desc + (size_t)dstIdx * 32, d, 32);
      }
    }
    *monoOut = mono;
    return 0;
  }
};

// core/lib/ORB/matcher.cc:1961-1975
inline int hamming256(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; ++i) {
    uint32_t x, y;
    std::memcpy(&x, a + 4 * i, 4);
    std::memcpy(&y, b + 4 * i, 4);
    uint32_t v = x ^ y;
    v = v - ((v >> 1) & 0x55555555u);
    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
    dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
  }
  return dist;
}

}  // namespace

extern "C" {

struct orc_extractor;  // opaque

orc_extractor* orc_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
  if (nlevels < 1 || nfeatures < 1 || !(scaleFactor > 1.0f)) return nullptr;
  return reinterpret_cast<orc_extractor*>(new Extractor(nfeatures, scaleFactor, nlevels, iniTh, minTh));
}
void orc_destroy(orc_extractor* h) { delete reinterpret_cast<Extractor*>(h); }

void orc_tables(orc_extractor* h, float* scale, float* inv, float* sigma2, float* invSigma2,
                int* quotas, int* umax16) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  for (int i = 0; i < e->nlevels; ++i) {
    if (scale) scale[i] = e->scale[i];
    if (inv) inv[i] = e->invScale[i];
    if (sigma2) sigma2[i] = e->sigma2[i];
    if (invSigma2) invSigma2[i] = e->invSigma2[i];
    if (quotas) quotas[i] = e->quota[i];
  }
  if (umax16) for (int i = 0; i < 16; ++i) umax16[i] = e->umax[i];
}

int orc_extract(orc_extractor* h, const uint8_t* img, int W, int H, int stride, int lap0,
                int lap1, void* kps28, uint8_t* desc, int cap, int* n, int* mono) {
  return reinterpret_cast<Extractor*>(h)->extract(img, W, H, stride, lap0, lap1,
                                                  (KeyPoint*)kps28, desc, cap, n, mono);
}

// Stage dumps of the last orc_extract call.
int orc_level_dims(orc_extractor* h, int level, int* w, int* hh, int* stride) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  if (level < 0 || level >= (int)e->pyr.size()) return -1;
  *w = e->pyr[level].w; *hh = e->pyr[level].h; *stride = e->pyr[level].stride;
  return 0;
}
const uint8_t* orc_level_padded(orc_extractor* h, int level) {
  return reinterpret_cast<Extractor*>(h)->pyr[level].data.data();
}
int orc_level_candidates(orc_extractor* h, int level, int* xys, int cap) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  const std::vector<Cand>& c = e->cands[level];
  for (size_t i = 0; i < c.size() && (int)i < cap; ++i) {
    xys[3 * i] = c[i].x; xys[3 * i + 1] = c[i].y; xys[3 * i + 2] = c[i].score;
  }
  return (int)c.size();
}
int orc_level_keys(orc_extractor* h, int level, void* kps28, int* m10m01, int cap) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  const std::vector<KeyPoint>& k = e->keys[level];
  for (size_t i = 0; i < k.size() && (int)i < cap; ++i) {
    ((KeyPoint*)kps28)[i] = k[i];
    if (m10m01) { m10m01[2 * i] = e->moments[level][2 * i]; m10m01[2 * i + 1] = e->moments[level][2 * i + 1]; }
  }
  return (int)k.size();
}
const uint8_t* orc_level_blurred(orc_extractor* h, int level) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  return e->blurred[level].empty() ? nullptr : e->blurred[level].data();
}

// ---- stand-alone primitives for unit tests ---------------------------------
int orc_cv_round(float v) { return cvRoundF(v); }
float orc_fast_atan2(float y, float x) { return fastAtan2(y, x); }
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw,
                          int dh, int dstride) {
  resizeLinearU8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orc_gauss7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  gaussianBlur7(src, w, h, sstride, dst, dstride);
}
int orc_fast(const uint8_t* img, int cols, int rows, int stride, int threshold, int* xys, int cap) {
  std::vector<Cand> out;
  fast9_16(img, cols, rows, stride, threshold, out);
  for (size_t i = 0; i < out.size() && (int)i < cap; ++i) {
    xys[3 * i] = out[i].x; xys[3 * i + 1] = out[i].y; xys[3 * i + 2] = out[i].score;
  }
  return (int)out.size();
}
// Quadtree selection on an explicit candidate list (x, y relative to minX/minY).
int orc_octree(const int* xys, int n, int minX, int maxX, int minY, int maxY, int N, int* sel,
               int cap) {
  std::vector<Cand> c(n);
  for (int i = 0; i < n; ++i) c[i] = {xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]};
  std::vector<int> r = distributeQuadTree(c, minX, maxX, minY, maxY, N);
  for (size_t i = 0; i < r.size() && (int)i < cap; ++i) sel[i] = r[i];
  return (int)r.size();
}
// std::sort permutation of (size, ulx) records with the reference comparator;
// perm[i] = original index of the element that ends at position i.
void orc_sort_sized_nodes(const int* size, const int* ulx, int n, int* perm) {
  std::vector<QNode> pool(n);
  std::vector<SizedNode> v(n);
  for (int i = 0; i < n; ++i) { pool[i].ulx = ulx[i]; v[i] = std::make_pair(size[i], &pool[i]); }
  std::sort(v.begin(), v.end(), sizedNodeLess);
  for (int i = 0; i < n; ++i) perm[i] = (int)(v[i].second - pool.data());
}

int orc_hamming(const uint8_t* a, const uint8_t* b) { return hamming256(a, b); }

// Brute-force Hamming 2-NN with cv::BFMatcher::knnMatch(k=2) ordering:
// ascending distance, earliest train index first among equals
// (core/src/Frame.cc:1177).  Missing neighbours: idx 0xFFFFFFFF, dist 0xFFFF.
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int64_t nt, uint32_t* idx, uint16_t* dist) {
  for (int i = 0; i < nq; ++i) {
    int d0 = INT_MAX, d1 = INT_MAX;
    int64_t i0 = -1, i1 = -1;
    for (int64_t j = 0; j < nt; ++j) {
      int d = hamming256(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < d1) {
        if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
        else { d1 = d; i1 = j; }
      }
    }
    idx[2 * i] = i0 < 0 ? 0xFFFFFFFFu : (uint32_t)i0;
    idx[2 * i + 1] = i1 < 0 ? 0xFFFFFFFFu : (uint32_t)i1;
    dist[2 * i] = i0 < 0 ? 0xFFFF : (uint16_t)d0;
    dist[2 * i + 1] = i1 < 0 ? 0xFFFF : (uint16_t)d1;
  }
}

// Row-band stereo descriptor search, core/src/Frame.cc:840-908.
// bestDist[i] stays 100 (TH_HIGH) and bestIdx[i] = -1 when nothing qualified.
void orc_stereo_band_match(const void* kpsL28, const uint8_t* descL, int nL, const void* kpsR28,
                           const uint8_t* descR, int nR, const float* scaleFactors, int nRows,
                           float minD, float maxD, int* bestIdx, int* bestDist) {
  const KeyPoint* L = (const KeyPoint*)kpsL28;
  const KeyPoint* R = (const KeyPoint*)kpsR28;
  std::vector<std::vector<int>> rows(nRows);
  for (int iR = 0; iR < nR; ++iR) {
    const float y = R[iR].y;
    const float r = 2.0f * scaleFactors[R[iR].octave];
    const int maxr = (int)std::ceil(y + r), minr = (int)std::floor(y - r);
    for (int yi = minr; yi <= maxr; ++yi)
      if (yi >= 0 && yi < nRows) rows[yi].push_back(iR);
  }
  for (int iL = 0; iL < nL; ++iL) {
    bestIdx[iL] = -1; bestDist[iL] = 100;
    const float vL = L[iL].y, uL = L[iL].x;
    const int lvl = L[iL].octave;
    const int row = (int)vL;
    if (row < 0 || row >= nRows) continue;
    const std::vector<int>& c = rows[row];
    if (c.empty()) continue;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int best = 100, bi = -1;
    for (int iR : c) {
      if (R[iR].octave < lvl - 1 || R[iR].octave > lvl + 1) continue;
      const float uR = R[iR].x;
      if (uR >= minU && uR <= maxU) {
        int d = hamming256(descL + (size_t)iL * 32, descR + (size_t)iR * 32);
        if (d < best) { best = d; bi = iR; }
      }
    }
    bestIdx[iL] = bi; bestDist[iL] = best;
  }
}

// Frame::ComputeStereoMatches, core/src/Frame.cc:831-996: row-band descriptor search, 11x11 L1
// refinement over +-5 columns on pyramid level kpL.octave of both cameras, parabola fit,
// disparity/depth, then removal of matches whose L1 distance is >= 1.5*1.4*median.
// hl / hr hold the pyramids of the last orc_extract calls on the left / right image; the
// keypoints and descriptors are those calls' outputs.  uRight / depth hold nL floats (-1 = no
// match, :832-833); sadDist (optional) receives the winning L1 distance of every keypoint that
// reached :979, else -1.  Where the reference would index outside a level image (cv::Mat
// rowRange / colRange raise cv::Exception) or outside the row table, the keypoint is skipped.
int orc_stereo_match(orc_extractor* hl, orc_extractor* hr, const void* kpsL28, const uint8_t* descL,
                     int nL, const void* kpsR28, const uint8_t* descR, int nR, float mbf, float mb,
                     float* uRight, float* depth, int* sadDist) {
  Extractor* eL = reinterpret_cast<Extractor*>(hl);
  Extractor* eR = reinterpret_cast<Extractor*>(hr);
  if (eL->pyr.empty() || eR->pyr.size() != eL->pyr.size()) return -1;
  const KeyPoint* L = (const KeyPoint*)kpsL28;
  const KeyPoint* R = (const KeyPoint*)kpsR28;
  for (int i = 0; i < nL; ++i) { uRight[i] = -1.0f; depth[i] = -1.0f; if (sadDist) sadDist[i] = -1; }
  const int thOrbDist = (100 + 50) / 2;                     // :835
  const int nRows = eL->pyr[0].h;                           // :837
  const float minZ = mb, minD = 0, maxD = mbf / minZ;       // :859-861
  std::vector<int> bestIdx(nL), bestDistV(nL);
  orc_stereo_band_match(kpsL28, descL, nL, kpsR28, descR, nR, eL->scale.data(), nRows, minD, maxD,
                        bestIdx.data(), bestDistV.data());  // :840-908
  std::vector<std::pair<int, int>> vDistIdx;
  for (int iL = 0; iL < nL; ++iL) {
    if (!(bestDistV[iL] < thOrbDist)) continue;             // :911
    const KeyPoint& kpL = L[iL];
    const float uL = kpL.x;
    const float uR0 = R[bestIdx[iL]].x;
    const float scaleFactor = eL->invScale[kpL.octave];
    const float scaleduL = std::round(kpL.x * scaleFactor);
    const float scaledvL = std::round(kpL.y * scaleFactor);
    const float scaleduR0 = std::round(uR0 * scaleFactor);
    const int w = 5, Lw = 5;
    const Plane& PL = eL->pyr[kpL.octave];
    const Plane& PR = eR->pyr[kpL.octave];
    const float iniu = scaleduR0 + Lw - w;
    const float endu = scaleduR0 + Lw + w + 1;
    const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
    // left window inside the left level (rowRange/colRange preconditions, :921-923)
    if (cv - w < 0 || cv + w + 1 > PL.h || cu - w < 0 || cu + w + 1 > PL.w) continue;
    if (iniu < 0 || endu >= PR.w) continue;                 // :933-935
    if (cv + w + 1 > PR.h || cr - Lw - w < 0) continue;     // right windows inside the right level
    int bestDist = INT_MAX, bestincR = 0;
    float vDists[2 * 5 + 1];
    for (int incR = -Lw; incR <= Lw; ++incR) {
      int sum = 0;                                          // cv::norm(IL, IR, NORM_L1), :943
      for (int dy = -w; dy <= w; ++dy) {
        const uint8_t* a = PL.roi() + (size_t)(cv + dy) * PL.stride + cu;
        const uint8_t* b = PR.roi() + (size_t)(cv + dy) * PR.stride + cr + incR;
        for (int dx = -w; dx <= w; ++dx) sum += std::abs((int)a[dx] - (int)b[dx]);
      }
      const float dist = (float)(double)sum;
      if (dist < bestDist) { bestDist = (int)dist; bestincR = incR; }
      vDists[Lw + incR] = dist;
    }
    if (bestincR == -Lw || bestincR == Lw) continue;        // :952-953
    const float dist1 = vDists[Lw + bestincR - 1];
    const float dist2 = vDists[Lw + bestincR];
    const float dist3 = vDists[Lw + bestincR + 1];
    const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
    if (deltaR < -1 || deltaR > 1) continue;
    float bestuR = eL->scale[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
    float disparity = (uL - bestuR);
    if (disparity >= minD && disparity < maxD) {
      if (disparity <= 0) {
        disparity = 0.01;
        bestuR = uL - 0.01;
      }
      depth[iL] = mbf / disparity;
      uRight[iL] = bestuR;
      if (sadDist) sadDist[iL] = bestDist;
      vDistIdx.push_back(std::make_pair(bestDist, iL));
    }
  }
  if (vDistIdx.empty()) return 0;                           // reference: undefined (reads [0])
  std::sort(vDistIdx.begin(), vDistIdx.end());              // :984
  const float median = vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  for (int i = (int)vDistIdx.size() - 1; i >= 0; --i) {
    if (vDistIdx[i].first < thDist) break;
    uRight[vDistIdx[i].second] = -1;
    depth[vDistIdx[i].second] = -1;
  }
  return 0;
}

// ---- frame grid (the consumer right after extraction) -------------------------------------
// cv::undistortPoints(src, dst, K, D, noArray(), P = K) as Frame::UndistortKeyPoints and
// Frame::ComputeImageBounds call it (core/src/Frame.cc:787-788, 814-815): OpenCV's iterative
// inverse of the radial-tangential model, 5 fixed-point iterations (the public entry point
// uses TermCriteria(MAX_ITER, 5, 0.01)), all in double, result stored as float.
// cam = fx, fy, cx, cy, k1, k2, p1, p2, k3.  xy / out hold n (x, y) pairs.
void orc_undistort_points(const float* cam, const float* xy, int n, float* out) {
  const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
  const double k1 = cam[4], k2 = cam[5], p1 = cam[6], p2 = cam[7], k3 = cam[8];
  const double ifx = 1. / fx, ify = 1. / fy;
  for (int i = 0; i < n; ++i) {
    const double u = xy[2 * i], v = xy[2 * i + 1];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
      if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
      const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
      const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    // new projection P = K, rectification = identity
    const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
    out[2 * i] = (float)(xx * ww);
    out[2 * i + 1] = (float)(yy * ww);
  }
}

// Frame::ComputeImageBounds, core/src/Frame.cc:801-829: bounds = mnMinX, mnMaxX, mnMinY, mnMaxY.
void orc_image_bounds(const float* cam, int cols, int rows, float* bounds) {
  if (cam && cam[4] != 0.0f) {
    const float c[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
    float m[8];
    orc_undistort_points(cam, c, 4, m);
    bounds[0] = std::min(m[0], m[4]);
    bounds[1] = std::max(m[2], m[6]);
    bounds[2] = std::min(m[1], m[3]);
    bounds[3] = std::max(m[5], m[7]);
  } else {
    bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows;
  }
}

// Frame::UndistortKeyPoints (:771-799) + Frame::AssignFeaturesToGrid / PosInGrid (:432-459,
// :751-762) for the monocular / rectified case (Nleft == -1).  kpsUn receives the keypoints
// with undistorted coordinates (a copy when k1 == 0, :772-775).  The grid is returned as a
// CSR over cells c = posX * 48 + posY (mGrid[posX][posY]): cellStart has 64*48+1 entries and
// entries lists keypoint indices in push_back order (ascending).
void orc_frame_grid(const float* cam, const float* bounds, const void* kps28, int n, void* kpsUn28,
                    int* cellStart, int* entries) {
  const KeyPoint* K = (const KeyPoint*)kps28;
  KeyPoint* U = (KeyPoint*)kpsUn28;
  for (int i = 0; i < n; ++i) U[i] = K[i];
  if (cam && cam[4] != 0.0f && n > 0) {
    std::vector<float> xy(2 * (size_t)n), o(2 * (size_t)n);
    for (int i = 0; i < n; ++i) { xy[2 * i] = K[i].x; xy[2 * i + 1] = K[i].y; }
    orc_undistort_points(cam, xy.data(), n, o.data());
    for (int i = 0; i < n; ++i) { U[i].x = o[2 * i]; U[i].y = o[2 * i + 1]; }
  }
  const int COLS = 64, ROWS = 48;
  const float invW = (float)COLS / (bounds[1] - bounds[0]);   // Frame.cc:193-196
  const float invH = (float)ROWS / (bounds[3] - bounds[2]);
  std::vector<std::vector<int>> grid(COLS * ROWS);
  for (int i = 0; i < n; ++i) {
    const int posX = (int)std::round((U[i].x - bounds[0]) * invW);
    const int posY = (int)std::round((U[i].y - bounds[2]) * invH);
    if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
    grid[posX * ROWS + posY].push_back(i);
  }
  int run = 0;
  for (int c = 0; c < COLS * ROWS; ++c) {
    cellStart[c] = run;
    for (int i : grid[c]) entries[run++] = i;
  }
  cellStart[COLS * ROWS] = run;
}

// Frame::GetFeaturesInArea, core/src/Frame.cc:682-749 (left / monocular grid), on the CSR
// above.  Returns the number of indices found; the first cap of them are written to out in
// the reference's order (ix ascending, iy ascending, cell order).
int orc_features_in_area(const float* bounds, const void* kpsUn28, const int* cellStart,
                         const int* entries, float x, float y, float r, int minLevel, int maxLevel,
                         int* out, int cap) {
  const KeyPoint* U = (const KeyPoint*)kpsUn28;
  const int COLS = 64, ROWS = 48;
  const float invW = (float)COLS / (bounds[1] - bounds[0]);
  const float invH = (float)ROWS / (bounds[3] - bounds[2]);
  const float factorX = r, factorY = r;
  const int nMinCellX = std::max(0, (int)std::floor((x - bounds[0] - factorX) * invW));
  if (nMinCellX >= COLS) return 0;
  const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x - bounds[0] + factorX) * invW));
  if (nMaxCellX < 0) return 0;
  const int nMinCellY = std::max(0, (int)std::floor((y - bounds[2] - factorY) * invH));
  if (nMinCellY >= ROWS) return 0;
  const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y - bounds[2] + factorY) * invH));
  if (nMaxCellY < 0) return 0;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  int found = 0;
  for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
    for (int iy = nMinCellY; iy <= nMaxCellY; ++iy)
      for (int e = cellStart[ix * ROWS + iy]; e < cellStart[ix * ROWS + iy + 1]; ++e) {
        const KeyPoint& kp = U[entries[e]];
        if (bCheckLevels) {
          if (kp.octave < minLevel) continue;
          if (maxLevel >= 0 && kp.octave > maxLevel) continue;
        }
        const float distx = kp.x - x, disty = kp.y - y;
        if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) {
          if (found < cap) out[found] = entries[e];
          ++found;
        }
      }
  return found;
}

// ---- guided matching (the per-frame tracking matchers that walk the frame grid) ----------------
// ORBmatcher::ComputeThreeMaxima, core/lib/ORB/matcher.cc:1925-1957.  sizes = bin sizes.
void orc_three_maxima(const int* sizes, int L, int* ind) {
  int max1 = 0, max2 = 0, max3 = 0;
  int ind1 = -1, ind2 = -1, ind3 = -1;
  for (int i = 0; i < L; ++i) {
    const int s = sizes[i];
    if (s > max1) {
      max3 = max2; max2 = max1; max1 = s;
      ind3 = ind2; ind2 = ind1; ind1 = i;
    } else if (s > max2) {
      max3 = max2; max2 = s;
      ind3 = ind2; ind2 = i;
    } else if (s > max3) {
      max3 = s;
      ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
  ind[0] = ind1; ind[1] = ind2; ind[2] = ind3;
}

namespace {
const int kHistoLength = 30;   // ORBmatcher::HISTO_LENGTH, matcher.cc:36
// rotation bin exactly as written at matcher.cc:694-700 (the product with 1/HISTO_LENGTH, not a
// division by the bin width: bins 0..12 are the ones that occur)
inline int rotationBin(float angle1, float angle2) {
  const float factor = 1.0f / kHistoLength;
  float rot = angle1 - angle2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == kHistoLength) bin = 0;
  return bin;
}
}  // namespace

// ORBmatcher::SearchForInitialization, core/lib/ORB/matcher.cc:629-733, on flat arrays.
// F2's grid is the CSR of orc_frame_grid (bounds, cellStart, entries over kps2Un).
// prevMatched: n1 (x, y) pairs, updated as at :728-730.  matches12: n1 entries (-1 = none).
int orc_search_for_initialization(const void* kps1Un28, const uint8_t* desc1, int n1,
                                  const void* kps2Un28, const uint8_t* desc2, int n2,
                                  const float* bounds2, const int* cellStart2, const int* entries2,
                                  float* prevMatched, int windowSize, float nnratio,
                                  int checkOrientation, int* matches12) {
  const KeyPoint* K1 = (const KeyPoint*)kps1Un28;
  const KeyPoint* K2 = (const KeyPoint*)kps2Un28;
  const int TH_LOW = 50;
  int nmatches = 0;
  for (int i = 0; i < n1; ++i) matches12[i] = -1;
  std::vector<std::vector<int>> rotHist(kHistoLength);
  std::vector<int> matchedDistance(n2, INT_MAX), matches21(n2, -1);
  std::vector<int> idx(std::max(n2, 1));
  for (int i1 = 0; i1 < n1; ++i1) {
    const int level1 = K1[i1].octave;
    if (level1 > 0) continue;
    const int found = orc_features_in_area(bounds2, kps2Un28, cellStart2, entries2, prevMatched[2 * i1],
                                           prevMatched[2 * i1 + 1], (float)windowSize, level1, level1,
                                           idx.data(), n2);
    if (found == 0) continue;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int c = 0; c < found; ++c) {
      const int i2 = idx[c];
      const int dist = hamming256(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
      if (matchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * nnratio) {
        if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
        matches12[i1] = bestIdx2;
        matches21[bestIdx2] = i1;
        matchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (checkOrientation) rotHist[rotationBin(K1[i1].angle, K2[bestIdx2].angle)].push_back(i1);
      }
    }
  }
  if (checkOrientation) {
    int sizes[kHistoLength], ind[3];
    for (int i = 0; i < kHistoLength; ++i) sizes[i] = (int)rotHist[i].size();
    orc_three_maxima(sizes, kHistoLength, ind);
    for (int i = 0; i < kHistoLength; ++i) {
      if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
      for (int idx1 : rotHist[i])
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < n1; ++i1)
    if (matches12[i1] >= 0) { prevMatched[2 * i1] = K2[matches12[i1]].x; prevMatched[2 * i1 + 1] = K2[matches12[i1]].y; }
  return nmatches;
}

// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), core/lib/ORB/matcher.cc:
// 1591-1808, for a monocular or rectified current frame (Nleft == -1), on flat arrays.  The
// projection of the last frame's map points (pose and camera model, :1614-1630) is the caller's:
//   lastValid[i]   map point present and not an outlier (:1611-1613)
//   uv[2i..]       its projection into the current frame; invz[i] = 1 / depth in the camera frame
//   lastOctave[i], lastAngle[i]  octave / angle of the last frame's keypoint i (:1632-1635, :1691)
//   lastDesc       the map point's representative descriptor (:1655), 32 bytes each
//   lastObs[i]     pMP->Observations() > 0, consulted when a later candidate meets this match (:1665)
// Current frame: undistorted keypoints, descriptors, mvuRight (may be NULL: monocular), the grid CSR
// with its bounds, the scale factors, and state[i2] = 0 free / 1 holds a map point with
// observations / 2 holds one without.  Outputs: assigned[i2] = last-frame index matched to
// keypoint i2, -1 untouched, -2 cleared by the rotation check (:1799-1800).  Returns nmatches.
int orc_search_by_projection_last(const uint8_t* lastValid, const float* uv, const float* invz,
                                  const int* lastOctave, const float* lastAngle, const uint8_t* lastDesc,
                                  const uint8_t* lastObs, int nLast, const void* kpsUn28,
                                  const uint8_t* desc, const float* uRight, const uint8_t* state, int n,
                                  const float* bounds, const int* cellStart, const int* entries,
                                  const float* scaleFactors, float th, int forward, int backward, float mbf,
                                  int checkOrientation, int* assigned) {
  const KeyPoint* K = (const KeyPoint*)kpsUn28;
  const int TH_HIGH = 100;
  int nmatches = 0;
  for (int i = 0; i < n; ++i) assigned[i] = -1;
  // whether keypoint i2 currently holds a map point with observations: initial state, then
  // whatever this loop assigned
  std::vector<uint8_t> blocked(std::max(n, 1));
  for (int i = 0; i < n; ++i) blocked[i] = state[i] == 1;
  std::vector<std::vector<int>> rotHist(kHistoLength);
  std::vector<int> idx(std::max(n, 1));
  for (int i = 0; i < nLast; ++i) {
    if (!lastValid[i]) continue;
    const float invzc = invz[i];
    if (invzc < 0) continue;
    const float u = uv[2 * i], v = uv[2 * i + 1];
    if (u < bounds[0] || u > bounds[1]) continue;
    if (v < bounds[2] || v > bounds[3]) continue;
    const int nLastOctave = lastOctave[i];
    const float radius = th * scaleFactors[nLastOctave];
    int found;
    if (forward) found = orc_features_in_area(bounds, kpsUn28, cellStart, entries, u, v, radius, nLastOctave, -1, idx.data(), n);
    else if (backward) found = orc_features_in_area(bounds, kpsUn28, cellStart, entries, u, v, radius, 0, nLastOctave, idx.data(), n);
    else found = orc_features_in_area(bounds, kpsUn28, cellStart, entries, u, v, radius, nLastOctave - 1, nLastOctave + 1, idx.data(), n);
    if (found == 0) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (int c = 0; c < found; ++c) {
      const int i2 = idx[c];
      if (blocked[i2]) continue;
      if (uRight && uRight[i2] > 0) {
        const float ur = u - mbf * invzc;
        const float er = std::fabs(ur - uRight[i2]);
        if (er > radius) continue;
      }
      const int dist = hamming256(lastDesc + (size_t)i * 32, desc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      assigned[bestIdx2] = i;
      blocked[bestIdx2] = lastObs[i] != 0;
      nmatches++;
      if (checkOrientation) rotHist[rotationBin(lastAngle[i], K[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (checkOrientation) {
    int sizes[kHistoLength], ind[3];
    for (int i = 0; i < kHistoLength; ++i) sizes[i] = (int)rotHist[i].size();
    orc_three_maxima(sizes, kHistoLength, ind);
    for (int i = 0; i < kHistoLength; ++i) {
      if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
      for (int i2 : rotHist[i]) { assigned[i2] = -2; nmatches--; }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints), core/lib/ORB/
// matcher.cc:41-213, for a monocular or rectified frame (Nleft == -1; the right-camera half,
// :144-210, only runs for fisheye pairs), on flat arrays.  What Frame::isInFrustum left in every
// map point is the caller's:
//   inView[i]    mbTrackInView, not bad, and not beyond thFarPoints when bFarPoints (:51-60)
//   proj[2i..]   mTrackProjX, mTrackProjY;  projXR[i]  mTrackProjXR
//   level[i]     mnTrackScaleLevel;  viewCos[i]  mTrackViewCos
//   ptDesc       the map point's descriptor, 32 bytes each
//   ptObs[i]     pMP->Observations() > 0, consulted when a later map point meets this match (:89-91)
// Frame: as in orc_search_by_projection_last.  assigned[i2] = index of the map point keypoint i2
// received (-1 untouched).  Returns nmatches (every assignment counts, :137).
int orc_search_by_projection_points(const uint8_t* inView, const float* proj, const float* projXR,
                                    const int* level, const float* viewCos, const uint8_t* ptDesc,
                                    const uint8_t* ptObs, int nPts, const void* kpsUn28, const uint8_t* desc,
                                    const float* uRight, const uint8_t* state, int n, const float* bounds,
                                    const int* cellStart, const int* entries, const float* scaleFactors,
                                    float th, float nnratio, int* assigned) {
  const KeyPoint* K = (const KeyPoint*)kpsUn28;
  const int TH_HIGH = 100;
  int nmatches = 0;
  for (int i = 0; i < n; ++i) assigned[i] = -1;
  std::vector<uint8_t> blocked(std::max(n, 1));
  for (int i = 0; i < n; ++i) blocked[i] = state[i] == 1;
  std::vector<int> idx(std::max(n, 1));
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < nPts; ++iMP) {
    if (!inView[iMP]) continue;
    const int nPredictedLevel = level[iMP];
    float r = viewCos[iMP] > 0.998 ? 2.5f : 4.0f;   // RadiusByViewingCos, :215-220 (compared as double)
    if (bFactor) r *= th;
    const float radius = r * scaleFactors[nPredictedLevel];
    const int found = orc_features_in_area(bounds, kpsUn28, cellStart, entries, proj[2 * iMP], proj[2 * iMP + 1],
                                           radius, nPredictedLevel - 1, nPredictedLevel, idx.data(), n);
    if (found == 0) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int c = 0; c < found; ++c) {
      const int i2 = idx[c];
      if (blocked[i2]) continue;
      if (uRight && uRight[i2] > 0) {
        const float er = std::fabs(projXR[iMP] - uRight[i2]);
        if (er > radius) continue;
      }
      const int dist = hamming256(ptDesc + (size_t)iMP * 32, desc + (size_t)i2 * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist;
        bestLevel2 = bestLevel; bestLevel = K[i2].octave;
        bestIdx = i2;
      } else if (dist < bestDist2) {
        bestLevel2 = K[i2].octave;
        bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
      if (bestLevel != bestLevel2 || (float)bestDist <= nnratio * (float)bestDist2) {
        assigned[bestIdx] = iMP;
        blocked[bestIdx] = ptObs[iMP] != 0;
        nmatches++;
      }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), core/lib/ORB/matcher.cc:1591-1808,
// for a current frame with two cameras (CurrentFrame.Nleft != -1: a fisheye rig): after the left
// camera's search (:1632-1711) every last-frame keypoint that got that far is searched again
// among the RIGHT camera's keypoints around its projection into the right camera (:1712-1786).
// The two keypoint sets are given separately (left: kpsL, descL, gridL; right: kpsR, descR, gridR;
// the reference keeps them in one array with the right ones at index + Nleft); uvR is the
// projection through Trl (:1713-1714), computed by the caller like uvL.  Quirks kept: the left
// camera's checks gate the right camera's search (behind the camera :1624, left projection
// outside the image :1629-1632, and an EMPTY left window, whose `continue` at :1652 leaves the
// loop body); the right projection is not checked against the image; with two cameras the
// mvuRight test of :1669-1674 does not apply.  One rotation histogram for both cameras.
int orc_search_by_projection_last_rig(const uint8_t* lastValid, const float* uvL, const float* invz,
                                      const float* uvR, const int* lastOctave, const float* lastAngle,
                                      const uint8_t* lastDesc, const uint8_t* lastObs, int nLast,
                                      const void* kpsL28, const uint8_t* descL, const uint8_t* stateL, int nL,
                                      const int* cellStartL, const int* entriesL, const void* kpsR28,
                                      const uint8_t* descR, const uint8_t* stateR, int nR, const int* cellStartR,
                                      const int* entriesR, const float* bounds, const float* scaleFactors,
                                      float th, int forward, int backward, int checkOrientation, int* assignedL,
                                      int* assignedR) {
  const KeyPoint* KL = (const KeyPoint*)kpsL28;
  const KeyPoint* KR = (const KeyPoint*)kpsR28;
  const int TH_HIGH = 100;
  int nmatches = 0;
  for (int i = 0; i < nL; ++i) assignedL[i] = -1;
  for (int i = 0; i < nR; ++i) assignedR[i] = -1;
  std::vector<uint8_t> blockedL(std::max(nL, 1)), blockedR(std::max(nR, 1));
  for (int i = 0; i < nL; ++i) blockedL[i] = stateL[i] == 1;
  for (int i = 0; i < nR; ++i) blockedR[i] = stateR[i] == 1;
  // histogram entries: keypoint index, +nL for the right camera (the reference's index + Nleft)
  std::vector<std::vector<int>> rotHist(kHistoLength);
  std::vector<int> idx(std::max(std::max(nL, nR), 1));
  for (int i = 0; i < nLast; ++i) {
    if (!lastValid[i]) continue;
    if (invz[i] < 0) continue;
    const float u = uvL[2 * i], v = uvL[2 * i + 1];
    if (u < bounds[0] || u > bounds[1]) continue;
    if (v < bounds[2] || v > bounds[3]) continue;
    const int nLastOctave = lastOctave[i];
    const float radius = th * scaleFactors[nLastOctave];
    int minL, maxL;
    if (forward) { minL = nLastOctave; maxL = -1; }
    else if (backward) { minL = 0; maxL = nLastOctave; }
    else { minL = nLastOctave - 1; maxL = nLastOctave + 1; }
    int found = orc_features_in_area(bounds, kpsL28, cellStartL, entriesL, u, v, radius, minL, maxL, idx.data(), nL);
    if (found == 0) continue;   // :1652, also skips the right camera
    int bestDist = 256, bestIdx2 = -1;
    for (int c = 0; c < found; ++c) {
      const int i2 = idx[c];
      if (blockedL[i2]) continue;
      const int dist = hamming256(lastDesc + (size_t)i * 32, descL + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      assignedL[bestIdx2] = i;
      blockedL[bestIdx2] = lastObs[i] != 0;
      nmatches++;
      if (checkOrientation) rotHist[rotationBin(lastAngle[i], KL[bestIdx2].angle)].push_back(bestIdx2);
    }
    // ---- the right camera, :1712-1786
    found = orc_features_in_area(bounds, kpsR28, cellStartR, entriesR, uvR[2 * i], uvR[2 * i + 1], radius, minL, maxL,
                                 idx.data(), nR);
    bestDist = 256; bestIdx2 = -1;
    for (int c = 0; c < found; ++c) {
      const int i2 = idx[c];
      if (blockedR[i2]) continue;
      const int dist = hamming256(lastDesc + (size_t)i * 32, descR + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      assignedR[bestIdx2] = i;
      blockedR[bestIdx2] = lastObs[i] != 0;
      nmatches++;
      if (checkOrientation) rotHist[rotationBin(lastAngle[i], KR[bestIdx2].angle)].push_back(bestIdx2 + nL);
    }
  }
  if (checkOrientation) {
    int sizes[kHistoLength], ind[3];
    for (int i = 0; i < kHistoLength; ++i) sizes[i] = (int)rotHist[i].size();
    orc_three_maxima(sizes, kHistoLength, ind);
    for (int i = 0; i < kHistoLength; ++i) {
      if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
      for (int e : rotHist[i]) {
        if (e < nL) assignedL[e] = -2; else assignedR[e - nL] = -2;
        nmatches--;
      }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints), matcher.cc:41-213,
// for a frame with two cameras (F.Nleft != -1).  Per map point first the left camera's search
// (:64-142; without the mvuRight test, which needs F.Nleft == -1), whose match is also given to
// the keypoint's stereo partner in the right camera (mvLeftToRightMatch, :129-135); then, if the
// point is in the right camera's view and has a predicted level there, the right camera's search
// (:144-210), whose match is also given to the partner in the left camera (mvRightToLeftMatch,
// :197-203).  Quirks kept: a left search that finds a best candidate within TH_HIGH but fails
// the ratio test leaves the loop body (`continue`, :124-125) and so skips the right camera; the
// right camera's radius is not multiplied by th (:147); a partner receives the map point whatever
// it held before, so a keypoint that held a point with observations can come to hold one without.
// inView / inViewR already contain the far-point and isBad tests (:51-60).  leftToRight[i] /
// rightToLeft[j]: partner index in the other camera's own numbering, or -1.
int orc_search_by_projection_points_rig(const uint8_t* inView, const float* proj, const int* level,
                                        const float* viewCos, const uint8_t* inViewR, const float* projR,
                                        const int* levelR, const float* viewCosR, const uint8_t* ptDesc,
                                        const uint8_t* ptObs, int nPts, const void* kpsL28, const uint8_t* descL,
                                        const uint8_t* stateL, int nL, const int* cellStartL, const int* entriesL,
                                        const void* kpsR28, const uint8_t* descR, const uint8_t* stateR, int nR,
                                        const int* cellStartR, const int* entriesR, const int* leftToRight,
                                        const int* rightToLeft, const float* bounds, const float* scaleFactors,
                                        float th, float nnratio, int* assignedL, int* assignedR) {
  const KeyPoint* KL = (const KeyPoint*)kpsL28;
  const KeyPoint* KR = (const KeyPoint*)kpsR28;
  const int TH_HIGH = 100;
  int nmatches = 0;
  for (int i = 0; i < nL; ++i) assignedL[i] = -1;
  for (int i = 0; i < nR; ++i) assignedR[i] = -1;
  std::vector<uint8_t> blockedL(std::max(nL, 1)), blockedR(std::max(nR, 1));
  for (int i = 0; i < nL; ++i) blockedL[i] = stateL[i] == 1;
  for (int i = 0; i < nR; ++i) blockedR[i] = stateR[i] == 1;
  std::vector<int> idx(std::max(std::max(nL, nR), 1));
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < nPts; ++iMP) {
    if (!inView[iMP] && !inViewR[iMP]) continue;
    const uint8_t obs = ptObs[iMP] != 0;
    if (inView[iMP]) {
      const int nPredictedLevel = level[iMP];
      float r = viewCos[iMP] > 0.998 ? 2.5f : 4.0f;
      if (bFactor) r *= th;
      const float radius = r * scaleFactors[nPredictedLevel];
      const int found = orc_features_in_area(bounds, kpsL28, cellStartL, entriesL, proj[2 * iMP], proj[2 * iMP + 1],
                                             radius, nPredictedLevel - 1, nPredictedLevel, idx.data(), nL);
      if (found > 0) {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < found; ++c) {
          const int i2 = idx[c];
          if (blockedL[i2]) continue;
          const int dist = hamming256(ptDesc + (size_t)iMP * 32, descL + (size_t)i2 * 32);
          if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist;
            bestLevel2 = bestLevel; bestLevel = KL[i2].octave;
            bestIdx = i2;
          } else if (dist < bestDist2) {
            bestLevel2 = KL[i2].octave;
            bestDist2 = dist;
          }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;   // :124-125
          if (bestLevel != bestLevel2 || (float)bestDist <= nnratio * (float)bestDist2) {
            assignedL[bestIdx] = iMP;
            blockedL[bestIdx] = obs;
            if (leftToRight[bestIdx] != -1) {
              assignedR[leftToRight[bestIdx]] = iMP;
              blockedR[leftToRight[bestIdx]] = obs;
              nmatches++;
            }
            nmatches++;
          }
        }
      }
    }
    if (inViewR[iMP]) {
      const int nPredictedLevel = levelR[iMP];
      if (nPredictedLevel != -1) {
        const float r = viewCosR[iMP] > 0.998 ? 2.5f : 4.0f;
        const float radius = r * scaleFactors[nPredictedLevel];
        const int found = orc_features_in_area(bounds, kpsR28, cellStartR, entriesR, projR[2 * iMP],
                                               projR[2 * iMP + 1], radius, nPredictedLevel - 1, nPredictedLevel,
                                               idx.data(), nR);
        if (found == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < found; ++c) {
          const int i2 = idx[c];
          if (blockedR[i2]) continue;
          const int dist = hamming256(ptDesc + (size_t)iMP * 32, descR + (size_t)i2 * 32);
          if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist;
            bestLevel2 = bestLevel; bestLevel = KR[i2].octave;
            bestIdx = i2;
          } else if (dist < bestDist2) {
            bestLevel2 = KR[i2].octave;
            bestDist2 = dist;
          }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
          if (rightToLeft[bestIdx] != -1) {
            assignedL[rightToLeft[bestIdx]] = iMP;
            blockedL[rightToLeft[bestIdx]] = obs;
            nmatches++;
          }
          assignedR[bestIdx] = iMP;
          blockedR[bestIdx] = obs;
          nmatches++;
        }
      }
    }
  }
  return nmatches;
}

// Frame::ComputeStereoFromRGBD, core/src/Frame.cc:998-1016.  depthImg: H rows of stride floats.
void orc_stereo_from_rgbd(const void* kps28, const void* kpsUn28, int n, const float* depthImg, int stride,
                          float mbf, float* uRight, float* depth) {
  const KeyPoint* K = (const KeyPoint*)kps28;
  const KeyPoint* U = (const KeyPoint*)kpsUn28;
  for (int i = 0; i < n; ++i) {
    uRight[i] = -1; depth[i] = -1;
    const float v = K[i].y, u = K[i].x;
    const float d = depthImg[(size_t)(int)v * stride + (int)u];   // Mat::at<float>(v, u): indices truncate
    if (d > 0) {
      depth[i] = d;
      uRight[i] = U[i].x - mbf / d;
    }
  }
}

// ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), core/lib/ORB/
// matcher.cc:1810-1923 (relocalisation), on flat arrays.  Per keypoint of the keyframe, the
// caller's part (:1826-1853):
//   kfValid[i]   a map point exists, is not bad, is not in sAlreadyFound, and its distance to the
//                camera centre lies inside its scale-invariance range
//   uv[2i..]     its projection;  level[i]  PredictScale(dist3D);  kfAngle[i]  pKF->mvKeysUn[i].angle
// hasPoint[i2]: CurrentFrame.mvpMapPoints[i2] is not null (any map point closes the keypoint, :1872).
// assigned / return value as in orc_search_by_projection_last.
int orc_search_by_projection_keyframe(const uint8_t* kfValid, const float* uv, const int* level,
                                      const float* kfAngle, const uint8_t* kfDesc, int nKf, const void* kpsUn28,
                                      const uint8_t* desc, const uint8_t* hasPoint, int n, const float* bounds,
                                      const int* cellStart, const int* entries, const float* scaleFactors,
                                      float th, int orbDist, int checkOrientation, int* assigned) {
  const KeyPoint* K = (const KeyPoint*)kpsUn28;
  int nmatches = 0;
  for (int i = 0; i < n; ++i) assigned[i] = -1;
  std::vector<uint8_t> taken(std::max(n, 1));
  for (int i = 0; i < n; ++i) taken[i] = hasPoint[i] != 0;
  std::vector<std::vector<int>> rotHist(kHistoLength);
  std::vector<int> idx(std::max(n, 1));
  for (int i = 0; i < nKf; ++i) {
    if (!kfValid[i]) continue;
    const float u = uv[2 * i], v = uv[2 * i + 1];
    if (u < bounds[0] || u > bounds[1]) continue;
    if (v < bounds[2] || v > bounds[3]) continue;
    const int nPredictedLevel = level[i];
    const float radius = th * scaleFactors[nPredictedLevel];
    const int found = orc_features_in_area(bounds, kpsUn28, cellStart, entries, u, v, radius, nPredictedLevel - 1,
                                           nPredictedLevel + 1, idx.data(), n);
    if (found == 0) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (int c = 0; c < found; ++c) {
      const int i2 = idx[c];
      if (taken[i2]) continue;
      const int dist = hamming256(kfDesc + (size_t)i * 32, desc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    // (with ORBdist >= 256 and no free candidate the reference writes mvpMapPoints[-1]; callers
    // pass 64 or 100, and this restatement requires a candidate)
    if (bestDist <= orbDist && bestIdx2 >= 0) {
      assigned[bestIdx2] = i;
      taken[bestIdx2] = 1;
      nmatches++;
      if (checkOrientation) rotHist[rotationBin(kfAngle[i], K[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (checkOrientation) {
    int sizes[kHistoLength], ind[3];
    for (int i = 0; i < kHistoLength; ++i) sizes[i] = (int)rotHist[i].size();
    orc_three_maxima(sizes, kHistoLength, ind);
    for (int i = 0; i < kHistoLength; ++i) {
      if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
      for (int i2 : rotHist[i]) { assigned[i2] = -2; nmatches--; }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches), core/lib/ORB/matcher.cc:222-405, for a
// monocular or rectified frame (F.Nleft == -1, no second camera), on flat arrays.  Both feature
// vectors (DBoW2::FeatureVector = std::map<NodeId, vector<unsigned>>) arrive as CSR in ascending
// node id, the maps' iteration order: nodeIds[r], features[nodeStart[r] .. nodeStart[r+1]).
//   kfValid[i]  the keyframe's map point i exists and is not bad (:252-258)
//   kfAngle[i]  pKF->mvKeysUn[i].angle (:321-325);  fAngle[i2]  F.mvKeys[i2].angle (:328-331)
// matches[i2] (nF entries) = keyframe keypoint whose map point frame keypoint i2 received, -1 none,
// -2 received and cleared by the rotation check (:394-401).  Returns nmatches.
int orc_search_by_bow(const uint8_t* kfValid, const float* kfAngle, const uint8_t* kfDesc, int nKfNodes,
                      const uint32_t* kfNodeIds, const int* kfNodeStart, const uint32_t* kfFeatures,
                      const float* fAngle, const uint8_t* fDesc, int nF, int nFNodes, const uint32_t* fNodeIds,
                      const int* fNodeStart, const uint32_t* fFeatures, float nnratio, int checkOrientation,
                      int* matches) {
  const int TH_LOW = 50;
  int nmatches = 0;
  for (int i = 0; i < nF; ++i) matches[i] = -1;
  std::vector<std::vector<int>> rotHist(kHistoLength);
  int a = 0, b = 0;
  while (a < nKfNodes && b < nFNodes) {
    if (kfNodeIds[a] == fNodeIds[b]) {
      for (int iKF = kfNodeStart[a]; iKF < kfNodeStart[a + 1]; ++iKF) {
        const unsigned realIdxKF = kfFeatures[iKF];
        if (!kfValid[realIdxKF]) continue;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int iF = fNodeStart[b]; iF < fNodeStart[b + 1]; ++iF) {
          const unsigned realIdxF = fFeatures[iF];
          if (matches[realIdxF] >= 0) continue;
          const int dist = hamming256(kfDesc + (size_t)realIdxKF * 32, fDesc + (size_t)realIdxF * 32);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
          else if (dist < bestDist2) { bestDist2 = dist; }
        }
        if (bestDist1 <= TH_LOW) {
          if ((float)bestDist1 < nnratio * (float)bestDist2) {
            matches[bestIdxF] = (int)realIdxKF;
            if (checkOrientation) rotHist[rotationBin(kfAngle[realIdxKF], fAngle[bestIdxF])].push_back(bestIdxF);
            nmatches++;
          }
        }
      }
      ++a; ++b;
    } else if (kfNodeIds[a] < fNodeIds[b]) {
      ++a;   // lower_bound(Fit->first) in a sorted map: skip ahead
    } else {
      ++b;
    }
  }
  if (checkOrientation) {
    int sizes[kHistoLength], ind[3];
    for (int i = 0; i < kHistoLength; ++i) sizes[i] = (int)rotHist[i].size();
    orc_three_maxima(sizes, kHistoLength, ind);
    for (int i = 0; i < kHistoLength; ++i) {
      if (i == ind[0] || i == ind[1] || i == ind[2]) continue;
      for (int i2 : rotHist[i]) { matches[i2] = -2; nmatches--; }
    }
  }
  return nmatches;
}

}  // extern "C"

// ---- bag of words (Frame::ComputeBoW, core/src/Frame.cc:764-769) -----------------------------
// The vocabulary class of the reference (core/lib/ORB/vocabulary.h:31-62) derives from DBoW2's
// TemplatedVocabulary<FORB::TDescriptor, FORB>.  DBoW2 is a git submodule (deps/DBoW2) whose
// directory is empty in the reference tree, so its version is not pinned there; what follows
// restates the published DBoW2 algorithm (TemplatedVocabulary::transform for one feature and
// for a feature vector, BowVector::addWeight / normalize, FeatureVector::addFeature,
// FORB::distance) - "parity unpinned" like the OpenCV primitives.  The file format is the
// reference's own: ORBVocabulary::loadFromText, core/lib/ORB/vocabulary.cc:96-160.  (Its
// binary loader, :180-227, sizes the node vector to 1 before a loop bounded by that size and
// reads no node, so the text format is the only one with defined behaviour.)
namespace {
struct VocNode {
  int parent = 0;
  std::vector<int> children;
  uint8_t desc[32] = {0};
  double weight = 0;
  unsigned word_id = 0;
};
struct Vocabulary {
  int k = 0, L = 0, scoring = 0, weighting = 0;
  std::vector<VocNode> nodes;   // nodes[0] = root
  int nWords = 0;
  // appends node nid = nodes.size() exactly as vocabulary.cc:130-156 does
  bool add(int pid, int isLeaf, const uint8_t* d, double w) {
    const int nid = (int)nodes.size();
    if (pid < 0 || pid >= nid) return false;   // the reference indexes m_nodes[pid] unchecked
    nodes.emplace_back();
    nodes[nid].parent = pid;
    nodes[pid].children.push_back(nid);
    std::memcpy(nodes[nid].desc, d, 32);
    nodes[nid].weight = w;
    if (isLeaf > 0) nodes[nid].word_id = (unsigned)nWords++;
    return true;
  }
};
}  // namespace

extern "C" {

// scoring: L1_NORM = 0; weighting: TF_IDF = 0, TF = 1, IDF = 2, BINARY = 3 (DBoW2 enums).
// Nodes 1..nNodes in file order: parent id, leaf flag, 32 descriptor bytes, weight.
void* orc_voc_create(int k, int L, int scoring, int weighting, int nNodes, const int* parent,
                     const uint8_t* isLeaf, const uint8_t* desc, const double* weight) {
  Vocabulary* v = new Vocabulary();
  v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
  v->nodes.resize(1);
  for (int i = 0; i < nNodes; ++i)
    if (!v->add(parent[i], isLeaf[i], desc + (size_t)i * 32, weight[i])) { delete v; return nullptr; }
  return v;
}

// ORBVocabulary::loadFromText, core/lib/ORB/vocabulary.cc:96-160.  Lines without a parent id
// (the empty line a trailing newline leaves, which the reference turns into a node with
// uninitialised contents) are skipped.
void* orc_voc_load_text(const char* path) {
  FILE* f = std::fopen(path, "r");
  if (!f) return nullptr;
  Vocabulary* v = new Vocabulary();
  int n1 = 0, n2 = 0;
  if (std::fscanf(f, "%d %d %d %d", &v->k, &v->L, &n1, &n2) != 4 || v->k < 0 || v->k > 20 || v->L < 1 ||
      v->L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
    std::fclose(f); delete v; return nullptr;
  }
  v->scoring = n1; v->weighting = n2;
  v->nodes.resize(1);
  for (;;) {
    int pid, leaf;
    const int got = std::fscanf(f, "%d %d", &pid, &leaf);
    if (got != 2) {   // only the end of the file ends the node list; a damaged line fails the load
      if (!(got == EOF && std::feof(f))) { std::fclose(f); delete v; return nullptr; }
      break;
    }
    uint8_t d[32];
    bool ok = true;
    for (int i = 0; i < 32 && ok; ++i) {
      int b;
      ok = std::fscanf(f, "%d", &b) == 1;
      d[i] = (uint8_t)b;
    }
    double w = 0;
    ok = ok && std::fscanf(f, "%lf", &w) == 1;
    if (!ok || !v->add(pid, leaf, d, w)) { std::fclose(f); delete v; return nullptr; }
  }
  std::fclose(f);
  return v;
}

void orc_voc_destroy(void* h) { delete (Vocabulary*)h; }

void orc_voc_info(void* h, int* k, int* L, int* nNodes, int* nWords) {
  Vocabulary* v = (Vocabulary*)h;
  *k = v->k; *L = v->L; *nNodes = (int)v->nodes.size(); *nWords = v->nWords;
}

// TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) for TF_IDF / TF
// weighting and L1 scoring (the stock configuration; anything else returns -1).
//   per feature: descend from the root, at every level the child at the smallest Hamming
//     distance (first one wins ties); the node met at level L - levelsup is the feature's node
//     id (0 when L - levelsup <= 0, and - where the original leaves it uninitialised, a leaf
//     above that level - also 0); the leaf's word id and weight;
//   BowVector (std::map<WordId, double>): weights with w > 0 are added up per word in feature
//     order, then divided by the L1 norm summed in ascending word order;
//   FeatureVector (std::map<NodeId, vector<unsigned>>): feature indices per node id.
// Outputs: words ascending with their values; node ids ascending with a CSR of feature indices.
int orc_bow_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* wordIds, double* wordVals,
                      int* nWordsOut, uint32_t* nodeIds, int* nodeStart, uint32_t* feats, int* nNodesOut) {
  Vocabulary* v = (Vocabulary*)h;
  *nWordsOut = 0; *nNodesOut = 0;
  if (v->scoring != 0 || (v->weighting != 0 && v->weighting != 1)) return -1;
  if (v->nodes.size() <= 1) return 0;   // empty()
  std::map<unsigned, double> bow;
  std::map<unsigned, std::vector<unsigned>> fv;
  const int nid_level = v->L - levelsup;
  for (int i = 0; i < n; ++i) {
    const uint8_t* f = desc + (size_t)i * 32;
    unsigned nid = 0;
    int final_id = 0, current_level = 0;
    do {
      ++current_level;
      const std::vector<int>& ch = v->nodes[final_id].children;
      final_id = ch[0];
      double best_d = hamming256(f, v->nodes[final_id].desc);
      for (size_t c = 1; c < ch.size(); ++c) {
        const double d = hamming256(f, v->nodes[ch[c]].desc);
        if (d < best_d) { best_d = d; final_id = ch[c]; }
      }
      if (current_level == nid_level) nid = (unsigned)final_id;
    } while (!v->nodes[final_id].children.empty());
    const unsigned word = v->nodes[final_id].word_id;
    // the stored weight under TF as under TF_IDF: the descent returns Node::weight and the
    // weighting mode only decides how weights are computed when a vocabulary is trained
    const double w = v->nodes[final_id].weight;
    if (w > 0) {
      auto it = bow.lower_bound(word);
      if (it != bow.end() && !(bow.key_comp()(word, it->first))) it->second += w;
      else bow.insert(it, std::make_pair(word, w));
      fv[nid].push_back((unsigned)i);
    }
  }
  double norm = 0.0;
  for (auto& kv : bow) norm += std::fabs(kv.second);
  if (norm > 0.0)
    for (auto& kv : bow) kv.second /= norm;
  int a = 0;
  for (auto& kv : bow) { wordIds[a] = kv.first; wordVals[a] = kv.second; ++a; }
  *nWordsOut = a;
  int b = 0, run = 0;
  for (auto& kv : fv) {
    nodeIds[b] = kv.first;
    nodeStart[b] = run;
    for (unsigned idx : kv.second) feats[run++] = idx;
    ++b;
  }
  nodeStart[b] = run;
  *nNodesOut = b;
  return 0;
}

}  // extern "C"
