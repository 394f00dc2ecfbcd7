// orbx_cv::ORBVocabulary / orbx_cv::ComputeBoW (include/orbx_cv.hpp) against the C ABI they wrap,
// compiled with the test-only OpenCV stand-in of tests/cpp/cvstub.  Needs a GPU.
// usage: test_adaptor_bow <vocabulary.txt>; prints "ok ..." on success.
#include <cstdio>
#include <map>
#include <vector>

#include "orbx_cv.hpp"

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: %s vocabulary.txt\n", argv[0]); return 2; }
  const int W = 752, H = 480;
  cv::Mat img(H, W, CV_8UC1);
  if (orbx_synth_frame(77, W, H, (int)img.step, img.data) != ORBX_OK) return 1;
  orbx_cv::ORBextractor ex(1000, 1.2f, 8, 20, 7);
  std::vector<cv::KeyPoint> kps;
  cv::Mat desc;
  std::vector<int> lap = {0, 0};
  ex(img, cv::Mat(), kps, desc, lap);
  orbx_cv::ORBVocabulary voc(argv[1]);
  if (voc.empty()) { std::printf("vocabulary not loaded\n"); return 1; }
  std::map<unsigned, double> bow;
  std::map<unsigned, std::vector<unsigned>> fv;
  orbx_cv::ComputeBoW(ex, voc, bow, fv, 2);
  // the same through the C ABI
  const int cap = orbx_max_keypoints(ex.handle());
  std::vector<uint32_t> words(cap), nodes(cap), feats(cap);
  std::vector<double> values(cap);
  std::vector<int32_t> start(cap + 1);
  int nw = 0, nn = 0;
  if (orbx_bow_download(ex.handle(), 0, words.data(), values.data(), &nw, nodes.data(), start.data(), feats.data(),
                        &nn, cap) != ORBX_OK)
    return 1;
  if ((int)bow.size() != nw || (int)fv.size() != nn || nw < 10) { std::printf("sizes differ\n"); return 1; }
  int i = 0;
  double sum = 0;
  for (auto& kv : bow) {
    if (kv.first != words[i] || kv.second != values[i]) { std::printf("word %d differs\n", i); return 1; }
    sum += kv.second;
    ++i;
  }
  int r = 0;
  size_t total = 0;
  for (auto& kv : fv) {
    if (kv.first != nodes[r] || (int)kv.second.size() != start[r + 1] - start[r]) { std::printf("node %d differs\n", r); return 1; }
    for (size_t k = 0; k < kv.second.size(); ++k)
      if (kv.second[k] != feats[start[r] + k]) { std::printf("feature list %d differs\n", r); return 1; }
    total += kv.second.size();
    ++r;
  }
  if (sum < 0.999999 || sum > 1.000001 || total > kps.size()) { std::printf("norm %f\n", sum); return 1; }
  std::printf("ok %d words, %d nodes, %zu features of %zu\n", nw, nn, total, kps.size());
  return 0;
}
