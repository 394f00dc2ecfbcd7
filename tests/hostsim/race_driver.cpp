// Calls every entry point of the C ABI once on small inputs.  Built together with the kernels
// under -fsanitize=thread (make race), it is what the race check runs: results are not compared
// here (the parity tests do that), only reports of the race detector count.  Test
// infrastructure only.
#include <orbx.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CHECK(call)                                                                        \
  do {                                                                                     \
    const int st_ = (call);                                                                \
    if (st_ != ORBX_OK) {                                                                  \
      fprintf(stderr, "race_driver: %s -> %d (%s)\n", #call, st_, orbx_strerror(st_));     \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)

extern "C" {
// HIP runtime subset of the simulation, for the calls that take device pointers
int hipMalloc(void** p, size_t n);
int hipFree(void* p);
int hipMemcpy(void* d, const void* s, size_t n, int kind);
}

// content: "synth" = the textured bench frame; "noise" = uniform noise (far more candidates than
// features wanted); "faint" = two gray values 19 apart (nothing at the first threshold, more
// survivors of the quick test at the second than the detection kernel queues)
static const char* g_content = "synth";
static std::vector<uint8_t> synth(uint32_t seed, int w, int h) {
  std::vector<uint8_t> img((size_t)w * h);
  if (!strcmp(g_content, "synth")) {
    CHECK(orbx_synth_frame(seed, w, h, w, img.data()));
  } else {
    std::mt19937 r(seed);
    const bool faint = !strcmp(g_content, "faint");
    for (auto& p : img) p = faint ? ((r() & 1) ? 100 : 119) : (uint8_t)r();
  }
  return img;
}

// a copy of descriptors with a few random bits flipped in each, so that nearest neighbours
// are neither exact nor unique
static std::vector<uint8_t> perturbed(const uint8_t* d, int n, int flips, std::mt19937& rng) {
  std::vector<uint8_t> out(d, d + (size_t)n * 32);
  for (int i = 0; i < n; ++i)
    for (int f = 0; f < flips; ++f) {
      const unsigned bit = rng() % 256;
      out[(size_t)i * 32 + bit / 8] ^= (uint8_t)(1u << (bit % 8));
    }
  return out;
}

// right image of a rectified pair: the left one shifted by a few pixels
static std::vector<uint8_t> shifted(const std::vector<uint8_t>& src, int w, int h, int d) {
  std::vector<uint8_t> out((size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) out[(size_t)y * w + x] = src[(size_t)y * w + std::min(w - 1, x + d)];
  return out;
}

static void step(const char* what) {
  fprintf(stderr, "race_driver: %s\n", what);
  fflush(stderr);
}

int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 320, H = argc > 2 ? atoi(argv[2]) : 200;
  const int nfeatures = argc > 3 ? atoi(argv[3]) : 400;
  if (argc > 4) g_content = argv[4];
  const int NF = 2;
  std::mt19937 rng(7);

  orbx_params prm;
  prm.nfeatures = nfeatures;
  prm.scale_factor = 1.2f;
  prm.nlevels = 5;
  prm.ini_th_fast = 20;
  prm.min_th_fast = 7;
  prm.device_id = 0;
  prm.max_batch = NF;
  orbx_extractor *left = nullptr, *right = nullptr;
  CHECK(orbx_create(&prm, &left));
  CHECK(orbx_create(&prm, &right));
  const int cap = orbx_max_keypoints(left);

  std::vector<std::vector<uint8_t>> imgsL, imgsR;
  for (int f = 0; f < NF; ++f) {
    imgsL.push_back(synth(0x0B5EEDu ^ (uint32_t)f, W, H));
    imgsR.push_back(shifted(imgsL.back(), W, H, 6 + 3 * f));
  }
  const uint8_t* pl[NF] = {imgsL[0].data(), imgsL[1].data()};
  const uint8_t* pr[NF] = {imgsR[0].data(), imgsR[1].data()};

  std::vector<orbx_keypoint> kps((size_t)NF * cap), kpsR((size_t)NF * cap);
  std::vector<uint8_t> desc((size_t)NF * cap * 32), descR((size_t)NF * cap * 32);
  int n[NF], mono[NF], nR[NF], monoR[NF];

  step("monocular extraction, single frame (keys fill the output back to front)");
  {
    int n1 = 0, m1 = 0;
    CHECK(orbx_extract(left, pl[0], W, H, W, 0, 1000, kps.data(), desc.data(), cap, &n1, &m1));
    const uint8_t* padded;
    int lw, lh, ls;
    CHECK(orbx_pyramid(left, 1, &padded, &lw, &lh, &ls));
  }
  step("extraction with a lapping area inside the image");
  CHECK(orbx_extract_batch(left, NF, pl, W, H, W, W / 3, 2 * W / 3, kps.data(), desc.data(), cap, n, mono));
  step("stereo pair extraction, batched");
  CHECK(orbx_extract_batch(left, NF, pl, W, H, W, 0, 0, kps.data(), desc.data(), cap, n, mono));
  CHECK(orbx_extract_batch(right, NF, pr, W, H, W, 0, 0, kpsR.data(), descR.data(), cap, nR, monoR));
  fprintf(stderr, "race_driver: keypoints %d %d | %d %d\n", n[0], n[1], nR[0], nR[1]);

  step("host batches in several groups: ordinary memory (staging threads) and page-locked memory");
  {
    setenv("ORBX_HOST_GROUP", "2", 1);     // read when the plan is built: 5 frames = 3 groups
    orbx_params pg = {300, 1.2f, 4, 20, 7, 0, 5};
    orbx_extractor* eg = nullptr;
    CHECK(orbx_create(&pg, &eg));
    const int capg = orbx_max_keypoints(eg);
    const uint8_t* five[5] = {pl[0], pl[1 % NF], pl[0], pl[1 % NF], pl[0]};
    std::vector<orbx_keypoint> kg((size_t)5 * capg);
    std::vector<uint8_t> dg((size_t)5 * capg * 32);
    int ng[5], mg[5];
    CHECK(orbx_extract_batch(eg, 5, five, W, H, W, 0, 1000, kg.data(), dg.data(), capg, ng, mg));
    void *hi = nullptr, *hk = nullptr, *hd = nullptr;
    CHECK(orbx_host_alloc((size_t)5 * W * H, &hi));
    CHECK(orbx_host_alloc((size_t)5 * capg * 28, &hk));
    CHECK(orbx_host_alloc((size_t)5 * capg * 32, &hd));
    const uint8_t* pin[5];
    for (int f = 0; f < 5; ++f) {
      memcpy((uint8_t*)hi + (size_t)f * W * H, five[f], (size_t)W * H);
      pin[f] = (const uint8_t*)hi + (size_t)f * W * H;
    }
    CHECK(orbx_extract_batch(eg, 5, pin, W, H, W, 0, 1000, (orbx_keypoint*)hk, (uint8_t*)hd, capg, ng, mg));
    CHECK(orbx_host_free(hi)); CHECK(orbx_host_free(hk)); CHECK(orbx_host_free(hd));
    orbx_destroy(eg);
    unsetenv("ORBX_HOST_GROUP");
  }

  step("lapping-area 2-NN on the two extractions");
  {
    std::vector<uint32_t> li((size_t)cap * 2);
    std::vector<uint16_t> ld((size_t)cap * 2);
    int nq = 0, ml = 0, nt = 0, mr = 0;
    CHECK(orbx_lapping_knn2(left, right, 1, li.data(), ld.data(), cap, &nq, &ml, &nt, &mr));
    CHECK(orbx_lapping_knn2_batch_device(left, right, NF));
    for (int f = 0; f < NF; ++f)
      CHECK(orbx_lapping_knn2_download(left, f, li.data(), ld.data(), cap, &nq, &ml, &nt, &mr));
    // the ratio test on the records where they lie
    CHECK(orbx_lapping_ratio_test_batch_device(left, 0.7));
    std::vector<int32_t> match(cap);
    int ngood = 0;
    for (int f = 0; f < NF; ++f) CHECK(orbx_lapping_matches_download(left, f, match.data(), cap, &nq, &ngood));
  }

  step("stereo association");
  CHECK(orbx_stereo_match_batch_device(left, right, NF, 40.0f, 0.1f));
  std::vector<float> uright(cap), depth(cap);
  std::vector<int32_t> l1(cap);
  int nn = 0;
  for (int f = 0; f < NF; ++f) CHECK(orbx_stereo_download(left, f, uright.data(), depth.data(), l1.data(), cap, &nn));

  step("row-band search on host arrays");
  {
    float scale[8];
    CHECK(orbx_get_tables(left, scale, nullptr, nullptr, nullptr, nullptr));
    std::vector<int32_t> bi(n[0]), bd(n[0]);
    CHECK(orbx_stereo_band_match(0, kps.data(), desc.data(), n[0], kpsR.data(), descR.data(), nR[0], scale,
                                 prm.nlevels, H, 0.0f, 40.0f, bi.data(), bd.data()));
  }

  step("frame grid, with and without distortion");
  orbx_pinhole cam = {0.6f * W, 0.6f * W, 0.5f * W, 0.5f * H, -0.28f, 0.07f, 0.0002f, 0.00002f, 0.0f};
  float bounds[4];
  CHECK(orbx_image_bounds(&cam, W, H, bounds));
  CHECK(orbx_frame_grid_batch_device(left, NF, &cam, bounds));
  std::vector<orbx_keypoint> kun(cap);
  std::vector<int32_t> cstart(64 * 48 + 1), centries(cap);
  CHECK(orbx_frame_grid_download(left, 1, kun.data(), cstart.data(), centries.data(), cap, &nn));
  CHECK(orbx_image_bounds(nullptr, W, H, bounds));
  CHECK(orbx_frame_grid_batch_device(left, NF, nullptr, bounds));
  CHECK(orbx_frame_grid_download(left, 0, kun.data(), cstart.data(), centries.data(), cap, &nn));

  step("depth association from a depth image");
  {
    std::vector<float> dimg((size_t)NF * W * H);
    for (auto& v : dimg) v = (rng() % 5 == 0) ? 0.0f : 0.5f + (float)(rng() % 1000) * 0.01f;
    void* dd;
    hipMalloc(&dd, dimg.size() * 4);
    hipMemcpy(dd, dimg.data(), dimg.size() * 4, 1);
    CHECK(orbx_stereo_from_rgbd_batch_device(left, NF, (const float*)dd, W, 40.0f));
    CHECK(orbx_stereo_download(left, 1, uright.data(), depth.data(), l1.data(), cap, &nn));
    hipFree(dd);
  }

  step("window queries on the grid");
  {
    const int nq = 300, pq = 64;
    std::vector<float> q(5 * nq);
    std::uniform_real_distribution<float> ux(-10.f, W + 10.f), uy(-10.f, H + 10.f), ur(2.f, 40.f);
    for (int i = 0; i < nq; ++i) {
      q[5 * i] = ux(rng);
      q[5 * i + 1] = uy(rng);
      q[5 * i + 2] = ur(rng);
      q[5 * i + 3] = (float)((int)(rng() % 3) - 1);
      q[5 * i + 4] = (i & 1) ? -1.0f : (float)(rng() % 5);
    }
    std::vector<int32_t> idx((size_t)nq * pq), cnt(nq);
    CHECK(orbx_features_in_area(left, 0, q.data(), nq, idx.data(), pq, cnt.data()));
  }

  // "last frame" = frame 1 of the batch seen from frame 0: every keypoint of frame 0 projects
  // near itself, so many of them compete for the same current keypoints
  step("search by projection from the last frame");
  {
    const int nl = n[0];
    std::vector<uint8_t> valid(nl), obs(nl), state(cap);
    std::vector<float> uv(2 * (size_t)nl), invz(nl), ang(nl), ur(cap);
    std::vector<int32_t> oct(nl), assigned(cap);
    std::uniform_real_distribution<float> jitter(-4.f, 4.f);
    for (int i = 0; i < nl; ++i) {
      valid[i] = rng() % 8 != 0;
      obs[i] = rng() % 4 != 0;
      uv[2 * i] = kun[i].x + jitter(rng);
      uv[2 * i + 1] = kun[i].y + jitter(rng);
      invz[i] = 0.05f + 0.001f * (float)(rng() % 100);
      ang[i] = kun[i].angle;
      oct[i] = kun[i].octave;
    }
    for (int i = 0; i < cap; ++i) {
      state[i] = (uint8_t)(rng() % 6 == 0 ? 1 + rng() % 2 : 0);
      ur[i] = (rng() % 3 == 0) ? kun[i % std::max(1, nl)].x - 5.0f : -1.0f;
    }
    const std::vector<uint8_t> dl = perturbed(desc.data(), nl, 24, rng);
    int nout = 0, nm = 0, rounds = 0;
    for (int variant = 0; variant < 3; ++variant) {
      CHECK(orbx_search_by_projection_last(left, 0, nl, valid.data(), uv.data(), invz.data(), oct.data(), ang.data(),
                                           dl.data(), obs.data(), variant == 1 ? ur.data() : nullptr,
                                           variant == 2 ? nullptr : state.data(), 15.0f, variant == 2, 0, 40.0f,
                                           variant != 2, assigned.data(), cap, &nout, &nm, &rounds));
      fprintf(stderr, "race_driver:   variant %d: %d matches in %d rounds\n", variant, nm, rounds);
    }
  }

  step("search by projection from a keyframe");
  {
    const int nk = n[0];
    std::vector<uint8_t> valid(nk, 1), has(cap);
    std::vector<float> uv(2 * (size_t)nk), ang(nk);
    std::vector<int32_t> lvl(nk), assigned(cap);
    std::uniform_real_distribution<float> jitter(-6.f, 6.f);
    for (int i = 0; i < nk; ++i) {
      uv[2 * i] = kun[i].x + jitter(rng);
      uv[2 * i + 1] = kun[i].y + jitter(rng);
      ang[i] = kun[i].angle;
      lvl[i] = kun[i].octave;
    }
    for (int i = 0; i < cap; ++i) has[i] = rng() % 5 == 0;
    const std::vector<uint8_t> dk = perturbed(desc.data(), nk, 16, rng);
    int nout = 0, nm = 0, rounds = 0;
    CHECK(orbx_search_by_projection_keyframe(left, 0, nk, valid.data(), uv.data(), lvl.data(), ang.data(), dk.data(),
                                             has.data(), 10.0f, 100, 1, assigned.data(), cap, &nout, &nm, &rounds));
    fprintf(stderr, "race_driver:   %d matches in %d rounds\n", nm, rounds);
  }

  step("search by projection of map points");
  {
    // three map points per keypoint of frame 0, each with its own jitter
    const int np = 3 * n[0];
    std::vector<uint8_t> view(np), obs(np), state(cap);
    std::vector<float> proj(2 * (size_t)np), xr(np), vcos(np), ur(cap);
    std::vector<int32_t> lvl(np), assigned(cap);
    std::vector<uint8_t> pd;
    std::uniform_real_distribution<float> jitter(-5.f, 5.f);
    for (int c = 0; c < 3; ++c) {
      const std::vector<uint8_t> d = perturbed(desc.data(), n[0], 10 * c, rng);
      pd.insert(pd.end(), d.begin(), d.end());
    }
    for (int i = 0; i < np; ++i) {
      const orbx_keypoint& kp = kun[i % n[0]];
      view[i] = rng() % 10 != 0;
      obs[i] = rng() % 4 != 0;
      proj[2 * i] = kp.x + jitter(rng);
      proj[2 * i + 1] = kp.y + jitter(rng);
      xr[i] = proj[2 * i] - 5.0f;
      vcos[i] = (rng() & 1) ? 0.9995f : 0.9f;
      lvl[i] = std::min(prm.nlevels - 1, kp.octave + (int)(rng() % 2));
    }
    for (int i = 0; i < cap; ++i) {
      state[i] = (uint8_t)(rng() % 6 == 0 ? 1 + rng() % 2 : 0);
      ur[i] = (rng() % 3 == 0) ? kun[i % std::max(1, n[0])].x - 5.0f : -1.0f;
    }
    int nout = 0, nm = 0, rounds = 0;
    for (int variant = 0; variant < 2; ++variant) {
      CHECK(orbx_search_by_projection_points(left, 0, np, view.data(), proj.data(), xr.data(), lvl.data(), vcos.data(),
                                             pd.data(), obs.data(), variant ? ur.data() : nullptr,
                                             variant ? state.data() : nullptr, variant ? 1.0f : 4.0f, 0.8f,
                                             assigned.data(), cap, &nout, &nm, &rounds));
      fprintf(stderr, "race_driver:   variant %d: %d matches in %d rounds\n", variant, nm, rounds);
    }
  }

  // the right camera of a rig: the other handle's extraction of the same batch, its own grid
  step("two-camera forms of the projection matchers");
  {
    CHECK(orbx_frame_grid_batch_device(right, NF, nullptr, bounds));
    const int nl = n[0];
    std::vector<uint8_t> valid(nl), obs(nl), stL(cap), stR(cap);
    std::vector<float> uvl(2 * (size_t)nl), uvr(2 * (size_t)nl), invz(nl), ang(nl), vcos(nl, 0.9995f);
    std::vector<int32_t> oct(nl), asgL(cap), asgR(cap), l2r(cap, -1), r2l(cap, -1);
    std::uniform_real_distribution<float> jitter(-5.f, 5.f);
    for (int i = 0; i < nl; ++i) {
      valid[i] = rng() % 8 != 0;
      obs[i] = rng() % 3 != 0;
      uvl[2 * i] = kun[i].x + jitter(rng);
      uvl[2 * i + 1] = kun[i].y + jitter(rng);
      uvr[2 * i] = kun[i].x + jitter(rng);
      uvr[2 * i + 1] = kun[i].y + jitter(rng);
      invz[i] = 0.05f + 0.001f * (float)(rng() % 100);
      ang[i] = kun[i].angle;
      oct[i] = kun[i].octave;
    }
    for (int i = 0; i < cap; ++i) {
      stL[i] = (uint8_t)(rng() % 6 == 0 ? 1 + rng() % 2 : 0);
      stR[i] = (uint8_t)(rng() % 6 == 0 ? 1 + rng() % 2 : 0);
    }
    const std::vector<uint8_t> dl = perturbed(desc.data(), nl, 24, rng);
    int nL = 0, nR = 0, nm = 0, rounds = 0;
    CHECK(orbx_search_by_projection_last_rig(left, right, 0, nl, valid.data(), uvl.data(), invz.data(), uvr.data(),
                                             oct.data(), ang.data(), dl.data(), obs.data(), stL.data(), stR.data(),
                                             15.0f, 0, 0, 1, asgL.data(), cap, &nL, asgR.data(), cap, &nR, &nm, &rounds));
    fprintf(stderr, "race_driver:   last frame, two cameras: %d matches in %d rounds\n", nm, rounds);
    // stereo partners: arbitrary links, so that partner writes re-open and close keypoints
    for (int i = 0; i < nL; ++i) l2r[i] = (nR > 0 && rng() % 3 == 0) ? (int)(rng() % nR) : -1;
    for (int i = 0; i < nR; ++i) r2l[i] = (nL > 0 && rng() % 3 == 0) ? (int)(rng() % nL) : -1;
    CHECK(orbx_search_by_projection_points_rig(left, right, 0, nl, valid.data(), uvl.data(), oct.data(), vcos.data(),
                                               valid.data(), uvr.data(), oct.data(), vcos.data(), dl.data(), obs.data(),
                                               l2r.data(), r2l.data(), stL.data(), stR.data(), 4.0f, 0.9f, asgL.data(),
                                               cap, &nL, asgR.data(), cap, &nR, &nm, &rounds));
    fprintf(stderr, "race_driver:   map points, two cameras: %d matches in %d rounds\n", nm, rounds);
  }

  step("search for initialization");
  {
    // F1 = the same keypoints with perturbed descriptors: neighbours compete for the same
    // keypoint of F2
    const int n1 = n[0];
    std::vector<orbx_keypoint> k1(kun.begin(), kun.begin() + n1);
    for (auto& kp : k1) kp.octave = 0;   // only level-0 keypoints of F1 take part (matcher.cc:642-645)
    const std::vector<uint8_t> d1 = perturbed(desc.data(), n1, 40, rng);
    std::vector<float> prev(2 * (size_t)n1);
    std::uniform_real_distribution<float> jitter(-6.f, 6.f);
    for (int i = 0; i < n1; ++i) {
      prev[2 * i] = k1[i].x + jitter(rng);
      prev[2 * i + 1] = k1[i].y + jitter(rng);
    }
    std::vector<int32_t> m12(n1);
    int nm = 0, rounds = 0;
    CHECK(orbx_search_for_initialization(left, 0, n1, k1.data(), d1.data(), prev.data(), 30,
                                         0.9f, 1, m12.data(), &nm, &rounds));
    fprintf(stderr, "race_driver:   %d matches in %d rounds\n", nm, rounds);
  }

  step("bag of words");
  {
    const int k = 6, L = 3;
    std::vector<int32_t> parent;
    std::vector<uint8_t> leaf, vd;
    std::vector<double> wts;
    std::vector<int> level = {0};
    // complete k-ary tree in breadth-first order, random descriptors
    for (size_t p = 0; p < level.size(); ++p) {
      if (level[p] == L) continue;
      for (int c = 0; c < k; ++c) {
        parent.push_back((int32_t)p);
        level.push_back(level[p] + 1);
        leaf.push_back(level[p] + 1 == L);
        for (int b = 0; b < 32; ++b) vd.push_back((uint8_t)rng());
        wts.push_back(level[p] + 1 == L ? 0.1 + (double)(rng() % 1000) / 500.0 : 0.0);
      }
    }
    orbx_vocabulary* voc = nullptr;
    CHECK(orbx_vocabulary_create(k, L, 0, 0, (int)parent.size(), parent.data(), leaf.data(), vd.data(), wts.data(),
                                 0, &voc));
    for (int levelsup : {4, 1, 0}) {
      CHECK(orbx_bow_transform_batch_device(left, voc, NF, levelsup));
      std::vector<uint32_t> wid(cap), nid(cap), feats(cap);
      std::vector<double> wv(cap);
      std::vector<int32_t> nstart(cap + 1);
      int nw = 0, nnodes = 0;
      for (int f = 0; f < NF; ++f)
        CHECK(orbx_bow_download(left, f, wid.data(), wv.data(), &nw, nid.data(), nstart.data(), feats.data(),
                                &nnodes, cap));
    }
    // the keyframe = the same frame with perturbed descriptors: its feature vector comes from a
    // transform on the right handle's copy of it (extract + transform + download)
    step("search by bag of words");
    for (int levelsup : {1, 4}) {
      CHECK(orbx_bow_transform_batch_device(left, voc, NF, levelsup));
      CHECK(orbx_bow_transform_batch_device(right, voc, NF, levelsup));
      std::vector<uint32_t> nid(cap), feats(cap);
      std::vector<int32_t> nstart(cap + 1);
      int nw = 0, nnodes = 0;
      CHECK(orbx_bow_download(right, 0, nullptr, nullptr, &nw, nid.data(), nstart.data(), feats.data(), &nnodes, cap));
      std::vector<uint8_t> valid(nR[0]);
      std::vector<float> ang(nR[0]);
      for (int i = 0; i < nR[0]; ++i) { valid[i] = rng() % 7 != 0; ang[i] = kpsR[i].angle; }
      std::vector<int32_t> m(cap);
      int nout = 0, nm = 0;
      CHECK(orbx_search_by_bow(left, 0, nR[0], valid.data(), ang.data(), descR.data(), nnodes, nid.data(),
                               nstart.data(), feats.data(), 0.9f, 1, m.data(), cap, &nout, &nm));
      fprintf(stderr, "race_driver:   levelsup %d: %d nodes, %d matches\n", levelsup, nnodes, nm);
    }
    orbx_vocabulary_destroy(voc);
  }

  step("brute-force 2-NN and the merge of partial results");
  {
    const int nq = 300;
    const int64_t nt = 1500;
    std::vector<uint8_t> q((size_t)nq * 32), t((size_t)nt * 32);
    for (auto& b : q) b = (uint8_t)rng();
    for (auto& b : t) b = (uint8_t)rng();
    std::vector<uint32_t> idx(2 * (size_t)nq);
    std::vector<uint16_t> dist(2 * (size_t)nq);
    CHECK(orbx_knn2(0, q.data(), nq, t.data(), nt, idx.data(), dist.data()));
    CHECK(orbx_knn2(0, q.data(), 3, t.data(), 1, idx.data(), dist.data()));
    // two shards on device pointers + merge
    void *dq, *dt, *di, *dd, *dio, *ddo;
    hipMalloc(&dq, q.size());
    hipMalloc(&dt, t.size());
    hipMalloc(&di, 2 * 2 * (size_t)nq * 4);
    hipMalloc(&dd, 2 * 2 * (size_t)nq * 2);
    hipMalloc(&dio, 2 * (size_t)nq * 4);
    hipMalloc(&ddo, 2 * (size_t)nq * 2);
    hipMemcpy(dq, q.data(), q.size(), 1);
    hipMemcpy(dt, t.data(), t.size(), 1);
    const int64_t half = nt / 2;
    CHECK(orbx_knn2_device((const uint8_t*)dq, nq, (const uint8_t*)dt, half, 0, (uint32_t*)di, (uint16_t*)dd, nullptr));
    CHECK(orbx_knn2_device((const uint8_t*)dq, nq, (const uint8_t*)dt + half * 32, nt - half, (uint32_t)half,
                           (uint32_t*)di + 2 * nq, (uint16_t*)dd + 2 * nq, nullptr));
    CHECK(orbx_top2_merge_device((const uint32_t*)di, (const uint16_t*)dd, 2, nq, (uint32_t*)dio, (uint16_t*)ddo,
                                 nullptr));
    // the allocation-free form (partial results in the library's own device array), twice
    CHECK(orbx_knn2_device_noalloc((const uint8_t*)dq, nq, (const uint8_t*)dt, nt, 0, (uint32_t*)dio, (uint16_t*)ddo,
                                   nullptr));
    CHECK(orbx_knn2_device_noalloc((const uint8_t*)dq, nq / 2, (const uint8_t*)dt, half, 7, (uint32_t*)dio,
                                   (uint16_t*)ddo, nullptr));
    hipFree(dq); hipFree(dt); hipFree(di); hipFree(dd); hipFree(dio); hipFree(ddo);
  }

  step("sort hook");
  {
    std::vector<uint64_t> rec(777);
    for (auto& r : rec) r = ((uint64_t)(rng() % 40) << 32) | (uint32_t)rng();
    CHECK(orbx_debug_sort_records(0, rec.data(), (int)rec.size()));
  }

  orbx_destroy(left);
  orbx_destroy(right);
  step("done");
  return 0;
}
