// C ABI of the extractor (include/orbx.h): handle, plan construction, stage
// orchestration on one HIP stream, host<->device transfers, stage dumps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/orbx.h"
#include "orbx_plan.h"
#include "orbx_bow.h"

namespace orbx {
void launch_copy_words2(const uint32_t* srcA, uint32_t* dstA, int nA, const uint32_t* srcB, uint32_t* dstB, int nB,
                        hipStream_t s);
void launch_export_few(const int* d_n, const int* d_mono, const float* d_kps, const uint8_t* d_desc, int nframes,
                       int kc, int cap, int G, int* hcnt, void* hk, void* hd, hipStream_t s);
void launch_pyramid(const Plan& hp, const uint8_t* d_img, int istride, size_t iframe, uint8_t* d_pyr,
                    int f0, int nframes, hipStream_t s);
void launch_knn2_frames(const uint8_t* descL, int capL, const int* nL, const int* monoL, const uint8_t* descR,
                        int capR, const int* nR, const int* monoR, uint32_t* idx, uint16_t* dist, int nframes,
                        hipStream_t s);
void launch_ratio_test_frames(const uint32_t* idx, const uint16_t* dist, int cap, const int* nq, const int* mono,
                              int nframes, double ratio, int32_t* match, hipStream_t s);
void launch_fast(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, uint32_t* d_cand,
                 int* d_cellCount, int l0, int l1, int f0, int nframes, hipStream_t s);
void launch_octree(const Plan& hp, const Plan* dp, const uint32_t* d_cand, const int* d_cellCount,
                   uint32_t* d_flatA, uint32_t* d_flatB, uint32_t* d_sel, int* d_selCount,
                   uint8_t* d_workspace, int l0, int l1, int f0, int nframes, hipStream_t s);
size_t octree_lds_bytes(const Plan& hp);
size_t fast_lds_bytes(const Plan& hp, int ts);
void launch_assemble(const Plan& hp, const Plan* dp, const uint32_t* d_sel, const int* d_selCount,
                     int lap0, int lap1, float* d_kps, WorkItem* d_work, int* d_n, int* d_mono,
                     int f0, int nframes, hipStream_t s);
void launch_blur(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, uint8_t* d_blur, int f0,
                 int nframes, hipStream_t s);
void launch_describe_fused(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, const WorkItem* d_work,
                           const int* d_n, float* d_kps, uint8_t* d_desc, int f0, int nframes, hipStream_t s);
void launch_describe(const Plan& hp, const Plan* dp, const uint8_t* d_pyr, const uint8_t* d_blur,
                     const WorkItem* d_work, const int* d_n, float* d_kps, uint8_t* d_desc,
                     int f0, int nframes, hipStream_t s);
int octree_set_lds_limit(const Plan& hp);
void launch_stereo_export_one(const int* nL, const float* uRight, const float* depth, int cap, int* hcnt, float* hu,
                              float* hd, hipStream_t s);
void launch_stereo(const Plan& hp, const Plan* dp, const uint8_t* pyrL, const uint8_t* pyrR,
                   const float* kpsL, const uint8_t* descL, const int* nL, int capL,
                   const float* kpsR, const uint8_t* descR, const int* nR, int capR, float mbf,
                   float mb, int* rowStart, int* entries, int* bandIdx, int* bandDist, int* sadDist,
                   float* uRight, float* depth, int nframes, hipStream_t s);
int stereo_entries_per_keypoint(const Plan& hp);
void launch_stereo_from_rgbd(const float* kps, const float* kpsUn, const int* n, int cap, const float* depthImg,
                             size_t frameFloats, int stride, int W, int H, float mbf, float* uRight, float* depth,
                             int* sadDist, int nframes, hipStream_t s);
size_t frame_grid_lds_bytes(int cap);
void launch_frame_grid(const float* d_kps, const int* d_n, int cap, const orbx_pinhole& cam, int distorted,
                       const float* bounds, float* d_kpsUn, int* d_cellStart, int* d_entries,
                       int nframes, hipStream_t s);
void launch_features_in_area(const float* d_kpsUn, int cap, const int* d_cellStart, const int* d_entries,
                             int frame, const float* bounds, const float* d_queries, int nq, int* d_out,
                             int percap, int* d_counts, hipStream_t s);
int launch_debug_sort(uint64_t* d_data, int n, int team, hipStream_t s);
size_t projection_work_ints(int nLast, int n);
size_t init_work_ints(int n1, int n2);
void launch_search_by_projection_last(const uint8_t* d_lastValid, const float* d_uv, const float* d_invz,
                                      const int* d_lastOctave, const float* d_lastAngle, const uint8_t* d_lastDesc,
                                      const uint8_t* d_lastObs, int nLast, const float* d_kpsUn,
                                      const uint8_t* d_desc, const float* d_uRight, const uint8_t* d_state, int n,
                                      const int* d_cellStart, const int* d_entries, const float* bounds,
                                      const float* scaleFactors, int nlevels, float th, int forward, int backward,
                                      float mbf, int checkOrientation, int maxDist, int* d_work, int* d_assigned,
                                      int* d_summary, hipStream_t s);
void launch_search_for_initialization(const float* d_kps1Un, const uint8_t* d_desc1, int n1, float* d_prevMatched,
                                      const float* d_kps2Un, const uint8_t* d_desc2, int n2,
                                      const int* d_cellStart2, const int* d_entries2, const float* bounds2,
                                      float window, float nnratio, int checkOrientation, int* d_work,
                                      int* d_matches12, int* d_summary, hipStream_t s);
void launch_search_by_bow(const uint8_t* d_kfValid, const float* d_kfAngle, const uint8_t* d_kfDesc, int nKfNodes,
                          const uint32_t* d_kfNode, const int* d_kfStart, const uint32_t* d_kfFeat,
                          const float* d_kps, const uint8_t* d_desc, const int* d_n, int cap, const uint32_t* d_fNode,
                          const int* d_fStart, const uint32_t* d_fFeat, const int* d_fCount, float nnratio,
                          int checkOrientation, int* d_matches, int* d_summary, hipStream_t s);
void launch_search_by_projection_last_rig(const uint8_t* d_lastValid, const float* d_uvL, const float* d_invz,
                                          const float* d_uvR, const int* d_lastOctave, const float* d_lastAngle,
                                          const uint8_t* d_lastDesc, const uint8_t* d_lastObs, int nLast,
                                          const float* d_kpsL, const uint8_t* d_descL, const uint8_t* d_stateL, int nL,
                                          const int* d_cellStartL, const int* d_entriesL, const float* d_kpsR,
                                          const uint8_t* d_descR, const uint8_t* d_stateR, int nR,
                                          const int* d_cellStartR, const int* d_entriesR, const float* bounds,
                                          const float* scaleFactors, int nlevels, float th, int forward, int backward,
                                          int checkOrientation, int* d_workL, int* d_workR, int* d_assignedL,
                                          int* d_assignedR, int* d_sumL, int* d_sumR, int* d_summary, hipStream_t s);
size_t points_rig_work_ints(int nPts, int nL, int nR);
void launch_search_by_projection_points_rig(const uint8_t* d_inView, const float* d_proj, const int* d_level,
                                            const float* d_viewCos, const uint8_t* d_inViewR, const float* d_projR,
                                            const int* d_levelR, const float* d_viewCosR, const uint8_t* d_ptDesc,
                                            const uint8_t* d_ptObs, int nPts, const float* d_kpsL,
                                            const uint8_t* d_descL, const uint8_t* d_stateL, int nL,
                                            const int* d_cellStartL, const int* d_entriesL, const float* d_kpsR,
                                            const uint8_t* d_descR, const uint8_t* d_stateR, int nR,
                                            const int* d_cellStartR, const int* d_entriesR, const int* d_leftToRight,
                                            const int* d_rightToLeft, const float* bounds, const float* scaleFactors,
                                            int nlevels, float th, float nnratio, int* d_work, int* d_assignedL,
                                            int* d_assignedR, int* d_summary, hipStream_t s);
void launch_search_by_projection_points(const uint8_t* d_inView, const float* d_proj, const float* d_projXR,
                                        const int* d_level, const float* d_viewCos, const uint8_t* d_ptDesc,
                                        const uint8_t* d_ptObs, int nPts, const float* d_kpsUn, const uint8_t* d_desc,
                                        const float* d_uRight, const uint8_t* d_state, int n, const int* d_cellStart,
                                        const int* d_entries, const float* bounds, const float* scaleFactors,
                                        int nlevels, float th, float nnratio, int* d_work, int* d_assigned,
                                        int* d_summary, hipStream_t s);
}  // namespace orbx

using orbx::LevelDesc;
using orbx::Plan;
using orbx::Tap;
using orbx::ColRec;
using orbx::WorkItem;

namespace {

// cvRound / cvFloor / cvCeil as OpenCV defines them for float and double.
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(float v) { int i = (int)v; return i + (i < v); }
inline int16_t sat_short(float v) {
  int r = cv_round(v);
  return (int16_t)std::min(std::max(r, -32768), 32767);
}
inline int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

#define HIP_TRY(expr)                       \
  do {                                      \
    hipError_t e_ = (expr);                 \
    if (e_ != hipSuccess) {                 \
      last_hip_error = e_;                  \
      return ORBX_ERR_HIP;                  \
    }                                       \
  } while (0)

}  // namespace

namespace {
// A few host threads that copy image rows into pinned staging and results out of it while the
// GPU works on the neighbouring group of frames (one memcpy thread moves ~10 GB/s, a frame is
// 0.36 MB in and 0.06 MB out).  Only the host-pointer batch entry point with pageable caller
// memory uses it; it is created on first use.
class HostPool {
 public:
  explicit HostPool(int nthreads) {
    // a thread that cannot be created (std::system_error) is done without: the pool works with
    // those that started, in the limit with the calling thread alone
    try {
      for (int i = 0; i < nthreads; ++i) workers_.emplace_back([this] { loop(); });
    } catch (...) {
    }
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  // fn(i) for i in [0, n), on the workers and the calling thread; returns when all are done
  void parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (n == 1 || workers_.empty()) {
      for (int i = 0; i < n; ++i) fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; n_ = n; next_.store(0); done_ = 0; ++epoch_;
    }
    cv_.notify_all();
    work();
    // A job is over when every item is done AND no thread is still inside work() with this job's
    // function: a worker that woke late must not claim an index of the next job with this one's
    // (by then destroyed) function.  Threads enter work() only under the lock and only while
    // fn_ is set, so nothing can start between here and the next job.
    std::unique_lock<std::mutex> lk(mu_);
    idle_.wait(lk, [this] { return done_ == n_ && active_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work() {
    int mine = 0;
    const std::function<void(int)>* fn;
    int n;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!fn_) return;
      fn = fn_; n = n_;
      ++active_;
    }
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= n) break;
      (*fn)(i);
      ++mine;
    }
    std::lock_guard<std::mutex> lk(mu_);
    done_ += mine;
    --active_;
    if (done_ == n_ && active_ == 0) idle_.notify_all();
  }
  void loop() {
    unsigned seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
      }
      work();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, idle_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, done_ = 0, active_ = 0;
  std::atomic<int> next_{0};
  unsigned epoch_ = 0;
  bool stop_ = false;
};
}  // namespace

struct orbx_extractor {
  orbx_params prm;
  double scaleFactorD;  // the reference keeps the float argument in a double member
  std::vector<float> scale, invScale, sigma2, invSigma2;
  std::vector<int> quota;
  int umax[16];
  int maxKeypoints = 0;

  int device = 0;
  hipStream_t stream = nullptr;
  // side stream(s): independent stages run beside the latency-bound quadtree
  // lane[0]: blur of the first half beside its quadtree; lane[1]: the second half of a large
  // batch, one stage behind the first; lane[2]: blur of the second half
  static const int kLanes = 3;
  static const int kSplitMin = 128;   // frames per half below which a batch is not split
  hipStream_t lane[kLanes] = {};
  hipEvent_t evFork = nullptr, evJoin[kLanes] = {};
  static const int kMaxParts = 3;
  int maxParts = 2;                   // parts a batch is cut into at most (timing knob ORBX_PARTS)
  int blurPlace = 3;                  // blur: 3 between pyramid and detection on the part's stream (the order
                                      // in which the two parts of a batch pair best: the other part's
                                      // detection then runs beside quadtree, descriptors and pyramid),
                                      // 0 after the quadtree, 2 after the detection, 1 beside the quadtree
                                      // on the side stream (timing knob ORBX_SIDE_BLUR)
  hipEvent_t evStart = nullptr, evForkB = nullptr, evPyr[kMaxParts] = {};
  hipEvent_t evForkP[kMaxParts] = {}, evJoinP[kMaxParts] = {}, evDone[kMaxParts] = {};
  hipError_t last_hip_error = hipSuccess;

  // plan for the current image size
  bool planned = false;
  Plan hp;
  Plan* dp = nullptr;
  std::vector<void*> d_taps;

  uint8_t* d_img = nullptr;   // staging for host-image entry points
  uint8_t* d_pyr = nullptr;
  uint8_t* d_blur = nullptr;
  uint32_t* d_cand = nullptr;
  int* d_cellCount = nullptr;
  uint32_t* d_flatA = nullptr;
  uint32_t* d_flatB = nullptr;
  uint32_t* d_sel = nullptr;
  int* d_selCount = nullptr;
  uint8_t* d_octws = nullptr;  // quadtree node arrays when they do not fit LDS
  float* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  WorkItem* d_work = nullptr;
  int* d_n = nullptr;
  int* d_mono = nullptr;
  size_t pyr_bytes = 0;
  size_t blur_bytes = 0;
  // stereo association results of this (left) handle, allocated by the first stereo call
  int* d_stBandIdx = nullptr;
  int* d_stRowStart = nullptr;   // row table of the right keypoints (CSR)
  int* d_stEntries = nullptr;
  int stRightCap = 0;            // right-handle keypoint capacity d_stEntries is sized for
  int* d_stBandDist = nullptr;
  int* d_stSad = nullptr;
  float* d_stURight = nullptr;
  float* d_stDepth = nullptr;
  int stereoFrames = 0;
  // lapping-area 2-NN results of this (left) handle for a whole batch, allocated by the first call
  uint8_t* d_guided = nullptr;     // inputs, work memory and results of one blocking guided-matcher call (grows)
  size_t guidedBytes = 0;
  uint8_t* d_lapOne = nullptr;     // records of one blocking orbx_lapping_knn2 call: [kp_cap] x (8 + 4) bytes
  std::vector<int> hostN, hostMono;   // counts the last host-pointer extraction read back, per frame
  int hostCountFrames = 0;            // frames for which they are valid (0 after a device-resident call)
  uint32_t* d_lapIdx = nullptr;    // [max_batch][kp_cap][2]
  uint16_t* d_lapDist = nullptr;
  int lapFrames = 0;
  int* d_lapCounts = nullptr;      // [4][max_batch]: Nleft, monoLeft, Nright, monoRight as the batch call saw them
  int32_t* d_lapMatch = nullptr;   // [max_batch][kp_cap]: the ratio test's verdict per query (train index or -1)
  bool lapMatchValid = false;
  // frame grid of the last grid call (undistorted keypoints + 64x48 cell CSR)
  float* d_kpsUn = nullptr;
  int* d_cellStart = nullptr;
  int* d_cellEntries = nullptr;
  int gridFrames = 0;
  float gridBounds[4] = {0, 0, 0, 0};
  hipEvent_t evStereo = nullptr;
  // bag-of-words results of the last transform call, allocated by the first one
  orbx::BowBuffers bow = {};
  void* d_bowBlock = nullptr;
  int bowFrames = 0;

  // Host-pointer entry points: a batch runs as groups of groupFrames frames, each group's
  // transfer in, kernels and transfer out on streams of their own, so that the transfer in of
  // group k+1 and the transfer out of group k-1 overlap the kernels of group k.  Pinned staging
  // (two buffers each way, allocated on first use) only serves pageable caller memory.
  int groupFrames = 0;
  hipStream_t sIn = nullptr, sOut = nullptr;
  uint8_t* h_in[2] = {nullptr, nullptr};
  uint8_t* h_out[2] = {nullptr, nullptr};
  int* h_cnt[2] = {nullptr, nullptr};      // n and mono of a group (pinned), 2 * groupFrames ints
  hipEvent_t evIn[2] = {nullptr, nullptr}, evComp[2] = {nullptr, nullptr}, evOut[2] = {nullptr, nullptr};
  HostPool* pool = nullptr;
  uint8_t* h_pyr = nullptr;         // pinned host mirror of frame 0's pyramid, all levels back to back
  size_t h_pyr_bytes = 0;
  bool h_pyr_valid = false;

  // A batch that ran as two parts leaves its second part on lane[1].  The main stream is made to
  // wait for it only when something other than the next equally split batch follows (a download,
  // a derived stage, a synchronisation): back-to-back batches then run as two independent
  // pipelines, each part following only its own predecessor, instead of every batch waiting for
  // the previous one's slower half.
  bool pyrWait = true;         // timing knob ORBX_PYR_WAIT: a part's pyramid waits for the previous part's
  bool pipelined = false;      // orbx_set_pipelined: leave the join of a batch's later parts to the next call
  bool pendingJoin = false;
  int pendingFrames = 0;       // frame count of the batch whose second part is still un-joined
  int pendingParts = 0;        // and the number of its parts
  int joinParts();
  int lastFrames = 0;
  bool profiling = false;
  hipEvent_t ev[ORBX_NUM_STAGES + 1] = {};
  // normal schedule: events around the detection kernel of every part of the last call
  // detection-launch events of the last kDetRing batched calls (a ring: slot = call number
  // modulo kDetRing), so that a run of back-to-back calls can be read out after it has ended
  static const int kDetRing = 32;
  hipEvent_t evDet[kDetRing][kMaxParts][2] = {};
  int detFrames[kDetRing][kMaxParts] = {};
  int detParts[kDetRing] = {};
  long detCalls = 0;                  // batched calls so far; the current call's slot is (detCalls - 1) % kDetRing
  int lastParts = 0;
  bool evValid = false;

  void buildTables();
  int ensurePlan(int W, int H);
  int buildPlan(int W, int H);
  void freePlan();
  int run(const uint8_t* d_images, int istride, size_t iframe, int nframes, int lap0, int lap1);
  int enqueuePart(int pk, hipStream_t m, hipStream_t side, hipEvent_t fork, hipEvent_t join, int f0, int n,
                  hipEvent_t pyrDone, const uint8_t* d_images, int istride, size_t iframe, int lap0, int lap1);
  void afterExtraction(int nframes);
  int ensureHostStaging(bool pageableIn, bool pageableOut);
};

void orbx_extractor::buildTables() {
  // core/lib/ORB/extractor.cc:406-463
  const int nl = prm.nlevels;
  scaleFactorD = (double)prm.scale_factor;
  scale.resize(nl); invScale.resize(nl); sigma2.resize(nl); invSigma2.resize(nl); quota.resize(nl);
  scale[0] = 1.0f; sigma2[0] = 1.0f;
  for (int i = 1; i < nl; ++i) {
    scale[i] = (float)(scale[i - 1] * scaleFactorD);
    sigma2[i] = scale[i] * scale[i];
  }
  for (int i = 0; i < nl; ++i) {
    invScale[i] = 1.0f / scale[i];
    invSigma2[i] = 1.0f / sigma2[i];
  }
  float factor = (float)(1.0f / scaleFactorD);
  float desired = prm.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; ++l) {
    quota[l] = cv_round(desired);
    sum += quota[l];
    desired *= factor;
  }
  quota[nl - 1] = std::max(prm.nfeatures - sum, 0);

  const int half = 15;
  int vmax = cv_floor(half * std::sqrt(2.f) / 2 + 1);
  int vmin = cv_ceil(half * std::sqrt(2.f) / 2);
  const double hp2 = half * half;
  for (int v = 0; v <= half; ++v) umax[v] = 0;
  for (int v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
  for (int v = half, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

void orbx_extractor::freePlan() {
  if (!planned) return;
  (void)hipSetDevice(device);
  (void)hipStreamSynchronize(stream);
  for (int k = 0; k < kLanes; ++k) if (lane[k]) (void)hipStreamSynchronize(lane[k]);   // a second part may still run
  if (sIn) (void)hipStreamSynchronize(sIn);
  if (sOut) (void)hipStreamSynchronize(sOut);
  pendingJoin = false;
  for (void* t : d_taps) (void)hipFree(t);
  d_taps.clear();
  if (d_bowBlock) (void)hipFree(d_bowBlock);
  d_bowBlock = nullptr; bowFrames = 0; bow = orbx::BowBuffers{};
  void* gbufs[] = {d_kpsUn, d_cellStart, d_cellEntries};
  for (void* b : gbufs) if (b) (void)hipFree(b);
  d_kpsUn = nullptr; d_cellStart = d_cellEntries = nullptr; gridFrames = 0;
  void* sbufs[] = {d_stRowStart, d_stEntries, d_stBandIdx, d_stBandDist, d_stSad, d_stURight, d_stDepth};
  for (void* b : sbufs) if (b) (void)hipFree(b);
  d_stRowStart = d_stEntries = nullptr; stRightCap = 0;
  d_stBandIdx = d_stBandDist = d_stSad = nullptr; d_stURight = d_stDepth = nullptr; stereoFrames = 0;
  if (d_lapOne) (void)hipFree(d_lapOne);
  d_lapOne = nullptr;
  if (d_guided) (void)hipFree(d_guided);
  d_guided = nullptr; guidedBytes = 0;
  if (d_lapIdx) (void)hipFree(d_lapIdx);
  if (d_lapDist) (void)hipFree(d_lapDist);
  if (d_lapCounts) (void)hipFree(d_lapCounts);
  if (d_lapMatch) (void)hipFree(d_lapMatch);
  d_lapIdx = nullptr; d_lapDist = nullptr; d_lapCounts = nullptr; d_lapMatch = nullptr; lapFrames = 0; lapMatchValid = false;
  void* bufs[] = {d_octws, dp, d_img, d_pyr, d_blur, d_cand, d_cellCount, d_flatA, d_flatB, d_sel, d_selCount,
                  d_kps, d_desc, d_work, d_n, d_mono};
  for (void* b : bufs) if (b) (void)hipFree(b);
  dp = nullptr; d_img = d_pyr = d_blur = d_desc = nullptr; d_cand = d_flatA = d_flatB = d_sel = nullptr;
  d_cellCount = d_selCount = d_n = d_mono = nullptr; d_octws = nullptr; d_kps = nullptr; d_work = nullptr;
  for (int i = 0; i < 2; ++i) {
    if (h_in[i]) (void)hipHostFree(h_in[i]);
    if (h_out[i]) (void)hipHostFree(h_out[i]);
    if (h_cnt[i]) (void)hipHostFree(h_cnt[i]);
    h_in[i] = h_out[i] = nullptr; h_cnt[i] = nullptr;
  }
  if (h_pyr) (void)hipHostFree(h_pyr);
  h_pyr = nullptr; h_pyr_bytes = 0; h_pyr_valid = false;
  planned = false;
}

// cv::resize INTER_LINEAR coefficient tables for one axis, indexed by the
// padded destination coordinate.  clampSingleTap selects the horizontal rule
// (out-of-range source -> single tap with weight 2048) versus the vertical
// rule (source rows clamped, weights kept).
static void build_taps(int sn, int dn, bool horizontal, std::vector<Tap>& out) {
  const double scale = 1.0 / ((double)dn / sn);
  std::vector<Tap> t(dn);
  for (int d = 0; d < dn; ++d) {
    float fx = (float)((d + 0.5) * scale - 0.5);
    int s = cv_floor(fx);
    fx -= s;
    Tap e;
    if (horizontal) {
      if (s < 0) { fx = 0; s = 0; }
      if (s >= sn - 1) { fx = 0; s = sn - 1; }
      e.s0 = (uint16_t)s;
      e.s1 = (uint16_t)std::min(s + 1, sn - 1);
    } else {
      e.s0 = (uint16_t)std::min(std::max(s, 0), sn - 1);
      e.s1 = (uint16_t)std::min(std::max(s + 1, 0), sn - 1);
    }
    e.w0 = sat_short((1.f - fx) * 2048);
    e.w1 = sat_short(fx * 2048);
    t[d] = e;
  }
  out.resize(dn + 2 * ORBX_EDGE);
  for (int p = 0; p < dn + 2 * ORBX_EDGE; ++p) out[p] = t[reflect101(p - ORBX_EDGE, dn)];
}

// buildPlan with the C ABI's promise kept: a failed host allocation becomes a status
int orbx_extractor::ensurePlan(int W, int H) {
  try {
    return buildPlan(W, H);
  } catch (...) {
    return ORBX_ERR_UNSUPPORTED;
  }
}

int orbx_extractor::buildPlan(int W, int H) {
  if (planned && hp.W == W && hp.H == H) return ORBX_OK;
  freePlan();
  const int nl = prm.nlevels;
  const int B = prm.max_batch;
  std::memset(&hp, 0, sizeof(hp));
  hp.nlevels = nl; hp.W = W; hp.H = H;
  hp.ini_th = std::min(std::max(prm.ini_th_fast, 0), 255);
  hp.min_th = std::min(std::max(prm.min_th_fast, 0), 255);
  hp.max_batch = B;
  for (int i = 0; i < 16; ++i) hp.umax[i] = umax[i];
  for (int i = 0; i < 16; ++i) hp.umax_packed |= (unsigned long long)(umax[i] & 15) << (4 * i);
  for (int j = 0; j < 32; ++j)
    for (int w = 0; w < 8; ++w) {
      uint32_t m = 0;
      for (int k = 0; k < 4; ++k) {
        const int u = 4 * w + k - 15, av = j < 31 ? std::abs(j - 15) : 16;
        if (av < 16 && std::abs(u) <= umax[av]) m |= 0xFFu << (8 * k);
      }
      hp.ic_mask[j][w] = m;
    }

  size_t pyrOff = 0, blurOff = 0;
  uint32_t candOff = 0, flatOff = 0;
  int cellFirst = 0, tileFirst = 0, selOff = 0, nodesCap = 0, cellsMax = 0, cwMax = 0, chMax = 0;
  for (int l = 0; l < nl; ++l) {
    LevelDesc& L = hp.lv[l];
    L.w = cv_round((float)W * invScale[l]);  // extractor.cc:1105-1107
    L.h = cv_round((float)H * invScale[l]);
    if (L.w >= 4096 + 32 || L.h >= 4096 + 32) return ORBX_ERR_UNSUPPORTED;  // 12-bit key fields
    // detection grid, extractor.cc:754-768
    const int minB = ORBX_DET_ORG;
    const int maxBX = L.w - ORBX_EDGE + 3, maxBY = L.h - ORBX_EDGE + 3;
    const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
    if (width < 35.f || height < 35.f) return ORBX_ERR_TOO_SMALL;
    const int nCols = (int)(width / 35.f), nRows = (int)(height / 35.f);
    L.wCell = (int)std::ceil(width / nCols);
    L.hCell = (int)std::ceil(height / nRows);
    if (L.wCell > ORBX_CELL_MAX - 2 || L.hCell > ORBX_CELL_MAX - 2) return ORBX_ERR_UNSUPPORTED;
    // cells whose detection domain [19 + j*wCell, ...) x [19 + i*hCell, ...) is
    // non-empty inside [19, w-19) x [19, h-19); the others find nothing in the
    // reference either (skipped at :774/:782 or a FAST ROI under 7 pixels)
    L.ncx = 0;
    while (L.ncx < nCols && ORBX_EDGE + L.ncx * L.wCell < L.w - ORBX_EDGE) ++L.ncx;
    L.ncy = 0;
    while (L.ncy < nRows && ORBX_EDGE + L.ncy * L.hCell < L.h - ORBX_EDGE) ++L.ncy;
    L.cell_first = cellFirst;
    cellFirst += L.ncx * L.ncy;
    cellsMax = std::max(cellsMax, L.ncx * L.ncy);
    cwMax = std::max(cwMax, L.wCell);
    chMax = std::max(chMax, L.hCell);
    L.cand_cap_cell = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);
    L.cand_off = candOff;
    const uint64_t levelCand = (uint64_t)L.ncx * L.ncy * L.cand_cap_cell;
    if (candOff + levelCand > 0x7fffffffu) return ORBX_ERR_UNSUPPORTED;
    candOff += (uint32_t)levelCand;
    L.flat_off = flatOff;
    L.flat_cap = (uint32_t)levelCand;
    flatOff += (uint32_t)levelCand;
    // quadtree roots, extractor.cc:549-551
    L.quota = quota[l];
    L.n_roots = (int)std::round(static_cast<float>(maxBX - minB) / (maxBY - minB));
    if (L.n_roots < 1) return ORBX_ERR_TOO_SMALL;
    if (L.n_roots > ORBX_MAX_ROOTS) return ORBX_ERR_UNSUPPORTED;
    L.hX = static_cast<float>(maxBX - minB) / L.n_roots;
    L.sel_cap = L.quota + 8 + 4 * L.n_roots;
    L.sel_off = selOff;
    selOff += L.sel_cap;
    nodesCap = std::max(nodesCap, L.sel_cap);
    L.scale = scale[l];
    L.patch_size = (int)(31 * scale[l]);  // extractor.cc:840
    // image planes
    L.pstride = round_up(L.w + 2 * ORBX_EDGE, 64);
    L.prows = L.h + 2 * ORBX_EDGE;
    L.pyr_plane = (size_t)L.pstride * L.prows;
    L.pyr_off = pyrOff;
    pyrOff += L.pyr_plane * B;
    L.bstride = round_up(L.w, 64);
    L.blur_plane = (size_t)L.bstride * L.h;
    L.blur_off = blurOff;
    blurOff += L.blur_plane * B;
    L.tiles_x = (L.w + 255) / 256;   // k_blur: a block covers 256 columns x 256 rows
    L.tiles_y = (L.h + 255) / 256;
    L.tile_first = tileFirst;
    tileFirst += L.tiles_x * L.tiles_y;
  }
  hp.cells_per_frame = cellFirst;
  hp.tiles_per_frame = tileFirst;
  hp.cand_frame = candOff;
  hp.flat_frame = flatOff;
  hp.sel_frame = selOff;
  hp.kp_cap = selOff;
  hp.fast_ts = cwMax <= 40 ? 48 : (cwMax <= 56 ? 64 : 80);
  hp.fast_tile_rows = chMax + 7;
  hp.fast_mm_rows = chMax + 2;
  hp.fast_qcap = std::min(512, (cwMax * chMax + 7) / 8 * 8);  // survivors queued per cell; more -> per-pixel path
  hp.fast_words = (cwMax * chMax + 31) / 32;
  {
    // LDS is handed out in granules of 1280 bytes (1/128 of a CU's 160 KB): what is left of the
    // kernel's last granule goes to the survivor queue (a cell with more survivors than the queue
    // holds is scored pixel by pixel, which costs several times as much)
    const long gran = 1280;
    const long bytes = (long)orbx::fast_lds_bytes(hp, hp.fast_ts);
    const long room = (bytes + gran - 1) / gran * gran - bytes;
    hp.fast_qcap = std::min({hp.fast_qcap + (int)(room / 2) / 8 * 8, 1024, (cwMax * chMax + 7) / 8 * 8});
  }
  if (const char* e = std::getenv("ORBX_FAST_QCAP")) {   // timing / test knob, results do not change
    const int v = std::atoi(e);
    if (v >= 8 && v <= 1024) hp.fast_qcap = std::min(v / 8 * 8, (cwMax * chMax + 7) / 8 * 8);
  }
  {
    const char* e = std::getenv("ORBX_FAST_SINGLE_PASS");   // timing knob, results do not change
    hp.fast_two_pass = (e && e[0] == '1') ? 0 : 1;
    hp.blur_waves = 1;
    if (const char* bw = std::getenv("ORBX_BLUR_WAVES")) hp.blur_waves = std::atoi(bw) == 4 ? 4 : 1;
    // ORBX_BLUR_PLANES=1 (timing / test knob, same results): the earlier schedule with a full-plane
    // blur launch and a descriptor kernel that reads blurred planes
    const char* bp = std::getenv("ORBX_BLUR_PLANES");
    hp.blur_in_describe = (bp && bp[0] == '1') ? 0 : 1;
    hp.pyr_xcd = 1;
    if (const char* px = std::getenv("ORBX_PYR_XCD")) hp.pyr_xcd = px[0] != '0';
    hp.pyr_waves = 1;
    if (const char* pw = std::getenv("ORBX_PYR_WAVES")) {
      const int v = std::atoi(pw);
      if (v == 1 || v == 2 || v == 4) hp.pyr_waves = v;
    }
  }
  hp.oct_nodes_cap = nodesCap;
  hp.oct_cells_max = cellsMax;
  {
    // LDS key buffers of k_octree: what fits beside the node arrays in ~48 KB per wave
    const long fixed = (long)nodesCap * 86 + (long)(cellsMax + 1) * 4 + 64;
    const long keys = (32 * 1024 - fixed) / 8;
    hp.oct_keys_lds = keys < 256 ? 0 : (int)std::min(keys, 8192L);
    // timing knob: fewer LDS keys per wave = more waves per CU, but more levels partition their
    // keys in the global buffers instead (results do not change)
    if (const char* e = std::getenv("ORBX_OCT_KEYS_LDS")) {
      const long v = std::strtol(e, nullptr, 10);
      if (v >= 0 && v <= 8192) hp.oct_keys_lds = (int)v;
    }
    hp.oct_ws_bytes = 0;
    if (orbx::octree_lds_bytes(hp) > 96 * 1024) {
      hp.oct_keys_lds = 0;
      hp.oct_ws_bytes = (uint32_t)((orbx::octree_lds_bytes(hp) + 255) / 256 * 256);
    }
  }
  {
    // k_octree_team: batches too small to fill the chip with one wave per (frame, level)
    // (measured at 752 x 480, 1000 features: 0.117 against 0.156 ms for one frame, 0.296 against
    // 0.316 ms for 64, 0.515 against 0.471 ms for 128).  Its key arrays take what is left of
    // 120 KB of LDS beside the node arrays.
    hp.oct_team_keys = 0;
    hp.oct_team_frames = 64;
    if (const char* e = std::getenv("ORBX_OCT_TEAM")) {   // timing / test knob, results do not change
      const long v = std::strtol(e, nullptr, 10);
      if (v >= 0 && v <= 1 << 20) hp.oct_team_frames = (int)v;
    }
    if (!hp.oct_ws_bytes && hp.oct_team_frames > 0) {
      const long fixed = (long)nodesCap * 80 + (long)(cellsMax + 1) * 4 + 256;
      const long keys = std::min((120L * 1024 - fixed) / 20, 6144L) / 8 * 8;
      if (keys >= 512) hp.oct_team_keys = (int)keys;
    }
  }
  maxKeypoints = selOff;
  pyr_bytes = pyrOff;

  if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (orbx::octree_set_lds_limit(hp) != 0) return ORBX_ERR_UNSUPPORTED;

  planned = true;  // from here on freePlan() releases whatever was allocated
  // resize tap tables + tiling records
  size_t resizeLds = 0;
  for (int l = 1; l < nl; ++l) {
    std::vector<Tap> tx, ty;
    build_taps(hp.lv[l - 1].w, hp.lv[l].w, true, tx);
    build_taps(hp.lv[l - 1].h, hp.lv[l].h, false, ty);
    LevelDesc& L = hp.lv[l];
    const int pw = L.w + 2 * ORBX_EDGE, ph = L.prows;
    const int ncol = L.pstride / 4;
    std::vector<ColRec> cr(ncol);
    std::vector<uint16_t> bc;
    for (int c = 0; c < ncol; ++c) {
      ColRec r;
      std::memset(&r, 0, sizeof(r));
      // pixels of the row's last dword that lie beyond the padded width repeat the last real
      // pixel's taps and are masked to zero when stored
      const int nvalid = std::min(4, pw - 4 * c);
      bool ok = nvalid >= 1;
      if (ok) {
        auto tapOf = [&](int k) -> const Tap& { return tx[std::min(4 * c + k, pw - 1)]; };
        // lowest left tap of the four pixels: in the reflected left/right border the pixels
        // run backwards through the source, which the per-pixel deltas express just as well
        int sxa = tapOf(0).s0;
        for (int k = 1; k < 4; ++k) sxa = std::min(sxa, (int)tapOf(k).s0);
        r.sxa = (uint16_t)sxa;
        r.nvalid = (uint16_t)nvalid;
        for (int k = 0; k < 4 && ok; ++k) {
          const Tap& t = tapOf(k);
          const int d = (int)t.s0 - sxa;
          ok = d >= 0 && d <= 6 && (t.s1 == t.s0 + 1 || (t.s1 == t.s0 && t.w1 == 0)) &&
               t.w0 + t.w1 == 2048 && t.w1 >= 0;
          r.deltas |= (uint16_t)(d << (3 * k));
          r.w1[k] = (uint16_t)t.w1;
        }
      }
      if (ok) r.deltas |= 0x8000; else r.deltas = 0;
      cr[c] = r;
      if (!ok && 4 * c < pw) bc.push_back((uint16_t)c);
    }
    const int nbx = (L.pstride + 255) / 256, nby = (ph + 15) / 16;
    std::vector<uint16_t> xb(2 * nbx), yb(2 * nby);
    int spanx = 0, spany = 0;
    for (int b = 0; b < nbx; ++b) {
      int lo = 1 << 30, hi = -1;
      for (int p = 256 * b; p < std::min(256 * b + 256, pw); ++p) {
        lo = std::min(lo, (int)tx[p].s0);
        hi = std::max(hi, (int)tx[p].s1);
      }
      if (hi < 0) { lo = 0; hi = 0; }
      xb[2 * b] = (uint16_t)lo; xb[2 * b + 1] = (uint16_t)hi;
      spanx = std::max(spanx, hi - lo + 1);
    }
    for (int b = 0; b < nby; ++b) {
      int lo = 1 << 30, hi = -1;
      for (int p = 16 * b; p < std::min(16 * b + 16, ph); ++p) {
        lo = std::min(lo, (int)ty[p].s0);
        hi = std::max(hi, (int)ty[p].s1);
      }
      yb[2 * b] = (uint16_t)lo; yb[2 * b + 1] = (uint16_t)hi;
      spany = std::max(spany, hi - lo + 1);
    }
    L.rs_ls = round_up(spanx + 3 + 12, 4);   // phase (<=3) + span + 3-dword window overrun
    L.rs_rows = spany;
    resizeLds = std::max(resizeLds, (size_t)L.rs_ls * L.rs_rows);
    void *dx = nullptr, *dy = nullptr, *dc = nullptr, *dxb = nullptr, *dyb = nullptr, *dbc = nullptr;
    HIP_TRY(hipMalloc(&dbc, bc.size() * 2 + 2)); d_taps.push_back(dbc);
    HIP_TRY(hipMemcpy(dbc, bc.data(), bc.size() * 2, hipMemcpyHostToDevice));
    L.bcols = (const uint16_t*)dbc; L.n_bcols = (int)bc.size();
    HIP_TRY(hipMalloc(&dx, tx.size() * sizeof(Tap))); d_taps.push_back(dx);
    HIP_TRY(hipMalloc(&dy, ty.size() * sizeof(Tap))); d_taps.push_back(dy);
    HIP_TRY(hipMalloc(&dc, cr.size() * sizeof(ColRec))); d_taps.push_back(dc);
    HIP_TRY(hipMalloc(&dxb, xb.size() * 2)); d_taps.push_back(dxb);
    HIP_TRY(hipMalloc(&dyb, yb.size() * 2)); d_taps.push_back(dyb);
    HIP_TRY(hipMemcpy(dx, tx.data(), tx.size() * sizeof(Tap), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy, ty.data(), ty.size() * sizeof(Tap), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dc, cr.data(), cr.size() * sizeof(ColRec), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dxb, xb.data(), xb.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dyb, yb.data(), yb.size() * 2, hipMemcpyHostToDevice));
    L.xtab = (const Tap*)dx; L.ytab = (const Tap*)dy; L.colrec = (const ColRec*)dc;
    L.xblk = (const uint16_t*)dxb; L.yblk = (const uint16_t*)dyb;
  }
  if (resizeLds + 64 > 60 * 1024) return ORBX_ERR_UNSUPPORTED;
  {
    // the detection kernel's cell table
    std::vector<orbx::FastCell> cells((size_t)hp.cells_per_frame);
    for (int l = 0; l < nl; ++l) {
      const LevelDesc& L = hp.lv[l];
      for (int cell = 0; cell < L.ncx * L.ncy; ++cell) {
        const int ci = cell / L.ncx, cj = cell - ci * L.ncx;
        const int X0 = ORBX_EDGE + cj * L.wCell, Y0 = ORBX_EDGE + ci * L.hCell;
        const int cw = std::min(L.wCell, L.w - ORBX_EDGE - X0), chh = std::min(L.hCell, L.h - ORBX_EDGE - Y0);
        const int shift = (X0 + 15) & 3;   // planes and rows are 64-byte aligned
        const int ngx = (cw + 3) >> 2;
        orbx::FastCell c;
        c.xy = (uint32_t)X0 | ((uint32_t)Y0 << 16);
        c.geo = (uint32_t)cw | ((uint32_t)chh << 8) | ((uint32_t)l << 16) | ((uint32_t)shift << 20) |
                ((uint32_t)(L.pstride / 64) << 24);
        c.cand = L.cand_off + (uint32_t)cell * (uint32_t)L.cand_cap_cell;
        c.tile = (uint32_t)((size_t)(Y0 - 3 + ORBX_EDGE) * L.pstride + (size_t)(X0 - 4 + ORBX_EDGE - shift));
        c.pyr64 = (uint32_t)(L.pyr_off / 64);
        c.plane64 = (uint32_t)(L.pyr_plane / 64);
        // floor(a / ngx) = (a * magic) >> 16 for 0 <= a <= 64 (ngx <= 64)
        c.magic = 65536u / (uint32_t)ngx + 1u;
        c.pad = 0;
        if (L.pstride / 64 > 255 || L.pyr_off / 64 > 0xffffffffull || cw > 255 || chh > 255) return ORBX_ERR_UNSUPPORTED;
        cells[(size_t)L.cell_first + cell] = c;
      }
    }
    void* dcells = nullptr;
    HIP_TRY(hipMalloc(&dcells, cells.size() * sizeof(orbx::FastCell) + 32)); d_taps.push_back(dcells);
    HIP_TRY(hipMemcpy(dcells, cells.data(), cells.size() * sizeof(orbx::FastCell), hipMemcpyHostToDevice));
    hp.fast_cells = (const orbx::FastCell*)dcells;
  }
  HIP_TRY(hipMalloc((void**)&dp, sizeof(Plan)));
  HIP_TRY(hipMemcpy(dp, &hp, sizeof(Plan), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&d_img, (size_t)B * W * H));
  HIP_TRY(hipMalloc((void**)&d_pyr, pyrOff + 256));  // tile loads may overrun the last row by a few bytes
  blur_bytes = blurOff + 256;   // window loads may overrun a row end by a few bytes
  // blurred planes exist only for the schedule that reads them (else on demand, for the stage dump)
  if (!hp.blur_in_describe) HIP_TRY(hipMalloc((void**)&d_blur, blur_bytes));
  HIP_TRY(hipMalloc((void**)&d_cand, (size_t)B * hp.cand_frame * 4));
  HIP_TRY(hipMalloc((void**)&d_cellCount, (size_t)B * hp.cells_per_frame * 4));
  HIP_TRY(hipMalloc((void**)&d_flatA, (size_t)B * hp.flat_frame * 4));
  HIP_TRY(hipMalloc((void**)&d_flatB, (size_t)B * hp.flat_frame * 4));
  HIP_TRY(hipMalloc((void**)&d_sel, (size_t)B * hp.sel_frame * 4));
  HIP_TRY(hipMalloc((void**)&d_selCount, (size_t)B * nl * 4));
  if (hp.oct_ws_bytes) HIP_TRY(hipMalloc((void**)&d_octws, (size_t)B * nl * hp.oct_ws_bytes));
  HIP_TRY(hipMalloc((void**)&d_kps, (size_t)B * hp.kp_cap * 28));
  HIP_TRY(hipMalloc((void**)&d_desc, (size_t)B * hp.kp_cap * 32));
  HIP_TRY(hipMalloc((void**)&d_work, (size_t)B * hp.kp_cap * sizeof(WorkItem)));
  HIP_TRY(hipMalloc((void**)&d_n, (size_t)B * 4));
  HIP_TRY(hipMalloc((void**)&d_mono, (size_t)B * 4));
  // frames per group of the host-pointer path: large enough for full-rate kernels, small enough
  // that two groups of pinned staging stay modest (64 MB of images per buffer at most)
  groupFrames = (int)std::max<size_t>(1, std::min<size_t>({(size_t)B, (size_t)128, ((size_t)64 << 20) / ((size_t)W * H)}));
  if (const char* e = std::getenv("ORBX_HOST_GROUP")) {   // timing / test knob, results do not change
    const long g = std::strtol(e, nullptr, 10);
    if (g >= 1 && g <= B) groupFrames = (int)g;
  }
  return ORBX_OK;
}

int orbx_extractor::ensureHostStaging(bool pageableIn, bool pageableOut) {
  const int G = groupFrames;
  for (int i = 0; i < 2; ++i) {
    if (!h_cnt[i]) HIP_TRY(hipHostMalloc((void**)&h_cnt[i], (size_t)G * 8, hipHostMallocDefault));
    if (pageableIn && !h_in[i])
      HIP_TRY(hipHostMalloc((void**)&h_in[i], (size_t)G * hp.W * hp.H, hipHostMallocDefault));
    if (pageableOut && !h_out[i])
      HIP_TRY(hipHostMalloc((void**)&h_out[i], (size_t)G * hp.kp_cap * 60, hipHostMallocDefault));
  }
  if ((pageableIn || pageableOut) && !pool && G > 1) {
    // packing threads: up to 12 (measured: 4 -> 79 k, 6 -> 93 k, 12 -> 98 k frames/s), within the
    // job's CPU share where the environment states one and within the machine's CPUs
    int nt = 12;
    if (const char* e = std::getenv("OMP_NUM_THREADS")) { const int v = std::atoi(e); if (v >= 1) nt = std::min(nt, v); }
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw >= 1) nt = std::min(nt, (int)hw);
    if (const char* e = std::getenv("ORBX_HOST_THREADS")) nt = std::min(std::max(std::atoi(e), 1), 64);
    // (thread creation can throw; nothing may cross the C ABI: without a pool the calling thread
    // packs alone)
    try {
      pool = new (std::nothrow) HostPool(nt - 1);   // the calling thread works too
    } catch (...) {
      pool = nullptr;
    }
  }
  return ORBX_OK;
}

int orbx_extractor::joinParts() {
  if (!pendingJoin) return ORBX_OK;
  pendingJoin = false;
  for (int k = 1; k < pendingParts; ++k) HIP_TRY(hipStreamWaitEvent(stream, evDone[k], 0));
  return ORBX_OK;
}

// The pyramid kernels address a frame and a plane with 32-bit byte offsets from a per-frame base.
static bool offsets_fit_32(const orbx::Plan& hp, int istride) {
  return (unsigned long long)istride * (unsigned long long)hp.H < (1ull << 32) &&
         (unsigned long long)hp.lv[0].pstride * (unsigned long long)hp.lv[0].prows < (1ull << 32);
}

int orbx_extractor::run(const uint8_t* d_images, int istride, size_t iframe, int nframes, int lap0,
                        int lap1) {
  if (!offsets_fit_32(hp, istride)) return ORBX_ERR_UNSUPPORTED;
  hipStream_t s = stream;
  const bool prof = profiling;
  lastParts = 0;
  ++detCalls;
  detParts[(detCalls - 1) % kDetRing] = 0;
  // the parts of this batch cover the same frames as those of the previous one only when the
  // batch is split the same way; anything else first waits for the previous second part
  if (pendingJoin && (prof || nframes != pendingFrames)) {
    const int js = joinParts();
    if (js != ORBX_OK) return js;
  }
  if (prof) {
    // timing mode: one stream, stages back to back, HIP events around each stage
    HIP_TRY(hipEventRecord(ev[0], s));
    orbx::launch_pyramid(hp, d_images, istride, iframe, d_pyr, 0, nframes, s);
    HIP_TRY(hipEventRecord(ev[1], s));
    orbx::launch_fast(hp, dp, d_pyr, d_cand, d_cellCount, 0, hp.nlevels, 0, nframes, s);
    HIP_TRY(hipEventRecord(ev[2], s));
    orbx::launch_octree(hp, dp, d_cand, d_cellCount, d_flatA, d_flatB, d_sel, d_selCount, d_octws, 0,
                        hp.nlevels, 0, nframes, s);
    HIP_TRY(hipEventRecord(ev[3], s));
    orbx::launch_assemble(hp, dp, d_sel, d_selCount, lap0, lap1, d_kps, d_work, d_n, d_mono, 0, nframes, s);
    HIP_TRY(hipEventRecord(ev[4], s));
    if (!hp.blur_in_describe) orbx::launch_blur(hp, dp, d_pyr, d_blur, 0, nframes, s);
    HIP_TRY(hipEventRecord(ev[5], s));
    if (hp.blur_in_describe) orbx::launch_describe_fused(hp, dp, d_pyr, d_work, d_n, d_kps, d_desc, 0, nframes, s);
    else orbx::launch_describe(hp, dp, d_pyr, d_blur, d_work, d_n, d_kps, d_desc, 0, nframes, s);
    HIP_TRY(hipEventRecord(ev[6], s));
  } else {
    // The quadtree runs one wave per (frame, level), is latency-bound and leaves most of
    // the chip idle; the blur only needs the pyramid, so it runs beside it on a second
    // stream.  (Starting the big levels' quadtrees earlier, beside the detection of the
    // small levels, measured slower: the two compete for LDS.)
    // A large batch is cut in up to kMaxParts parts that alternate between two streams, each
    // starting one stage behind the previous one, so that the latency-bound pyramid of a part
    // overlaps the VALU-bound detection of the part before it, and its detection overlaps
    // that part's quadtree and descriptors.
    // (a part must be big enough to fill the chip by itself: kSplitMin frames of 752 x 480, fewer
    // frames of a larger image in proportion)
    const long long refPixels = (long long)kSplitMin * 752 * 480;
    const int splitMin = (int)std::max(1LL, std::min((long long)kSplitMin, (refPixels + (long long)hp.W * hp.H - 1) / ((long long)hp.W * hp.H)));
    const int nparts = std::max(1, std::min(maxParts, nframes / splitMin));
    if (nparts > 1) {
      HIP_TRY(hipEventRecord(evStart, s));            // the other part streams follow earlier work on s
      for (int k = 1; k < nparts; ++k) HIP_TRY(hipStreamWaitEvent(lane[k], evStart, 0));
    }
    for (int k = 0, f0 = 0; k < nparts; ++k) {
      const int n = (nframes - f0) / (nparts - k);
      hipStream_t m = k == 0 ? s : lane[k];
      // a part starts its pyramid when the previous part's pyramid is done
      if (k > 0 && pyrWait) HIP_TRY(hipStreamWaitEvent(m, evPyr[k - 1], 0));
      const int st = enqueuePart(k, m, lane[0], evForkP[k], evJoinP[k], f0, n, k + 1 < nparts ? evPyr[k] : nullptr,
                                 d_images, istride, iframe, lap0, lap1);
      if (st != ORBX_OK) return st;
      f0 += n;
    }
    lastParts = nparts;
    detParts[(detCalls - 1) % kDetRing] = nparts;
    if (nparts > 1) {
      for (int k = 1; k < nparts; ++k) HIP_TRY(hipEventRecord(evDone[k], lane[k]));
      pendingJoin = true;
      pendingFrames = nframes;
      pendingParts = nparts;
      // By default the handle's stream covers the whole batch when the call returns: a caller may
      // have fetched orbx_stream() once and order its own work behind it.  Pipelined handles leave
      // the join to whatever follows (see the comment at pendingJoin).
      if (!pipelined) {
        const int js = joinParts();
        if (js != ORBX_OK) return js;
      }
    }
  }
  evValid = prof;
  HIP_TRY(hipGetLastError());
  afterExtraction(nframes);
  return ORBX_OK;
}

// One part of a batch (frames [f0, f0 + n)) on main stream m with the blur on `side`: the blur
// only needs the pyramid, so it runs beside the latency-bound quadtree.
int orbx_extractor::enqueuePart(int pk, hipStream_t m, hipStream_t side, hipEvent_t fork, hipEvent_t join,
                                int f0, int n, hipEvent_t pyrDone, const uint8_t* d_images, int istride,
                                size_t iframe, int lap0, int lap1) {
  if (!offsets_fit_32(hp, istride)) return ORBX_ERR_UNSUPPORTED;
  orbx::launch_pyramid(hp, d_images, istride, iframe, d_pyr, f0, n, m);
  if (pyrDone) HIP_TRY(hipEventRecord(pyrDone, m));
  const bool planes = !hp.blur_in_describe;   // the full-plane blur launch exists only in that schedule
  if (planes && blurPlace == 3) orbx::launch_blur(hp, dp, d_pyr, d_blur, f0, n, m);
  const int ds = (int)((detCalls > 0 ? detCalls - 1 : 0) % kDetRing);
  HIP_TRY(hipEventRecord(evDet[ds][pk][0], m));
  orbx::launch_fast(hp, dp, d_pyr, d_cand, d_cellCount, 0, hp.nlevels, f0, n, m);
  HIP_TRY(hipEventRecord(evDet[ds][pk][1], m));
  detFrames[ds][pk] = n;
  if (planes && blurPlace == 2) orbx::launch_blur(hp, dp, d_pyr, d_blur, f0, n, m);
  if (planes && blurPlace == 1) {
    HIP_TRY(hipEventRecord(fork, m));
    HIP_TRY(hipStreamWaitEvent(side, fork, 0));
    orbx::launch_blur(hp, dp, d_pyr, d_blur, f0, n, side);
    HIP_TRY(hipEventRecord(join, side));
  }
  orbx::launch_octree(hp, dp, d_cand, d_cellCount, d_flatA, d_flatB, d_sel, d_selCount, d_octws, 0,
                      hp.nlevels, f0, n, m);
  orbx::launch_assemble(hp, dp, d_sel, d_selCount, lap0, lap1, d_kps, d_work, d_n, d_mono, f0, n, m);
  if (planes && blurPlace == 1) HIP_TRY(hipStreamWaitEvent(m, join, 0));
  if (planes && blurPlace == 0) orbx::launch_blur(hp, dp, d_pyr, d_blur, f0, n, m);
  if (planes) orbx::launch_describe(hp, dp, d_pyr, d_blur, d_work, d_n, d_kps, d_desc, f0, n, m);
  else orbx::launch_describe_fused(hp, dp, d_pyr, d_work, d_n, d_kps, d_desc, f0, n, m);
  return ORBX_OK;
}

void orbx_extractor::afterExtraction(int nframes) {
  lastFrames = nframes;
  hostCountFrames = 0;
  // results derived from the previous extraction (stereo association, frame grid, bag of words)
  // describe keypoints that no longer exist: their download / query calls fail until the
  // corresponding call has run again
  stereoFrames = 0;
  lapFrames = 0;
  gridFrames = 0;
  bowFrames = 0;
  h_pyr_valid = false;
}

// ---------------------------------------------------------------------------
extern "C" {

const char* orbx_strerror(int status) {
  switch (status) {
    case ORBX_OK: return "ok";
    case ORBX_ERR_EMPTY_IMAGE: return "empty image";
    case ORBX_ERR_TOO_SMALL: return "image too small: a pyramid level has no 35-pixel detection cell (or no quadtree root: much taller than wide)";
    case ORBX_ERR_CAPACITY: return "output capacity too small";
    case ORBX_ERR_BAD_ARG: return "bad argument";
    case ORBX_ERR_NO_DEVICE: return "no usable HIP device";
    case ORBX_ERR_HIP: return "HIP runtime error";
    case ORBX_ERR_UNSUPPORTED: return "parameters beyond what the kernels are sized for";
    case ORBX_ERR_BATCH: return "more frames than the handle's max_batch";
    case ORBX_ERR_SCRATCH: return "input too large for the fixed scratch buffer of the allocation-free entry point";
    case ORBX_ERR_COMM: return "RCCL library not found or an RCCL call failed";
    default: return "unknown status";
  }
}

extern "C++" {
namespace orbx {
// Several entry points take short-lived device memory from the device's stream-ordered pool
// (hipMallocAsync).  With the pool's default release threshold of zero every such allocation is
// given back to the system at the next synchronisation and mapped again by the next call (close to
// a millisecond each time); once per device the threshold is raised so that the pool keeps what it
// has handed out.  The pool is HIP's per-device default pool; PyTorch's caching allocator does not
// use it.
void keep_pool_memory(int device) {
  static std::mutex mu;
  static bool done[64] = {};
  if (device < 0 || device >= 64) return;
  std::lock_guard<std::mutex> lk(mu);
  if (done[device]) return;
  done[device] = true;
  hipMemPool_t pool = nullptr;
  if (hipDeviceGetDefaultMemPool(&pool, device) != hipSuccess || !pool) { (void)hipGetLastError(); return; }
  uint64_t keep = UINT64_MAX;
  if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) != hipSuccess) (void)hipGetLastError();
}
}  // namespace orbx
}  // extern "C++"
using orbx::keep_pool_memory;

int orbx_create(const orbx_params* p, orbx_extractor** out) {
  if (!p || !out) return ORBX_ERR_BAD_ARG;
  *out = nullptr;
  if (p->nfeatures < 1 || p->nlevels < 1 || p->nlevels > ORBX_MAX_LEVELS || !(p->scale_factor > 1.0f) ||
      p->max_batch < 1)
    return ORBX_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return ORBX_ERR_NO_DEVICE;
  if (p->device_id < 0 || p->device_id >= ndev) return ORBX_ERR_NO_DEVICE;
  orbx_extractor* ex = new (std::nothrow) orbx_extractor();
  if (!ex) return ORBX_ERR_BAD_ARG;
  ex->prm = *p;
  ex->device = p->device_id;
  try {
    ex->buildTables();
  } catch (...) {
    delete ex;
    return ORBX_ERR_UNSUPPORTED;
  }
  if (hipSetDevice(ex->device) == hipSuccess) keep_pool_memory(ex->device);
  if (const char* e = std::getenv("ORBX_PYR_WAIT")) ex->pyrWait = e[0] != '0';
  if (hipSetDevice(ex->device) != hipSuccess ||
      hipStreamCreateWithFlags(&ex->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ex;
    return ORBX_ERR_NO_DEVICE;
  }
  for (int i = 0; i <= ORBX_NUM_STAGES; ++i)
    if (hipEventCreate(&ex->ev[i]) != hipSuccess) { orbx_destroy(ex); return ORBX_ERR_NO_DEVICE; }
  for (int r = 0; r < orbx_extractor::kDetRing; ++r)
    for (int i = 0; i < 2 * orbx_extractor::kMaxParts; ++i)
      if (hipEventCreate(&ex->evDet[r][i >> 1][i & 1]) != hipSuccess) { orbx_destroy(ex); return ORBX_ERR_NO_DEVICE; }
  bool ok = hipEventCreateWithFlags(&ex->evFork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ex->evStart, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ex->evForkB, hipEventDisableTiming) == hipSuccess;
  for (int k = 0; k < orbx_extractor::kMaxParts && ok; ++k)
    ok = hipEventCreateWithFlags(&ex->evPyr[k], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&ex->evForkP[k], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&ex->evJoinP[k], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&ex->evDone[k], hipEventDisableTiming) == hipSuccess;
  if (const char* e = std::getenv("ORBX_PARTS")) {
    const int v = std::atoi(e);
    if (v >= 1 && v <= orbx_extractor::kMaxParts) ex->maxParts = v;
  }
  if (const char* e = std::getenv("ORBX_SIDE_BLUR")) {
    const int v = std::atoi(e);
    if (v >= 0 && v <= 3) ex->blurPlace = v;
  }
  for (int k = 0; k < orbx_extractor::kLanes && ok; ++k)
    ok = hipStreamCreateWithFlags(&ex->lane[k], hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&ex->evJoin[k], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&ex->sIn, hipStreamNonBlocking) == hipSuccess &&
       hipStreamCreateWithFlags(&ex->sOut, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 2 && ok; ++i)
    ok = hipEventCreateWithFlags(&ex->evIn[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&ex->evComp[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&ex->evOut[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) { orbx_destroy(ex); return ORBX_ERR_NO_DEVICE; }
  *out = ex;
  return ORBX_OK;
}

void orbx_destroy(orbx_extractor* ex) {
  if (!ex) return;
  (void)hipSetDevice(ex->device);
  ex->freePlan();
  for (int i = 0; i <= ORBX_NUM_STAGES; ++i) if (ex->ev[i]) (void)hipEventDestroy(ex->ev[i]);
  for (int r = 0; r < orbx_extractor::kDetRing; ++r)
    for (int i = 0; i < 2 * orbx_extractor::kMaxParts; ++i)
      if (ex->evDet[r][i >> 1][i & 1]) (void)hipEventDestroy(ex->evDet[r][i >> 1][i & 1]);
  for (int k = 0; k < orbx_extractor::kLanes; ++k) {
    if (ex->lane[k]) { (void)hipStreamSynchronize(ex->lane[k]); (void)hipStreamDestroy(ex->lane[k]); }
    if (ex->evJoin[k]) (void)hipEventDestroy(ex->evJoin[k]);
  }
  if (ex->evFork) (void)hipEventDestroy(ex->evFork);
  if (ex->evStart) (void)hipEventDestroy(ex->evStart);
  for (int k = 0; k < orbx_extractor::kMaxParts; ++k) {
    if (ex->evPyr[k]) (void)hipEventDestroy(ex->evPyr[k]);
    if (ex->evForkP[k]) (void)hipEventDestroy(ex->evForkP[k]);
    if (ex->evJoinP[k]) (void)hipEventDestroy(ex->evJoinP[k]);
    if (ex->evDone[k]) (void)hipEventDestroy(ex->evDone[k]);
  }
  if (ex->evForkB) (void)hipEventDestroy(ex->evForkB);
  if (ex->evStereo) (void)hipEventDestroy(ex->evStereo);
  for (int i = 0; i < 2; ++i) {
    if (ex->evIn[i]) (void)hipEventDestroy(ex->evIn[i]);
    if (ex->evComp[i]) (void)hipEventDestroy(ex->evComp[i]);
    if (ex->evOut[i]) (void)hipEventDestroy(ex->evOut[i]);
  }
  if (ex->sIn) { (void)hipStreamSynchronize(ex->sIn); (void)hipStreamDestroy(ex->sIn); }
  if (ex->sOut) { (void)hipStreamSynchronize(ex->sOut); (void)hipStreamDestroy(ex->sOut); }
  if (ex->stream) (void)hipStreamDestroy(ex->stream);
  delete ex->pool;
  delete ex;
}

int orbx_get_tables(const orbx_extractor* ex, float* scale, float* inv_scale, float* sigma2,
                    float* inv_sigma2, int32_t* features_per_level) {
  if (!ex) return ORBX_ERR_BAD_ARG;
  for (int i = 0; i < ex->prm.nlevels; ++i) {
    if (scale) scale[i] = ex->scale[i];
    if (inv_scale) inv_scale[i] = ex->invScale[i];
    if (sigma2) sigma2[i] = ex->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = ex->invSigma2[i];
    if (features_per_level) features_per_level[i] = ex->quota[i];
  }
  return ORBX_OK;
}

int orbx_max_keypoints(const orbx_extractor* ex) {
  if (!ex) return ORBX_ERR_BAD_ARG;
  if (ex->planned) return ex->maxKeypoints;
  int sum = 0;
  for (int l = 0; l < ex->prm.nlevels; ++l) sum += ex->quota[l] + 8 + 4 * ORBX_MAX_ROOTS;
  return sum;
}

void* orbx_stream(orbx_extractor* ex) {
  if (!ex) return nullptr;
  // callers order their own work against this stream: it must cover everything enqueued so far
  if (hipSetDevice(ex->device) == hipSuccess) (void)ex->joinParts();
  return (void*)ex->stream;
}

int orbx_sync(orbx_extractor* ex) {
  if (!ex) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  hipError_t e = hipStreamSynchronize(ex->stream);
  if (e != hipSuccess) { ex->last_hip_error = e; return ORBX_ERR_HIP; }
  return ORBX_OK;
}

int orbx_extract_batch_device(orbx_extractor* ex, int n_frames, const uint8_t* d_imgs, int width,
                              int height, int stride, int lap0, int lap1) {
  if (!ex) return ORBX_ERR_BAD_ARG;
  if (!d_imgs || width <= 0 || height <= 0 || n_frames <= 0) return ORBX_ERR_EMPTY_IMAGE;
  if (stride < width) return ORBX_ERR_BAD_ARG;
  if (n_frames > ex->prm.max_batch) return ORBX_ERR_BATCH;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  int st = ex->ensurePlan(width, height);
  if (st != ORBX_OK) { ex->freePlan(); return st; }
  return ex->run(d_imgs, stride, (size_t)stride * height, n_frames, lap0, lap1);
}

int orbx_set_pipelined(orbx_extractor* ex, int enabled) {
  if (!ex) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  ex->pipelined = enabled != 0;
  return ORBX_OK;
}

int orbx_batch_results_device(orbx_extractor* ex, orbx_keypoint** d_kps, uint8_t** d_desc,
                              int32_t** d_n, int32_t** d_mono, int* cap) {
  if (!ex || !ex->planned) return ORBX_ERR_BAD_ARG;
  // whoever asks for the buffers is about to read them behind the handle's stream
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  if (d_kps) *d_kps = (orbx_keypoint*)ex->d_kps;
  if (d_desc) *d_desc = ex->d_desc;
  if (d_n) *d_n = ex->d_n;
  if (d_mono) *d_mono = ex->d_mono;
  if (cap) *cap = ex->hp.kp_cap;
  return ORBX_OK;
}

int orbx_batch_download(orbx_extractor* ex, int frame, orbx_keypoint* kps, uint8_t* desc, int cap,
                        int* n_out, int* mono_out) {
  if (!ex || !ex->planned || frame < 0 || frame >= ex->lastFrames) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  hipStream_t s = ex->stream;
  int nm[2];
  {
    hipError_t e;
    e = hipMemcpyAsync(&nm[0], ex->d_n + frame, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&nm[1], ex->d_mono + frame, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { ex->last_hip_error = e; return ORBX_ERR_HIP; }
  }
  const int n = nm[0];
  if (n_out) *n_out = n;
  if (mono_out) *mono_out = nm[1];
  if (n > cap) return ORBX_ERR_CAPACITY;
  if (n > 0) {
    hipError_t e = hipSuccess;
    const size_t base = (size_t)frame * ex->hp.kp_cap;
    if (kps) e = hipMemcpyAsync(kps, ex->d_kps + base * 7, (size_t)n * 28, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && desc)
      e = hipMemcpyAsync(desc, ex->d_desc + base * 32, (size_t)n * 32, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { ex->last_hip_error = e; return ORBX_ERR_HIP; }
  }
  return ORBX_OK;
}

namespace {
// Is [p, p + bytes) page-locked host memory the GPU can reach directly (hipHostMalloc /
// hipHostRegister)?  Ordinary (pageable) memory makes the query fail; that failure is expected
// and must not stay behind as the thread's last HIP error.
bool is_pinned_host(const void* p, size_t bytes) {
  if (!p) return false;
  hipPointerAttribute_t a;
  const uint8_t* ends[2] = {(const uint8_t*)p, (const uint8_t*)p + (bytes ? bytes - 1 : 0)};
  for (const uint8_t* q : ends) {
    std::memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
  }
  return true;
}
}  // namespace

// A few frames from host memory - one is the call Frame::ExtractORB makes
// (core/src/Frame.cc:461-470), two a stereo pair, four a rig: no groups, no helper streams, no
// packing threads and no copy engines.  The images are packed into the handle's page-locked
// buffer, which the first pyramid kernel reads over the bus; a last small kernel writes the counts
// and every frame's n records into page-locked memory; one synchronisation.
// (Asking the runtime whether the caller's image is page-locked costs ten times what packing a
// frame of a few hundred KB costs, so only a single large frame asks, and that one goes through a
// copy.)  Returns kNotFew when the frames are too many or too large for this path.
constexpr int kNotFew = 1;
constexpr int kFewFrames = 8;

static int extract_few(orbx_extractor* ex, int n_frames, const uint8_t* const* imgs, int width, int height,
                       int stride, int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                       int* mono_out) {
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  const size_t frameBytes = (size_t)width * height;
  const size_t srcBytes = (size_t)stride * (height - 1) + width;
  const int kc = ex->hp.kp_cap, G = ex->groupFrames;
  const bool small = (size_t)n_frames * frameBytes <= ((size_t)4 << 20);
  if (n_frames > kFewFrames || n_frames > G || (!small && n_frames > 1)) return kNotFew;
  const int ccap = std::min(cap, kc);
  const bool wantK = kps && ccap > 0, wantD = desc && ccap > 0;
  const bool pinnedIn = !small && is_pinned_host(imgs[0], srcBytes);
  int st = ex->ensureHostStaging(!pinnedIn, true);
  if (st != ORBX_OK) return st;
  const uint8_t* src = imgs[0];
  int sstride = stride;
  if (!pinnedIn) {
    for (int f = 0; f < n_frames; ++f) {
      uint8_t* hin = ex->h_in[0] + (size_t)f * frameBytes;
      const uint8_t* img = imgs[f];
      if (stride == width) std::memcpy(hin, img, frameBytes);
      else for (int y = 0; y < height; ++y) std::memcpy(hin + (size_t)y * width, img + (size_t)y * stride, width);
    }
    src = ex->h_in[0];
    sstride = width;
  }
  void* srcDev = nullptr;
  if (small && hipHostGetDevicePointer(&srcDev, (void*)src, 0) == hipSuccess && srcDev) {
    st = ex->run((const uint8_t*)srcDev, width, frameBytes, n_frames, lap0, lap1);
  } else {
    (void)hipGetLastError();
    if (sstride == width) HIP_TRY(hipMemcpyAsync(ex->d_img, src, (size_t)n_frames * frameBytes, hipMemcpyHostToDevice, s));
    else HIP_TRY(hipMemcpy2DAsync(ex->d_img, width, src, sstride, width, height, hipMemcpyHostToDevice, s));
    st = ex->run(ex->d_img, width, frameBytes, n_frames, lap0, lap1);
  }
  if (st != ORBX_OK) return st;
  int* hn = ex->h_cnt[0];
  uint8_t* hk = ex->h_out[0];
  uint8_t* hd = hk + (size_t)G * kc * 28;
  orbx::launch_export_few(ex->d_n, ex->d_mono, ex->d_kps, ex->d_desc, n_frames, kc, ccap, G, hn, hk, hd, s);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  int worst = ORBX_OK;
  try {
    ex->hostN.assign(hn, hn + n_frames);
    ex->hostMono.assign(hn + G, hn + G + n_frames);
    ex->hostCountFrames = n_frames;
  } catch (...) {
    ex->hostCountFrames = 0;
  }
  for (int f = 0; f < n_frames; ++f) {
    const int n = hn[f];
    if (n_out) n_out[f] = n;
    if (mono_out) mono_out[f] = hn[G + f];
    if (n > cap) { worst = ORBX_ERR_CAPACITY; continue; }
    if (n > 0) {
      if (wantK) std::memcpy(kps + (size_t)f * cap, hk + (size_t)f * kc * 28, (size_t)n * 28);
      if (wantD) std::memcpy(desc + (size_t)f * cap * 32, hd + (size_t)f * kc * 32, (size_t)n * 32);
    }
  }
  return worst;
}

int orbx_extract_batch(orbx_extractor* ex, int n_frames, const uint8_t* const* imgs, int width,
                       int height, int stride, int lap0, int lap1, orbx_keypoint* kps,
                       uint8_t* desc, int cap, int* n_out, int* mono_out) {
  if (!ex) return ORBX_ERR_BAD_ARG;
  if (!imgs || width <= 0 || height <= 0 || n_frames <= 0) return ORBX_ERR_EMPTY_IMAGE;
  for (int f = 0; f < n_frames; ++f) if (!imgs[f]) return ORBX_ERR_EMPTY_IMAGE;
  if (stride < width || cap < 0) return ORBX_ERR_BAD_ARG;
  if (n_frames > ex->prm.max_batch) return ORBX_ERR_BATCH;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  int st = ex->ensurePlan(width, height);
  if (st != ORBX_OK) { ex->freePlan(); return st; }
  if (n_frames <= kFewFrames) {
    st = extract_few(ex, n_frames, imgs, width, height, stride, lap0, lap1, kps, desc, cap, n_out, mono_out);
    if (st != kNotFew) return st;
  }
  auto& last_hip_error = ex->last_hip_error;
  const size_t frameBytes = (size_t)width * height;
  const size_t srcBytes = (size_t)stride * (height - 1) + width;
  const int kc = ex->hp.kp_cap, G = ex->groupFrames;
  const int ngroups = (n_frames + G - 1) / G;

  // caller memory the copy engines can use directly needs no staging and no CPU copy
  bool pinnedIn = true;
  for (int f = 0; f < n_frames && pinnedIn; ++f) pinnedIn = is_pinned_host(imgs[f], srcBytes);
  const int ccap = std::min(cap, kc);                      // slots per frame the caller can take
  const bool wantK = kps && ccap > 0, wantD = desc && ccap > 0;
  const bool pinnedOut = (!wantK || is_pinned_host(kps, (size_t)n_frames * cap * 28)) &&
                         (!wantD || is_pinned_host(desc, (size_t)n_frames * cap * 32));
  const bool stageOut = !pinnedOut && (wantK || wantD);
  st = ex->ensureHostStaging(!pinnedIn, stageOut);
  if (st != ORBX_OK) return st;
  HostPool* pool = ex->pool;
  auto par = [&](int n, const std::function<void(int)>& fn) {
    if (pool) pool->parallel_for(n, fn); else for (int i = 0; i < n; ++i) fn(i);
  };

  // work enqueued earlier on the handle's stream (a previous device-resident call) comes first
  HIP_TRY(hipEventRecord(ex->evStart, ex->stream));
  HIP_TRY(hipStreamWaitEvent(ex->lane[1], ex->evStart, 0));
  HIP_TRY(hipStreamWaitEvent(ex->sIn, ex->evStart, 0));

  int worst = ORBX_OK;
  bool countsKept = true;   // the counts read back below stay with the handle for later blocking calls
  try {
    ex->hostN.assign((size_t)n_frames, 0);
    ex->hostMono.assign((size_t)n_frames, 0);
  } catch (...) {
    countsKept = false;
  }
  // group k is done on the device and its results are where the caller wants them
  auto finish = [&](int k) -> int {
    const int b = k & 1, f0 = k * G, m = std::min(G, n_frames - f0);
    HIP_TRY(hipEventSynchronize(ex->evOut[b]));
    const int* hn = ex->h_cnt[b];
    const int* hm = hn + G;
    for (int i = 0; i < m; ++i) {
      if (n_out) n_out[f0 + i] = hn[i];
      if (mono_out) mono_out[f0 + i] = hm[i];
      if (hn[i] > cap) worst = ORBX_ERR_CAPACITY;
      if (countsKept) { ex->hostN[f0 + i] = hn[i]; ex->hostMono[f0 + i] = hm[i]; }
    }
    if (stageOut) {
      const uint8_t* hk = ex->h_out[b];
      const uint8_t* hd = hk + (size_t)G * kc * 28;
      par(m, [&](int i) {
        const int n = std::min(hn[i], ccap);
        if (hn[i] > cap || n <= 0) return;
        if (wantK) std::memcpy(kps + (size_t)(f0 + i) * cap, hk + (size_t)i * kc * 28, (size_t)n * 28);
        if (wantD) std::memcpy(desc + (size_t)(f0 + i) * cap * 32, hd + (size_t)i * kc * 32, (size_t)n * 32);
      });
    }
    return ORBX_OK;
  };

  for (int k = 0; k < ngroups; ++k) {
    const int b = k & 1, f0 = k * G, m = std::min(G, n_frames - f0);
    uint8_t* dimg = ex->d_img + (size_t)f0 * frameBytes;
    // ---- images to the device (copy-in stream)
    if (pinnedIn) {
      for (int i = 0; i < m;) {
        if (stride == width) {
          int j = i + 1;    // frames that follow each other in memory travel as one transfer
          while (j < m && imgs[f0 + j] == imgs[f0 + j - 1] + frameBytes) ++j;
          HIP_TRY(hipMemcpyAsync(dimg + (size_t)i * frameBytes, imgs[f0 + i], (size_t)(j - i) * frameBytes,
                                 hipMemcpyHostToDevice, ex->sIn));
          i = j;
        } else {
          HIP_TRY(hipMemcpy2DAsync(dimg + (size_t)i * frameBytes, width, imgs[f0 + i], stride, width, height,
                                   hipMemcpyHostToDevice, ex->sIn));
          ++i;
        }
      }
    } else {
      if (k >= 2) HIP_TRY(hipEventSynchronize(ex->evIn[b]));   // the staging buffer's last transfer is done
      uint8_t* hin = ex->h_in[b];
      par(m, [&](int i) {
        uint8_t* dst = hin + (size_t)i * frameBytes;
        const uint8_t* src = imgs[f0 + i];
        if (stride == width) std::memcpy(dst, src, frameBytes);
        else for (int y = 0; y < height; ++y) std::memcpy(dst + (size_t)y * width, src + (size_t)y * stride, width);
      });
      HIP_TRY(hipMemcpyAsync(dimg, hin, (size_t)m * frameBytes, hipMemcpyHostToDevice, ex->sIn));
    }
    HIP_TRY(hipEventRecord(ex->evIn[b], ex->sIn));
    // ---- kernels: groups alternate between the handle's two stream pairs, like the parts of a
    // device-resident batch
    const bool odd = k & 1;
    hipStream_t ms = odd ? ex->lane[1] : ex->stream;
    HIP_TRY(hipStreamWaitEvent(ms, ex->evIn[b], 0));
    // (a slot of its own in the launch-event ring, reporting no launches: the groups of the host
    // path are not the schedule those events are kept for)
    ++ex->detCalls;
    ex->detParts[(ex->detCalls - 1) % orbx_extractor::kDetRing] = 0;
    st = ex->enqueuePart(b, ms, ex->lane[0], ex->evForkP[b], ex->evJoinP[b], f0, m, nullptr, ex->d_img, width, frameBytes,
                         lap0, lap1);
    if (st != ORBX_OK) return st;
    HIP_TRY(hipEventRecord(ex->evComp[b], ms));
    // ---- results to the host (copy-out stream); the staging buffers of group k-2 were emptied
    // by finish(k-2) before this point
    HIP_TRY(hipStreamWaitEvent(ex->sOut, ex->evComp[b], 0));
    HIP_TRY(hipMemcpyAsync(ex->h_cnt[b], ex->d_n + f0, (size_t)m * 4, hipMemcpyDeviceToHost, ex->sOut));
    HIP_TRY(hipMemcpyAsync(ex->h_cnt[b] + G, ex->d_mono + f0, (size_t)m * 4, hipMemcpyDeviceToHost, ex->sOut));
    if (stageOut) {
      uint8_t* hk = ex->h_out[b];
      uint8_t* hd = hk + (size_t)G * kc * 28;
      if (wantK) HIP_TRY(hipMemcpyAsync(hk, ex->d_kps + (size_t)f0 * kc * 7, (size_t)m * kc * 28, hipMemcpyDeviceToHost, ex->sOut));
      if (wantD) HIP_TRY(hipMemcpyAsync(hd, ex->d_desc + (size_t)f0 * kc * 32, (size_t)m * kc * 32, hipMemcpyDeviceToHost, ex->sOut));
    } else {
      // straight into the caller's arrays: the first min(cap, slots) entries of every frame;
      // entries beyond a frame's n_out are unspecified
      if (wantK) HIP_TRY(hipMemcpy2DAsync(kps + (size_t)f0 * cap, (size_t)cap * 28, ex->d_kps + (size_t)f0 * kc * 7,
                                          (size_t)kc * 28, (size_t)ccap * 28, m, hipMemcpyDeviceToHost, ex->sOut));
      if (wantD) HIP_TRY(hipMemcpy2DAsync(desc + (size_t)f0 * cap * 32, (size_t)cap * 32, ex->d_desc + (size_t)f0 * kc * 32,
                                          (size_t)kc * 32, (size_t)ccap * 32, m, hipMemcpyDeviceToHost, ex->sOut));
    }
    HIP_TRY(hipEventRecord(ex->evOut[b], ex->sOut));
    if (k >= 1) { st = finish(k - 1); if (st != ORBX_OK) return st; }
  }
  st = finish(ngroups - 1);
  if (st != ORBX_OK) return st;
  // later work on the handle's stream follows everything this call ran on the other streams
  HIP_TRY(hipEventRecord(ex->evJoin[1], ex->lane[1]));
  HIP_TRY(hipStreamWaitEvent(ex->stream, ex->evJoin[1], 0));
  ex->lastParts = 0;
  ex->evValid = false;
  HIP_TRY(hipGetLastError());
  ex->afterExtraction(n_frames);
  if (countsKept) ex->hostCountFrames = n_frames;
  return worst;
}

int orbx_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return ORBX_ERR_BAD_ARG;
  *out = nullptr;
  return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? ORBX_OK : ORBX_ERR_HIP;
}

int orbx_host_free(void* p) {
  if (!p) return ORBX_OK;
  return hipHostFree(p) == hipSuccess ? ORBX_OK : ORBX_ERR_HIP;
}

int orbx_extract(orbx_extractor* ex, const uint8_t* img, int width, int height, int stride, int lap0,
                 int lap1, orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out, int* mono_out) {
  if (n_out) *n_out = 0;
  if (mono_out) *mono_out = -1;
  if (!ex) return ORBX_ERR_BAD_ARG;
  if (!img || width <= 0 || height <= 0) return ORBX_ERR_EMPTY_IMAGE;
  const uint8_t* one[1] = {img};
  int n = 0, mono = -1;
  int st = orbx_extract_batch(ex, 1, one, width, height, stride, lap0, lap1, kps, desc, cap, &n, &mono);
  if (n_out) *n_out = n;
  if (mono_out) *mono_out = mono;
  return st;
}

// ---- stereo association on two device-resident extractions ---------------------------------
static bool same_geometry(const orbx_extractor* a, const orbx_extractor* b) {
  return a->planned && b->planned && a->device == b->device && a->hp.W == b->hp.W &&
         a->hp.H == b->hp.H && a->prm.nlevels == b->prm.nlevels &&
         a->prm.scale_factor == b->prm.scale_factor;
}

// the per-keypoint buffers of a stereo association (allocated on first use, freed with the plan)
static int ensure_stereo_buffers(orbx_extractor* ex) {
  auto& last_hip_error = ex->last_hip_error;
  const size_t slots = (size_t)ex->prm.max_batch * ex->hp.kp_cap;
  if (!ex->d_stRowStart)
    HIP_TRY(hipMalloc((void**)&ex->d_stRowStart, (size_t)ex->prm.max_batch * (ex->hp.lv[0].h + 1) * 4));
  if (!ex->d_stBandIdx) HIP_TRY(hipMalloc((void**)&ex->d_stBandIdx, slots * 4));
  if (!ex->d_stBandDist) HIP_TRY(hipMalloc((void**)&ex->d_stBandDist, slots * 4));
  if (!ex->d_stSad) HIP_TRY(hipMalloc((void**)&ex->d_stSad, slots * 4));
  if (!ex->d_stDepth) HIP_TRY(hipMalloc((void**)&ex->d_stDepth, slots * 4));
  if (!ex->d_stURight) HIP_TRY(hipMalloc((void**)&ex->d_stURight, slots * 4));
  return ORBX_OK;
}

int orbx_stereo_from_rgbd_batch_device(orbx_extractor* ex, int n_frames, const float* d_depth, int stride_floats,
                                       float mbf) {
  if (!ex || !ex->planned || !ex->d_kpsUn || n_frames <= 0 || !d_depth || stride_floats < ex->hp.W)
    return ORBX_ERR_BAD_ARG;
  if (n_frames > ex->gridFrames) return ORBX_ERR_BATCH;   // the undistorted keypoints come from the grid call
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  const int st = ensure_stereo_buffers(ex);
  if (st != ORBX_OK) return st;
  orbx::launch_stereo_from_rgbd(ex->d_kps, ex->d_kpsUn, ex->d_n, ex->hp.kp_cap, d_depth,
                                (size_t)stride_floats * ex->hp.H, stride_floats, ex->hp.W, ex->hp.H, mbf,
                                ex->d_stURight, ex->d_stDepth, ex->d_stSad, n_frames, ex->stream);
  HIP_TRY(hipGetLastError());
  ex->stereoFrames = n_frames;
  return ORBX_OK;
}

// `blocking`: the caller waits for the handle's stream before it returns, so the right handle needs
// no order behind this work.
static int stereo_enqueue(orbx_extractor* left, orbx_extractor* right, int n_frames, float mbf, float mb,
                          bool blocking) {
  if (!left || !right || left == right || n_frames <= 0 || !(mb > 0.f) || !(mbf > 0.f))
    return ORBX_ERR_BAD_ARG;
  if (!same_geometry(left, right)) return ORBX_ERR_BAD_ARG;
  if (n_frames > left->lastFrames || n_frames > right->lastFrames) return ORBX_ERR_BATCH;
  orbx_extractor* ex = left;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  {
    const int st = ensure_stereo_buffers(ex);
    if (st != ORBX_OK) return st;
  }
  if (right->hp.kp_cap > ex->stRightCap) {
    HIP_TRY(hipStreamSynchronize(ex->stream));
    if (ex->d_stEntries) HIP_TRY(hipFree(ex->d_stEntries));
    ex->d_stEntries = nullptr;
    HIP_TRY(hipMalloc((void**)&ex->d_stEntries, (size_t)ex->prm.max_batch * right->hp.kp_cap *
                                                    orbx::stereo_entries_per_keypoint(ex->hp) * 4));
    ex->stRightCap = right->hp.kp_cap;
  }
  if (!ex->evStereo) HIP_TRY(hipEventCreateWithFlags(&ex->evStereo, hipEventDisableTiming));
  // the right extraction runs on its own stream: order this work after it (nothing to order when
  // that stream has drained, as after a blocking extraction call)
  if (const int js_ = right->joinParts()) return js_;
  const hipError_t busy = hipStreamQuery(right->stream);
  if (busy == hipErrorNotReady) {
    (void)hipGetLastError();
    HIP_TRY(hipEventRecord(ex->evStereo, right->stream));
    HIP_TRY(hipStreamWaitEvent(ex->stream, ex->evStereo, 0));
  } else {
    HIP_TRY(busy);
  }
  orbx::launch_stereo(ex->hp, ex->dp, ex->d_pyr, right->d_pyr, ex->d_kps, ex->d_desc, ex->d_n,
                      ex->hp.kp_cap, right->d_kps, right->d_desc, right->d_n, right->hp.kp_cap, mbf, mb,
                      ex->d_stRowStart, ex->d_stEntries, ex->d_stBandIdx, ex->d_stBandDist, ex->d_stSad, ex->d_stURight, ex->d_stDepth,
                      n_frames, ex->stream);
  HIP_TRY(hipGetLastError());
  // the right handle's buffers must not be overwritten before this work has read them
  if (!blocking) {
    HIP_TRY(hipEventRecord(ex->evStereo, ex->stream));
    HIP_TRY(hipStreamWaitEvent(right->stream, ex->evStereo, 0));
  }
  ex->stereoFrames = n_frames;
  return ORBX_OK;
}

int orbx_stereo_match_batch_device(orbx_extractor* left, orbx_extractor* right, int n_frames, float mbf,
                                   float mb) {
  return stereo_enqueue(left, right, n_frames, mbf, mb, false);
}

int orbx_stereo_results_device(orbx_extractor* left, float** d_uright, float** d_depth, int* cap) {
  if (!left || !left->planned || !left->d_stURight) return ORBX_ERR_BAD_ARG;
  if (d_uright) *d_uright = left->d_stURight;
  if (d_depth) *d_depth = left->d_stDepth;
  if (cap) *cap = left->hp.kp_cap;
  return ORBX_OK;
}

int orbx_stereo_download(orbx_extractor* left, int frame, float* uright, float* depth,
                         int32_t* l1_dist, int cap, int* n_out) {
  orbx_extractor* ex = left;
  if (!ex || !ex->planned || !ex->d_stURight || frame < 0 || frame >= ex->stereoFrames)
    return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  int n = 0;
  HIP_TRY(hipMemcpyAsync(&n, ex->d_n + frame, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n_out) *n_out = n;
  if (n > cap) return ORBX_ERR_CAPACITY;
  if (n > 0) {
    const size_t base = (size_t)frame * ex->hp.kp_cap;
    if (uright) HIP_TRY(hipMemcpyAsync(uright, ex->d_stURight + base, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (depth) HIP_TRY(hipMemcpyAsync(depth, ex->d_stDepth + base, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (l1_dist) HIP_TRY(hipMemcpyAsync(l1_dist, ex->d_stSad + base, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return ORBX_OK;
}

int orbx_stereo_match(orbx_extractor* left, orbx_extractor* right, float mbf, float mb, float* uright,
                      float* depth, int cap, int* n_out) {
  if (cap < 0) return ORBX_ERR_BAD_ARG;
  int st = stereo_enqueue(left, right, 1, mbf, mb, true);
  if (st != ORBX_OK) return st;
  // count and results through page-locked memory the GPU writes, one wait (as in extract_few)
  orbx_extractor* ex = left;
  auto& last_hip_error = ex->last_hip_error;
  const hipError_t queued = hipGetLastError();
  st = ex->ensureHostStaging(false, true);
  if (queued != hipSuccess || st != ORBX_OK) {
    (void)hipStreamSynchronize(ex->stream);   // blocking promise: nothing of this call is left running
    HIP_TRY(queued);
    return st;
  }
  const int kc = ex->hp.kp_cap, ccap = std::min(cap, kc);
  int* hn = ex->h_cnt[0];
  float* hu = (float*)ex->h_out[0];
  float* hdp = hu + kc;
  orbx::launch_stereo_export_one(ex->d_n, ex->d_stURight, ex->d_stDepth, ccap, hn, hu, hdp, ex->stream);
  const hipError_t launched = hipGetLastError();
  HIP_TRY(hipStreamSynchronize(ex->stream));
  HIP_TRY(launched);
  const int n = hn[0];
  if (n_out) *n_out = n;
  if (n > cap) return ORBX_ERR_CAPACITY;
  if (n > 0) {
    if (uright) std::memcpy(uright, hu, (size_t)n * 4);
    if (depth) std::memcpy(depth, hdp, (size_t)n * 4);
  }
  return ORBX_OK;
}

int orbx_lapping_knn2(orbx_extractor* left, orbx_extractor* right, int frame, uint32_t* idx, uint16_t* dist,
                      int cap, int* n_query, int* mono_left, int* n_train, int* mono_right) {
  if (!left || !right || left == right || !left->planned || !right->planned || left->device != right->device ||
      frame < 0 || frame >= left->lastFrames || frame >= right->lastFrames || cap < 0)
    return ORBX_ERR_BAD_ARG;
  orbx_extractor* ex = left;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  if (!ex->evStereo) HIP_TRY(hipEventCreateWithFlags(&ex->evStereo, hipEventDisableTiming));
  // the right extraction runs on its own stream: order this work after it (nothing to order when
  // that stream has drained, as after a blocking extraction call)
  if (const int js_ = right->joinParts()) return js_;
  const hipError_t busy = hipStreamQuery(right->stream);
  if (busy == hipErrorNotReady) {
    (void)hipGetLastError();
    HIP_TRY(hipEventRecord(ex->evStereo, right->stream));
    HIP_TRY(hipStreamWaitEvent(s, ex->evStereo, 0));
  } else {
    HIP_TRY(busy);
  }
  int h[4] = {0, 0, 0, 0};   // Nleft, monoLeft, Nright, monoRight
  if (frame < left->hostCountFrames && frame < right->hostCountFrames) {
    // both extractions came through host-pointer calls, which read these counts back
    h[0] = left->hostN[frame]; h[1] = left->hostMono[frame];
    h[2] = right->hostN[frame]; h[3] = right->hostMono[frame];
  } else {
    HIP_TRY(hipMemcpyAsync(&h[0], left->d_n + frame, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&h[1], left->d_mono + frame, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&h[2], right->d_n + frame, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&h[3], right->d_mono + frame, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  const int nq = h[0] - h[1], nt = h[2] - h[3];
  if (n_query) *n_query = nq;
  if (mono_left) *mono_left = h[1];
  if (n_train) *n_train = nt;
  if (mono_right) *mono_right = h[3];
  if (nq < 0 || nt < 0) return ORBX_ERR_BAD_ARG;
  if (nq > cap) return ORBX_ERR_CAPACITY;
  if (nq == 0) return ORBX_OK;
  if (!idx || !dist) return ORBX_ERR_BAD_ARG;
  if (nq > left->hp.kp_cap) return ORBX_ERR_BAD_ARG;
  // records on the device, then through page-locked memory the GPU writes: one wait
  const size_t oDist = (size_t)left->hp.kp_cap * 8;
  if (!ex->d_lapOne) HIP_TRY(hipMalloc((void**)&ex->d_lapOne, (size_t)left->hp.kp_cap * 12));
  int st = ex->ensureHostStaging(false, true);
  if (st != ORBX_OK) return st;
  uint8_t* d = ex->d_lapOne;
  const uint8_t* q = left->d_desc + ((size_t)frame * left->hp.kp_cap + h[1]) * 32;
  const uint8_t* t = right->d_desc + ((size_t)frame * right->hp.kp_cap + h[3]) * 32;
  st = orbx_knn2_device(q, nq, t, nt, 0, (uint32_t*)d, (uint16_t*)(d + oDist), (void*)s);
  hipError_t e = hipSuccess;
  uint8_t* hb = ex->h_out[0];
  if (st == ORBX_OK) {
    orbx::launch_copy_words2((const uint32_t*)d, (uint32_t*)hb, 2 * nq, (const uint32_t*)(d + oDist),
                             (uint32_t*)(hb + (size_t)nq * 8), nq, s);
    e = hipGetLastError();
  }
  {
    const hipError_t es = hipStreamSynchronize(s);
    if (e == hipSuccess) e = es;
  }
  if (st != ORBX_OK) return st;
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  std::memcpy(idx, hb, (size_t)nq * 8);
  std::memcpy(dist, hb + (size_t)nq * 8, (size_t)nq * 4);
  return ORBX_OK;
}

// The same search for frames [0, n_frames) of the two last batched extractions in one launch,
// results kept on the device.  The four counts per frame (Nleft, monoLeft, Nright, monoRight) are
// copied into a buffer of the left handle in the same stream order, so that the download calls
// read what this call saw whatever happens to the right handle afterwards.
int orbx_lapping_knn2_batch_device(orbx_extractor* left, orbx_extractor* right, int n_frames) {
  if (!left || !right || left == right || !left->planned || !right->planned || left->device != right->device ||
      n_frames <= 0)
    return ORBX_ERR_BAD_ARG;
  if (n_frames > left->lastFrames || n_frames > right->lastFrames) return ORBX_ERR_BATCH;
  orbx_extractor* ex = left;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  if (const int js_ = right->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  if (!ex->evStereo) HIP_TRY(hipEventCreateWithFlags(&ex->evStereo, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ex->evStereo, right->stream));   // the right extraction runs on its own stream
  HIP_TRY(hipStreamWaitEvent(s, ex->evStereo, 0));
  const size_t slots = (size_t)ex->prm.max_batch * ex->hp.kp_cap;
  const size_t B = (size_t)ex->prm.max_batch;
  if (!ex->d_lapIdx) HIP_TRY(hipMalloc((void**)&ex->d_lapIdx, slots * 8));
  if (!ex->d_lapDist) HIP_TRY(hipMalloc((void**)&ex->d_lapDist, slots * 4));
  if (!ex->d_lapCounts) HIP_TRY(hipMalloc((void**)&ex->d_lapCounts, 4 * B * 4));
  const int* src[4] = {ex->d_n, ex->d_mono, right->d_n, right->d_mono};
  for (int k = 0; k < 4; ++k)
    HIP_TRY(hipMemcpyAsync(ex->d_lapCounts + k * B, src[k], (size_t)n_frames * 4, hipMemcpyDeviceToDevice, s));
  orbx::launch_knn2_frames(ex->d_desc, ex->hp.kp_cap, ex->d_lapCounts, ex->d_lapCounts + B, right->d_desc,
                           right->hp.kp_cap, ex->d_lapCounts + 2 * B, ex->d_lapCounts + 3 * B, ex->d_lapIdx,
                           ex->d_lapDist, n_frames, s);
  HIP_TRY(hipGetLastError());
  ex->lapFrames = n_frames;
  ex->lapMatchValid = false;
  return ORBX_OK;
}

// the four counts of one frame as the batch call saw them
static int lapping_counts(orbx_extractor* ex, int frame, int h[4]) {
  auto& last_hip_error = ex->last_hip_error;
  const size_t B = (size_t)ex->prm.max_batch;
  HIP_TRY(hipMemcpy2DAsync(h, 4, ex->d_lapCounts + frame, B * 4, 4, 4, hipMemcpyDeviceToHost, ex->stream));
  HIP_TRY(hipStreamSynchronize(ex->stream));
  return ORBX_OK;
}

int orbx_lapping_knn2_download(orbx_extractor* left, int frame, uint32_t* idx, uint16_t* dist, int cap,
                               int* n_query, int* mono_left, int* n_train, int* mono_right) {
  if (!left || !left->planned || !left->d_lapCounts || frame < 0 || frame >= left->lapFrames || cap < 0)
    return ORBX_ERR_BAD_ARG;
  orbx_extractor* ex = left;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  int h[4] = {0, 0, 0, 0};   // Nleft, monoLeft, Nright, monoRight
  if (const int cs_ = lapping_counts(ex, frame, h)) return cs_;
  const int nq = h[0] - h[1], nt = h[2] - h[3];
  if (n_query) *n_query = nq;
  if (mono_left) *mono_left = h[1];
  if (n_train) *n_train = nt;
  if (mono_right) *mono_right = h[3];
  if (nq < 0 || nt < 0) return ORBX_ERR_BAD_ARG;
  if (nq > cap) return ORBX_ERR_CAPACITY;
  if (nq == 0) return ORBX_OK;
  if (!idx || !dist) return ORBX_ERR_BAD_ARG;
  const size_t base = (size_t)frame * ex->hp.kp_cap;
  HIP_TRY(hipMemcpyAsync(idx, ex->d_lapIdx + base * 2, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(dist, ex->d_lapDist + base * 2, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return ORBX_OK;
}

// The ratio test of core/src/Frame.cc:1185 on the records the batch call left on the device.
int orbx_lapping_ratio_test_batch_device(orbx_extractor* left, double ratio) {
  if (!left || !left->planned || !left->d_lapCounts || left->lapFrames <= 0 || !(ratio >= 0.0)) return ORBX_ERR_BAD_ARG;
  orbx_extractor* ex = left;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  auto& last_hip_error = ex->last_hip_error;
  const size_t B = (size_t)ex->prm.max_batch;
  if (!ex->d_lapMatch) HIP_TRY(hipMalloc((void**)&ex->d_lapMatch, B * ex->hp.kp_cap * 4));
  orbx::launch_ratio_test_frames(ex->d_lapIdx, ex->d_lapDist, ex->hp.kp_cap, ex->d_lapCounts, ex->d_lapCounts + B,
                                 ex->lapFrames, ratio, ex->d_lapMatch, ex->stream);
  HIP_TRY(hipGetLastError());
  ex->lapMatchValid = true;
  return ORBX_OK;
}

int orbx_lapping_matches_device(orbx_extractor* left, int32_t** d_match, int* cap) {
  if (!left || !left->planned || !left->lapMatchValid || left->lapFrames <= 0) return ORBX_ERR_BAD_ARG;
  if (d_match) *d_match = left->d_lapMatch;
  if (cap) *cap = left->hp.kp_cap;
  return ORBX_OK;
}

int orbx_lapping_matches_download(orbx_extractor* left, int frame, int32_t* match, int cap, int* n_query,
                                  int* n_good) {
  if (!left || !left->planned || !left->lapMatchValid || frame < 0 || frame >= left->lapFrames || cap < 0)
    return ORBX_ERR_BAD_ARG;
  orbx_extractor* ex = left;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  auto& last_hip_error = ex->last_hip_error;
  int h[4] = {0, 0, 0, 0};
  if (const int cs_ = lapping_counts(ex, frame, h)) return cs_;
  const int nq = h[0] - h[1];
  if (n_query) *n_query = nq;
  if (n_good) *n_good = 0;
  if (nq < 0) return ORBX_ERR_BAD_ARG;
  if (nq > cap) return ORBX_ERR_CAPACITY;
  if (nq == 0) return ORBX_OK;
  if (!match) return ORBX_ERR_BAD_ARG;
  HIP_TRY(hipMemcpyAsync(match, ex->d_lapMatch + (size_t)frame * ex->hp.kp_cap, (size_t)nq * 4, hipMemcpyDeviceToHost,
                         ex->stream));
  HIP_TRY(hipStreamSynchronize(ex->stream));
  if (n_good) {
    int g = 0;
    for (int i = 0; i < nq; ++i) g += match[i] >= 0;
    *n_good = g;
  }
  return ORBX_OK;
}

// ---- frame grid on a device-resident extraction -----------------------------------------------
#define ORBX_GRID_CELLS (64 * 48)

int orbx_frame_grid_batch_device(orbx_extractor* ex, int n_frames, const orbx_pinhole* cam,
                                 const float* bounds) {
  if (!ex || !ex->planned || n_frames <= 0 || !bounds) return ORBX_ERR_BAD_ARG;
  if (!(bounds[1] > bounds[0]) || !(bounds[3] > bounds[2])) return ORBX_ERR_BAD_ARG;
  if (n_frames > ex->lastFrames) return ORBX_ERR_BATCH;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  const int cap = ex->hp.kp_cap;
  if (cap >= 0xFFFF || orbx::frame_grid_lds_bytes(cap) > 64 * 1024) return ORBX_ERR_UNSUPPORTED;
  if (!ex->d_kpsUn) {
    const size_t B = (size_t)ex->prm.max_batch;
    HIP_TRY(hipMalloc((void**)&ex->d_kpsUn, B * cap * 28));
    HIP_TRY(hipMalloc((void**)&ex->d_cellStart, B * (ORBX_GRID_CELLS + 1) * 4));
    HIP_TRY(hipMalloc((void**)&ex->d_cellEntries, B * cap * 4));
  }
  orbx_pinhole c;
  std::memset(&c, 0, sizeof(c));
  if (cam) c = *cam;
  const int distorted = cam && cam->k1 != 0.0f;      // Frame.cc:772
  if (distorted && (!(c.fx != 0.f) || !(c.fy != 0.f))) return ORBX_ERR_BAD_ARG;
  orbx::launch_frame_grid(ex->d_kps, ex->d_n, cap, c, distorted, bounds, ex->d_kpsUn, ex->d_cellStart,
                          ex->d_cellEntries, n_frames, ex->stream);
  HIP_TRY(hipGetLastError());
  ex->gridFrames = n_frames;
  std::memcpy(ex->gridBounds, bounds, sizeof(ex->gridBounds));
  return ORBX_OK;
}

int orbx_frame_grid_results_device(orbx_extractor* ex, orbx_keypoint** d_kps_un, int32_t** d_cell_start,
                                   int32_t** d_cell_entries, int* cap) {
  if (!ex || !ex->planned || !ex->d_kpsUn) return ORBX_ERR_BAD_ARG;
  if (d_kps_un) *d_kps_un = (orbx_keypoint*)ex->d_kpsUn;
  if (d_cell_start) *d_cell_start = ex->d_cellStart;
  if (d_cell_entries) *d_cell_entries = ex->d_cellEntries;
  if (cap) *cap = ex->hp.kp_cap;
  return ORBX_OK;
}

int orbx_frame_grid_download(orbx_extractor* ex, int frame, orbx_keypoint* kps_un, int32_t* cell_start,
                             int32_t* cell_entries, int cap, int* n_out) {
  if (!ex || !ex->planned || !ex->d_kpsUn || frame < 0 || frame >= ex->gridFrames) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  int n = 0;
  HIP_TRY(hipMemcpyAsync(&n, ex->d_n + frame, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n_out) *n_out = n;
  if (n > cap) return ORBX_ERR_CAPACITY;
  const size_t base = (size_t)frame * ex->hp.kp_cap;
  if (kps_un && n > 0) HIP_TRY(hipMemcpyAsync(kps_un, ex->d_kpsUn + base * 7, (size_t)n * 28, hipMemcpyDeviceToHost, s));
  if (cell_start)
    HIP_TRY(hipMemcpyAsync(cell_start, ex->d_cellStart + (size_t)frame * (ORBX_GRID_CELLS + 1),
                           (size_t)(ORBX_GRID_CELLS + 1) * 4, hipMemcpyDeviceToHost, s));
  if (cell_entries && n > 0)
    HIP_TRY(hipMemcpyAsync(cell_entries, ex->d_cellEntries + base, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return ORBX_OK;
}

int orbx_features_in_area(orbx_extractor* ex, int frame, const float* queries, int n_queries,
                          int32_t* indices, int per_query_cap, int32_t* counts) {
  if (!ex || !ex->planned || !ex->d_kpsUn || frame < 0 || frame >= ex->gridFrames || n_queries < 0 ||
      per_query_cap < 0 || (n_queries > 0 && (!queries || !counts || (per_query_cap > 0 && !indices))))
    return ORBX_ERR_BAD_ARG;
  if (n_queries == 0) return ORBX_OK;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  float* dq = nullptr;
  int *dout = nullptr, *dcnt = nullptr;
  hipError_t e = hipMallocAsync((void**)&dq, (size_t)n_queries * 20, s);
  if (e == hipSuccess) e = hipMallocAsync((void**)&dout, (size_t)n_queries * std::max(per_query_cap, 1) * 4, s);
  if (e == hipSuccess) e = hipMallocAsync((void**)&dcnt, (size_t)n_queries * 4, s);
  if (e == hipSuccess) e = hipMemcpyAsync(dq, queries, (size_t)n_queries * 20, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    orbx::launch_features_in_area(ex->d_kpsUn, ex->hp.kp_cap, ex->d_cellStart, ex->d_cellEntries, frame,
                                  ex->gridBounds, dq, n_queries, dout, per_query_cap, dcnt, s);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(counts, dcnt, (size_t)n_queries * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && per_query_cap > 0)
    e = hipMemcpyAsync(indices, dout, (size_t)n_queries * per_query_cap * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (dq) (void)hipFreeAsync(dq, s);
  if (dout) (void)hipFreeAsync(dout, s);
  if (dcnt) (void)hipFreeAsync(dcnt, s);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  return ORBX_OK;
}

// ---- guided matchers on a frame with a device-resident grid ------------------------------------
namespace {
// carves one stream-ordered allocation into 256-byte aligned pieces
struct Carver {
  size_t total = 0;
  size_t add(size_t bytes) { const size_t at = total; total += (bytes + 255) / 256 * 256; return at; }
};
const int kGuidedMaxQueries = 1 << 20;

// The host arrays of one call, packed by the CPU into one block and moved with one transfer
// (every separate copy from pageable memory costs a round trip through the driver).  The block
// must stay alive until the stream has been synchronised.
struct PackedUpload {
  std::vector<uint8_t> host;
  size_t lo = (size_t)-1, hi = 0;
  bool init(size_t bytes) noexcept {
    try { host.assign(bytes, 0); return true; } catch (...) { return false; }
  }
  void add(size_t off, const void* src, size_t bytes) {
    if (!bytes || off + bytes > host.size()) return;
    std::memcpy(host.data() + off, src, bytes);
    lo = std::min(lo, off);
    hi = std::max(hi, off + bytes);
  }
  hipError_t send(uint8_t* d, hipStream_t s) const {
    if (hi <= lo) return hipSuccess;
    return hipMemcpyAsync(d + lo, host.data() + lo, hi - lo, hipMemcpyHostToDevice, s);
  }
};

// One blocking guided-matcher call on a handle, the way the single-frame extraction is served:
// the keypoint count from the last host-pointer extraction where that is the frame in question
// (no first wait), one device block that stays with the handle, the host arrays packed into the
// handle's page-locked buffer when they fit (else into ordinary memory), the results written to
// page-locked memory by a kernel, one wait.
struct GuidedCall {
  orbx_extractor* ex;
  hipStream_t s;
  uint8_t* d = nullptr;        // device block
  uint8_t* hup = nullptr;      // host block the arrays are packed into
  bool pinnedUp = false;
  PackedUpload pageable;
  size_t lo = (size_t)-1, hi = 0;

  int count(int frame, int* n) {
    auto& last_hip_error = ex->last_hip_error;
    if (frame < ex->hostCountFrames) { *n = ex->hostN[frame]; return ORBX_OK; }
    HIP_TRY(hipMemcpyAsync(n, ex->d_n + frame, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return ORBX_OK;
  }
  int begin(size_t totalBytes, size_t uploadBytes) {
    auto& last_hip_error = ex->last_hip_error;
    if (totalBytes > ex->guidedBytes) {
      HIP_TRY(hipStreamSynchronize(s));
      if (ex->d_guided) HIP_TRY(hipFree(ex->d_guided));
      ex->d_guided = nullptr; ex->guidedBytes = 0;
      const size_t want = totalBytes + totalBytes / 4;
      HIP_TRY(hipMalloc((void**)&ex->d_guided, want));
      ex->guidedBytes = want;
    }
    d = ex->d_guided;
    const int st = ex->ensureHostStaging(true, true);
    if (st != ORBX_OK) return st;
    if (uploadBytes <= (size_t)ex->groupFrames * ex->hp.W * ex->hp.H) {
      hup = ex->h_in[0];
      pinnedUp = true;
    } else {
      if (!pageable.init(uploadBytes)) return ORBX_ERR_UNSUPPORTED;
      hup = pageable.host.data();
    }
    return ORBX_OK;
  }
  void add(size_t off, const void* src, size_t bytes) {
    if (!bytes) return;
    std::memcpy(hup + off, src, bytes);
    lo = std::min(lo, off);
    hi = std::max(hi, off + bytes);
  }
  hipError_t send() const {
    if (hi <= lo) return hipSuccess;
    return hipMemcpyAsync(d + lo, hup + lo, hi - lo, hipMemcpyHostToDevice, s);
  }
  // assigned[0, nn) and the two summary words to the host; waits for the stream (also when e
  // already holds an error: the packed block must outlive whatever was enqueued)
  hipError_t finish(hipError_t e, size_t oAsg, size_t nn, size_t oSum, int32_t* assigned, int* sum) {
    uint8_t* hres = ex->h_out[0];
    if (e == hipSuccess) {
      orbx::launch_copy_words2((const uint32_t*)(d + oAsg), (uint32_t*)hres, (int)nn, (const uint32_t*)(d + oSum),
                               (uint32_t*)(hres + nn * 4), 2, s);
      e = hipGetLastError();
    }
    const hipError_t es = hipStreamSynchronize(s);
    if (e == hipSuccess) e = es;
    if (e == hipSuccess) {
      std::memcpy(assigned, hres, nn * 4);
      std::memcpy(sum, hres + nn * 4, 8);
    }
    return e;
  }
};
}  // namespace

namespace {
// the last-frame matcher and the keyframe matcher: one loop, the second with a caller-given
// distance limit, every keypoint with a map point closed, every assignment closing
int search_by_projection(orbx_extractor* cur, int frame, int n_last, const uint8_t* last_valid, const float* uv,
                         const float* invz, const int32_t* last_octave, const float* last_angle,
                         const uint8_t* last_desc, const uint8_t* last_obs, const float* u_right,
                         const uint8_t* state, float th, int forward, int backward, float mbf,
                         int check_orientation, int max_dist, int32_t* assigned, int cap, int* n_out, int* nmatches,
                         int* rounds) {
  orbx_extractor* ex = cur;
  if (!ex || !ex->planned || !ex->d_kpsUn || frame < 0 || frame >= ex->gridFrames || n_last < 0 ||
      n_last > kGuidedMaxQueries || cap < 0 || !nmatches)
    return ORBX_ERR_BAD_ARG;
  if (n_last > 0 && (!last_valid || !uv || !invz || !last_octave || !last_angle || !last_desc || !last_obs))
    return ORBX_ERR_BAD_ARG;
  for (int i = 0; i < n_last; ++i)
    if (last_valid[i] && (last_octave[i] < 0 || last_octave[i] >= ex->prm.nlevels)) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  GuidedCall call{ex, s};
  int n = 0;
  {
    const int st = call.count(frame, &n);
    if (st != ORBX_OK) return st;
  }
  if (n_out) *n_out = n;
  *nmatches = 0;
  if (rounds) *rounds = 0;
  if (n > cap) return ORBX_ERR_CAPACITY;
  if (n > 0 && !assigned) return ORBX_ERR_BAD_ARG;
  if (n == 0) return ORBX_OK;
  if (n_last == 0) {
    for (int i = 0; i < n; ++i) assigned[i] = -1;
    return ORBX_OK;
  }
  const size_t nl = (size_t)n_last, nn = (size_t)n;
  Carver c;
  const size_t oValid = c.add(nl), oUv = c.add(nl * 8), oInvz = c.add(nl * 4), oOct = c.add(nl * 4),
               oAng = c.add(nl * 4), oDesc = c.add(nl * 32), oObs = c.add(nl), oUr = c.add(nn * 4),
               oState = c.add(nn), oWork = c.add(orbx::projection_work_ints(n_last, n) * 4),
               oAsg = c.add(nn * 4), oSum = c.add(8);
  {
    const int st = call.begin(c.total, oWork);
    if (st != ORBX_OK) return st;
  }
  uint8_t* d = call.d;
  auto up = [&](size_t off, const void* src, size_t bytes) { call.add(off, src, bytes); };
  up(oValid, last_valid, nl); up(oUv, uv, nl * 8); up(oInvz, invz, nl * 4); up(oOct, last_octave, nl * 4);
  up(oAng, last_angle, nl * 4); up(oDesc, last_desc, nl * 32); up(oObs, last_obs, nl);
  if (u_right) up(oUr, u_right, nn * 4);
  if (state) up(oState, state, nn);
  hipError_t e = call.send();
  int sum[2] = {0, 0};
  if (e == hipSuccess) {
    const size_t base = (size_t)frame * ex->hp.kp_cap;
    orbx::launch_search_by_projection_last(
        d + oValid, (const float*)(d + oUv), (const float*)(d + oInvz), (const int*)(d + oOct),
        (const float*)(d + oAng), d + oDesc, d + oObs, n_last, ex->d_kpsUn + base * 7, ex->d_desc + base * 32,
        u_right ? (const float*)(d + oUr) : nullptr, state ? d + oState : nullptr, n,
        ex->d_cellStart + (size_t)frame * (ORBX_GRID_CELLS + 1), ex->d_cellEntries + base, ex->gridBounds,
        ex->scale.data(), ex->prm.nlevels, th, forward != 0, backward != 0, mbf, check_orientation != 0, max_dist,
        (int*)(d + oWork), (int*)(d + oAsg), (int*)(d + oSum), s);
    e = hipGetLastError();
  }
  e = call.finish(e, oAsg, nn, oSum, assigned, sum);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  *nmatches = sum[0];
  if (rounds) *rounds = sum[1];
  return ORBX_OK;
}
}  // namespace

int orbx_search_by_projection_last(orbx_extractor* cur, int frame, int n_last, const uint8_t* last_valid,
                                   const float* uv, const float* invz, const int32_t* last_octave,
                                   const float* last_angle, const uint8_t* last_desc, const uint8_t* last_obs,
                                   const float* u_right, const uint8_t* state, float th, int forward,
                                   int backward, float mbf, int check_orientation, int32_t* assigned, int cap,
                                   int* n_out, int* nmatches, int* rounds) {
  return search_by_projection(cur, frame, n_last, last_valid, uv, invz, last_octave, last_angle, last_desc, last_obs,
                              u_right, state, th, forward, backward, mbf, check_orientation, 100 /* TH_HIGH */,
                              assigned, cap, n_out, nmatches, rounds);
}

int orbx_search_by_projection_keyframe(orbx_extractor* cur, int frame, int n_kf, const uint8_t* kf_valid,
                                       const float* uv, const int32_t* level, const float* kf_angle,
                                       const uint8_t* kf_desc, const uint8_t* has_point, float th, int orb_dist,
                                       int check_orientation, int32_t* assigned, int cap, int* n_out,
                                       int* nmatches, int* rounds) {
  if (n_kf < 0 || n_kf > kGuidedMaxQueries || orb_dist < 0 || orb_dist > 255) return ORBX_ERR_BAD_ARG;
  try {
    // in the terms of the shared loop: nothing is behind the camera, every map point has
    // observations (an assignment closes the keypoint, matcher.cc:1872), and a keypoint that
    // holds any map point is closed from the start
    const std::vector<float> invz((size_t)n_kf, 1.0f);
    const std::vector<uint8_t> obs((size_t)n_kf, 1);
    return search_by_projection(cur, frame, n_kf, kf_valid, uv, invz.data(), level, kf_angle, kf_desc, obs.data(),
                                nullptr, has_point, th, 0, 0, 0.0f, check_orientation, orb_dist, assigned, cap,
                                n_out, nmatches, rounds);
  } catch (...) {
    return ORBX_ERR_UNSUPPORTED;
  }
}

// the two handles of a rig frame: same device, both with a grid of the same bounds for `frame`
static int rig_check(const orbx_extractor* l, const orbx_extractor* r, int frame) {
  if (!l || !r || l == r || !l->planned || !r->planned || !l->d_kpsUn || !r->d_kpsUn || l->device != r->device)
    return ORBX_ERR_BAD_ARG;
  if (frame < 0 || frame >= l->gridFrames || frame >= r->gridFrames) return ORBX_ERR_BAD_ARG;
  if (l->prm.nlevels != r->prm.nlevels || l->prm.scale_factor != r->prm.scale_factor) return ORBX_ERR_BAD_ARG;
  for (int k = 0; k < 4; ++k)
    if (l->gridBounds[k] != r->gridBounds[k]) return ORBX_ERR_BAD_ARG;   // one Frame, one set of image bounds
  return ORBX_OK;
}

int orbx_search_by_projection_last_rig(orbx_extractor* left, orbx_extractor* right, int frame, int n_last,
                                       const uint8_t* last_valid, const float* uv_left, const float* invz,
                                       const float* uv_right, const int32_t* last_octave, const float* last_angle,
                                       const uint8_t* last_desc, const uint8_t* last_obs, const uint8_t* state_left,
                                       const uint8_t* state_right, float th, int forward, int backward,
                                       int check_orientation, int32_t* assigned_left, int cap_left, int* n_left,
                                       int32_t* assigned_right, int cap_right, int* n_right, int* nmatches,
                                       int* rounds) {
  if (const int rc_ = rig_check(left, right, frame)) return rc_;
  if (n_last < 0 || n_last > kGuidedMaxQueries || cap_left < 0 || cap_right < 0 || !nmatches) return ORBX_ERR_BAD_ARG;
  if (n_last > 0 && (!last_valid || !uv_left || !uv_right || !invz || !last_octave || !last_angle || !last_desc || !last_obs))
    return ORBX_ERR_BAD_ARG;
  orbx_extractor* ex = left;
  for (int i = 0; i < n_last; ++i)
    if (last_valid[i] && (last_octave[i] < 0 || last_octave[i] >= ex->prm.nlevels)) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  if (const int js_ = right->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  if (!ex->evStereo) HIP_TRY(hipEventCreateWithFlags(&ex->evStereo, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ex->evStereo, right->stream));   // the right camera's extraction and grid ran on its own stream
  HIP_TRY(hipStreamWaitEvent(s, ex->evStereo, 0));
  int nn[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(&nn[0], ex->d_n + frame, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&nn[1], right->d_n + frame, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const int nL = nn[0], nR = nn[1];
  if (n_left) *n_left = nL;
  if (n_right) *n_right = nR;
  *nmatches = 0;
  if (rounds) *rounds = 0;
  if (nL > cap_left || nR > cap_right) return ORBX_ERR_CAPACITY;
  if ((nL > 0 && !assigned_left) || (nR > 0 && !assigned_right)) return ORBX_ERR_BAD_ARG;
  for (int i = 0; i < nL; ++i) assigned_left[i] = -1;
  for (int i = 0; i < nR; ++i) assigned_right[i] = -1;
  if (n_last == 0 || nL == 0) return ORBX_OK;   // nothing passes the left camera's search: the right one is never reached
  const size_t nl = (size_t)n_last, nnl = (size_t)nL, nnr = (size_t)std::max(nR, 1);
  Carver c;
  const size_t oValid = c.add(nl), oUvL = c.add(nl * 8), oUvR = c.add(nl * 8), oInvz = c.add(nl * 4),
               oOct = c.add(nl * 4), oAng = c.add(nl * 4), oDesc = c.add(nl * 32), oObs = c.add(nl),
               oStL = c.add(nnl), oStR = c.add(nnr),
               oWorkL = c.add((orbx::projection_work_ints(n_last, nL) + nl) * 4),
               oWorkR = c.add((orbx::projection_work_ints(n_last, (int)nnr) + nl) * 4), oAsgL = c.add(nnl * 4),
               oAsgR = c.add(nnr * 4), oSum = c.add(24);
  uint8_t* d = nullptr;
  hipError_t e = hipMallocAsync((void**)&d, c.total, s);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  PackedUpload pk;
  if (!pk.init(oWorkL)) { (void)hipFreeAsync(d, s); return ORBX_ERR_UNSUPPORTED; }
  auto up = [&](size_t off, const void* src, size_t bytes) { pk.add(off, src, bytes); };
  up(oValid, last_valid, nl); up(oUvL, uv_left, nl * 8); up(oUvR, uv_right, nl * 8); up(oInvz, invz, nl * 4);
  up(oOct, last_octave, nl * 4); up(oAng, last_angle, nl * 4); up(oDesc, last_desc, nl * 32); up(oObs, last_obs, nl);
  if (state_left) up(oStL, state_left, nnl);
  if (state_right && nR > 0) up(oStR, state_right, (size_t)nR);
  e = pk.send(d, s);
  int sum[2] = {0, 0};
  if (e == hipSuccess) {
    const size_t bL = (size_t)frame * ex->hp.kp_cap, bR = (size_t)frame * right->hp.kp_cap;
    orbx::launch_search_by_projection_last_rig(
        d + oValid, (const float*)(d + oUvL), (const float*)(d + oInvz), (const float*)(d + oUvR),
        (const int*)(d + oOct), (const float*)(d + oAng), d + oDesc, d + oObs, n_last, ex->d_kpsUn + bL * 7,
        ex->d_desc + bL * 32, state_left ? d + oStL : nullptr, nL,
        ex->d_cellStart + (size_t)frame * (ORBX_GRID_CELLS + 1), ex->d_cellEntries + bL, right->d_kpsUn + bR * 7,
        right->d_desc + bR * 32, (state_right && nR > 0) ? d + oStR : nullptr, nR,
        right->d_cellStart + (size_t)frame * (ORBX_GRID_CELLS + 1), right->d_cellEntries + bR, ex->gridBounds,
        ex->scale.data(), ex->prm.nlevels, th, forward != 0, backward != 0, check_orientation != 0,
        (int*)(d + oWorkL), (int*)(d + oWorkR), (int*)(d + oAsgL), (int*)(d + oAsgR), (int*)(d + oSum),
        (int*)(d + oSum + 8), (int*)(d + oSum + 16), s);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(assigned_left, d + oAsgL, nnl * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && nR > 0) e = hipMemcpyAsync(assigned_right, d + oAsgR, (size_t)nR * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(sum, d + oSum + 16, 8, hipMemcpyDeviceToHost, s);
  {   // also after an error: the packed host block must outlive whatever was enqueued
    const hipError_t es = hipStreamSynchronize(s);
    if (e == hipSuccess) e = es;
  }
  (void)hipFreeAsync(d, s);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  *nmatches = sum[0];
  if (rounds) *rounds = sum[1];
  return ORBX_OK;
}

int orbx_search_by_projection_points(orbx_extractor* f, int frame, int n_points, const uint8_t* in_view,
                                     const float* proj, const float* proj_xr, const int32_t* level,
                                     const float* view_cos, const uint8_t* desc, const uint8_t* obs,
                                     const float* u_right, const uint8_t* state, float th, float nnratio,
                                     int32_t* assigned, int cap, int* n_out, int* nmatches, int* rounds) {
  orbx_extractor* ex = f;
  if (!ex || !ex->planned || !ex->d_kpsUn || frame < 0 || frame >= ex->gridFrames || n_points < 0 ||
      n_points > kGuidedMaxQueries || cap < 0 || !nmatches)
    return ORBX_ERR_BAD_ARG;
  if (n_points > 0 && (!in_view || !proj || !proj_xr || !level || !view_cos || !desc || !obs)) return ORBX_ERR_BAD_ARG;
  for (int i = 0; i < n_points; ++i)
    if (in_view[i] && (level[i] < 0 || level[i] >= ex->prm.nlevels)) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  GuidedCall call{ex, s};
  int n = 0;
  {
    const int st = call.count(frame, &n);
    if (st != ORBX_OK) return st;
  }
  if (n_out) *n_out = n;
  *nmatches = 0;
  if (rounds) *rounds = 0;
  if (n > cap) return ORBX_ERR_CAPACITY;
  if (n > 0 && !assigned) return ORBX_ERR_BAD_ARG;
  if (n == 0) return ORBX_OK;
  if (n_points == 0) {
    for (int i = 0; i < n; ++i) assigned[i] = -1;
    return ORBX_OK;
  }
  const size_t np = (size_t)n_points, nn = (size_t)n;
  Carver c;
  const size_t oView = c.add(np), oProj = c.add(np * 8), oXr = c.add(np * 4), oLvl = c.add(np * 4),
               oCos = c.add(np * 4), oDesc = c.add(np * 32), oObs = c.add(np), oUr = c.add(nn * 4),
               oState = c.add(nn), oWork = c.add(orbx::projection_work_ints(n_points, n) * 4),
               oAsg = c.add(nn * 4), oSum = c.add(8);
  {
    const int st = call.begin(c.total, oWork);
    if (st != ORBX_OK) return st;
  }
  uint8_t* d = call.d;
  auto up = [&](size_t off, const void* src, size_t bytes) { call.add(off, src, bytes); };
  up(oView, in_view, np); up(oProj, proj, np * 8); up(oXr, proj_xr, np * 4); up(oLvl, level, np * 4);
  up(oCos, view_cos, np * 4); up(oDesc, desc, np * 32); up(oObs, obs, np);
  if (u_right) up(oUr, u_right, nn * 4);
  if (state) up(oState, state, nn);
  hipError_t e = call.send();
  int sum[2] = {0, 0};
  if (e == hipSuccess) {
    const size_t base = (size_t)frame * ex->hp.kp_cap;
    orbx::launch_search_by_projection_points(
        d + oView, (const float*)(d + oProj), (const float*)(d + oXr), (const int*)(d + oLvl),
        (const float*)(d + oCos), d + oDesc, d + oObs, n_points, ex->d_kpsUn + base * 7, ex->d_desc + base * 32,
        u_right ? (const float*)(d + oUr) : nullptr, state ? d + oState : nullptr, n,
        ex->d_cellStart + (size_t)frame * (ORBX_GRID_CELLS + 1), ex->d_cellEntries + base, ex->gridBounds,
        ex->scale.data(), ex->prm.nlevels, th, nnratio, (int*)(d + oWork), (int*)(d + oAsg), (int*)(d + oSum), s);
    e = hipGetLastError();
  }
  e = call.finish(e, oAsg, nn, oSum, assigned, sum);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  *nmatches = sum[0];
  if (rounds) *rounds = sum[1];
  return ORBX_OK;
}

int orbx_search_by_projection_points_rig(orbx_extractor* left, orbx_extractor* right, int frame, int n_points,
                                         const uint8_t* in_view, const float* proj, const int32_t* level,
                                         const float* view_cos, const uint8_t* in_view_right, const float* proj_right,
                                         const int32_t* level_right, const float* view_cos_right, const uint8_t* desc,
                                         const uint8_t* obs, const int32_t* left_to_right,
                                         const int32_t* right_to_left, const uint8_t* state_left,
                                         const uint8_t* state_right, float th, float nnratio, int32_t* assigned_left,
                                         int cap_left, int* n_left, int32_t* assigned_right, int cap_right,
                                         int* n_right, int* nmatches, int* rounds) {
  if (const int rc_ = rig_check(left, right, frame)) return rc_;
  if (n_points < 0 || n_points > kGuidedMaxQueries / 2 || cap_left < 0 || cap_right < 0 || !nmatches)
    return ORBX_ERR_BAD_ARG;
  if (n_points > 0 && (!in_view || !proj || !level || !view_cos || !in_view_right || !proj_right || !level_right ||
                       !view_cos_right || !desc || !obs))
    return ORBX_ERR_BAD_ARG;
  orbx_extractor* ex = left;
  // the right camera's search needs a predicted level there (:146); fold that into its in-view flag
  std::vector<uint8_t> viewR((size_t)std::max(n_points, 1));
  for (int i = 0; i < n_points; ++i) {
    if (in_view[i] && (level[i] < 0 || level[i] >= ex->prm.nlevels)) return ORBX_ERR_BAD_ARG;
    viewR[i] = in_view_right[i] && level_right[i] != -1;
    if (viewR[i] && (level_right[i] < 0 || level_right[i] >= ex->prm.nlevels)) return ORBX_ERR_BAD_ARG;
  }
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  if (const int js_ = right->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  if (!ex->evStereo) HIP_TRY(hipEventCreateWithFlags(&ex->evStereo, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ex->evStereo, right->stream));
  HIP_TRY(hipStreamWaitEvent(s, ex->evStereo, 0));
  int nn[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(&nn[0], ex->d_n + frame, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&nn[1], right->d_n + frame, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const int nL = nn[0], nR = nn[1];
  if (n_left) *n_left = nL;
  if (n_right) *n_right = nR;
  *nmatches = 0;
  if (rounds) *rounds = 0;
  if (nL > cap_left || nR > cap_right) return ORBX_ERR_CAPACITY;
  if ((nL > 0 && (!assigned_left || !left_to_right)) || (nR > 0 && (!assigned_right || !right_to_left)))
    return ORBX_ERR_BAD_ARG;
  for (int i = 0; i < nL; ++i)
    if (left_to_right[i] < -1 || left_to_right[i] >= nR) return ORBX_ERR_BAD_ARG;
  for (int i = 0; i < nR; ++i)
    if (right_to_left[i] < -1 || right_to_left[i] >= nL) return ORBX_ERR_BAD_ARG;
  for (int i = 0; i < nL; ++i) assigned_left[i] = -1;
  for (int i = 0; i < nR; ++i) assigned_right[i] = -1;
  if (n_points == 0 || (nL == 0 && nR == 0)) return ORBX_OK;
  const size_t np = (size_t)n_points, nnl = (size_t)std::max(nL, 1), nnr = (size_t)std::max(nR, 1);
  Carver c;
  const size_t oView = c.add(np), oProj = c.add(np * 8), oLvl = c.add(np * 4), oCos = c.add(np * 4),
               oViewR = c.add(np), oProjR = c.add(np * 8), oLvlR = c.add(np * 4), oCosR = c.add(np * 4),
               oDesc = c.add(np * 32), oObs = c.add(np), oL2R = c.add(nnl * 4), oR2L = c.add(nnr * 4),
               oStL = c.add(nnl), oStR = c.add(nnr),
               oWork = c.add(orbx::points_rig_work_ints(n_points, nL, nR) * 4), oAsgL = c.add(nnl * 4),
               oAsgR = c.add(nnr * 4), oSum = c.add(8);
  uint8_t* d = nullptr;
  hipError_t e = hipMallocAsync((void**)&d, c.total, s);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  PackedUpload pk;
  if (!pk.init(oWork)) { (void)hipFreeAsync(d, s); return ORBX_ERR_UNSUPPORTED; }
  auto up = [&](size_t off, const void* src, size_t bytes) { pk.add(off, src, bytes); };
  up(oView, in_view, np); up(oProj, proj, np * 8); up(oLvl, level, np * 4); up(oCos, view_cos, np * 4);
  up(oViewR, viewR.data(), np); up(oProjR, proj_right, np * 8); up(oLvlR, level_right, np * 4);
  up(oCosR, view_cos_right, np * 4); up(oDesc, desc, np * 32); up(oObs, obs, np);
  if (nL > 0) up(oL2R, left_to_right, (size_t)nL * 4);
  if (nR > 0) up(oR2L, right_to_left, (size_t)nR * 4);
  if (state_left && nL > 0) up(oStL, state_left, (size_t)nL);
  if (state_right && nR > 0) up(oStR, state_right, (size_t)nR);
  e = pk.send(d, s);
  int sum[2] = {0, 0};
  if (e == hipSuccess) {
    const size_t bL = (size_t)frame * ex->hp.kp_cap, bR = (size_t)frame * right->hp.kp_cap;
    orbx::launch_search_by_projection_points_rig(
        d + oView, (const float*)(d + oProj), (const int*)(d + oLvl), (const float*)(d + oCos), d + oViewR,
        (const float*)(d + oProjR), (const int*)(d + oLvlR), (const float*)(d + oCosR), d + oDesc, d + oObs, n_points,
        ex->d_kpsUn + bL * 7, ex->d_desc + bL * 32, (state_left && nL > 0) ? d + oStL : nullptr, nL,
        ex->d_cellStart + (size_t)frame * (ORBX_GRID_CELLS + 1), ex->d_cellEntries + bL, right->d_kpsUn + bR * 7,
        right->d_desc + bR * 32, (state_right && nR > 0) ? d + oStR : nullptr, nR,
        right->d_cellStart + (size_t)frame * (ORBX_GRID_CELLS + 1), right->d_cellEntries + bR,
        (const int*)(d + oL2R), (const int*)(d + oR2L), ex->gridBounds, ex->scale.data(), ex->prm.nlevels, th,
        nnratio, (int*)(d + oWork), (int*)(d + oAsgL), (int*)(d + oAsgR), (int*)(d + oSum), s);
    e = hipGetLastError();
  }
  if (e == hipSuccess && nL > 0) e = hipMemcpyAsync(assigned_left, d + oAsgL, (size_t)nL * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && nR > 0) e = hipMemcpyAsync(assigned_right, d + oAsgR, (size_t)nR * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(sum, d + oSum, 8, hipMemcpyDeviceToHost, s);
  {   // also after an error: the packed host block must outlive whatever was enqueued
    const hipError_t es = hipStreamSynchronize(s);
    if (e == hipSuccess) e = es;
  }
  (void)hipFreeAsync(d, s);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  *nmatches = sum[0];
  if (rounds) *rounds = sum[1];
  return ORBX_OK;
}

int orbx_search_for_initialization(orbx_extractor* f2, int frame2, int n1, const orbx_keypoint* kps1_un,
                                   const uint8_t* desc1, float* prev_matched, int window_size, float nnratio,
                                   int check_orientation, int32_t* matches12, int* nmatches, int* rounds) {
  orbx_extractor* ex = f2;
  if (!ex || !ex->planned || !ex->d_kpsUn || frame2 < 0 || frame2 >= ex->gridFrames || n1 < 0 ||
      n1 > kGuidedMaxQueries || !nmatches)
    return ORBX_ERR_BAD_ARG;
  if (n1 > 0 && (!kps1_un || !desc1 || !prev_matched || !matches12)) return ORBX_ERR_BAD_ARG;
  *nmatches = 0;
  if (rounds) *rounds = 0;
  if (n1 == 0) return ORBX_OK;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  int n2 = 0;
  HIP_TRY(hipMemcpyAsync(&n2, ex->d_n + frame2, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const size_t a1 = (size_t)n1, a2 = (size_t)std::max(n2, 1);
  Carver c;
  const size_t oK1 = c.add(a1 * 28), oD1 = c.add(a1 * 32), oPm = c.add(a1 * 8),
               oWork = c.add(orbx::init_work_ints(n1, (int)a2) * 4), oM12 = c.add(a1 * 4), oSum = c.add(8);
  uint8_t* d = nullptr;
  hipError_t e = hipMallocAsync((void**)&d, c.total, s);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  PackedUpload pk;
  if (!pk.init(oWork)) { (void)hipFreeAsync(d, s); return ORBX_ERR_UNSUPPORTED; }
  pk.add(oK1, kps1_un, a1 * 28); pk.add(oD1, desc1, a1 * 32); pk.add(oPm, prev_matched, a1 * 8);
  e = pk.send(d, s);
  int sum[2] = {0, 0};
  if (e == hipSuccess) {
    const size_t base = (size_t)frame2 * ex->hp.kp_cap;
    orbx::launch_search_for_initialization(
        (const float*)(d + oK1), d + oD1, n1, (float*)(d + oPm), ex->d_kpsUn + base * 7, ex->d_desc + base * 32, n2,
        ex->d_cellStart + (size_t)frame2 * (ORBX_GRID_CELLS + 1), ex->d_cellEntries + base, ex->gridBounds,
        (float)window_size, nnratio, check_orientation != 0, (int*)(d + oWork), (int*)(d + oM12),
        (int*)(d + oSum), s);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(matches12, d + oM12, a1 * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(prev_matched, d + oPm, a1 * 8, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(sum, d + oSum, 8, hipMemcpyDeviceToHost, s);
  {   // also after an error: the packed host block must outlive whatever was enqueued
    const hipError_t es = hipStreamSynchronize(s);
    if (e == hipSuccess) e = es;
  }
  (void)hipFreeAsync(d, s);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  *nmatches = sum[0];
  if (rounds) *rounds = sum[1];
  return ORBX_OK;
}

// ---- bag of words on a device-resident extraction ----------------------------------------------
int orbx_bow_transform_batch_device(orbx_extractor* ex, const orbx_vocabulary* voc, int n_frames, int levelsup) {
  if (!ex || !ex->planned || !voc || n_frames <= 0 || levelsup < 0) return ORBX_ERR_BAD_ARG;
  if (n_frames > ex->lastFrames) return ORBX_ERR_BATCH;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  const orbx::VocDevice* vd = orbx::voc_device(voc, ex->device);
  if (!vd) return ORBX_ERR_BAD_ARG;   // vocabulary lives on another device
  auto& last_hip_error = ex->last_hip_error;
  const int cap = ex->hp.kp_cap;
  if (!ex->d_bowBlock) {
    const size_t B = (size_t)ex->prm.max_batch;
    int P = 2;
    while (P < cap) P <<= 1;
    const bool keysInLds = orbx::bow_keys_lds_bytes(P) <= 60 * 1024;
    Carver c;
    const size_t oFW = c.add(B * cap * 4), oFWt = c.add(B * cap * 8), oFN = c.add(B * cap * 4),
                 oKeys = c.add(keysInLds ? 0 : B * P * 8), oBW = c.add(B * cap * 4), oBV = c.add(B * cap * 8),
                 oBC = c.add(B * 4), oVN = c.add(B * cap * 4), oVS = c.add(B * ((size_t)cap + 1) * 4),
                 oVF = c.add(B * cap * 4), oVC = c.add(B * 4);
    uint8_t* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, c.total));
    ex->d_bowBlock = d;
    orbx::BowBuffers& b = ex->bow;
    b.featWord = (uint32_t*)(d + oFW); b.featWeight = (double*)(d + oFWt); b.featNode = (uint32_t*)(d + oFN);
    b.keyWs = keysInLds ? nullptr : (uint64_t*)(d + oKeys); b.P = P;
    b.bowWord = (uint32_t*)(d + oBW); b.bowVal = (double*)(d + oBV); b.bowCount = (int*)(d + oBC);
    b.fvNode = (uint32_t*)(d + oVN); b.fvStart = (int*)(d + oVS); b.fvFeat = (uint32_t*)(d + oVF);
    b.fvCount = (int*)(d + oVC);
  }
  orbx::launch_bow(*vd, ex->d_desc, ex->d_n, cap, n_frames, levelsup, ex->bow, ex->stream);
  HIP_TRY(hipGetLastError());
  ex->bowFrames = n_frames;
  return ORBX_OK;
}

int orbx_bow_download(orbx_extractor* ex, int frame, uint32_t* word_ids, double* word_values, int* n_words,
                      uint32_t* node_ids, int32_t* node_start, uint32_t* features, int* n_nodes, int cap) {
  if (!ex || !ex->planned || !ex->d_bowBlock || frame < 0 || frame >= ex->bowFrames || cap < 0)
    return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  const orbx::BowBuffers& b = ex->bow;
  int cnt[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(&cnt[0], b.bowCount + frame, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&cnt[1], b.fvCount + frame, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n_words) *n_words = cnt[0];
  if (n_nodes) *n_nodes = cnt[1];
  if (cnt[0] > cap || cnt[1] > cap) return ORBX_ERR_CAPACITY;
  const size_t fo = (size_t)frame * ex->hp.kp_cap;
  int m = 0;   // features that entered the vectors = end of the last node's range
  HIP_TRY(hipMemcpyAsync(&m, b.fvStart + (size_t)frame * (ex->hp.kp_cap + 1) + cnt[1], 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (m > cap) return ORBX_ERR_CAPACITY;
  if (word_ids && cnt[0]) HIP_TRY(hipMemcpyAsync(word_ids, b.bowWord + fo, (size_t)cnt[0] * 4, hipMemcpyDeviceToHost, s));
  if (word_values && cnt[0]) HIP_TRY(hipMemcpyAsync(word_values, b.bowVal + fo, (size_t)cnt[0] * 8, hipMemcpyDeviceToHost, s));
  if (node_ids && cnt[1]) HIP_TRY(hipMemcpyAsync(node_ids, b.fvNode + fo, (size_t)cnt[1] * 4, hipMemcpyDeviceToHost, s));
  if (node_start)
    HIP_TRY(hipMemcpyAsync(node_start, b.fvStart + (size_t)frame * (ex->hp.kp_cap + 1), (size_t)(cnt[1] + 1) * 4,
                           hipMemcpyDeviceToHost, s));
  if (features && m) HIP_TRY(hipMemcpyAsync(features, b.fvFeat + fo, (size_t)m * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return ORBX_OK;
}

int orbx_search_by_bow(orbx_extractor* f, int frame, int n_kf, const uint8_t* kf_valid, const float* kf_angle,
                       const uint8_t* kf_desc, int n_kf_nodes, const uint32_t* kf_node_ids,
                       const int32_t* kf_node_start, const uint32_t* kf_features, float nnratio,
                       int check_orientation, int32_t* matches, int cap, int* n_out, int* nmatches) {
  orbx_extractor* ex = f;
  if (!ex || !ex->planned || !ex->d_bowBlock || frame < 0 || frame >= ex->bowFrames || n_kf < 0 ||
      n_kf_nodes < 0 || cap < 0 || !nmatches)
    return ORBX_ERR_BAD_ARG;
  if (n_kf > 0 && (!kf_valid || !kf_angle || !kf_desc)) return ORBX_ERR_BAD_ARG;
  if (n_kf_nodes > 0 && (!kf_node_ids || !kf_node_start || !kf_features)) return ORBX_ERR_BAD_ARG;
  // the keyframe's feature vector must be a CSR in ascending node id over keypoints [0, n_kf)
  int nFeat = 0;
  if (n_kf_nodes > 0) {
    if (kf_node_start[0] != 0) return ORBX_ERR_BAD_ARG;
    for (int r = 0; r < n_kf_nodes; ++r) {
      if (kf_node_start[r + 1] < kf_node_start[r]) return ORBX_ERR_BAD_ARG;
      if (r > 0 && kf_node_ids[r] <= kf_node_ids[r - 1]) return ORBX_ERR_BAD_ARG;
    }
    nFeat = kf_node_start[n_kf_nodes];
    for (int i = 0; i < nFeat; ++i)
      if (kf_features[i] >= (uint32_t)n_kf) return ORBX_ERR_BAD_ARG;
  }
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  hipStream_t s = ex->stream;
  GuidedCall call{ex, s};
  int n = 0;
  {
    const int st = call.count(frame, &n);
    if (st != ORBX_OK) return st;
  }
  if (n_out) *n_out = n;
  *nmatches = 0;
  if (n > cap) return ORBX_ERR_CAPACITY;
  if (n > 0 && !matches) return ORBX_ERR_BAD_ARG;
  if (n == 0) return ORBX_OK;
  const size_t nk = (size_t)std::max(n_kf, 1), nn = (size_t)std::max(n_kf_nodes, 1), nf = (size_t)std::max(nFeat, 1);
  const int kc = ex->hp.kp_cap;
  Carver c;
  const size_t oValid = c.add(nk), oAng = c.add(nk * 4), oDesc = c.add(nk * 32), oNode = c.add(nn * 4),
               oStart = c.add((nn + 1) * 4), oFeat = c.add(nf * 4), oM = c.add((size_t)kc * 4), oSum = c.add(8);
  {
    const int st = call.begin(c.total, oM);
    if (st != ORBX_OK) return st;
  }
  uint8_t* d = call.d;
  auto up = [&](size_t off, const void* src, size_t bytes) { call.add(off, src, bytes); };
  up(oValid, kf_valid, (size_t)n_kf); up(oAng, kf_angle, (size_t)n_kf * 4); up(oDesc, kf_desc, (size_t)n_kf * 32);
  up(oNode, kf_node_ids, (size_t)n_kf_nodes * 4);
  if (n_kf_nodes > 0) up(oStart, kf_node_start, ((size_t)n_kf_nodes + 1) * 4);
  up(oFeat, kf_features, (size_t)nFeat * 4);
  hipError_t e = call.send();
  int sum[2] = {0, 0};
  if (e == hipSuccess) {
    const size_t base = (size_t)frame * kc;
    const orbx::BowBuffers& b = ex->bow;
    orbx::launch_search_by_bow(d + oValid, (const float*)(d + oAng), d + oDesc, n_kf_nodes,
                               (const uint32_t*)(d + oNode), (const int*)(d + oStart), (const uint32_t*)(d + oFeat),
                               ex->d_kps + base * 7, ex->d_desc + base * 32, ex->d_n + frame, kc, b.fvNode + base,
                               b.fvStart + (size_t)frame * (kc + 1), b.fvFeat + base, b.fvCount + frame, nnratio,
                               check_orientation != 0, (int*)(d + oM), (int*)(d + oSum), s);
    e = hipGetLastError();
  }
  e = call.finish(e, oM, (size_t)n, oSum, matches, sum);
  if (e != hipSuccess) { last_hip_error = e; return ORBX_ERR_HIP; }
  *nmatches = sum[0];
  return ORBX_OK;
}

int orbx_level_dims(orbx_extractor* ex, int level, int* width, int* height, int* stride) {
  if (!ex || !ex->planned || level < 0 || level >= ex->hp.nlevels) return ORBX_ERR_BAD_ARG;
  if (width) *width = ex->hp.lv[level].w;
  if (height) *height = ex->hp.lv[level].h;
  if (stride) *stride = ex->hp.lv[level].pstride;
  return ORBX_OK;
}

int orbx_debug_level_padded(orbx_extractor* ex, int frame, int level, uint8_t* dst) {
  if (!ex || !ex->planned || !dst || level < 0 || level >= ex->hp.nlevels || frame < 0 ||
      frame >= ex->lastFrames)
    return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  const LevelDesc& L = ex->hp.lv[level];
  {
    hipError_t e = hipStreamSynchronize(ex->stream);
    if (e == hipSuccess)
      e = hipMemcpy(dst, ex->d_pyr + L.pyr_off + (size_t)frame * L.pyr_plane, L.pyr_plane,
                    hipMemcpyDeviceToHost);
    if (e != hipSuccess) { ex->last_hip_error = e; return ORBX_ERR_HIP; }
  }
  return ORBX_OK;
}

int orbx_pyramid(orbx_extractor* ex, int level, const uint8_t** padded, int* width, int* height,
                 int* stride) {
  if (!ex || !ex->planned || !padded || level < 0 || level >= ex->hp.nlevels || ex->lastFrames < 1)
    return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  auto& last_hip_error = ex->last_hip_error;
  // host mirror of frame 0's pyramid: all levels back to back in pinned memory, fetched together
  // on the first request after an extraction (eight transfers, one wait)
  size_t total = 0, off = 0;
  for (int l = 0; l < ex->hp.nlevels; ++l) {
    if (l == level) off = total;
    total += ex->hp.lv[l].pyr_plane;
  }
  if (ex->h_pyr_bytes != total) {
    if (ex->h_pyr) (void)hipHostFree(ex->h_pyr);
    ex->h_pyr = nullptr; ex->h_pyr_bytes = 0; ex->h_pyr_valid = false;
    HIP_TRY(hipHostMalloc((void**)&ex->h_pyr, total, hipHostMallocDefault));
    ex->h_pyr_bytes = total;
  }
  if (!ex->h_pyr_valid) {
    size_t o = 0;
    for (int l = 0; l < ex->hp.nlevels; ++l) {
      const LevelDesc& L = ex->hp.lv[l];
      HIP_TRY(hipMemcpyAsync(ex->h_pyr + o, ex->d_pyr + L.pyr_off, L.pyr_plane, hipMemcpyDeviceToHost, ex->stream));
      o += L.pyr_plane;
    }
    HIP_TRY(hipStreamSynchronize(ex->stream));
    ex->h_pyr_valid = true;
  }
  *padded = ex->h_pyr + off;
  return orbx_level_dims(ex, level, width, height, stride);
}

int orbx_debug_level_candidates(orbx_extractor* ex, int frame, int level, int32_t* xys, int cap,
                                int* n_out) {
  if (!ex || !ex->planned || level < 0 || level >= ex->hp.nlevels || frame < 0 ||
      frame >= ex->lastFrames || !n_out)
    return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  const LevelDesc& L = ex->hp.lv[level];
  const int nc = L.ncx * L.ncy;
  std::vector<int> counts(nc);
  std::vector<uint32_t> seg((size_t)nc * L.cand_cap_cell);
  hipError_t e = hipStreamSynchronize(ex->stream);
  if (e == hipSuccess)
    e = hipMemcpy(counts.data(), ex->d_cellCount + (size_t)frame * ex->hp.cells_per_frame + L.cell_first,
                  (size_t)nc * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess)
    e = hipMemcpy(seg.data(), ex->d_cand + (size_t)frame * ex->hp.cand_frame + L.cand_off,
                  seg.size() * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { ex->last_hip_error = e; return ORBX_ERR_HIP; }
  int n = 0;
  for (int c = 0; c < nc; ++c)
    for (int k = 0; k < counts[c]; ++k, ++n)
      if (n < cap && xys) {
        const uint32_t key = seg[(size_t)c * L.cand_cap_cell + k];
        xys[3 * n] = key & 0xFFF; xys[3 * n + 1] = (key >> 12) & 0xFFF; xys[3 * n + 2] = key >> 24;
      }
  *n_out = n;
  return n > cap ? ORBX_ERR_CAPACITY : ORBX_OK;
}

int orbx_debug_level_selected(orbx_extractor* ex, int frame, int level, int32_t* xys, int cap,
                              int* n_out) {
  if (!ex || !ex->planned || level < 0 || level >= ex->hp.nlevels || frame < 0 ||
      frame >= ex->lastFrames || !n_out)
    return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  const LevelDesc& L = ex->hp.lv[level];
  int n = 0;
  std::vector<uint32_t> keys(L.sel_cap);
  hipError_t e = hipStreamSynchronize(ex->stream);
  if (e == hipSuccess)
    e = hipMemcpy(&n, ex->d_selCount + (size_t)frame * ex->hp.nlevels + level, 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess)
    e = hipMemcpy(keys.data(), ex->d_sel + (size_t)frame * ex->hp.sel_frame + L.sel_off,
                  keys.size() * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { ex->last_hip_error = e; return ORBX_ERR_HIP; }
  for (int i = 0; i < n && i < cap && xys; ++i) {
    xys[3 * i] = keys[i] & 0xFFF; xys[3 * i + 1] = (keys[i] >> 12) & 0xFFF; xys[3 * i + 2] = keys[i] >> 24;
  }
  *n_out = n;
  return n > cap ? ORBX_ERR_CAPACITY : ORBX_OK;
}

int orbx_debug_level_blurred(orbx_extractor* ex, int frame, int level, uint8_t* dst) {
  if (!ex || !ex->planned || !dst || level < 0 || level >= ex->hp.nlevels || frame < 0 ||
      frame >= ex->lastFrames)
    return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  const LevelDesc& L = ex->hp.lv[level];
  hipError_t e = hipSuccess;
  if (ex->hp.blur_in_describe) {
    // the default schedule produces no blurred planes (the descriptor kernel blurs each keypoint's
    // window itself): the dump runs the full-plane blur kernel on the frame's pyramid, which is
    // still where the extraction left it
    if (!ex->d_blur) e = hipMalloc((void**)&ex->d_blur, ex->blur_bytes);
    if (e == hipSuccess) {
      orbx::launch_blur(ex->hp, ex->dp, ex->d_pyr, ex->d_blur, frame, 1, ex->stream);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ex->stream);
  if (e == hipSuccess)
    e = hipMemcpy2D(dst, L.w, ex->d_blur + L.blur_off + (size_t)frame * L.blur_plane, L.bstride, L.w,
                    L.h, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { ex->last_hip_error = e; return ORBX_ERR_HIP; }
  return ORBX_OK;
}

static int debug_sort(int device_id, uint64_t* records, int n, int team);

int orbx_debug_sort_records(int device_id, uint64_t* records, int n) { return debug_sort(device_id, records, n, 0); }

int orbx_debug_sort_records_team(int device_id, uint64_t* records, int n) { return debug_sort(device_id, records, n, 1); }

static int debug_sort(int device_id, uint64_t* records, int n, int team) {
  if (!records || n < 0 || n > 3000) return ORBX_ERR_BAD_ARG;
  if (n == 0) return ORBX_OK;
  if (hipSetDevice(device_id) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  uint64_t* d = nullptr;
  if (hipMalloc((void**)&d, (size_t)n * 8) != hipSuccess) return ORBX_ERR_HIP;
  hipError_t e = hipMemcpy(d, records, (size_t)n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && orbx::launch_debug_sort(d, n, team, nullptr) != 0) e = hipErrorInvalidValue;
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(records, d, (size_t)n * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return e == hipSuccess ? ORBX_OK : ORBX_ERR_HIP;
}

int orbx_set_profiling(orbx_extractor* ex, int enabled) {
  if (!ex) return ORBX_ERR_BAD_ARG;
  ex->profiling = enabled != 0;
  ex->evValid = false;
  return ORBX_OK;
}

int orbx_stage_times(orbx_extractor* ex, float* ms, int n) {
  if (!ex || !ms || n < ORBX_NUM_STAGES || !ex->evValid) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  hipError_t e = hipEventSynchronize(ex->ev[ORBX_NUM_STAGES]);
  for (int i = 0; i < ORBX_NUM_STAGES && e == hipSuccess; ++i)
    e = hipEventElapsedTime(&ms[i], ex->ev[i], ex->ev[i + 1]);
  if (e != hipSuccess) { ex->last_hip_error = e; return ORBX_ERR_HIP; }
  return ORBX_OK;
}

int orbx_detect_launch_history(orbx_extractor* ex, int last_calls, float* ms, int32_t* frames, int cap,
                               int* n_launches) {
  if (!ex || !ms || !frames || !n_launches || last_calls < 1) return ORBX_ERR_BAD_ARG;
  if (hipSetDevice(ex->device) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  if (const int js_ = ex->joinParts()) return js_;
  hipError_t e = hipStreamSynchronize(ex->stream);
  const long ncalls = std::min<long>({(long)last_calls, ex->detCalls, (long)orbx_extractor::kDetRing});
  int n = 0;
  for (long c = ex->detCalls - ncalls; c < ex->detCalls && e == hipSuccess; ++c) {   // oldest first
    const int ds = (int)(c % orbx_extractor::kDetRing);
    for (int k = 0; k < ex->detParts[ds] && e == hipSuccess; ++k) {
      if (n >= cap) return ORBX_ERR_CAPACITY;
      e = hipEventElapsedTime(&ms[n], ex->evDet[ds][k][0], ex->evDet[ds][k][1]);
      frames[n] = ex->detFrames[ds][k];
      ++n;
    }
  }
  if (e != hipSuccess) { ex->last_hip_error = e; return ORBX_ERR_HIP; }
  *n_launches = n;
  return ORBX_OK;
}

int orbx_detect_launch_times(orbx_extractor* ex, float* ms, int32_t* frames, int cap, int* n_launches) {
  if (!ex || ex->lastParts <= 0 || cap < ex->lastParts) return ORBX_ERR_BAD_ARG;
  return orbx_detect_launch_history(ex, 1, ms, frames, cap, n_launches);
}

// ---------------------------------------------------------------------------
// Synthetic textured frame: mid-gray, 400 random axis-aligned rectangles
// (side 6..96, gray 0..255), +-6 uniform noise, one 3x3 box blur (REFLECT_101
// borders).  PCG32 (XSH-RR), stream constant 54.
// ---------------------------------------------------------------------------
namespace {
struct Pcg32 {
  uint64_t state, inc;
  explicit Pcg32(uint64_t seed) {
    state = 0; inc = (54u << 1) | 1u;
    next(); state += seed; next();
  }
  uint32_t next() {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32 - rot) & 31));
  }
  uint32_t below(uint32_t n) { return (uint32_t)(((uint64_t)next() * n) >> 32); }
};
}  // namespace

int orbx_synth_frame(uint32_t seed, int width, int height, int stride, uint8_t* dst) {
  if (!dst || width < 8 || height < 8 || stride < width) return ORBX_ERR_BAD_ARG;
  Pcg32 rng(0x0B5EEDull ^ (uint64_t)seed);
  std::vector<int> img((size_t)width * height, 128);
  for (int r = 0; r < 400; ++r) {
    int rw = 6 + (int)rng.below(91), rh = 6 + (int)rng.below(91);
    int x0 = (int)rng.below((uint32_t)width) - rw / 2, y0 = (int)rng.below((uint32_t)height) - rh / 2;
    int g = (int)rng.below(256);
    for (int y = std::max(y0, 0); y < std::min(y0 + rh, height); ++y)
      for (int x = std::max(x0, 0); x < std::min(x0 + rw, width); ++x) img[(size_t)y * width + x] = g;
  }
  for (size_t i = 0; i < img.size(); ++i) {
    int v = img[i] + (int)rng.below(13) - 6;
    img[i] = std::min(std::max(v, 0), 255);
  }
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      int s = 0;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx)
          s += img[(size_t)reflect101(y + dy, height) * width + reflect101(x + dx, width)];
      dst[(size_t)y * stride + x] = (uint8_t)((s + 4) / 9);
    }
  return ORBX_OK;
}

}  // extern "C"
