/*
 * orbx - MI355X (gfx950) ORB feature front-end, C ABI.
 *
 * This header is the whole drop-in boundary: plain C types, no C++ exceptions,
 * no torch types.  Every function returns an orbx_status (0 = ok, negative =
 * error) unless stated otherwise.  Each entry point names the reference
 * interface (paths relative to the ORB-SLAM3 tree) whose work it performs.
 *
 * Threading: one orbx_extractor owns one HIP stream and private device
 * buffers.  Different extractors may be used concurrently from different host
 * threads (the reference runs one extractor per camera on its own thread,
 * core/src/Frame.cc:151-154); a single extractor must not be called
 * concurrently with itself.
 */
#ifndef ORBX_H_
#define ORBX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum orbx_status {
  ORBX_OK = 0,
  ORBX_ERR_EMPTY_IMAGE = -1,  /* reference returns -1, core/lib/ORB/extractor.cc:1023-1024 */
  ORBX_ERR_TOO_SMALL = -2,    /* a pyramid level has no 35-px cell, or is so much taller than
                               * wide that extractor.cc:549 rounds nIni to 0 (reference: undefined) */
  ORBX_ERR_CAPACITY = -3,     /* output capacity too small; *n_out holds the need */
  ORBX_ERR_BAD_ARG = -4,
  ORBX_ERR_NO_DEVICE = -5,    /* no usable HIP device / HIP runtime error at set-up */
  ORBX_ERR_HIP = -6,          /* HIP runtime error during a call */
  ORBX_ERR_UNSUPPORTED = -7,  /* parameters beyond what the kernels are sized for */
  ORBX_ERR_BATCH = -8,        /* more frames than the handle's max_batch */
  ORBX_ERR_SCRATCH = -9,      /* input beyond what the fixed scratch buffer of an allocation-free
                               * entry point holds */
  ORBX_ERR_COMM = -10         /* RCCL library missing or an RCCL call failed */
} orbx_status;

/* cv::KeyPoint layout (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id. */
typedef struct orbx_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orbx_keypoint;

/* Constructor arguments of ORB_SLAM3::ORBextractor (core/lib/ORB/extractor.h:49-50)
 * plus where to run. */
typedef struct orbx_params {
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels;
  int32_t ini_th_fast;
  int32_t min_th_fast;
  int32_t device_id; /* HIP device ordinal */
  int32_t max_batch; /* frames per batched call this handle is sized for, >= 1 */
} orbx_params;

typedef struct orbx_extractor orbx_extractor;

const char* orbx_strerror(int status);

/* ORBextractor::ORBextractor, core/lib/ORB/extractor.cc:406-463. */
int orbx_create(const orbx_params* params, orbx_extractor** out);
void orbx_destroy(orbx_extractor* ex);

/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (extractor.h:61-73) and mnFeaturesPerLevel.
 * Each array holds nlevels entries; any pointer may be NULL. */
int orbx_get_tables(const orbx_extractor* ex, float* scale, float* inv_scale, float* sigma2,
                    float* inv_sigma2, int32_t* features_per_level);

/* Upper bound on the number of keypoints one frame can produce with this
 * handle (the quadtree may exceed nfeatures, extractor.cc:595-723). */
int orbx_max_keypoints(const orbx_extractor* ex);

/* ORBextractor::operator(), core/lib/ORB/extractor.cc:1018-1101.
 * Host image in (8-bit, row stride in bytes), host keypoints/descriptors out.
 * kps and desc hold cap entries (cap*28 and cap*32 bytes).  *n_out = number
 * of keypoints, *mono_out = the reference's return value (monoIndex): entries
 * [0, mono) lie outside [lap0, lap1] in x, entries [mono, n) inside, the
 * latter filled back to front.  Blocks until the outputs are on the host. */
int orbx_extract(orbx_extractor* ex, const uint8_t* img, int width, int height, int stride,
                 int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                 int* mono_out);

/* mvImagePyramid of the last orbx_extract call (extractor.h:75,
 * extractor.cc:1103-1127): host pointer to the padded level (19-pixel
 * REFLECT_101 border on every side); the level image starts at
 * padded + 19*stride + 19.  Downloaded lazily; valid until the next call on
 * this handle. */
int orbx_pyramid(orbx_extractor* ex, int level, const uint8_t** padded, int* width, int* height,
                 int* stride);

/* Batched operator(): n_frames equally sized host images -> per-frame outputs.
 * kps/desc hold n_frames*cap entries, frame f at offset f*cap.  The batch runs
 * as groups of up to 128 frames whose transfers in, kernels and transfers out
 * overlap.  Images and result arrays in page-locked host memory
 * (orbx_host_alloc, hipHostMalloc, hipHostRegister) are read and written by the
 * copy engines directly - then entries of a frame beyond its n_out are
 * unspecified; ordinary memory goes through pinned staging filled by a few
 * host threads (up to 12, within OMP_NUM_THREADS and the machine's CPUs;
 * ORBX_HOST_THREADS overrides the number).  A call with at most eight frames
 * and 4 MB of images (one frame, a stereo pair, a rig) is served for latency
 * instead: no groups, helper streams or copy engines - the images are packed
 * into page-locked memory the first kernel reads, the results are written
 * to page-locked memory by a last small kernel, one wait. */
int orbx_extract_batch(orbx_extractor* ex, int n_frames, const uint8_t* const* imgs, int width,
                       int height, int stride, int lap0, int lap1, orbx_keypoint* kps,
                       uint8_t* desc, int cap, int* n_out, int* mono_out);

/* Page-locked host memory (hipHostMalloc / hipHostFree) for images and results
 * of the host-pointer entry points, for callers that do not link HIP. */
int orbx_host_alloc(size_t bytes, void** out);
int orbx_host_free(void* p);

/* Device-resident batched operator(): d_imgs points at n_frames contiguous
 * images (height*stride bytes each) in device memory.  Work is enqueued on the
 * handle's stream and the call returns without waiting; results stay in the
 * handle's device buffers (orbx_batch_results_device) until the next call.
 * orbx_sync waits for completion.  One frame (height*stride bytes) and one
 * padded level-0 plane must each stay below 4 GiB: the kernels address them
 * with 32-bit offsets, and larger ones return ORBX_ERR_UNSUPPORTED. */
int orbx_extract_batch_device(orbx_extractor* ex, int n_frames, const uint8_t* d_imgs, int width,
                              int height, int stride, int lap0, int lap1);
int orbx_sync(orbx_extractor* ex);
/* A batch of 256 frames or more runs as two parts on two streams of the
 * handle.  By default the handle's stream (orbx_stream) waits for both parts
 * before orbx_extract_batch_device returns, so work a caller orders behind that
 * stream - its own kernels, or overwriting d_imgs - follows the whole batch.
 * With pipelining enabled that wait is left out between equally sized
 * back-to-back batches (each part then follows only its own predecessor, about
 * a tenth more frames per second): after orbx_extract_batch_device only
 * orbx_sync, a fresh orbx_stream() / orbx_batch_results_device call, the
 * download calls and the derived stages (stereo, grid, bag of words, matchers)
 * cover all parts; a stream pointer fetched earlier does not.  d_imgs must then
 * stay unchanged until one of those has been used.  No counterpart in the
 * reference. */
int orbx_set_pipelined(orbx_extractor* ex, int enabled);

/* Device buffers of the last batched call: keypoints [n_frames][cap] (28 B),
 * descriptors [n_frames][cap][32], n and mono as int32[n_frames]. */
int orbx_batch_results_device(orbx_extractor* ex, orbx_keypoint** d_kps, uint8_t** d_desc,
                              int32_t** d_n, int32_t** d_mono, int* cap);
/* Copy frame f of the last batched call to the host. */
int orbx_batch_download(orbx_extractor* ex, int frame, orbx_keypoint* kps, uint8_t* desc, int cap,
                        int* n_out, int* mono_out);

/* The handle's HIP stream (hipStream_t as void*), for callers that order
 * their own work against it.  The pointer is constant for the handle's life;
 * on a pipelined handle (orbx_set_pipelined) the call itself makes the stream
 * wait for every part of the last batch, so call it again after each batch. */
void* orbx_stream(orbx_extractor* ex);

/* ---- stage dumps (parity tests; frame indexes the last batched call) ------ */
/* Padded pyramid level of one frame into dst (rows*stride bytes, stride from
 * orbx_level_dims). */
int orbx_level_dims(orbx_extractor* ex, int level, int* width, int* height, int* stride);
int orbx_debug_level_padded(orbx_extractor* ex, int frame, int level, uint8_t* dst);
/* FAST candidates of one level in quadtree input order: (x, y, score) triples,
 * coordinates relative to (16,16) as in extractor.cc:822-829. */
int orbx_debug_level_candidates(orbx_extractor* ex, int frame, int level, int32_t* xys, int cap,
                                int* n_out);
/* Quadtree-selected keys of one level, in list order, same triples. */
int orbx_debug_level_selected(orbx_extractor* ex, int frame, int level, int32_t* xys, int cap,
                              int* n_out);
/* Blurred level image (width*height bytes, contiguous). */
int orbx_debug_level_blurred(orbx_extractor* ex, int frame, int level, uint8_t* dst);

/* Sorts n 64-bit records (high word = comparison key, low word = payload) on the device
 * with the quadtree kernel's sort, which must reproduce the permutation of
 * std::sort(..., compareNodes) (extractor.cc:528-541, 673-674) including the order of
 * records with equal keys.  n <= 3000.  Test hook. */
int orbx_debug_sort_records(int device_id, uint64_t* records, int n);
/* The same with the sort of the multi-wave quadtree kernel that small batches use. */
int orbx_debug_sort_records_team(int device_id, uint64_t* records, int n);

/* ---- timing --------------------------------------------------------------- */
#define ORBX_STAGE_PYRAMID 0
#define ORBX_STAGE_FAST 1
#define ORBX_STAGE_OCTREE 2
#define ORBX_STAGE_ASSEMBLE 3
#define ORBX_STAGE_BLUR 4
#define ORBX_STAGE_DESCRIBE 5
#define ORBX_NUM_STAGES 6
/* When enabled, a device-resident batched call (orbx_extract_batch_device) runs
 * its stages one after the other on the handle's stream with HIP events between
 * them; orbx_stage_times (after orbx_sync) returns the milliseconds of the last
 * such call per stage.  The host-pointer calls (orbx_extract,
 * orbx_extract_batch) keep their own pipelined schedule and record no stage
 * times: orbx_stage_times returns ORBX_ERR_BAD_ARG after them. */
int orbx_set_profiling(orbx_extractor* ex, int enabled);
int orbx_stage_times(orbx_extractor* ex, float* ms, int n);
/* Detection-kernel launches of the last batched call in the normal
 * (non-profiling) schedule, where a large batch runs as two parts on two
 * streams: per launch the HIP-event duration in milliseconds, measured on the
 * stream it ran on, and the frames it covered.  Waits for the call. */
int orbx_detect_launch_times(orbx_extractor* ex, float* ms, int32_t* frames, int cap,
                             int* n_launches);
/* The same for the last `last_calls` batched calls (at most 32 are kept), oldest
 * first: the launches of a run of back-to-back calls, read out after the run,
 * so that reading them does not put a wait between the calls.  Calls made in
 * profiling mode contribute no launches.  (A measurement aid: no counterpart
 * in the reference.) */
int orbx_detect_launch_history(orbx_extractor* ex, int last_calls, float* ms, int32_t* frames,
                               int cap, int* n_launches);

/* ---- Hamming matching ------------------------------------------------------ */
/* ORBmatcher::DescriptorDistance, core/lib/ORB/matcher.cc:1961-1975 (host). */
int orbx_hamming(const uint8_t* a, const uint8_t* b);

/* Brute-force Hamming 2-NN: what cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, m, 2)
 * returns at core/src/Frame.cc:1177: per query the two nearest train rows by
 * (distance, train index) ascending.  idx holds nq*2 uint32, dist nq*2
 * uint16; missing neighbours (nt < 2) are idx 0xFFFFFFFF / dist 0xFFFF.
 * Host pointers; runs on device_id. */
int orbx_knn2(int device_id, const uint8_t* q, int nq, const uint8_t* t, int64_t nt,
              uint32_t* idx, uint16_t* dist);
/* Same on device pointers, asynchronous on stream (hipStream_t as void*, NULL
 * = default stream).  Train row j is reported as index j + index_offset, so a
 * database shard reports global indices. */
int orbx_knn2_device(const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt,
                     uint32_t index_offset, uint32_t* d_idx, uint16_t* d_dist, void* stream);
/* orbx_knn2_device for callers that cannot allocate (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): same
 * arguments and results, but the per-segment partial results live in a fixed
 * 64 MiB device array inside the library (one copy per GPU, held until the
 * process ends) and no allocation routine is called.  No counterpart in the
 * reference.  Limit: a scan of more than 4 194 304 train rows runs as
 * s = ceil(nt / 4 194 304) segments at least, whose partial results take
 * s * (nq * 12 + 512) bytes; a call for which that exceeds the array returns
 * ORBX_ERR_SCRATCH (up to 4 194 304 train rows any nq fits).  Runs on the device
 * that is current at the call (the stream's).  All calls on one GPU share the
 * array: calls that overlap on one GPU (different streams or threads) race.
 * The 4-byte aligned d_dist / 8-byte aligned d_idx of orbx_knn2_device apply. */
int orbx_knn2_device_noalloc(const uint8_t* d_q, int nq, const uint8_t* d_t, int64_t nt,
                             uint32_t index_offset, uint32_t* d_idx, uint16_t* d_dist, void* stream);
size_t orbx_knn2_noalloc_scratch_bytes(void);
/* Merge nparts partial 2-NN results (layout [part][nq][2]) into the global
 * 2-NN with the same (distance, index) ordering.  Device pointers; the index
 * array 8-byte and the distance array 4-byte aligned. */
int orbx_top2_merge_device(const uint32_t* d_idx_parts, const uint16_t* d_dist_parts, int nparts,
                           int nq, uint32_t* d_idx, uint16_t* d_dist, void* stream);

/* ---- 2-NN against a train set sharded over the GPUs of a node ---------------- */
/* One process per GPU.  The query rows are replicated, every rank holds one
 * contiguous range of the train rows (a keyframe descriptor database, the
 * scaled-up form of core/src/Frame.cc:1177); the only exchange is one
 * all-gather of the per-rank top-2 records over RCCL, followed by the merge of
 * orbx_top2_merge_device in rank order.  With contiguous ranges in rank order
 * that is global index order, so results equal the unsharded orbx_knn2_device
 * bit for bit, ties included.
 *
 * nccl_comm: an ncclComm_t (as void*) whose ranks each call this function with
 * the same nq and their own shard; shard_offset = global index of the shard's
 * first row.  d_workspace: orbx_knn2_sharded_workspace_bytes(nranks, nq) bytes
 * of device memory, 256-byte aligned, or NULL (then the call allocates
 * stream-ordered memory itself).  Every rank receives the global result.  The
 * call returns once the exchange has completed on stream (it reads one status
 * word per rank from the gathered blocks); the merge that follows is
 * asynchronous on stream.  RCCL is taken from the copy already loaded in the
 * process, else from librccl.so.1 on the loader path; ORBX_ERR_COMM if neither
 * exists.
 *
 * Failure: a rank whose own part fails (bad shard pointer, allocation failure
 * inside its search, a workspace it cannot use) still enters the all-gather,
 * with "no neighbour" records and its status in its block's header, and EVERY
 * rank returns the worst status of all ranks; no rank is left waiting in the
 * collective and the communicator stays usable.  Exceptions, which return
 * before the collective and leave the peers waiting: a NULL or foreign
 * communicator, RCCL missing, and failure to allocate the exchange memory when
 * d_workspace is NULL or unusable; nq must be the same on every rank.  A
 * non-OK status that comes from RCCL itself (ORBX_ERR_COMM) leaves the
 * communicator unusable. */
int orbx_knn2_sharded(void* nccl_comm, const uint8_t* d_q, int nq, const uint8_t* d_t_shard,
                      int64_t nt_shard, uint32_t shard_offset, uint32_t* d_idx, uint16_t* d_dist,
                      void* d_workspace, size_t workspace_bytes, void* stream);
size_t orbx_knn2_sharded_workspace_bytes(int nranks, int nq);
/* For callers without an RCCL binding of their own: ncclGetUniqueId (128 bytes,
 * made by one rank and handed to the others by any means), ncclCommInitRank on
 * device_id, ncclCommDestroy. */
int orbx_rccl_unique_id(uint8_t* id128);
int orbx_rccl_comm_create(const uint8_t* id128, int nranks, int rank, int device_id, void** comm);
int orbx_rccl_comm_destroy(void* comm);

/* Row-band stereo descriptor search: the first half of
 * Frame::ComputeStereoMatches, core/src/Frame.cc:840-908.  For each left
 * keypoint the right keypoint with the smallest Hamming distance among those
 * whose row band contains the left row, whose octave differs by at most one
 * and whose x lies in [uL - max_d, uL - min_d]; best_dist starts at 100 and
 * the first candidate in ascending right index wins ties.  best_idx = -1 and
 * best_dist = 100 when nothing qualified.  Host pointers; runs on device_id. */
int orbx_stereo_band_match(int device_id, const orbx_keypoint* kps_l, const uint8_t* desc_l,
                           int n_l, const orbx_keypoint* kps_r, const uint8_t* desc_r, int n_r,
                           const float* scale_factors, int nlevels, int n_rows, float min_d,
                           float max_d, int32_t* best_idx, int32_t* best_dist);

/* ---- stereo association on device-resident extractions --------------------- */
/* Frame::ComputeStereoMatches, core/src/Frame.cc:831-996, for frames
 * [0, n_frames) of the last batched (or single-frame) calls on `left` and
 * `right` (same device, image size, nlevels and scale factor): row-band
 * descriptor search (:840-908), 11x11 L1 search over +-5 columns on pyramid
 * level kpL.octave of both handles with parabola refinement (:910-981),
 * disparity/depth, and removal of matches whose L1 distance is >= 1.5*1.4*
 * median (:984-995).  mbf and mb are Frame::mbf and Frame::mb.  The pyramids,
 * keypoints and descriptors are read where the extractions left them in
 * device memory.  Asynchronous on left's stream (ordered after right's
 * stream); results stay in left's device buffers.  Keypoints whose windows
 * would leave the level image (a cv::Exception in the reference) get no
 * match. */
int orbx_stereo_match_batch_device(orbx_extractor* left, orbx_extractor* right, int n_frames,
                                   float mbf, float mb);
/* mvuRight / mvDepth of the last stereo call: float [n_frames][cap], entry i
 * belongs to left keypoint i, -1 = no match (Frame.cc:832-833). */
int orbx_stereo_results_device(orbx_extractor* left, float** d_uright, float** d_depth, int* cap);
/* Copy frame f to the host (waits for the stereo call).  l1_dist (may be
 * NULL) receives the winning L1 distance of every keypoint that reached
 * Frame.cc:979, before the median rule, else -1. */
int orbx_stereo_download(orbx_extractor* left, int frame, float* uright, float* depth,
                         int32_t* l1_dist, int cap, int* n_out);
/* Single stereo frame after orbx_extract on both handles: the drop-in for
 * Frame::ComputeStereoMatches.  uright / depth hold cap floats. */
int orbx_stereo_match(orbx_extractor* left, orbx_extractor* right, float mbf, float mb,
                      float* uright, float* depth, int cap, int* n_out);

/* The descriptor search of Frame::ComputeStereoFishEyeMatches, core/src/Frame.cc:
 * 1157-1185, on two device-resident extractions made with the cameras' lapping
 * areas: BFmatcher.knnMatch(stereoDescLeft, stereoDescRight, matches, 2) (:1177)
 * with stereoDescLeft = rows [monoLeft, Nleft) of the left frame's descriptors
 * and stereoDescRight = rows [monoRight, Nright) of the right one (:1159-1165),
 * mono* being the two operator() return values.  idx / dist hold cap*2 entries
 * in the layout of orbx_knn2; idx values are trainIdx (relative to monoRight).
 * *n_query = Nleft - monoLeft (ORBX_ERR_CAPACITY if above cap), *n_train =
 * Nright - monoRight; any of the four outputs may be NULL.  Blocks. */
int orbx_lapping_knn2(orbx_extractor* left, orbx_extractor* right, int frame, uint32_t* idx, uint16_t* dist,
                      int cap, int* n_query, int* mono_left, int* n_train, int* mono_right);
/* The same search (core/src/Frame.cc:1159-1177) for frames [0, n_frames) of the
 * last batched extractions on the two handles in one launch; results stay on
 * the device.  Asynchronous on the left handle's stream.
 * orbx_lapping_knn2_download fetches one frame's results in the layout of
 * orbx_lapping_knn2 (blocks; the counts it reports are those the batch call
 * saw, kept with the left handle: the right handle may be reused or destroyed
 * in between). */
int orbx_lapping_knn2_batch_device(orbx_extractor* left, orbx_extractor* right, int n_frames);
int orbx_lapping_knn2_download(orbx_extractor* left, int frame, uint32_t* idx, uint16_t* dist, int cap,
                               int* n_query, int* mono_left, int* n_train, int* mono_right);
/* Lowe's ratio test as core/src/Frame.cc:1185 applies it to those records:
 * matches.size() >= 2 && matches[0].distance < matches[1].distance * ratio
 * (float distances, product and comparison in double; the reference's ratio is
 * 0.7).  match[i] = trainIdx of the nearest neighbour when query i passes,
 * else -1.  The generic form works on any device-resident (idx, dist) records
 * in the layout of orbx_knn2; the batch form on the records the last
 * orbx_lapping_knn2_batch_device call left on the left handle (frame f's
 * verdicts at d_match + f * cap).  Asynchronous. */
int orbx_ratio_test_device(const uint32_t* d_idx, const uint16_t* d_dist, int nq, double ratio,
                           int32_t* d_match, void* stream);
int orbx_lapping_ratio_test_batch_device(orbx_extractor* left, double ratio);
int orbx_lapping_matches_device(orbx_extractor* left, int32_t** d_match, int* cap);
/* One frame's verdicts to the host (blocks); *n_good = queries that passed. */
int orbx_lapping_matches_download(orbx_extractor* left, int frame, int32_t* match, int cap, int* n_query,
                                  int* n_good);

/* Frame::ComputeStereoFromRGBD, core/src/Frame.cc:998-1016, for frames
 * [0, n_frames) of the last extraction + grid call on ex (the undistorted x
 * comes from the grid call): d_depth points at n_frames float images in device
 * memory (height rows of stride_floats floats each, already scaled by the depth
 * map factor as in the Frame constructor, Frame.cc:264-266).  Asynchronous on
 * the handle's stream; results through orbx_stereo_results_device /
 * orbx_stereo_download like a stereo pair's (l1_dist is -1). */
int orbx_stereo_from_rgbd_batch_device(orbx_extractor* ex, int n_frames, const float* d_depth, int stride_floats,
                                       float mbf);

/* Results derived from an extraction (stereo association, frame grid, bag of
 * words) belong to the keypoints of that extraction: the next extraction on the
 * handle invalidates them, and their download / query calls return
 * ORBX_ERR_BAD_ARG until the deriving call has run again. */

/* ---- frame grid on device-resident extractions ------------------------------ */
/* Pinhole camera with OpenCV's radial-tangential coefficients, as Frame holds
 * them in mK and mDistCoef (k3 = 0 when the settings give four coefficients). */
typedef struct orbx_pinhole {
  float fx, fy, cx, cy;
  float k1, k2, p1, p2, k3;
} orbx_pinhole;

/* cv::undistortPoints(src, dst, K, D, noArray(), K) as called at
 * core/src/Frame.cc:787-788 and :814-815 (five fixed-point iterations in
 * double).  xy / out_xy hold n (x, y) pairs.  Host. */
int orbx_undistort_points(const orbx_pinhole* cam, const float* xy, int n, float* out_xy);
/* Frame::ComputeImageBounds, core/src/Frame.cc:801-829:
 * bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY}; cam NULL or k1 == 0 gives the
 * image rectangle.  Host. */
int orbx_image_bounds(const orbx_pinhole* cam, int width, int height, float* bounds);

/* Frame::UndistortKeyPoints (core/src/Frame.cc:771-799; a copy when cam is
 * NULL or k1 == 0) followed by Frame::AssignFeaturesToGrid / PosInGrid
 * (:432-459, :751-762) for frames [0, n_frames) of the last batched call on
 * ex, on the keypoints where the extraction left them.  Asynchronous on the
 * handle's stream.  The grid is a CSR over cells c = posX*48 + posY
 * (mGrid[posX][posY]); a cell's keypoint indices are ascending, as push_back
 * leaves them.  Bounds without extent in x or y are refused
 * (ORBX_ERR_BAD_ARG): distortion strong enough to fold the image's corners onto
 * one point gives such bounds, and the reference would go on with an infinite
 * cell size (Frame.cc:193-196). */
int orbx_frame_grid_batch_device(orbx_extractor* ex, int n_frames, const orbx_pinhole* cam,
                                 const float* bounds);
/* Device buffers of the last grid call: mvKeysUn [n_frames][cap] (28 B each),
 * cell starts int32 [n_frames][64*48+1], cell entries int32 [n_frames][cap]. */
int orbx_frame_grid_results_device(orbx_extractor* ex, orbx_keypoint** d_kps_un,
                                   int32_t** d_cell_start, int32_t** d_cell_entries, int* cap);
/* Copy frame f to the host: kps_un holds cap keypoints, cell_start 64*48+1
 * entries, cell_entries cap entries; any pointer may be NULL. */
int orbx_frame_grid_download(orbx_extractor* ex, int frame, orbx_keypoint* kps_un,
                             int32_t* cell_start, int32_t* cell_entries, int cap, int* n_out);
/* Frame::GetFeaturesInArea, core/src/Frame.cc:682-749, for n_queries windows
 * on one frame of the last grid call.  queries holds 5 floats per window: x, y,
 * r, minLevel, maxLevel.  counts[q] = number of keypoints found; the first
 * per_query_cap of them go to indices[q*per_query_cap ...] in the reference's
 * order.  Host pointers; blocks until done. */
int orbx_features_in_area(orbx_extractor* ex, int frame, const float* queries, int n_queries,
                          int32_t* indices, int per_query_cap, int32_t* counts);

/* ---- guided matchers on a frame with a device-resident grid ------------------ */
/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono),
 * core/lib/ORB/matcher.cc:1591-1808, monocular / rectified case (Nleft == -1).
 * The current frame is frame `frame` of the last extraction + grid call on
 * `cur` (undistorted keypoints, descriptors and grid are read where they lie in
 * device memory).  The last frame arrives as flat host arrays, one entry per
 * keypoint of the last frame; projecting its map points with the current pose
 * (:1614-1630) is the caller's:
 *   last_valid[i]   a map point exists and is not an outlier (:1607-1613)
 *   uv[2i], uv[2i+1]  its projection into the current frame; invz[i] = 1 / depth
 *   last_octave[i], last_angle[i]  of the last frame's keypoint i (:1632, :1691)
 *   last_desc       the map point's descriptor (:1655), 32 bytes each
 *   last_obs[i]     pMP->Observations() > 0 (what :1665 finds once it is assigned)
 * u_right: the current frame's mvuRight (n entries) or NULL (monocular).
 * state:   per current keypoint 0 = no map point, 1 = holds one with
 *          observations, 2 = holds one without; NULL = all 0.
 * forward / backward: the flags of :1621-1622.  th, mbf as in the reference.
 * assigned[i2] (cap entries, n used) = index of the last-frame keypoint whose
 * map point keypoint i2 received, -1 = untouched, -2 = set and cleared again by
 * the rotation check (:1799-1800).  *nmatches = the reference's return value.
 * rounds (may be NULL) = decision rounds the device took.  Blocks. */
int orbx_search_by_projection_last(orbx_extractor* cur, int frame, int n_last, const uint8_t* last_valid,
                                   const float* uv, const float* invz, const int32_t* last_octave,
                                   const float* last_angle, const uint8_t* last_desc,
                                   const uint8_t* last_obs, const float* u_right, const uint8_t* state,
                                   float th, int forward, int backward, float mbf, int check_orientation,
                                   int32_t* assigned, int cap, int* n_out, int* nmatches, int* rounds);
/* The same matcher for a current frame with two cameras (CurrentFrame.Nleft != -1,
 * a fisheye rig), core/lib/ORB/matcher.cc:1591-1808 with the right camera's pass
 * of :1712-1786: `left` and `right` hold the two cameras' extractions and grids
 * for `frame` (orbx_frame_grid_batch_device on each, same image bounds), the
 * reference's keypoint index i2 >= Nleft being right keypoint i2 - Nleft here.
 * uv_left / uv_right: the map point's projections into the two cameras (the
 * second through GetRelativePoseTrl, :1713-1714).  As in the reference the left
 * camera's checks gate the right camera's search: a point behind the camera,
 * with a left projection outside the image, or with an empty left window is not
 * searched in the right camera either; the right projection itself is not
 * checked against the image; the mvuRight test does not apply.  One rotation
 * histogram covers both cameras.  assigned_left / assigned_right, state_left /
 * state_right as assigned / state above, per camera.  *nmatches = the return
 * value.  Blocks. */
int orbx_search_by_projection_last_rig(orbx_extractor* left, orbx_extractor* right, int frame, int n_last,
                                       const uint8_t* last_valid, const float* uv_left, const float* invz,
                                       const float* uv_right, const int32_t* last_octave, const float* last_angle,
                                       const uint8_t* last_desc, const uint8_t* last_obs, const uint8_t* state_left,
                                       const uint8_t* state_right, float th, int forward, int backward,
                                       int check_orientation, int32_t* assigned_left, int cap_left, int* n_left,
                                       int32_t* assigned_right, int cap_right, int* n_right, int* nmatches,
                                       int* rounds);
/* ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints) for
 * a frame with two cameras (F.Nleft != -1), core/lib/ORB/matcher.cc:41-213: per
 * map point the left camera's search (:60-142), whose match is also written
 * into the keypoint's stereo partner in the right camera (mvLeftToRightMatch,
 * :129-135), then the right camera's search (:144-210) around proj_right at
 * level_right, whose match is also written into the partner in the left camera
 * (mvRightToLeftMatch, :197-203).  in_view_right / proj_right / level_right /
 * view_cos_right: mbTrackInViewR, mTrackProjXR / mTrackProjYR,
 * mnTrackScaleLevelR (-1 = none: no right search, :146), mTrackViewCosR.
 * left_to_right[n_left] / right_to_left[n_right]: partner index in the other
 * camera's own numbering, -1 none.  Kept from the reference: a left search whose
 * best candidate fails the ratio test skips the right camera too (:123-124); the
 * right camera's radius ignores th (:147); a partner receives the map point
 * whatever it held, so a keypoint closed by a point with observations opens again
 * when it receives one without; every write counts in *nmatches.  assigned_* /
 * state_* per camera as in orbx_search_by_projection_points.  n_points at most
 * half the single-camera limit.  Blocks. */
int orbx_search_by_projection_points_rig(orbx_extractor* left, orbx_extractor* right, int frame, int n_points,
                                         const uint8_t* in_view, const float* proj, const int32_t* level,
                                         const float* view_cos, const uint8_t* in_view_right, const float* proj_right,
                                         const int32_t* level_right, const float* view_cos_right, const uint8_t* desc,
                                         const uint8_t* obs, const int32_t* left_to_right,
                                         const int32_t* right_to_left, const uint8_t* state_left,
                                         const uint8_t* state_right, float th, float nnratio, int32_t* assigned_left,
                                         int cap_left, int* n_left, int32_t* assigned_right, int cap_right,
                                         int* n_right, int* nmatches, int* rounds);
/* ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist),
 * core/lib/ORB/matcher.cc:1810-1923 (relocalisation).  One entry per keypoint
 * of the keyframe; the caller's part is :1826-1853:
 *   kf_valid[i]  a map point exists, is not bad, is not in sAlreadyFound, and its
 *                distance to the camera centre is inside its invariance range
 *   uv           its projection; level[i] = PredictScale(dist3D, &CurrentFrame)
 *   kf_angle[i]  pKF->mvKeysUn[i].angle;  kf_desc  the map point's descriptor
 * has_point[i2] (0 or 1; NULL = none): CurrentFrame.mvpMapPoints[i2] is not null (:1872).
 * orb_dist in [0, 255].  Outputs as in orbx_search_by_projection_last. */
int orbx_search_by_projection_keyframe(orbx_extractor* cur, int frame, int n_kf, const uint8_t* kf_valid,
                                       const float* uv, const int32_t* level, const float* kf_angle,
                                       const uint8_t* kf_desc, const uint8_t* has_point, float th, int orb_dist,
                                       int check_orientation, int32_t* assigned, int cap, int* n_out,
                                       int* nmatches, int* rounds);
/* ORBmatcher::SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints),
 * core/lib/ORB/matcher.cc:41-142, monocular / rectified case (Nleft == -1): the
 * local-map matcher of Tracking::SearchLocalPoints.  F is frame `frame` of the
 * last extraction + grid call on `f`.  The map points arrive as flat host
 * arrays holding what Frame::isInFrustum left in them:
 *   in_view[i]   mbTrackInView, not bad, not beyond thFarPoints (:51-60)
 *   proj[2i], proj[2i+1]  mTrackProjX, mTrackProjY;  proj_xr[i]  mTrackProjXR
 *   level[i]     mnTrackScaleLevel;  view_cos[i]  mTrackViewCos (:64, :215-220)
 *   desc         the map point's descriptor, 32 bytes each
 *   obs[i]       pMP->Observations() > 0 (what :89-91 finds once it is assigned)
 * u_right, state, assigned, cap, n_out, rounds as in
 * orbx_search_by_projection_last (no rotation check here, so no -2).  nnratio
 * is the matcher's constructor argument.  *nmatches = the return value. */
int orbx_search_by_projection_points(orbx_extractor* f, int frame, int n_points, const uint8_t* in_view,
                                     const float* proj, const float* proj_xr, const int32_t* level,
                                     const float* view_cos, const uint8_t* desc, const uint8_t* obs,
                                     const float* u_right, const uint8_t* state, float th, float nnratio,
                                     int32_t* assigned, int cap, int* n_out, int* nmatches, int* rounds);
/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12,
 * windowSize), core/lib/ORB/matcher.cc:629-733.  F2 is frame `frame2` of the
 * last extraction + grid call on `f2`; F1 arrives as host arrays (mvKeysUn and
 * descriptors, n1 entries).  prev_matched: n1 (x, y) pairs, updated as at
 * :728-730.  matches12: n1 entries, -1 = none.  nnratio and check_orientation
 * are the matcher's constructor arguments.  *nmatches = the return value. */
int orbx_search_for_initialization(orbx_extractor* f2, int frame2, int n1, const orbx_keypoint* kps1_un,
                                   const uint8_t* desc1, float* prev_matched, int window_size,
                                   float nnratio, int check_orientation, int32_t* matches12, int* nmatches,
                                   int* rounds);

/* ---- bag of words ------------------------------------------------------------ */
/* The reference's ORBVocabulary (core/lib/ORB/vocabulary.h:31-62) in device
 * memory.  Supported: TF_IDF (0) or TF (1) weighting with L1 scoring (0), the
 * stock configuration; other combinations return ORBX_ERR_UNSUPPORTED.  Under
 * both modes a word counts with the weight its node stores (the file's). */
typedef struct orbx_vocabulary orbx_vocabulary;
/* ORBVocabulary::loadFromText, core/lib/ORB/vocabulary.cc:96-160: first line
 * "k L scoring weighting", then one node per line "parent isLeaf d0 .. d31
 * weight"; nodes and words are numbered in file order.  (The reference's binary
 * loader, :180-227, reads no node - the node vector is resized to 1 before a
 * loop bounded by its size - so the text format is the one with defined
 * behaviour.)  A line without a node, such as the one a trailing newline
 * leaves, ends the file; the reference turns it into a node with
 * uninitialised contents. */
int orbx_vocabulary_load_text(const char* path, int device_id, orbx_vocabulary** out);
/* The same from arrays: nodes 1..n_nodes in file order (node 0 is the root). */
int orbx_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                           const uint8_t* is_leaf, const uint8_t* descriptors, const double* weights,
                           int device_id, orbx_vocabulary** out);
void orbx_vocabulary_destroy(orbx_vocabulary* voc);
/* n_nodes includes the root. */
int orbx_vocabulary_info(const orbx_vocabulary* voc, int* k, int* L, int* n_nodes, int* n_words);
/* Frame::ComputeBoW, core/src/Frame.cc:764-769 (and KeyFrame::ComputeBoW,
 * core/src/KeyFrame.cc:117-122): vocabulary.transform(descriptors, mBowVec,
 * mFeatVec, levelsup = 4) for frames [0, n_frames) of the last batched call on
 * ex, on the descriptors where the extraction left them.  Asynchronous on the
 * handle's stream. */
int orbx_bow_transform_batch_device(orbx_extractor* ex, const orbx_vocabulary* voc, int n_frames,
                                    int levelsup);
/* Copy frame f to the host (waits).  mBowVec as parallel arrays in ascending
 * word id: word_ids / word_values (doubles, L1-normalised), *n_words entries.
 * mFeatVec as a CSR in ascending node id: node_ids (*n_nodes entries),
 * node_start (*n_nodes + 1), features (keypoint indices, ascending inside a
 * node).  Every array holds cap entries (node_start cap + 1); any may be NULL. */
int orbx_bow_download(orbx_extractor* ex, int frame, uint32_t* word_ids, double* word_values, int* n_words,
                      uint32_t* node_ids, int32_t* node_start, uint32_t* features, int* n_nodes, int cap);

/* ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches), core/lib/ORB/matcher.cc:
 * 222-405, monocular / rectified case (F.Nleft == -1, no second camera): the
 * matcher of Tracking::TrackReferenceKeyFrame and Relocalization.  F is frame
 * `frame` of the last extraction on `f` for which
 * orbx_bow_transform_batch_device has run (its FeatureVector and descriptors
 * are read where they lie in device memory).  The keyframe arrives as host
 * arrays over its n_kf keypoints:
 *   kf_valid[i]  its map point i exists and is not bad (:252-258)
 *   kf_angle[i]  pKF->mvKeysUn[i].angle (:321-325);  kf_desc  32 bytes each
 *   pKF->mFeatVec as a CSR in ascending node id (the layout orbx_bow_download
 *   returns): kf_node_ids[n_kf_nodes], kf_node_start[n_kf_nodes + 1], kf_features
 * nnratio and check_orientation are the matcher's constructor arguments.
 * matches[i2] (cap entries, *n_out used) = keyframe keypoint whose map point
 * frame keypoint i2 received, -1 = none, -2 = received and cleared by the
 * rotation check (:394-401).  *nmatches = the return value.  Blocks. */
int orbx_search_by_bow(orbx_extractor* f, int frame, int n_kf, const uint8_t* kf_valid, const float* kf_angle,
                       const uint8_t* kf_desc, int n_kf_nodes, const uint32_t* kf_node_ids,
                       const int32_t* kf_node_start, const uint32_t* kf_features, float nnratio,
                       int check_orientation, int32_t* matches, int cap, int* n_out, int* nmatches);

/* ---- synthetic input ------------------------------------------------------- */
/* Deterministic textured test frame (PCG32-seeded rectangles + noise + 3x3
 * box blur), written to dst (height rows of stride bytes).  Host only. */
int orbx_synth_frame(uint32_t seed, int width, int height, int stride, uint8_t* dst);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H_ */
