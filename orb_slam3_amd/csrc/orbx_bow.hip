// Bag-of-words transform of device-resident descriptors:
//   Frame::ComputeBoW  core/src/Frame.cc:764-769
//     = ORBVocabulary::transform(descriptors, mBowVec, mFeatVec, 4), the vocabulary class of
//       core/lib/ORB/vocabulary.h:31-62 on top of DBoW2's TemplatedVocabulary (deps/DBoW2, an
//       empty submodule directory in the reference tree; its published algorithm is followed).
//   vocabulary file: ORBVocabulary::loadFromText  core/lib/ORB/vocabulary.cc:96-160
//
//   k_bow_descend   per feature, descend the vocabulary tree: at every level the child at the
//                   smallest Hamming distance, the first one on ties (16 lanes per feature, one
//                   child per lane, children's descriptors contiguous in memory);
//   k_bow_assemble  per frame, the two std::map results as sorted arrays: BowVector (word id ->
//                   weight summed in feature order, L1-normalised in ascending word order) and
//                   FeatureVector (node id -> feature indices, ascending).
// Doubles are added in exactly the order the reference adds them, so values are bit-identical.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/orbx.h"
#include "orbx_bow.h"

namespace orbx {

#define BOW_THREADS 256
#define BOW_LANES 16   // lanes per feature in k_bow_descend

__global__ __launch_bounds__(BOW_THREADS) void k_bow_descend(VocDevice voc, const uint32_t* __restrict__ desc,
                                                             const int* __restrict__ nP, int cap,
                                                             int nidLevel, uint32_t* __restrict__ featWord,
                                                             double* __restrict__ featWeight,
                                                             uint32_t* __restrict__ featNode) {
  const int f = blockIdx.y;
  const int j = threadIdx.x & (BOW_LANES - 1);
  const int i = blockIdx.x * (BOW_THREADS / BOW_LANES) + (threadIdx.x / BOW_LANES);
  if (i >= nP[f]) return;   // uniform over the 16 lanes of a feature
  const uint32_t* q = desc + ((size_t)f * cap + i) * 8;
  uint32_t a[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) a[w] = q[w];
  int node = 0, level = 0;
  uint32_t nid = 0;
  for (;;) {
    const int s = voc.childStart[node], e = voc.childStart[node + 1];
    if (s == e) break;   // Node::isLeaf(): no children
    ++level;
    // key = distance << 20 | position among the children: its minimum is the closest child,
    // the earliest one on ties (strict '<' in the reference's scan)
    uint32_t best = 0xFFFFFFFFu;
    for (int c0 = s; c0 < e; c0 += BOW_LANES) {
      const int c = c0 + j;
      if (c < e) {
        const uint32_t* d = voc.childDesc + (size_t)c * 8;
        uint32_t dist = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) dist += (uint32_t)__builtin_popcount(a[w] ^ d[w]);
        best = min(best, (dist << 20) | (uint32_t)(c - s));
      }
    }
#pragma unroll
    for (int dd = BOW_LANES / 2; dd >= 1; dd >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, dd, BOW_LANES));
    node = voc.childId[s + (int)(best & 0xFFFFFu)];
    if (level == nidLevel) nid = (uint32_t)node;
  }
  if (j == 0) {
    const size_t o = (size_t)f * cap + i;
    featWord[o] = voc.nodeWord[node];
    // the stored Node::weight under TF_IDF and TF alike (a vocabulary built with TF stores 1
    // in every word, but a file may say otherwise and transform() does not look at the mode)
    featWeight[o] = voc.nodeWeight[node];
    featNode[o] = nid;
  }
}

// ascending bitonic sort of P (a power of two) 64-bit keys by the whole workgroup: one
// compare-exchange pair per thread and trip.  A wave's 64 pairs of a trip lie in one aligned block
// of 128 keys whenever the distance is at most 64, in every such step the same block: those steps
// need the wave's program order only, not a workgroup barrier (51 of the 66 steps for 2048 keys).
#define BOWA_THREADS 1024   /* k_bow_assemble: one pair per thread for 2048 keys */
__device__ __forceinline__ void block_sort(uint64_t* keys, int P) {
  const int half = P >> 1;
  for (int k = 2; k <= P; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      if (jj > 64) __syncthreads();
      for (int i = threadIdx.x; i < half; i += BOWA_THREADS) {
        const int t = ((i & ~(jj - 1)) << 1) | (i & (jj - 1)), o = t | jj;
        const uint64_t x = keys[t], y = keys[o];
        const bool up = (t & k) == 0;
        if ((x > y) == up) { keys[t] = y; keys[o] = x; }
      }
      if (jj > 64) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  __syncthreads();
}

// For sorted keys[0..m): run r = maximal range of equal high words.  Writes runStart[r] for
// every run and runStart[R] = m; returns R.  (Block-wide; scratch = one int of LDS per wave.)
__device__ __forceinline__ int block_runs(const uint64_t* keys, int m, int* runStart, int* scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NWAVES = BOWA_THREADS / 64;
  int base = 0;
  for (int c0 = 0; c0 < m; c0 += BOWA_THREADS) {
    const int t = c0 + tid;
    const bool head = t < m && (t == 0 || (uint32_t)(keys[t] >> 32) != (uint32_t)(keys[t - 1] >> 32));
    // a run's number = heads before it: inside the wave from the ballot, across waves from LDS
    const unsigned long long mask = __ballot(head);
    if (lane == 0) scratch[wave] = __popcll(mask);
    __syncthreads();
    int before = base, total = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
      const int c = scratch[w];
      if (w < wave) before += c;
      total += c;
    }
    if (head)
      runStart[before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] = t;
    __syncthreads();
    base += total;
  }
  if (tid == 0) runStart[base] = m;
  __syncthreads();
  return base;
}

// KEYS_GLOBAL: the sort keys live in the caller's workspace (a frame with more keys than LDS
// holds); else in LDS, and instantiated apart so that the sort's accesses are LDS instructions
// rather than flat ones (a flat access to LDS takes the long way through the memory pipeline:
// the kernel took 137 us per frame that way, most of it in the two sorts).
template <bool KEYS_GLOBAL>
__global__ __launch_bounds__(BOWA_THREADS) void k_bow_assemble(const int* __restrict__ nP, int cap, int P,
                                                              const uint32_t* __restrict__ featWord,
                                                              const double* __restrict__ featWeight,
                                                              const uint32_t* __restrict__ featNode,
                                                              uint64_t* __restrict__ keyWs,
                                                              uint32_t* __restrict__ bowWord,
                                                              double* __restrict__ bowVal, int* __restrict__ bowCount,
                                                              uint32_t* __restrict__ fvNode, int* __restrict__ fvStart,
                                                              uint32_t* __restrict__ fvFeat, int* __restrict__ fvCount) {
  extern __shared__ uint64_t sKeys[];
  __shared__ int scratch[BOWA_THREADS];
  __shared__ int sM;
  __shared__ double sNorm;
  const int f = blockIdx.x, tid = threadIdx.x;
  const int n = nP[f];
  const size_t fo = (size_t)f * cap;
  uint64_t* keys;
  if constexpr (KEYS_GLOBAL) keys = keyWs + (size_t)f * P;
  else keys = sKeys;
  int* runStart = fvStart + (size_t)f * (cap + 1);   // also used for the word runs, rewritten below
  // features that take part: weight > 0 ("not stopped")
  if (tid == 0) sM = 0;
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < n; i += BOWA_THREADS) mine += featWeight[fo + i] > 0 ? 1 : 0;
  atomicAdd(&sM, mine);
  __syncthreads();
  const int m = sM;

  // ---- BowVector -------------------------------------------------------------------------
  for (int t = tid; t < P; t += BOWA_THREADS)
    keys[t] = (t < n && featWeight[fo + t] > 0) ? (((uint64_t)featWord[fo + t] << 32) | (unsigned)t)
                                                : ~(uint64_t)0;
  __syncthreads();
  block_sort(keys, P);
  const int R = block_runs(keys, m, runStart, scratch);
  for (int r = tid; r < R; r += BOWA_THREADS) {
    const int s = runStart[r], e = runStart[r + 1];
    // BowVector::addWeight in feature order: the first occurrence stores w, the others add w
    const double w = featWeight[fo + (uint32_t)keys[s]];
    double v = w;
    for (int c = s + 1; c < e; ++c) v += w;
    bowWord[fo + r] = (uint32_t)(keys[s] >> 32);
    bowVal[fo + r] = v;
  }
  __syncthreads();
  // BowVector::normalize(L1): the norm is accumulated in ascending word order, by one thread.
  // The values go through the key array (free until the second sort), eight loads in flight:
  // read one by one from where the other threads just wrote them, this sum was half the kernel.
  double* vals = (double*)keys;
  for (int r = tid; r < R; r += BOWA_THREADS) vals[r] = bowVal[fo + r];
  __syncthreads();
  if (tid == 0) {
    double norm = 0.0;
    int r = 0;
    for (; r + 8 <= R; r += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = vals[r + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) norm += fabs(v[u]);
    }
    for (; r < R; ++r) norm += fabs(vals[r]);
    sNorm = norm;
    bowCount[f] = R;
  }
  __syncthreads();
  const double norm = sNorm;
  if (norm > 0.0)
    for (int r = tid; r < R; r += BOWA_THREADS) bowVal[fo + r] = bowVal[fo + r] / norm;
  __syncthreads();

  // ---- FeatureVector ---------------------------------------------------------------------
  for (int t = tid; t < P; t += BOWA_THREADS)
    keys[t] = (t < n && featWeight[fo + t] > 0) ? (((uint64_t)featNode[fo + t] << 32) | (unsigned)t)
                                                : ~(uint64_t)0;
  __syncthreads();
  block_sort(keys, P);
  const int Rn = block_runs(keys, m, runStart, scratch);
  for (int r = tid; r < Rn; r += BOWA_THREADS) fvNode[fo + r] = (uint32_t)(keys[runStart[r]] >> 32);
  for (int t = tid; t < m; t += BOWA_THREADS) fvFeat[fo + t] = (uint32_t)keys[t];
  if (tid == 0) fvCount[f] = Rn;
}

size_t bow_keys_lds_bytes(int P) { return (size_t)P * 8; }

void launch_bow(const VocDevice& voc, const uint8_t* d_desc, const int* d_n, int cap, int nframes, int levelsup,
                const BowBuffers& b, hipStream_t s) {
  const int perBlock = BOW_THREADS / BOW_LANES;
  hipLaunchKernelGGL(k_bow_descend, dim3((cap + perBlock - 1) / perBlock, nframes), dim3(BOW_THREADS), 0, s, voc,
                     (const uint32_t*)d_desc, d_n, cap, voc.L - levelsup, b.featWord, b.featWeight, b.featNode);
  const size_t lds = b.keyWs ? 0 : bow_keys_lds_bytes(b.P);
  if (b.keyWs)
    hipLaunchKernelGGL(k_bow_assemble<true>, dim3(nframes), dim3(BOWA_THREADS), lds, s, d_n, cap, b.P, b.featWord,
                       b.featWeight, b.featNode, b.keyWs, b.bowWord, b.bowVal, b.bowCount, b.fvNode, b.fvStart,
                       b.fvFeat, b.fvCount);
  else
    hipLaunchKernelGGL(k_bow_assemble<false>, dim3(nframes), dim3(BOWA_THREADS), lds, s, d_n, cap, b.P, b.featWord,
                       b.featWeight, b.featNode, b.keyWs, b.bowWord, b.bowVal, b.bowCount, b.fvNode, b.fvStart,
                       b.fvFeat, b.fvCount);
}

}  // namespace orbx

// ---------------------------------------------------------------------------------------------
// vocabulary object
// ---------------------------------------------------------------------------------------------
struct orbx_vocabulary {
  int k = 0, L = 0, scoring = 0, weighting = 0;
  int nNodes = 0;   // including the root
  int nWords = 0;
  int device = 0;
  void* d_block = nullptr;
  orbx::VocDevice dev;
};

namespace orbx {
const VocDevice* voc_device(const orbx_vocabulary* v, int device) {
  return (v && v->device == device && v->d_block) ? &v->dev : nullptr;
}
}  // namespace orbx

namespace {

// Builds the device form from nodes 1..n in file order (parent, leaf flag, descriptor, weight),
// numbering nodes and words as ORBVocabulary::loadFromText does (vocabulary.cc:130-156).
int build_vocabulary(int k, int L, int scoring, int weighting, int n, const int32_t* parent, const uint8_t* isLeaf,
                     const uint8_t* desc, const double* weight, int device_id, orbx_vocabulary** out) {
  if (!out) return ORBX_ERR_BAD_ARG;
  *out = nullptr;
  if (k < 0 || k > 20 || L < 1 || L > 10 || n < 0 || (n > 0 && (!parent || !isLeaf || !desc || !weight)))
    return ORBX_ERR_BAD_ARG;
  // TF_IDF (0) or TF (1) weighting with L1 scoring (0): what transform() implements here
  if (scoring != 0 || (weighting != 0 && weighting != 1)) return ORBX_ERR_UNSUPPORTED;
  if (n >= (1 << 30)) return ORBX_ERR_UNSUPPORTED;
  const int N = n + 1;
  std::vector<int> childCount(N, 0);
  for (int i = 0; i < n; ++i) {
    if (parent[i] < 0 || parent[i] > i) return ORBX_ERR_BAD_ARG;   // node i+1 must name an earlier node
    childCount[parent[i]]++;
  }
  for (int c : childCount) if (c >= (1 << 20)) return ORBX_ERR_UNSUPPORTED;   // child position field
  std::vector<int> childStart(N + 1, 0);
  for (int i = 0; i < N; ++i) childStart[i + 1] = childStart[i] + childCount[i];
  std::vector<int> cursor(childStart.begin(), childStart.end() - 1);
  std::vector<int> childId(std::max(n, 1));
  std::vector<uint32_t> childDesc((size_t)std::max(n, 1) * 8);
  std::vector<uint32_t> nodeWord(N, 0);
  std::vector<double> nodeWeight(N, 0.0);
  int words = 0;
  for (int i = 0; i < n; ++i) {
    const int nid = i + 1, e = cursor[parent[i]]++;   // children in push_back order
    childId[e] = nid;
    std::memcpy(&childDesc[(size_t)e * 8], desc + (size_t)i * 32, 32);
    nodeWeight[nid] = weight[i];
    if (isLeaf[i] > 0) nodeWord[nid] = (uint32_t)words++;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return ORBX_ERR_NO_DEVICE;
  if (hipSetDevice(device_id) != hipSuccess) return ORBX_ERR_NO_DEVICE;
  orbx_vocabulary* v = new (std::nothrow) orbx_vocabulary();
  if (!v) return ORBX_ERR_BAD_ARG;
  v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
  v->nNodes = N; v->nWords = words; v->device = device_id;
  auto up256 = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t oStart = 0, oId = oStart + up256((size_t)(N + 1) * 4), oDesc = oId + up256(childId.size() * 4),
               oWord = oDesc + up256(childDesc.size() * 4), oWeight = oWord + up256((size_t)N * 4),
               total = oWeight + up256((size_t)N * 8);
  uint8_t* d = nullptr;
  if (hipMalloc((void**)&d, total) != hipSuccess) { delete v; return ORBX_ERR_HIP; }
  v->d_block = d;
  hipError_t e = hipMemcpy(d + oStart, childStart.data(), (size_t)(N + 1) * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + oId, childId.data(), childId.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + oDesc, childDesc.data(), childDesc.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + oWord, nodeWord.data(), (size_t)N * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + oWeight, nodeWeight.data(), (size_t)N * 8, hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(d); delete v; return ORBX_ERR_HIP; }
  v->dev.childStart = (const int*)(d + oStart);
  v->dev.childId = (const int*)(d + oId);
  v->dev.childDesc = (const uint32_t*)(d + oDesc);
  v->dev.nodeWord = (const uint32_t*)(d + oWord);
  v->dev.nodeWeight = (const double*)(d + oWeight);
  v->dev.nNodes = N; v->dev.L = L; v->dev.weighting = weighting;
  *out = v;
  return ORBX_OK;
}

}  // namespace

extern "C" {

int orbx_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                           const uint8_t* is_leaf, const uint8_t* descriptors, const double* weights,
                           int device_id, orbx_vocabulary** out) {
  try {
    return build_vocabulary(k, L, scoring, weighting, n_nodes, parent, is_leaf, descriptors, weights, device_id, out);
  } catch (...) {   // host allocation failure: no exception leaves the C ABI
    return ORBX_ERR_UNSUPPORTED;
  }
}

int orbx_vocabulary_load_text(const char* path, int device_id, orbx_vocabulary** out) {
  if (!path || !out) return ORBX_ERR_BAD_ARG;
  *out = nullptr;
  FILE* f = std::fopen(path, "r");
  if (!f) return ORBX_ERR_BAD_ARG;
  try {
  int k = 0, L = 0, n1 = 0, n2 = 0;
  // header checks of vocabulary.cc:111-117
  if (std::fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 ||
      n1 > 5 || n2 < 0 || n2 > 3) {
    std::fclose(f);
    return ORBX_ERR_BAD_ARG;
  }
  std::vector<int32_t> parent;
  std::vector<uint8_t> leaf, desc;
  std::vector<double> weight;
  bool ok = true;
  for (;;) {
    int pid, isLeaf;
    const int got = std::fscanf(f, "%d %d", &pid, &isLeaf);
    if (got != 2) {
      // only the end of the file (after optional white space, such as a trailing newline) ends the
      // list; anything else that is not a node is a damaged file, not a shorter vocabulary
      ok = got == EOF && std::feof(f);
      break;
    }
    for (int i = 0; i < 32 && ok; ++i) {
      int b;
      ok = std::fscanf(f, "%d", &b) == 1;
      desc.push_back((uint8_t)b);
    }
    double w = 0;
    ok = ok && std::fscanf(f, "%lf", &w) == 1;
    if (!ok) break;
    parent.push_back(pid);
    leaf.push_back(isLeaf > 0 ? 1 : 0);
    weight.push_back(w);
  }
  std::fclose(f);
  f = nullptr;
  if (!ok) return ORBX_ERR_BAD_ARG;
  return build_vocabulary(k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(),
                          device_id, out);
  } catch (...) {   // host allocation failure: no exception leaves the C ABI
    if (f) std::fclose(f);
    return ORBX_ERR_UNSUPPORTED;
  }
}

void orbx_vocabulary_destroy(orbx_vocabulary* v) {
  if (!v) return;
  (void)hipSetDevice(v->device);
  if (v->d_block) (void)hipFree(v->d_block);
  delete v;
}

int orbx_vocabulary_info(const orbx_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words) {
  if (!v) return ORBX_ERR_BAD_ARG;
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (n_nodes) *n_nodes = v->nNodes;
  if (n_words) *n_words = v->nWords;
  return ORBX_OK;
}

}  // extern "C"
