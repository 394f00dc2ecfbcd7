// Internal interface between the vocabulary / bag-of-words kernels (orbx_bow.hip) and the
// extractor handle (orbx_api.hip).
#ifndef ORBX_BOW_H_
#define ORBX_BOW_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct orbx_vocabulary;

namespace orbx {

// Vocabulary tree in device memory.  Node 0 is the root; the children of node p are entries
// childStart[p] .. childStart[p+1] of childId / childDesc, in the order the file lists them
// (the order of DBoW2's Node::children), so one level's candidates are contiguous.
struct VocDevice {
  const int* childStart;       // nNodes + 1
  const int* childId;          // node id per child entry
  const uint32_t* childDesc;   // 8 words per child entry
  const uint32_t* nodeWord;    // word id of a leaf (0 otherwise)
  const double* nodeWeight;    // Node::weight
  int nNodes, L, weighting;
};

// Per-handle buffers of the transform, [max_batch][cap] unless noted.
struct BowBuffers {
  uint32_t* featWord;          // per feature: word id, weight, node id at level L - levelsup
  double* featWeight;
  uint32_t* featNode;
  uint64_t* keyWs;   // [max_batch][P] sort keys when they do not fit LDS, else null
  int P;                       // power of two >= cap
  uint32_t* bowWord;           // BowVector, ascending word id
  double* bowVal;
  int* bowCount;               // [max_batch]
  uint32_t* fvNode;            // FeatureVector, ascending node id
  int* fvStart;                // [max_batch][cap + 1]
  uint32_t* fvFeat;
  int* fvCount;                // [max_batch]
};

const VocDevice* voc_device(const orbx_vocabulary* v, int device);
size_t bow_keys_lds_bytes(int P);
void launch_bow(const VocDevice& voc, const uint8_t* d_desc, const int* d_n, int cap, int nframes, int levelsup,
                const BowBuffers& b, hipStream_t s);

}  // namespace orbx

#endif
