// Host model of the data-parallel formulation of libstdc++'s std::sort used by the
// quadtree kernel: the quicksort phase is replayed partition by partition, each partition
// computed from the *original* segment contents (stopper lists, number of swaps, cut),
// and the final insertion sort is a stable sort inside each final segment.  Checked
// against the real std::sort.
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>
#include "../../orb_slam3_amd/csrc/stdsort_emul.h"

using orbx_sort::elem_t;
using orbx_sort::less;
static bool cmp(const uint64_t& a, const uint64_t& b) { return (uint32_t)(a >> 32) < (uint32_t)(b >> 32); }

static int partition_model(std::vector<elem_t>& v, int lo, int hi, int pivot) {
  const elem_t P = v[pivot];
  std::vector<int> A, B;
  for (int i = lo; i < hi; ++i) if (!less(v[i], P)) A.push_back(i);
  for (int i = hi - 1; i >= lo; --i) if (!less(P, v[i])) B.push_back(i);
  size_t k = 0;
  while (k < A.size() && k < B.size() && A[k] < B[k]) ++k;
  for (size_t j = 0; j < k; ++j) std::swap(v[A[j]], v[B[j]]);
  const long INF = 1L << 40;
  long a = k < A.size() ? A[k] : INF;
  long bprev = k > 0 ? B[k - 1] : INF;
  return (int)std::min(a, bprev);
}

static bool parsort_model(std::vector<elem_t>& v) {
  const int n = (int)v.size();
  std::vector<int> cuts = {0};
  if (n > 16) {
    struct Seg { int first, last, depth; };
    std::vector<Seg> stk = {{0, n, 2 * orbx_sort::floor_log2(n)}};
    while (!stk.empty()) {
      Seg s = stk.back(); stk.pop_back();
      while (s.last - s.first > 16) {
        if (s.depth == 0) return false;
        --s.depth;
        int mid = s.first + (s.last - s.first) / 2;
        orbx_sort::move_median_to_first(v.data(), s.first, s.first + 1, mid, s.last - 1);
        int cut = partition_model(v, s.first + 1, s.last, s.first);
        stk.push_back({cut, s.last, s.depth});
        cuts.push_back(cut);
        s.last = cut;
      }
    }
  }
  cuts.push_back(n);
  std::sort(cuts.begin(), cuts.end());
  for (size_t i = 0; i + 1 < cuts.size(); ++i)
    std::stable_sort(v.begin() + cuts[i], v.begin() + cuts[i + 1], cmp);
  return true;
}

int main() {
  std::mt19937 rng(777);
  long fallback = 0;
  for (int rep = 0; rep < 60000; ++rep) {
    int n = 1 + rng() % (rep % 40 == 0 ? 4000 : 500);
    int nkeys = 1 + rng() % (rep % 3 == 0 ? 3 : (rep % 3 == 1 ? 30 : 100000));
    int mode = rng() % 5;
    std::vector<uint64_t> a(n), b;
    for (int i = 0; i < n; ++i) {
      uint32_t key = rng() % nkeys;
      if (mode == 1) key = i % nkeys;
      if (mode == 2) key = (n - i) % nkeys;
      if (mode == 3) key = (i < n / 2) ? i % nkeys : (n - i) % nkeys;
      a[i] = ((uint64_t)key << 32) | (uint32_t)i;
    }
    b = a;
    std::sort(a.begin(), a.end(), cmp);
    if (!parsort_model(b)) { ++fallback; continue; }
    if (a != b) { std::printf("MISMATCH n=%d nkeys=%d mode=%d\n", n, nkeys, mode); return 1; }
  }
  std::printf("OK fallbacks=%ld\n", fallback);
  return 0;
}
