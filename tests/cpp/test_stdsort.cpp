// Host check: orbx_sort::std_sort must produce the same permutation as
// libstdc++'s std::sort for comparators with many equal elements, and its
// heap path must match std::partial_sort(first, last, last).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../orb_slam3_amd/csrc/stdsort_emul.h"

static bool cmp(const uint64_t& a, const uint64_t& b) { return (uint32_t)(a >> 32) < (uint32_t)(b >> 32); }

int main() {
  std::mt19937 rng(12345);
  long cases = 0;
  for (int rep = 0; rep < 40000; ++rep) {
    int n = 1 + rng() % (rep % 50 == 0 ? 5000 : 400);
    int nkeys = 1 + rng() % (rep % 3 == 0 ? 4 : (rep % 3 == 1 ? 40 : 100000));
    std::vector<uint64_t> a(n), b;
    int mode = rng() % 5;
    for (int i = 0; i < n; ++i) {
      uint32_t key = rng() % nkeys;
      if (mode == 1) key = i % nkeys;            // saw-tooth
      if (mode == 2) key = (n - i) % nkeys;      // descending saw
      if (mode == 3) key = (i < n / 2) ? i % nkeys : (n - i) % nkeys;  // organ pipe
      a[i] = ((uint64_t)key << 32) | (uint32_t)i;
    }
    b = a;
    std::sort(a.begin(), a.end(), cmp);
    orbx_sort::std_sort(b.data(), n);
    if (a != b) { std::printf("MISMATCH sort n=%d nkeys=%d mode=%d\n", n, nkeys, mode); return 1; }
    // heap path
    std::vector<uint64_t> c(n), d;
    for (int i = 0; i < n; ++i) c[i] = ((uint64_t)(rng() % nkeys) << 32) | (uint32_t)i;
    d = c;
    std::partial_sort(c.begin(), c.end(), c.end(), cmp);
    orbx_sort::heap_sort_range(d.data(), 0, n);
    if (c != d) { std::printf("MISMATCH heap n=%d\n", n); return 1; }
    ++cases;
  }
  // median-of-3 killer style input to drive the depth limit to zero
  for (int n = 64; n <= 4096; n *= 2) {
    std::vector<uint64_t> a(n), b;
    // classic adversary: v[i] built so that median-of-3 always picks a bad pivot
    int k = n / 2;
    for (int i = 0; i < k; ++i) {
      if (i % 2 == 0) a[i] = ((uint64_t)(i + 1) << 32) | i;
      else a[i] = ((uint64_t)(k + i + (k % 2 == 0 ? 0 : 1)) << 32) | i;
      a[k + i] = ((uint64_t)(2 * (i + 1)) << 32) | (k + i);
    }
    b = a;
    std::sort(a.begin(), a.end(), cmp);
    orbx_sort::std_sort(b.data(), n);
    if (a != b) { std::printf("MISMATCH killer n=%d\n", n); return 1; }
  }
  std::printf("OK %ld\n", cases);
  return 0;
}
