// Single-thread restatement of libstdc++'s std::sort (introsort: median-of-3
// quicksort with a 2*floor(log2 n) depth budget, heap sort on exhaustion,
// final insertion sort with threshold 16).
//
// The quadtree's last rounds order the nodes to split with std::sort and a
// comparator under which distinct nodes can compare equal
// (core/lib/ORB/extractor.cc:528-541, 673-674), so which of two equal nodes
// is split first depends on the exact sequence of swaps std::sort performs.
// To select the same keypoints the device has to reproduce that permutation;
// this header is that reproduction, usable from host code (unit-tested
// against the real std::sort) and from one lane of a HIP kernel.
//
// Elements are 64-bit: the high 32 bits are the comparison key, the low 32
// bits an opaque payload.  less(a,b) = key(a) < key(b).
#ifndef ORBX_STDSORT_EMUL_H_
#define ORBX_STDSORT_EMUL_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__
#else
#define ORBX_HD
#endif

namespace orbx_sort {

typedef uint64_t elem_t;

ORBX_HD inline bool less(elem_t a, elem_t b) { return (uint32_t)(a >> 32) < (uint32_t)(b >> 32); }

template <typename P>
ORBX_HD inline void swap_at(P v, int i, int j) {
  elem_t t = v[i];
  v[i] = v[j];
  v[j] = t;
}

template <typename P>
ORBX_HD inline void move_median_to_first(P v, int result, int a, int b, int c) {
  if (less(v[a], v[b])) {
    if (less(v[b], v[c])) swap_at(v, result, b);
    else if (less(v[a], v[c])) swap_at(v, result, c);
    else swap_at(v, result, a);
  } else if (less(v[a], v[c])) swap_at(v, result, a);
  else if (less(v[b], v[c])) swap_at(v, result, c);
  else swap_at(v, result, b);
}

template <typename P>
ORBX_HD inline int unguarded_partition(P v, int first, int last, int pivot) {
  for (;;) {
    while (less(v[first], v[pivot])) ++first;
    --last;
    while (less(v[pivot], v[last])) --last;
    if (!(first < last)) return first;
    swap_at(v, first, last);
    ++first;
  }
}

template <typename P>
ORBX_HD inline void push_heap(P v, int first, int hole, int top, elem_t value) {
  int parent = (hole - 1) / 2;
  while (hole > top && less(v[first + parent], value)) {
    v[first + hole] = v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[first + hole] = value;
}

template <typename P>
ORBX_HD inline void adjust_heap(P v, int first, int hole, int len, elem_t value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(v[first + child], v[first + child - 1])) --child;
    v[first + hole] = v[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v[first + hole] = v[first + child - 1];
    hole = child - 1;
  }
  push_heap(v, first, hole, top, value);
}

// std::__partial_sort(first, last, last): make_heap + sort_heap on the range.
template <typename P>
ORBX_HD inline void heap_sort_range(P v, int first, int last) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      elem_t value = v[first + parent];
      adjust_heap(v, first, parent, len, value);
      if (parent == 0) break;
      --parent;
    }
  }
  // heap_select's second loop runs over [middle, last) = empty here.
  int end = last;
  while (end - first > 1) {
    --end;
    elem_t value = v[end];
    v[end] = v[first];
    adjust_heap(v, first, 0, end - first, value);
  }
}

template <typename P>
ORBX_HD inline void unguarded_linear_insert(P v, int last) {
  elem_t val = v[last];
  int next = last - 1;
  while (less(val, v[next])) {
    v[last] = v[next];
    last = next;
    --next;
  }
  v[last] = val;
}

template <typename P>
ORBX_HD inline void insertion_sort(P v, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(v[i], v[first])) {
      elem_t val = v[i];
      for (int k = i; k > first; --k) v[k] = v[k - 1];
      v[first] = val;
    } else {
      unguarded_linear_insert(v, i);
    }
  }
}

ORBX_HD inline int floor_log2(int n) {
  int r = 0;
  while (n > 1) { n >>= 1; ++r; }
  return r;
}

// std::sort(v, v + n) with comparator less().  P is any random-access
// "pointer-like" with operator[] returning an lvalue of elem_t.
template <typename P>
ORBX_HD inline void std_sort(P v, int n) {
  if (n <= 0) return;
  // introsort loop, recursion on the right part replaced by an explicit stack
  // (the two parts are disjoint, so the processing order does not matter).
  int stk_first[64], stk_last[64], stk_depth[64];
  int sp = 0;
  stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = 2 * floor_log2(n);
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
    while (last - first > 16) {
      if (depth == 0) {
        heap_sort_range(v, first, last);
        break;
      }
      --depth;
      int mid = first + (last - first) / 2;
      move_median_to_first(v, first, first + 1, mid, last - 1);
      int cut = unguarded_partition(v, first + 1, last, first);
      stk_first[sp] = cut; stk_last[sp] = last; stk_depth[sp] = depth;
      ++sp;
      last = cut;
    }
  }
  if (n > 16) {
    insertion_sort(v, 0, 16);
    for (int i = 16; i != n; ++i) unguarded_linear_insert(v, i);
  } else {
    insertion_sort(v, 0, n);
  }
}

}  // namespace orbx_sort

#endif  // ORBX_STDSORT_EMUL_H_
