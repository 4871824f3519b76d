// fx_sort_replay.h — order replay of pcl::EuclideanClusterExtraction's final
//   std::sort(clusters.rbegin(), clusters.rend(), comparePointClusters)
// (called from ref: src/feature_extraction_node.cpp:229 and :276 through PCL).
//
// The comparator is size(a) < size(b) on reverse iterators, i.e. an ascending introsort of
// the REVERSED sequence.  libstdc++'s introsort is not stable, so the order of equally
// sized clusters is a property of the algorithm itself; keypoint ordinals (and through the
// 3DSC RNG stream, descriptor values) depend on it.  This file therefore re-implements the
// algorithm's published structure — median-of-3 quicksort down to 16-element runs with a
// 2*floor(log2 n) depth limit and heap-sort fallback, then one guarded + one unguarded
// insertion pass — as a single sequential routine over packed records that runs on one GPU
// lane (and on the host for the CPU test-suite).  tests/test_sort_replay.py checks the
// permutation against the oracle's real std::sort call.
//
// Records are packed (size << 16) | ordinal; only the size takes part in comparisons.
#ifndef FX_SORT_REPLAY_H_
#define FX_SORT_REPLAY_H_
#include <stdint.h>

#if defined(__HIPCC__)
#define FX_HD __host__ __device__
#else
#define FX_HD
#endif

namespace fx_sort_detail {

// View of the record array through reverse iterators: logical position i is rec[n-1-i].
struct RevView {
  uint32_t *rec;
  int n;
  FX_HD uint32_t get(int i) const { return rec[n - 1 - i]; }
  FX_HD void set(int i, uint32_t v) { rec[n - 1 - i] = v; }
  FX_HD void swap(int i, int j) {
    uint32_t a = get(i), b = get(j);
    set(i, b);
    set(j, a);
  }
};
FX_HD inline bool less_size(uint32_t a, uint32_t b) { return (a >> 16) < (b >> 16); }

template <class V>
FX_HD inline void push_heap(V &v, int first, int hole, int top, uint32_t value) {
  int parent = (hole - 1) / 2;
  while (hole > top && less_size(v.get(first + parent), value)) {
    v.set(first + hole, v.get(first + parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v.set(first + hole, value);
}
template <class V>
FX_HD inline void adjust_heap(V &v, int first, int hole, int len, uint32_t value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less_size(v.get(first + child), v.get(first + (child - 1)))) child--;
    v.set(first + hole, v.get(first + child));
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v.set(first + hole, v.get(first + (child - 1)));
    hole = child - 1;
  }
  push_heap(v, first, hole, top, value);
}
// partial_sort(first, last, last): make_heap + sort_heap over [first, last)
template <class V>
FX_HD inline void heap_sort(V &v, int first, int last) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      uint32_t value = v.get(first + parent);
      adjust_heap(v, first, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  int l = last;
  while (l - first > 1) {
    --l;
    uint32_t value = v.get(l);
    v.set(l, v.get(first));
    adjust_heap(v, first, 0, l - first, value);
  }
}
template <class V>
FX_HD inline void median_to_first(V &v, int result, int a, int b, int c) {
  const uint32_t va = v.get(a), vb = v.get(b), vc = v.get(c);
  if (less_size(va, vb)) {
    if (less_size(vb, vc))
      v.swap(result, b);
    else if (less_size(va, vc))
      v.swap(result, c);
    else
      v.swap(result, a);
  } else if (less_size(va, vc))
    v.swap(result, a);
  else if (less_size(vb, vc))
    v.swap(result, c);
  else
    v.swap(result, b);
}
template <class V>
FX_HD inline int partition_pivot(V &v, int first, int last) {
  const int mid = first + (last - first) / 2;
  median_to_first(v, first, first + 1, mid, last - 1);
  int lo = first + 1, hi = last;
  while (true) {
    const uint32_t pivot = v.get(first);
    while (less_size(v.get(lo), pivot)) ++lo;
    --hi;
    while (less_size(pivot, v.get(hi))) --hi;
    if (!(lo < hi)) return lo;
    v.swap(lo, hi);
    ++lo;
  }
}
template <class V>
FX_HD inline void linear_insert_unguarded(V &v, int last) {
  const uint32_t val = v.get(last);
  int next = last - 1;
  while (less_size(val, v.get(next))) {
    v.set(last, v.get(next));
    last = next;
    --next;
  }
  v.set(last, val);
}
template <class V>
FX_HD inline void insertion_sort(V &v, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less_size(v.get(i), v.get(first))) {
      const uint32_t val = v.get(i);
      for (int j = i; j > first; --j) v.set(j, v.get(j - 1));
      v.set(first, val);
    } else {
      linear_insert_unguarded(v, i);
    }
  }
}

}  // namespace fx_sort_detail

// Phase 1 of std::sort (std::__introsort_loop): median-of-3 quicksort partitions until every
// pending range has <= 16 elements; heap sort for a range whose depth budget ran out.
// Sequential by nature.  stk: FX_SORT_STACK_WORDS ints of scratch.
#define FX_SORT_STACK_WORDS 120
#define FX_SORT_THRESHOLD 16
template <class V>
FX_HD inline void fx_sort_partition_phase(V &v, int n, int *stk) {
  using namespace fx_sort_detail;
  if (n <= FX_SORT_THRESHOLD) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  // explicit stack for the recursion on the right-hand part; pending ranges are disjoint,
  // so the order they are processed in does not change the result
  int *stk_first = stk, *stk_last = stk + 40, *stk_depth = stk + 80;
  int sp = 0;
  stk_first[0] = 0;
  stk_last[0] = n;
  stk_depth[0] = 2 * lg;
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
    while (last - first > FX_SORT_THRESHOLD) {
      if (depth == 0) {
        heap_sort(v, first, last);
        break;
      }
      --depth;
      const int cut = partition_pivot(v, first, last);
      stk_first[sp] = cut;
      stk_last[sp] = last;
      stk_depth[sp] = depth;
      ++sp;
      last = cut;
    }
  }
}

// The same partition step stated without the two walking pointers, which is what lets a wavefront do
// it in a handful of steps (fx_kernels.hip: sort_partition_wave).  After the median has moved to
// `first`, let L_0 < L_1 < ... be the positions in (first, last) holding an element that is NOT
// smaller than the pivot, and R_0 > R_1 > ... the positions in [first, last) holding one that is NOT
// larger (R ends with `first` itself).  The k-th stop of the left pointer is L_k and of the right one
// R_k as long as L_k < R_k — positions between the pointers are untouched by earlier swaps — so the
// loop swaps exactly the pairs (L_k, R_k), k < s, where s is the first k with L_k >= R_k, and returns
// min(L_s, R_{s-1}): the left pointer stops at L_s unless it first meets the element just swapped to
// R_{s-1}.  This routine is the sequential statement of that rule (tests check it against std::sort);
// Lpos / Rpos: scratch for the two position lists.
template <class V>
FX_HD inline int partition_pivot_lists(V &v, int first, int last, uint16_t *Lpos, uint16_t *Rpos) {
  using namespace fx_sort_detail;
  const int mid = first + (last - first) / 2;
  median_to_first(v, first, first + 1, mid, last - 1);
  const uint32_t pivot = v.get(first);
  int nL = 0, nR = 0;
  for (int p = first + 1; p < last; ++p)
    if (!less_size(v.get(p), pivot)) Lpos[nL++] = (uint16_t)p;
  for (int p = last - 1; p >= first; --p)
    if (!less_size(pivot, v.get(p))) Rpos[nR++] = (uint16_t)p;
  int s = 0;
  while (s < nL && s < nR && Lpos[s] < Rpos[s]) ++s;
  for (int k = 0; k < s; ++k) v.swap(Lpos[k], Rpos[k]);
  int cut = 0x7fffffff;
  if (s < nL) cut = Lpos[s];
  if (s >= 1 && Rpos[s - 1] < cut) cut = Rpos[s - 1];
  return cut;
}
// Phase 1 with the list form of the partition step (host test of the rule; n < 65536).
template <class V>
FX_HD inline void fx_sort_partition_phase_lists(V &v, int n, int *stk, uint16_t *Lpos, uint16_t *Rpos) {
  using namespace fx_sort_detail;
  if (n <= FX_SORT_THRESHOLD) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  int *stk_first = stk, *stk_last = stk + 40, *stk_depth = stk + 80;
  int sp = 1;
  stk_first[0] = 0;
  stk_last[0] = n;
  stk_depth[0] = 2 * lg;
  while (sp > 0) {
    --sp;
    int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
    while (last - first > FX_SORT_THRESHOLD) {
      if (depth == 0) {
        heap_sort(v, first, last);
        break;
      }
      --depth;
      const int cut = partition_pivot_lists(v, first, last, Lpos, Rpos);
      stk_first[sp] = cut;
      stk_last[sp] = last;
      stk_depth[sp] = depth;
      ++sp;
      last = cut;
    }
  }
}

// Phase 2 of std::sort (std::__final_insertion_sort): a guarded insertion sort of the first 16
// elements and an unguarded one of the rest.  Both move an element left only past strictly
// greater ones, i.e. together they are a STABLE sort of whatever phase 1 left behind — which is
// why the kernels replace this phase by a parallel stable ranking (cc_order in fx_kernels.hip).
template <class V>
FX_HD inline void fx_sort_insertion_phase(V &v, int n) {
  using namespace fx_sort_detail;
  if (n > FX_SORT_THRESHOLD) {
    insertion_sort(v, 0, FX_SORT_THRESHOLD);
    for (int i = FX_SORT_THRESHOLD; i != n; ++i) linear_insert_unguarded(v, i);
  } else {
    insertion_sort(v, 0, n);
  }
}

// Records in memory, literal two-phase replay: descending size, ties as libstdc++ leaves them.
FX_HD inline void fx_sort_replay_desc(uint32_t *rec, uint32_t n, int *stk) {
  if (n < 2) return;
  fx_sort_detail::RevView v{rec, (int)n};
  fx_sort_partition_phase(v, (int)n, stk);
  fx_sort_insertion_phase(v, (int)n);
}

// Same result with phase 2 as a stable ranking (what the GPU does, O(n^2) here for the test):
// in forward order, position = #larger + #equal-sized records in front.
FX_HD inline void fx_sort_replay_desc_ranked(uint32_t *rec, uint32_t n, int *stk, uint32_t *tmp) {
  if (n < 2) return;
  fx_sort_detail::RevView v{rec, (int)n};
  fx_sort_partition_phase(v, (int)n, stk);
  for (uint32_t c = 0; c < n; ++c) {
    const uint32_t sz = rec[c] >> 16;
    uint32_t pos = 0;
    for (uint32_t d = 0; d < n; ++d) {
      const uint32_t sd = rec[d] >> 16;
      pos += (sd > sz || (sd == sz && d < c)) ? 1u : 0u;
    }
    tmp[pos] = rec[c];
  }
  for (uint32_t c = 0; c < n; ++c) rec[c] = tmp[c];
}

#endif  // FX_SORT_REPLAY_H_
