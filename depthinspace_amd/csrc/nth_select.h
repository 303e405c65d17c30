// Which k of n keys does `torch.topk(key, k, largest=False, sorted=False)` return on the CPU when keys tie?
//
// The reference picks Conv3D's 9 neighbours with exactly that call (model/multi_frame_networks.py:498).  ATen's CPU
// kernel (TopKImpl.h, topk_impl_loop) copies the row into a vector of (value, index) pairs and, for k*64 > n, runs
// std::nth_element(begin, begin + k - 1, end) with the comparator below; the result is the first k pairs.  With tied
// keys at the k-th position (masked candidates all carry the same fill value; equidistant planar candidates) the set
// depends on the algorithm, so this header restates libstdc++'s introselect (bits/stl_algo.h: __introselect,
// __unguarded_partition_pivot, __move_median_to_first, __unguarded_partition, __insertion_sort, __heap_select and the
// stl_heap.h helpers) step for step on an array of (key, id).  It is shared by the HIP selection kernel
// (conv3d_knn.hip) and by the host build that tests/test_nth_select.py checks against std::nth_element and torch.topk.
#pragma once

#ifndef NTH_HD
#ifdef __HIPCC__
#define NTH_HD __host__ __device__ __forceinline__
#else
#define NTH_HD inline
#endif
#endif

struct NthPair {
  float key;
  int id;
};

// ATen: ((_isnan(y) && !_isnan(x)) || (x < y))   (NaN sorts last)
NTH_HD bool nth_less(const NthPair& x, const NthPair& y) {
  return ((y.key != y.key) && !(x.key != x.key)) || (x.key < y.key);
}
template <class Q>
NTH_HD void nth_swap(Q q, int a, int b) {
  const NthPair t = q[a];
  q[a] = q[b];
  q[b] = t;
}

template <class Q>
NTH_HD void nth_push_heap(Q q, int first, int hole, int top, NthPair value) {
  int parent = (hole - 1) / 2;
  while (hole > top && nth_less(q[first + parent], value)) {
    q[first + hole] = q[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  q[first + hole] = value;
}
template <class Q>
NTH_HD void nth_adjust_heap(Q q, int first, int hole, int len, NthPair value) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (nth_less(q[first + second], q[first + (second - 1)])) second--;
    q[first + hole] = q[first + second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    q[first + hole] = q[first + (second - 1)];
    hole = second - 1;
  }
  nth_push_heap(q, first, hole, top, value);
}
template <class Q>
NTH_HD void nth_heap_select(Q q, int first, int middle, int last) {
  const int len = middle - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      const NthPair v = q[first + parent];
      nth_adjust_heap(q, first, parent, len, v);
      if (parent == 0) break;
      parent--;
    }
  }
  for (int i = middle; i < last; ++i)
    if (nth_less(q[i], q[first])) {  // __pop_heap(first, middle, i)
      const NthPair v = q[i];
      q[i] = q[first];
      nth_adjust_heap(q, first, 0, len, v);
    }
}

// std::nth_element(q, q + nth, q + n) with the comparator above; Q is NthPair* or anything indexable like it
// (the HIP kernel passes a strided view of an LDS array)
template <class Q>
NTH_HD void nth_element_pairs(Q q, int n, int nth) {
  if (n == 0 || nth == n) return;
  int first = 0, last = n;
  int depth = 0;
  for (int m = n; m > 1; m >>= 1) depth++;  // std::__lg(n)
  depth *= 2;
  while (last - first > 3) {
    if (depth == 0) {
      nth_heap_select(q, first, nth + 1, last);
      nth_swap(q, first, nth);
      return;
    }
    --depth;
    // __unguarded_partition_pivot
    const int mid = first + (last - first) / 2;
    {  // __move_median_to_first(first, first + 1, mid, last - 1)
      const int a = first + 1, b = mid, c = last - 1;
      if (nth_less(q[a], q[b])) {
        if (nth_less(q[b], q[c])) nth_swap(q, first, b);
        else if (nth_less(q[a], q[c])) nth_swap(q, first, c);
        else nth_swap(q, first, a);
      } else if (nth_less(q[a], q[c])) nth_swap(q, first, a);
      else if (nth_less(q[b], q[c])) nth_swap(q, first, c);
      else nth_swap(q, first, b);
    }
    int lo = first + 1, hi = last;
    while (true) {  // __unguarded_partition(first + 1, last, pivot = first)
      while (nth_less(q[lo], q[first])) ++lo;
      --hi;
      while (nth_less(q[first], q[hi])) --hi;
      if (!(lo < hi)) break;
      nth_swap(q, lo, hi);
      ++lo;
    }
    const int cut = lo;
    if (cut <= nth) first = cut;
    else last = cut;
  }
  // __insertion_sort(first, last)
  for (int i = first + 1; i < last; ++i) {
    const NthPair v = q[i];
    if (nth_less(v, q[first])) {
      for (int j = i; j > first; --j) q[j] = q[j - 1];
      q[first] = v;
    } else {
      int j = i;
      while (nth_less(v, q[j - 1])) {
        q[j] = q[j - 1];
        --j;
      }
      q[j] = v;
    }
  }
}
