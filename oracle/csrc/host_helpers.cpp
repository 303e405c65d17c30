// TEST INFRASTRUCTURE (tests/ only): host helpers.  (1) vfma: correctly rounded fused multiply-add over arrays, the
// one operation numpy lacks, for tests/bitexact.py.  (2) host build of depthinspace_amd/csrc/nth_select.h (the tie-breaking rule of the HIP neighbour
// selection) next to libstdc++'s own std::nth_element / std::__heap_select, for tests/test_nth_select.py.
#include <algorithm>
#include <utility>
#include <vector>
#include "../../depthinspace_amd/csrc/nth_select.h"

namespace {
struct AtenLess {  // ATen TopKImpl.h comparator (largest = false)
  bool operator()(const std::pair<float, long>& x, const std::pair<float, long>& y) const {
    return ((y.first != y.first) && !(x.first != x.first)) || (x.first < y.first);
  }
};
}  // namespace

extern "C" {
void vfma(const float* a, const float* b, const float* c, float* o, long n) {
  for (long i = 0; i < n; ++i) o[i] = __builtin_fmaf(a[i], b[i], c[i]);
}
// first k ids after nth_element(begin, begin + k - 1, end) over each row of `keys` (rows x n)
void nthsel_rows(const float* keys, long rows, int n, int k, int* out) {
  std::vector<NthPair> q(n);
  for (long r = 0; r < rows; ++r) {
    for (int j = 0; j < n; ++j) q[j] = NthPair{keys[r * n + j], j};
    nth_element_pairs(q.data(), n, k - 1);
    for (int j = 0; j < k; ++j) out[r * k + j] = q[j].id;
  }
}
void stdsel_rows(const float* keys, long rows, int n, int k, int* out) {
  std::vector<std::pair<float, long>> q(n);
  for (long r = 0; r < rows; ++r) {
    for (int j = 0; j < n; ++j) q[j] = {keys[r * n + j], j};
    std::nth_element(q.begin(), q.begin() + k - 1, q.end(), AtenLess());
    for (int j = 0; j < k; ++j) out[r * k + j] = (int)q[j].second;
  }
}
// the depth-limit branch of introselect (never reached by 36 keys in practice): whole-array permutation after
// __heap_select(first, middle, last), ours vs libstdc++'s.  returns the number of rows whose permutations differ
long heapsel_mismatches(const float* keys, long rows, int n, int middle) {
  long bad = 0;
  std::vector<NthPair> q(n);
  std::vector<std::pair<float, long>> s(n);
  for (long r = 0; r < rows; ++r) {
    for (int j = 0; j < n; ++j) {
      q[j] = NthPair{keys[r * n + j], j};
      s[j] = {keys[r * n + j], j};
    }
    nth_heap_select(q.data(), 0, middle, n);
    std::__heap_select(s.begin(), s.begin() + middle, s.end(), __gnu_cxx::__ops::__iter_comp_iter(AtenLess()));
    for (int j = 0; j < n; ++j)
      if (q[j].id != (int)s[j].second) {
        ++bad;
        break;
      }
  }
  return bad;
}
}
