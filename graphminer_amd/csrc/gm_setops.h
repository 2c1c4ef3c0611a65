// gm_setops.h -- wave64 sorted-set primitives (binary search + ballot compaction).
//
// MI355X re-design of the reference's warp set-op library:
//   binary_search / binary_search_2phase   include/search.cuh:25-35,53-77
//   intersect_num / intersect (+upper)     include/set_intersect.cuh:73-105,152-187,273-298,352,392-424
//   difference_num / difference_set        include/set_difference.cuh:20-35,62-81,112-135,171-196
// Semantics are those of the CPU oracle (Appendix B of SURVEY.md): results are exact sets /
// counts, materialised output is ascending. The "also drop other.vid" quirk of the CPU
// difference (src/common/VertexSet.cc:29,37) is applied by the CALLER through `skip`.
//
// One wave (64 lanes) cooperates on one (A,B) pair: lanes stride the lookup list in tiles of
// 64 (coalesced 256 B loads) and bisect the search list; the 64-bit ballot + v_mbcnt gives the
// compaction rank. All 64 lanes must call these functions together.
#pragma once
#include "gm_wave.h"

namespace gm {

// first index in S[0..len) with S[idx] >= key   (S ascending; any address space)
template <class Ptr>
__device__ __forceinline__ int lower_bound(Ptr S, int len, int key) {
  int lo = 0, hi = len;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    bool lt = S[mid] < key;
    lo = lt ? mid + 1 : lo;
    hi = lt ? hi : mid;
  }
  return lo;
}

// membership test; returns position through *pos (valid only when found)
template <class Ptr>
__device__ __forceinline__ bool contains(Ptr S, int len, int key, int *pos) {
  int lo = 0, hi = len;
  bool found = false;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    int x = S[mid];
    if (x == key) { found = true; lo = mid; break; }
    bool lt = x < key;
    lo = lt ? mid + 1 : lo;
    hi = lt ? hi : mid;
  }
  *pos = lo;
  return found;
}

// |{x in A ^ B}| -- per-lane partial count (sum over the wave for the total)
__device__ __forceinline__ unsigned wave_intersect_num(const int *A, int a, const int *B, int b) {
  const int lane = lane_id();
  const int *L = A, *S = B;
  int nl = a, ns = b;
  if (b < a) { L = B; S = A; nl = b; ns = a; }  // lookup = shorter list (set_intersect.cuh:277-285)
  unsigned cnt = 0;
  for (int i = lane; i < nl; i += GM_WAVE) {
    int pos;
    cnt += contains(S, ns, L[i], &pos) ? 1u : 0u;
  }
  return cnt;
}

// bounded variant: only elements < upper. Both lists are trimmed first (wave-uniform
// bisection, every lane reads the same address -> one broadcast request), which is
// exactly "stop as soon as either cursor reaches upper" (include/VertexSet.h:110-122).
__device__ __forceinline__ unsigned wave_intersect_num_upper(const int *A, int a, const int *B, int b, int upper) {
  a = lower_bound(A, a, upper);
  b = lower_bound(B, b, upper);
  return wave_intersect_num(A, a, B, b);
}

// |{x in A ^ B : x != ex, x != ey}| per-lane partial (intersect_num with two excluded ancestors,
// include/set_intersect.cuh:471 / VertexSet::intersect_ns_except, include/VertexSet.h:178-188)
__device__ __forceinline__ unsigned wave_intersect_num_except2(const int *A, int a, const int *B, int b, int ex, int ey) {
  const int lane = lane_id();
  const int *L = A, *S = B;
  int nl = a, ns = b;
  if (b < a) { L = B; S = A; nl = b; ns = a; }
  unsigned cnt = 0;
  for (int i = lane; i < nl; i += GM_WAVE) {
    int pos;
    const int key = L[i];
    cnt += (contains(S, ns, key, &pos) && key != ex && key != ey) ? 1u : 0u;
  }
  return cnt;
}

// |{x in A ^ B : x < upper, x != ex}| per-lane partial (intersect_num(..., upper, ancestor), set_intersect.cuh:436 /
// VertexSet::intersect_ns_bound_except, include/VertexSet.h:152-164)
__device__ __forceinline__ unsigned wave_intersect_num_upper_except(const int *A, int a, const int *B, int b, int upper, int ex) {
  a = lower_bound(A, a, upper);
  b = lower_bound(B, b, upper);
  return wave_intersect_num_except2(A, a, B, b, ex, ex);
}

// A ^ B -> out (ascending, capacity min(a,b)); returns the size in every lane
__device__ __forceinline__ int wave_intersect_set(const int *A, int a, const int *B, int b, int *out) {
  const int lane = lane_id();
  const int *L = A, *S = B;
  int nl = a, ns = b;
  if (b < a) { L = B; S = A; nl = b; ns = a; }
  int n = 0;
  for (int base = 0; base < nl; base += GM_WAVE) {  // wave-uniform trip count
    int i = base + lane;
    bool f = false;
    int key = 0;
    if (i < nl) {
      int pos;
      key = L[i];
      f = contains(S, ns, key, &pos);
    }
    unsigned long long m = __ballot(f);
    if (f) out[n + rank_below(m)] = key;
    n += __popcll(m);
  }
  return n;
}

__device__ __forceinline__ int wave_intersect_set_upper(const int *A, int a, const int *B, int b, int upper, int *out) {
  a = lower_bound(A, a, upper);
  b = lower_bound(B, b, upper);
  return wave_intersect_set(A, a, B, b, out);
}

// count_smaller (include/operations.cuh:61-105): number of elements of the ascending list a[0..n) that are < bound, as a
// per-lane partial (sum over the wave = the count): lanes stride the list -- the listing form of diamond walks S with it
// (src/sgl/gpu_kernels/diamond_nested.cuh:23-27); lower_bound gives the same number with one lane.
__device__ __forceinline__ unsigned wave_count_smaller(int bound, const int *a, int n) {
  const int lane = lane_id();
  unsigned c = 0;
  for (int i = lane; i < n; i += GM_WAVE) c += (a[i] < bound) ? 1u : 0u;
  return c;
}

// |{x in A : x not in B, x != skip}| per-lane partial. skip = -1 disables the exclusion.
__device__ __forceinline__ unsigned wave_difference_num(const int *A, int a, const int *B, int b, int skip) {
  const int lane = lane_id();
  unsigned cnt = 0;
  for (int i = lane; i < a; i += GM_WAVE) {
    int pos, key = A[i];
    bool in = contains(B, b, key, &pos);
    cnt += (!in && key != skip) ? 1u : 0u;
  }
  return cnt;
}

__device__ __forceinline__ unsigned wave_difference_num_upper(const int *A, int a, const int *B, int b, int skip, int upper) {
  a = lower_bound(A, a, upper);  // elements of B >= upper can never match the kept prefix of A
  return wave_difference_num(A, a, B, b, skip);
}

// A \ B -> out (ascending, capacity a; out may alias A only if the caller owns A)
__device__ __forceinline__ int wave_difference_set(const int *A, int a, const int *B, int b, int skip, int *out) {
  const int lane = lane_id();
  int n = 0;
  for (int base = 0; base < a; base += GM_WAVE) {
    int i = base + lane;
    bool keep = false;
    int key = 0;
    if (i < a) {
      int pos;
      key = A[i];
      keep = !contains(B, b, key, &pos) && key != skip;
    }
    unsigned long long m = __ballot(keep);
    if (keep) out[n + rank_below(m)] = key;
    n += __popcll(m);
  }
  return n;
}

__device__ __forceinline__ int wave_difference_set_upper(const int *A, int a, const int *B, int b, int skip, int upper, int *out) {
  a = lower_bound(A, a, upper);
  return wave_difference_set(A, a, B, b, skip, out);
}

}  // namespace gm
