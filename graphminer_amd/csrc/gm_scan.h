// gm_scan.h -- the hipCUB (rocPRIM) exclusive scan over a growable temp buffer; included by the units that run device scans /
// sorts / selections in the untimed setup (gm_graph.hip, gm_tables.hip).
#pragma once
#include "gm_host.h"

#include <hipcub/hipcub.hpp>

template <class TI, class TO>
static hipError_t dev_exclusive_sum(ScanTemp &tmp, const TI *d_in, TO *d_out, size_t n, hipStream_t stream = 0) {
  size_t bytes = 0;
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_in, d_out, (int)n, stream);
  if (e != hipSuccess) return e;
  if ((e = tmp.reserve(bytes)) != hipSuccess) return e;
  return hipcub::DeviceScan::ExclusiveSum(tmp.buf.p, bytes, d_in, d_out, (int)n, stream);
}
