// gel_tables.h -- what every translation unit with kernels shares besides the physics: the fused kernel's workgroup size and the
// staging of the atmosphere / wind / CA tables into LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "gel_device.h"
#include "gel_launch.h"
#include "gel_rhs_parts.h"

namespace gel {

#ifndef GEL_BLOCK
#define GEL_BLOCK 256
#endif
constexpr int kBlock = GEL_BLOCK;  // threads per workgroup of the fused kernel (a multiple of 64)
static_assert(kAtmDoubles == kAtmTableDoubles, "atmosphere table size mismatch between host and device");

GEL_DEV Tables table_view(const double* base, int Kw, int Kc) {
  Tables tb;
  tb.atm = base;
  tb.wind = base + kAtmDoubles;
  tb.ca = tb.wind + 3 * Kw;
  tb.winds = tb.ca + 2 * Kc;
  tb.cas = tb.winds + 2 * (Kw - 1);
  tb.Kw = Kw;
  tb.Kc = Kc;
  return tb;
}

// sync = false: the caller reaches a workgroup barrier of its own before the first table lookup (the cooperative D.X
// forms do), so the copy shares that barrier -- and its memory latency -- with the caller's own first loads
// The cooperative forms of the fused kernel split the copy: stage_tables_issue() requests this thread's table entry at the top of
// the kernel (no wait), stage_tables_commit() writes it to LDS right before the workgroup's first barrier -- behind the state-row
// loads, so the two memory latencies overlap instead of adding up (the entry-to-descriptor stage of a wavefront was 4-7 k cycles).
// Tables longer than the workgroup fall back to the loop at the commit.
GEL_DEV double stage_tables_issue(const ProblemDev& P) {
  const int ntab = table_doubles(P.Kw, P.Kc);
  return P.tables[min((int)threadIdx.x, ntab - 1)];   // no branch around the load (a block boundary the argument fetches behind it would not cross)
}
GEL_DEV void stage_tables_commit(const ProblemDev& P, double* lds, double mine) {
  const int ntab = table_doubles(P.Kw, P.Kc);
  if ((int)threadIdx.x < ntab) lds[threadIdx.x] = mine;
  for (int i = threadIdx.x + kBlock; i < ntab; i += kBlock) lds[i] = P.tables[i];   // (cooperative forms only: kBlock threads -- no look at the dispatch packet)
}
GEL_DEV Tables stage_tables(const ProblemDev& P, double* lds, bool sync = true) {
  const int ntab = table_doubles(P.Kw, P.Kc);
  for (int i = threadIdx.x; i < ntab; i += blockDim.x) lds[i] = P.tables[i];
  if (sync) __syncthreads();
  return table_view(lds, P.Kw, P.Kc);
}

}  // namespace gel
