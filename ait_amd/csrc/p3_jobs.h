// ait_amd/csrc/p3_jobs.h -- the conversion jobs of the P3 operand format (p3_impl.h): plain data + the host entry, for
// the composites that convert their own weights (csrc/transformer.hip) without pulling in the kernels.
#pragma once
#include "common.h"

namespace ait_p3 {

struct Job {
  const float* src;       // [rows][cols], row pitch ld (floats)
  unsigned short* dst;    // transpose == 0: P3 [rows][cols / 8][3][8];  1: P3 of the transpose, [cols][rows / 8][3][8]
  int rows, cols, ld, transpose;
  int first_block;        // blocks of the jobs before this one
};
constexpr int kMaxJobs = 40;
struct Jobs {
  Job j[kMaxJobs];
  int n;
};

// a P3 operand as a product sees it: planes of a [rows][ld values] matrix (6 bytes per value; gemm_p3.hip).  A null
// Ref means "not pre-split" (the product splits the raw weight in registers).  sub(rows, vals): the view that starts
// `rows` rows / `vals` reduction values further.
struct Ref {
  const unsigned short* p = nullptr;
  long long ld = 0;
  Ref sub(long long rows, long long vals) const { return p ? Ref{p + (rows * ld + vals) * 3, ld} : Ref{}; }
};
// the two conversions of one weight W [N_out][K_in]: P3 of W (forward), P3 of W^T (input gradient; training only)
struct Pair { Ref w, wt; };

// enqueue the conversion of jobs.n matrices, one launch (csrc/gemm_p3.hip)
int split(Jobs& jobs, hipStream_t s);

}  // namespace ait_p3
