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

// enqueue the conversion of jobs.n matrices, one launch (csrc/gemm_p3.hip)
int split(Jobs& jobs, hipStream_t s);

}  // namespace ait_p3
