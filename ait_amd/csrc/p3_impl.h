// ait_amd/csrc/p3_impl.h -- the "P3" operand format of the split-product GEMM and the kernels that produce it.
//
// P3 stores an f32 matrix X [rows][K] (K = the reduction dimension of the product it will enter) as its three bf16
// planes, interleaved in groups of eight values along K:
//     P3[row][K / 8][plane = h, m, l][8]   (bf16)   --   48 bytes per 8 values, 6 bytes per value, row pitch 6 K bytes
// with x = h + m + l EXACTLY (h = bf16(x) rounded to nearest even, m = the top 8 bits of x - h, l = x - h - m: the
// remainders are exact f32 subtractions and l needs at most 8 significant bits; gemm_f32_impl.h split2<true>): P3 is a
// lossless re-encoding of the f32 tensor.  One 16-byte chunk is what one lane of v_mfma_f32_32x32x16_bf16 takes as its eight k-values of one
// plane, so the GEMM moves P3 rows global -> LDS with global_load_lds_dwordx4 and fetches operands with
// ds_read_b128 -- no vector arithmetic on that operand at all (gemm_f32_impl.h, KNOB_BP3).  Weights are converted
// once per step (ait_p3_split / the composites' own multi-tensor pass); their 6 B/value against 4 is irrelevant
// (33 MB of AIT weights against 150-600 MB activations per product).
#pragma once
#include "gemm_f32_impl.h"
#include "p3_jobs.h"

namespace ait_p3 {

// one thread per group of eight reduction values; 256 threads per block
__global__ __launch_bounds__(256) void p3_split_kernel(const Jobs jobs) {
  int ji = 0;
#pragma unroll 1
  for (int i = 1; i < jobs.n; i++)
    if ((int)blockIdx.x >= jobs.j[i].first_block) ji = i;
  // (constant-index copy: a run-time index into the kernel-argument struct sends it to scratch)
  const float* src = jobs.j[0].src;
  unsigned short* dst = jobs.j[0].dst;
  int rows = jobs.j[0].rows, cols = jobs.j[0].cols, ld = jobs.j[0].ld, tr = jobs.j[0].transpose, fb = 0;
#pragma unroll
  for (int i = 1; i < kMaxJobs; i++)
    if (i == ji) {
      src = jobs.j[i].src; dst = jobs.j[i].dst; rows = jobs.j[i].rows; cols = jobs.j[i].cols; ld = jobs.j[i].ld;
      tr = jobs.j[i].transpose; fb = jobs.j[i].first_block;
    }
  const long long t = (long long)(blockIdx.x - fb) * 256 + threadIdx.x;
  float x[8];
  long long out_group;       // index of the 48-byte group in dst
  if (!tr) {
    const int gpr = cols >> 3;                       // groups per row
    const long long row = t / gpr;
    const int g = (int)(t - row * gpr);
    if (row >= rows) return;
    const float4 a = *reinterpret_cast<const float4*>(src + row * ld + g * 8);
    const float4 b = *reinterpret_cast<const float4*>(src + row * ld + g * 8 + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    out_group = t;
  } else {
    // consecutive threads take consecutive columns (coalesced 4-byte reads of eight source rows)
    const long long rg = t / cols;                   // group of eight source rows
    const int c = (int)(t - rg * cols);
    if (rg * 8 >= rows) return;
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = src[(rg * 8 + j) * ld + c];
    out_group = (long long)c * (rows >> 3) + rg;
  }
  ait_gemm::u32x4 h, m, l;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    unsigned a, b, c;
    ait_gemm::split2<true>(x[2 * j], x[2 * j + 1], a, b, c);
    h[j] = a; m[j] = b; l[j] = c;
  }
  ait_gemm::u32x4* o = reinterpret_cast<ait_gemm::u32x4*>(dst + out_group * 24);
  o[0] = h; o[1] = m; o[2] = l;
}

inline int blocks_of(const Job& j) {
  const long long groups = (long long)j.rows * j.cols / 8;
  return (int)((groups + 255) / 256);
}
// enqueue the conversion of jobs.n matrices (one launch)
inline int launch_split(Jobs& jobs, hipStream_t s) {
  if (jobs.n <= 0) return AIT_OK;
  int total = 0;
  for (int i = 0; i < jobs.n; i++) {
    const Job& j = jobs.j[i];
    const int red = j.transpose ? j.rows : j.cols;
    if (!j.src || !j.dst || j.rows <= 0 || j.cols <= 0 || (red & 7) || (j.ld & 3) || (reinterpret_cast<uintptr_t>(j.src) & 15) ||
        (reinterpret_cast<uintptr_t>(j.dst) & 15))
      return AIT_EUNSUPPORTED;
    jobs.j[i].first_block = total;
    total += blocks_of(j);
  }
  hipLaunchKernelGGL(p3_split_kernel, dim3(total), dim3(256), 0, s, jobs);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

}  // namespace ait_p3
