// ait_amd/csrc/gemm_internal.h -- library-internal form of ait_gemm_f32 (csrc/gemm_f32.hip) with the epilogue
// options only the library's own composites use (csrc/tail.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

// C = alpha opA(A).opB(B) (+ bias) (+ residual), then zeroed where gate <= 0 (gate: same addressing as C, needs
// `residual`; NULL = ait_gemm_f32 exactly).
int ait_gemm_f32_ex(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                    int ldb, float* C, int ldc, const float* bias, const float* residual, const float* gate, int flags,
                    int split_k, int c_colblk, long long c_batch_stride, const ait_launch_ctx* ctx, void* stream);
