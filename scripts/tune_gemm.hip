// scripts/tune_gemm.hip -- tuning harness (NOT part of the product library): instantiates several
// tile configurations of ait_amd/csrc/gemm_f32_impl.h behind one C entry point.
#include "../ait_amd/csrc/gemm_f32_impl.h"
using namespace ait_gemm;

template <class C>
static int run(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
               float* Cc, int ldc, int flags, int split_k, void* stream) {
  GemmArgs g;
  int rc = make_args(ta, tb, M, N, K, 1.f, A, lda, B, ldb, Cc, ldc, nullptr, nullptr, flags, split_k,
                     0, 0, C::BK, g);
  if (rc) return rc;
  return dispatch<C>(g, !ta, tb != 0, ait_stream(stream));
}

extern "C" __attribute__((visibility("default"))) int tune_gemm(
    int variant, int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B,
    int ldb, float* C, int ldc, int flags, int split_k, void* stream) {
  switch (variant) {
    case 0: return run<Cfg<128, 128, 16, 2, 2, 2>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 1: return run<Cfg<256, 128, 16, 4, 2, 2>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 2: return run<Cfg<256, 128, 16, 4, 2, 2, 2>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 3: return run<Cfg<128, 128, 16, 2, 2, 2, 2>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 4: return run<Cfg<256, 128, 16, 4, 2, 2, 6>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 5: return run<Cfg<256, 128, 16, 4, 2, 2, 14>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 6: return run<Cfg<256, 128, 16, 4, 2, 2, 15>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 22: return run<Cfg<128, 128, 32, 2, 2, 2, 2 + 2048>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 23: return run<Cfg<256, 128, 32, 4, 2, 1, 2 + 2048>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 20: return run<Cfg<256, 128, 16, 2, 2, 2, 6 + 256>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 21: return run<Cfg<256, 256, 16, 2, 4, 1, 6 + 256>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 19: return run<Cfg<256, 128, 16, 4, 2, 2, 6 + 256 + 1024>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 17: return run<Cfg<256, 256, 16, 4, 4, 1, 6 + 256>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 18: return run<Cfg<128, 128, 16, 2, 2, 3, 6 + 256>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 16: return run<Cfg<256, 128, 16, 4, 2, 2, 6 + 256 + 512>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 15: return run<Cfg<256, 128, 16, 4, 2, 2, 6 + 256>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 12: return run<Cfg<256, 128, 16, 4, 2, 2, 6 + 32>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 13: return run<Cfg<256, 128, 16, 4, 2, 2, 6 + 32 + 64>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 14: return run<Cfg<256, 128, 16, 4, 2, 2, 6 + 32 + 64 + 128>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 9: return run<Cfg<256, 128, 16, 2, 2, 2, 6>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 10: return run<Cfg<128, 256, 16, 2, 2, 2, 6>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 11: return run<Cfg<256, 128, 16, 2, 2, 2, 22>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 8: return run<Cfg<256, 128, 16, 4, 2, 2, 22>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    case 7: return run<Cfg<256, 128, 32, 4, 2, 2, 14>>(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, flags, split_k, stream);
    default: return -1;
  }
}
