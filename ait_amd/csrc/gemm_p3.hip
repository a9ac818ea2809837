// ait_amd/csrc/gemm_p3.hip -- products whose WEIGHT operand arrives pre-split ("P3", p3_impl.h) and the pass that
// pre-splits.
//
// Why: the split-product GEMM (gemm_f32_impl.h, KNOB_SPLIT) forms every f32 product from six bf16 partial products
// and was paying 5.5 vector instructions per fetched operand value to split BOTH operands in registers, in every
// wave that touches a value -- a weight value once per 256-row tile, i.e. 300 times per product.  A weight is the
// same for all of them: it is split ONCE per step into P3 rows, moved global -> LDS by the same LDS-DMA stream and
// fetched as ready bf16 planes (three ds_read_b128 per 32x16 operand block, no arithmetic).  With the wave tile
// turned to 64 x 128 (two A blocks, four B blocks) a wave splits two operand blocks per slab instead of six:
// 88 vector instructions per 48 MFMAs against 264.  Tile 256 x 256, eight waves, one workgroup per CU.
//   forward   y  = x W^T : B = P3 of W   [N_out][K_in]
//   dgrad     dx = dy W  : B = P3 of W^T [K_in][N_out]      (the transposed conversion, once per step as well)
// Weight gradients (both operands are activations) stay on the kernel that splits both (gemm_f32.hip).
// Measured (scripts/gemm_lab.hip, profiles/r04_gemm_lab_bp3.txt): 195-208 TFLOP/s on the transformer's forward /
// dgrad shapes against 182-188 for the 256x128 tile that splits both operands, same run.
#include "p3_impl.h"
#include "gemm_internal.h"

namespace {
using namespace ait_gemm;
using TileP3 = Cfg<256, 256, 16, 4, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_BP3 | KNOB_RNE>;

template <class C>
int dispatch_nt(const GemmArgs& g, hipStream_t s, const SchedWs& ws) {
  const bool row_bias = g.bias && (g.flags & AIT_GEMM_BIAS_ROW);
  if ((g.flags & (AIT_GEMM_ATOMIC | AIT_GEMM_ACCUMULATE)) || row_bias) return AIT_EUNSUPPORTED;
  if (g.gate) {
    if (!g.residual || (g.flags & AIT_GEMM_MASK_POS)) return AIT_EINVAL;
    return launch<C, true, true, EPI_RESG>(g, s, ws);
  }
  if (g.residual) return launch<C, true, true, EPI_RES>(g, s, ws);
  return launch<C, true, true, EPI_STORE>(g, s, ws);
}
}  // namespace

bool ait_gemm_p3b_takes(int M, int N, int K, const ait_launch_ctx* ctx) {
  if (ctx && (ctx->flags & (AIT_CTX_NATIVE_F32 | AIT_CTX_BF16))) return false;
  if (!ctx || !ctx->sched_ws) return false;         // (the 256 x 256 tile list needs the stream-K cut of its last round)
  const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
  return M >= 512 && N >= 256 && K >= 128 && (K % 16) == 0 && tiles >= ait_lab::Knobs::p3_min_tiles;
}

int ait_gemm_f32_p3b(int M, int N, int K, float alpha, const float* A, int lda, const void* B_p3, long long ldb_values,
                     float* C, int ldc, const float* bias, const float* residual, const float* gate, int flags,
                     int c_colblk, long long c_batch_stride, const ait_launch_ctx* ctx, void* stream) {
  if (M == 0 || N == 0) return (M < 0 || N < 0 || K < 0) ? AIT_EINVAL : AIT_OK;
  if (!B_p3 || K <= 0 || (K % 16) || (ldb_values % 8) || ldb_values < K || ldb_values * 3 / 2 > 0x7fffffffLL ||
      (reinterpret_cast<uintptr_t>(B_p3) & 15))
    return AIT_EUNSUPPORTED;
  GemmArgs g;
  // (make_args checks A / C / epilogue arguments; B's pitch is in floats of a P3 row: 1.5 per value)
  const int rc = make_args(0, 1, M, N, K, alpha, A, lda, reinterpret_cast<const float*>(B_p3), (int)(ldb_values * 3 / 2), C, ldc,
                           bias, residual, flags, 1, c_colblk, c_batch_stride, 16, g);
  if (rc != AIT_OK) return rc;
  g.gate = gate;
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, 2.0 * M * N * K, ait_stream(stream), M, N, K, 0, 1, 1);
  return dispatch_nt<TileP3>(g, ait_stream(stream), sched_ws_of(ctx));
}

int ait_p3::split(ait_p3::Jobs& jobs, hipStream_t s) { return ait_p3::launch_split(jobs, s); }

// ---- C ABI ---------------------------------------------------------------------------------------------
AIT_API size_t ait_p3_bytes(long long rows, long long cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return (size_t)rows * (size_t)cols * 6;
}

AIT_API int ait_p3_split(const float* src, int rows, int cols, int ld, int transpose, void* dst, void* stream) {
  if (rows < 0 || cols < 0) return AIT_EINVAL;
  if (rows == 0 || cols == 0) return AIT_OK;
  if (!src || !dst) return AIT_EINVAL;
  ait_p3::Jobs jobs;
  jobs.n = 1;
  jobs.j[0] = ait_p3::Job{src, static_cast<unsigned short*>(dst), rows, cols, ld, transpose ? 1 : 0, 0};
  return ait_p3::launch_split(jobs, ait_stream(stream));
}

AIT_API int ait_gemm_f32_p3(int M, int N, int K, float alpha, const float* A, int lda, const void* B_p3, long long ldb_values,
                            float* C, int ldc, const float* bias, const float* residual, int flags, int c_colblk,
                            long long c_batch_stride, const ait_launch_ctx* ctx, void* stream) {
  return ait_gemm_f32_p3b(M, N, K, alpha, A, lda, B_p3, ldb_values, C, ldc, bias, residual, nullptr, flags, c_colblk,
                          c_batch_stride, ctx, stream);
}
