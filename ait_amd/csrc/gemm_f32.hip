// ait_amd/csrc/gemm_f32.hip -- fp32-in / fp32-accumulate GEMM on the gfx950 matrix cores.
//
// Every dense contraction of the AIT path (1x1-conv embeddings, QKV / output projections, the
// position-wise feed-forward, and all of their backward products) runs through this kernel:
//
//     C[M,N] (op)= alpha * opA(A) . opB(B)  [+ bias] [+ residual] [relu]
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulate, no reduced
// precision anywhere (north_star: logits within 1e-4 of the fp32 reference).  Dense peak on
// MI355X: 157.3 TFLOP/s (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz).
//
// Tiling (wave64, not a warp tiling): 128x128 output tile per 256-thread workgroup, 4 waves in
// 2x2, each wave owns 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).  K is consumed in slabs of
// 16: both operand slabs are staged in LDS K-MAJOR ([k][m] / [k][n]), which makes the MFMA
// operand fetch (lane l needs A[m0 + (l&31)][k + (l>>5)]) a conflict-free ds_read_b32 for every
// storage layout of A and B; the layout only changes how the slab is written:
//   operand stored with the reduction dim contiguous ([m][k]): float4 global loads along k,
//       transposed on the way into LDS (4 x ds_write_b32, row pitch 132 -> at most 2-way);
//   operand stored with the reduction dim outermost ([k][m]):  float4 loads along m,
//       ds_write_b128 straight into the K-major slab.
// The next slab's global loads are issued before the 32 MFMAs of the current slab and written
// to the other LDS buffer after them (register-staged double buffering, one barrier per slab);
// ~100 VGPRs and 33 KiB LDS per workgroup leave 4 workgroups per CU to cover barrier bubbles.
// One MFMA takes 64 cycles on its SIMD and needs one ds_read_b32 per operand per 2 MFMAs, so
// the LDS pipe is <15 % busy: the kernel is bound by the matrix pipe, which is the roofline it
// is measured against (bench.py "roofline").
//
// XCD-aware launch: workgroups are dealt round-robin to the 8 XCDs; block ids are remapped so
// that the N-tiles of one M-panel run back to back on ONE XCD (its A panel stays in that
// XCD's L2) while the (small, shared) B operand lives in every L2 / the Infinity Cache.
//
// Split-K (gridDim.z > 1) serves the weight-gradient products (reduction over bp*64 tokens,
// tiny output): partial tiles are combined with one fp32 atomic per element, issued as whole
// 128-B row segments straight from the accumulator layout.
#include "gemm_f32_impl.h"
#include "gemm_internal.h"

namespace {
using namespace ait_gemm;
// Product tiles (measured on MI355X with scripts/gemm_lab.hip / scripts/tune_gemm.py):
//   Tile256D   256x128, FOUR waves owning 128x64 each (8 MFMA tiles per wave, 0.75 operand fetches per MFMA;
//              two workgroups per CU put two waves of DIFFERENT workgroups on every SIMD, so their barrier
//              phases are decoupled), slabs global -> LDS directly, PERSISTENT (the slab ring runs across
//              tile boundaries): every large product with K % 16 == 0.  Measured against the 8-wave
//              (64x64 per wave) form of the same tile: +2..5 % on every layout (scripts/gemm_lab.hip sweep,
//              profiles/r02_gemm_lab_sweep.txt).
//              Products: every f32 operand value split exactly into three bf16 planes in registers, six
//              v_mfma_f32_32x32x16_bf16 per 32x32x16 block (KNOB_SPLIT, gemm_f32_impl.h "f32 products on the bf16
//              matrix pipe"): 185-190 TF/s on the transformer's shapes against 131-136 for the same tile on
//              v_mfma_f32_32x32x2_f32 (Tile256N, selected by AIT_CTX_NATIVE_F32; 157.3 TF/s is that instruction's peak).
//   Tile256    256x128, register-staged three-slab ring: large products whose K is not a multiple of 16.
//   Tile128    128x128 register-staged double buffer: outputs with few rows.
//   TileN64    256x64 for the 64-column SHBlock / fc products (no dead half tile).
//   Tile64     64x64 for few-tile problems (the bs*64-row query side): latency, not throughput.
using Tile256D = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_RNE>;
using Tile256B = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_BF16>;   // operands rounded to bf16, one MFMA per block (AIT_CTX_BF16)
using Tile256N = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPREAD>;      // the same tile on v_mfma_f32_32x32x2_f32 (ait_gemm_f32_products(0))
using Tile256 = Cfg<256, 128, 16, 4, 2, 2, MODE_RING>;
using Tile128 = Cfg<128, 128, 16, 2, 2, 2, MODE_DB>;
using TileN64 = Cfg<256, 64, 16, 4, 1, 2, MODE_DB>;
using TileN64D = Cfg<256, 64, 16, 4, 1, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_RNE>;      // the same 256x64 outputs on the persistent split tile (four waves of 64x64)
using Tile64 = Cfg<64, 64, 16, 2, 2, 2, MODE_DB>;
//   (lab, round 6: the same 64x64 outputs with K-slabs of 64 -- a quarter of the global-load round trips -- measured no
//   faster: 33-35 us for 256x512x1024 either way, profiles/r06_gemm_tail_experiments.txt.  The few-tile launches are bound
//   by one wave's serial MFMA chain, not by the slab round trips.)
//   Tile128S   128x128 split tile, four waves of 64x64, four-slot ring, up to two workgroups per CU: products of a few
//              hundred 256x128 tiles (lab knob mid_tile)
using Tile128S = Cfg<128, 128, 16, 2, 2, 2, MODE_DLDS, 4, KNOB_SPLIT | KNOB_RNE>;
//   TileCoop   256x256, eight waves of 64x128, one workgroup per CU, every operand value split ONCE per workgroup into an LDS
//              plane image (KNOB_COOP): 88 vector instructions per 48 MFMAs against 264.  The weight-gradient products (both
//              operands are activations, K-outer: nothing to pre-split): 203-204 TF/s against 185 for Tile256D on the
//              transformer's shapes (profiles/r04_gemm_lab_coop.txt).  Static work lists (its LDS is the CU's 160 KB).
using TileCoop = Cfg<256, 256, 16, 4, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_COOP | KNOB_NOTICKET | KNOB_RNE>;
}  // namespace

// the weight-gradient launches the cooperative-split tile takes (csrc/transformer.hip sizes its K-splits for the tile)
bool ait_gemm_coop_takes(int trans_a, int trans_b, int M, int N, int K, int flags, const ait_launch_ctx* ctx) {
  if (ctx && (ctx->flags & (AIT_CTX_NATIVE_F32 | AIT_CTX_BF16))) return false;
  return trans_a && !trans_b && (flags & AIT_GEMM_ATOMIC) && M >= 256 && N >= 256 && (M % 4) == 0 && (N % 4) == 0 &&
         K >= 4096 && (K % 16) == 0;
}

// The product entry point with everything the library's own composites may ask for (csrc/gemm_internal.h): `gate`
// (same addressing as C; with `residual`: C = (alpha A.B + bias + residual) zeroed where gate <= 0 -- the input
// gradient of a residual block's ReLU output, formed in the epilogue of the product that completes it).
int ait_gemm_f32_ex(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                    int ldb, float* C, int ldc, const float* bias, const float* residual, const float* gate, int flags,
                    int split_k, int c_colblk, long long c_batch_stride, const ait_launch_ctx* ctx, void* stream) {
  if (M == 0 || N == 0) return (M < 0 || N < 0 || K < 0) ? AIT_EINVAL : AIT_OK;
  GemmArgs g;
  const int rc = make_args(trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, residual,
                           flags, split_k, c_colblk, c_batch_stride, Tile128::BK, g);
  if (rc != AIT_OK) return rc;
  g.gate = gate;
  const SchedWs ws = sched_ws_of(ctx);
  if (ctx && (ctx->flags & AIT_CTX_BF16)) {
    // bf16 products: the persistent tile rounds in registers (below); products it does not take go to the kernel that
    // rounds on the way into LDS (gemm_bf16.hip) -- except the few-tile launches with a gate / column-sum / ReLU-mask
    // epilogue, which stay on the f32 instruction
    const long long t256 = (long long)((M + 255) / 256) * ((N + 127) / 128) * g.splits;
    const bool dir = K > 0 && (K % 16 == 0) && (!trans_a || (M % 4 == 0 && M >= 4)) && (trans_b || (N % 4 == 0 && N >= 4));
    const bool sk = dir && K >= 512 && t256 >= 96 &&
                    ((g.splits == 1 && !(flags & AIT_GEMM_ATOMIC) && ws.p != nullptr) || (flags & AIT_GEMM_ATOMIC));
    const bool persistent = !(N <= 64 && (long long)((M + 255) / 256) * g.splits >= 128) && M >= 512 && (t256 >= 512 || sk) && dir;
    if (!persistent && !gate && !(flags & (AIT_GEMM_COLSUM | AIT_GEMM_MASK_POS)))
      return ait_gemm_bf16(trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, residual, flags, split_k, c_colblk,
                           c_batch_stride, ctx, stream);
  }
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, 2.0 * M * N * K, ait_stream(stream), M, N, K, trans_a, trans_b,
                      g.splits);
  // operand "K-contiguous" means the reduction dimension is the fast one in memory:
  //   A: !trans_a  (A is [M,K]);   B: trans_b (B is [N,K])
  const long long tiles256 = (long long)((M + 255) / 256) * ((N + 127) / 128) * g.splits;
  if (N <= 64 && (long long)((M + 255) / 256) * g.splits >= 128) {
    const bool direct64 = K > 0 && (K % 16 == 0) && N == 64 && (!trans_a || (M % 4 == 0 && M >= 4)) &&
                          !(ctx && (ctx->flags & (AIT_CTX_NATIVE_F32 | AIT_CTX_BF16)));
    if (direct64) return dispatch<TileN64D>(g, !trans_a, trans_b != 0, ait_stream(stream), ws);
    return dispatch<TileN64>(g, !trans_a, trans_b != 0, ait_stream(stream));
  }
  const bool direct = K > 0 && (K % 16 == 0) && (!trans_a || (M % 4 == 0 && M >= 4)) && (trans_b || (N % 4 == 0 && N >= 4));
  // with the stream-K work list (a scheduler workspace) the persistent tile also serves products of a few hundred
  // tiles (their slabs are spread over all workgroups): layer4-sized and co-attention-sized products; split-K
  // launches cut their last round without any scratch
  const bool few_tiles_sk = direct && K >= 512 && tiles256 >= 96 &&
                            ((g.splits == 1 && !(flags & AIT_GEMM_ATOMIC) && ws.p != nullptr) || (flags & AIT_GEMM_ATOMIC));
  if (direct && ait_gemm_coop_takes(trans_a, trans_b, M, N, K, flags, ctx) &&
      (long long)((M + 255) / 256) * ((N + 255) / 256) * g.splits >= 128)
    return launch<TileCoop, false, false, EPI_ATOMIC>(g, ait_stream(stream), ws);
  if constexpr (ait_lab::Knobs::mid_tile != 0) {      // lab: products of fewer than 512 256x128 tiles on the 128x128 split tile
    const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * g.splits;
    if (direct && M >= 512 && tiles256 < 512 && t128 >= 128 && !(ctx && (ctx->flags & (AIT_CTX_NATIVE_F32 | AIT_CTX_BF16))))
      return dispatch<Tile128S>(g, !trans_a, trans_b != 0, ait_stream(stream), ait_lab::Knobs::mid_tile == 2 ? ws : SchedWs());
  }
  if (M >= 512 && (tiles256 >= 512 || few_tiles_sk)) {
    if (direct) {
      if (ctx && (ctx->flags & AIT_CTX_BF16)) return dispatch<Tile256B>(g, !trans_a, trans_b != 0, ait_stream(stream), ws);
      if (ctx && (ctx->flags & AIT_CTX_NATIVE_F32)) return dispatch<Tile256N>(g, !trans_a, trans_b != 0, ait_stream(stream), ws);
      return dispatch<Tile256D>(g, !trans_a, trans_b != 0, ait_stream(stream), ws);
    }
    return dispatch<Tile256>(g, !trans_a, trans_b != 0, ait_stream(stream));
  }
  const long long tiles128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * g.splits;
  if (tiles128 < 128) return dispatch<Tile64>(g, !trans_a, trans_b != 0, ait_stream(stream));
  return dispatch<Tile128>(g, !trans_a, trans_b != 0, ait_stream(stream));
}

AIT_API int ait_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha,
                         const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         const float* bias, const float* residual, int flags, int split_k,
                         int c_colblk, long long c_batch_stride, const ait_launch_ctx* ctx, void* stream) {
  return ait_gemm_f32_ex(trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, residual, nullptr, flags, split_k,
                         c_colblk, c_batch_stride, ctx, stream);
}

// Scheduler scratch of the persistent kernel (include/ait_hip.h, ait_launch_ctx): sized for the widest product tile
// at the occupancy this device admits; the caller allocates it and has its control words zeroed once.
AIT_API size_t ait_gemm_workspace_bytes(void) {
  const void* k = reinterpret_cast<const void*>(gemm_f32_stream_kernel<Tile256D, true, true, EPI_STORE>);
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Tile256D::LDS) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  const int slots = stream_slots<Tile256D>(k);
  // every persistent tile configuration fits: a smaller tile at a higher occupancy never needs more than
  // (max slots) x (largest tile)
  size_t per = (size_t)Tile256D::BM * Tile256D::BN * sizeof(float);
  int s2 = slots;
  const int cap = 4 * (slots / 2 > 0 ? slots / 2 : 1);         // 128x128 tiles: at most 4 per CU
  if ((size_t)cap * 128 * 128 * sizeof(float) > (size_t)s2 * per) { per = 128 * 128 * sizeof(float); s2 = cap; }
  return kCtlBytes + (size_t)s2 * per;
}

AIT_API int ait_gemm_workspace_init(void* sched_ws, size_t sched_ws_bytes, void* stream) {
  if (!sched_ws || sched_ws_bytes < kCtlBytes) return AIT_EINVAL;
  if (hipMemsetAsync(sched_ws, 0, kCtlBytes, ait_stream(stream)) != hipSuccess) return AIT_ELAUNCH;
  return AIT_OK;
}

// Batched form: batch x batch2 independent products C_ij (op)= alpha * opA(A_ij) . opB(B_ij), operand (i, j) at
// base + i*stride + j*stride2 floats, one launch (gridDim.y x gridDim.z).  The image-level co-attention products of the COCO variant
// (lib/model/modules/blocks_coatt_transformer_sk.py:86-110: rel = rho(qry) . phi(img), i2q . emb(qry),
// q2i . emb(img), three torch.matmul over the batch) and their backward.  Few-tile problems: the
// register-staged tiles.
AIT_API int ait_gemm_f32_batched(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, int lda,
                                 long long stride_a, long long stride_a2, const float* B, int ldb, long long stride_b,
                                 long long stride_b2, float* C, int ldc, long long stride_c, long long stride_c2,
                                 int batch, int batch2, int flags, int split_k, const ait_launch_ctx* ctx, void* stream) {
  if (batch < 0 || batch2 < 0 || (flags & ~(AIT_GEMM_ACCUMULATE | AIT_GEMM_ATOMIC))) return AIT_EINVAL;
  if (batch == 0 || batch2 == 0 || M == 0 || N == 0) return (M < 0 || N < 0 || K < 0) ? AIT_EINVAL : AIT_OK;
  if (batch > 65535 || batch2 > 65535 || ((stride_a | stride_b | stride_a2 | stride_b2) & 3)) return AIT_EUNSUPPORTED;
  GemmArgs g;
  const int rc = make_args(trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, C, ldc, nullptr, nullptr, flags, split_k, 0, 0,
                           Tile128::BK, g);
  if (rc != AIT_OK) return rc;
  g.batch = batch; g.sA = stride_a; g.sB = stride_b; g.sC = stride_c;
  g.batch2 = batch2; g.sA2 = stride_a2; g.sB2 = stride_b2; g.sC2 = stride_c2;
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, 2.0 * M * N * K * batch * batch2, ait_stream(stream), M, N, K,
                      trans_a, trans_b, batch * batch2);
  const long long tiles128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * batch * batch2 * g.splits;
  if (tiles128 < 128) return dispatch<Tile64>(g, !trans_a, trans_b != 0, ait_stream(stream));
  return dispatch<Tile128>(g, !trans_a, trans_b != 0, ait_stream(stream));
}
