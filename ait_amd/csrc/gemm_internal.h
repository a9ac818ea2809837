// ait_amd/csrc/gemm_internal.h -- library-internal form of ait_gemm_f32 (csrc/gemm_f32.hip) with the epilogue
// options only the library's own composites use (csrc/tail.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

// C = alpha opA(A).opB(B) (+ bias) (+ residual), then zeroed where gate <= 0 (gate: same addressing as C, needs
// `residual`; NULL = ait_gemm_f32 exactly).
int ait_gemm_f32_ex(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                    int ldb, float* C, int ldc, const float* bias, const float* residual, const float* gate, int flags,
                    int split_k, int c_colblk, long long c_batch_stride, const ait_launch_ctx* ctx, void* stream);

// the weight-gradient launches that run on the cooperative-split 256 x 256 tile (csrc/gemm_f32.hip)
bool ait_gemm_coop_takes(int trans_a, int trans_b, int M, int N, int K, int flags, const ait_launch_ctx* ctx);

// Products whose weight operand is pre-split (csrc/gemm_p3.hip).  ait_gemm_p3b_takes: the shapes that path serves
// (others stay on ait_gemm_f32_ex with the raw weight).
bool ait_gemm_p3b_takes(int M, int N, int K, const ait_launch_ctx* ctx);
int ait_gemm_f32_p3b(int M, int N, int K, float alpha, const float* A, int lda, const void* B_p3, long long ldb_values,
                     float* C, int ldc, const float* bias, const float* residual, const float* gate, int flags,
                     int c_colblk, long long c_batch_stride, const ait_launch_ctx* ctx, void* stream);

// dx = conv^T(dy) (+ conv1^T(dy1): a 1x1 stride-2 convolution of the same input) (+ residual / gate) for a STRIDE-2
// convolution, by parity class of the input positions, one launch (csrc/gemm_f32.hip).  AIT_EUNSUPPORTED: not applicable.
int ait_conv_bwd_data_s2(const float* dy, int lddy, const float* w, const ait_conv_geom* q, const float* dy1, int lddy1,
                         const float* w1, int cin, int cout, const float* residual, int flags, float* dx, int lddx,
                         const float* zeros, const ait_launch_ctx* ctx, void* stream);

// the three convolution entries with POSITION-MAJOR rows (pm != 0: GEMM row = position * q->n + map in every activation
// tensor; gemm_f32_impl.h ConvGeom::pm_maps): csrc/tail.hip's layer4.  pm == 0: the public entries exactly.
int ait_conv_fwd_f32_pm(const float* x, int ldx, const float* w, const ait_conv_geom* q, int cin, int cout, const float* bias,
                        const float* residual, int flags, float* y, int ldy, const float* zeros, size_t zeros_floats, int pm,
                        const ait_launch_ctx* ctx, void* stream);
int ait_conv_bwd_data_f32_pm(const float* dy, int lddy, const float* w, const ait_conv_geom* q, int cin, int cout,
                             const float* residual, int flags, float* dx, int lddx, const float* zeros, size_t zeros_floats,
                             int pm, const ait_launch_ctx* ctx, void* stream);
int ait_conv_bwd_weight_f32_pm(const float* dy, int lddy, const float* x, int ldx, const ait_conv_geom* q, int cin, int cout,
                               float* dw, int split_k, const float* zeros, size_t zeros_floats, int pm,
                               const ait_launch_ctx* ctx, void* stream);

// ait_attn_bwd with the three gradients written as bf16 (out_bf16 != 0; pitches in elements) -- csrc/attn_bwd.hip
int ait_attn_bwd_ex(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* P, const float* dO,
                    int n_seq, int H, int T, int d, int kv_rows, float scale, float p_drop, unsigned long long seed, void* dq,
                    int lddq, void* dk, int lddk, void* dv, int lddv, int out_bf16, void* stream);

// ait_mha_core_fwd reading q / k / v from bf16 tensors (qkv_bf16 != 0; pitches in elements) -- csrc/mha_fused.hip
int ait_mha_core_fwd_ex(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int n_seq, int kv_rows,
                        int mask_mode, int n_valid_keys, float scale, float p_attn, unsigned long long seed_attn, const float* sk_w,
                        const float* sk_b, const float* fc_w, const float* residual, const float* ln_g, const float* ln_b, float eps,
                        float p_fc, unsigned long long seed_fc, int out_rows, int q_rep, float* P, float* O, float* u, float* gate,
                        float* s, float* f, float* y, float* mean, float* rstd, int qkv_bf16, void* stream);

// ait_mha_core_bwd with the three gradients written as bf16 (out_bf16 != 0) and / or q / k / v read from bf16 tensors
// (qkv_bf16 != 0); pitches in elements -- csrc/mha_fused_bwd.hip
int ait_mha_core_bwd_ex(const float* df, const float* fc_w, const float* O, const float* gate, const float* sk_w, const void* q,
                        int ldq, const void* k, int ldk, const void* v, int ldv, const float* P, int n_seq, int kv_rows,
                        float scale, float p_attn, unsigned long long seed_attn, void* dq, int lddq, void* dk, int lddk, void* dv,
                        int lddv, float* dg, int out_bf16, int qkv_bf16, void* stream);

// ait_ln_bwd with the gradient `da` also / instead written as bf16 (da16: [rows, 512] bf16 or NULL) -- csrc/rowwise.hip
int ait_ln_bwd_ex(const float* dy, const float* a, const float* pos, const float* residual, const float* gamma,
                  const float* mean, const float* rstd, long long rows, int d, int seq_len, int src_rows_per_seq, int rep,
                  int dy_rows_per_seq, float p_drop, unsigned long long seed, float* da, float* dres, float* dgamma,
                  float* dbeta, float* dcolsum, void* da16, void* stream);

// the caller's scheduler scratch (ait_launch_ctx::sched_ws; layout: csrc/gemm_f32_impl.h): [0, kCtlBytes) the f32 kernel's control
// words, then room for partial tiles -- which the bf16 kernel's last-round cut borrows too (launches ordered on one stream)
namespace ait_ws {
constexpr size_t kCtlBytes = 16384;
}

// ---- bf16-STORAGE products (csrc/gemm_bf16s.hip) with what only the library's composites use: a bf16 addend, an addend AND
// a gate, and operands gathered through a convolution window (csrc/tail.hip's layer4 in the bf16 configuration) -------------
namespace ait_bf16s {
// A stride-1 "same" convolution (odd square window, pad = k / 2) over maps of H x W positions, H * W and W powers of two,
// rows (map, y, x) of a channels-last tensor with cin (a power of two >= 64) channels: the reduction index of the product is
// (tap, channel), a K-slab lies inside one tap and reads the rows of the neighbouring position -- or `zeros` (>= 512 bytes)
// where the window hangs over the edge.
struct Conv {
  int on;
  int hw_mask, w_shift, w_mask, H, W, kw, pad, cin_shift;
  const unsigned short* zeros;
};
// C[M, N] = A[M, K] . B[N, K]^T (+ bias) (+ res32 | res16) (kept where the gate > 0: gate16, or res32 when that is the only
// tensor given) (ReLU) -> C32 and / or C16.  cv.on: A is the map [M, cin] (pitch lda), K = taps * cin, B = [N][taps][cin].
struct Gemm {
  const void* A; const void* B;
  float* C32; void* C16;
  const float* bias; const float* res32; const void* res16; const void* gate16;
  int M, N, K;
  long long lda, ldb, ldc32, ldc16, ldr, ldg;
  bool relu, gate;
  Conv cv;
};
int gemm(const Gemm& p, const ait_launch_ctx* ctx, void* stream);
// C[Mo, No] (f32) += A[R, Mo]^T . B[R, No] over K-ranges of the rows.  cv.on: B is the map [R, cin] (pitch ldb), No = taps * cin,
// column (tap, channel) of row r reads the map row of r's neighbour under that tap (the convolution's weight gradient).
struct Wgrad {
  const void* A; const void* B;
  float* C;
  int Mo, No, R, split_k;
  long long lda, ldb, ldc;
  void* partials; size_t partials_bytes;
  Conv cv;
};
int wgrad(const Wgrad& p, const ait_launch_ctx* ctx, void* stream);
// f32 weights [n_out][taps][cin] (x scale[n_out], may be NULL) -> bf16: as they lie, or (flipped) as the data gradient's
// operand [cin][taps][n_out] with the window mirrored; up to 24 matrices in one launch
struct WeightJob { const float* src; const float* scale; void* dst; int n_out, taps, cin, flipped; };
int fold_weights(const WeightJob* jobs, int n, void* stream);
}  // namespace ait_bf16s
