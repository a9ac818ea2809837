// ait_amd/csrc/transformer.hip -- the whole AIT forward as ONE C entry point.
//
// ait_transformer_fwd composes the kernels of this library exactly as ait_amd/system.py composes
// them for inference (Transformer.forward in eval mode, lib/model/system/Models.py:231-280 with
// n_layers = 1, the configuration of faster_rcnn_sys_transformer_sk_dilat.py:148-158):
//
//   tokens  -> enc_emb / dec_emb (1x1 conv = GEMM + bias)             Models.py:246-247
//   encoder :  LN(pad49->64(x) + pos) ; self-attention (key padding) ; selective heads ; fc ;
//              LN(+residual) ; [rows compacted to the n_src real tokens] feed-forward ; LN
//   decoder :  LN(repeat_P(q) + pos) ; causal self-attention block ; cross-attention block over
//              the unpadded encoder memory ; feed-forward block
//   dec_trans (GEMM + bias)                                           Models.py:278
//
// A C / C++ caller (or any FFI) gets the operator without Python; inputs and output are
// token-major (= channels-last feature maps), the layouts ait_roi_align_nhwc_fwd produces and the
// SK / layer4 stage consumes.  All intermediates live in a caller-owned workspace
// (ait_transformer_workspace_bytes); nothing is allocated, nothing is kept between calls.
// Training: ait_transformer_fwd_train saves every activation the backward reads into a caller-owned
// buffer; ait_transformer_bwd runs the whole backward (input gradients written, parameter gradients
// accumulated), with the residual-gradient adds fused into the dgrad GEMM epilogues and the bias
// gradients into the LayerNorm-backward / column-sum kernels.  The two sub-layer blocks are exported
// the same way (ait_mha_block_*, ait_ffn_*).
#include "common.h"
#include "gemm_internal.h"
#include "p3_jobs.h"

namespace {

constexpr int D = 512, DI = 2048, C2 = 1024, H = 8, T = 64, DK = 64;
constexpr float kEps = 1e-6f;   // nn.LayerNorm(d_model, eps=1e-6), SubLayers.py:65 / Models.py:81

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Bump {
  char* p;
  size_t left;
  float* take(size_t floats) {
    const size_t bytes = align_up(floats * sizeof(float), 256);
    if (bytes > left) return nullptr;
    float* r = reinterpret_cast<float*>(p);
    p += bytes;
    left -= bytes;
    return r;
  }
};

#define AIT_TRY(expr)            \
  do {                           \
    const int rc__ = (expr);     \
    if (rc__ != AIT_OK) return rc__; \
  } while (0)

// where a block runs: the launch stream and the caller's launch context (scheduler scratch of the GEMM, probe)
struct Run {
  void* stream;
  const ait_launch_ctx* ctx;
};

// a weight in its pre-split form, and the forward / transposed pair of one weight (csrc/p3_jobs.h)
using P3Ref = ait_p3::Ref;
using P3W = ait_p3::Pair;

// y = x W^T (+ b) (+ relu) on the matrix cores
inline int linear(const float* x, int M, int K, const float* w, int N, const float* b, bool relu, float* y,
                  const Run& s, const P3Ref& p3 = P3Ref()) {
  if (p3.p && ait_gemm_p3b_takes(M, N, K, s.ctx))
    return ait_gemm_f32_p3b(M, N, K, 1.f, x, K, p3.p, p3.ld, y, N, b, nullptr, nullptr, relu ? AIT_GEMM_RELU : 0, 0, 0, s.ctx,
                            s.stream);
  return ait_gemm_f32(0, 1, M, N, K, 1.f, x, K, w, K, y, N, b, nullptr, relu ? AIT_GEMM_RELU : 0, 1, 0, 0, s.ctx, s.stream);
}
// dx [M, K_in] = dy [M, N_out] . W [N_out, K_in]  (+ residual, or gated by `residual` > 0 with mask_pos)
// `colsum` (optional): float[K_in] into which the column sums of dx are ADDED in the product's epilogue (the bias
// gradient of the layer dx flows into)
inline int dgrad(const float* dy, int M, int N_out, const float* w, int K_in, const float* residual, bool mask_pos,
                 float* dx, const Run& s, float* colsum = nullptr, const P3Ref& p3t = P3Ref()) {
  if (p3t.p && ait_gemm_p3b_takes(M, K_in, N_out, s.ctx))      // B = P3 of W^T: rows K_in, reduction over N_out
    return ait_gemm_f32_p3b(M, K_in, N_out, 1.f, dy, N_out, p3t.p, p3t.ld, dx, K_in, colsum, residual, nullptr,
                            (mask_pos ? AIT_GEMM_MASK_POS : 0) | (colsum ? AIT_GEMM_COLSUM : 0), 0, 0, s.ctx, s.stream);
  return ait_gemm_f32(0, 0, M, K_in, N_out, 1.f, dy, N_out, w, K_in, dx, K_in, colsum, residual,
                      (mask_pos ? AIT_GEMM_MASK_POS : 0) | (colsum ? AIT_GEMM_COLSUM : 0), 1, 0, 0, s.ctx, s.stream);
}
// K-splits of a weight gradient [M_out, N_out] = sum over K tokens: multiples of 8 (each XCD owns whole
// K-ranges), chosen so that tiles x splits fills the resident workgroup slots of the 256x128 kernel in
// whole rounds with at least 256 tokens per split (the heuristic ait_amd/system.py used in round 1).
inline int wgrad_splits(int M_out, int N_out, long long K, bool coop) {
  long long tiles, slots;
  if (coop) {
    // the 256 x 256 tile, one per CU.  Its atomic launches cut an under-filled round evenly over the workgroups (stream-K
    // pieces ADD their partial tiles), so "whole rounds" is not what decides: every K-range costs each output element one
    // more atomic add, and the fewest ranges the tile takes (tiles x ranges >= 128: ait_gemm_coop_takes' caller) win --
    // 1536 x 512 x 76800 (the self-attention blocks' W_qkv gradient): 16 ranges 630 us, 32 640, 40 636, 64 (round 5's
    // choice: three whole rounds) 701; the 4- / 8- / 16-tile gradients measure the same for every admissible count
    // (profiles/r06_gemm_tail_experiments.txt)
    // ... PROVIDED the ranges are equal (K a multiple of 16 x ranges): the kernel cuts an under-filled round only then.  The
    // encoder's 58800 rows (1200 x 49) are not: 16 tiles x 8 ranges left half the chip idle (1.05 ms against 0.62 with the 16
    // ranges of the whole-rounds rule) -- those keep the rule below.
    tiles = (long long)((M_out + 255) / 256) * ((N_out + 255) / 256);
    int sp = 8;
    while (sp < 64 && tiles * sp < 128 && K / (sp + 8) >= 256) sp += 8;
    if (K % ((long long)sp * 16) == 0) return sp;
    slots = 256;
  } else if (M_out >= 512) { tiles = (long long)((M_out + 255) / 256) * ((N_out + 127) / 128); slots = 512; }
  else { tiles = (long long)((M_out + 127) / 128) * ((N_out + 127) / 128); slots = 1024; }
  int best = 8;
  double best_eff = -1.0;
  for (int sp = 8; sp < (tiles <= 4 ? 129 : 65); sp += 8) {
    if (K / sp < 256 && sp > 8) break;
    const double rounds = (double)(tiles * sp) / (double)slots;
    const double eff = rounds / (double)((tiles * sp + slots - 1) / slots > 0 ? (tiles * sp + slots - 1) / slots : 1);
    if (eff > best_eff + 1e-9) { best = sp; best_eff = eff; }
  }
  return best;
}
// dW [N_out, K_in] += dy [M, N_out]^T . x [M, K_in]   (split-K, fp32 atomics: accumulates)
inline int wgrad(const float* dy, long long M, int N_out, const float* x, int K_in, float* dw, const Run& s) {
  if (!dw) return AIT_OK;
  const bool coop = M >= 512 && ait_gemm_coop_takes(1, 0, N_out, K_in, (int)M, AIT_GEMM_ATOMIC, s.ctx);
  const int sp = M >= 512 ? wgrad_splits(N_out, K_in, M, coop) : 1;
  return ait_gemm_f32(1, 0, N_out, K_in, (int)M, 1.f, dy, N_out, x, K_in, dw, K_in, nullptr, nullptr, AIT_GEMM_ATOMIC,
                      sp, 0, 0, s.ctx, s.stream);
}

// K-ranges of a bf16 weight-gradient product [Mo, No] over R token rows (ait_gemm_bf16s_tn): enough that tiles x ranges
// fill the chip's 256 workgroup slots of the 256 x 256 tile in whole rounds (64 at most), in whole 32-row slabs per range,
// and -- decisive -- only as many as `scratch_bytes` holds partial tiles for: stored once and reduced by a second launch the
// ranges cost next to nothing (866-906 TFLOP/s on cfg5's feed-forward gradients), added with f32 atomics they cost a third
// of the product at 16 ranges and more beyond (590 / 350 / 250 TFLOP/s at 16 / 32 / 64: profiles/r05_bf16_storage_ffn.txt).
// 0: no split fits the rows.
inline int bf16_tn_split(int Mo, int No, long long R, size_t scratch_bytes) {
  const long long tiles = (long long)(Mo / 256) * (No % 256 == 0 ? No / 256 : No / 128);
  if (tiles <= 0 || R <= 0) return 0;
  // ONE round of the 256 x 256 tile's 256 slots where that is possible (the kernel takes ranges that differ by one 64-row slab:
  // 21 ranges for the 12 tiles of W_qkv's gradient instead of 32 -- a round and a half)
  if (No % 256 == 0 && R % 64 == 0) {
    int sp = (int)(256 / tiles);
    sp = sp < 1 ? 1 : (sp > 64 ? 64 : sp);
    if (R / 64 >= sp && (sp == 1 || R / sp >= 512) && (sp <= 16 || (size_t)sp * Mo * No * sizeof(float) <= scratch_bytes)) return sp;
  }
  int want = 1;
  while (want < 64 && tiles * want < 256) want *= 2;
  for (int sp = want; sp >= 1; sp /= 2) {
    if (R % sp || (R / sp) % 32 || (R / sp < 512 && sp > 1)) continue;
    if (sp > 16 && (size_t)sp * Mo * No * sizeof(float) > scratch_bytes) continue;      // (beyond 16 only without atomics)
    return sp;
  }
  return 0;
}
// buffers of one MultiHeadAttention block: scratch in eval, the saved activations in training
struct MhaBuf {
  float *qkv;          // self: [M, 1536];  cross: q [M, 512] then kv [n*kv_rows, 1024]
  float *P;            // [n, 8, 64, 64] probabilities before dropout (training only)
  float *O, *u, *gate, *s, *f;
  float *mean, *rstd;  // of the closing LayerNorm (training only)
};
inline size_t mha_buf_floats(long long n, int kv_rows, bool cross, bool train) {
  const size_t M = (size_t)n * T;
  size_t f = (cross ? M * D + (size_t)n * kv_rows * 2 * D : M * 3 * D) + M * D + M * DK + (size_t)n * D +
             (size_t)n * DK + M * D;
  if (train) f += (size_t)n * H * T * T + 2 * M;
  return f + 10 * 64;       // 256-B alignment slack per buffer
}
inline bool carve(Bump& b, MhaBuf& m, long long n, int kv_rows, bool cross, bool train) {
  const size_t M = (size_t)n * T;
  m.qkv = b.take(cross ? M * D + (size_t)n * kv_rows * 2 * D : M * 3 * D);
  m.P = train ? b.take((size_t)n * H * T * T) : nullptr;
  m.O = b.take(M * D);
  m.u = b.take(M * DK);
  m.gate = b.take((size_t)n * D);
  m.s = b.take((size_t)n * DK);
  m.f = b.take(M * D);
  m.mean = train ? b.take(M) : nullptr;
  m.rstd = train ? b.take(M) : nullptr;
  return m.qkv && m.O && m.u && m.gate && m.s && m.f && (!train || (m.P && m.mean && m.rstd));
}
struct Qkv { const float *q, *k, *v; int ldq, ldkv; };
inline Qkv views(const MhaBuf& m, long long n, bool cross) {
  if (!cross) return Qkv{m.qkv, m.qkv + D, m.qkv + 2 * D, 3 * D, 3 * D};
  const float* kv = m.qkv + (size_t)n * T * D;
  return Qkv{m.qkv, kv, kv + D, D, 2 * D};
}

// ---- q / k / v STORED in bf16 (AIT_CTX_BF16, round 5) ------------------------------------------------------------------
// Taken when the mode is on, the block has its pre-split-weight scratch (unused in this mode: W_qkv^T and W_qkv as bf16 take
// 4 of its 6 bytes per value) and the rows fill the bf16 kernel's tiles.  The SAME predicate in the forward and the
// backward of a step: the backward reads m.qkv in the format the forward wrote.  Layout inside m.qkv (f32-sized regions, the
// bf16 tensors take the first half of theirs, the bf16 copies of the block's inputs the second):
//   self:  qkv16 [M, 1536] | x16 [M, 512]            cross:  q16 [M, 512] | x16 [M, 512] ... kv16 [R2, 1024] | xkv16 [R2, 512]
inline bool qkv16_on(const Run& s, const P3W& pq, long long M) {
  if (ait_lab::Knobs::no_bf16_qkv) return false;
  return s.ctx && (s.ctx->flags & AIT_CTX_BF16) && pq.w.p && M >= 256;
}
inline unsigned short* qkv_w16(const P3W& pq) { return const_cast<unsigned short*>(pq.w.p) + (size_t)3 * D * D; }
struct Qkv16 { unsigned short *q, *k, *v, *x, *xkv; int ldq, ldkv; };
inline Qkv16 views16(const MhaBuf& m, long long n, long long M, int kv_rows, bool cross) {
  unsigned short* b = reinterpret_cast<unsigned short*>(m.qkv);
  if (!cross) return Qkv16{b, b + D, b + 2 * D, b + (size_t)M * 3 * D, nullptr, 3 * D, 3 * D};
  unsigned short* kv = reinterpret_cast<unsigned short*>(m.qkv + (size_t)n * T * D);
  return Qkv16{b, kv, kv + D, b + (size_t)M * D, kv + (size_t)n * kv_rows * 2 * D, D, 2 * D};
}

// one MultiHeadAttention block (SubLayers.py:68-102 with the selective heads of :22-39):
//   xq [n*64, 512] queries (and residual); keys/values from xkv [n*kv_rows, 512] (xkv == xq: self).
// p_fc: dropout rate behind fc (SubLayers.py:64,97), p_attn: on the probabilities (Modules.py:14,24);
// both 0 in eval.  seed: the block's seed; its two sites derive theirs with ait_dropout_seed.
int mha_block(const float* xq, const float* xkv, int n, int kv_rows, int mask_mode, int n_valid,
              const ait_mha_weights& w, const MhaBuf& m, float p_fc, float p_attn, unsigned long long seed,
              float* y, const Run& s, const P3W& pq = P3W(), int out_rows = T, int q_rep = 1) {
  // q_rep > 1 (inference, cross-attention): xq holds one sequence per q_rep sequences of the block (the query side of
  // a pair, equal for all of its proposals): the query projection and the residual are taken per pair
  const int M = (n / q_rep) * T;
  const bool cross = xkv != xq;
  const bool train = m.mean != nullptr;
  if (qkv16_on(s, pq, M)) {
    // bf16 storage of q / k / v (AIT_CTX_BF16): the block's input(s) and W_qkv are converted once -- the copies of the inputs
    // stay in the second half of m.qkv for the weight gradient of the backward --, the projection runs on the bf16-operand
    // kernel with a bf16 result, and the attention kernel widens what it loads
    const long long R2 = (long long)n * kv_rows;
    unsigned short* w16 = qkv_w16(pq);
    AIT_TRY(ait_f32_to_bf16(w.w_qkv, 3 * D, D, D, w16, D, 0, s.stream));
    const Qkv16 v = views16(m, n, M, kv_rows, cross);
    AIT_TRY(ait_f32_to_bf16(xq, M, D, D, v.x, D, 0, s.stream));
    if (!cross) {
      AIT_TRY(ait_gemm_bf16s(M, 3 * D, D, v.x, D, w16, D, nullptr, 0, v.q, 3 * D, nullptr, nullptr, nullptr, 0, 0, s.ctx, s.stream));
    } else {
      AIT_TRY(ait_f32_to_bf16(xkv, R2, D, D, v.xkv, D, 0, s.stream));
      AIT_TRY(ait_gemm_bf16s(M, D, D, v.x, D, w16, D, nullptr, 0, v.q, D, nullptr, nullptr, nullptr, 0, 0, s.ctx, s.stream));
      AIT_TRY(ait_gemm_bf16s((int)R2, 2 * D, D, v.xkv, D, w16 + (size_t)D * D, D, nullptr, 0, v.k, 2 * D, nullptr, nullptr, nullptr,
                             0, 0, s.ctx, s.stream));
    }
    return ait_mha_core_fwd_ex(v.q, v.ldq, v.k, v.ldkv, v.v, v.ldkv, n, kv_rows, mask_mode, n_valid, 0.125f, p_attn,
                               ait_dropout_seed(seed, 0), w.sk_w, w.sk_b, w.fc_w, xq, w.ln_g, w.ln_b, kEps, p_fc,
                               ait_dropout_seed(seed, 1), out_rows, q_rep, train ? m.P : nullptr, train ? m.O : nullptr,
                               train ? m.u : nullptr, train ? m.gate : nullptr, train ? m.s : nullptr, train ? m.f : nullptr, y,
                               m.mean, m.rstd, 1, s.stream);
  }
  if (!cross) {
    AIT_TRY(linear(xq, M, D, w.w_qkv, 3 * D, nullptr, false, m.qkv, s, pq.w));
  } else {
    AIT_TRY(linear(xq, M, D, w.w_qkv, D, nullptr, false, m.qkv, s, pq.w));
    AIT_TRY(linear(xkv, n * kv_rows, D, w.w_qkv + (size_t)D * D, 2 * D, nullptr, false, m.qkv + (size_t)n * T * D, s, pq.w.sub(D, 0)));
  }
  const Qkv v = views(m, n, cross);
  // attention tiles, selective heads, fc, dropout, residual and the closing LayerNorm: one kernel, all eight heads of a
  // sequence resident (csrc/mha_fused.hip).  Training saves what the backward reads; inference writes nothing but y.
  // (out_rows < 64: only the first out_rows rows of every sequence are written, compacted -- the encoder, whose padded
  // rows are never read again)
  return ait_mha_core_fwd(v.q, v.ldq, v.k, v.ldkv, v.v, v.ldkv, n, kv_rows, mask_mode, n_valid, 0.125f, p_attn,
                          ait_dropout_seed(seed, 0), w.sk_w, w.sk_b, w.fc_w, xq, w.ln_g, w.ln_b, kEps, p_fc,
                          ait_dropout_seed(seed, 1), out_rows, q_rep, train ? m.P : nullptr, train ? m.O : nullptr,
                          train ? m.u : nullptr, train ? m.gate : nullptr, train ? m.s : nullptr, train ? m.f : nullptr, y,
                          m.mean, m.rstd, s.stream);
}

// scratch of the block's backward
struct MhaBwdWs { float *df, *dres, *du, *dO, *dg, *dqkv; };
inline size_t mha_bwd_ws_floats(long long n, int kv_rows, bool cross) {
  const size_t M = (size_t)n * T;
  return M * D * 3 + M * DK + (size_t)n * D + (cross ? M * D + (size_t)n * kv_rows * 2 * D : M * 3 * D) + 6 * 64;
}
inline bool carve(Bump& b, MhaBwdWs& w, long long n, int kv_rows, bool cross) {
  const size_t M = (size_t)n * T;
  w.df = b.take(M * D); w.dres = b.take(M * D); w.du = b.take(M * DK); w.dO = b.take(M * D);
  w.dg = b.take((size_t)n * D);
  w.dqkv = b.take(cross ? M * D + (size_t)n * kv_rows * 2 * D : M * 3 * D);
  return w.df && w.dres && w.du && w.dO && w.dg && w.dqkv;
}

// backward of mha_block.  dy holds dy_rows (<= 64) rows per sequence (the rest received no gradient).
// dxq [n*64, 512] and (cross) dxkv [n*kv_rows, 512] are WRITTEN; parameter gradients are ACCUMULATED.
int mha_block_bwd(const float* dy, int dy_rows, const float* xq, const float* xkv, int n, int kv_rows,
                  const ait_mha_weights& w, const MhaBuf& m, const MhaBwdWs& t, float p_fc, float p_attn,
                  unsigned long long seed, float* dxq, float* dxkv, const ait_mha_grads& g, const Run& s,
                  const P3W& pq = P3W()) {
  const int M = n * T;
  const bool cross = xkv != xq;
  // closing LayerNorm + dropout + residual: df (at fc's output), dres (the residual branch)
  // (bf16 configuration: df also as bf16 and a bf16 copy of u -- into the du / dO scratch nothing else uses since the fused
  // backward kernel -- for d fc_w on the bf16 weight-gradient kernel's 64-column tile: the f32 product spends 0.26 ms per
  // block on splitting 512 x M values for 64 columns of matrix work, 66 TFLOP/s)
  const bool fcw16 = ait_lab::Knobs::fcw_bf16 && g.fc_w && s.ctx && (s.ctx->flags & AIT_CTX_BF16) &&
                     !(s.ctx->flags & AIT_CTX_NATIVE_F32) && M >= 4096 && (M % 32) == 0;
  AIT_TRY(ait_ln_bwd_ex(dy, m.f, nullptr, xq, w.ln_g, m.mean, m.rstd, M, D, T, T, 1, dy_rows, p_fc, ait_dropout_seed(seed, 1), t.df,
                        t.dres, g.ln_g, g.ln_b, nullptr, fcw16 ? static_cast<void*>(t.dO) : nullptr, s.stream));
  if (fcw16) {
    AIT_TRY(ait_f32_to_bf16(m.u, M, DK, DK, t.du, DK, 0, s.stream));
    ait_bf16s::Wgrad p{};
    p.A = t.dO; p.B = t.du; p.C = g.fc_w; p.Mo = D; p.No = DK; p.R = M;
    p.lda = D; p.ldb = DK; p.ldc = DK;
    p.split_k = M / 32 < 128 ? M / 32 : 128;                                 // 2 tiles x 128 ranges: a round of the narrow tile's slots
    p.partials = reinterpret_cast<char*>(t.dO) + (size_t)M * D * 2;          // (the other half of dO)
    p.partials_bytes = (size_t)M * D * 2;
    AIT_TRY(ait_bf16s::wgrad(p, s.ctx, s.stream));                           // d fc_w += df^T u
  } else {
    AIT_TRY(wgrad(t.df, M, D, m.u, DK, g.fc_w, s));                          // d fc_w += df^T u
  }
  const Qkv v = views(m, n, cross);
  // (the forward of this step stored q / k / v -- and bf16 copies of the block's inputs -- in bf16: same predicate)
  const bool in16 = qkv16_on(s, pq, M);
  const Qkv16 v16 = views16(m, n, M, kv_rows, cross);
  // fc's input gradient, the selective heads and the attention tiles backwards: one kernel per sequence
  // (csrc/mha_fused_bwd.hip; du and dO stay on the chip)
  auto attn_back = [&](void* dq_, int lddq, void* dk_, int lddk, void* dv_, int lddv, int out16) -> int {
    if (in16)
      AIT_TRY(ait_mha_core_bwd_ex(t.df, w.fc_w, m.O, m.gate, w.sk_w, v16.q, v16.ldq, v16.k, v16.ldkv, v16.v, v16.ldkv, m.P, n,
                                  kv_rows, 0.125f, p_attn, ait_dropout_seed(seed, 0), dq_, lddq, dk_, lddk, dv_, lddv, t.dg, out16, 1,
                                  s.stream));
    else
      AIT_TRY(ait_mha_core_bwd_ex(t.df, w.fc_w, m.O, m.gate, w.sk_w, v.q, v.ldq, v.k, v.ldkv, v.v, v.ldkv, m.P, n, kv_rows, 0.125f,
                                  p_attn, ait_dropout_seed(seed, 0), dq_, lddq, dk_, lddk, dv_, lddv, t.dg, out16, 0, s.stream));
    AIT_TRY(wgrad(t.dg, n, D, m.s, DK, g.sk_w, s));                            // d sk_w += dg^T s
    if (g.sk_b) AIT_TRY(ait_colsum_f32(t.dg, n, D, D, g.sk_b, s.stream));
    return AIT_OK;
  };
  float* dq = t.dqkv;
  // ---- bf16 storage of the attention gradients (AIT_CTX_BF16, round 5): dq / dk / dv leave the attention kernel as bf16
  // (their only consumers are the two products below), the block's input and the transposed projection weight are
  // converted once, and the input gradient / weight gradient run on the bf16-operand kernels (csrc/gemm_bf16s.hip).
  // Shapes permitting (whole 32-row slabs per K-range of the weight gradients), else the f32-storage products below.
  {
    const long long R2 = (long long)n * kv_rows;
    int sp = 0, sp2 = 0;
    size_t part_bytes = 0;
    if (!ait_lab::Knobs::no_bf16_attn && s.ctx && (s.ctx->flags & AIT_CTX_BF16) && pq.w.p && M >= 256) {
      // (scratch for the K-ranges' partial tiles: t.dqkv behind its bf16 tensors -- with the inputs' bf16 copies left in m.qkv
      // by the forward, its whole second half)
      if (!cross) part_bytes = in16 ? (size_t)M * 3 * D * 2 : (size_t)M * 2 * D * 2;
      else part_bytes = in16 ? ((size_t)M * D + (size_t)R2 * 2 * D) * 2 : (size_t)R2 * 2 * D;
      if (!cross) sp = bf16_tn_split(3 * D, D, M, part_bytes);
      else {
        sp = bf16_tn_split(D, D, M, part_bytes);
        sp2 = bf16_tn_split(2 * D, D, R2, part_bytes);
      }
    }
    if (sp && (!cross || sp2)) {
      unsigned short* wt16 = const_cast<unsigned short*>(pq.w.p);              // W_qkv^T as bf16 [D, 3D] (2 of the 6 carved bytes per value)
      unsigned short* g16 = reinterpret_cast<unsigned short*>(t.dqkv);          // the gradients, bf16, in the first half of t.dqkv
      AIT_TRY(ait_f32_to_bf16(w.w_qkv, 3 * D, D, D, wt16, 3 * D, 1, s.stream));
      if (!cross) {
        unsigned short* x16 = in16 ? v16.x : g16 + (size_t)M * 3 * D;            // [M, D]: the forward's copy, or made here
        void* part = in16 ? g16 + (size_t)M * 3 * D : x16 + (size_t)M * D;
        AIT_TRY(attn_back(g16, 3 * D, g16 + D, 3 * D, g16 + 2 * D, 3 * D, 1));
        if (!in16) AIT_TRY(ait_f32_to_bf16(xq, M, D, D, x16, D, 0, s.stream));
        AIT_TRY(ait_gemm_bf16s(M, D, 3 * D, g16, 3 * D, wt16, 3 * D, dxq, D, nullptr, 0, nullptr, t.dres, nullptr, D, 0, s.ctx,
                               s.stream));                                                                   // dx = dqkv W_qkv + dres
        if (g.w_qkv) AIT_TRY(ait_gemm_bf16s_tn(3 * D, D, M, g16, 3 * D, x16, D, g.w_qkv, D, sp, part, part_bytes, s.ctx, s.stream));
        return AIT_OK;
      }
      unsigned short* dq16 = g16;                                               // [M, D]
      unsigned short* dkv16 = dq16 + (size_t)M * D;                             // [R2, 2D]
      unsigned short* x16 = in16 ? v16.x : dkv16 + (size_t)R2 * 2 * D;          // [M, D]
      unsigned short* xkv16 = in16 ? v16.xkv : x16 + (size_t)M * D;             // [R2, D]
      void* part = in16 ? dkv16 + (size_t)R2 * 2 * D : xkv16 + (size_t)R2 * D;
      AIT_TRY(attn_back(dq16, D, dkv16, 2 * D, dkv16 + D, 2 * D, 1));
      if (!in16) {
        AIT_TRY(ait_f32_to_bf16(xq, M, D, D, x16, D, 0, s.stream));
        AIT_TRY(ait_f32_to_bf16(xkv, R2, D, D, xkv16, D, 0, s.stream));
      }
      AIT_TRY(ait_gemm_bf16s(M, D, D, dq16, D, wt16, 3 * D, dxq, D, nullptr, 0, nullptr, t.dres, nullptr, D, 0, s.ctx, s.stream));
      if (dxkv)
        AIT_TRY(ait_gemm_bf16s((int)R2, D, 2 * D, dkv16, 2 * D, wt16 + D, 3 * D, dxkv, D, nullptr, 0, nullptr, nullptr, nullptr, 0, 0,
                               s.ctx, s.stream));
      if (g.w_qkv) {
        AIT_TRY(ait_gemm_bf16s_tn(D, D, M, dq16, D, x16, D, g.w_qkv, D, sp, part, part_bytes, s.ctx, s.stream));
        AIT_TRY(ait_gemm_bf16s_tn(2 * D, D, (int)R2, dkv16, 2 * D, xkv16, D, g.w_qkv + (size_t)D * D, D, sp2, part, part_bytes,
                                  s.ctx, s.stream));
      }
      return AIT_OK;
    }
  }
  if (!cross) {
    AIT_TRY(attn_back(dq, 3 * D, dq + D, 3 * D, dq + 2 * D, 3 * D, 0));
    AIT_TRY(dgrad(dq, M, 3 * D, w.w_qkv, D, t.dres, false, dxq, s, nullptr, pq.wt));          // dx = dqkv W_qkv + dres
    return wgrad(dq, M, 3 * D, xq, D, g.w_qkv, s);
  }
  float* dkv = t.dqkv + (size_t)M * D;
  AIT_TRY(attn_back(dq, D, dkv, 2 * D, dkv + D, 2 * D, 0));
  AIT_TRY(dgrad(dq, M, D, w.w_qkv, D, t.dres, false, dxq, s, nullptr, pq.wt));
  if (dxkv) AIT_TRY(dgrad(dkv, n * kv_rows, 2 * D, w.w_qkv + (size_t)D * D, D, nullptr, false, dxkv, s, nullptr, pq.wt.sub(0, D)));
  AIT_TRY(wgrad(dq, M, D, xq, D, g.w_qkv, s));
  return wgrad(dkv, (long long)n * kv_rows, 2 * D, xkv, D, g.w_qkv ? g.w_qkv + (size_t)D * D : nullptr, s);
}

// PositionwiseFeedForward (SubLayers.py:177-187) on `rows` token rows; h / f are scratch in eval and the
// saved activations in training (with mean / rstd of the closing LayerNorm)
struct FfnBuf { float *h, *f, *mean, *rstd; };

// ---- the feed-forward block with its 2048-wide tensors STORED in bf16 (AIT_CTX_BF16; BASELINE configs[4]) ------------
// h = relu(x W1^T + b1) and dh exist only as bf16 (in the first half of the f32 buffers the block owns anyway), the two
// 512-wide operands x and df are converted on the way in (a sixth of the bytes of h), the weights and their transposes are
// converted once per forward into the block's pre-split-weight scratch (unused in this mode); all six products run on
// ait_gemm_bf16s / ait_gemm_bf16s_tn (csrc/gemm_bf16s.hip): bf16 operands from memory, f32 accumulate.  Same products as
// the f32-storage bf16 mode (which rounds the same values in registers); the summation order differs.
// Taken when the scratch is there and the shapes fit the bf16 kernels (whole 32-row slabs per split).
struct Bf16Ffn {
  bool on = false;
  int split = 1;
  unsigned short *w1, *w1t, *w2, *w2t;     // [DI, D], [D, DI], [D, DI], [DI, D]
};
inline Bf16Ffn bf16_ffn_plan(long long rows, const Run& s, const P3W& p1, const P3W& p2) {
  Bf16Ffn b;
  if (ait_lab::Knobs::no_bf16_ffn) return b;
  if (!s.ctx || !(s.ctx->flags & AIT_CTX_BF16) || !p1.w.p || !p2.w.p || rows < 256 || rows > 0x7fffffffLL / DI) return b;
  b.split = bf16_tn_split(D, DI, rows, (size_t)rows * (DI - D) * 2);      // (both weight gradients: 16 tiles of 256 x 256)
  b.on = b.split > 0;
  if (!b.on) return b;
  b.w1 = const_cast<unsigned short*>(p1.w.p); b.w1t = b.w1 + (size_t)DI * D;      // (6 bytes per value are carved: 4 used)
  b.w2 = const_cast<unsigned short*>(p2.w.p); b.w2t = b.w2 + (size_t)DI * D;
  return b;
}

// PositionwiseFeedForward (SubLayers.py:177-187) on `rows` token rows; h / f are scratch in eval and the
// saved activations in training (with mean / rstd of the closing LayerNorm)
int ffn_block(const float* x, long long rows, const ait_ffn_weights& w, const FfnBuf& m, float p,
              unsigned long long seed, float* y, const Run& s, const P3W& p1 = P3W(), const P3W& p2 = P3W()) {
  const Bf16Ffn b = bf16_ffn_plan(rows, s, p1, p2);
  if (b.on) {
    unsigned short* h16 = reinterpret_cast<unsigned short*>(m.h);                 // [rows, DI] bf16: the first half of m.h
    unsigned short* x16 = h16 + (size_t)rows * DI;                                // [rows, D] bf16, in the second half
    AIT_TRY(ait_f32_to_bf16(w.w1, DI, D, D, b.w1, D, 0, s.stream));
    AIT_TRY(ait_f32_to_bf16(w.w2, D, DI, DI, b.w2, DI, 0, s.stream));
    if (p1.wt.p) {          // training: the transposes the backward multiplies by
      AIT_TRY(ait_f32_to_bf16(w.w1, DI, D, D, b.w1t, DI, 1, s.stream));
      AIT_TRY(ait_f32_to_bf16(w.w2, D, DI, DI, b.w2t, D, 1, s.stream));
    }
    AIT_TRY(ait_f32_to_bf16(x, rows, D, D, x16, D, 0, s.stream));
    AIT_TRY(ait_gemm_bf16s((int)rows, DI, D, x16, D, b.w1, D, nullptr, 0, h16, DI, w.b1, nullptr, nullptr, 0, AIT_GEMM_RELU,
                           s.ctx, s.stream));
    AIT_TRY(ait_gemm_bf16s((int)rows, D, DI, h16, DI, b.w2, DI, m.f, D, nullptr, 0, w.b2, nullptr, nullptr, 0, 0, s.ctx,
                           s.stream));
  } else {
    AIT_TRY(linear(x, (int)rows, D, w.w1, DI, w.b1, true, m.h, s, p1.w));
    AIT_TRY(linear(m.h, (int)rows, DI, w.w2, D, w.b2, false, m.f, s, p2.w));
  }
  return ait_ln_fwd(m.f, nullptr, x, w.ln_g, w.ln_b, rows, D, T, T, 1, kEps, p, ait_dropout_seed(seed, 0), y, m.mean,
                    m.rstd, s.stream);
}
struct FfnBwdWs { float *df, *dres, *dh; };
int ffn_block_bwd(const float* dy, const float* x, long long rows, const ait_ffn_weights& w, const FfnBuf& m,
                  const FfnBwdWs& t, float p, unsigned long long seed, float* dx, const ait_ffn_grads& g, const Run& s,
                  const P3W& p1 = P3W(), const P3W& p2 = P3W()) {
  const int R = (int)rows;
  const Bf16Ffn b = bf16_ffn_plan(rows, s, p1, p2);
  const bool b16 = b.on && p1.wt.p;
  // df at w_2's output (its column sums are d b2), dres on the residual branch; in the bf16-storage mode df is written
  // as bf16 only (its consumers are the two bf16 products below)
  unsigned short* df16_ = b16 ? reinterpret_cast<unsigned short*>(t.dh) + (size_t)rows * DI : nullptr;
  AIT_TRY(ait_ln_bwd_ex(dy, m.f, nullptr, x, w.ln_g, m.mean, m.rstd, rows, D, T, T, 1, T, p, ait_dropout_seed(seed, 0),
                        b16 ? nullptr : t.df, t.dres, g.ln_g, g.ln_b, g.b2, df16_, s.stream));
  if (b16) {
    // (the forward of this step stored h and the weight copies in bf16: same plan, same predicate)
    const unsigned short* h16 = reinterpret_cast<const unsigned short*>(m.h);
    const unsigned short* x16 = h16 + (size_t)rows * DI;                          // the forward's bf16 copy of x, still there
    unsigned short* dh16 = reinterpret_cast<unsigned short*>(t.dh);               // [rows, DI] bf16: first half of t.dh
    const unsigned short* df16 = df16_;                                           // [rows, D]: in the second half of t.dh
    // (scratch for the K-ranges' partial tiles of the two weight gradients: t.dh's second half behind df16)
    void* part = const_cast<unsigned short*>(df16) + (size_t)rows * D;
    const size_t part_bytes = (size_t)rows * (DI - D) * 2;
    if (g.w2) AIT_TRY(ait_gemm_bf16s_tn(D, DI, R, df16, D, h16, DI, g.w2, DI, b.split, part, part_bytes, s.ctx, s.stream));   // d W2 += df^T h
    AIT_TRY(ait_gemm_bf16s(R, DI, D, df16, D, b.w2t, D, nullptr, 0, dh16, DI, nullptr, nullptr, h16, DI, AIT_GEMM_MASK_POS,
                           s.ctx, s.stream));                                                                   // dh = (df W2) [h > 0]
    if (g.b1) AIT_TRY(ait_colsum_bf16(dh16, rows, DI, DI, g.b1, s.stream));                                    // d b1
    if (g.w1) AIT_TRY(ait_gemm_bf16s_tn(DI, D, R, dh16, DI, x16, D, g.w1, D, b.split, part, part_bytes, s.ctx, s.stream));    // d W1 += dh^T x
    return ait_gemm_bf16s(R, D, DI, dh16, DI, b.w1t, DI, dx, D, nullptr, 0, nullptr, t.dres, nullptr, D, 0, s.ctx,
                          s.stream);                                                                            // dx = dh W1 + dres
  }
  AIT_TRY(wgrad(t.df, rows, D, m.h, DI, g.w2, s));                             // d W2 += df^T h
  AIT_TRY(dgrad(t.df, R, D, w.w2, DI, m.h, true, t.dh, s, g.b1, p2.wt));      // dh = (df W2) [h > 0];  d b1 += column sums
  AIT_TRY(wgrad(t.dh, rows, DI, x, D, g.w1, s));                               // d W1 += dh^T x
  return dgrad(t.dh, R, DI, w.w1, D, t.dres, false, dx, s, nullptr, p1.wt);    // dx = dh W1 + dres
}

}  // namespace

// Pure function: the seed of dropout site `site` under base seed `base` (splitmix64 finaliser).  The
// training entry points derive every site's seed with it; a caller that composes the building blocks
// itself (ait_amd/system.py's fine-grained path) uses the same function and gets the same masks.
AIT_API unsigned long long ait_dropout_seed(unsigned long long base, int site) {
  unsigned long long z = base + 0x9E3779B97F4A7C15ull * (unsigned long long)(site + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---------------------------------------------------------------------------------------------------
// sub-layer blocks: inference
// ---------------------------------------------------------------------------------------------------
AIT_API size_t ait_mha_block_workspace_bytes(int n_seq, int kv_rows) {
  if (n_seq <= 0 || kv_rows <= 0 || kv_rows > T) return 0;
  return mha_buf_floats(n_seq, kv_rows, true, false) * sizeof(float) + (size_t)n_seq * T * 2 * D * sizeof(float);
}

static int check_mha(const float* xq, const float*& xkv, int n_seq, int kv_rows, int mask_mode, const void* w) {
  if (n_seq < 0 || kv_rows <= 0 || kv_rows > T || mask_mode < 0 || mask_mode > 2 || !w) return AIT_EINVAL;
  if ((long long)n_seq * T > 0x7fffffffLL) return AIT_EUNSUPPORTED;
  if (n_seq > 0 && !xq) return AIT_EINVAL;
  if (!xkv || xkv == xq) {
    if (kv_rows != T) return AIT_EINVAL;       // self-attention: keys are the 64 query tokens
    xkv = xq;
  }
  return AIT_OK;
}

AIT_API int ait_mha_block_fwd(const float* xq, const float* xkv, int n_seq, int kv_rows, int mask_mode,
                              int n_valid_keys, const ait_mha_weights* w, void* workspace, size_t workspace_bytes,
                              float* y, const ait_launch_ctx* ctx, void* stream) {
  AIT_TRY(check_mha(xq, xkv, n_seq, kv_rows, mask_mode, w));
  if (n_seq == 0) return AIT_OK;
  if (!y || !workspace) return AIT_EINVAL;
  if (workspace_bytes < ait_mha_block_workspace_bytes(n_seq, kv_rows)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(workspace), workspace_bytes};
  MhaBuf m;
  if (!carve(b, m, n_seq, kv_rows, xkv != xq, false)) return AIT_EWORKSPACE;
  return mha_block(xq, xkv, n_seq, kv_rows, mask_mode, n_valid_keys, *w, m, 0.f, 0.f, 0, y, Run{stream, ctx});
}

AIT_API size_t ait_ffn_workspace_bytes(long long rows) {
  if (rows <= 0) return 0;
  return ((size_t)rows * DI + (size_t)rows * D) * sizeof(float) + 4 * 256;
}

AIT_API int ait_ffn_fwd(const float* x, long long rows, const ait_ffn_weights* w, void* workspace,
                        size_t workspace_bytes, float* y, const ait_launch_ctx* ctx, void* stream) {
  if (rows < 0 || rows > 0x7fffffffLL || !w) return AIT_EINVAL;
  if (rows == 0) return AIT_OK;
  if (!x || !y || !workspace) return AIT_EINVAL;
  if (workspace_bytes < ait_ffn_workspace_bytes(rows)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(workspace), workspace_bytes};
  FfnBuf m{b.take((size_t)rows * DI), b.take((size_t)rows * D), nullptr, nullptr};
  if (!m.h || !m.f) return AIT_EWORKSPACE;
  return ffn_block(x, rows, *w, m, 0.f, 0, y, Run{stream, ctx});
}

// ---------------------------------------------------------------------------------------------------
// sub-layer blocks: training (forward that saves, backward)
// ---------------------------------------------------------------------------------------------------
static bool bad_p(float p) { return !(p >= 0.f && p < 1.f); }

AIT_API size_t ait_mha_block_saved_bytes(int n_seq, int kv_rows) {
  if (n_seq <= 0 || kv_rows <= 0 || kv_rows > T) return 0;
  return (mha_buf_floats(n_seq, kv_rows, true, true) + (size_t)n_seq * T * 2 * D) * sizeof(float);
}

AIT_API int ait_mha_block_fwd_train(const float* xq, const float* xkv, int n_seq, int kv_rows, int mask_mode,
                                    int n_valid_keys, const ait_mha_weights* w, float p_drop, float p_attn_drop,
                                    unsigned long long seed, void* saved, size_t saved_bytes, float* y,
                                    const ait_launch_ctx* ctx, void* stream) {
  AIT_TRY(check_mha(xq, xkv, n_seq, kv_rows, mask_mode, w));
  if (bad_p(p_drop) || bad_p(p_attn_drop)) return AIT_EINVAL;
  if (n_seq == 0) return AIT_OK;
  if (!y || !saved) return AIT_EINVAL;
  if (saved_bytes < ait_mha_block_saved_bytes(n_seq, kv_rows)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(saved), saved_bytes};
  MhaBuf m;
  if (!carve(b, m, n_seq, kv_rows, xkv != xq, true)) return AIT_EWORKSPACE;
  return mha_block(xq, xkv, n_seq, kv_rows, mask_mode, n_valid_keys, *w, m, p_drop, p_attn_drop, seed, y, Run{stream, ctx});
}

AIT_API size_t ait_mha_block_bwd_workspace_bytes(int n_seq, int kv_rows) {
  if (n_seq <= 0 || kv_rows <= 0 || kv_rows > T) return 0;
  return (mha_bwd_ws_floats(n_seq, kv_rows, true) + (size_t)n_seq * T * 2 * D) * sizeof(float);
}

AIT_API int ait_mha_block_bwd(const float* dy, const float* xq, const float* xkv, int n_seq, int kv_rows,
                              int mask_mode, int n_valid_keys, const ait_mha_weights* w, float p_drop,
                              float p_attn_drop, unsigned long long seed, const void* saved, size_t saved_bytes,
                              void* workspace, size_t workspace_bytes, float* dxq, float* dxkv,
                              const ait_mha_grads* grads, const ait_launch_ctx* ctx, void* stream) {
  (void)n_valid_keys;     // masked probabilities are exactly 0 in the saved P: the backward needs no mask
  AIT_TRY(check_mha(xq, xkv, n_seq, kv_rows, mask_mode, w));
  if (bad_p(p_drop) || bad_p(p_attn_drop) || !grads) return AIT_EINVAL;
  if (n_seq == 0) return AIT_OK;
  if (!dy || !dxq || !saved || !workspace) return AIT_EINVAL;
  if (saved_bytes < ait_mha_block_saved_bytes(n_seq, kv_rows)) return AIT_EWORKSPACE;
  if (workspace_bytes < ait_mha_block_bwd_workspace_bytes(n_seq, kv_rows)) return AIT_EWORKSPACE;
  const bool cross = xkv != xq;
  Bump b{static_cast<char*>(const_cast<void*>(saved)), saved_bytes};
  MhaBuf m;
  if (!carve(b, m, n_seq, kv_rows, cross, true)) return AIT_EWORKSPACE;
  Bump bw{static_cast<char*>(workspace), workspace_bytes};
  MhaBwdWs t;
  if (!carve(bw, t, n_seq, kv_rows, cross)) return AIT_EWORKSPACE;
  return mha_block_bwd(dy, T, xq, xkv, n_seq, kv_rows, *w, m, t, p_drop, p_attn_drop, seed, dxq, dxkv, *grads,
                       Run{stream, ctx});
}

AIT_API size_t ait_ffn_saved_bytes(long long rows) {
  if (rows <= 0) return 0;
  return ((size_t)rows * (DI + D + 2)) * sizeof(float) + 6 * 256;
}
static bool carve_ffn(Bump& b, FfnBuf& m, long long rows) {
  m.h = b.take((size_t)rows * DI); m.f = b.take((size_t)rows * D);
  m.mean = b.take((size_t)rows); m.rstd = b.take((size_t)rows);
  return m.h && m.f && m.mean && m.rstd;
}

AIT_API int ait_ffn_fwd_train(const float* x, long long rows, const ait_ffn_weights* w, float p_drop,
                              unsigned long long seed, void* saved, size_t saved_bytes, float* y,
                              const ait_launch_ctx* ctx, void* stream) {
  if (rows < 0 || rows > 0x7fffffffLL || !w || bad_p(p_drop)) return AIT_EINVAL;
  if (rows == 0) return AIT_OK;
  if (!x || !y || !saved) return AIT_EINVAL;
  if (saved_bytes < ait_ffn_saved_bytes(rows)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(saved), saved_bytes};
  FfnBuf m;
  if (!carve_ffn(b, m, rows)) return AIT_EWORKSPACE;
  return ffn_block(x, rows, *w, m, p_drop, seed, y, Run{stream, ctx});
}

AIT_API size_t ait_ffn_bwd_workspace_bytes(long long rows) {
  if (rows <= 0) return 0;
  return ((size_t)rows * (2 * D + DI)) * sizeof(float) + 4 * 256;
}
static bool carve_ffn_ws(Bump& b, FfnBwdWs& t, long long rows) {
  t.df = b.take((size_t)rows * D); t.dres = b.take((size_t)rows * D); t.dh = b.take((size_t)rows * DI);
  return t.df && t.dres && t.dh;
}

AIT_API int ait_ffn_bwd(const float* dy, const float* x, long long rows, const ait_ffn_weights* w, float p_drop,
                        unsigned long long seed, const void* saved, size_t saved_bytes, void* workspace,
                        size_t workspace_bytes, float* dx, const ait_ffn_grads* grads, const ait_launch_ctx* ctx,
                        void* stream) {
  if (rows < 0 || rows > 0x7fffffffLL || !w || !grads || bad_p(p_drop)) return AIT_EINVAL;
  if (rows == 0) return AIT_OK;
  if (!dy || !x || !dx || !saved || !workspace) return AIT_EINVAL;
  if (saved_bytes < ait_ffn_saved_bytes(rows) || workspace_bytes < ait_ffn_bwd_workspace_bytes(rows)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(const_cast<void*>(saved)), saved_bytes};
  FfnBuf m;
  if (!carve_ffn(b, m, rows)) return AIT_EWORKSPACE;
  Bump bw{static_cast<char*>(workspace), workspace_bytes};
  FfnBwdWs t;
  if (!carve_ffn_ws(bw, t, rows)) return AIT_EWORKSPACE;
  return ffn_block_bwd(dy, x, rows, *w, m, t, p_drop, seed, dx, *grads, Run{stream, ctx});
}

// ---------------------------------------------------------------------------------------------------
// the whole operator
// ---------------------------------------------------------------------------------------------------
namespace {
// everything a forward produces; in eval the buffers of consecutive blocks alias, in training all are kept
struct AitBufs {
  float *emb_p, *emb_q;
  float *x0, *mean0, *rstd0;        // encoder prologue output / LayerNorm statistics
  MhaBuf enc_slf;
  float *y1, *xc;                   // encoder self-attention block output; its n_src real rows
  FfnBuf enc_ffn;
  float *mem;                       // encoder memory [bp*n_src, 512]
  float *xd, *meand, *rstdd;        // decoder prologue
  MhaBuf dec_slf;
  float *d1;
  MhaBuf dec_enc;
  float *d2;
  FfnBuf dec_ffn;
  float *d3;
  // the weights of the large products, pre-split at the start of the call (forward form; in training also the
  // transposed form for the input gradients, kept in `saved` for the backward)
  P3W p_enc_emb, p_enc_qkv, p_dec_qkv, p_x_qkv, p_enc_w1, p_enc_w2, p_dec_w1, p_dec_w2, p_dec_trans;
};
// values of all pre-split weights (one conversion each)
constexpr size_t kP3Values = (size_t)D * C2 + 3 * (size_t)3 * D * D + 4 * (size_t)DI * D + (size_t)C2 * D;
inline size_t p3_floats(bool train) { return (train ? 2 : 1) * (kP3Values * 3 / 2) + 20 * 64; }
inline bool carve_p3(Bump& b, AitBufs& a, bool train) {
  bool ok = true;
  auto one = [&](P3W& w, int n_out, int k_in) {
    float* f = b.take((size_t)n_out * k_in * 3 / 2);
    float* t = train ? b.take((size_t)n_out * k_in * 3 / 2) : nullptr;
    ok = ok && f && (!train || t);
    w.w = P3Ref{reinterpret_cast<const unsigned short*>(f), k_in};
    w.wt = P3Ref{reinterpret_cast<const unsigned short*>(t), n_out};
  };
  one(a.p_enc_emb, D, C2);
  one(a.p_enc_qkv, 3 * D, D); one(a.p_dec_qkv, 3 * D, D); one(a.p_x_qkv, 3 * D, D);
  one(a.p_enc_w1, DI, D); one(a.p_enc_w2, D, DI); one(a.p_dec_w1, DI, D); one(a.p_dec_w2, D, DI);
  one(a.p_dec_trans, C2, D);
  return ok;
}
// the conversion pass: one launch for all of them
inline int p3_convert(const ait_transformer_weights* w, const AitBufs& a, void* stream) {
  ait_p3::Jobs jobs;
  jobs.n = 0;
  auto add = [&](const P3W& pw, const float* src, int n_out, int k_in) {
    jobs.j[jobs.n++] = ait_p3::Job{src, const_cast<unsigned short*>(pw.w.p), n_out, k_in, k_in, 0, 0};
    if (pw.wt.p) jobs.j[jobs.n++] = ait_p3::Job{src, const_cast<unsigned short*>(pw.wt.p), n_out, k_in, k_in, 1, 0};
  };
  add(a.p_enc_emb, w->enc_emb_w, D, C2);
  add(a.p_enc_qkv, w->enc_slf.w_qkv, 3 * D, D); add(a.p_dec_qkv, w->dec_slf.w_qkv, 3 * D, D); add(a.p_x_qkv, w->dec_enc.w_qkv, 3 * D, D);
  add(a.p_enc_w1, w->enc_ffn.w1, DI, D); add(a.p_enc_w2, w->enc_ffn.w2, D, DI);
  add(a.p_dec_w1, w->dec_ffn.w1, DI, D); add(a.p_dec_w2, w->dec_ffn.w2, D, DI);
  add(a.p_dec_trans, w->dec_trans_w, C2, D);
  return ait_p3::split(jobs, ait_stream(stream));
}
inline size_t ait_saved_floats(long long bp, long long bs, long long ns) {
  const size_t M = (size_t)bp * T;
  return (size_t)bp * ns * D + (size_t)bs * T * D + M * D + 2 * M + mha_buf_floats(bp, T, false, true) + M * D +
         (size_t)bp * ns * D + (size_t)bp * ns * (DI + D + 2) + (size_t)bp * ns * D + M * D + 2 * M +
         mha_buf_floats(bp, T, false, true) + M * D + mha_buf_floats(bp, (int)ns, true, true) + M * D +
         M * (DI + D + 2) + M * D + 64 * 64 + p3_floats(true);
}
inline bool carve_train(Bump& b, AitBufs& a, long long bp, long long bs, int ns) {
  const size_t M = (size_t)bp * T, Mc = (size_t)bp * ns;
  a.emb_p = b.take(Mc * D); a.emb_q = b.take((size_t)bs * T * D);
  a.x0 = b.take(M * D); a.mean0 = b.take(M); a.rstd0 = b.take(M);
  bool ok = carve(b, a.enc_slf, bp, T, false, true);
  a.y1 = b.take(M * D);
  a.xc = ns < T ? b.take(Mc * D) : a.y1;
  ok = ok && carve_ffn(b, a.enc_ffn, (long long)Mc);
  a.mem = b.take(Mc * D);
  a.xd = b.take(M * D); a.meand = b.take(M); a.rstdd = b.take(M);
  ok = ok && carve(b, a.dec_slf, bp, T, false, true);
  a.d1 = b.take(M * D);
  ok = ok && carve(b, a.dec_enc, bp, ns, true, true);
  a.d2 = b.take(M * D);
  ok = ok && carve_ffn(b, a.dec_ffn, (long long)M);
  a.d3 = b.take(M * D);
  ok = ok && carve_p3(b, a, true);
  return ok && a.emb_p && a.emb_q && a.x0 && a.mean0 && a.rstd0 && a.y1 && a.xc && a.mem && a.xd && a.meand &&
         a.rstdd && a.d1 && a.d2 && a.d3;
}
inline size_t ws_floats(long long bp, long long bs, long long ns) {
  const size_t M = (size_t)bp * T;
  return (size_t)bp * ns * D + (size_t)bs * T * D + M * D + mha_buf_floats(bp, (int)ns, true, false) + M * 2 * D +
         M * D + (size_t)bp * ns * D + M * DI + M * D + (size_t)bp * ns * D + 3 * M * D + 32 * 64 + p3_floats(false);
}
inline bool carve_eval(Bump& b, AitBufs& a, long long bp, long long bs, int ns) {
  const size_t M = (size_t)bp * T, Mc = (size_t)bp * ns;
  a.emb_p = b.take(Mc * D); a.emb_q = b.take((size_t)bs * T * D);
  a.x0 = b.take(M * D); a.mean0 = a.rstd0 = nullptr;
  // one attention scratch set, sized for the widest use (q | kv of the cross block fits in the self block's qkv)
  MhaBuf m;
  m.qkv = b.take(M * 3 * D + Mc * 2 * D);
  m.P = nullptr;
  m.O = b.take(M * D); m.u = b.take(M * DK); m.gate = b.take((size_t)bp * D); m.s = b.take((size_t)bp * DK);
  m.f = b.take(M * D); m.mean = m.rstd = nullptr;
  a.enc_slf = a.dec_slf = a.dec_enc = m;
  a.y1 = b.take(M * D);
  a.xc = ns < T ? b.take(Mc * D) : a.y1;
  FfnBuf f{b.take(M * DI), m.f, nullptr, nullptr};
  a.enc_ffn = a.dec_ffn = f;
  a.mem = b.take(Mc * D);
  a.xd = a.x0; a.meand = a.rstdd = nullptr;
  a.d1 = b.take(M * D); a.d2 = b.take(M * D); a.d3 = a.d1;
  if (!carve_p3(b, a, false)) return false;
  return m.qkv && m.O && m.u && m.gate && m.s && m.f && a.emb_p && a.emb_q && a.x0 && a.y1 && a.xc && f.h && a.mem &&
         a.d1 && a.d2;
}

// ---- dec_trans with a bf16 result / a bf16 output gradient (AIT_CTX_BF16 | AIT_CTX_IO_BF16) -----------------------------
// The consumer of the operator's output in the bf16 configuration is a bf16 convolution stack (the proposal tail on MIOpen):
// `out` leaves as bf16 [M, 1024] and `d_out` arrives as bf16, d3's bf16 copy (the product's left operand, and the right
// operand of the weight gradient) lives in what the bf16-storage feed-forward leaves unused of its h buffer, the weight and
// its transpose in the pre-split scratch.  Taken when the decoder's feed-forward runs in bf16 storage (the h buffer is laid
// out that way) and the rows suit the weight-gradient kernel.
// Round 6: the same three products on bf16 operands ALSO when the consumer wants f32 (the proposal tail in the library: its SK
// blocks read f32) -- `out` then leaves the bf16 product as f32 and the backward converts `d_out` once (a pass over 1.5 GB that
// the two gradient products and the column sum repay twice over: -0.8 ms at cfg5).  io = false: that form.
struct Io16 {
  bool on = false, io = false;      // io: `out` / `d_out` are bf16 tensors (AIT_CTX_IO_BF16)
  unsigned short *d3, *w, *wt;      // [M, D], [C2, D], [D, C2]
};
inline Io16 io16_plan(long long M, const Run& s, const AitBufs& a) {
  Io16 o;
  if (!s.ctx || !(s.ctx->flags & AIT_CTX_BF16)) return o;
  o.io = (s.ctx->flags & AIT_CTX_IO_BF16) != 0;
  if (!o.io && !ait_lab::Knobs::dec_trans_bf16) return o;
  const Bf16Ffn f = bf16_ffn_plan(M, s, a.p_dec_w1, a.p_dec_w2);
  if (!f.on || !a.p_dec_trans.w.p) return o;
  if (!bf16_tn_split(C2, D, M, 0)) return o;      // (the weight gradient's rows: whole 32-row slabs per K-range)
  o.on = true;
  o.d3 = reinterpret_cast<unsigned short*>(a.dec_ffn.h) + (size_t)M * DI + (size_t)M * D;      // behind h16 and x16
  o.w = const_cast<unsigned short*>(a.p_dec_trans.w.p);
  o.wt = o.w + (size_t)C2 * D;
  return o;
}

// ---- the storage format of a training forward's `saved` buffer ---------------------------------------------------------
// Which blocks stored what as bf16 is decided by the call's ctx flags and sizes (qkv16_on, bf16_ffn_plan, io16_plan above).
// The backward re-derives those decisions from ITS ctx; nothing in device memory could tell it (reading a header back would
// cost a device-to-host round trip per step).  So the forward REPORTS its decisions as one word, the caller hands that word
// to the backward, and a backward whose own derivation differs -- other flags, another ctx -- returns AIT_EINVAL instead of
// reading bf16 bytes as f32.  One function, called with the same arguments by both.
constexpr unsigned kFmtMagic = 0xA1700000u;
inline unsigned saved_format(const Run& run, const AitBufs& a, int bp, int n_src) {
  const long long M = (long long)bp * T, Mc = (long long)bp * n_src;
  unsigned f = kFmtMagic;
  if (qkv16_on(run, a.p_enc_qkv, M)) f |= AIT_SAVED_ENC_QKV16;
  if (qkv16_on(run, a.p_dec_qkv, M)) f |= AIT_SAVED_DEC_QKV16;
  if (qkv16_on(run, a.p_x_qkv, M)) f |= AIT_SAVED_X_QKV16;
  if (bf16_ffn_plan(Mc, run, a.p_enc_w1, a.p_enc_w2).on && a.p_enc_w1.wt.p) f |= AIT_SAVED_ENC_FFN16;
  if (bf16_ffn_plan(M, run, a.p_dec_w1, a.p_dec_w2).on && a.p_dec_w1.wt.p) f |= AIT_SAVED_DEC_FFN16;
  const Io16 io = io16_plan(M, run, a);
  if (io.on) f |= io.io ? AIT_SAVED_IO16 : AIT_SAVED_DT16;
  return f;
}

// block seeds of the operator's ten dropout sites (two per attention block, one per feed-forward / prologue)
enum { kSeedEncPro = 16, kSeedEncSlf, kSeedEncFfn, kSeedDecPro, kSeedDecSlf, kSeedDecEnc, kSeedDecFfn };

int ait_forward(const float* x_props, const float* x_query, int bp, int bs, int n_src, const ait_transformer_weights* w,
                const AitBufs& a, float p, float p_attn, unsigned long long seed, float* out, const Run& run) {
  void* stream = run.stream;
  const int M = bp * T, P = bp / bs;
  AIT_TRY(p3_convert(w, a, stream));     // this call's weights, pre-split (33 MB read, one launch)
  // embeddings (1x1 convolutions on token rows)
  AIT_TRY(linear(x_props, bp * n_src, C2, w->enc_emb_w, D, w->enc_emb_b, false, a.emb_p, run, a.p_enc_emb.w));
  AIT_TRY(linear(x_query, bs * T, C2, w->dec_emb_w, D, w->dec_emb_b, false, a.emb_q, run));
  // ---- encoder (Models.py:83-111): zero-pad n_src -> 64 rows inside the LayerNorm row map --------
  AIT_TRY(ait_ln_fwd(a.emb_p, w->pos_table, nullptr, w->enc_ln_g, w->enc_ln_b, M, D, T, n_src, 1, kEps, p,
                     ait_dropout_seed(seed, kSeedEncPro), a.x0, a.mean0, a.rstd0, stream));
  // only the n_src real rows of each sequence are read again (dead padded rows are masked as keys everywhere
  // downstream): the block's closing LayerNorm writes them compacted, straight into xc
  AIT_TRY(mha_block(a.x0, a.x0, bp, T, /*key padding*/ 1, n_src, w->enc_slf, a.enc_slf, p, p_attn,
                    ait_dropout_seed(seed, kSeedEncSlf), a.xc, run, a.p_enc_qkv, n_src));
  AIT_TRY(ffn_block(a.xc, (long long)bp * n_src, w->enc_ffn, a.enc_ffn, p, ait_dropout_seed(seed, kSeedEncFfn), a.mem,
                    run, a.p_enc_w1, a.p_enc_w2));
  // ---- decoder (Models.py:143-172): the query sequence of a pair repeated over its P proposals ----
  const bool train = a.meand != nullptr;
  if (!train && p == 0.f && p_attn == 0.f && P > 1) {
    // Inference: without dropout the P repeats of a pair's query sequence are equal all the way through the decoder's
    // prologue, its self-attention block and the query projection of the cross-attention (-147 MFLOP per
    // proposal).  Run them once per PAIR; the cross-attention reads queries and residual of sequence n / P.
    AIT_TRY(ait_ln_fwd(a.emb_q, w->pos_table, nullptr, w->dec_ln_g, w->dec_ln_b, bs * T, D, T, T, 1, kEps, 0.f, 0, a.xd,
                       nullptr, nullptr, stream));
    AIT_TRY(mha_block(a.xd, a.xd, bs, T, /*causal*/ 2, 0, w->dec_slf, a.dec_slf, 0.f, 0.f, 0, a.d1, run, a.p_dec_qkv));
    AIT_TRY(mha_block(a.d1, a.mem, bp, n_src, n_src < T ? 0 : 1, n_src, w->dec_enc, a.dec_enc, 0.f, 0.f, 0, a.d2, run,
                      a.p_x_qkv, T, P));
  } else {
    AIT_TRY(ait_ln_fwd(a.emb_q, w->pos_table, nullptr, w->dec_ln_g, w->dec_ln_b, M, D, T, T, P, kEps, p,
                       ait_dropout_seed(seed, kSeedDecPro), a.xd, a.meand, a.rstdd, stream));
    AIT_TRY(mha_block(a.xd, a.xd, bp, T, /*causal*/ 2, 0, w->dec_slf, a.dec_slf, p, p_attn,
                      ait_dropout_seed(seed, kSeedDecSlf), a.d1, run, a.p_dec_qkv));
    AIT_TRY(mha_block(a.d1, a.mem, bp, n_src, /*none: the memory is unpadded*/ n_src < T ? 0 : 1, n_src, w->dec_enc,
                      a.dec_enc, p, p_attn, ait_dropout_seed(seed, kSeedDecEnc), a.d2, run, a.p_x_qkv));
  }
  AIT_TRY(ffn_block(a.d2, M, w->dec_ffn, a.dec_ffn, p, ait_dropout_seed(seed, kSeedDecFfn), a.d3, run, a.p_dec_w1, a.p_dec_w2));
  // dec_trans back to 2d channels per token
  const Io16 io = io16_plan(M, run, a);
  if (run.ctx && (run.ctx->flags & AIT_CTX_IO_BF16) && !io.on) return AIT_EUNSUPPORTED;
  if (io.on) {
    AIT_TRY(ait_f32_to_bf16(w->dec_trans_w, C2, D, D, io.w, D, 0, stream));
    AIT_TRY(ait_f32_to_bf16(a.d3, M, D, D, io.d3, D, 0, stream));
    return ait_gemm_bf16s(M, C2, D, io.d3, D, io.w, D, io.io ? nullptr : out, C2, io.io ? out : nullptr, C2, w->dec_trans_b, nullptr,
                          nullptr, 0, 0, run.ctx, stream);
  }
  return linear(a.d3, M, D, w->dec_trans_w, C2, w->dec_trans_b, false, out, run, a.p_dec_trans.w);
}

int check_ait(int bp, int bs, int n_src, const void* w) {
  if (bp < 0 || bs <= 0 || n_src <= 0 || n_src > T || !w) return AIT_EINVAL;
  if (bp % bs) return AIT_EINVAL;
  if ((long long)bp * T > 0x7fffffffLL) return AIT_EUNSUPPORTED;   // GEMM row counts are ints
  return AIT_OK;
}
}  // namespace

AIT_API size_t ait_transformer_workspace_bytes(int bp, int bs, int n_src) {
  if (bp <= 0 || bs <= 0 || n_src <= 0 || n_src > T) return 0;
  return ws_floats(bp, bs, n_src) * sizeof(float) + 32 * 256;
}

AIT_API int ait_transformer_fwd(const float* x_props, const float* x_query, int bp, int bs, int n_src,
                                const ait_transformer_weights* w, void* workspace, size_t workspace_bytes,
                                float* out, const ait_launch_ctx* ctx, void* stream) {
  AIT_TRY(check_ait(bp, bs, n_src, w));
  if (bp == 0) return AIT_OK;
  if (!x_props || !x_query || !out || !workspace) return AIT_EINVAL;
  if (workspace_bytes < ait_transformer_workspace_bytes(bp, bs, n_src)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(workspace), workspace_bytes};
  AitBufs a;
  if (!carve_eval(b, a, bp, bs, n_src)) return AIT_EWORKSPACE;
  return ait_forward(x_props, x_query, bp, bs, n_src, w, a, 0.f, 0.f, 0, out, Run{stream, ctx});
}

AIT_API size_t ait_transformer_saved_bytes(int bp, int bs, int n_src) {
  if (bp <= 0 || bs <= 0 || n_src <= 0 || n_src > T) return 0;
  return ait_saved_floats(bp, bs, n_src) * sizeof(float) + 64 * 256;
}

AIT_API int ait_transformer_fwd_train(const float* x_props, const float* x_query, int bp, int bs, int n_src,
                                      const ait_transformer_weights* w, float p_drop, float p_attn_drop,
                                      unsigned long long seed, void* saved, size_t saved_bytes,
                                      unsigned* saved_format_out, float* out, const ait_launch_ctx* ctx, void* stream) {
  AIT_TRY(check_ait(bp, bs, n_src, w));
  if (bad_p(p_drop) || bad_p(p_attn_drop) || !saved_format_out) return AIT_EINVAL;
  *saved_format_out = kFmtMagic;
  if (bp == 0) return AIT_OK;
  if (!x_props || !x_query || !out || !saved) return AIT_EINVAL;
  if (saved_bytes < ait_transformer_saved_bytes(bp, bs, n_src)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(saved), saved_bytes};
  AitBufs a;
  if (!carve_train(b, a, bp, bs, n_src)) return AIT_EWORKSPACE;
  *saved_format_out = saved_format(Run{stream, ctx}, a, bp, n_src);
  return ait_forward(x_props, x_query, bp, bs, n_src, w, a, p_drop, p_attn_drop, seed, out, Run{stream, ctx});
}

// 1 if ait_transformer_fwd_train / _fwd / _bwd take AIT_CTX_BF16 | AIT_CTX_IO_BF16 at this size (else they return
// AIT_EUNSUPPORTED under that flag)
AIT_API int ait_transformer_io_bf16_ok(int bp, int bs, int n_src) {
  if (check_ait(bp, bs, n_src, &bp) != AIT_OK || bp == 0) return 0;
  const long long M = (long long)bp * T;
  if (M < 256 || M > 0x7fffffffLL / DI) return 0;
  return bf16_tn_split(D, DI, M, (size_t)M * (DI - D) * 2) > 0 && bf16_tn_split(C2, D, M, 0) > 0;
}

AIT_API size_t ait_transformer_bwd_workspace_bytes(int bp, int bs, int n_src) {
  if (bp <= 0 || bs <= 0 || n_src <= 0 || n_src > T) return 0;
  const size_t M = (size_t)bp * T;
  // three gradient carriers [M, 512], d_emb_q, the widest block scratch (attention or feed-forward)
  size_t blk = mha_bwd_ws_floats(bp, T, true) + M * 2 * D;
  const size_t ffn = M * (2 * D + DI);
  if (ffn > blk) blk = ffn;
  return (3 * M * D + (size_t)bs * T * D + blk) * sizeof(float) + 32 * 256;
}

// parts: bit 0 = decoder head (dec_trans, decoder feed-forward), bit 1 = decoder attention (cross, self, prologue,
// dec_emb), bit 2 = encoder (feed-forward, self-attention, prologue, enc_emb).  The gradient carriers between the
// parts live at fixed places of `workspace`: a caller that runs the parts as separate calls (in this order) hands
// every call the same workspace.
static int ait_backward_parts(int parts, const float* d_out, const float* x_props, const float* x_query, int bp, int bs,
                              int n_src, const ait_transformer_weights* w, float p_drop, float p_attn_drop,
                              unsigned long long seed, const void* saved, size_t saved_bytes, unsigned saved_fmt,
                              void* workspace, size_t workspace_bytes, float* d_x_props, float* d_x_query,
                              const ait_transformer_grads* g, const ait_launch_ctx* ctx, void* stream) {
  AIT_TRY(check_ait(bp, bs, n_src, w));
  if (bad_p(p_drop) || bad_p(p_attn_drop) || !g) return AIT_EINVAL;
  if ((saved_fmt & 0xFFF00000u) != kFmtMagic) return AIT_EINVAL;      // not a word ait_transformer_fwd_train reported
  if (bp == 0) return AIT_OK;
  if (((parts & 1) && !d_out) || !x_props || !x_query || !saved || !workspace) return AIT_EINVAL;
  if (saved_bytes < ait_transformer_saved_bytes(bp, bs, n_src)) return AIT_EWORKSPACE;
  if (workspace_bytes < ait_transformer_bwd_workspace_bytes(bp, bs, n_src)) return AIT_EWORKSPACE;
  hipStream_t hs = ait_stream(stream);
  const Run run{stream, ctx};
  const int M = bp * T, P = bp / bs, Mc = bp * n_src;
  Bump bs_{static_cast<char*>(const_cast<void*>(saved)), saved_bytes};
  AitBufs a;
  if (!carve_train(bs_, a, bp, bs, n_src)) return AIT_EWORKSPACE;
  // the forward stored its tensors in the format it reported; this call would read them in the format ITS ctx implies
  if (saved_format(run, a, bp, n_src) != saved_fmt) return AIT_EINVAL;
  Bump b{static_cast<char*>(workspace), workspace_bytes};
  float* ga = b.take((size_t)M * D);
  float* gb = b.take((size_t)M * D);
  float* gc = b.take((size_t)M * D);
  float* d_emb_q = b.take((size_t)bs * T * D);
  if (!ga || !gb || !gc || !d_emb_q) return AIT_EWORKSPACE;
  const Bump blk = b;        // every block carves its scratch from the same remainder
  const float p = p_drop, pa = p_attn_drop;

  if (parts & 1) {
  // dec_trans: out = d3 W^T + b
  const Io16 io = io16_plan(M, run, a);
  if (ctx && (ctx->flags & AIT_CTX_IO_BF16) && !io.on) return AIT_EUNSUPPORTED;
  if (io.on) {
    // (the forward of this step left d3's bf16 copy behind: same plan.  d_out: a bf16 tensor, or f32 and converted here once)
    Bump bb = blk;
    const void* d_out16 = d_out;
    if (!io.io) {
      float* cv = bb.take((size_t)M * C2 / 2);
      if (!cv) return AIT_EWORKSPACE;
      AIT_TRY(ait_f32_to_bf16(d_out, M, C2, C2, cv, C2, 0, stream));
      d_out16 = cv;
    }
    // (K-ranges of the weight gradient: as many as the block scratch holds partial tiles for)
    const int sp = bf16_tn_split(C2, D, M, bb.left > 4096 ? bb.left - 4096 : 0);
    const size_t part_floats = (size_t)sp * C2 * D;
    float* part = bb.take(part_floats);
    if (g->dec_trans_b) AIT_TRY(ait_colsum_bf16(d_out16, M, C2, C2, g->dec_trans_b, stream));
    if (g->dec_trans_w)
      AIT_TRY(ait_gemm_bf16s_tn(C2, D, M, d_out16, C2, io.d3, D, g->dec_trans_w, D, sp, part, part ? part_floats * sizeof(float) : 0,
                                ctx, stream));
    AIT_TRY(ait_f32_to_bf16(w->dec_trans_w, C2, D, D, io.wt, C2, 1, stream));
    AIT_TRY(ait_gemm_bf16s(M, D, C2, d_out16, C2, io.wt, C2, ga, D, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, ctx, stream));   // ga = d d3
  } else {
  if (g->dec_trans_b) AIT_TRY(ait_colsum_f32(d_out, M, C2, C2, g->dec_trans_b, stream));
  AIT_TRY(wgrad(d_out, M, C2, a.d3, D, g->dec_trans_w, run));
  AIT_TRY(dgrad(d_out, M, C2, w->dec_trans_w, D, nullptr, false, ga, run, nullptr, a.p_dec_trans.wt));      // ga = d d3
  }
  {  // decoder feed-forward: d d3 -> d d2
    Bump bb = blk; FfnBwdWs t;
    if (!carve_ffn_ws(bb, t, M)) return AIT_EWORKSPACE;
    AIT_TRY(ffn_block_bwd(ga, a.d2, M, w->dec_ffn, a.dec_ffn, t, p, ait_dropout_seed(seed, kSeedDecFfn), gb, g->dec_ffn, run,
                          a.p_dec_w1, a.p_dec_w2));
  }
  }
  float* d_mem = gc;     // [Mc, 512]
  if (parts & 2) {
  {  // decoder cross-attention: d d2 -> d d1, d mem
    Bump bb = blk; MhaBwdWs t;
    if (!carve(bb, t, bp, n_src, true)) return AIT_EWORKSPACE;
    AIT_TRY(mha_block_bwd(gb, T, a.d1, a.mem, bp, n_src, w->dec_enc, a.dec_enc, t, p, pa,
                          ait_dropout_seed(seed, kSeedDecEnc), ga, d_mem, g->dec_enc, run, a.p_x_qkv));
  }
  {  // decoder self-attention: d d1 -> d xd
    Bump bb = blk; MhaBwdWs t;
    if (!carve(bb, t, bp, T, false)) return AIT_EWORKSPACE;
    AIT_TRY(mha_block_bwd(ga, T, a.xd, a.xd, bp, T, w->dec_slf, a.dec_slf, t, p, pa, ait_dropout_seed(seed, kSeedDecSlf),
                          gb, nullptr, g->dec_slf, run, a.p_dec_qkv));
  }
  // decoder prologue: LayerNorm(dropout(repeat_P(emb_q) + pos)); the P copies' gradients are summed
  AIT_TRY(ait_ln_bwd(gb, a.emb_q, w->pos_table, nullptr, w->dec_ln_g, a.meand, a.rstdd, M, D, T, T, P, T, p,
                     ait_dropout_seed(seed, kSeedDecPro), ga, nullptr, g->dec_ln_g, g->dec_ln_b, g->dec_emb_b, stream));
  if (hipMemsetAsync(d_emb_q, 0, (size_t)bs * T * D * sizeof(float), hs) != hipSuccess) return AIT_ELAUNCH;
  AIT_TRY(ait_rep_sum_f32(ga, bs, P, (long long)T * D, d_emb_q, stream));
  AIT_TRY(wgrad(d_emb_q, (long long)bs * T, D, x_query, C2, g->dec_emb_w, run));
  if (d_x_query) AIT_TRY(dgrad(d_emb_q, bs * T, D, w->dec_emb_w, C2, nullptr, false, d_x_query, run));
  }
  if (!(parts & 4)) return AIT_OK;

  {  // encoder feed-forward on the compacted rows: d mem -> d xc
    Bump bb = blk; FfnBwdWs t;
    if (!carve_ffn_ws(bb, t, Mc)) return AIT_EWORKSPACE;
    AIT_TRY(ffn_block_bwd(d_mem, a.xc, Mc, w->enc_ffn, a.enc_ffn, t, p, ait_dropout_seed(seed, kSeedEncFfn), ga,
                          g->enc_ffn, run, a.p_enc_w1, a.p_enc_w2));
  }
  {  // encoder self-attention: its output received a gradient only on the n_src real rows of a sequence
    Bump bb = blk; MhaBwdWs t;
    if (!carve(bb, t, bp, T, false)) return AIT_EWORKSPACE;
    AIT_TRY(mha_block_bwd(ga, n_src, a.x0, a.x0, bp, T, w->enc_slf, a.enc_slf, t, p, pa,
                          ait_dropout_seed(seed, kSeedEncSlf), gb, nullptr, g->enc_slf, run, a.p_enc_qkv));
  }
  // encoder prologue: LayerNorm(dropout(pad(emb_p) + pos)); the gradient is indexed by source row
  AIT_TRY(ait_ln_bwd(gb, a.emb_p, w->pos_table, nullptr, w->enc_ln_g, a.mean0, a.rstd0, M, D, T, n_src, 1, T, p,
                     ait_dropout_seed(seed, kSeedEncPro), ga, nullptr, g->enc_ln_g, g->enc_ln_b, g->enc_emb_b, stream));
  AIT_TRY(wgrad(ga, Mc, D, x_props, C2, g->enc_emb_w, run));
  if (d_x_props) AIT_TRY(dgrad(ga, Mc, D, w->enc_emb_w, C2, nullptr, false, d_x_props, run, nullptr, a.p_enc_emb.wt));
  return AIT_OK;
}

AIT_API int ait_transformer_bwd(const float* d_out, const float* x_props, const float* x_query, int bp, int bs,
                                int n_src, const ait_transformer_weights* w, float p_drop, float p_attn_drop,
                                unsigned long long seed, const void* saved, size_t saved_bytes, unsigned saved_format,
                                void* workspace, size_t workspace_bytes, float* d_x_props, float* d_x_query,
                                const ait_transformer_grads* g, const ait_launch_ctx* ctx, void* stream) {
  return ait_backward_parts(7, d_out, x_props, x_query, bp, bs, n_src, w, p_drop, p_attn_drop, seed, saved, saved_bytes,
                            saved_format, workspace, workspace_bytes, d_x_props, d_x_query, g, ctx, stream);
}

AIT_API int ait_transformer_bwd_part(int part, const float* d_out, const float* x_props, const float* x_query, int bp,
                                     int bs, int n_src, const ait_transformer_weights* w, float p_drop, float p_attn_drop,
                                     unsigned long long seed, const void* saved, size_t saved_bytes,
                                     unsigned saved_format, void* workspace, size_t workspace_bytes, float* d_x_props,
                                     float* d_x_query, const ait_transformer_grads* g, const ait_launch_ctx* ctx,
                                     void* stream) {
  if (part < 0 || part > 2) return AIT_EINVAL;
  return ait_backward_parts(1 << part, d_out, x_props, x_query, bp, bs, n_src, w, p_drop, p_attn_drop, seed, saved,
                            saved_bytes, saved_format, workspace, workspace_bytes, d_x_props, d_x_query, g, ctx, stream);
}
