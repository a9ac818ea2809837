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
// Training goes through the Python autograd wrappers, which save what the backward needs.
#include "common.h"

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

inline size_t ws_floats(long long bp, long long bs, long long ns) {
  // emb_p, emb_q, x, qkv (also q | kv), O, u, gate, s, f, y, xc, h, f2, mem, d0, d1, d2 (+ slack per buffer)
  const long long M = bp * T;
  long long f = bp * ns * D + bs * T * D + M * D + M * 3 * D + M * D + M * DK + bp * D + bp * DK + M * D +
                M * D + bp * ns * D + M * DI + M * D + bp * ns * D + M * D + M * D + M * D + bp * ns * 2 * D;
  return (size_t)f + 32 * 64;
}

#define AIT_TRY(expr)            \
  do {                           \
    const int rc__ = (expr);     \
    if (rc__ != AIT_OK) return rc__; \
  } while (0)

// y = x W^T (+ b) (+ relu) on the fp32 matrix cores
inline int linear(const float* x, int M, int K, const float* w, int N, const float* b, bool relu, float* y,
                  void* s) {
  return ait_gemm_f32(0, 1, M, N, K, 1.f, x, K, w, K, y, N, b, nullptr, relu ? AIT_GEMM_RELU : 0, 1, 0, 0, s);
}

// one MultiHeadAttention block (SubLayers.py:68-102 with the selective heads of :22-39):
//   xq [n*64, 512] queries (and residual); keys/values from xkv [n*kv_rows, 512] (xkv == xq: self)
int mha_block(const float* xq, const float* xkv, int n, int kv_rows, int mask_mode, int n_valid,
              const ait_mha_weights& w, float* qkv, float* O, float* u, float* gate, float* sp, float* f,
              float* y, void* s) {
  const int M = n * T;
  const float *q, *k, *v;
  int ldq, ldkv;
  if (xkv == xq) {
    AIT_TRY(linear(xq, M, D, w.w_qkv, 3 * D, nullptr, false, qkv, s));
    q = qkv; k = qkv + D; v = qkv + 2 * D;
    ldq = ldkv = 3 * D;
  } else {
    float* qp = qkv;
    float* kv = qkv + (size_t)M * D;
    AIT_TRY(linear(xq, M, D, w.w_qkv, D, nullptr, false, qp, s));
    AIT_TRY(linear(xkv, n * kv_rows, D, w.w_qkv + (size_t)D * D, 2 * D, nullptr, false, kv, s));
    q = qp; k = kv; v = kv + D;
    ldq = D; ldkv = 2 * D;
  }
  AIT_TRY(ait_attn_fwd(q, ldq, k, ldkv, v, ldkv, n, H, T, DK, kv_rows, mask_mode, n_valid, 0.125f, 0.f, 0,
                       nullptr, O, s));
  AIT_TRY(ait_sh_fwd(O, w.sk_w, w.sk_b, n, H, T, DK, u, gate, sp, s));
  AIT_TRY(linear(u, M, DK, w.fc_w, D, nullptr, false, f, s));
  return ait_ln_fwd(f, nullptr, xq, w.ln_g, w.ln_b, M, D, T, T, 1, kEps, 0.f, 0, y, nullptr, nullptr, s);
}

// PositionwiseFeedForward (SubLayers.py:177-187) on `rows` token rows
int ffn_block(const float* x, long long rows, const ait_ffn_weights& w, float* h, float* f, float* y, void* s) {
  AIT_TRY(linear(x, (int)rows, D, w.w1, DI, w.b1, true, h, s));
  AIT_TRY(linear(h, (int)rows, DI, w.w2, D, w.b2, false, f, s));
  return ait_ln_fwd(f, nullptr, x, w.ln_g, w.ln_b, rows, D, T, T, 1, kEps, 0.f, 0, y, nullptr, nullptr, s);
}

}  // namespace

AIT_API size_t ait_mha_block_workspace_bytes(int n_seq, int kv_rows) {
  if (n_seq <= 0 || kv_rows <= 0 || kv_rows > T) return 0;
  const size_t M = (size_t)n_seq * T;
  // qkv (or q | kv), O, u, gate, s, f
  return (M * 3 * D + (size_t)n_seq * kv_rows * 2 * D + M * D + M * DK + (size_t)n_seq * D + (size_t)n_seq * DK + M * D) *
             sizeof(float) + 8 * 256;
}

AIT_API int ait_mha_block_fwd(const float* xq, const float* xkv, int n_seq, int kv_rows, int mask_mode,
                              int n_valid_keys, const ait_mha_weights* w, void* workspace, size_t workspace_bytes,
                              float* y, void* stream) {
  if (n_seq < 0 || kv_rows <= 0 || kv_rows > T || mask_mode < 0 || mask_mode > 2 || !w) return AIT_EINVAL;
  if (n_seq == 0) return AIT_OK;
  if (!xq || !y || !workspace) return AIT_EINVAL;
  if (!xkv || xkv == xq) {
    if (kv_rows != T) return AIT_EINVAL;       // self-attention: keys are the 64 query tokens
    xkv = xq;
  }
  if (workspace_bytes < ait_mha_block_workspace_bytes(n_seq, kv_rows)) return AIT_EWORKSPACE;
  const size_t M = (size_t)n_seq * T;
  Bump b{static_cast<char*>(workspace), workspace_bytes};
  float* qkv = b.take(M * 3 * D + (size_t)n_seq * kv_rows * 2 * D);
  float* O = b.take(M * D);
  float* u = b.take(M * DK);
  float* gate = b.take((size_t)n_seq * D);
  float* sp = b.take((size_t)n_seq * DK);
  float* f = b.take(M * D);
  if (!qkv || !O || !u || !gate || !sp || !f) return AIT_EWORKSPACE;
  return mha_block(xq, xkv, n_seq, kv_rows, mask_mode, n_valid_keys, *w, qkv, O, u, gate, sp, f, y, stream);
}

AIT_API size_t ait_ffn_workspace_bytes(long long rows) {
  if (rows <= 0) return 0;
  return ((size_t)rows * DI + (size_t)rows * D) * sizeof(float) + 4 * 256;
}

AIT_API int ait_ffn_fwd(const float* x, long long rows, const ait_ffn_weights* w, void* workspace,
                        size_t workspace_bytes, float* y, void* stream) {
  if (rows < 0 || rows > 0x7fffffffLL || !w) return AIT_EINVAL;
  if (rows == 0) return AIT_OK;
  if (!x || !y || !workspace) return AIT_EINVAL;
  if (workspace_bytes < ait_ffn_workspace_bytes(rows)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(workspace), workspace_bytes};
  float* h = b.take((size_t)rows * DI);
  float* f = b.take((size_t)rows * D);
  if (!h || !f) return AIT_EWORKSPACE;
  return ffn_block(x, rows, *w, h, f, y, stream);
}

AIT_API size_t ait_transformer_workspace_bytes(int bp, int bs, int n_src) {
  if (bp <= 0 || bs <= 0 || n_src <= 0 || n_src > T) return 0;
  return ws_floats(bp, bs, n_src) * sizeof(float) + 32 * 256;
}

AIT_API int ait_transformer_fwd(const float* x_props, const float* x_query, int bp, int bs, int n_src,
                                const ait_transformer_weights* w, void* workspace, size_t workspace_bytes,
                                float* out, void* stream) {
  if (bp < 0 || bs <= 0 || n_src <= 0 || n_src > T || !w) return AIT_EINVAL;
  if (bp % bs) return AIT_EINVAL;
  if (bp == 0) return AIT_OK;
  if (!x_props || !x_query || !out || !workspace) return AIT_EINVAL;
  if (workspace_bytes < ait_transformer_workspace_bytes(bp, bs, n_src)) return AIT_EWORKSPACE;
  if ((long long)bp * T > 0x7fffffffLL) return AIT_EUNSUPPORTED;   // GEMM row counts are ints
  hipStream_t hs = ait_stream(stream);
  const int M = bp * T, P = bp / bs;
  Bump b{static_cast<char*>(workspace), workspace_bytes};
  float* emb_p = b.take((size_t)bp * n_src * D);
  float* emb_q = b.take((size_t)bs * T * D);
  float* x = b.take((size_t)M * D);
  float* qkv = b.take((size_t)M * 3 * D + (size_t)bp * n_src * 2 * D);
  float* O = b.take((size_t)M * D);
  float* u = b.take((size_t)M * DK);
  float* gate = b.take((size_t)bp * D);
  float* sp = b.take((size_t)bp * DK);
  float* f = b.take((size_t)M * D);
  float* y = b.take((size_t)M * D);
  float* xc = b.take((size_t)bp * n_src * D);
  float* h = b.take((size_t)M * DI);
  float* mem = b.take((size_t)bp * n_src * D);
  float* d1 = b.take((size_t)M * D);
  float* d2 = b.take((size_t)M * D);
  if (!emb_p || !emb_q || !x || !qkv || !O || !u || !gate || !sp || !f || !y || !xc || !h || !mem || !d1 || !d2)
    return AIT_EWORKSPACE;

  // embeddings (1x1 convolutions on token rows)
  AIT_TRY(linear(x_props, bp * n_src, C2, w->enc_emb_w, D, w->enc_emb_b, false, emb_p, stream));
  AIT_TRY(linear(x_query, bs * T, C2, w->dec_emb_w, D, w->dec_emb_b, false, emb_q, stream));

  // ---- encoder (Models.py:83-111): zero-pad n_src -> 64 rows inside the LayerNorm row map --------
  AIT_TRY(ait_ln_fwd(emb_p, w->pos_table, nullptr, w->enc_ln_g, w->enc_ln_b, M, D, T, n_src, 1, kEps, 0.f, 0, x,
                     nullptr, nullptr, stream));
  AIT_TRY(mha_block(x, x, bp, T, /*key padding*/ 1, n_src, w->enc_slf, qkv, O, u, gate, sp, f, y, stream));
  // only the n_src real rows of each sequence are read again: compact them (dead padded rows are
  // masked as keys everywhere downstream)
  if (n_src < T) {
    if (hipMemcpy2DAsync(xc, (size_t)n_src * D * sizeof(float), y, (size_t)T * D * sizeof(float),
                         (size_t)n_src * D * sizeof(float), bp, hipMemcpyDeviceToDevice, hs) != hipSuccess)
      return AIT_ELAUNCH;
  } else {
    xc = y;
  }
  AIT_TRY(ffn_block(xc, (long long)bp * n_src, w->enc_ffn, h, f, mem, stream));

  // ---- decoder (Models.py:143-172): the query sequence of a pair repeated over its P proposals ----
  AIT_TRY(ait_ln_fwd(emb_q, w->pos_table, nullptr, w->dec_ln_g, w->dec_ln_b, M, D, T, T, P, kEps, 0.f, 0, x,
                     nullptr, nullptr, stream));
  AIT_TRY(mha_block(x, x, bp, T, /*causal*/ 2, 0, w->dec_slf, qkv, O, u, gate, sp, f, d1, stream));
  AIT_TRY(mha_block(d1, mem, bp, n_src, /*none: the memory is unpadded*/ n_src < T ? 0 : 1, n_src, w->dec_enc, qkv, O,
                    u, gate, sp, f, d2, stream));
  AIT_TRY(ffn_block(d2, M, w->dec_ffn, h, f, d1, stream));

  // dec_trans back to 2d channels per token
  return linear(d1, M, D, w->dec_trans_w, C2, w->dec_trans_b, false, out, stream);
}
