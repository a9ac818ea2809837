// ait_amd/csrc/heads.hip -- the detector's two heads behind the proposal tail, forward and backward
// (lib/model/faster_rcnn/faster_rcnn_sys_transformer_sk_dilat.py:283-290 with the modules of
// resnet_sys_transformer_sk_dilat.py:425-433):
//     bbox_pred = RCNN_bbox_pred(props)                                  Linear(F -> n_bbox)
//     score     = RCNN_cls_score(cat(props, repeat_P(query)))            Linear(2F -> 8) . Linear(8 -> 2)
// `score` is the per-proposal similarity logit pair north_star states its tolerance on.  The reference builds the
// [R, 2F] concatenation (the query row repeated over the P proposals of its pair) and hands three tiny matrices to the
// vendor GEMM; here one wave per proposal row reads its F-vector once, forms the n_bbox + 8 dot products against the
// props halves of the weights, adds the query half (eight dots of the pair's query row, recomputed per wave: 16 KB from
// L2) and applies the 8 -> 2 layer in registers.  No concatenation, no vendor call.  HBM-bound by the proposal features:
// R * F * 4 bytes read once (9.8 MB at 1200 x 2048); the weights (n_bbox + 16 rows of F floats) stay in L2.
//
// Backward: d_hidden = d_score w2; d_props = d_hidden w1[:, :F] + d_bbox w_bbox (a wave per row); the weight gradients
// are [n_bbox + 8, R] x [R, F] products with a tiny left side: a thread per feature column walks a chunk of rows with the
// row's d_hidden / d_bbox values broadcast from LDS, chunk partials are added with fp32 atomics (zero-initialised or
// running-sum buffers of the caller: ACCUMULATED like every parameter gradient of this library).
#include "common.h"

namespace {

constexpr int kHid = 8;        // Linear(2F, 8)
constexpr int kCls = 2;        // Linear(8, 2)
constexpr int kMaxBox = 8;     // n_bbox <= 8 in registers (class-agnostic: 4)
constexpr int kRowsPerBlock = 4;

struct HeadsArgs {
  const float *props, *query;      // [R, F], [bs, F]
  const float *w_bbox, *b_bbox;    // [n_bbox, F], [n_bbox]
  const float *w1, *b1;            // [8, 2F], [8]
  const float *w2, *b2;            // [2, 8], [2]
  int R, bs, P, F, n_bbox;
};

// PER = float4 per lane of an F-vector (F = 256 PER): the row's and the pair's query share are fetched ONCE into registers
// (2 PER independent 16-B loads in flight), then every weight row is PER more independent loads against them.
template <int PER>
__global__ __launch_bounds__(64 * kRowsPerBlock) void heads_fwd_kernel(const HeadsArgs g, float* __restrict__ bbox,
                                                                       float* __restrict__ hidden, float* __restrict__ score) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * kRowsPerBlock + wave;
  if (r >= g.R) return;
  const int n4 = g.F / 4;
  const float4* x = reinterpret_cast<const float4*>(g.props + (size_t)r * g.F);
  const float4* q = reinterpret_cast<const float4*>(g.query + (size_t)(r / g.P) * g.F);
  float4 xs[PER], qs[PER];
#pragma unroll
  for (int i = 0; i < PER; i++) {
    xs[i] = x[lane + 64 * i];
    qs[i] = q[lane + 64 * i];
  }
  auto dot = [&](const float4 (&v)[PER], const float4* __restrict__ w) __attribute__((always_inline)) -> float {
    float4 wv[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) wv[i] = w[lane + 64 * i];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++) s += v[i].x * wv[i].x + v[i].y * wv[i].y + v[i].z * wv[i].z + v[i].w * wv[i].w;
    return s;
  };
  float hb[kMaxBox], hh[kHid];
#pragma unroll
  for (int k = 0; k < kMaxBox; k++)
    hb[k] = k < g.n_bbox ? dot(xs, reinterpret_cast<const float4*>(g.w_bbox + (size_t)k * g.F)) : 0.f;
#pragma unroll
  for (int j = 0; j < kHid; j++) {
    const float4* wp = reinterpret_cast<const float4*>(g.w1 + (size_t)j * 2 * g.F);
    hh[j] = dot(xs, wp) + dot(qs, wp + n4);
  }
#pragma unroll
  for (int k = 0; k < kMaxBox; k++) hb[k] = wave_sum(hb[k]);
#pragma unroll
  for (int j = 0; j < kHid; j++) hh[j] = wave_sum(hh[j]) + g.b1[j];
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < kMaxBox; k++)
      if (k < g.n_bbox) bbox[(size_t)r * g.n_bbox + k] = hb[k] + g.b_bbox[k];
#pragma unroll
    for (int j = 0; j < kHid; j++) hidden[(size_t)r * kHid + j] = hh[j];
#pragma unroll
    for (int c = 0; c < kCls; c++) {
      float s = g.b2[c];
#pragma unroll
      for (int j = 0; j < kHid; j++) s += g.w2[c * kHid + j] * hh[j];
      score[(size_t)r * kCls + c] = s;
    }
  }
}

// d_hidden [R, 8] = d_score w2, its sum over the P proposals of a pair dhq [bs, 8], and the small parameter gradients
// d w2, d b2, d b1, d b_bbox: one workgroup per PAIR walks the pair's P rows (a thread per row), block-level sums, one
// atomic per block and output
__global__ __launch_bounds__(256) void heads_bwd_small_kernel(const HeadsArgs g, const float* __restrict__ d_score,
                                                              const float* __restrict__ d_bbox, const float* __restrict__ hidden,
                                                              float* __restrict__ dh, float* __restrict__ dhq, float* __restrict__ d_w2,
                                                              float* __restrict__ d_b2, float* __restrict__ d_b1,
                                                              float* __restrict__ d_b_bbox) {
  constexpr int kAcc = kCls * kHid + kCls + kHid + kMaxBox;       // d w2 | d b2 | sum d_hidden (= d b1 = dhq) | d b_bbox
  __shared__ float acc[kAcc];
  for (int i = threadIdx.x; i < kAcc; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int b = blockIdx.x;
  float v[kAcc];
#pragma unroll
  for (int i = 0; i < kAcc; i++) v[i] = 0.f;
  for (int p = threadIdx.x; p < g.P; p += 256) {
    const size_t r = (size_t)b * g.P + p;
    float ds[kCls];
#pragma unroll
    for (int c = 0; c < kCls; c++) ds[c] = d_score ? d_score[r * kCls + c] : 0.f;
#pragma unroll
    for (int j = 0; j < kHid; j++) {
      const float h = hidden[r * kHid + j];
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < kCls; c++) {
        d += ds[c] * g.w2[c * kHid + j];
        v[c * kHid + j] += ds[c] * h;
      }
      dh[r * kHid + j] = d;
      v[kCls * kHid + kCls + j] += d;
    }
#pragma unroll
    for (int c = 0; c < kCls; c++) v[kCls * kHid + c] += ds[c];
#pragma unroll
    for (int k = 0; k < kMaxBox; k++)
      if (d_bbox && k < g.n_bbox) v[kCls * kHid + kCls + kHid + k] += d_bbox[r * g.n_bbox + k];
  }
#pragma unroll
  for (int i = 0; i < kAcc; i++) {
    const float t = wave_sum(v[i]);
    if ((threadIdx.x & 63) == 0) atomicAdd(&acc[i], t);
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    const int i = threadIdx.x;
    const float t = acc[i];
    float* dst = nullptr;
    if (i < kCls * kHid) dst = d_w2 ? d_w2 + i : nullptr;
    else if (i < kCls * kHid + kCls) dst = d_b2 ? d_b2 + (i - kCls * kHid) : nullptr;
    else if (i < kCls * kHid + kCls + kHid) {
      dhq[(size_t)b * kHid + (i - kCls * kHid - kCls)] = t;
      dst = d_b1 ? d_b1 + (i - kCls * kHid - kCls) : nullptr;
    } else if (i - (kCls * kHid + kCls + kHid) < g.n_bbox) dst = d_b_bbox ? d_b_bbox + (i - kCls * kHid - kCls - kHid) : nullptr;
    if (dst) atomicAdd(dst, t);
  }
}

// d_props[r, :] = d_hidden[r] w1[:, :F] + d_bbox[r] w_bbox     (a wave per row, 16-B stores)
__global__ __launch_bounds__(64 * kRowsPerBlock) void heads_bwd_props_kernel(const HeadsArgs g, const float* __restrict__ dh,
                                                                             const float* __restrict__ d_bbox,
                                                                             float* __restrict__ d_props) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * kRowsPerBlock + wave;
  if (r >= g.R) return;
  float d[kHid], db[kMaxBox];
#pragma unroll
  for (int j = 0; j < kHid; j++) d[j] = dh[(size_t)r * kHid + j];
#pragma unroll
  for (int k = 0; k < kMaxBox; k++) db[k] = (d_bbox && k < g.n_bbox) ? d_bbox[(size_t)r * g.n_bbox + k] : 0.f;
  const int n4 = g.F / 4;
  float4* out = reinterpret_cast<float4*>(d_props + (size_t)r * g.F);
  for (int c = lane; c < n4; c += 64) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < kHid; j++) {
      const float4 w = reinterpret_cast<const float4*>(g.w1 + (size_t)j * 2 * g.F)[c];
      s.x += d[j] * w.x; s.y += d[j] * w.y; s.z += d[j] * w.z; s.w += d[j] * w.w;
    }
#pragma unroll
    for (int k = 0; k < kMaxBox; k++)
      if (k < g.n_bbox) {
        const float4 w = reinterpret_cast<const float4*>(g.w_bbox + (size_t)k * g.F)[c];
        s.x += db[k] * w.x; s.y += db[k] * w.y; s.z += db[k] * w.z; s.w += db[k] * w.w;
      }
    out[c] = s;
  }
}

// the wide parameter gradients and d_query.  grid (F / 256, row chunks): thread = feature column c.
//   d w1[j, c]     += sum_r dh[r, j] props[r, c]          d w_bbox[k, c] += sum_r d_bbox[r, k] props[r, c]
//   (chunk 0 only) d w1[j, F + c] += sum_b dhq[b, j] query[b, c] ;  d_query[b, c] = sum_j dhq[b, j] w1[j, F + c]
constexpr int kChunkRows = 48;
__global__ __launch_bounds__(256) void heads_bwd_wide_kernel(const HeadsArgs g, const float* __restrict__ dh,
                                                             const float* __restrict__ dhq, const float* __restrict__ d_bbox,
                                                             float* __restrict__ d_w1, float* __restrict__ d_w_bbox,
                                                             float* __restrict__ d_query) {
  __shared__ float coef[kChunkRows][kHid + kMaxBox];
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * kChunkRows, r1 = min(g.R, r0 + kChunkRows);
  for (int i = threadIdx.x; i < (r1 - r0) * (kHid + kMaxBox); i += 256) {
    const int rr = i / (kHid + kMaxBox), k = i % (kHid + kMaxBox);
    coef[rr][k] = k < kHid ? dh[(size_t)(r0 + rr) * kHid + k]
                           : ((d_bbox && k - kHid < g.n_bbox) ? d_bbox[(size_t)(r0 + rr) * g.n_bbox + (k - kHid)] : 0.f);
  }
  __syncthreads();
  if (c >= g.F) return;
  float a[kHid + kMaxBox];
#pragma unroll
  for (int k = 0; k < kHid + kMaxBox; k++) a[k] = 0.f;
  for (int r = r0; r < r1; r++) {
    const float x = g.props[(size_t)r * g.F + c];
#pragma unroll
    for (int k = 0; k < kHid + kMaxBox; k++) a[k] += coef[r - r0][k] * x;
  }
  if (d_w1) {
#pragma unroll
    for (int j = 0; j < kHid; j++) atomicAdd(d_w1 + (size_t)j * 2 * g.F + c, a[j]);
  }
  if (d_w_bbox) {
#pragma unroll
    for (int k = 0; k < kMaxBox; k++)
      if (k < g.n_bbox) atomicAdd(d_w_bbox + (size_t)k * g.F + c, a[kHid + k]);
  }
  if (blockIdx.y == 0) {       // the query half: bs rows only
    float wq[kHid];
#pragma unroll
    for (int j = 0; j < kHid; j++) wq[j] = g.w1[(size_t)j * 2 * g.F + g.F + c];
    float aw[kHid];
#pragma unroll
    for (int j = 0; j < kHid; j++) aw[j] = 0.f;
    for (int b = 0; b < g.bs; b++) {
      const float x = g.query[(size_t)b * g.F + c];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < kHid; j++) {
        const float d = dhq[(size_t)b * kHid + j];
        aw[j] += d * x;
        s += d * wq[j];
      }
      if (d_query) d_query[(size_t)b * g.F + c] = s;
    }
    if (d_w1) {
#pragma unroll
      for (int j = 0; j < kHid; j++) atomicAdd(d_w1 + (size_t)j * 2 * g.F + g.F + c, aw[j]);
    }
  }
}

inline int check_args(int R, int bs, int F, int n_bbox) {
  if (R < 0 || bs <= 0 || F <= 0 || n_bbox < 0) return AIT_EINVAL;
  if (R % bs) return AIT_EINVAL;
  if ((F % 256) || F > 4096 || n_bbox > kMaxBox) return AIT_EUNSUPPORTED;      // a lane owns F / 256 float4 of a row
  return AIT_OK;
}

}  // namespace

AIT_API int ait_heads_fwd(const float* props, const float* query, int R, int bs, int F, const float* w_bbox,
                          const float* b_bbox, int n_bbox, const float* w1, const float* b1, const float* w2, const float* b2,
                          float* bbox_pred, float* hidden, float* score, void* stream) {
  AIT_TRY_RC(check_args(R, bs, F, n_bbox));
  if (R == 0) return AIT_OK;
  if (!props || !query || !w1 || !b1 || !w2 || !b2 || !hidden || !score || (n_bbox > 0 && (!w_bbox || !b_bbox || !bbox_pred)))
    return AIT_EINVAL;
  HeadsArgs g{props, query, w_bbox, b_bbox, w1, b1, w2, b2, R, bs, R / bs, F, n_bbox};
  const dim3 grid((R + kRowsPerBlock - 1) / kRowsPerBlock), block(64 * kRowsPerBlock);
  switch (F / 256) {
    case 1: hipLaunchKernelGGL(heads_fwd_kernel<1>, grid, block, 0, ait_stream(stream), g, bbox_pred, hidden, score); break;
    case 2: hipLaunchKernelGGL(heads_fwd_kernel<2>, grid, block, 0, ait_stream(stream), g, bbox_pred, hidden, score); break;
    case 4: hipLaunchKernelGGL(heads_fwd_kernel<4>, grid, block, 0, ait_stream(stream), g, bbox_pred, hidden, score); break;
    case 8: hipLaunchKernelGGL(heads_fwd_kernel<8>, grid, block, 0, ait_stream(stream), g, bbox_pred, hidden, score); break;
    case 16: hipLaunchKernelGGL(heads_fwd_kernel<16>, grid, block, 0, ait_stream(stream), g, bbox_pred, hidden, score); break;
    default: return AIT_EUNSUPPORTED;
  }
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API size_t ait_heads_bwd_workspace_bytes(int R, int bs) {
  if (R < 0 || bs <= 0) return 0;
  return ((size_t)R + (size_t)bs) * kHid * sizeof(float);
}

AIT_API int ait_heads_bwd(const float* d_bbox, const float* d_score, const float* props, const float* query, int R, int bs,
                          int F, const float* w_bbox, int n_bbox, const float* w1, const float* w2, const float* hidden,
                          void* workspace, size_t workspace_bytes, float* d_props, float* d_query, float* d_w_bbox,
                          float* d_b_bbox, float* d_w1, float* d_b1, float* d_w2, float* d_b2, void* stream) {
  AIT_TRY_RC(check_args(R, bs, F, n_bbox));
  if (R == 0) return AIT_OK;
  if (!props || !query || !w1 || !w2 || !hidden || !workspace || (n_bbox > 0 && !w_bbox)) return AIT_EINVAL;
  if (workspace_bytes < ait_heads_bwd_workspace_bytes(R, bs)) return AIT_EWORKSPACE;
  hipStream_t s = ait_stream(stream);
  float* dh = static_cast<float*>(workspace);
  float* dhq = dh + (size_t)R * kHid;
  HeadsArgs g{props, query, w_bbox, nullptr, w1, nullptr, w2, nullptr, R, bs, R / bs, F, n_bbox};
  hipLaunchKernelGGL(heads_bwd_small_kernel, dim3(bs), dim3(256), 0, s, g, d_score, d_bbox, hidden, dh, dhq, d_w2, d_b2, d_b1,
                     d_b_bbox);
  AIT_CHECK_LAUNCH();
  if (d_props) {
    hipLaunchKernelGGL(heads_bwd_props_kernel, dim3((R + kRowsPerBlock - 1) / kRowsPerBlock), dim3(64 * kRowsPerBlock), 0, s, g,
                       dh, d_bbox, d_props);
    AIT_CHECK_LAUNCH();
  }
  if (d_w1 || d_w_bbox || d_query) {
    hipLaunchKernelGGL(heads_bwd_wide_kernel, dim3((F + 255) / 256, (R + kChunkRows - 1) / kChunkRows), dim3(256), 0, s, g, dh,
                       dhq, d_bbox, d_w1, d_w_bbox, d_query);
    AIT_CHECK_LAUNCH();
  }
  return AIT_OK;
}
