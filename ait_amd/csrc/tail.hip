// ait_amd/csrc/tail.hip -- the proposal tail behind the AIT as ONE C entry point per direction:
//
//   AIT output [bp, 8, 8, C] --SKBlock--> [bp, 4, 4, C] --RCNN_top (ResNet layer4)--> mean over positions --> [bp, 4*planes]
//   query feature [bs, 8, 8, C] --its own SKBlock--> ... the same layer4 ... --> [bs, 4*planes]
//
// Reference: lib/model/modules/blocks_sys_transformer_sk_dilat.py:915-997 (SKBlock / SKNet: two grouped
// convolutions -- 1x1 and 3x3, 8 groups, ReLU -- whose squares are summed; the branch-attention weights are computed
// and NOT used, :974-981), lib/model/faster_rcnn/resnet_sys_transformer_sk_dilat.py:85-111 (Bottleneck: stride on
// the first 1x1), :422 (RCNN_top = layer4), :482-491 (_head_to_tail = layer4 -> mean(3).mean(2)), and the call site
// faster_rcnn_sys_transformer_sk_dilat.py:247-253.
//
// What is NOT computed: layer4 opens with stride-2 1x1 convolutions, so of the SK block's 8x8
// output only the 16 even positions are ever read; the block is position-wise behind its convolutions, so it is
// evaluated at those positions (its convolutions run at stride 2) and layer4 takes the result at stride 1.  Same
// sums, and the skipped positions receive exactly zero gradient in the reference.
//
// How it runs: every convolution is a product on the matrix-core kernel of gemm_f32_impl.h -- 1x1 convolutions
// as token-major GEMMs, 3x3 as implicit GEMMs, the SK branches as grouped implicit GEMMs -- with everything
// elementwise in the epilogues: frozen-BN scale folded into the weight rows (one multi-tensor pass per call), BN
// shift / residual / ReLU in the forward epilogues; in the backward the ReLU masks ride in the epilogue of the
// product that forms each gradient (AIT_GEMM_MASK_POS; "+ residual gradient, gated by the block input's sign" for
// the product that closes a bottleneck), so not one elementwise pass runs between two products of layer4.  The
// stride-2 data gradients of the SK branches run per parity class of the input positions (gemm_f32.hip).  The
// proposals' and the queries' rows go through layer4 TOGETHER (it is the same module): one set of launches, one
// weight gradient.  Rows are padded to a multiple of 128 with zero-gradient rows so that the weight gradients'
// K-splits are equal (their last round is then cut evenly over the workgroups without scratch).
#include "common.h"
#include "gemm_internal.h"
#include "p3_jobs.h"

namespace {

constexpr int kPos = 16;          // positions per map behind the SK block (4 x 4)
constexpr int kMaxBlocks = 4;

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

#define AIT_TRY(expr)                \
  do {                               \
    const int rc__ = (expr);         \
    if (rc__ != AIT_OK) return rc__; \
  } while (0)

struct Run {
  void* stream;
  const ait_launch_ctx* ctx;
};

// ---- small kernels -------------------------------------------------------------------------------------------
// row-scaled copies of up to 16 matrices in one launch: out[r][c] (op)= in[r][c] * s[r]
struct ScaleDesc { const float* in; const float* s; float* out; int rows, cols; };
struct ScaleBatch { ScaleDesc d[16]; int n; int add; };
__global__ __launch_bounds__(256) void scale_rows_kernel(const ScaleBatch b) {
  const ScaleDesc d = b.d[blockIdx.y];
  const long long n4 = (long long)d.rows * (d.cols / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / (d.cols / 4));
    const float s = d.s ? d.s[r] : 1.f;
    float4 v = reinterpret_cast<const float4*>(d.in)[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    if (b.add) {
      const float4 o = reinterpret_cast<const float4*>(d.out)[i];
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    reinterpret_cast<float4*>(d.out)[i] = v;
  }
}
int scale_rows(const ScaleBatch& b, hipStream_t s) {
  if (b.n == 0) return AIT_OK;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(256, (unsigned)b.n), dim3(256), 0, s, b);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

// ---- layer4's row order --------------------------------------------------------------------------------------------
// POSITION-MAJOR: row(position p, map m) = p * n_maps_pad + m for every activation of layer4 (xtop, a1, a2, o and their
// gradients).  The 1x1 convolutions, column sums and epilogues do not care about the order of rows; the 3x3 convolutions
// gather through ConvGeom::pm_maps; the SK blocks' closing pass writes its result into this order and the pooling reads it.
// Why: on a 4 x 4 map a 3x3 window hangs over the edge for 12 of the 16 positions -- 31 % of all (position, tap) pairs
// multiply a row of zeros -- and in this order the rows of one position are ONE contiguous block, so that whole blocks of
// the weight gradient's reduction can be skipped (conv_f32.hip).  map-major (the lab knob's other value): row = m * 16 + p.
// (per call: the position-major gather exists in the split-product kernels only -- a call in the bf16 or the f32-instruction
// product form keeps the map-major order)
constexpr bool kPM = ait_lab::Knobs::l4_pm;
inline bool l4_pm_on(const ait_launch_ctx* ctx) { return kPM && !(ctx && (ctx->flags & (AIT_CTX_BF16 | AIT_CTX_NATIVE_F32))); }
__host__ __device__ __forceinline__ long long l4_row(int m, int p, int n_maps_pad, int pm) {
  return pm ? (long long)p * n_maps_pad + m : (long long)m * kPos + p;
}

// pooled[i][c] = mean of the 16 rows of map i (mean(3).mean(2) of the reference as one reduction over equal groups)
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ o, int n_maps, int n_maps_pad, int pm, int C,
                                                       float* __restrict__ pooled) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_maps * (C / 4)) return;
  const int m = (int)(i / (C / 4)), c4 = (int)(i - (long long)m * (C / 4));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < kPos; p++) {
    const float4 v = reinterpret_cast<const float4*>(o + (size_t)l4_row(m, p, n_maps_pad, pm) * C)[c4];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float k = 1.f / kPos;
  reinterpret_cast<float4*>(pooled + (size_t)m * C)[c4] = make_float4(acc.x * k, acc.y * k, acc.z * k, acc.w * k);
}
// g[r][c] = dpooled[r / 16][c] / 16 where o[r][c] > 0 (the ReLU that closes layer4), 0 elsewhere and on padding maps
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ o,
                                                       long long rows, int n_maps, int n_maps_pad, int pm, int C, float* __restrict__ g) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * (C / 4)) return;
  const long long r = i / (C / 4);
  const int c4 = (int)(i - r * (C / 4)), m = pm ? (int)(r % n_maps_pad) : (int)(r / kPos);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m < n_maps) {
    const float4 d = reinterpret_cast<const float4*>(dpooled + (size_t)m * C)[c4];
    const float4 y = reinterpret_cast<const float4*>(o + (size_t)r * C)[c4];
    const float k = 1.f / kPos;
    v = make_float4(y.x > 0.f ? d.x * k : 0.f, y.y > 0.f ? d.y * k : 0.f, y.z > 0.f ? d.z * k : 0.f, y.w > 0.f ? d.w * k : 0.f);
  }
  reinterpret_cast<float4*>(g + (size_t)r * C)[c4] = v;
}

// The SK blocks' closing pass y = relu(a)^2 + relu(b)^2 (csrc/bn_act.hip: ait_sk_sqsum_*) with its result in layer4's row
// order: a / b hold n maps of 16 rows (map-major, as the SK convolutions write them), y row l4_row(map0 + m, p).
// n_zero more maps behind them are zero-filled (the padding maps, by the call that writes the last real ones).
__global__ __launch_bounds__(256) void sqsum_l4_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, int n_zero,
                                                           int map0, int n_maps_pad, int pm, int C, float* __restrict__ y) {
  const long long n4 = (long long)(n + n_zero) * kPos * (C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long r = i / (C / 4);
    const int c4 = (int)(i - r * (C / 4)), m = (int)(r / kPos), p = (int)(r - (long long)m * kPos);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < n) {
      const float4 x = reinterpret_cast<const float4*>(a)[i], z = reinterpret_cast<const float4*>(b)[i];
      float u, v;
      u = fmaxf(x.x, 0.f); v = fmaxf(z.x, 0.f); o.x = u * u + v * v;
      u = fmaxf(x.y, 0.f); v = fmaxf(z.y, 0.f); o.y = u * u + v * v;
      u = fmaxf(x.z, 0.f); v = fmaxf(z.z, 0.f); o.z = u * u + v * v;
      u = fmaxf(x.w, 0.f); v = fmaxf(z.w, 0.f); o.w = u * u + v * v;
    }
    reinterpret_cast<float4*>(y + (size_t)l4_row(map0 + m, p, n_maps_pad, pm) * C)[c4] = o;
  }
}
__global__ __launch_bounds__(256) void sqsum_l4_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                           const float* __restrict__ b, int n, int map0, int n_maps_pad, int pm, int C,
                                                           float* __restrict__ da, float* __restrict__ db) {
  const long long n4 = (long long)n * kPos * (C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long r = i / (C / 4);
    const int c4 = (int)(i - r * (C / 4)), m = (int)(r / kPos), p = (int)(r - (long long)m * kPos);
    const float4 g = reinterpret_cast<const float4*>(dy + (size_t)l4_row(map0 + m, p, n_maps_pad, pm) * C)[c4];
    const float4 x = reinterpret_cast<const float4*>(a)[i], z = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(da)[i] = make_float4(2.f * fmaxf(x.x, 0.f) * g.x, 2.f * fmaxf(x.y, 0.f) * g.y,
                                                   2.f * fmaxf(x.z, 0.f) * g.z, 2.f * fmaxf(x.w, 0.f) * g.w);
    reinterpret_cast<float4*>(db)[i] = make_float4(2.f * fmaxf(z.x, 0.f) * g.x, 2.f * fmaxf(z.y, 0.f) * g.y,
                                                   2.f * fmaxf(z.z, 0.f) * g.z, 2.f * fmaxf(z.w, 0.f) * g.w);
  }
}
inline unsigned l4_grid(long long n4) {
  const long long b = (n4 + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ---- layer4 on bf16 STORAGE (the bf16 configuration, BASELINE configs[4]: cfgs/res101.yml) ----------------------------------
// Under AIT_CTX_BF16 every activation and gradient of layer4 -- its input (the SK blocks' result), a1 / a2 / o of every
// bottleneck, the four gradient buffers -- is HELD in bf16 (map-major rows, padded to a multiple of 256: whole row tiles and
// whole 64-row slabs of the weight gradients' reduction) and the folded weights are converted once per call, both
// orientations: the products run on gemm_bf16s.hip's kernels (bf16 operands from memory, 2-byte results), the 3x3 convolutions
// through their window gather.  Same buffers: every bf16 tensor lives in the first half of the f32 tensor it replaces.  The SK
// blocks keep their f32 tensors (their products round to bf16 in registers): their closing pass writes layer4's input in
// bf16, and the last two data-gradient products of layer4 write the gradient they hand back in f32.
constexpr bool kTail16 = ait_lab::Knobs::tail_bf16s;
typedef unsigned short bf16_t;
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {         // v_cvt_pk_bf16_f32, nearest even
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 t;
  t[0] = (__bf16)a;
  t[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float4 ld_bf16x4(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ void st_bf16x4(bf16_t* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
}
__global__ __launch_bounds__(256) void pool_fwd16_kernel(const bf16_t* __restrict__ o, int n_maps, int C, float* __restrict__ pooled) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_maps * (C / 4)) return;
  const int m = (int)(i / (C / 4)), c4 = (int)(i - (long long)m * (C / 4));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < kPos; p++) {
    const float4 v = ld_bf16x4(o + ((size_t)m * kPos + p) * C + 4 * c4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float k = 1.f / kPos;
  reinterpret_cast<float4*>(pooled + (size_t)m * C)[c4] = make_float4(acc.x * k, acc.y * k, acc.z * k, acc.w * k);
}
__global__ __launch_bounds__(256) void pool_bwd16_kernel(const float* __restrict__ dpooled, const bf16_t* __restrict__ o, long long rows,
                                                         int n_maps, int C, bf16_t* __restrict__ g) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * (C / 4)) return;
  const long long r = i / (C / 4);
  const int c4 = (int)(i - r * (C / 4)), m = (int)(r / kPos);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m < n_maps) {
    const float4 d = reinterpret_cast<const float4*>(dpooled + (size_t)m * C)[c4];
    const float4 y = ld_bf16x4(o + (size_t)r * C + 4 * c4);
    const float k = 1.f / kPos;
    v = make_float4(y.x > 0.f ? d.x * k : 0.f, y.y > 0.f ? d.y * k : 0.f, y.z > 0.f ? d.z * k : 0.f, y.w > 0.f ? d.w * k : 0.f);
  }
  st_bf16x4(g + (size_t)r * C + 4 * c4, v);
}
// sqsum_l4_fwd_kernel with its result in bf16, map-major
__global__ __launch_bounds__(256) void sqsum_l4_fwd16_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, int n_zero,
                                                             int map0, int C, bf16_t* __restrict__ y) {
  const long long n4 = (long long)(n + n_zero) * kPos * (C / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long r = i / (C / 4);
    const int c4 = (int)(i - r * (C / 4));
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < (long long)n * kPos) {
      const float4 x = reinterpret_cast<const float4*>(a)[i], z = reinterpret_cast<const float4*>(b)[i];
      float u, v;
      u = fmaxf(x.x, 0.f); v = fmaxf(z.x, 0.f); o.x = u * u + v * v;
      u = fmaxf(x.y, 0.f); v = fmaxf(z.y, 0.f); o.y = u * u + v * v;
      u = fmaxf(x.z, 0.f); v = fmaxf(z.z, 0.f); o.z = u * u + v * v;
      u = fmaxf(x.w, 0.f); v = fmaxf(z.w, 0.f); o.w = u * u + v * v;
    }
    st_bf16x4(y + ((size_t)map0 * kPos + r) * C + 4 * c4, o);
  }
}

// ---- geometry and buffers ------------------------------------------------------------------------------------
struct Dims {
  int bp, bs, C, P, E;          // C = SK / layer4 input channels, P = planes, E = 4 * planes
  int n_blocks;
  long long Rp, Rq, R;          // rows behind the SK block: proposals, queries, padded total
  int n_maps;                   // bp + bs
  int n_maps_pad;               // R / 16
};
inline int make_dims(int bp, int bs, int C, int planes, int n_blocks, Dims& d) {
  if (bp < 0 || bs < 0 || bp + bs <= 0 || C <= 0 || planes <= 0 || n_blocks < 1 || n_blocks > kMaxBlocks) return AIT_EINVAL;
  if (n_blocks < 2) return AIT_EUNSUPPORTED;      // a one-block tail has nowhere to park its shortcut (ait_hip.h: 2 <= n_blocks)
  if ((C % 1024) || (planes % 128)) return AIT_EUNSUPPORTED;      // 8 groups of a multiple of 128 channels; 128-wide tiles
  d.bp = bp; d.bs = bs; d.C = C; d.P = planes; d.E = 4 * planes; d.n_blocks = n_blocks;
  d.Rp = (long long)bp * kPos; d.Rq = (long long)bs * kPos;
  d.R = (long long)align_up((size_t)(d.Rp + d.Rq), 128);
  d.n_maps = bp + bs; d.n_maps_pad = (int)(d.R / kPos);
  if (d.R * 4 > 0x7fffffffLL / 4 || (long long)bp * 64 > 0x7fffffffLL / 4) return AIT_EUNSUPPORTED;
  if ((unsigned long long)d.R * (unsigned long long)(d.C > d.E ? d.C : d.E) >= (1ull << 32)) return AIT_EUNSUPPORTED;   // 32-bit epilogue offsets
  return AIT_OK;
}
// folded weights of layer4, in launch order
struct BlockW { float *w1, *w2, *w3, *wd; };
// the folded 1x1 weights once more in the pre-split operand format (csrc/p3_jobs.h): planes of W' for the forward
// product, of W'^T for the data gradient -- the 256 x 256 tile of gemm_p3.hip takes them
struct BlockP3 { ait_p3::Pair w1, w3, wd; };
inline size_t p3_floats(const Dims& d) {
  size_t v = 0;
  for (int k = 0; k < d.n_blocks; k++) {
    const size_t cin = k == 0 ? d.C : d.E;
    v += (size_t)d.P * cin + (size_t)d.E * d.P + (k == 0 ? (size_t)d.E * cin : 0);
  }
  return 2 * (v * 3 / 2) + 16 * 64;
}
inline bool carve_p3(Bump& b, const Dims& d, BlockP3 (&p)[kMaxBlocks]) {
  bool ok = true;
  auto one = [&](ait_p3::Pair& w, size_t n_out, size_t k_in) {
    float* f = b.take(n_out * k_in * 3 / 2);
    float* t = b.take(n_out * k_in * 3 / 2);
    ok = ok && f && t;
    w.w = ait_p3::Ref{reinterpret_cast<const unsigned short*>(f), (long long)k_in};
    w.wt = ait_p3::Ref{reinterpret_cast<const unsigned short*>(t), (long long)n_out};
  };
  for (int k = 0; k < d.n_blocks; k++) {
    const size_t cin = k == 0 ? d.C : d.E;
    one(p[k].w1, d.P, cin);
    one(p[k].w3, d.E, d.P);
    if (k == 0) one(p[k].wd, d.E, cin);
  }
  return ok;
}
inline size_t folded_floats(const Dims& d) {
  size_t f = 0;
  for (int k = 0; k < d.n_blocks; k++) {
    const size_t cin = k == 0 ? d.C : d.E;
    f += align_up((size_t)d.P * cin, 64) + align_up((size_t)d.P * 9 * d.P, 64) + align_up((size_t)d.E * d.P, 64);
    if (k == 0) f += align_up((size_t)d.E * cin, 64);
  }
  return f;
}
inline bool carve_folded(Bump& b, const Dims& d, BlockW (&w)[kMaxBlocks]) {
  bool ok = true;
  for (int k = 0; k < d.n_blocks; k++) {
    const size_t cin = k == 0 ? d.C : d.E;
    w[k].w1 = b.take((size_t)d.P * cin);
    w[k].w2 = b.take((size_t)d.P * 9 * d.P);
    w[k].w3 = b.take((size_t)d.E * d.P);
    w[k].wd = k == 0 ? b.take((size_t)d.E * cin) : nullptr;
    ok = ok && w[k].w1 && w[k].w2 && w[k].w3 && (k != 0 || w[k].wd);
  }
  return ok;
}
struct Saved {
  float *zeros;                       // the row of zeros an out-of-map window tap reads
  float *f1p, *f3p, *f1q, *f3q;       // SK branches behind their ReLU, proposals / queries
  float *xtop;                        // [R, C] layer4 input (SK outputs, padded)
  float *a1[kMaxBlocks], *a2[kMaxBlocks], *o[kMaxBlocks];
  BlockW wf[kMaxBlocks];
  BlockP3 p3[kMaxBlocks];
};
constexpr size_t kZeros = 8192;
inline size_t saved_floats(const Dims& d) {
  const size_t R = (size_t)d.R, slack = 64 * 64;
  return kZeros + 2 * (size_t)(d.Rp + d.Rq) * d.C + R * d.C + (size_t)d.n_blocks * (2 * R * d.P + R * d.E) + folded_floats(d) + p3_floats(d) + slack;
}
inline bool carve(Bump& b, const Dims& d, Saved& s) {
  s.zeros = b.take(kZeros);
  s.f1p = b.take((size_t)d.Rp * d.C + 4); s.f3p = b.take((size_t)d.Rp * d.C + 4);
  s.f1q = b.take((size_t)d.Rq * d.C + 4); s.f3q = b.take((size_t)d.Rq * d.C + 4);
  s.xtop = b.take((size_t)d.R * d.C);
  bool ok = s.zeros && s.f1p && s.f3p && s.f1q && s.f3q && s.xtop;
  for (int k = 0; k < d.n_blocks; k++) {
    s.a1[k] = b.take((size_t)d.R * d.P); s.a2[k] = b.take((size_t)d.R * d.P); s.o[k] = b.take((size_t)d.R * d.E);
    ok = ok && s.a1[k] && s.a2[k] && s.o[k];
  }
  return ok && carve_folded(b, d, s.wf) && carve_p3(b, d, s.p3);
}

inline ait_conv_geom sk_geom(int n, int k) { return ait_conv_geom{n, 8, 8, 4, 4, k, k, 2, k / 2, 8}; }
inline ait_conv_geom l4_geom(int n) { return ait_conv_geom{n, 4, 4, 4, 4, 3, 3, 1, 1, 1}; }

// y = relu?(x W^T + bias (+ residual))
inline int linear(const float* x, long long M, int K, const float* w, int N, const float* bias, const float* residual,
                  bool relu, float* y, const Run& r, const ait_p3::Ref& p3 = ait_p3::Ref()) {
  if (p3.p && ait_gemm_p3b_takes((int)M, N, K, r.ctx))
    return ait_gemm_f32_p3b((int)M, N, K, 1.f, x, K, p3.p, p3.ld, y, N, bias, residual, nullptr, relu ? AIT_GEMM_RELU : 0, 0, 0,
                            r.ctx, r.stream);
  return ait_gemm_f32_ex(0, 1, (int)M, N, K, 1.f, x, K, w, K, y, N, bias, residual, nullptr, relu ? AIT_GEMM_RELU : 0, 1, 0, 0,
                         r.ctx, r.stream);
}
// dx = dy W  (+ residual) (gated by mask > 0: MASK_POS when there is no residual, the gate operand when there is)
inline int dgrad(const float* dy, long long M, int N_out, const float* w, int K_in, const float* residual,
                 const float* mask, float* dx, const Run& r, const ait_p3::Ref& p3t = ait_p3::Ref()) {
  // (the residual + gate epilogue reads two more tensors per output: on these short reductions the 256 x 256 tile's
  // epilogue then weighs more than the pre-split planes save -- 35 % against 39 % of the matrix pipe -- so it stays on
  // the 256 x 128 tile)
  constexpr bool both_ok = ait_lab::Knobs::tail_resg_p3;
  if (p3t.p && (both_ok || !(residual && mask)) && ait_gemm_p3b_takes((int)M, K_in, N_out, r.ctx)) {      // B = planes of W'^T: rows K_in, reduction over N_out
    const bool both = residual && mask;
    return ait_gemm_f32_p3b((int)M, K_in, N_out, 1.f, dy, N_out, p3t.p, p3t.ld, dx, K_in, nullptr, both ? residual : (mask ? mask : residual),
                            both ? mask : nullptr, (!both && mask) ? AIT_GEMM_MASK_POS : 0, 0, 0, r.ctx, r.stream);
  }
  if (residual && mask)
    return ait_gemm_f32_ex(0, 0, (int)M, K_in, N_out, 1.f, dy, N_out, w, K_in, dx, K_in, nullptr, residual, mask, 0, 1, 0, 0,
                           r.ctx, r.stream);
  return ait_gemm_f32_ex(0, 0, (int)M, K_in, N_out, 1.f, dy, N_out, w, K_in, dx, K_in, nullptr, mask ? mask : residual, nullptr,
                         mask ? AIT_GEMM_MASK_POS : 0, 1, 0, 0, r.ctx, r.stream);
}
// dW [N_out, K_in] += dy^T x: one K-range per XCD; the persistent kernel cuts the last round evenly (equal splits)
inline int wgrad(const float* dy, long long M, int N_out, const float* x, int K_in, float* dw, const Run& r) {
  const int sp = M >= 4096 ? 8 : 1;
  return ait_gemm_f32_ex(1, 0, N_out, K_in, (int)M, 1.f, dy, N_out, x, K_in, dw, K_in, nullptr, nullptr, nullptr, AIT_GEMM_ATOMIC,
                         sp, 0, 0, r.ctx, r.stream);
}

int check_weights(const ait_tail_weights* w, const Dims& d) {
  if (!w) return AIT_EINVAL;
  const ait_sk_weights* sk[2] = {&w->sk_props, &w->sk_query};
  for (int i = 0; i < 2; i++)
    if (!sk[i]->w1 || !sk[i]->b1 || !sk[i]->w3 || !sk[i]->b3) return AIT_EINVAL;
  for (int k = 0; k < d.n_blocks; k++) {
    const ait_bottleneck_weights& b = w->block[k];
    if (!b.conv1 || !b.conv2 || !b.conv3 || !b.bn1_scale || !b.bn1_shift || !b.bn2_scale || !b.bn2_shift || !b.bn3_scale ||
        !b.bn3_shift)
      return AIT_EINVAL;
    if (k == 0 && (!b.down || !b.bnd_scale || !b.bnd_shift)) return AIT_EINVAL;
  }
  return AIT_OK;
}

// one SKBlock at stride 2: f1 = relu(conv1x1_g8(x) + b1), f3 = relu(conv3x3_g8(x) + b3), y = f1^2 + f3^2 written into
// layer4's input at maps map0 .. map0 + n - 1 (+ n_zero zero-filled maps behind them)
int sk_forward(const float* x, int n, int n_zero, int map0, const Dims& d, const ait_sk_weights& w, float* f1, float* f3, float* xtop,
               const float* zeros, const Run& r, bool out16 = false) {
  if (n > 0) {
    const ait_conv_geom g1 = sk_geom(n, 1), g3 = sk_geom(n, 3);
    AIT_TRY(ait_conv_fwd_f32(x, d.C, w.w1, &g1, d.C, d.C, w.b1, nullptr, AIT_GEMM_RELU, f1, d.C, zeros, kZeros, r.ctx, r.stream));
    AIT_TRY(ait_conv_fwd_f32(x, d.C, w.w3, &g3, d.C, d.C, w.b3, nullptr, AIT_GEMM_RELU, f3, d.C, zeros, kZeros, r.ctx, r.stream));
  }
  if (n + n_zero == 0) return AIT_OK;
  if (out16)
    hipLaunchKernelGGL(sqsum_l4_fwd16_kernel, dim3(l4_grid((long long)(n + n_zero) * kPos * (d.C / 4))), dim3(256), 0,
                       ait_stream(r.stream), f1, f3, n, n_zero, map0, d.C, reinterpret_cast<bf16_t*>(xtop));
  else
    hipLaunchKernelGGL(sqsum_l4_fwd_kernel, dim3(l4_grid((long long)(n + n_zero) * kPos * (d.C / 4))), dim3(256), 0, ait_stream(r.stream),
                       f1, f3, n, n_zero, map0, d.n_maps_pad, (int)l4_pm_on(r.ctx), d.C, xtop);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

// ---- layer4 on bf16 storage: when, its row count, its views of the buffers ---------------------------------------------
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline bool tail16_on(const ait_launch_ctx* ctx, const Dims& d) {
  // (from 1024 rows -- below that the launches are latency-bound either way and the f32 form keeps its position-major
  // tricks; the convolution gather wants a power-of-two channel count)
  return kTail16 && ctx && (ctx->flags & AIT_CTX_BF16) && !(ctx->flags & AIT_CTX_NATIVE_F32) && d.Rp + d.Rq >= 1024 &&
         pow2(d.P) && d.P >= 256;
}
inline long long rows16(const Dims& d) { return (long long)align_up((size_t)(d.Rp + d.Rq), 256); }
struct W16 { bf16_t *w1, *w1t, *w2, *w2d, *w3, *w3t, *wd, *wdt; };
inline W16 weights16(const BlockW& f, const Dims& d, int k) {
  const size_t cin = k == 0 ? d.C : d.E, P = d.P, E = d.E;
  W16 w;
  w.w1 = reinterpret_cast<bf16_t*>(f.w1); w.w1t = w.w1 + P * cin;
  w.w2 = reinterpret_cast<bf16_t*>(f.w2); w.w2d = w.w2 + P * 9 * P;
  w.w3 = reinterpret_cast<bf16_t*>(f.w3); w.w3t = w.w3 + E * P;
  w.wd = reinterpret_cast<bf16_t*>(f.wd); w.wdt = f.wd ? w.wd + E * cin : nullptr;
  return w;
}
inline ait_bf16s::Conv l4_conv16(const float* zeros, int channels) {
  return ait_bf16s::Conv{1, kPos - 1, 2, 3, 4, 4, 3, 1, __builtin_ctz((unsigned)channels), reinterpret_cast<const unsigned short*>(zeros)};
}
// y16 [M, N] = relu?(x16 [M, K] W16[N, K]^T + bias (+ res16)) (kept where gate16 > 0), or the same into y32; conv: through the 3x3 window
inline int mm16(const bf16_t* x, long long M, int K, const bf16_t* w, int N, const float* bias, const bf16_t* res16, const float* res32,
                const bf16_t* gate16, bool relu, bf16_t* y16, float* y32, const Run& r, const ait_bf16s::Conv* cv = nullptr) {
  // (One launch per product.  At cfg5 -- 4096 proposal + 8 query maps = 256.5 row tiles -- every 512-column product would run
  // a third round for two tiles: the 4608-deep convolutions get that round cut along K inside gemm_bf16s.hip (453 -> 370 us, two
  // whole rounds: 374); the shorter reductions keep it -- launching the proposals' rows alone instead collects 1.6 ms on the big
  // launches and gives it back on the query maps' twenty small ones, serial K loops on one or two workgroups, 33-163 us each.)
  ait_bf16s::Gemm p{};
  p.A = x; p.B = w; p.C16 = y16; p.C32 = y32; p.bias = bias; p.res16 = res16; p.res32 = res32; p.gate16 = gate16;
  p.gate = gate16 != nullptr;
  p.M = (int)M; p.N = N; p.K = K;
  p.lda = cv ? (1 << cv->cin_shift) : K; p.ldb = K; p.ldc16 = N; p.ldc32 = N; p.ldr = N; p.ldg = N;
  p.relu = relu;
  if (cv) p.cv = *cv;
  return ait_bf16s::gemm(p, r.ctx, r.stream);
}
// dW [N_out, cols] (f32) += dy16 [R, N_out]^T x16 [R, .] over 16 equal ranges of the rows
// (scratch: room for the ranges' partial tiles -- stored once and added in range order instead of f32 atomics, which cost these
// 4160-row ranges three times the product itself; too small: atomics)
struct Scratch { void* p; size_t bytes; };
inline int wg16(const bf16_t* dy, long long R, int N_out, const bf16_t* x, int K_in, float* dw, const Run& r, const Scratch& scratch,
                const ait_bf16s::Conv* cv = nullptr) {
  ait_bf16s::Wgrad p{};
  p.A = dy; p.B = x; p.C = dw; p.Mo = N_out; p.No = cv ? 9 * K_in : K_in; p.R = (int)R;
  // K-ranges: as many as make tiles x ranges ONE round of the 256 workgroup slots of the 256 x 256 tile (7 for the 36 tiles
  // of the 3x3 gradient, 16 / 32 / 8 for the 1x1 ones); the kernel takes ranges that differ by a slab
  const int tiles = (p.Mo / 256) * (p.No / 256);
  int sp = tiles > 0 ? 256 / tiles : 1;
  sp = sp < 1 ? 1 : (sp > 64 ? 64 : sp);
  if (sp > R / 64) sp = (int)(R / 64) > 0 ? (int)(R / 64) : 1;
  p.split_k = sp;
  p.lda = N_out; p.ldb = K_in; p.ldc = p.No;
  p.partials = scratch.p; p.partials_bytes = scratch.bytes;
  if (cv) p.cv = *cv;
  return ait_bf16s::wgrad(p, r.ctx, r.stream);
}

}  // namespace

AIT_API size_t ait_tail_saved_bytes(int bp, int bs, int channels, int planes, int n_blocks) {
  Dims d;
  if (make_dims(bp, bs, channels, planes, n_blocks, d) != AIT_OK) return 0;
  return saved_floats(d) * sizeof(float) + 64 * 256;
}

constexpr unsigned kTailFmtMagic = 0xA1800000u;
inline unsigned tail_format(const ait_launch_ctx* ctx, const Dims& d) {
  return kTailFmtMagic | (l4_pm_on(ctx) ? AIT_TAIL_SAVED_PM : 0u) | (tail16_on(ctx, d) ? AIT_TAIL_SAVED_BF16 : 0u);
}

AIT_API int ait_tail_fwd(const float* x_props, const float* x_query, int bp, int bs, int channels, int planes,
                         int n_blocks, const ait_tail_weights* w, void* saved, size_t saved_bytes, unsigned* saved_format,
                         float* pooled, const ait_launch_ctx* ctx, void* stream) {
  Dims d;
  AIT_TRY(make_dims(bp, bs, channels, planes, n_blocks, d));
  AIT_TRY(check_weights(w, d));
  if ((bp > 0 && !x_props) || (bs > 0 && !x_query) || !saved || !pooled || !saved_format) return AIT_EINVAL;
  *saved_format = tail_format(ctx, d);
  if (saved_bytes < ait_tail_saved_bytes(bp, bs, channels, planes, n_blocks)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(saved), saved_bytes};
  Saved s;
  if (!carve(b, d, s)) return AIT_EWORKSPACE;
  hipStream_t hs = ait_stream(stream);
  const Run run{stream, ctx};
  const int C = d.C, P = d.P, E = d.E;
  if (hipMemsetAsync(s.zeros, 0, kZeros * sizeof(float), hs) != hipSuccess) return AIT_ELAUNCH;
  if (tail16_on(ctx, d)) {
    // ---- layer4 on bf16 storage: folded weights -> bf16 (both orientations), the SK blocks' result in bf16, bf16 products
    const long long R16 = rows16(d);
    const int npad16 = (int)(R16 / kPos);
    for (int k = 0; k < d.n_blocks; k++) {
      const ait_bottleneck_weights& bw = w->block[k];
      const int cin = k == 0 ? C : E;
      const W16 w16 = weights16(s.wf[k], d, k);
      ait_bf16s::WeightJob jobs[8];
      int n = 0;
      jobs[n++] = ait_bf16s::WeightJob{bw.conv1, bw.bn1_scale, w16.w1, P, 1, cin, 0};
      jobs[n++] = ait_bf16s::WeightJob{bw.conv1, bw.bn1_scale, w16.w1t, P, 1, cin, 1};
      jobs[n++] = ait_bf16s::WeightJob{bw.conv2, bw.bn2_scale, w16.w2, P, 9, P, 0};
      jobs[n++] = ait_bf16s::WeightJob{bw.conv2, bw.bn2_scale, w16.w2d, P, 9, P, 1};
      jobs[n++] = ait_bf16s::WeightJob{bw.conv3, bw.bn3_scale, w16.w3, E, 1, P, 0};
      jobs[n++] = ait_bf16s::WeightJob{bw.conv3, bw.bn3_scale, w16.w3t, E, 1, P, 1};
      if (k == 0) {
        jobs[n++] = ait_bf16s::WeightJob{bw.down, bw.bnd_scale, w16.wd, E, 1, cin, 0};
        jobs[n++] = ait_bf16s::WeightJob{bw.down, bw.bnd_scale, w16.wdt, E, 1, cin, 1};
      }
      AIT_TRY(ait_bf16s::fold_weights(jobs, n, stream));
    }
    AIT_TRY(sk_forward(x_props, bp, 0, 0, d, w->sk_props, s.f1p, s.f3p, s.xtop, s.zeros, run, true));
    AIT_TRY(sk_forward(x_query, bs, npad16 - d.n_maps, bp, d, w->sk_query, s.f1q, s.f3q, s.xtop, s.zeros, run, true));
    const ait_bf16s::Conv cv = l4_conv16(s.zeros, P);
    const bf16_t* xin = reinterpret_cast<const bf16_t*>(s.xtop);
    for (int k = 0; k < d.n_blocks; k++) {
      const ait_bottleneck_weights& bw = w->block[k];
      const int cin = k == 0 ? C : E;
      const W16 w16 = weights16(s.wf[k], d, k);
      bf16_t* a1 = reinterpret_cast<bf16_t*>(s.a1[k]);
      bf16_t* a2 = reinterpret_cast<bf16_t*>(s.a2[k]);
      bf16_t* o = reinterpret_cast<bf16_t*>(s.o[k]);
      AIT_TRY(mm16(xin, R16, cin, w16.w1, P, bw.bn1_shift, nullptr, nullptr, nullptr, true, a1, nullptr, run));
      AIT_TRY(mm16(a1, R16, 9 * P, w16.w2, P, bw.bn2_shift, nullptr, nullptr, nullptr, true, a2, nullptr, run, &cv));
      const bf16_t* idn = xin;
      if (k == 0) {
        bf16_t* park = reinterpret_cast<bf16_t*>(s.o[1]);       // (free until block 1 writes its output; n_blocks >= 2)
        AIT_TRY(mm16(xin, R16, cin, w16.wd, E, bw.bnd_shift, nullptr, nullptr, nullptr, false, park, nullptr, run));
        idn = park;
      }
      AIT_TRY(mm16(a2, R16, P, w16.w3, E, bw.bn3_shift, idn, nullptr, nullptr, true, o, nullptr, run));
      xin = o;
    }
    const long long n4 = (long long)d.n_maps * (E / 4);
    hipLaunchKernelGGL(pool_fwd16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, hs,
                       reinterpret_cast<const bf16_t*>(s.o[d.n_blocks - 1]), d.n_maps, E, pooled);
    AIT_CHECK_LAUNCH();
    return AIT_OK;
  }
  // ---- frozen-BN scales into the weight rows (resnet_sys_transformer_sk_dilat.py:435-441,474-480: every BatchNorm of
  // RCNN_top is frozen and in eval mode): y = bn(conv(x)) = x (diag(scale) W)^T + shift
  {
    ScaleBatch sb{};
    for (int k = 0; k < d.n_blocks; k++) {
      const ait_bottleneck_weights& bw = w->block[k];
      const int cin = k == 0 ? C : E;
      sb.d[sb.n++] = ScaleDesc{bw.conv1, bw.bn1_scale, s.wf[k].w1, P, cin};
      sb.d[sb.n++] = ScaleDesc{bw.conv2, bw.bn2_scale, s.wf[k].w2, P, 9 * P};
      sb.d[sb.n++] = ScaleDesc{bw.conv3, bw.bn3_scale, s.wf[k].w3, E, P};
      if (k == 0) sb.d[sb.n++] = ScaleDesc{bw.down, bw.bnd_scale, s.wf[k].wd, E, cin};
    }
    sb.add = 0;
    AIT_TRY(scale_rows(sb, hs));
    // ... and the folded 1x1 weights once more as bf16 planes, both orientations (one launch)
    ait_p3::Jobs jobs;
    jobs.n = 0;
    auto add = [&](const ait_p3::Pair& pw, const float* src, int n_out, int k_in) {
      jobs.j[jobs.n++] = ait_p3::Job{src, const_cast<unsigned short*>(pw.w.p), n_out, k_in, k_in, 0, 0};
      jobs.j[jobs.n++] = ait_p3::Job{src, const_cast<unsigned short*>(pw.wt.p), n_out, k_in, k_in, 1, 0};
    };
    for (int k = 0; k < d.n_blocks; k++) {
      const int cin = k == 0 ? C : E;
      add(s.p3[k].w1, s.wf[k].w1, P, cin);
      add(s.p3[k].w3, s.wf[k].w3, E, P);
      if (k == 0) add(s.p3[k].wd, s.wf[k].wd, E, cin);
    }
    AIT_TRY(ait_p3::split(jobs, hs));
  }
  // ---- the two SK blocks write their halves of layer4's input; the padding rows are zero
  AIT_TRY(sk_forward(x_props, bp, 0, 0, d, w->sk_props, s.f1p, s.f3p, s.xtop, s.zeros, run));
  AIT_TRY(sk_forward(x_query, bs, d.n_maps_pad - d.n_maps, bp, d, w->sk_query, s.f1q, s.f3q, s.xtop, s.zeros, run));
  // ---- layer4: bottlenecks on [R, .] token rows of 4x4 maps
  const ait_conv_geom g3 = l4_geom(d.n_maps_pad);
  const float* xin = s.xtop;
  for (int k = 0; k < d.n_blocks; k++) {
    const ait_bottleneck_weights& bw = w->block[k];
    const int cin = k == 0 ? C : E;
    AIT_TRY(linear(xin, d.R, cin, s.wf[k].w1, P, bw.bn1_shift, nullptr, true, s.a1[k], run, s.p3[k].w1.w));
    AIT_TRY(ait_conv_fwd_f32_pm(s.a1[k], P, s.wf[k].w2, &g3, P, P, bw.bn2_shift, nullptr, AIT_GEMM_RELU, s.a2[k], P, s.zeros, kZeros,
                                (int)l4_pm_on(ctx), ctx, stream));
    const float* idn = xin;
    if (k == 0) {
      // the projection shortcut; parked in the buffer of the NEXT block's output (free until then), or in the pooled
      // staging of a one-block tail's own output buffer is impossible -> a2 of block 0 is still needed: use o[k] twice
      float* park = d.n_blocks > 1 ? s.o[1] : nullptr;
      if (!park) return AIT_EUNSUPPORTED;
      AIT_TRY(linear(xin, d.R, cin, s.wf[k].wd, E, bw.bnd_shift, nullptr, false, park, run, s.p3[k].wd.w));
      idn = park;
    }
    AIT_TRY(linear(s.a2[k], d.R, P, s.wf[k].w3, E, bw.bn3_shift, idn, true, s.o[k], run, s.p3[k].w3.w));
    xin = s.o[k];
  }
  {
    const long long n4 = (long long)d.n_maps * (E / 4);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, hs, s.o[d.n_blocks - 1], d.n_maps,
                       d.n_maps_pad, (int)l4_pm_on(ctx), E, pooled);
    AIT_CHECK_LAUNCH();
  }
  return AIT_OK;
}

AIT_API size_t ait_tail_bwd_workspace_bytes(int bp, int bs, int channels, int planes, int n_blocks) {
  Dims d;
  if (make_dims(bp, bs, channels, planes, n_blocks, d) != AIT_OK) return 0;
  const size_t R = (size_t)d.R;
  const size_t f = 2 * R * d.E + 2 * R * d.P + R * d.C + 2 * (size_t)(d.Rp + d.Rq) * d.C + folded_floats(d) + 64 * 64;
  return f * sizeof(float) + 64 * 256;
}

AIT_API int ait_tail_bwd(const float* d_pooled, const float* x_props, const float* x_query, int bp, int bs, int channels,
                         int planes, int n_blocks, const ait_tail_weights* w, const void* saved, size_t saved_bytes,
                         unsigned saved_format, void* workspace, size_t workspace_bytes, float* d_x_props, float* d_x_query,
                         const ait_tail_grads* grads, const ait_launch_ctx* ctx, void* stream) {
  Dims d;
  AIT_TRY(make_dims(bp, bs, channels, planes, n_blocks, d));
  AIT_TRY(check_weights(w, d));
  // the forward laid `saved` out in the row order it reported; this call would read it in the order ITS ctx implies
  if (saved_format != tail_format(ctx, d)) return AIT_EINVAL;
  if (!d_pooled || (bp > 0 && !x_props) || (bs > 0 && !x_query) || !saved || !workspace || !grads) return AIT_EINVAL;
  if (saved_bytes < ait_tail_saved_bytes(bp, bs, channels, planes, n_blocks)) return AIT_EWORKSPACE;
  if (workspace_bytes < ait_tail_bwd_workspace_bytes(bp, bs, channels, planes, n_blocks)) return AIT_EWORKSPACE;
  Bump bsv{static_cast<char*>(const_cast<void*>(saved)), saved_bytes};
  Saved s;
  if (!carve(bsv, d, s)) return AIT_EWORKSPACE;
  Bump b{static_cast<char*>(workspace), workspace_bytes};
  const int C = d.C, P = d.P, E = d.E;
  float* ga = b.take((size_t)d.R * E);
  float* gb = b.take((size_t)d.R * E);
  float* g2 = b.take((size_t)d.R * P);
  float* g1 = b.take((size_t)d.R * P);
  float* dxt = b.take((size_t)d.R * C);
  float* df1 = b.take((size_t)(d.Rp + d.Rq) * C + 8);
  float* df3 = b.take((size_t)(d.Rp + d.Rq) * C + 8);
  BlockW dwf[kMaxBlocks];
  float* dwf_base = reinterpret_cast<float*>(b.p);
  if (!ga || !gb || !g2 || !g1 || !dxt || !df1 || !df3 || !carve_folded(b, d, dwf)) return AIT_EWORKSPACE;
  const size_t dwf_bytes = (size_t)(b.p - reinterpret_cast<char*>(dwf_base));
  hipStream_t hs = ait_stream(stream);
  const Run run{stream, ctx};
  if (hipMemsetAsync(dwf_base, 0, dwf_bytes, hs) != hipSuccess) return AIT_ELAUNCH;

  const bool t16 = tail16_on(ctx, d);
  if (t16) {
    const long long R16 = rows16(d);
    bf16_t* gout = reinterpret_cast<bf16_t*>(ga);
    bf16_t* gnext = reinterpret_cast<bf16_t*>(gb);
    bf16_t* g2h = reinterpret_cast<bf16_t*>(g2);
    bf16_t* g1h = reinterpret_cast<bf16_t*>(g1);
    {
      const long long n4 = R16 * (E / 4);
      hipLaunchKernelGGL(pool_bwd16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, hs, d_pooled,
                         reinterpret_cast<const bf16_t*>(s.o[d.n_blocks - 1]), R16, d.n_maps, E, gout);
      AIT_CHECK_LAUNCH();
    }
    const ait_bf16s::Conv cv = l4_conv16(s.zeros, P);
    const Scratch scr{dxt, (size_t)d.R * C * sizeof(float)};       // (dxt is written by the last two products only)
    for (int k = d.n_blocks - 1; k >= 0; k--) {
      const int cin = k == 0 ? C : E;
      const W16 w16 = weights16(s.wf[k], d, k);
      const bf16_t* xin = reinterpret_cast<const bf16_t*>(k == 0 ? s.xtop : s.o[k - 1]);
      const bf16_t* a1 = reinterpret_cast<const bf16_t*>(s.a1[k]);
      const bf16_t* a2 = reinterpret_cast<const bf16_t*>(s.a2[k]);
      AIT_TRY(wg16(gout, R16, E, a2, P, dwf[k].w3, run, scr));                                              // d W3' += g^T a2
      AIT_TRY(mm16(gout, R16, E, w16.w3t, P, nullptr, nullptr, nullptr, a2, false, g2h, nullptr, run));     // g2 = (g W3') [a2 > 0]
      AIT_TRY(wg16(g2h, R16, P, a1, P, dwf[k].w2, run, scr, &cv));
      AIT_TRY(mm16(g2h, R16, 9 * P, w16.w2d, P, nullptr, nullptr, nullptr, a1, false, g1h, nullptr, run, &cv));
      AIT_TRY(wg16(g1h, R16, P, xin, cin, dwf[k].w1, run, scr));                                            // d W1' += g1^T x_in
      if (k > 0) {
        // conv1's data gradient + the identity shortcut's, behind the previous block's ReLU
        AIT_TRY(mm16(g1h, R16, P, w16.w1t, cin, nullptr, gout, nullptr, xin, false, gnext, nullptr, run));
        bf16_t* t = gout; gout = gnext; gnext = t;
      } else {
        AIT_TRY(wg16(gout, R16, E, xin, cin, dwf[k].wd, run, scr));                                         // projection shortcut
        // (the gradient handed to the SK blocks in f32, rows of real maps only: dxt has d.R rows)
        AIT_TRY(mm16(gout, d.R, E, w16.wdt, cin, nullptr, nullptr, nullptr, nullptr, false, nullptr, dxt, run));
        AIT_TRY(mm16(g1h, d.R, P, w16.w1t, cin, nullptr, nullptr, dxt, nullptr, false, nullptr, dxt, run));  // (+=, in place)
      }
    }
  }
  // gradient at layer4's output, behind its closing ReLU
  if (!t16) {
    const long long n4 = d.R * (E / 4);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, hs, d_pooled, s.o[d.n_blocks - 1], d.R,
                       d.n_maps, d.n_maps_pad, (int)l4_pm_on(ctx), E, ga);
    AIT_CHECK_LAUNCH();
  }
  const ait_conv_geom g3 = l4_geom(d.n_maps_pad);
  float* gout = ga;        // gradient at this block's output (masked)
  float* gnext = gb;       // where the gradient at the block's input goes
  for (int k = t16 ? -1 : d.n_blocks - 1; k >= 0; k--) {
    const int cin = k == 0 ? C : E;
    const float* xin = k == 0 ? s.xtop : s.o[k - 1];
    AIT_TRY(wgrad(gout, d.R, E, s.a2[k], P, dwf[k].w3, run));                                     // d W3' += g^T a2
    AIT_TRY(dgrad(gout, d.R, E, s.wf[k].w3, P, nullptr, s.a2[k], g2, run, s.p3[k].w3.wt));                      // g2 = (g W3') [a2 > 0]
    AIT_TRY(ait_conv_bwd_weight_f32_pm(g2, P, s.a1[k], P, &g3, P, P, dwf[k].w2, 8, s.zeros, kZeros, (int)l4_pm_on(ctx), ctx, stream));
    AIT_TRY(ait_conv_bwd_data_f32_pm(g2, P, s.wf[k].w2, &g3, P, P, s.a1[k], AIT_GEMM_MASK_POS, g1, P, s.zeros, kZeros, (int)l4_pm_on(ctx), ctx, stream));
    AIT_TRY(wgrad(g1, d.R, P, xin, cin, dwf[k].w1, run));                                         // d W1' += g1^T x_in
    if (k > 0) {
      // gradient at the previous block's output: conv1's data gradient + the identity shortcut's, behind that block's ReLU
      AIT_TRY(dgrad(g1, d.R, P, s.wf[k].w1, cin, gout, xin, gnext, run, s.p3[k].w1.wt));
      float* t = gout; gout = gnext; gnext = t;
    } else {
      AIT_TRY(wgrad(gout, d.R, E, xin, cin, dwf[k].wd, run));                                     // projection shortcut
      AIT_TRY(dgrad(gout, d.R, E, s.wf[k].wd, cin, nullptr, nullptr, dxt, run, s.p3[k].wd.wt));
      AIT_TRY(dgrad(g1, d.R, P, s.wf[k].w1, cin, dxt, nullptr, dxt, run, s.p3[k].w1.wt));                      // (+=, in place)
    }
  }
  // weight gradients: from the folded weights back to the parameters (d W = diag(scale) d W'), ACCUMULATED
  {
    ScaleBatch sb{};
    for (int k = 0; k < d.n_blocks; k++) {
      const ait_bottleneck_weights& bw = w->block[k];
      const ait_bottleneck_grads& bg = grads->block[k];
      const int cin = k == 0 ? C : E;
      if (bg.conv1) sb.d[sb.n++] = ScaleDesc{dwf[k].w1, bw.bn1_scale, bg.conv1, P, cin};
      if (bg.conv2) sb.d[sb.n++] = ScaleDesc{dwf[k].w2, bw.bn2_scale, bg.conv2, P, 9 * P};
      if (bg.conv3) sb.d[sb.n++] = ScaleDesc{dwf[k].w3, bw.bn3_scale, bg.conv3, E, P};
      if (k == 0 && bg.down) sb.d[sb.n++] = ScaleDesc{dwf[k].wd, bw.bnd_scale, bg.down, E, cin};
    }
    sb.add = 1;
    AIT_TRY(scale_rows(sb, hs));
  }
  // ---- the two SK blocks: y = f1^2 + f3^2
  const long long Rsk = d.Rp + d.Rq;
  // (f1p | f1q and f3p | f3q are separate buffers: two passes each)
  if (bp > 0) {
    hipLaunchKernelGGL(sqsum_l4_bwd_kernel, dim3(l4_grid((long long)bp * kPos * (C / 4))), dim3(256), 0, hs, dxt, s.f1p, s.f3p, bp, 0,
                       d.n_maps_pad, (int)l4_pm_on(ctx), C, df1, df3);
    AIT_CHECK_LAUNCH();
  }
  if (bs > 0) {
    hipLaunchKernelGGL(sqsum_l4_bwd_kernel, dim3(l4_grid((long long)bs * kPos * (C / 4))), dim3(256), 0, hs, dxt, s.f1q, s.f3q, bs, bp,
                       d.n_maps_pad, (int)l4_pm_on(ctx), C, df1 + (size_t)d.Rp * C, df3 + (size_t)d.Rp * C);
    AIT_CHECK_LAUNCH();
  }
  (void)Rsk;
  struct Side { const float* x; int n; long long row0; const ait_sk_weights* w; const ait_sk_grads* g; float* dx; };
  const Side sides[2] = {{x_props, bp, 0, &w->sk_props, &grads->sk_props, d_x_props},
                         {x_query, bs, d.Rp, &w->sk_query, &grads->sk_query, d_x_query}};
  for (const Side& sd : sides) {
    if (sd.n == 0) continue;
    const float* a1 = df1 + (size_t)sd.row0 * C;
    const float* a3 = df3 + (size_t)sd.row0 * C;
    const long long rows = (long long)sd.n * kPos;
    const ait_conv_geom q1 = sk_geom(sd.n, 1), q3 = sk_geom(sd.n, 3);
    const int sp = rows >= 4096 ? 8 : 1;
    if (sd.g->w1) AIT_TRY(ait_conv_bwd_weight_f32(a1, C, sd.x, C, &q1, C, C, sd.g->w1, sp, s.zeros, kZeros, ctx, stream));
    if (sd.g->w3) AIT_TRY(ait_conv_bwd_weight_f32(a3, C, sd.x, C, &q3, C, C, sd.g->w3, sp, s.zeros, kZeros, ctx, stream));
    if (sd.g->b1) AIT_TRY(ait_colsum_f32(a1, rows, C, C, sd.g->b1, stream));
    if (sd.g->b3) AIT_TRY(ait_colsum_f32(a3, rows, C, C, sd.g->b3, stream));
    if (sd.dx) {
      // both branches' input gradient in ONE launch: by parity class of the 8x8 positions, the 1x1 branch as one more
      // tap of class (even, even) (gemm_f32.hip); two calls (3x3, then the 1x1 accumulated in place) where that
      // does not apply (row counts that are not a multiple of the tile)
      int rc = ait_conv_bwd_data_s2(a3, C, sd.w->w3, &q3, a1, C, sd.w->w1, C, C, nullptr, 0, sd.dx, C, s.zeros, ctx, stream);
      if (rc == AIT_EUNSUPPORTED) {
        AIT_TRY(ait_conv_bwd_data_f32(a3, C, sd.w->w3, &q3, C, C, nullptr, 0, sd.dx, C, s.zeros, kZeros, ctx, stream));
        rc = ait_conv_bwd_data_f32(a1, C, sd.w->w1, &q1, C, C, sd.dx, 0, sd.dx, C, s.zeros, kZeros, ctx, stream);     // (+= in place)
      }
      AIT_TRY(rc);
    }
  }
  return AIT_OK;
}
