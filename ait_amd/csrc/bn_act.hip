// ait_amd/csrc/bn_act.hip -- frozen batch-norm + residual + ReLU in one pass (and its backward).
//
// Every BatchNorm of the detector is frozen and runs in eval mode during training
// (lib/model/faster_rcnn/resnet_sys_transformer_sk_dilat.py:435-441,457-480), i.e. it is the
// per-channel affine map y = x*scale[c] + shift[c] with scale = gamma/sqrt(var+eps),
// shift = beta - mean*scale.  The bottleneck tail "bn3(conv3) ; out += residual ; relu"
// (:99-111) and "bn ; relu" (:88-96) are HBM-bound elementwise chains that the reference runs as
// 2-3 separate passes; here each is ONE read-modify-write pass (16 B per lane), forward and
// backward (dx = dy * [y>0] * scale[c], dres = dy * [y>0]).  The backward takes the incoming gradient as up to TWO addends
// (dy + dy2): a bottleneck's output feeds the next block's convolution path AND its shortcut, and summing the two gradients
// while this pass streams them replaces a separate add kernel (three tensor passes) by one more read.
#include "common.h"

namespace {

constexpr int kThreads = 256;

// MODE 0: scalar (any shape); 1: NCHW planes with HW % 4 == 0 (a float4 shares one channel);
// 2: channels-last rows (HW == 1, C % 4 == 0: a float4 covers four consecutive channels)
template <int MODE>
__global__ __launch_bounds__(kThreads) void bn_act_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ res, int relu, long long total, int C, int HW, float* __restrict__ y) {
  const long long stride = (long long)gridDim.x * kThreads * 4;
  for (long long i = ((long long)blockIdx.x * kThreads + threadIdx.x) * 4; i < total; i += stride) {
    float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 0) {
      const float4 a = *reinterpret_cast<const float4*>(x + i);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
      if (res) {
        const float4 b = *reinterpret_cast<const float4*>(res + i);
        r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w;
      }
      float s[4], t[4];
      if (MODE == 1) {
        const int c = (int)((i / HW) % C);
        s[0] = s[1] = s[2] = s[3] = scale[c];
        t[0] = t[1] = t[2] = t[3] = shift[c];
      } else {
        const int c = (int)(i % C);
        const float4 s4 = *reinterpret_cast<const float4*>(scale + c);
        const float4 t4 = *reinterpret_cast<const float4*>(shift + c);
        s[0] = s4.x; s[1] = s4.y; s[2] = s4.z; s[3] = s4.w;
        t[0] = t4.x; t[1] = t4.y; t[2] = t4.z; t[3] = t4.w;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        v[k] = v[k] * s[k] + t[k] + r[k];
        if (relu) v[k] = fmaxf(v[k], 0.f);
      }
      *reinterpret_cast<float4*>(y + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const long long j = i + k;
        if (j >= total) break;
        const int c = (int)((j / HW) % C);
        float o = x[j] * scale[c] + shift[c] + (res ? res[j] : 0.f);
        if (relu) o = fmaxf(o, 0.f);
        y[j] = o;
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void bn_act_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ dy2, const float* __restrict__ y, const float* __restrict__ scale,
    int relu, long long total, int C, int HW, float* __restrict__ dx, float* __restrict__ dres) {
  const long long stride = (long long)gridDim.x * kThreads * 4;
  for (long long i = ((long long)blockIdx.x * kThreads + threadIdx.x) * 4; i < total; i += stride) {
    if (MODE != 0) {
      const float4 g4 = *reinterpret_cast<const float4*>(dy + i);
      float g[4] = {g4.x, g4.y, g4.z, g4.w};
      if (dy2) {
        const float4 h4 = *reinterpret_cast<const float4*>(dy2 + i);
        g[0] += h4.x; g[1] += h4.y; g[2] += h4.z; g[3] += h4.w;
      }
      if (relu) {
        const float4 o = *reinterpret_cast<const float4*>(y + i);
        g[0] = o.x > 0.f ? g[0] : 0.f;
        g[1] = o.y > 0.f ? g[1] : 0.f;
        g[2] = o.z > 0.f ? g[2] : 0.f;
        g[3] = o.w > 0.f ? g[3] : 0.f;
      }
      if (dres) *reinterpret_cast<float4*>(dres + i) = make_float4(g[0], g[1], g[2], g[3]);
      if (MODE == 1) {
        const float s = scale[(int)((i / HW) % C)];
        *reinterpret_cast<float4*>(dx + i) = make_float4(g[0] * s, g[1] * s, g[2] * s, g[3] * s);
      } else {
        const float4 s4 = *reinterpret_cast<const float4*>(scale + (int)(i % C));
        *reinterpret_cast<float4*>(dx + i) = make_float4(g[0] * s4.x, g[1] * s4.y, g[2] * s4.z, g[3] * s4.w);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const long long j = i + k;
        if (j >= total) break;
        float g = dy[j] + (dy2 ? dy2[j] : 0.f);
        if (relu && !(y[j] > 0.f)) g = 0.f;
        if (dres) dres[j] = g;
        dx[j] = g * scale[(int)((j / HW) % C)];
      }
    }
  }
}

// ---- bf16 tensors (the C4 trunk of the bf16 configuration, BASELINE configs[4]): channels-last rows, eight values
// (16 B) per lane, f32 arithmetic, nearest-even rounding on the way out --------------------------------------------------
__device__ __forceinline__ void unpack8(const uint4 v, float (&o)[8]) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 t;
  t[0] = (__bf16)a;
  t[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
__global__ __launch_bounds__(kThreads) void bn_act_fwd16_kernel(const unsigned short* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const unsigned short* __restrict__ res,
                                                                int relu, long long total, int C, unsigned short* __restrict__ y) {
  const long long stride = (long long)gridDim.x * kThreads * 8;
  for (long long i = ((long long)blockIdx.x * kThreads + threadIdx.x) * 8; i < total; i += stride) {
    float v[8], r[8], s[8], t[8];
    unpack8(*reinterpret_cast<const uint4*>(x + i), v);
    if (res) unpack8(*reinterpret_cast<const uint4*>(res + i), r);
    const int c = (int)(i % C);
    *reinterpret_cast<float4*>(s) = *reinterpret_cast<const float4*>(scale + c);
    *reinterpret_cast<float4*>(s + 4) = *reinterpret_cast<const float4*>(scale + c + 4);
    *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(shift + c);
    *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(shift + c + 4);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      v[k] = v[k] * s[k] + t[k] + (res ? r[k] : 0.f);
      if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    *reinterpret_cast<uint4*>(y + i) = pack8(v);
  }
}
__global__ __launch_bounds__(kThreads) void bn_act_bwd16_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ dy2,
                                                                const unsigned short* __restrict__ y,
                                                                const float* __restrict__ scale, int relu, long long total, int C,
                                                                unsigned short* __restrict__ dx, unsigned short* __restrict__ dres) {
  const long long stride = (long long)gridDim.x * kThreads * 8;
  for (long long i = ((long long)blockIdx.x * kThreads + threadIdx.x) * 8; i < total; i += stride) {
    float g[8], o[8], s[8];
    const uint4 graw = *reinterpret_cast<const uint4*>(dy + i);
    unpack8(graw, g);
    if (dy2) {      // (the sum of the two bf16 addends is formed in f32 and rounded ONCE, into dres / dx below)
      unpack8(*reinterpret_cast<const uint4*>(dy2 + i), o);
#pragma unroll
      for (int k = 0; k < 8; k++) g[k] += o[k];
    }
    if (relu) {
      unpack8(*reinterpret_cast<const uint4*>(y + i), o);
#pragma unroll
      for (int k = 0; k < 8; k++) g[k] = o[k] > 0.f ? g[k] : 0.f;
    }
    if (dres) *reinterpret_cast<uint4*>(dres + i) = pack8(g);       // (one addend: a masked copy of bf16 values, exact)
    const int c = (int)(i % C);
    *reinterpret_cast<float4*>(s) = *reinterpret_cast<const float4*>(scale + c);
    *reinterpret_cast<float4*>(s + 4) = *reinterpret_cast<const float4*>(scale + c + 4);
#pragma unroll
    for (int k = 0; k < 8; k++) g[k] *= s[k];
    *reinterpret_cast<uint4*>(dx + i) = pack8(g);
  }
}

inline unsigned grid_for(long long total) {
  long long b = (total / 4 + kThreads - 1) / kThreads;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

AIT_API int ait_bn_act_fwd(const float* x, const float* scale, const float* shift,
                           const float* residual, int relu, long long n, int C, int HW, float* y,
                           void* stream) {
  if (n < 0 || C <= 0 || HW <= 0) return AIT_EINVAL;
  const long long total = n * C * HW;
  if (total == 0) return AIT_OK;
  if (!x || !scale || !shift || !y) return AIT_EINVAL;
  const bool al = aligned16(x) && aligned16(y) && (!residual || aligned16(residual));
  if (al && HW % 4 == 0)
    hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(grid_for(total)), dim3(kThreads), 0,
                       ait_stream(stream), x, scale, shift, residual, relu, total, C, HW, y);
  else if (al && HW == 1 && C % 4 == 0 && aligned16(scale) && aligned16(shift))
    hipLaunchKernelGGL(bn_act_fwd_kernel<2>, dim3(grid_for(total)), dim3(kThreads), 0,
                       ait_stream(stream), x, scale, shift, residual, relu, total, C, HW, y);
  else
    hipLaunchKernelGGL(bn_act_fwd_kernel<0>, dim3(grid_for(total)), dim3(kThreads), 0,
                       ait_stream(stream), x, scale, shift, residual, relu, total, C, HW, y);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_bn_act_bwd(const float* dy, const float* dy2, const float* y, const float* scale, int relu,
                           long long n, int C, int HW, float* dx, float* dres, void* stream) {
  if (n < 0 || C <= 0 || HW <= 0) return AIT_EINVAL;
  const long long total = n * C * HW;
  if (total == 0) return AIT_OK;
  if (!dy || !scale || !dx || (relu && !y)) return AIT_EINVAL;
  const bool al = aligned16(dy) && aligned16(dx) && (!relu || aligned16(y)) && (!dres || aligned16(dres)) && (!dy2 || aligned16(dy2));
  if (al && HW % 4 == 0)
    hipLaunchKernelGGL(bn_act_bwd_kernel<1>, dim3(grid_for(total)), dim3(kThreads), 0,
                       ait_stream(stream), dy, dy2, y, scale, relu, total, C, HW, dx, dres);
  else if (al && HW == 1 && C % 4 == 0 && aligned16(scale))
    hipLaunchKernelGGL(bn_act_bwd_kernel<2>, dim3(grid_for(total)), dim3(kThreads), 0,
                       ait_stream(stream), dy, dy2, y, scale, relu, total, C, HW, dx, dres);
  else
    hipLaunchKernelGGL(bn_act_bwd_kernel<0>, dim3(grid_for(total)), dim3(kThreads), 0,
                       ait_stream(stream), dy, dy2, y, scale, relu, total, C, HW, dx, dres);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_bn_act_fwd_bf16(const void* x, const float* scale, const float* shift, const void* residual, int relu,
                                long long rows, int C, void* y, void* stream) {
  if (rows < 0 || C <= 0) return AIT_EINVAL;
  const long long total = rows * C;
  if (total == 0) return AIT_OK;
  if (!x || !scale || !shift || !y) return AIT_EINVAL;
  if ((C % 8) || !aligned16(x) || !aligned16(y) || (residual && !aligned16(residual)) || !aligned16(scale) || !aligned16(shift))
    return AIT_EUNSUPPORTED;
  long long b = (total / 8 + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(bn_act_fwd16_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(kThreads), 0, ait_stream(stream),
                     static_cast<const unsigned short*>(x), scale, shift, static_cast<const unsigned short*>(residual), relu, total, C,
                     static_cast<unsigned short*>(y));
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_bn_act_bwd_bf16(const void* dy, const void* dy2, const void* y, const float* scale, int relu, long long rows, int C,
                                void* dx, void* dres, void* stream) {
  if (rows < 0 || C <= 0) return AIT_EINVAL;
  const long long total = rows * C;
  if (total == 0) return AIT_OK;
  if (!dy || !scale || !dx || (relu && !y)) return AIT_EINVAL;
  if ((C % 8) || !aligned16(dy) || (dy2 && !aligned16(dy2)) || !aligned16(dx) || (relu && !aligned16(y)) || (dres && !aligned16(dres)) || !aligned16(scale))
    return AIT_EUNSUPPORTED;
  long long b = (total / 8 + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(bn_act_bwd16_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(kThreads), 0, ait_stream(stream),
                     static_cast<const unsigned short*>(dy), static_cast<const unsigned short*>(dy2), static_cast<const unsigned short*>(y), scale, relu, total, C,
                     static_cast<unsigned short*>(dx), static_cast<unsigned short*>(dres));
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

// ---------------------------------------------------------------------------------------------
// SKBlock tail: out = relu(a)^2 + relu(b)^2  (lib/model/modules/blocks_sys_transformer_sk_dilat.py
// :966-981 as actually executed: two conv+ReLU branches, `v = f * f`, sum over the branches; the
// branch-attention weights are computed upstream but never used).  One pass instead of five.
// ---------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(kThreads) void sk_sqsum_fwd_kernel(const float4* __restrict__ a,
                                                                const float4* __restrict__ b,
                                                                long long n4, float4* __restrict__ y) {
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    const float4 x = a[i], z = b[i];
    float4 o;
    float p, q;
    p = fmaxf(x.x, 0.f); q = fmaxf(z.x, 0.f); o.x = p * p + q * q;
    p = fmaxf(x.y, 0.f); q = fmaxf(z.y, 0.f); o.y = p * p + q * q;
    p = fmaxf(x.z, 0.f); q = fmaxf(z.z, 0.f); o.z = p * p + q * q;
    p = fmaxf(x.w, 0.f); q = fmaxf(z.w, 0.f); o.w = p * p + q * q;
    y[i] = o;
  }
}

__global__ __launch_bounds__(kThreads) void sk_sqsum_bwd_kernel(
    const float4* __restrict__ dy, const float4* __restrict__ a, const float4* __restrict__ b,
    long long n4, float4* __restrict__ da, float4* __restrict__ db) {
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    const float4 g = dy[i], x = a[i], z = b[i];
    // d/da relu(a)^2 = 2*relu(a)
    da[i] = make_float4(2.f * fmaxf(x.x, 0.f) * g.x, 2.f * fmaxf(x.y, 0.f) * g.y,
                        2.f * fmaxf(x.z, 0.f) * g.z, 2.f * fmaxf(x.w, 0.f) * g.w);
    db[i] = make_float4(2.f * fmaxf(z.x, 0.f) * g.x, 2.f * fmaxf(z.y, 0.f) * g.y,
                        2.f * fmaxf(z.z, 0.f) * g.z, 2.f * fmaxf(z.w, 0.f) * g.w);
  }
}

}  // namespace

AIT_API int ait_sk_sqsum_fwd(const float* a, const float* b, long long n, float* y, void* stream) {
  if (n < 0) return AIT_EINVAL;
  if (n == 0) return AIT_OK;
  if (!a || !b || !y) return AIT_EINVAL;
  if ((n & 3) || !aligned16(a) || !aligned16(b) || !aligned16(y)) return AIT_EUNSUPPORTED;
  hipLaunchKernelGGL(sk_sqsum_fwd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, ait_stream(stream),
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), n / 4,
                     reinterpret_cast<float4*>(y));
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_sk_sqsum_bwd(const float* dy, const float* a, const float* b, long long n, float* da,
                             float* db, void* stream) {
  if (n < 0) return AIT_EINVAL;
  if (n == 0) return AIT_OK;
  if (!dy || !a || !b || !da || !db) return AIT_EINVAL;
  if ((n & 3) || !aligned16(dy) || !aligned16(a) || !aligned16(b) || !aligned16(da) || !aligned16(db))
    return AIT_EUNSUPPORTED;
  hipLaunchKernelGGL(sk_sqsum_bwd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, ait_stream(stream),
                     reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(a),
                     reinterpret_cast<const float4*>(b), n / 4, reinterpret_cast<float4*>(da),
                     reinterpret_cast<float4*>(db));
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}
