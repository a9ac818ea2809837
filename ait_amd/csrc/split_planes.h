// ait_amd/csrc/split_planes.h -- an f32 value as three bf16 planes, and the six-term product of two such values on
// v_mfma_f32_32x32x16_bf16 (the product form of this library: gemm_f32_impl.h "f32 products on the bf16 matrix pipe").
// Shared by the GEMM kernels (gemm_f32_impl.h), the P3 conversion (p3_impl.h) and the attention tiles (attn.hip).
#pragma once
#include "common.h"

namespace ait_gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Planes { bf16x8 h, m, l; };

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned round2(float x0, float x1) {         // v_cvt_pk_bf16_f32: nearest even, NaN stays NaN
  bf16x2 t;
  t[0] = (__bf16)x0;
  t[1] = (__bf16)x1;
  return __builtin_bit_cast(unsigned, t);
}

// x = h + m + l exactly (the remainders are exact f32 subtractions, l needs at most 8 significant bits).
//   RNE  : h = bf16(x) rounded to nearest, m = the top 8 bits of the (either-signed) remainder: |m| <= 2^-9 |x|,
//          |l| < 2^-16 |x|; the partial products the six-term form drops (m l' + l m' + l l') are <= 2^-23 |x x'|;
//   else : h, m by TRUNCATION (the top 16 bits): |m| < 2^-7 |x|, |l| < 2^-15 |x|, dropped terms <= 2^-21 |x x'|.
// 11 vector instructions per two values either way (tests/test_host_cpu.py restates both in numpy).
template <bool RNE>
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  if constexpr (RNE) {
    // h to NEAREST (v_cvt_pk_bf16_f32); the remainder r = x - h is exact, has at most 16 significant bits and EITHER
    // sign whatever the sign of x, so its two halves are taken by truncation (v_perm_b32: one cycle cheaper per
    // instruction than the conversion, and the kernel that splits both operands is bound by vector issue):
    // m = top 8 bits of r, l = r - m (exact, <= 8 bits), both with the sign of r -- zero-mean planes, no drift on
    // same-signed data (profiles/r04_split_bias.txt), |m| <= 2^-9 |x|, |l| < 2^-16 |x|: dropped terms <= 2^-23 |x x'|.
    h = round2(x0, x1);                                                   // (bf16(x1) << 16) | bf16(x0)
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  } else {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                      // (hi16(x1) << 16) | hi16(x0)
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
}

// TERMS = 6: the exact 3-way split; TERMS = 1: the operand rounded to bf16 (only plane h is formed)
template <int TERMS = 6, bool RNE = false>
__device__ __forceinline__ Planes split8(const float4& p, const float4& q) {     // k = 8*lk + 0..3 | 4..7
  u32x4 h, m, l;
  if constexpr (TERMS == 1) {
    h[0] = round2(p.x, p.y); h[1] = round2(p.z, p.w); h[2] = round2(q.x, q.y); h[3] = round2(q.z, q.w);
    Planes r;
    r.h = __builtin_bit_cast(bf16x8, h);
    r.m = r.h;
    r.l = r.h;
    return r;
  }
  unsigned a, b, c;
  split2<RNE>(p.x, p.y, a, b, c); h[0] = a; m[0] = b; l[0] = c;
  split2<RNE>(p.z, p.w, a, b, c); h[1] = a; m[1] = b; l[1] = c;
  split2<RNE>(q.x, q.y, a, b, c); h[2] = a; m[2] = b; l[2] = c;
  split2<RNE>(q.z, q.w, a, b, c); h[3] = a; m[3] = b; l[3] = c;
  Planes r;
  r.h = __builtin_bit_cast(bf16x8, h);
  r.m = __builtin_bit_cast(bf16x8, m);
  r.l = __builtin_bit_cast(bf16x8, l);
  return r;
}

template <int TERMS = 6>
__device__ __forceinline__ f32x16 mfma_split(const Planes& a, const Planes& b, f32x16 acc) {
  if constexpr (TERMS == 1) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
}


}  // namespace ait_gemm
