// ait_amd/csrc/attn_impl.h (csrc/attn.hip: forward, csrc/attn_bwd.hip: backward -- one kernel per translation unit:
// co-compiled, each perturbs the other's register allocation) -- the proposal x query score matrix of AIT on the matrix cores.
//
//   ait_attn_fwd   per (sequence, head):  S = (Q K^T) * scale ; mask ; P = softmax(S) ;
//                  O = dropout(P) V                (lib/model/system/Modules.py:16-29)
//   ait_attn_bwd   the five backward products dV, dP, dS, dQ, dK of the same unit.
//
// Shapes on the AIT path are fixed: T = 64 tokens (8x8 query cells; 7x7 proposal cells zero
// padded to 64, lib/model/system/Models.py:269-270), d_k = d_v = 64, 8 heads.  One WAVEFRONT
// owns one (sequence, head) unit: its 64x64 score tile is exactly 2x2 MFMA tiles of 32x32 (64 accumulator
// VGPRs), so the whole softmax lives in registers and the probabilities never round-trip through HBM inside
// the kernel.  Four units per 256-thread workgroup.
//
// Products of the FORWARD: the library's f32 product form (split_planes.h): every operand value split exactly into
// three bf16 planes in registers, six v_mfma_f32_32x32x16_bf16 per 32x32x16 block, f32 accumulate -- 96 MFMAs of 32
// cycles per 64x64x64 product against 128 of 64 cycles on v_mfma_f32_32x32x2_f32 (0.246 ms per block against 0.294).
// The backward stays on the f32 instruction (see attn_bwd_kernel).  A lane's operand fragment of a k-block is eight
// consecutive k-values of its row / column.
//
// Operands are staged through a wave-private 64x65 fp32 LDS panel (odd pitch:
// both access patterns the MFMA needs -- "rows down the lanes" for X as a left operand / X^T as
// a right operand, and "columns along the lanes" for the other two cases -- are conflict-free
// ds_read_b32); the other operand of every product lives in registers (see OpRegs).  16.6 KB per
// wave, 66.5 KB per workgroup -> 2 workgroups (8 waves, 2 per SIMD) per CU.
// Global loads/stores are whole 256-B head rows (64 floats), coalesced.
//
// Masks are the two compile-time predicates of the reference: key padding
// (k < n_valid) and causal (k <= q); masked scores are set to -1e9 before the softmax exactly
// as masked_fill does, so a masked probability is exactly 0.
//
// P (pre-dropout) is written to HBM once for the backward pass; the dropout mask itself is
// recomputed from the stateless hash.
#pragma once
#include "common.h"
#include "split_planes.h"

namespace ait_attn {


using ait_gemm::f32x16;
using ait_gemm::Planes;

constexpr int T = 64, D = 64, PITCH = 65;
constexpr bool kF32Pairs = ait_lab::Knobs::f32_pairs;      // (lab: the column-paired right-operand loads for f32 q / k / v as well)
constexpr int kPanel = T * PITCH;          // floats per LDS panel
// a [rows, 8 heads x 64] tensor as the fused kernels address it: at least 512 columns per row, the pitch a whole number of
// 16-byte vectors (4 f32 / 8 bf16), the base 16-byte aligned
inline bool rows_ok(const void* p, int ld, int is_bf16) {
  return ld >= 8 * D && (ld % (is_bf16 ? 8 : 4)) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}
constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned long long idx,
                                            float p, float inv_keep) {
  unsigned h = mix32((unsigned)idx ^ mix32((unsigned)(idx >> 32) + (unsigned)seed) ^
                     (unsigned)(seed >> 32) * 0x9e3779b9u);
  float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.f;
}

// The same decision for element base + off of a block whose base is wave-uniform and a multiple of a power of two > off
// (a unit's 4096 probabilities, a sequence's 32768 fc outputs): the high word of the index and everything that depends
// on it are computed once per block on the scalar side -- 9 vector instructions per element instead of 17, two of them
// the quarter-rate v_mul_lo_u32 either way.  Bit-identical to drop_scale(seed, base + off, ...).
struct DropBlock {
  unsigned lo0, key;
  __device__ __forceinline__ DropBlock(unsigned long long seed, unsigned long long base)
      : lo0((unsigned)base), key(mix32((unsigned)(base >> 32) + (unsigned)seed) ^ (unsigned)(seed >> 32) * 0x9e3779b9u) {}
  __device__ __forceinline__ float scale(unsigned off, float p, float inv_keep) const {
    const unsigned h = mix32((lo0 + off) ^ key);
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.f;
  }
};

// One 64x64 operand panel on its way global -> registers -> LDS [64][65].  All 16 loads of a lane
// (16 B each: lane = (row & 3, 16-B column chunk), 4 rows per wave-wide load) are issued back to
// back, so a panel costs ONE memory round trip, and they can be issued long before the panel is
// needed (the caller runs the previous product in between).  The LDS writes are four ds_write_b32
// per chunk; with the odd pitch, (row & 3) + 4*chunk + j covers all 64 banks exactly once.
struct Stage {
  float4 v[16];
  // rows >= `rows` (K / V panels of an unpadded memory) read as zero; their loads are clamped to
  // the last valid row so that nothing is branched around
  __device__ __forceinline__ void load(const float* __restrict__ g, int ld, int lane, int rows = T) {
    const int c = (lane & 15) * 4, r0 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r = min(i * 4 + r0, rows - 1);
      v[i] = *reinterpret_cast<const float4*>(g + (unsigned)(r * ld + c));      // unsigned 32-bit offset off a wave-uniform (SGPR) base: saddr + voffset addressing
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ s, int lane, int rows = T) const {
    const int c = (lane & 15) * 4, r0 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r = i * 4 + r0;
      const bool live = r < rows;
      float* __restrict__ d = s + r * PITCH + c;
      d[0] = live ? v[i].x : 0.f;
      d[1] = live ? v[i].y : 0.f;
      d[2] = live ? v[i].z : 0.f;
      d[3] = live ? v[i].w : 0.f;
    }
  }
};

// The same panel from a bf16 tensor (the bf16-storage mode's q / k / v, csrc/transformer.hip): 16 B = eight values per
// lane and load (lane = (row & 7, 16-B chunk): eight rows per wave-wide load, eight loads per panel), the values widened
// to f32 on their way into the (f32) panel, so every product downstream is unchanged -- and exact for these operands, whose
// two lower planes are zero.  LDS writes: (row & 7) + 8 * chunk + j covers all 64 banks exactly once.
__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
struct Stage16 {
  uint4 v[8];
  __device__ __forceinline__ void load(const unsigned short* __restrict__ g, int ld, int lane, int rows = T) {
    const int c = (lane & 7) * 8, r0 = lane >> 3;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int r = min(i * 8 + r0, rows - 1);
      v[i] = *reinterpret_cast<const uint4*>(g + (unsigned)(r * ld + c));
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ s, int lane, int rows = T) const {
    const int c = (lane & 7) * 8, r0 = lane >> 3;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int r = i * 8 + r0;
      const bool live = r < rows;
      float* __restrict__ d = s + r * PITCH + c;
      d[0] = live ? bf16_lo(v[i].x) : 0.f; d[1] = live ? bf16_hi(v[i].x) : 0.f;
      d[2] = live ? bf16_lo(v[i].y) : 0.f; d[3] = live ? bf16_hi(v[i].y) : 0.f;
      d[4] = live ? bf16_lo(v[i].z) : 0.f; d[5] = live ? bf16_hi(v[i].z) : 0.f;
      d[6] = live ? bf16_lo(v[i].w) : 0.f; d[7] = live ? bf16_hi(v[i].w) : 0.f;
    }
  }
};

// ---- operand fragments ---------------------------------------------------------------------------------------
// A 64x64x64 product runs as four k-blocks of 16; in block kb lane (li, lk) holds, for each of its two 32-row (or
// 32-column) tiles t, the eight values k = 16 kb + 8 lk + 0..7 of row / column 32 t + li -- the operand layout of
// v_mfma_f32_32x32x16_bf16.  Fragments come from the wave's LDS panel (either orientation: conflict-free
// ds_read_b32 with the odd pitch) or from registers (OpRegs).
//   rows:  X(i, k) = Xs[i][k]  (the lane's row down the panel)        cols:  X(k, j) = Xs[k][j]
struct Frag { float4 lo, hi; };
__device__ __forceinline__ Frag frag_rows(const float* __restrict__ Xs, int t, int kb, int li, int lk) {
  const float* p = Xs + (t * 32 + li) * PITCH + 16 * kb + 8 * lk;
  return Frag{make_float4(p[0], p[1], p[2], p[3]), make_float4(p[4], p[5], p[6], p[7])};
}
__device__ __forceinline__ Frag frag_cols(const float* __restrict__ Xs, int t, int kb, int li, int lk) {
  const float* p = Xs + (16 * kb + 8 * lk) * PITCH + t * 32 + li;
  return Frag{make_float4(p[0], p[PITCH], p[2 * PITCH], p[3 * PITCH]),
              make_float4(p[4 * PITCH], p[5 * PITCH], p[6 * PITCH], p[7 * PITCH])};
}

// ---- register-resident operands -----------------------------------------------------------------
// A right operand that is row-major in memory (R(k,j) = X[k][j]) is loaded straight from global into registers --
// lane li reads 32 consecutive floats of row k, whole 128-B segments -- so it never needs an LDS panel.  A left
// operand "rows down the lanes" is not coalescable from global; it is staged once through the wave's panel and then
// lifted into registers, which frees the panel for the other operand.  Either way a wave needs ONE 64x65 panel, i.e.
// 66.5 KB per workgroup and two workgroups (2 waves per SIMD) per CU.
struct OpRegs {
  float v[2][4][8];   // [tile][k-block][k within the lane's eight]
  __device__ __forceinline__ Frag frag(int t, int kb) const {
    return Frag{make_float4(v[t][kb][0], v[t][kb][1], v[t][kb][2], v[t][kb][3]),
                make_float4(v[t][kb][4], v[t][kb][5], v[t][kb][6], v[t][kb][7])};
  }
};

// B(k, j) = g[k*ld + j] for k < rows, else 0
__device__ __forceinline__ void breg_load(OpRegs& b, const float* __restrict__ g, int ld, int lane,
                                          int rows = T) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < 4; kb++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = 16 * kb + 8 * lk + j;
      const unsigned off = (unsigned)(min(k, rows - 1) * ld + li);          // unsigned 32-bit offset off a wave-uniform base
      const float x0 = g[off], x1 = g[off + 32];
      b.v[0][kb][j] = k < rows ? x0 : 0.f;
      b.v[1][kb][j] = k < rows ? x1 : 0.f;
    }
}

// The same from a bf16 tensor, with the COLUMNS PAIRED: lane li loads ONE dword of row k -- columns 2 li and 2 li + 1, a
// whole 128-B row per half-wave -- and takes the low half as its column of tile 0, the high half as its column of tile 1:
// B's column of (tile t, lane li) is 2 li + t, not 32 t + li.  A permutation of B's columns is the same permutation of
// the product's: whoever reads the accumulators of such a product addresses columns with acc_col_of<true>.  32 loads
// instead of the 64 two-byte loads the unpaired layout would take.
__device__ __forceinline__ void breg_load_pairs(OpRegs& b, const unsigned short* __restrict__ g, int ld, int lane, int rows = T) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < 4; kb++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = 16 * kb + 8 * lk + j;
      const unsigned w = *reinterpret_cast<const unsigned*>(g + (unsigned)(min(k, rows - 1) * ld + 2 * li));
      b.v[0][kb][j] = k < rows ? bf16_lo(w) : 0.f;
      b.v[1][kb][j] = k < rows ? bf16_hi(w) : 0.f;
    }
}

// ... and the column-paired form for an f32 tensor: one 8-byte load per lane and row (columns 2 li, 2 li + 1)
__device__ __forceinline__ void breg_load_pairs(OpRegs& b, const float* __restrict__ g, int ld, int lane, int rows = T) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < 4; kb++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = 16 * kb + 8 * lk + j;
      const float2 w = *reinterpret_cast<const float2*>(g + (unsigned)(min(k, rows - 1) * ld + 2 * li));
      b.v[0][kb][j] = k < rows ? w.x : 0.f;
      b.v[1][kb][j] = k < rows ? w.y : 0.f;
    }
}

// A(i, k) = Ls[i][k] lifted out of an LDS panel
__device__ __forceinline__ void areg_from_lds(OpRegs& a, const float* __restrict__ Ls, int lane) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
      for (int j = 0; j < 8; j++) a.v[t][kb][j] = Ls[(t * 32 + li) * PITCH + 16 * kb + 8 * lk + j];
}

__device__ __forceinline__ float frag_at(const Frag& f, int j) {
  return j == 0 ? f.lo.x : j == 1 ? f.lo.y : j == 2 ? f.lo.z : j == 3 ? f.lo.w : j == 4 ? f.hi.x : j == 5 ? f.hi.y : j == 6 ? f.hi.z : f.hi.w;
}
// acc += L . R over the four k-blocks; fa(t, kb) / fb(t, kb) return the lane's fragment of left tile t / right tile t.
//   SPLIT = true : the library's f32 product form, 24 v_mfma_f32_32x32x16_bf16 + 176 vector instructions per block
//   SPLIT = false: v_mfma_f32_32x32x2_f32, 32 MFMAs of twice the cycles per block and no vector work -- MFMA step j of
//                  block kb multiplies k = 16 kb + j (lane half 0) and 16 kb + 8 + j (half 1): any pairing of k-values is
//                  a valid step as long as both operands use it.
// The forward and the fused backward (mha_fused_bwd.hip) use the split form, attn_bwd_kernel the f32 instruction (it says why).
// PA / PB: the planes operand A / B HAS -- 3: any f32 value; 1: the values are bf16 already (the bf16-storage mode's
// q / k / v, widened on their way in): that operand's lower planes are zero, its split is one conversion per pair, and the
// product is EXACT in three MFMAs (one, if both operands are bf16) instead of six.
template <bool SPLIT, int PA_ = 3, int PB_ = 3, class FA, class FB>
__device__ __forceinline__ void mm_split(FA fa, FB fb, f32x16 (&acc)[2][2]) {
  constexpr int PA = ait_lab::Knobs::no_bf16_planes ? 3 : PA_, PB = ait_lab::Knobs::no_bf16_planes ? 3 : PB_;
#pragma unroll
  for (int kb = 0; kb < 4; kb++) {
    // (fence per k-block: the loop must be fully unrolled -- static register indices -- but the scheduler must not
    // lift every block's fragments and planes above the first MFMA)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!SPLIT) {
      const Frag x0 = fa(0, kb), x1 = fa(1, kb), y0 = fb(0, kb), y1 = fb(1, kb);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float a0 = frag_at(x0, j), a1 = frag_at(x1, j), b0 = frag_at(y0, j), b1 = frag_at(y1, j);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      continue;
    }
    Planes pa[2], pb[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const Frag x = fa(t, kb);
      pa[t] = ait_gemm::split8<PA == 1 ? 1 : 6, true>(x.lo, x.hi);
      const Frag y = fb(t, kb);
      pb[t] = ait_gemm::split8<PB == 1 ? 1 : 6, true>(y.lo, y.hi);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) {
        if constexpr (PA == 3 && PB == 3) {
          acc[a][b] = ait_gemm::mfma_split<6>(pa[a], pb[b], acc[a][b]);
        } else if constexpr (PA == 3) {       // (smallest terms first, as the six-term form)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[a].l, pb[b].h, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[a].m, pb[b].h, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[a].h, pb[b].h, acc[a][b], 0, 0, 0);
        } else if constexpr (PB == 3) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[a].h, pb[b].l, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[a].h, pb[b].m, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[a].h, pb[b].h, acc[a][b], 0, 0, 0);
        } else {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[a].h, pb[b].h, acc[a][b], 0, 0, 0);
        }
      }
  }
}
// the four operand placements the kernels use
//   A in registers, R(k, j) = Rs[j][k]  (right operand transposed, from the panel)
template <bool SPLIT = true, int PA = 3, int PB = 3>
__device__ __forceinline__ void mm_areg_bldsT(const OpRegs& a, const float* __restrict__ Rs,
                                              f32x16 (&acc)[2][2], int lane) {
  const int li = lane & 31, lk = lane >> 5;
  mm_split<SPLIT, PA, PB>([&](int t, int kb) { return a.frag(t, kb); }, [&](int t, int kb) { return frag_rows(Rs, t, kb, li, lk); }, acc);
}
//   L from the panel (L(i,k) = Ls[i][k], or Ls[k][i] with LT), B in registers
template <bool LT, bool SPLIT = true, int PA = 3, int PB = 3>
__device__ __forceinline__ void mm_alds_breg(const float* __restrict__ Ls, const OpRegs& b,
                                             f32x16 (&acc)[2][2], int lane) {
  const int li = lane & 31, lk = lane >> 5;
  mm_split<SPLIT, PA, PB>([&](int t, int kb) { return LT ? frag_cols(Ls, t, kb, li, lk) : frag_rows(Ls, t, kb, li, lk); },
                          [&](int t, int kb) { return b.frag(t, kb); }, acc);
}

__device__ __forceinline__ void zero(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
}

// accumulator element (a, b, r) of this lane sits at row / col:
__device__ __forceinline__ int acc_row(int a, int r, int lane) {
  return a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int b, int lane) { return b * 32 + (lane & 31); }
// ... of a product whose right operand came from breg_load_pairs (PAIRED), or not
template <bool PAIRED>
__device__ __forceinline__ int acc_col_of(int b, int lane) { return PAIRED ? 2 * (lane & 31) + b : b * 32 + (lane & 31); }

// reductions across the 32 lanes that hold one accumulator row (same lane>>5), result in every lane -- on the vector
// pipe only: four DPP steps inside a row of 16 lanes (quad_perm 1032 / 2301, then row_half_mirror / row_mirror, which
// pair equal-valued groups exactly as xor 4 / xor 8 would), and gfx950's v_permlane16_swap_b32 for the step between the
// two rows.  (__shfl_xor compiles to ds_bpermute_b32: an LDS-pipe round trip per step -- the 320 of them in the softmax
// were 45 % of the attention tile's time.)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// v of lane ^ 16 beside the lane's own: a = b = v, then rows 1 / 3 of a swap with rows 0 / 2 of b
__device__ __forceinline__ void swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float half_sum(float v) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);     // row_half_mirror
  v += dpp_f<0x140>(v);     // row_mirror
  float w = v;
  swap16(v, w);
  return v + w;
}
__device__ __forceinline__ float half_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  float w = v;
  swap16(v, w);
  return fmaxf(v, w);
}

// e^x for x <= 0 (softmax arguments) in six vector instructions: v_exp_f32 of x log2(e), the product carried in two
// floats (t + lo) so that the result is as close as expf's (|rel| < 2e-7) -- libm's expf is ~20 instructions of range
// and denormal handling the softmax does not need (masked scores, -1e9, underflow to exactly 0 either way).
__device__ __forceinline__ float exp_neg(float x) {
  const float t = x * 1.44269504f;
  const float lo = fmaf(x, 1.44269504f, -t) + x * 1.92596303e-8f;      // rounding of the product + log2(e) - float(log2(e))
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, lo * 0.693147182f, e);
}

template <typename F>
__device__ __forceinline__ void for_acc(f32x16 (&acc)[2][2], int lane, F f) {
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        acc[a][b][r] = f((float)acc[a][b][r], acc_row(a, r, lane), acc_col(b, lane));
}

__device__ __forceinline__ void acc_to_global(const f32x16 (&acc)[2][2], float* __restrict__ g,
                                              int ld, int lane, float mul) {
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        g[(unsigned)(acc_row(a, r, lane) * ld + acc_col(b, lane))] = acc[a][b][r] * mul;
}
__device__ __forceinline__ void acc_to_global_rows(const f32x16 (&acc)[2][2], float* __restrict__ g,
                                                   int ld, int lane, int rows) {
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = acc_row(a, r, lane);
        if (row < rows) g[(unsigned)(row * ld + acc_col(b, lane))] = acc[a][b][r];
      }
}
__device__ __forceinline__ void acc_to_lds(const f32x16 (&acc)[2][2], float* __restrict__ s,
                                           int lane) {
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) s[acc_row(a, r, lane) * PITCH + acc_col(b, lane)] = acc[a][b][r];
}

struct AttnArgs {
  const float *q, *k, *v;
  int ldq, ldk, ldv;
  int n_seq, H;
  int mask_mode, n_valid;
  int kv_rows;   // rows per sequence in the K / V tensors (64, or fewer when the memory is unpadded)
  float scale, p;
  unsigned long long seed;
};

// scale, mask (masked scores are -1e9 exactly as masked_fill writes them)
__device__ __forceinline__ void scale_mask(f32x16 (&acc)[2][2], int lane, const AttnArgs& g) {
  const int mode = g.mask_mode, nv = g.n_valid, kvr = g.kv_rows;
  for_acc(acc, lane, [&](float x, int row, int col) {
    const bool dead = col >= kvr || (mode == 1 && col >= nv) || (mode == 2 && col > row);
    return dead ? -1e9f : x * g.scale;
  });
}

constexpr size_t kLds = 0;  // panels are static LDS (66.5 KB per workgroup)

inline bool bad(int n_seq, int H, int Tt, int d, int mask_mode, int n_valid, float p) {
  return n_seq < 0 || H <= 0 || mask_mode < 0 || mask_mode > 2 || p < 0.f || p >= 1.f ||
         (mask_mode == 1 && (n_valid <= 0 || n_valid > Tt));
}


}  // namespace ait_attn
