// ait_amd/csrc/attn.hip -- the proposal x query score matrix of AIT on the fp32 matrix cores.
//
//   ait_attn_fwd   per (sequence, head):  S = (Q K^T) * scale ; mask ; P = softmax(S) ;
//                  O = dropout(P) V                (lib/model/system/Modules.py:16-29)
//   ait_attn_bwd   the five backward products dV, dP, dS, dQ, dK of the same unit.
//
// Shapes on the AIT path are fixed: T = 64 tokens (8x8 query cells; 7x7 proposal cells zero
// padded to 64, lib/model/system/Models.py:269-270), d_k = d_v = 64, 8 heads.  One WAVEFRONT
// owns one (sequence, head) unit: its 64x64 score tile is exactly 2x2 v_mfma_f32_32x32x2_f32
// tiles (64 accumulator VGPRs), so the whole softmax lives in registers and the probabilities
// never round-trip through HBM inside the kernel.  Four units per 256-thread workgroup.
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
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int T = 64, D = 64, PITCH = 65;
constexpr int kPanel = T * PITCH;          // floats per LDS panel
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
      v[i] = *reinterpret_cast<const float4*>(g + (r * ld + c));      // 32-bit offset off a wave-uniform base
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

// acc[a][b] += sum_k L(i,k) * R(k,j) for a 64x64x64 product out of two LDS panels.
//   LT = false: L(i,k) = Ls[i][k]      LT = true: L(i,k) = Ls[k][i]   (left operand transposed)
//   RT = false: R(k,j) = Rs[k][j]      RT = true: R(k,j) = Rs[j][k]   (right operand transposed)
template <bool LT, bool RT>
__device__ __forceinline__ void mm64(const float* __restrict__ Ls, const float* __restrict__ Rs,
                                     f32x16 (&acc)[2][2], int lane) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll 8
  for (int k = 0; k < D; k += 2) {
    const int kk = k + lk;
    float a0, a1, b0, b1;
    if (LT) {
      a0 = Ls[kk * PITCH + li];
      a1 = Ls[kk * PITCH + li + 32];
    } else {
      a0 = Ls[li * PITCH + kk];
      a1 = Ls[(li + 32) * PITCH + kk];
    }
    if (RT) {
      b0 = Rs[li * PITCH + kk];
      b1 = Rs[(li + 32) * PITCH + kk];
    } else {
      b0 = Rs[kk * PITCH + li];
      b1 = Rs[kk * PITCH + li + 32];
    }
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// ---- register-resident operands -----------------------------------------------------------------
// The MFMA wants, per k-step j (k = 2j + lk):  A operand  lane(li,lk) = L(li + 32a, k),
//                                              B operand  lane(li,lk) = R(k, li + 32b).
// A B operand that is row-major in memory (R(k,j) = X[k][j]) can be loaded straight from global
// into registers -- lane li reads 32 consecutive floats of row k, whole 128-B segments -- so it
// never needs an LDS panel.  An A operand "rows down the lanes" is not coalescable from global;
// it is staged once through the wave's panel and then lifted into registers, which frees the
// panel for the other operand.  Either way a wave needs ONE 64x65 panel instead of two, i.e.
// 66.5 KB per workgroup and two workgroups (2 waves per SIMD) per CU.
struct OpRegs {
  float v[2][32];   // [tile][k-step]
};

// B(k, j) = g[k*ld + j] for k < rows, else 0
__device__ __forceinline__ void breg_load(OpRegs& b, const float* __restrict__ g, int ld, int lane,
                                          int rows = T) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int j = 0; j < 32; j++) {
    const int k = 2 * j + lk;
    const int off = min(k, rows - 1) * ld + li;          // 32-bit offset off a wave-uniform base
    const float x0 = g[off], x1 = g[off + 32];
    b.v[0][j] = k < rows ? x0 : 0.f;
    b.v[1][j] = k < rows ? x1 : 0.f;
  }
}

// A(i, k) = Ls[i][k] lifted out of an LDS panel
__device__ __forceinline__ void areg_from_lds(OpRegs& a, const float* __restrict__ Ls, int lane) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int j = 0; j < 32; j++) {
    a.v[0][j] = Ls[li * PITCH + 2 * j + lk];
    a.v[1][j] = Ls[(li + 32) * PITCH + 2 * j + lk];
  }
}

// acc += A(regs) * R   with R(k, j) = Rs[j][k]   (right operand transposed, from the panel)
__device__ __forceinline__ void mm_areg_bldsT(const OpRegs& a, const float* __restrict__ Rs,
                                              f32x16 (&acc)[2][2], int lane) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int j = 0; j < 32; j++) {
    // fence every 8 k-steps: the loop must be fully unrolled (static register indices), but the
    // scheduler must not lift all 64 panel reads above the first MFMA (64 more live registers)
    if ((j & 7) == 0) __builtin_amdgcn_sched_barrier(0);
    const int kk = 2 * j + lk;
    const float b0 = Rs[li * PITCH + kk], b1 = Rs[(li + 32) * PITCH + kk];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[0][j], b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[0][j], b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[1][j], b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[1][j], b1, acc[1][1], 0, 0, 0);
  }
}

// acc += L * B(regs)   with L(i,k) = Ls[i][k] (LT = false) or Ls[k][i] (LT = true) from the panel
template <bool LT>
__device__ __forceinline__ void mm_alds_breg(const float* __restrict__ Ls, const OpRegs& b,
                                             f32x16 (&acc)[2][2], int lane) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int j = 0; j < 32; j++) {
    if ((j & 7) == 0) __builtin_amdgcn_sched_barrier(0);
    const int kk = 2 * j + lk;
    float a0, a1;
    if (LT) {
      a0 = Ls[kk * PITCH + li];
      a1 = Ls[kk * PITCH + li + 32];
    } else {
      a0 = Ls[li * PITCH + kk];
      a1 = Ls[(li + 32) * PITCH + kk];
    }
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.v[0][j], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.v[1][j], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.v[0][j], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.v[1][j], acc[1][1], 0, 0, 0);
  }
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

// reductions across the 32 lanes that hold one accumulator row (same lane>>5)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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
        g[acc_row(a, r, lane) * ld + acc_col(b, lane)] = acc[a][b][r] * mul;
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
        if (row < rows) g[row * ld + acc_col(b, lane)] = acc[a][b][r];
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

__global__ __launch_bounds__(kThreads, 2) void attn_fwd_kernel(const AttnArgs g, float* __restrict__ P,
                                                               float* __restrict__ O) {
  __shared__ __attribute__((aligned(16))) float lds[kWaves * kPanel];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long unit = (long long)blockIdx.x * kWaves + wave;  // (sequence, head)
  if (unit >= (long long)g.n_seq * g.H) return;
  const int n = (int)(unit / g.H), h = (int)(unit % g.H);
  float* s0 = lds + wave * kPanel;
  OpRegs op;
  {
    Stage sq, sk;
    sq.load(g.q + ((size_t)n * T) * g.ldq + h * D, g.ldq, lane);
    sk.load(g.k + ((size_t)n * g.kv_rows) * g.ldk + h * D, g.ldk, lane, g.kv_rows);
    sq.store(s0, lane);
    areg_from_lds(op, s0, lane);          // Q as the left operand, in registers
    sk.store(s0, lane, g.kv_rows);        // the panel now holds K
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc[2][2];
  zero(acc);
  mm_areg_bldsT(op, s0, acc, lane);       // S = Q K^T
  __builtin_amdgcn_sched_barrier(0);
  // V goes straight from global into registers (row-major right operand); in flight under the softmax
  breg_load(op, g.v + ((size_t)n * g.kv_rows) * g.ldv + h * D, g.ldv, lane, g.kv_rows);
  // ---- scale, mask, softmax over keys (columns) ----------------------------------------
  const int mode = g.mask_mode, nv = g.n_valid, kvr = g.kv_rows;
  for_acc(acc, lane, [&](float x, int row, int col) {
    const bool dead = col >= kvr || (mode == 1 && col >= nv) || (mode == 2 && col > row);
    return dead ? -1e9f : x * g.scale;
  });
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float m = half_max(fmaxf(acc[a][0][r], acc[a][1][r]));
      float e0 = expf(acc[a][0][r] - m), e1 = expf(acc[a][1][r] - m);
      float inv = 1.f / half_sum(e0 + e1);
      acc[a][0][r] = e0 * inv;
      acc[a][1][r] = e1 * inv;
    }
  const size_t pbase = (size_t)unit * T * T;
  if (P) acc_to_global(acc, P + pbase, T, lane, 1.f);
  if (g.p > 0.f) {
    const float inv_keep = 1.f / (1.f - g.p);
    for_acc(acc, lane, [&](float x, int row, int col) {
      return x * drop_scale(g.seed, pbase + (size_t)row * T + col, g.p, inv_keep);
    });
  }
  acc_to_lds(acc, s0, lane);  // P_drop over the K panel (this wave's reads of it are done)
  zero(acc);
  mm_alds_breg<false>(s0, op, acc, lane);  // O = P V
  acc_to_global(acc, O + (size_t)unit * T * D, D, lane, 1.f);
}

struct AttnBwdArgs {
  AttnArgs f;
  const float *P, *dO;
  float *dq, *dk, *dv;
  int lddq, lddk, lddv;
};

__global__ __launch_bounds__(kThreads, 2) void attn_bwd_kernel(const AttnBwdArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[kWaves * kPanel];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long unit = (long long)blockIdx.x * kWaves + wave;
  if (unit >= (long long)g.f.n_seq * g.f.H) return;
  const int n = (int)(unit / g.f.H), h = (int)(unit % g.f.H);
  float* s0 = lds + wave * kPanel;
  const size_t pbase = (size_t)unit * T * T;
  const float p = g.f.p, inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const float* __restrict__ dO = g.dO + (size_t)unit * T * D;
  const float* __restrict__ Vg = g.f.v + ((size_t)n * g.f.kv_rows) * g.f.ldv + h * D;
  const float* __restrict__ Kg = g.f.k + ((size_t)n * g.f.kv_rows) * g.f.ldk + h * D;
  const float* __restrict__ Qg = g.f.q + ((size_t)n * T) * g.f.ldq + h * D;
  const float* __restrict__ Pu = g.P + pbase;
  OpRegs op;
  f32x16 acc[2][2];
  // ---- dV = dropout(P)^T dO :  panel <- dropout(P), dO as the register right operand -------------
  // P (pre-dropout) comes in accumulator layout straight from HBM (128-B row segments); it is read a
  // second time for dS (still in L2) rather than held in 64 registers across two products.
  breg_load(op, dO, D, lane);
  {
    f32x16 pd[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = acc_row(a, r, lane), col = acc_col(b, lane);
          // branch-free: at p = 0 the hash test u >= p always passes and inv_keep is 1
          pd[a][b][r] = Pu[row * T + col] * drop_scale(g.f.seed, pbase + (size_t)row * T + col, p, inv_keep);
        }
    acc_to_lds(pd, s0, lane);
  }
  __builtin_amdgcn_sched_barrier(0);   // phase fence: keeps later loads from being hoisted above
  zero(acc);
  mm_alds_breg<true>(s0, op, acc, lane);
  acc_to_global_rows(acc, g.dv + ((size_t)n * g.f.kv_rows) * g.lddv + h * D, g.lddv, lane, g.f.kv_rows);
  __builtin_amdgcn_sched_barrier(0);
  // ---- dPd = dO V^T :  dO through the panel into registers (left operand), then the panel holds V
  {
    Stage st;
    st.load(dO, D, lane);
    st.store(s0, lane);
    areg_from_lds(op, s0, lane);
    __builtin_amdgcn_sched_barrier(0);
    st.load(Vg, g.f.ldv, lane, g.f.kv_rows);
    st.store(s0, lane, g.f.kv_rows);
  }
  __builtin_amdgcn_sched_barrier(0);
  zero(acc);
  mm_areg_bldsT(op, s0, acc, lane);
  __builtin_amdgcn_sched_barrier(0);
  // ---- dS = P * (dP - rowsum(dP * P)) with dP = dPd * mask/(1-p);  then the 1/sqrt(dk) scale ------
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // four rows' shuffle chains at a time
      const int row = acc_row(a, r, lane);
      const float p0 = Pu[row * T + acc_col(0, lane)], p1 = Pu[row * T + acc_col(1, lane)];
      const float d0 = acc[a][0][r] * drop_scale(g.f.seed, pbase + (size_t)row * T + acc_col(0, lane), p, inv_keep);
      const float d1 = acc[a][1][r] * drop_scale(g.f.seed, pbase + (size_t)row * T + acc_col(1, lane), p, inv_keep);
      const float dot = half_sum(d0 * p0 + d1 * p1);
      acc[a][0][r] = p0 * (d0 - dot) * g.f.scale;
      acc[a][1][r] = p1 * (d1 - dot) * g.f.scale;
    }
  __builtin_amdgcn_sched_barrier(0);
  breg_load(op, Kg, g.f.ldk, lane, g.f.kv_rows);
  acc_to_lds(acc, s0, lane);  // dS (already scaled) over the V panel
  zero(acc);
  mm_alds_breg<false>(s0, op, acc, lane);  // dQ = dS K
  __builtin_amdgcn_sched_barrier(0);
  breg_load(op, Qg, g.f.ldq, lane);
  acc_to_global(acc, g.dq + ((size_t)n * T) * g.lddq + h * D, g.lddq, lane, 1.f);
  zero(acc);
  mm_alds_breg<true>(s0, op, acc, lane);   // dK = dS^T Q
  acc_to_global_rows(acc, g.dk + ((size_t)n * g.f.kv_rows) * g.lddk + h * D, g.lddk, lane, g.f.kv_rows);
}

constexpr size_t kLds = 0;  // panels are static LDS (66.5 KB per workgroup)

inline bool bad(int n_seq, int H, int Tt, int d, int mask_mode, int n_valid, float p) {
  return n_seq < 0 || H <= 0 || mask_mode < 0 || mask_mode > 2 || p < 0.f || p >= 1.f ||
         (mask_mode == 1 && (n_valid <= 0 || n_valid > Tt));
}

}  // namespace

AIT_API int ait_attn_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                         int n_seq, int H, int Tt, int d, int kv_rows, int mask_mode, int n_valid_keys,
                         float scale, float p_drop, unsigned long long seed, float* P, float* O,
                         void* stream) {
  if (bad(n_seq, H, Tt, d, mask_mode, n_valid_keys, p_drop)) return AIT_EINVAL;
  if (Tt != T || d != D) return AIT_EUNSUPPORTED;
  if (n_seq == 0) return AIT_OK;
  if (!q || !k || !v || !O) return AIT_EINVAL;
  if (kv_rows <= 0 || kv_rows > T) return AIT_EINVAL;
  AttnArgs a{q, k, v, ldq, ldk, ldv, n_seq, H, mask_mode, n_valid_keys, kv_rows, scale, p_drop, seed};
  const long long units = (long long)n_seq * H;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)((units + kWaves - 1) / kWaves)),
                     dim3(kThreads), kLds, ait_stream(stream), a, P, O);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_attn_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                         const float* P, const float* dO, int n_seq, int H, int Tt, int d, int kv_rows,
                         float scale, float p_drop, unsigned long long seed, float* dq, int lddq,
                         float* dk, int lddk, float* dv, int lddv, void* stream) {
  if (bad(n_seq, H, Tt, d, 0, 0, p_drop)) return AIT_EINVAL;
  if (Tt != T || d != D) return AIT_EUNSUPPORTED;
  if (n_seq == 0) return AIT_OK;
  if (!q || !k || !v || !P || !dO || !dq || !dk || !dv) return AIT_EINVAL;
  AttnBwdArgs b;
  if (kv_rows <= 0 || kv_rows > T) return AIT_EINVAL;
  b.f = AttnArgs{q, k, v, ldq, ldk, ldv, n_seq, H, 0, 0, kv_rows, scale, p_drop, seed};
  b.P = P; b.dO = dO; b.dq = dq; b.dk = dk; b.dv = dv;
  b.lddq = lddq; b.lddk = lddk; b.lddv = lddv;
  const long long units = (long long)n_seq * H;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)((units + kWaves - 1) / kWaves)),
                     dim3(kThreads), kLds, ait_stream(stream), b);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}
