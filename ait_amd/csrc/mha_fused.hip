// ait_amd/csrc/mha_fused.hip -- ait_mha_core_fwd: everything of MultiHeadAttention.forward behind the Q/K/V projections
// as ONE kernel, one workgroup per sequence with all eight heads resident (lib/model/system/SubLayers.py:82-100,
// Modules.py:16-29, SHBlock SubLayers.py:22-39):
//
//     per head:   S = (Q K^T) / 8 ; mask ; P = softmax(S) ; O_h = dropout(P) V                   (phase A)
//     SHBlock:    s = mean_t sum_h O_h ; gate = softmax_h(sk_w s + sk_b) ; u = sum_h gate_h * O_h    (phases B, C)
//     closing:    y = LayerNorm(dropout(u fc_w^T) + residual)                                     (phases D, E)
//
// Wave = head.  Phase A is the attention tile of attn.hip (attn_impl.h) with the wave's O_h left in its accumulator
// registers instead of going to memory.  The head sum runs through the waves' LDS panels (fixed order: bit-identical
// from launch to launch), u (64 x 64) becomes the left operand of the fc product, whose 64 x 512 result is again
// accumulators -- 64 output columns per wave, fc_w's rows straight from global (L2) in operand order -- and the
// LayerNorm's row statistics are reduced across lanes on the vector pipe (DPP) and across the eight waves through LDS.  Products in the library's f32 form (split_planes.h).
//
// What never touches HBM in inference: P, O, u, the fc output; in training they are written once (the backward
// kernels read them) and never re-read by the forward.  Replaces four launches (attention tiles, selective heads,
// the K = 64 fc product at a quarter of the product kernel's usual rate, LayerNorm rows).
//
// LDS: eight 64 x 65 panels (133 KB) + u (16.6 KB) + 9 KB of vectors -> one workgroup (8 waves, 2 per SIMD) per CU.
#include "attn_impl.h"
#include "gemm_internal.h"

namespace {
using namespace ait_attn;

constexpr int kHeads = 8;
constexpr int kDm = kHeads * D;            // 512: model width
constexpr int kFusedThreads = kHeads * 64;

struct CoreArgs {
  AttnArgs at;
  const float *sk_w, *sk_b, *fc_w, *residual, *ln_g, *ln_b;
  float eps, p_fc;
  unsigned long long seed_fc;
  int out_rows;
  int q_rep;       // sequence n takes its queries and its residual from sequence n / q_rep of q / residual
  float *P, *O, *u, *gate, *s, *f, *y, *mean, *rstd;
};

// raw barrier: __syncthreads() is fence + barrier, and the fence drains vmcnt(0) -- the P / O / u / f stores still in
// flight and the fc_w / residual loads issued ahead -- at every one of the nine phase boundaries.  LDS traffic only
// needs lgkmcnt(0).
typedef __attribute__((address_space(3))) void lds_void;
// pull the 128-B line at `src` towards this XCD's L2: a one-dword LDS-DMA transfer per lane into `sink` (wave-uniform
// LDS byte address)
__device__ __forceinline__ void prefetch_line(const float* src, unsigned sink) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src), "s"(sink) : "m0", "memory");
}
// a pointer the optimiser cannot see through: the weights are the same for every sequence of the persistent loop, and
// hoisted out of it their 128 registers (fc_w and sk_w rows) would live across the whole body and spill
template <class P>
__device__ __forceinline__ P* each_time(P* p) {
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// IN16: q / k / v are bf16 tensors behind the float pointers (pitches in elements either way)
template <bool IN16>
__global__ __launch_bounds__(kFusedThreads, 1) void mha_core_fwd_kernel(const CoreArgs c) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const U = lds + kHeads * kPanel;          // [64][65]
  float* const part = U + kPanel;                  // [8][64]  per-head column sums, later per-wave row partials
  float* const vec = part + kHeads * 64;           // [512]    gate logits, then gates
  float* const sv = vec + kDm;                     // [64]     s
  float* const rstat = sv + 64;                    // [2][64]  row mean / rstd
  const AttnArgs& g = c.at;
  constexpr bool PAIRED = IN16 || kF32Pairs;      // O_h's columns: 2 li + t (breg_load_pairs), else 32 t + li
  const int lane = threadIdx.x & 63, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform, in an SGPR)
  const int li = lane & 31, lk = lane >> 5, tid = h * 64 + lane;
  float* s0 = lds + h * kPanel;
  const unsigned sink = (unsigned)(size_t)(lds_void*)(rstat + 128);      // 256 B nobody reads: where prefetches land
  // One workgroup per sequence, NOT a persistent loop over sequences: inside a loop the optimiser hoists the ~400
  // lane-dependent address computations of the body out of it and keeps them live across it (1.9 KB of scratch per lane,
  // 2.4x the time; hiding the lane id and the induction variable from it still left 200 B and 1.3x).
  const int n = blockIdx.x;
  const int nq = c.q_rep > 1 ? n / c.q_rep : n;      // (inference: the decoder's query side is one sequence per PAIR)
  const long long unit = (long long)n * kHeads + h;
  // ---- phase A: the attention tile of head h (attn.hip) -----------------------------------------------------
  OpRegs op;
  if constexpr (IN16) {
    Stage16 sq, sk;
    sq.load(reinterpret_cast<const unsigned short*>(g.q) + ((size_t)nq * T) * g.ldq + h * D, g.ldq, lane);
    sk.load(reinterpret_cast<const unsigned short*>(g.k) + ((size_t)n * g.kv_rows) * g.ldk + h * D, g.ldk, lane, g.kv_rows);
    sq.store(s0, lane);
    areg_from_lds(op, s0, lane);
    sk.store(s0, lane, g.kv_rows);
  } else {
    Stage sq, sk;
    sq.load(g.q + ((size_t)nq * T) * g.ldq + h * D, g.ldq, lane);
    sk.load(g.k + ((size_t)n * g.kv_rows) * g.ldk + h * D, g.ldk, lane, g.kv_rows);
    sq.store(s0, lane);
    areg_from_lds(op, s0, lane);
    sk.store(s0, lane, g.kv_rows);
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc[2][2];
  zero(acc);
  mm_areg_bldsT<true, IN16 ? 1 : 3, IN16 ? 1 : 3>(op, s0, acc, lane);       // S = Q K^T
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (IN16)      // (column-paired: O_h's columns below are acc_col_of<IN16>)
    breg_load_pairs(op, reinterpret_cast<const unsigned short*>(g.v) + ((size_t)n * g.kv_rows) * g.ldv + h * D, g.ldv, lane, g.kv_rows);
  else if constexpr (PAIRED)
    breg_load_pairs(op, g.v + ((size_t)n * g.kv_rows) * g.ldv + h * D, g.ldv, lane, g.kv_rows);
  else
    breg_load(op, g.v + ((size_t)n * g.kv_rows) * g.ldv + h * D, g.ldv, lane, g.kv_rows);
  scale_mask(acc, lane, g);
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float m = half_max(fmaxf(acc[a][0][r], acc[a][1][r]));
      float e0 = exp_neg(acc[a][0][r] - m), e1 = exp_neg(acc[a][1][r] - m);
      float inv = __builtin_amdgcn_rcpf(half_sum(e0 + e1));      // (1 ulp; a full-precision divide is ten instructions)
      acc[a][0][r] = e0 * inv;
      acc[a][1][r] = e1 * inv;
    }
  const size_t pbase = (size_t)unit * T * T;
  if (c.P) acc_to_global(acc, c.P + pbase, T, lane, 1.f);
  if (g.p > 0.f) {
    const float inv_keep = 1.f / (1.f - g.p);
    const DropBlock db(g.seed, pbase);
    for_acc(acc, lane, [&](float x, int row, int col) { return x * db.scale(row * T + col, g.p, inv_keep); });
  }
  acc_to_lds(acc, s0, lane);
  zero(acc);
  mm_alds_breg<false, true, 3, IN16 ? 1 : 3>(s0, op, acc, lane);  // O_h = P V, kept in acc
  __builtin_amdgcn_sched_barrier(0);
  if (c.O) {
    float* __restrict__ og = c.O + (size_t)unit * T * D;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = acc_row(a, r, lane);
        if constexpr (PAIRED) {      // columns 2 li and 2 li + 1: one 8-byte store
          *reinterpret_cast<float2*>(og + (unsigned)(row * D + 2 * li)) = make_float2(acc[a][0][r], acc[a][1][r]);
        } else {
          og[(unsigned)(row * D + li)] = acc[a][0][r];
          og[(unsigned)(row * D + 32 + li)] = acc[a][1][r];
        }
      }
  }
  // ---- phase B: s = mean over tokens of the head sum; gate = softmax over heads of sk_w s + sk_b ----------------
  // rows 64 h .. 64 h + 63 of sk_w (this wave's 64 gate logits) go through the wave's own panel, whose P the product
  // above has read: coalesced loads, conflict-free row reads, and nothing here waits for another wave
  {
    Stage sw;
    sw.load(each_time(c.sk_w) + (size_t)(h * 64) * D, D, lane);
#pragma unroll
    for (int b = 0; b < 2; b++) {
      float cs = 0.f;
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) cs += acc[a][b][r];
      cs += __shfl_xor(cs, 32, 64);
      if (lk == 0) part[h * 64 + acc_col_of<PAIRED>(b, lane)] = cs;
    }
    sw.store(s0, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  // fc_w rows 64 h .. 64 h + 63 (this wave's output columns) as the right operand of phase D in flight until then:
  // R(k, j) = fc_w[64 h + j][k]; lane (li, lk) holds k = 16 kb + 8 lk + 0..7 of column 32 t + li
  {
    const float* w = each_time(c.fc_w) + (size_t)(h * 64) * D;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int kb = 0; kb < 4; kb++) {
        const float4* p = reinterpret_cast<const float4*>(w + (t * 32 + li) * D + 16 * kb + 8 * lk);
        const float4 lo = p[0], hi = p[1];
        op.v[t][kb][0] = lo.x; op.v[t][kb][1] = lo.y; op.v[t][kb][2] = lo.z; op.v[t][kb][3] = lo.w;
        op.v[t][kb][4] = hi.x; op.v[t][kb][5] = hi.y; op.v[t][kb][6] = hi.z; op.v[t][kb][7] = hi.w;
      }
  }
  wg_barrier();
  float gj;                                     // gate of (head h, channel `lane`)
  {
    float sc = 0.f;                             // s[lane], in every wave
#pragma unroll
    for (int hh = 0; hh < kHeads; hh++) sc += part[hh * 64 + lane];
    sc *= 1.f / T;
    if (h == 0 && c.s) c.s[(size_t)n * 64 + lane] = sc;
    float d = each_time(c.sk_b)[tid];
    const float* wr = s0 + lane * PITCH;
#pragma unroll
    for (int q = 0; q < D; q++)
      d += wr[q] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), q));
    vec[tid] = d;
    wg_barrier();
    float mx = vec[lane];
#pragma unroll
    for (int hh = 1; hh < kHeads; hh++) mx = fmaxf(mx, vec[hh * 64 + lane]);
    float sum = 0.f;
#pragma unroll
    for (int hh = 0; hh < kHeads; hh++) sum += exp_neg(vec[hh * 64 + lane] - mx);
    gj = exp_neg(d - mx) / sum;
    if (c.gate) c.gate[(size_t)n * kDm + tid] = gj;
  }
  // ---- phase C: u = sum_h gate_h * O_h through the panels, heads in fixed order ---------------------------------
  {
    float g0, g1;                                           // gates of the lane's two channels
    if constexpr (PAIRED) {
      g0 = __shfl(gj, 2 * li, 64);
      g1 = __shfl(gj, 2 * li + 1, 64);
    } else {
      const float gx = __shfl_xor(gj, 32, 64);
      g0 = lk ? gx : gj;                                    // channels li and 32 + li
      g1 = lk ? gj : gx;
    }
    const int c0 = acc_col_of<PAIRED>(0, lane), c1 = acc_col_of<PAIRED>(1, lane);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = acc_row(a, r, lane);
        s0[row * PITCH + c0] = acc[a][0][r] * g0;
        s0[row * PITCH + c1] = acc[a][1][r] * g1;
      }
  }
  wg_barrier();
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = h * 8 + i;                   // 512 threads x 8 = 64 rows x 64 channels
    float v = 0.f;
#pragma unroll
    for (int hh = 0; hh < kHeads; hh++) v += lds[hh * kPanel + row * PITCH + lane];
    U[row * PITCH + lane] = v;
    if (c.u) c.u[((size_t)n * T + row) * D + lane] = v;
  }
  wg_barrier();
  // ---- phase D: f = u fc_w^T, this wave's 64 columns ---------------------------------------------------------
  zero(acc);
  mm_alds_breg<false>(U, op, acc, lane);
  __builtin_amdgcn_sched_barrier(0);
  const size_t row0 = (size_t)n * T;
  const int col0 = h * 64;
  if (c.f) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          c.f[(row0 + acc_row(a, r, lane)) * kDm + col0 + acc_col(b, lane)] = acc[a][b][r];
  }
  // ---- phase E: z = dropout(f) + residual ; LayerNorm over the 512 columns of every token row ------------------
  {
    const float inv_keep = c.p_fc > 0.f ? 1.f / (1.f - c.p_fc) : 1.f;
    const DropBlock db(c.seed_fc, row0 * kDm);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const size_t idx = (row0 + acc_row(a, r, lane)) * kDm + col0 + acc_col(b, lane);
          float z = acc[a][b][r];
          if (c.p_fc > 0.f) z *= db.scale((unsigned)(acc_row(a, r, lane) * kDm + col0 + acc_col(b, lane)), c.p_fc, inv_keep);
          acc[a][b][r] = z + c.residual[idx - (size_t)(n - nq) * T * kDm];
        }
  }
  float rv[32];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) rv[a * 16 + r] = acc[a][0][r] + acc[a][1][r];
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const float tot = half_sum(rv[i]);
    if (li == 0) part[h * 64 + acc_row(i >> 4, i & 15, lane)] = tot;
  }
  wg_barrier();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kHeads; w++) v += part[w * 64 + tid];
    rstat[tid] = v * (1.f / kDm);
  }
  wg_barrier();
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float mu = rstat[acc_row(a, r, lane)];
      const float d0 = acc[a][0][r] - mu, d1 = acc[a][1][r] - mu;
      acc[a][0][r] = d0;
      acc[a][1][r] = d1;
      rv[a * 16 + r] = d0 * d0 + d1 * d1;
    }
  wg_barrier();                                 // (the mean partials have been read)
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const float tot = half_sum(rv[i]);
    if (li == 0) part[h * 64 + acc_row(i >> 4, i & 15, lane)] = tot;
  }
  wg_barrier();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kHeads; w++) v += part[w * 64 + tid];
    const float rs = 1.f / sqrtf(v * (1.f / kDm) + c.eps);      // biased variance, like nn.LayerNorm
    rstat[64 + tid] = rs;
    if (c.mean) c.mean[row0 + tid] = rstat[tid];
    if (c.rstd) c.rstd[row0 + tid] = rs;
  }
  wg_barrier();
  {
    const float* lg = each_time(c.ln_g);
    const float* lb = each_time(c.ln_b);
    const float g0 = lg[col0 + li], g1 = lg[col0 + 32 + li];
    const float b0 = lb[col0 + li], b1 = lb[col0 + 32 + li];
    const int orows = c.out_rows;
    float* __restrict__ y = c.y + (size_t)n * orows * kDm + col0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = acc_row(a, r, lane);
        const float rs = rstat[64 + row];
        if (row < orows) {
          y[row * kDm + li] = acc[a][0][r] * rs * g0 + b0;
          y[row * kDm + 32 + li] = acc[a][1][r] * rs * g1 + b1;
        }
      }
  }
}

constexpr size_t kFusedLds = (size_t)(kHeads * kPanel + kPanel + kHeads * 64 + kDm + 64 + 128 + 64) * sizeof(float);

}  // namespace

// qkv_bf16 != 0: q / k / v point at bf16 tensors (library-internal: csrc/transformer.hip's bf16-storage mode)
int ait_mha_core_fwd_ex(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int n_seq, int kv_rows,
                        int mask_mode, int n_valid_keys, float scale, float p_attn, unsigned long long seed_attn, const float* sk_w,
                        const float* sk_b, const float* fc_w, const float* residual, const float* ln_g, const float* ln_b, float eps,
                        float p_fc, unsigned long long seed_fc, int out_rows, int q_rep, float* P, float* O, float* u, float* gate,
                        float* s, float* f, float* y, float* mean, float* rstd, int qkv_bf16, void* stream) {
  if (bad(n_seq, kHeads, T, D, mask_mode, n_valid_keys, p_attn) || p_fc < 0.f || p_fc >= 1.f) return AIT_EINVAL;
  if (n_seq == 0) return AIT_OK;
  if (!q || !k || !v || !sk_w || !sk_b || !fc_w || !residual || !ln_g || !ln_b || !y) return AIT_EINVAL;
  if (kv_rows <= 0 || kv_rows > T || out_rows <= 0 || out_rows > T || q_rep < 1) return AIT_EINVAL;
  if ((long long)n_seq * T * kDm > 0x7fffffffLL * 4) return AIT_EUNSUPPORTED;
  if (!ait_attn::rows_ok(q, ldq, qkv_bf16) || !ait_attn::rows_ok(k, ldk, qkv_bf16) || !ait_attn::rows_ok(v, ldv, qkv_bf16))
    return AIT_EUNSUPPORTED;      // (512 columns per row, pitches in whole 16-byte vectors, aligned bases)
  const void* fn = qkv_bf16 ? reinterpret_cast<const void*>(mha_core_fwd_kernel<true>)
                            : reinterpret_cast<const void*>(mha_core_fwd_kernel<false>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds) != hipSuccess) return AIT_ELAUNCH;
  CoreArgs c{AttnArgs{static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v), ldq, ldk, ldv, n_seq,
                      kHeads, mask_mode, n_valid_keys, kv_rows, scale, p_attn, seed_attn},
             sk_w, sk_b, fc_w, residual, ln_g, ln_b, eps, p_fc, seed_fc, out_rows, q_rep, P, O, u, gate, s, f, y, mean, rstd};
  if (qkv_bf16)
    hipLaunchKernelGGL(mha_core_fwd_kernel<true>, dim3((unsigned)n_seq), dim3(kFusedThreads), kFusedLds, ait_stream(stream), c);
  else
    hipLaunchKernelGGL(mha_core_fwd_kernel<false>, dim3((unsigned)n_seq), dim3(kFusedThreads), kFusedLds, ait_stream(stream), c);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_mha_core_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int n_seq,
                             int kv_rows, int mask_mode, int n_valid_keys, float scale, float p_attn,
                             unsigned long long seed_attn, const float* sk_w, const float* sk_b, const float* fc_w,
                             const float* residual, const float* ln_g, const float* ln_b, float eps, float p_fc,
                             unsigned long long seed_fc, int out_rows, int q_rep, float* P, float* O, float* u, float* gate,
                             float* s, float* f, float* y, float* mean, float* rstd, void* stream) {
  return ait_mha_core_fwd_ex(q, ldq, k, ldk, v, ldv, n_seq, kv_rows, mask_mode, n_valid_keys, scale, p_attn, seed_attn, sk_w, sk_b,
                             fc_w, residual, ln_g, ln_b, eps, p_fc, seed_fc, out_rows, q_rep, P, O, u, gate, s, f, y, mean, rstd, 0,
                             stream);
}
