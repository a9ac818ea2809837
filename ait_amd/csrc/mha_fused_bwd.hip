// ait_amd/csrc/mha_fused_bwd.hip -- ait_mha_core_bwd: the backward of MultiHeadAttention between the closing LayerNorm and
// the Q/K/V projections as ONE kernel, one workgroup per sequence with all eight heads resident -- the mirror of
// mha_fused.hip (lib/model/system/SubLayers.py:82-100 backwards: fc, SHBlock :22-39, ScaledDotProductAttention
// Modules.py:16-29):
//
//     fc:        du = df fc_w                                     [64 x 512] . [512 x 64]: wave h takes columns 64 h .. of
//                                                                 df, the partial products are summed through the panels
//     SHBlock:   dgate_h[c] = sum_t du[t,c] O_h[t,c] ; dg = softmax'_h(gate, dgate) ; ds = sk_w^T dg / 64
//                dO_h[t,c] = du[t,c] gate_h[c] + ds[c]             (never written: formed from the shared du panel where a
//                                                                 product needs it as an operand)
//     per head:  dV = dropout(P)^T dO ; dPd = dO V^T ; dS = P (dP - rowsum(dP P)) / 8 ; dQ = dS K ; dK = dS^T Q
//
// Replaces three launches (the K = 64 input-gradient product of fc at ~ 70 TFLOP/s, ait_sh_bwd, ait_attn_bwd) and the
// two tensors between them: du [M, 64] and dO [M, 512] are neither written nor read, O is read once.  What stays outside:
// the LayerNorm backward in front (ait_ln_bwd: df, the residual branch, d gamma / d beta) and the three products whose
// reduction runs over ALL sequences (d fc_w = df^T u, d sk_w = dg^T s, d W_qkv): they are M-deep GEMMs, not per-sequence
// work.  Wave = head, as in the forward.  Products in the library's f32 form (split_planes.h: three bf16 planes per value,
// six v_mfma_f32_32x32x16_bf16 per block) -- UNLIKE attn_bwd_kernel, whose four-wave workgroups measured faster on
// v_mfma_f32_32x32x2_f32: here every product moved to the split form took ~ 3 % off (1200 sequences: none 0.401 ms, the
// two last 0.391, all 0.368 against 0.486 for the three launches; profiles/r05_mha_fused_bwd.txt).
//
// LDS: eight 64 x 65 panels (133 KB) + du (16.6 KB) + 8 KB of vectors -> one workgroup (8 waves, 2 per SIMD) per CU.
#include "attn_impl.h"
#include "gemm_internal.h"
#include <type_traits>

namespace {
using namespace ait_attn;

constexpr int kHeads = 8;
constexpr int kDm = kHeads * D;
constexpr int kFusedThreads = kHeads * 64;
constexpr int kSplitMask = ait_lab::Knobs::fb_split;      // the attention tile's products in the split form: all four

struct CoreBwdArgs {
  AttnArgs at;
  const float *df, *fc_w, *O, *gate, *sk_w, *P;
  float *dq, *dk, *dv;          // OUT16: bf16 tensors behind these pointers (pitches in elements either way)
  int lddq, lddk, lddv;
  float* dg;
};

template <class P>
__device__ __forceinline__ P* each_time(P* p) {
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ void wg_barrier() {      // (mha_fused.hip: LDS traffic only needs lgkmcnt(0), not the fence's vmcnt(0))
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// the lane id as a value the optimiser cannot trace: what is computed from it (store offsets, dropout hashes) is computed
// where it is used instead of once for the whole kernel and kept -- or spilled -- across the phases in between
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ unsigned short bf16_bits(float x) {
  const __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}
// the accumulator tile to rows < `rows` of a tensor whose first element of this unit is `first` ELEMENTS behind `base`
// (OUT16: bf16 elements, nearest even).  PAIRED: the product's right operand came from breg_load_pairs -- the lane's two
// values of a row are neighbouring columns and leave as one store.
template <bool OUT16, bool PAIRED>
__device__ __forceinline__ void store_rows(const f32x16 (&acc)[2][2], float* base, size_t first, int ld, int lane, int rows) {
  float* __restrict__ g32 = base + first;
  unsigned short* __restrict__ g16 = reinterpret_cast<unsigned short*>(base) + first;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = acc_row(a, r, lane);
      if (row >= rows) continue;
      if constexpr (PAIRED) {
        const unsigned off = (unsigned)(row * ld + 2 * (lane & 31));
        if constexpr (OUT16)
          *reinterpret_cast<unsigned*>(g16 + off) = (unsigned)bf16_bits(acc[a][0][r]) | ((unsigned)bf16_bits(acc[a][1][r]) << 16);
        else
          *reinterpret_cast<float2*>(g32 + off) = make_float2(acc[a][0][r], acc[a][1][r]);
      } else {
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const unsigned off = (unsigned)(row * ld + acc_col(b, lane));
          if constexpr (OUT16) g16[off] = bf16_bits(acc[a][b][r]);
          else g32[off] = acc[a][b][r];
        }
      }
    }
}

// OUT16: dq / dk / dv are written as bf16;  IN16: q / k / v are bf16 tensors behind the float pointers
template <bool OUT16, bool IN16>
__global__ __launch_bounds__(kFusedThreads, 1) void mha_core_bwd_kernel(const CoreBwdArgs c) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const U = lds + kHeads * kPanel;          // [64][65]   du, shared by the eight heads
  float* const part = U + kPanel;                  // [8][64]    per-wave partials of ds
  float* const vdg = part + kHeads * 64;           // [512]      d gate (before the softmax backward)
  float* const wvec = vdg + kDm;                   // [8][128]   per head: gate_h[64], then ds[64]
  const AttnArgs& g = c.at;
  const int lane = threadIdx.x & 63, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lk = lane >> 5, tid = h * 64 + lane;
  float* s0 = lds + h * kPanel;
  const int n = blockIdx.x;
  const long long unit = (long long)n * kHeads + h;
  OpRegs op;
  f32x16 acc[2][2];
  // ---- fc backward: du = df fc_w, this wave's 64 of the 512 inner indices --------------------------------------------
  {
    Stage sd;
    sd.load(c.df + ((size_t)n * T) * kDm + h * 64, kDm, lane);
    breg_load(op, each_time(c.fc_w) + (size_t)(h * 64) * D, D, lane);      // R(k, j) = fc_w[64 h + k][j]
    sd.store(s0, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  zero(acc);
  mm_alds_breg<false, !ait_lab::Knobs::fb_du_f32>(s0, op, acc, lane);       // (split form: nothing on the vector pipe beside this product)
  __builtin_amdgcn_sched_barrier(0);
  // this head's O (64 x 64, whole rows) and its gates in flight across the head sum
  Stage so;
  so.load(c.O + (size_t)unit * T * D, D, lane);
  const float gme = c.gate[(size_t)n * kDm + tid];      // gate of (head h, channel `lane`)
  acc_to_lds(acc, s0, lane);
  wvec[h * 128 + lane] = gme;
  wg_barrier();
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = h * 8 + i;                   // 512 threads x 8 = 64 rows x 64 channels, heads in fixed order
    float v = 0.f;
#pragma unroll
    for (int hh = 0; hh < kHeads; hh++) v += lds[hh * kPanel + row * PITCH + lane];
    U[row * PITCH + lane] = v;
  }
  wg_barrier();
  // ---- SHBlock backward ----------------------------------------------------------------------------------------------
  {
    const int cc = (lane & 15) * 4, r0 = lane >> 4;      // (Stage's layout: four channels of rows r0 + 4 i)
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const float* u = U + (i * 4 + r0) * PITCH + cc;
      d0 += u[0] * so.v[i].x;
      d1 += u[1] * so.v[i].y;
      d2 += u[2] * so.v[i].z;
      d3 += u[3] * so.v[i].w;
    }
    d0 += __shfl_xor(d0, 16, 64); d1 += __shfl_xor(d1, 16, 64); d2 += __shfl_xor(d2, 16, 64); d3 += __shfl_xor(d3, 16, 64);
    d0 += __shfl_xor(d0, 32, 64); d1 += __shfl_xor(d1, 32, 64); d2 += __shfl_xor(d2, 32, 64); d3 += __shfl_xor(d3, 32, 64);
    if (r0 == 0) {
      float* o = vdg + h * 64 + cc;
      o[0] = d0; o[1] = d1; o[2] = d2; o[3] = d3;
    }
  }
  wg_barrier();
  {
    // softmax over the heads, backwards, for channel `lane`; then this wave's 64 rows of ds = sk_w^T dg
    float dot = 0.f;
#pragma unroll
    for (int hh = 0; hh < kHeads; hh++) dot += vdg[hh * 64 + lane] * wvec[hh * 128 + lane];
    const float dgv = gme * (vdg[tid] - dot);
    c.dg[(size_t)n * kDm + tid] = dgv;
    const float* w = each_time(c.sk_w) + (size_t)(h * 64) * D + lane;
    float dsp = 0.f;
#pragma unroll
    for (int q = 0; q < 64; q++)
      dsp += w[q * D] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dgv), q));
    part[tid] = dsp;
  }
  wg_barrier();
  {
    float dsl = 0.f;
#pragma unroll
    for (int hh = 0; hh < kHeads; hh++) dsl += part[hh * 64 + lane];
    wvec[h * 128 + 64 + lane] = dsl * (1.f / T);      // (s is a mean over the tokens)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the wave reads its own vector back: in order, no barrier)
  const float* const gv = wvec + h * 128;
  const float* const dsv = gv + 64;
  // ---- the attention tile of head h backwards (attn_bwd_kernel), dO_h(t, c) = du(t, c) gate_h(c) + ds(c) from U -----------
  const size_t pbase = (size_t)unit * T * T;
  const float p = g.p, inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const DropBlock db(g.seed, pbase);
  using In = typename std::conditional<IN16, unsigned short, float>::type;
  using StageIn = typename std::conditional<IN16, Stage16, Stage>::type;
  const In* __restrict__ Vg = reinterpret_cast<const In*>(g.v) + ((size_t)n * g.kv_rows) * g.ldv + h * D;
  const In* __restrict__ Kg = reinterpret_cast<const In*>(g.k) + ((size_t)n * g.kv_rows) * g.ldk + h * D;
  const In* __restrict__ Qg = reinterpret_cast<const In*>(g.q) + ((size_t)n * T) * g.ldq + h * D;
  const float* __restrict__ Pu = c.P + pbase;
  // dV = dropout(P)^T dO : dO as the register right operand, R(k, j) = dO(k, j)
  {
    const float g0 = gv[li], g1 = gv[32 + li], e0 = dsv[li], e1 = dsv[32 + li];
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float* u = U + (16 * kb + 8 * lk + j) * PITCH + li;
        op.v[0][kb][j] = u[0] * g0 + e0;
        op.v[1][kb][j] = u[32] * g1 + e1;
      }
  }
  {
    f32x16 pd[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = acc_row(a, r, lane), col = acc_col(b, lane);
          pd[a][b][r] = Pu[(unsigned)(row * T + col)] * db.scale(row * T + col, p, inv_keep);
        }
    acc_to_lds(pd, s0, lane);
  }
  __builtin_amdgcn_sched_barrier(0);
  zero(acc);
  mm_alds_breg<true, (kSplitMask & 1) != 0>(s0, op, acc, lane);
  store_rows<OUT16, false>(acc, c.dv, ((size_t)n * g.kv_rows) * c.lddv + h * D, c.lddv, opaque(lane), g.kv_rows);
  __builtin_amdgcn_sched_barrier(0);
  // dPd = dO V^T : dO as the register left operand, L(i, k) = dO(i, k); the panel holds V
  {
    StageIn st;
    st.load(Vg, g.ldv, lane, g.kv_rows);
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = 16 * kb + 8 * lk + j;
        const float gk = gv[k], ek = dsv[k];
        op.v[0][kb][j] = U[li * PITCH + k] * gk + ek;
        op.v[1][kb][j] = U[(32 + li) * PITCH + k] * gk + ek;
      }
    st.store(s0, lane, g.kv_rows);
  }
  __builtin_amdgcn_sched_barrier(0);
  zero(acc);
  mm_areg_bldsT<(kSplitMask & 2) != 0, 3, IN16 ? 1 : 3>(op, s0, acc, lane);
  __builtin_amdgcn_sched_barrier(0);
  // dS = P * (dP - rowsum(dP * P)) with dP = dPd * mask / (1 - p); then the 1 / sqrt(d_k) scale
  const int lane_s = opaque(lane);
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
      const int row = acc_row(a, r, lane_s);
      const float p0 = Pu[(unsigned)(row * T + acc_col(0, lane_s))], p1 = Pu[(unsigned)(row * T + acc_col(1, lane_s))];
      const float d0 = acc[a][0][r] * db.scale(row * T + acc_col(0, lane_s), p, inv_keep);
      const float d1 = acc[a][1][r] * db.scale(row * T + acc_col(1, lane_s), p, inv_keep);
      const float dot = half_sum(d0 * p0 + d1 * p1);
      acc[a][0][r] = p0 * (d0 - dot) * g.scale;
      acc[a][1][r] = p1 * (d1 - dot) * g.scale;
    }
  __builtin_amdgcn_sched_barrier(0);
  constexpr bool PAIRED = IN16 || kF32Pairs;
  if constexpr (PAIRED) breg_load_pairs(op, Kg, g.ldk, lane, g.kv_rows);      // (column-paired: so are dQ's columns)
  else breg_load(op, Kg, g.ldk, lane, g.kv_rows);
  acc_to_lds(acc, s0, lane);
  zero(acc);
  mm_alds_breg<false, (kSplitMask & 4) != 0, 3, IN16 ? 1 : 3>(s0, op, acc, lane);      // dQ = dS K
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (PAIRED) breg_load_pairs(op, Qg, g.ldq, lane);
  else breg_load(op, Qg, g.ldq, lane);
  store_rows<OUT16, PAIRED>(acc, c.dq, ((size_t)n * T) * c.lddq + h * D, c.lddq, opaque(lane), T);
  zero(acc);
  mm_alds_breg<true, (kSplitMask & 8) != 0, 3, IN16 ? 1 : 3>(s0, op, acc, lane);       // dK = dS^T Q
  store_rows<OUT16, PAIRED>(acc, c.dk, ((size_t)n * g.kv_rows) * c.lddk + h * D, c.lddk, opaque(lane), g.kv_rows);
}

constexpr size_t kBwdLds = (size_t)(kHeads * kPanel + kPanel + kHeads * 64 + kDm + kHeads * 128) * sizeof(float);
}  // namespace

// out_bf16 != 0: dq / dk / dv point at bf16 tensors; qkv_bf16 != 0: so do q / k / v (library-internal: csrc/transformer.hip's
// bf16-storage mode)
int ait_mha_core_bwd_ex(const float* df, const float* fc_w, const float* O, const float* gate, const float* sk_w, const void* q,
                        int ldq, const void* k, int ldk, const void* v, int ldv, const float* P, int n_seq, int kv_rows,
                        float scale, float p_attn, unsigned long long seed_attn, void* dq, int lddq, void* dk, int lddk, void* dv,
                        int lddv, float* dg, int out_bf16, int qkv_bf16, void* stream) {
  if (bad(n_seq, kHeads, T, D, 0, 0, p_attn)) return AIT_EINVAL;
  if (n_seq == 0) return AIT_OK;
  if (!df || !fc_w || !O || !gate || !sk_w || !q || !k || !v || !P || !dq || !dk || !dv || !dg) return AIT_EINVAL;
  if (kv_rows <= 0 || kv_rows > T) return AIT_EINVAL;
  if ((long long)n_seq * T * kDm > 0x7fffffffLL * 4) return AIT_EUNSUPPORTED;
  // the kernels' 16-byte (paired: 8-byte) loads and stores: rows of at least the eight heads' 512 columns, pitches in whole
  // vectors (4 f32 / 8 bf16), 16-byte aligned bases -- a strided view that breaks one of these is refused, not faulted on
  if (!ait_attn::rows_ok(q, ldq, qkv_bf16) || !ait_attn::rows_ok(k, ldk, qkv_bf16) || !ait_attn::rows_ok(v, ldv, qkv_bf16) ||
      !ait_attn::rows_ok(dq, lddq, out_bf16) || !ait_attn::rows_ok(dk, lddk, out_bf16) || !ait_attn::rows_ok(dv, lddv, out_bf16))
    return AIT_EUNSUPPORTED;
  for (const void* p : {(const void*)df, (const void*)fc_w, (const void*)O, (const void*)gate, (const void*)sk_w, (const void*)P,
                        (const void*)dg})
    if (reinterpret_cast<uintptr_t>(p) & 15) return AIT_EUNSUPPORTED;
  CoreBwdArgs c;
  c.at = AttnArgs{static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v), ldq, ldk, ldv, n_seq,
                  kHeads, 0, 0, kv_rows, scale, p_attn, seed_attn};
  c.df = df; c.fc_w = fc_w; c.O = O; c.gate = gate; c.sk_w = sk_w; c.P = P;
  c.dq = static_cast<float*>(dq); c.dk = static_cast<float*>(dk); c.dv = static_cast<float*>(dv);
  c.lddq = lddq; c.lddk = lddk; c.lddv = lddv; c.dg = dg;
  void (*fn)(const CoreBwdArgs) = out_bf16 ? (qkv_bf16 ? mha_core_bwd_kernel<true, true> : mha_core_bwd_kernel<true, false>)
                                           : (qkv_bf16 ? mha_core_bwd_kernel<false, true> : mha_core_bwd_kernel<false, false>);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds) != hipSuccess)
    return AIT_ELAUNCH;
  hipLaunchKernelGGL(fn, dim3((unsigned)n_seq), dim3(kFusedThreads), kBwdLds, ait_stream(stream), c);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_mha_core_bwd(const float* df, const float* fc_w, const float* O, const float* gate, const float* sk_w,
                             const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* P, int n_seq,
                             int kv_rows, float scale, float p_attn, unsigned long long seed_attn, float* dq, int lddq, float* dk,
                             int lddk, float* dv, int lddv, float* dg, void* stream) {
  return ait_mha_core_bwd_ex(df, fc_w, O, gate, sk_w, q, ldq, k, ldk, v, ldv, P, n_seq, kv_rows, scale, p_attn, seed_attn, dq, lddq,
                             dk, lddk, dv, lddv, dg, 0, 0, stream);
}
