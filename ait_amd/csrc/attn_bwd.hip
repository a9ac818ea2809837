// ait_amd/csrc/attn_bwd.hip -- ait_attn_bwd (design notes: attn_impl.h)
#include "attn_impl.h"

namespace {
using namespace ait_attn;

struct AttnBwdArgs {
  AttnArgs f;
  const float *P, *dO;
  float *dq, *dk, *dv;          // OUT16: bf16 tensors behind these pointers (pitches in elements either way)
  int lddq, lddk, lddv;
};

// the accumulator tile as bf16 (nearest even): the gradients' only consumers in the bf16-storage mode are the two bf16
// products of the projection's backward (transformer.hip)
__device__ __forceinline__ unsigned short to_bf16_bits(float x) {
  const __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void acc_to_global_rows16(const f32x16 (&acc)[2][2], unsigned short* __restrict__ g, int ld, int lane,
                                                     int rows) {
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = acc_row(a, r, lane);
        if (row < rows) g[(unsigned)(row * ld + acc_col(b, lane))] = to_bf16_bits(acc[a][b][r]);
      }
}

// The backward keeps v_mfma_f32_32x32x2_f32 (mm_* with SPLIT = false) and reads the saved P.  Measured alternatives
// (profiles/r04_attention.txt): its vector pipe is already the busier one -- the dropout hash twice over 4096
// elements, the dS algebra -- so the split form's 176 vector instructions per k-block made it 0.59 ms per block
// against 0.34; recomputing P from saved row statistics (no 157-MB P tensor) with only the score product split:
// 0.48 ms (an extra product and an exp pass in front of everything else).
template <bool OUT16>
__global__ __launch_bounds__(kThreads, 2) void attn_bwd_kernel(const AttnBwdArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[kWaves * kPanel];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: unit bases in SGPRs)
  const long long unit = (long long)blockIdx.x * kWaves + wave;
  if (unit >= (long long)g.f.n_seq * g.f.H) return;
  const int n = (int)(unit / g.f.H), h = (int)(unit % g.f.H);
  float* s0 = lds + wave * kPanel;
  const size_t pbase = (size_t)unit * T * T;
  const float p = g.f.p, inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const DropBlock db(g.f.seed, pbase);
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
          pd[a][b][r] = Pu[(unsigned)(row * T + col)] * db.scale(row * T + col, p, inv_keep);
        }
    acc_to_lds(pd, s0, lane);
  }
  __builtin_amdgcn_sched_barrier(0);   // phase fence: keeps later loads from being hoisted above
  zero(acc);
  mm_alds_breg<true, false>(s0, op, acc, lane);
  if constexpr (OUT16)
    acc_to_global_rows16(acc, reinterpret_cast<unsigned short*>(g.dv) + ((size_t)n * g.f.kv_rows) * g.lddv + h * D, g.lddv, lane,
                         g.f.kv_rows);
  else
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
  mm_areg_bldsT<false>(op, s0, acc, lane);
  __builtin_amdgcn_sched_barrier(0);
  // ---- dS = P * (dP - rowsum(dP * P)) with dP = dPd * mask/(1-p);  then the 1/sqrt(dk) scale ------
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // four rows' shuffle chains at a time
      const int row = acc_row(a, r, lane);
      const float p0 = Pu[(unsigned)(row * T + acc_col(0, lane))], p1 = Pu[(unsigned)(row * T + acc_col(1, lane))];
      const float d0 = acc[a][0][r] * db.scale(row * T + acc_col(0, lane), p, inv_keep);
      const float d1 = acc[a][1][r] * db.scale(row * T + acc_col(1, lane), p, inv_keep);
      const float dot = half_sum(d0 * p0 + d1 * p1);
      acc[a][0][r] = p0 * (d0 - dot) * g.f.scale;
      acc[a][1][r] = p1 * (d1 - dot) * g.f.scale;
    }
  __builtin_amdgcn_sched_barrier(0);
  breg_load(op, Kg, g.f.ldk, lane, g.f.kv_rows);
  acc_to_lds(acc, s0, lane);  // dS (already scaled) over the V panel
  zero(acc);
  mm_alds_breg<false, false>(s0, op, acc, lane);  // dQ = dS K
  __builtin_amdgcn_sched_barrier(0);
  breg_load(op, Qg, g.f.ldq, lane);
  if constexpr (OUT16)
    acc_to_global_rows16(acc, reinterpret_cast<unsigned short*>(g.dq) + ((size_t)n * T) * g.lddq + h * D, g.lddq, lane, T);
  else
    acc_to_global(acc, g.dq + ((size_t)n * T) * g.lddq + h * D, g.lddq, lane, 1.f);
  zero(acc);
  mm_alds_breg<true, false>(s0, op, acc, lane);   // dK = dS^T Q
  if constexpr (OUT16)
    acc_to_global_rows16(acc, reinterpret_cast<unsigned short*>(g.dk) + ((size_t)n * g.f.kv_rows) * g.lddk + h * D, g.lddk, lane,
                         g.f.kv_rows);
  else
    acc_to_global_rows(acc, g.dk + ((size_t)n * g.f.kv_rows) * g.lddk + h * D, g.lddk, lane, g.f.kv_rows);
}

}  // namespace

// out_bf16 != 0: dq / dk / dv point at bf16 tensors (library-internal: csrc/transformer.hip's bf16-storage backward)
int ait_attn_bwd_ex(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* P, const float* dO,
                    int n_seq, int H, int Tt, int d, int kv_rows, float scale, float p_drop, unsigned long long seed, void* dq,
                    int lddq, void* dk, int lddk, void* dv, int lddv, int out_bf16, void* stream) {
  if (bad(n_seq, H, Tt, d, 0, 0, p_drop)) return AIT_EINVAL;
  if (Tt != T || d != D) return AIT_EUNSUPPORTED;
  if (n_seq == 0) return AIT_OK;
  if (!q || !k || !v || !P || !dO || !dq || !dk || !dv) return AIT_EINVAL;
  AttnBwdArgs b;
  if (kv_rows <= 0 || kv_rows > T) return AIT_EINVAL;
  b.f = AttnArgs{q, k, v, ldq, ldk, ldv, n_seq, H, 0, 0, kv_rows, scale, p_drop, seed};
  b.P = P; b.dO = dO; b.dq = static_cast<float*>(dq); b.dk = static_cast<float*>(dk); b.dv = static_cast<float*>(dv);
  b.lddq = lddq; b.lddk = lddk; b.lddv = lddv;
  const long long units = (long long)n_seq * H;
  const dim3 grid((unsigned)((units + kWaves - 1) / kWaves));
  if (out_bf16) hipLaunchKernelGGL(attn_bwd_kernel<true>, grid, dim3(kThreads), kLds, ait_stream(stream), b);
  else hipLaunchKernelGGL(attn_bwd_kernel<false>, grid, dim3(kThreads), kLds, ait_stream(stream), b);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_attn_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                         const float* P, const float* dO, int n_seq, int H, int Tt, int d, int kv_rows,
                         float scale, float p_drop, unsigned long long seed, float* dq, int lddq,
                         float* dk, int lddk, float* dv, int lddv, void* stream) {
  return ait_attn_bwd_ex(q, ldq, k, ldk, v, ldv, P, dO, n_seq, H, Tt, d, kv_rows, scale, p_drop, seed, dq, lddq, dk, lddk, dv,
                         lddv, 0, stream);
}
