// ait_amd/csrc/attn.hip -- ait_attn_fwd (design notes: attn_impl.h)
#include "attn_impl.h"

namespace {
using namespace ait_attn;

__global__ __launch_bounds__(kThreads, 2) void attn_fwd_kernel(const AttnArgs g, float* __restrict__ P,
                                                               float* __restrict__ O) {
  __shared__ __attribute__((aligned(16))) float lds[kWaves * kPanel];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: unit bases in SGPRs)
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
  if (P) acc_to_global(acc, P + pbase, T, lane, 1.f);
  if (g.p > 0.f) {
    const float inv_keep = 1.f / (1.f - g.p);
    const DropBlock db(g.seed, pbase);
    for_acc(acc, lane, [&](float x, int row, int col) { return x * db.scale(row * T + col, g.p, inv_keep); });
  }
  acc_to_lds(acc, s0, lane);  // P_drop over the K panel (this wave's reads of it are done)
  zero(acc);
  mm_alds_breg<false>(s0, op, acc, lane);  // O = P V
  acc_to_global(acc, O + (size_t)unit * T * D, D, lane, 1.f);
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

