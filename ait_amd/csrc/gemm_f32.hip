// ait_amd/csrc/gemm_f32.hip -- fp32-in / fp32-accumulate GEMM on the gfx950 matrix cores.
//
// Every dense contraction of the AIT path (1x1-conv embeddings, QKV / output projections, the
// position-wise feed-forward, and all of their backward products) runs through this kernel:
//
//     C[M,N] (op)= alpha * opA(A) . opB(B)  [+ bias] [+ residual] [relu]
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulate, no reduced
// precision anywhere (north_star: logits within 1e-4 of the fp32 reference).  Dense peak on
// MI355X: 157.3 TFLOP/s (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz).
//
// Tiling (wave64, not a warp tiling): 128x128 output tile per 256-thread workgroup, 4 waves in
// 2x2, each wave owns 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).  K is consumed in slabs of
// 16: both operand slabs are staged in LDS K-MAJOR ([k][m] / [k][n]), which makes the MFMA
// operand fetch (lane l needs A[m0 + (l&31)][k + (l>>5)]) a conflict-free ds_read_b32 for every
// storage layout of A and B; the layout only changes how the slab is written:
//   operand stored with the reduction dim contiguous ([m][k]): float4 global loads along k,
//       transposed on the way into LDS (4 x ds_write_b32, row pitch 132 -> at most 2-way);
//   operand stored with the reduction dim outermost ([k][m]):  float4 loads along m,
//       ds_write_b128 straight into the K-major slab.
// The next slab's global loads are issued before the 32 MFMAs of the current slab and written
// to the other LDS buffer after them (register-staged double buffering, one barrier per slab);
// ~100 VGPRs and 33 KiB LDS per workgroup leave 4 workgroups per CU to cover barrier bubbles.
// One MFMA takes 64 cycles on its SIMD and needs one ds_read_b32 per operand per 2 MFMAs, so
// the LDS pipe is <15 % busy: the kernel is bound by the matrix pipe, which is the roofline it
// is measured against (bench.py "roofline").
//
// XCD-aware launch: workgroups are dealt round-robin to the 8 XCDs; block ids are remapped so
// that the N-tiles of one M-panel run back to back on ONE XCD (its A panel stays in that
// XCD's L2) while the (small, shared) B operand lives in every L2 / the Infinity Cache.
//
// Split-K (gridDim.z > 1) serves the weight-gradient products (reduction over bp*64 tokens,
// tiny output): partial tiles are combined with one fp32 atomic per element, issued as whole
// 128-B row segments straight from the accumulator layout.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_PITCH = BM + 4;  // floats; keeps ds_write_b128 16-B aligned
constexpr int kThreads = 256;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;      // [N] (or [M] with AIT_GEMM_BIAS_ROW)
  const float* residual;  // same addressing as C
  int M, N, K;
  int lda, ldb, ldc;
  int c_colblk;             // 0: plain row-major C.  >0: C(i,j) at (j/colblk)*c_batch + i*ldc + j%colblk
  long long c_batch;
  float alpha;
  int flags;
  int k_per_split;
  int splits;
};

// Stage one BK x 128 slab of an operand into registers.
//   KCONTIG = true : element (r, k) at p[r*ld + k]   (reduction dim contiguous)
//   KCONTIG = false: element (r, k) at p[k*ld + r]
// Rows >= R and k >= Kend read as zero.  ld % 4 == 0 and 16-B aligned bases are required.
template <bool KCONTIG>
__device__ __forceinline__ void load_slab(const float* __restrict__ p, int ld, int r0, int R,
                                          int k0, int Kend, float4 (&v)[2]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int e = t + i * kThreads;  // 512 float4 per slab
    if (KCONTIG) {
      const int r = r0 + (e >> 2), k = k0 + ((e & 3) << 2);
      if (r < R && k < Kend)
        v[i] = *reinterpret_cast<const float4*>(p + (size_t)r * ld + k);
      else
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const int k = k0 + (e >> 5), r = r0 + ((e & 31) << 2);
      if (k < Kend && r + 3 < R) {
        v[i] = *reinterpret_cast<const float4*>(p + (size_t)k * ld + r);
      } else if (k < Kend && r < R) {  // ragged right edge
        const float* q = p + (size_t)k * ld + r;
        v[i].x = q[0];
        v[i].y = (r + 1 < R) ? q[1] : 0.f;
        v[i].z = (r + 2 < R) ? q[2] : 0.f;
        v[i].w = 0.f;
      } else {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_slab(float* __restrict__ s, const float4 (&v)[2]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int e = t + i * kThreads;
    if (KCONTIG) {
      const int r = e >> 2, k = (e & 3) << 2;
      s[(k + 0) * LDS_PITCH + r] = v[i].x;
      s[(k + 1) * LDS_PITCH + r] = v[i].y;
      s[(k + 2) * LDS_PITCH + r] = v[i].z;
      s[(k + 3) * LDS_PITCH + r] = v[i].w;
    } else {
      const int k = e >> 5, r = (e & 31) << 2;
      *reinterpret_cast<float4*>(s + k * LDS_PITCH + r) = v[i];
    }
  }
}

// AK / BK_: true when that operand is stored with the reduction dimension contiguous.
//   forward  y = x W^T   : A = x [M,K] (AK), B = W [N,K] (BK_)
//   dgrad    dx = dy W   : A = dy [M,K'] (AK), B = W [K',N] (!BK_)
//   wgrad    dW = dy^T x : A = dy [K',M] (!AK), B = x [K',N] (!BK_)
template <bool AK, bool BK_>
__global__ __launch_bounds__(kThreads, 2) void gemm_f32_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                           // [2][BK][LDS_PITCH]
  float* Bs = lds + 2 * BK * LDS_PITCH;      // [2][BK][LDS_PITCH]

  // ---- XCD-aware work assignment (blocks b and b+8 share an XCD / L2) ----------------------
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int bid = blockIdx.x;
  const int xcd = bid % AIT_NXCD, j = bid / AIT_NXCD;
  int tm, tn, split;
  if (g.splits == 1) {
    // the N-tiles of one M-panel run back to back on ONE XCD: its A panel stays in that L2
    tm = (j / tiles_n) * AIT_NXCD + xcd;
    tn = j % tiles_n;
    split = 0;
  } else {
    // split-K (weight gradients): every XCD owns splits/8 K-ranges and runs ALL output tiles of
    // them concurrently, so each byte of A and B crosses the fabric once and the 16-row slabs
    // that the co-running tiles walk in step are served from that XCD's L2
    const int tiles = tiles_m * tiles_n;
    const int per_xcd = (g.splits + AIT_NXCD - 1) / AIT_NXCD;
    split = xcd * per_xcd + j / tiles;
    const int t = j % tiles;
    tm = t / tiles_n;
    tn = t % tiles_n;
    if (split >= g.splits) return;
  }
  if (tm >= tiles_m) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  if (kbeg >= kend) return;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  float4 ra[2], rb[2];
  load_slab<AK>(g.A, g.lda, m0, g.M, kbeg, kend, ra);
  load_slab<BK_>(g.B, g.ldb, n0, g.N, kbeg, kend, rb);
  store_slab<AK>(As, ra);
  store_slab<BK_>(Bs, rb);
  __syncthreads();

  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      load_slab<AK>(g.A, g.lda, m0, g.M, k0 + BK, kend, ra);
      load_slab<BK_>(g.B, g.ldb, n0, g.N, k0 + BK, kend, rb);
    }
    const float* as = As + cur * BK * LDS_PITCH + wm + li;
    const float* bs = Bs + cur * BK * LDS_PITCH + wn + li;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = as[(kk + lk) * LDS_PITCH], a1 = as[(kk + lk) * LDS_PITCH + 32];
      const float b0 = bs[(kk + lk) * LDS_PITCH], b1 = bs[(kk + lk) * LDS_PITCH + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      store_slab<AK>(As + (cur ^ 1) * BK * LDS_PITCH, ra);
      store_slab<BK_>(Bs + (cur ^ 1) * BK * LDS_PITCH, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool atomic = (g.flags & AIT_GEMM_ATOMIC) != 0;
  const bool accum = (g.flags & AIT_GEMM_ACCUMULATE) != 0;
  const bool relu = (g.flags & AIT_GEMM_RELU) != 0;
  const bool bias_row = (g.flags & AIT_GEMM_BIAS_ROW) != 0;
  const bool mask_pos = (g.flags & AIT_GEMM_MASK_POS) != 0;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int col = n0 + wn + b * 32 + li;
      if (col >= g.N) continue;
      size_t cbase;
      if (g.c_colblk > 0)
        cbase = (size_t)(col / g.c_colblk) * g.c_batch + (col % g.c_colblk);
      else
        cbase = col;
      const float bcol = (g.bias && !bias_row) ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= g.M) continue;
        const size_t off = cbase + (size_t)row * g.ldc;
        float v = g.alpha * acc[a][b][r];
        if (atomic) {
          unsafeAtomicAdd(g.C + off, v);
          continue;
        }
        v += bias_row ? (g.bias ? g.bias[row] : 0.f) : bcol;
        if (mask_pos) {
          if (!(g.residual[off] > 0.f)) v = 0.f;  // ReLU backward: gate by the saved activation
        } else if (g.residual) {
          v += g.residual[off];
        }
        if (accum) v += g.C[off];
        if (relu) v = fmaxf(v, 0.f);
        g.C[off] = v;
      }
    }
}

template <bool AK, bool BK_>
int launch(const GemmArgs& g, hipStream_t s) {
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  unsigned blocks;
  if (g.splits == 1) {
    const int tm_pad = (tiles_m + AIT_NXCD - 1) / AIT_NXCD * AIT_NXCD;
    blocks = (unsigned)(tm_pad * tiles_n);
  } else {
    const int per_xcd = (g.splits + AIT_NXCD - 1) / AIT_NXCD;
    blocks = (unsigned)(per_xcd * AIT_NXCD * tiles_m * tiles_n);
  }
  const size_t lds = sizeof(float) * 4 * BK * LDS_PITCH;
  hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_>), dim3(blocks), dim3(kThreads), lds, s, g);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

}  // namespace

AIT_API int ait_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha,
                         const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         const float* bias, const float* residual, int flags, int split_k,
                         int c_colblk, long long c_batch_stride, void* stream) {
  if (M < 0 || N < 0 || K < 0) return AIT_EINVAL;
  if (M == 0 || N == 0) return AIT_OK;
  if (!A || !B || !C) return AIT_EINVAL;
  // float4 staging: row pitches and bases 16-B aligned; K % 4 only matters for an operand whose
  // reduction dimension is the contiguous one
  if ((lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
      (reinterpret_cast<uintptr_t>(B) & 15) || ((K & 3) && (!trans_a || trans_b)))
    return AIT_EUNSUPPORTED;
  if (split_k < 1) split_k = 1;
  if (split_k > 1 && !(flags & AIT_GEMM_ATOMIC)) return AIT_EINVAL;
  if ((flags & AIT_GEMM_ATOMIC) && (bias || residual || (flags & AIT_GEMM_RELU)))
    return AIT_EINVAL;
  if ((flags & AIT_GEMM_MASK_POS) && !residual) return AIT_EINVAL;
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.residual = residual;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.c_colblk = c_colblk; g.c_batch = c_batch_stride; g.alpha = alpha; g.flags = flags;
  int kps = (K + split_k - 1) / split_k;
  kps = (kps + BK - 1) / BK * BK;
  g.k_per_split = kps;
  g.splits = (K + kps - 1) / kps;
  hipStream_t s = ait_stream(stream);
  // operand "K-contiguous" means the reduction dimension is the fast one in memory:
  //   A: !trans_a  (A is [M,K]);   B: trans_b (B is [N,K])
  const bool ak = !trans_a, bk = trans_b != 0;
  if (ak && bk) return launch<true, true>(g, s);
  if (ak && !bk) return launch<true, false>(g, s);
  if (!ak && bk) return launch<false, true>(g, s);
  return launch<false, false>(g, s);
}
