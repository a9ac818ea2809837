// ait_amd/csrc/gemm_bf16.hip -- the bf16 matrix-core variant of the AIT GEMM (BASELINE cfg 5:
// "bf16 ... fp16 MFMA path").  Same interface, layouts and fused epilogues as ait_gemm_f32; the
// operands stay fp32 in HBM and are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32)
// on their way into LDS, the products run on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate)
// and accumulate in fp32.  At that rate the kernel is no longer matrix-pipe bound but bound by
// the fp32 operand traffic out of L2/HBM, so the tile is the larger 256x128 and K advances 32
// per slab.
//
// LDS image of an operand whose reduction dim is contiguous in memory (for the other kind see
// Stage::store): row-major [rows][32 bf16] = 64 B per row in 16-B chunks of 8 consecutive k; the
// bf16 MFMA wants exactly one such chunk per lane (lane l: row l&31, k-chunk l>>5), fetched with
// one ds_read_b128.  Chunk index XOR ((row>>2)&3) makes that read conflict-free (four rows share
// a 256-B bank row; the four 16-lane groups of a b128 read then hit four different chunks).
//   reduction dim contiguous in memory ([m][k]): 2 float4 -> 8 bf16 -> one ds_write_b128
//   reduction dim outermost ([k][m]):            4 float4 (4 k x 4 m) -> four ds_write_b64
#include "gemm_f32_impl.h"

namespace {
using namespace ait_gemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int ROWB = BK * 2;                // bytes per LDS row

template <int BM_, int BN_, int WM_, int WN_>
struct BCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr size_t LDS_BYTES = (size_t)2 * (BM + BN) * ROWB;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {   // byte offset of a 16-B chunk
  return row * ROWB + ((chunk ^ ((row >> 2) & 3)) << 4);
}

// ---- staging: global fp32 -> registers (raw) -> bf16 -> LDS -----------------------------------
// KCONTIG: item = (row, chunk of 8 k): 2 float4.   !KCONTIG: item = (4 rows, 4 k): 4 float4.
template <bool KCONTIG, int ROWS, int NT>
struct Stage {
  static constexpr int ITEMS = KCONTIG ? ROWS * 4 : (ROWS / 4) * 8;
  static constexpr int PER = (ITEMS + NT - 1) / NT;
  static constexpr int NV = KCONTIG ? 2 : 4;
  float4 v[PER][NV];

  __device__ __forceinline__ void load(const float* __restrict__ p, int ld, int r0, int R, int k0,
                                       int Kend) {
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const int e = threadIdx.x + i * NT;
      if (KCONTIG) {
        const int row = e >> 2, c = e & 3;
        const int r = r0 + row, k = k0 + c * 8;
#pragma unroll
        for (int q = 0; q < 2; q++) {
          if (e < ITEMS && r < R && k + 4 * q < Kend)
            v[i][q] = *reinterpret_cast<const float4*>(p + (size_t)r * ld + k + 4 * q);
          else
            v[i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        const int rq = e % (ROWS / 4), c8 = e / (ROWS / 4);
        const int r = r0 + rq * 4;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int k = k0 + c8 * 4 + q;
          if (e < ITEMS && k < Kend && r + 3 < R) {
            v[i][q] = *reinterpret_cast<const float4*>(p + (size_t)k * ld + r);
          } else if (e < ITEMS && k < Kend && r < R) {
            const float* s = p + (size_t)k * ld + r;
            v[i][q].x = s[0];
            v[i][q].y = (r + 1 < R) ? s[1] : 0.f;
            v[i][q].z = (r + 2 < R) ? s[2] : 0.f;
            v[i][q].w = 0.f;
          } else {
            v[i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
  }

  __device__ __forceinline__ void store(char* __restrict__ lds) const {
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const int e = threadIdx.x + i * NT;
      if (e >= ITEMS) continue;
      if (KCONTIG) {
        const int row = e >> 2, c = e & 3;
        const float f[8] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w, v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w};
        bf16x8 o;
#pragma unroll
        for (int q = 0; q < 8; q++) o[q] = (__bf16)f[q];
        *reinterpret_cast<bf16x8*>(lds + lds_off(row, c)) = o;
      } else {
        // K-outer operand: "k-pair-major" image  word[kp][row] = (bf16 k = 2kp, bf16 k = 2kp + 1).
        // The item's 4 rows x 4 k become two rows of four consecutive words: two conflict-free
        // ds_write_b128 (consecutive lanes -> consecutive 16 B).  The MFMA side reads its 8
        // consecutive k of one row as four ds_read_b32 down the kp axis (lanes = consecutive rows).
        const int rq = e % (ROWS / 4), c8 = e / (ROWS / 4);
        const float* f = reinterpret_cast<const float*>(&v[i][0]);   // f[q*4 + j] = (k = c8*4+q, row rq*4+j)
#pragma unroll
        for (int h = 0; h < 2; h++) {                                 // kp = c8*2 + h  <-  k = c8*4 + 2h, +1
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float x0 = f[(2 * h) * 4 + j], x1 = f[(2 * h + 1) * 4 + j];
            o[2 * j] = (__bf16)x0;
            o[2 * j + 1] = (__bf16)x1;
          }
          const int off = ((c8 * 2 + h) * ROWS + rq * 4) * 4;
          *reinterpret_cast<bf16x8*>(lds + off) = o;
        }
      }
    }
  }
};

// one operand tile's 8 consecutive k (k-step ks, lane half lk) for row `row` out of an LDS image
template <bool KCONTIG, int ROWS>
__device__ __forceinline__ bf16x8 fetch8(const char* __restrict__ img, int row, int ks, int lk) {
  if (KCONTIG) return *reinterpret_cast<const bf16x8*>(img + lds_off(row, ks * 2 + lk));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned* __restrict__ w = reinterpret_cast<const unsigned*>(img) + (ks * 8 + lk * 4) * ROWS + row;
  u32x4 p = {w[0], w[ROWS], w[2 * ROWS], w[3 * ROWS]};
  return __builtin_bit_cast(bf16x8, p);
}

template <class C, bool AK, bool BKC, int EPI>
__global__ __launch_bounds__(C::NT, 2) void gemm_bf16_kernel(const GemmArgs g) {
  constexpr int BM = C::BM, BN = C::BN, TM = C::TM, TN = C::TN, WN = C::WN;
  constexpr int A_BUF = BM * ROWB, B_BUF = BN * ROWB;            // bytes of one operand image
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* As = lds;                          // [2 buffers][BM][64 B]
  char* Bs = lds + 2 * A_BUF;              // [2 buffers][BN][64 B]

  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int bid = blockIdx.x;
  const int xcd = bid % AIT_NXCD, j = bid / AIT_NXCD;
  int tm, tn, split;
  if (g.splits == 1) {
    const int total = tiles_m * tiles_n;
    const int chunk = (total + AIT_NXCD - 1) / AIT_NXCD;
    const int id = xcd * chunk + j;
    if (j >= chunk || id >= total) return;
    tm = id / tiles_n;
    tn = id % tiles_n;
    split = 0;
  } else {
    const int tiles = tiles_m * tiles_n;
    const int per_xcd = (g.splits + AIT_NXCD - 1) / AIT_NXCD;
    split = xcd * per_xcd + j / tiles;
    const int t = j % tiles;
    tm = t / tiles_n;
    tn = t % tiles_n;
    if (split >= g.splits) return;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  if (kbeg >= kend) return;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave / WN) * (TM * 32), wn = (wave % WN) * (TN * 32);
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  Stage<AK, BM, C::NT> sa;
  Stage<BKC, BN, C::NT> sb;
  sa.load(g.A, g.lda, m0, g.M, kbeg, kend);
  sb.load(g.B, g.ldb, n0, g.N, kbeg, kend);
  sa.store(As);
  sb.store(Bs);
  __syncthreads();

  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      sa.load(g.A, g.lda, m0, g.M, k0 + BK, kend);
      sb.load(g.B, g.ldb, n0, g.N, k0 + BK, kend);
    }
    const char* as = As + cur * A_BUF;
    const char* bs = Bs + cur * B_BUF;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      bf16x8 av[TM], bv[TN];
#pragma unroll
      for (int a = 0; a < TM; a++) av[a] = fetch8<AK, BM>(as, wm + a * 32 + li, ks, lk);
#pragma unroll
      for (int b = 0; b < TN; b++) bv[b] = fetch8<BKC, BN>(bs, wn + b * 32 + li, ks, lk);
#pragma unroll
      for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    if (more) {
      sa.store(As + (cur ^ 1) * A_BUF);
      sb.store(Bs + (cur ^ 1) * B_BUF);
    }
    __syncthreads();
    cur ^= 1;
  }
  epilogue<TM, TN, EPI>(acc, g, m0, n0, wm, wn, li, lk);
}

template <class C, bool AK, bool BKC, int EPI>
int launch_bf16(const GemmArgs& g, hipStream_t s) {
  const int tiles_n = (g.N + C::BN - 1) / C::BN;
  const int tiles_m = (g.M + C::BM - 1) / C::BM;
  unsigned blocks;
  if (g.splits == 1) {
    blocks = (unsigned)((tiles_m * tiles_n + AIT_NXCD - 1) / AIT_NXCD * AIT_NXCD);
  } else {
    const int per_xcd = (g.splits + AIT_NXCD - 1) / AIT_NXCD;
    blocks = (unsigned)(per_xcd * AIT_NXCD * tiles_m * tiles_n);
  }
  auto kern = gemm_bf16_kernel<C, AK, BKC, EPI>;
  if (C::LDS_BYTES > 64 * 1024)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)C::LDS_BYTES) != hipSuccess)
      return AIT_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(C::NT), C::LDS_BYTES, s, g);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

template <class C, int EPI>
int dispatch_layout_bf16(const GemmArgs& g, bool ak, bool bk, hipStream_t s) {
  if (ak && bk) return launch_bf16<C, true, true, EPI>(g, s);
  if (ak && !bk) return launch_bf16<C, true, false, EPI>(g, s);
  if (!ak && bk) return launch_bf16<C, false, true, EPI>(g, s);
  return launch_bf16<C, false, false, EPI>(g, s);
}

template <class C>
int dispatch_bf16(const GemmArgs& g, bool ak, bool bk, hipStream_t s) {
  if (g.flags & AIT_GEMM_ATOMIC) return dispatch_layout_bf16<C, EPI_ATOMIC>(g, ak, bk, s);
  const bool row_bias = g.bias && (g.flags & AIT_GEMM_BIAS_ROW);
  if (g.residual && !(g.flags & AIT_GEMM_ACCUMULATE) && !row_bias) return dispatch_layout_bf16<C, EPI_RES>(g, ak, bk, s);
  if (g.residual || (g.flags & (AIT_GEMM_ACCUMULATE | AIT_GEMM_MASK_POS)) || row_bias)
    return dispatch_layout_bf16<C, EPI_AUX>(g, ak, bk, s);
  return dispatch_layout_bf16<C, EPI_STORE>(g, ak, bk, s);
}

using Bf16Tile = BCfg<256, 128, 4, 2>;

int run_gemm(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A,
             int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
             const float* residual, int flags, int split_k, int c_colblk, long long c_batch_stride,
             const ait_launch_ctx* ctx, void* stream) {
  if (M == 0 || N == 0) return (M < 0 || N < 0 || K < 0) ? AIT_EINVAL : AIT_OK;
  GemmArgs g;
  const int rc = make_args(trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, residual,
                           flags, split_k, c_colblk, c_batch_stride, BK, g);
  if (rc != AIT_OK) return rc;
  hipStream_t s = ait_stream(stream);
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, 2.0 * M * N * K, s, M, N, K, trans_a, trans_b, g.splits);
  return dispatch_bf16<Bf16Tile>(g, !trans_a, trans_b != 0, s);
}

}  // namespace

AIT_API int ait_gemm_bf16(int trans_a, int trans_b, int M, int N, int K, float alpha,
                          const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          const float* bias, const float* residual, int flags, int split_k,
                          int c_colblk, long long c_batch_stride, const ait_launch_ctx* ctx, void* stream) {
  return run_gemm(trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, residual,
                  flags, split_k, c_colblk, c_batch_stride, ctx, stream);
}
