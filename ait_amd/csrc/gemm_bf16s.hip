// ait_amd/csrc/gemm_bf16s.hip -- GEMM with bf16 operands STORED in memory (BASELINE configs[4], cfgs/res101.yml:
// "bf16 ... fp16 MFMA path"): the linears of the AIT (Models.py:246-247,278, SubLayers.py:77-79,97,181-183) when the
// activations between sub-layers and the per-step weight copies are kept in bf16.
//
//   C[M, N] (f32 and / or bf16) = A[M, K] (bf16, K contiguous) . B[N, K]^T (bf16, K contiguous)  (+ bias[N]) (ReLU)
//                                 (+ residual[M, N] f32 | zeroed where gate[M, N] <= 0)
//   one v_mfma_f32_32x32x16_bf16 per 32x32x16 block, f32 accumulate.
//
// Why a second kernel: the f32-storage tiles (gemm_f32_impl.h, KNOB_BF16) stream 4-byte operands through LDS for one MFMA
// per block and are bound by that traffic at 345-470 TFLOP/s (profiles/r04_cfg5_bench_line.json: 374).  With 2-byte
// operands a 32-deep slab of a 256 x 128 tile is 24 KB: three slabs in flight and two to three workgroups per CU, so that
// one workgroup's epilogue (the products of the AIT are 512 deep: an output tile per eight slabs) runs under the other's
// MFMAs.  Output layout: the MFMA is issued with the operands swapped (D^T = B A^T), so a lane holds FOUR CONSECUTIVE
// columns of one output row per accumulator quad: 16-byte (f32) / 8-byte (bf16) stores, and bias / residual / gate are
// fetched with the same vector width.
//
// Persistent launch, XCD-aware tile order (blocks b and b + 8 share an L2; every XCD owns a contiguous chunk of the
// row-major tile list: an A row panel's column tiles meet in one L2); operand slabs global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 16 rows x 64 B per instruction), 16-B chunk c of row r stored at chunk position
// c ^ ((r >> 2) & 3): the ds_read_b128 of 32 consecutive rows is bank-conflict-free.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BM = 256, BN = 128, BK = 32, NT = 256, NS = 3;
constexpr int ROWB = BK * 2;                       // bytes per operand row of a slab (64)
constexpr int STAGE = (BM + BN) * ROWB;            // 24 KB
constexpr int GRAN = 16 * ROWB;                    // one LDS-DMA instruction: 16 rows x 64 B = 1 KB
constexpr int GA = BM / 16, GB = BN / 16;          // granules per slab: 16 + 8

struct Args {
  const unsigned short* A;
  const unsigned short* B;
  float* C32;
  unsigned short* C16;
  const float* bias;
  const float* residual;      // added (EPI_RES) or read as the gate (EPI_GATE: value kept where residual > 0)
  const unsigned short* gate16;   // EPI_GATE with a bf16 gate tensor (the stored ReLU output) instead of `residual`
  int M, N, K;
  long long lda, ldb, ldc32, ldc16, ldr;
  int relu;
};
enum { EPI_PLAIN = 0, EPI_RES = 1, EPI_GATE = 2 };

__device__ __forceinline__ void glds16(const void* src, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "m0", "memory");
}
__device__ __forceinline__ unsigned pack2(float a, float b) {         // v_cvt_pk_bf16_f32, nearest even
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 t;
  t[0] = (__bf16)a;
  t[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, t);
}

template <int EPI>
__global__ __launch_bounds__(NT, 2) void gemm_bf16s_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = g.N / BN, tiles_m = (g.M + BM - 1) / BM, tiles = tiles_m * tiles_n;
  const int per = (tiles + AIT_NXCD - 1) / AIT_NXCD, wg_per_xcd = gridDim.x / AIT_NXCD;
  const int xcd = blockIdx.x % AIT_NXCD, j = blockIdx.x / AIT_NXCD;
  const int chunk_end = min(per, tiles - xcd * per);
  const int mine = j < chunk_end ? (chunk_end - j + wg_per_xcd - 1) / wg_per_xcd : 0;
  if (mine <= 0) return;
  const int wm = (wave >> 1) * 128, wn = (wave & 1) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int slabs = g.K / BK, total = mine * slabs;
  const int rr = lane >> 2, pos = lane & 3;                        // row within a granule, chunk POSITION in LDS
  const int cfetch = pos ^ ((rr >> 2) & 3);                          // the 16-B chunk of the row that lands there
  const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds;
  auto tile_origin = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
    const int t = xcd * per + j + i * wg_per_xcd;
    m0 = (t / tiles_n) * BM;
    n0 = (t % tiles_n) * BN;
  };
  // loader: 24 granules per slab, six per wave (A granules 0..15: waves take q = wave, wave + 4, ...; then B's eight)
  auto issue = [&](int s, int stage) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(s / slabs, m0, n0);
    const int k0 = (s % slabs) * BK;
#pragma unroll
    for (int i = 0; i < GA / 4; i++) {
      const int q = wave + i * 4;
      int row = m0 + q * 16 + rr;
      row = row < g.M ? row : g.M - 1;                               // (rows past M: any finite data, never stored)
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE + q * GRAN);
      glds16(g.A + (size_t)row * g.lda + k0 + cfetch * 8, dst);
    }
#pragma unroll
    for (int i = 0; i < GB / 4; i++) {
      const int q = wave + i * 4;
      const int row = n0 + q * 16 + rr;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE + BM * ROWB + q * GRAN);
      glds16(g.B + (size_t)row * g.ldb + k0 + cfetch * 8, dst);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
  issue(0, 0);
  if (total > 1) issue(1, 1);
  int done = 0, ti = 0, stage = 0;
  bool stores_pending = false;
  for (int s = 0; s < total; s++) {
    // slab s must have landed: everything but the six loads of slab s + 1 (loads complete in order; stores of an
    // epilogue may overtake loads in the counter, so after one everything is awaited)
    if (stores_pending || s + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    stores_pending = false;
    __builtin_amdgcn_s_barrier();                                    // ... for every wave; and all are done with slab s - 1
    if (s + 2 < total) issue(s + 2, (stage + 2) % NS);
    const unsigned char* sa = lds + stage * STAGE;
    const unsigned char* sb = sa + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      const int c = 2 * ks + lk;
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int row = wm + a * 32 + li;
        fa[a] = *reinterpret_cast<const bf16x8*>(sa + row * ROWB + ((c ^ ((row >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int row = wn + b * 32 + li;
        fb[b] = *reinterpret_cast<const bf16x8*>(sb + row * ROWB + ((c ^ ((row >> 2) & 3)) << 4));
      }
      // operands swapped: D^T = B A^T, lane (li, lk) holds row m = li, columns 8 q + 4 lk + (r & 3), q = r >> 2
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
    if (++done == slabs) {
      int m0, n0;
      tile_origin(ti, m0, n0);
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int row = m0 + wm + a * 32 + li;
        const bool ok = row < g.M;
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int col = n0 + wn + b * 32 + 8 * q + 4 * lk;
            float4 v = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
            if (g.bias) {
              const float4 bb = *reinterpret_cast<const float4*>(g.bias + col);
              v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (ok) {
              if constexpr (EPI == EPI_RES) {
                const float4 rv = *reinterpret_cast<const float4*>(g.residual + (size_t)row * g.ldr + col);
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
              }
              if constexpr (EPI == EPI_GATE) {
                if (g.gate16) {
                  const uint2 gv = *reinterpret_cast<const uint2*>(g.gate16 + (size_t)row * g.ldr + col);
                  // (a bf16 value is positive iff its 16 bits, read as a signed short, are > 0: +0 is 0, negatives and -0 < 0)
                  v.x = (short)(gv.x & 0xffffu) > 0 ? v.x : 0.f; v.y = (short)(gv.x >> 16) > 0 ? v.y : 0.f;
                  v.z = (short)(gv.y & 0xffffu) > 0 ? v.z : 0.f; v.w = (short)(gv.y >> 16) > 0 ? v.w : 0.f;
                } else {
                  const float4 rv = *reinterpret_cast<const float4*>(g.residual + (size_t)row * g.ldr + col);
                  v.x = rv.x > 0.f ? v.x : 0.f; v.y = rv.y > 0.f ? v.y : 0.f;
                  v.z = rv.z > 0.f ? v.z : 0.f; v.w = rv.w > 0.f ? v.w : 0.f;
                }
              }
              if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
              if (g.C32) *reinterpret_cast<float4*>(g.C32 + (size_t)row * g.ldc32 + col) = v;
              if (g.C16) *reinterpret_cast<uint2*>(g.C16 + (size_t)row * g.ldc16 + col) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
            }
#pragma unroll
            for (int r = 0; r < 4; r++) acc[a][b][4 * q + r] = 0.f;
          }
      }
      done = 0;
      ti++;
      stores_pending = true;
    }
  }
}

// f32 [rows, cols] (row pitch ld_src) -> bf16 [rows, cols] (row pitch ld_dst), nearest even; 16 B in, 8 B out per lane
__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ src, long long rows, int cols, long long ld_src,
                                                      unsigned short* __restrict__ dst, long long ld_dst) {
  const int c4 = cols / 4;
  const long long n = rows * c4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
    *reinterpret_cast<uint2*>(dst + r * ld_dst + c) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
  }
}
// ... and the transposed copy: dst[c, r] = bf16(src[r, c]) through a 64 x 64 LDS tile (wgrad-side operands, weights^T)
__global__ __launch_bounds__(256) void to_bf16_t_kernel(const float* __restrict__ src, long long rows, int cols, long long ld_src,
                                                        unsigned short* __restrict__ dst, long long ld_dst) {
  __shared__ float tile[64][65];
  const long long r0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(r0 + r) * ld_src + c0 + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 32; i += 256) {
    const int c = i >> 5, r = (i & 31) * 2;
    if (c0 + c < cols && r0 + r < rows) {
      if (r0 + r + 1 < rows)
        *reinterpret_cast<unsigned*>(dst + (long long)(c0 + c) * ld_dst + r0 + r) = pack2(tile[r][c], tile[r + 1][c]);
      else
        dst[(long long)(c0 + c) * ld_dst + r0 + r] = (unsigned short)(pack2(tile[r][c], 0.f) & 0xffffu);
    }
  }
}

template <int EPI>
int launch(const Args& g, hipStream_t s) {
  const void* kern = reinterpret_cast<const void*>(gemm_bf16s_kernel<EPI>);
  constexpr int kLds = NS * STAGE;
  static int slots = 0;           // (a constant of the code object and the device model)
  if (slots == 0) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return AIT_ELAUNCH;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
      cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, kLds) != hipSuccess || per_cu <= 0) {
      (void)hipGetLastError();
      per_cu = 2;
    }
    slots = (per_cu > 2 ? 2 : per_cu) * cus;
  }
  const int tiles = ((g.M + BM - 1) / BM) * (g.N / BN);
  const int per = (tiles + AIT_NXCD - 1) / AIT_NXCD;
  int w = slots / AIT_NXCD;
  if (w > per) w = per;
  if (w < 1) w = 1;
  hipLaunchKernelGGL(gemm_bf16s_kernel<EPI>, dim3(w * AIT_NXCD), dim3(NT), kLds, s, g);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

}  // namespace

AIT_API int ait_gemm_bf16s(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, float* C32,
                           long long ldc32, void* C16, long long ldc16, const float* bias, const float* residual,
                           const void* gate16, long long ldr, int flags, const ait_launch_ctx* ctx, void* stream) {
  if (M < 0 || N < 0 || K < 0) return AIT_EINVAL;
  if (M == 0 || N == 0) return AIT_OK;
  if (!A || !B || (!C32 && !C16)) return AIT_EINVAL;
  if (K == 0 || (K % BK) || (N % BN) || (lda % 8) || (ldb % 8) || lda < K || ldb < K ||
      (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15))
    return AIT_EUNSUPPORTED;
  if ((C32 && ((ldc32 % 4) || ldc32 < N || (reinterpret_cast<uintptr_t>(C32) & 15))) ||
      (C16 && ((ldc16 % 4) || ldc16 < N || (reinterpret_cast<uintptr_t>(C16) & 7))))
    return AIT_EUNSUPPORTED;
  if (flags & ~(AIT_GEMM_RELU | AIT_GEMM_MASK_POS)) return AIT_EUNSUPPORTED;
  const bool gate = (flags & AIT_GEMM_MASK_POS) != 0;
  if (gate && !residual && !gate16) return AIT_EINVAL;
  if (!gate && gate16) return AIT_EINVAL;
  if ((residual || gate16) && ((ldr % 4) || ldr < N)) return AIT_EUNSUPPORTED;
  Args g;
  g.A = static_cast<const unsigned short*>(A); g.B = static_cast<const unsigned short*>(B);
  g.C32 = C32; g.C16 = static_cast<unsigned short*>(C16);
  g.bias = bias; g.residual = residual; g.gate16 = static_cast<const unsigned short*>(gate16);
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc32 = ldc32; g.ldc16 = ldc16; g.ldr = ldr;
  g.relu = (flags & AIT_GEMM_RELU) ? 1 : 0;
  hipStream_t s = ait_stream(stream);
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, 2.0 * M * N * K, s, M, N, K, 0, 1, 1);
  if (gate) return launch<EPI_GATE>(g, s);
  if (residual) return launch<EPI_RES>(g, s);
  return launch<EPI_PLAIN>(g, s);
}

AIT_API int ait_f32_to_bf16(const float* src, long long rows, int cols, long long ld_src, void* dst, long long ld_dst,
                            int transpose, void* stream) {
  if (rows < 0 || cols < 0) return AIT_EINVAL;
  if (rows == 0 || cols == 0) return AIT_OK;
  if (!src || !dst || ld_src < cols) return AIT_EINVAL;
  hipStream_t s = ait_stream(stream);
  if (transpose) {
    if (ld_dst < rows || (ld_dst % 2) || (reinterpret_cast<uintptr_t>(dst) & 3)) return AIT_EUNSUPPORTED;
    const long long bx = (rows + 63) / 64;
    if (bx > 0x7fffffffLL || (cols + 63) / 64 > 65535) return AIT_EUNSUPPORTED;
    hipLaunchKernelGGL(to_bf16_t_kernel, dim3((unsigned)bx, (unsigned)((cols + 63) / 64)), dim3(256), 0, s, src, rows, cols, ld_src,
                       static_cast<unsigned short*>(dst), ld_dst);
  } else {
    if ((cols % 4) || (ld_src % 4) || (ld_dst % 4) || ld_dst < cols || (reinterpret_cast<uintptr_t>(src) & 15) ||
        (reinterpret_cast<uintptr_t>(dst) & 7))
      return AIT_EUNSUPPORTED;
    const long long want = (rows * (cols / 4) + 255) / 256;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, s, src, rows, cols, ld_src,
                       static_cast<unsigned short*>(dst), ld_dst);
  }
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}
