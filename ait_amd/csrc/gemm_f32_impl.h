// ait_amd/csrc/gemm_f32_impl.h -- the fp32 MFMA GEMM kernel, templated on its tile configuration.
// See gemm_f32.hip for the design notes.  Included by gemm_f32.hip (product instantiations) and by
// scripts/tune_gemm.hip (the tuning harness).
#pragma once
#include "common.h"

namespace ait_gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;      // [N] (or [M] with AIT_GEMM_BIAS_ROW)
  const float* residual;  // same addressing as C
  int M, N, K;
  int lda, ldb, ldc;
  int c_colblk;           // 0: plain row-major C.  >0: C(i,j) at (j/colblk)*c_batch + i*ldc + j%colblk
  long long c_batch;
  float alpha;
  int flags;
  int k_per_split;
  int splits;
};

// Tile configuration: BM x BN output tile, K-slabs of BK, WM x WN wavefronts each owning
// (BM/WM/32) x (BN/WN/32) MFMA tiles of 32x32.
template <int BM_, int BN_, int BK_, int WM_, int WN_, int MINW_, int OPT_ = 0>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, MINW = MINW_, OPT = OPT_;
  static constexpr int NT = 64 * WM * WN;          // threads
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr int PA = BM + 4, PB = BN + 4;   // LDS pitches (floats), 16-B aligned rows
  static constexpr int VA = BM * BK / 4 / NT;      // float4 per thread per A slab
  static constexpr int VB = BN * BK / 4 / NT;
  static constexpr int NBUF = (OPT_ & 4) ? 3 : 2;   // OPT bit 2: three-slab LDS ring
  static constexpr size_t LDS = sizeof(float) * NBUF * BK * (PA + PB);
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "tile / wave mismatch");
  static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "slab / thread mismatch");
};

// Stage one BK x ROWS slab of an operand into registers.
//   KCONTIG = true : element (r, k) at p[r*ld + k]   (reduction dim contiguous)
//   KCONTIG = false: element (r, k) at p[k*ld + r]
// Rows >= R and k >= Kend read as zero.  ld % 4 == 0 and 16-B aligned bases are required.
template <bool KCONTIG, int ROWS, int BK, int NT, int NV>
__device__ __forceinline__ void load_slab(const float* __restrict__ p, int ld, int r0, int R,
                                          int k0, int Kend, float4 (&v)[NV]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int e = t + i * NT;
    if (KCONTIG) {
      const int r = r0 + e / (BK / 4), k = k0 + (e % (BK / 4)) * 4;
      if (r < R && k < Kend)
        v[i] = *reinterpret_cast<const float4*>(p + (size_t)r * ld + k);
      else
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const int k = k0 + e / (ROWS / 4), r = r0 + (e % (ROWS / 4)) * 4;
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

template <bool KCONTIG, int ROWS, int BK, int NT, int NV, int PITCH>
__device__ __forceinline__ void store_slab(float* __restrict__ s, const float4 (&v)[NV]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int e = t + i * NT;
    if (KCONTIG) {
      const int r = e / (BK / 4), k = (e % (BK / 4)) * 4;
      s[(k + 0) * PITCH + r] = v[i].x;
      s[(k + 1) * PITCH + r] = v[i].y;
      s[(k + 2) * PITCH + r] = v[i].z;
      s[(k + 3) * PITCH + r] = v[i].w;
    } else {
      const int k = e / (ROWS / 4), r = (e % (ROWS / 4)) * 4;
      *reinterpret_cast<float4*>(s + k * PITCH + r) = v[i];
    }
  }
}

// ---- OPT bit 4: "row image" for operands whose reduction dimension is contiguous in memory ------
// Instead of transposing such an operand into the K-major slab (4 x ds_write_b32 per float4, one
// ds_read_b32 per MFMA operand), its slab is kept as it arrives: [row][BK = 16 floats] = 64-B rows
// of four 16-B chunks, chunk index XOR ((row >> 2) & 3) (conflict-free ds_write_b128 AND
// ds_read_b128, same geometry as gemm_bf16.hip).  A lane then fetches FOUR k-steps of one operand
// tile with one ds_read_b128.  This needs the k order of a slab to be: lane half lk works through
// k = 8*lk + s for MFMA step s = 0..7 (any order is legal as long as A and B agree; a K-major
// operand simply reads row 8*lk + s).
__device__ __forceinline__ int rowimg_off(int row, int chunk) {     // float offset of a 16-B chunk
  return row * 16 + ((chunk ^ ((row >> 2) & 3)) << 2);
}

template <int ROWS, int NT, int NV>
__device__ __forceinline__ void store_rowimg(float* __restrict__ s, const float4 (&v)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int e = threadIdx.x + i * NT;
    *reinterpret_cast<float4*>(s + rowimg_off(e >> 2, e & 3)) = v[i];
  }
}

// operands of k-step group g (4 MFMA steps) of one slab for TILES 32-row tiles starting at row0
template <bool ROWIMG, int TILES, int PITCH>
__device__ __forceinline__ void fetch_group(const float* __restrict__ slab, int row0, int li, int lk, int g,
                                            float4 (&x)[TILES]) {
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    if (ROWIMG) {
      x[t] = *reinterpret_cast<const float4*>(slab + rowimg_off(row0 + t * 32 + li, 2 * lk + g));
    } else {
      const float* p = slab + (8 * lk + 4 * g) * PITCH + row0 + t * 32 + li;
      x[t] = make_float4(p[0], p[PITCH], p[2 * PITCH], p[3 * PITCH]);
    }
  }
}

// ---- OPT bit 8 (256): slabs go global -> LDS directly (gfx950 global_load_lds_dwordx4) ------------
// One wave-wide instruction moves 64 x 16 B = 1 KB: every lane supplies its own global address, the
// data lands at (wave-uniform LDS base) + lane * 16 B.  No staging registers, no ds_write, no
// VGPR write-back traffic next to the MFMA results.  LDS images (unpadded, 1 KB granules):
//   reduction dim contiguous in memory: the row image of rowimg_off() -- granule q = rows
//       16q..16q+15; lane L fills slot (row 16q + L/4, position L%4) and therefore FETCHES the chunk
//       that the swizzle assigns to that slot;
//   reduction dim outermost: K-major [16][ROWS], granule q = 256 consecutive elements of it.
// Out-of-range rows are clamped to the last valid row (their products are never stored); the
// caller guarantees whole slabs (K range a multiple of 16) and ROWS-dim % 4 == 0 for K-outer.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <bool KCONTIG, int ROWS, int NWAVES>
__device__ __forceinline__ void dlds_load(const float* __restrict__ p, int ld, int r0, int R, int k0,
                                          float* __restrict__ slab, int wave, int lane) {
  constexpr int GRAN = ROWS * 16 / 256;          // 1-KB granules per slab
#pragma unroll
  for (int q0 = 0; q0 < GRAN; q0 += NWAVES) {
    const int q = q0 + wave;                     // wave-uniform
    if (GRAN % NWAVES != 0 && q >= GRAN) break;
    const float* src;
    if (KCONTIG) {
      const int row = q * 16 + (lane >> 2), pos = lane & 3;
      const int chunk = pos ^ ((row >> 2) & 3);
      src = p + (size_t)min(r0 + row, R - 1) * ld + k0 + chunk * 4;
    } else {
      const int e = q * 256 + lane * 4;          // element of the [16][ROWS] image
      const int k = e / ROWS, r = e % ROWS;
      src = p + (size_t)(k0 + k) * ld + min(r0 + r, R - 4);
    }
    // Issued as inline assembly on purpose: through the builtin the compiler treats the transfer
    // as a store to LDS that may alias every later ds_read and puts s_waitcnt vmcnt(0) in front of
    // the next operand fetch, i.e. it serialises the slab's memory latency with the MFMA stream.
    // Ordering is explicit here instead: a slot is requested only after the barrier that retired
    // its last reader, and awaited (vmcnt(0)) before the barrier that publishes it.
    const unsigned dst = (unsigned)(size_t)(lds_void*)(slab + q * 256);      // wave-uniform LDS address
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "m0", "memory");
  }
}

// ---- OPT bit 11 (2048): 32-float slabs, two stages, direct to LDS (experiment) ----------------------
// K-contiguous operands are then fetched as whole 128-B lines (a 16-float slab touches every
// line twice, in two consecutive slabs).  Row image: [row][32 floats] = eight 16-B chunks, chunk
// index XOR (row & 7); lane half lk works through k = 16*lk + s, s = 0..15.
__device__ __forceinline__ int rowimg32_off(int row, int chunk) { return row * 32 + ((chunk ^ (row & 7)) << 2); }

template <bool KCONTIG, int ROWS, int NWAVES>
__device__ __forceinline__ void dlds_load32(const float* __restrict__ p, int ld, int r0, int R, int k0,
                                            float* __restrict__ slab, int wave, int lane) {
  constexpr int GRAN = ROWS * 32 / 256;
#pragma unroll
  for (int q0 = 0; q0 < GRAN; q0 += NWAVES) {
    const int q = q0 + wave;
    if (GRAN % NWAVES != 0 && q >= GRAN) break;
    const float* src;
    if (KCONTIG) {
      const int row = q * 8 + (lane >> 3), pos = lane & 7;
      src = p + (size_t)min(r0 + row, R - 1) * ld + k0 + ((pos ^ (row & 7)) << 2);
    } else {
      const int e = q * 256 + lane * 4;
      const int k = e / ROWS, r = e % ROWS;
      src = p + (size_t)(k0 + k) * ld + min(r0 + r, R - 4);
    }
    const unsigned dst = (unsigned)(size_t)(lds_void*)(slab + q * 256);
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "m0", "memory");
  }
}

template <bool ROWIMG, int TILES, int ROWS>
__device__ __forceinline__ void fetch_group32(const float* __restrict__ slab, int row0, int li, int lk, int g,
                                              float4 (&x)[TILES]) {
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    const int row = row0 + t * 32 + li;
    if (ROWIMG) {
      x[t] = *reinterpret_cast<const float4*>(slab + rowimg32_off(row, 4 * lk + g));
    } else {
      const float* p = slab + (16 * lk + 4 * g) * ROWS + row;
      x[t] = make_float4(p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]);
    }
  }
}

// AK / BKC: true when that operand is stored with the reduction dimension contiguous.
//   forward  y = x W^T   : A = x [M,K] (AK), B = W [N,K] (BKC)
//   dgrad    dx = dy W   : A = dy [M,K'] (AK), B = W [K',N] (!BKC)
//   wgrad    dW = dy^T x : A = dy [K',M] (!AK), B = x [K',N] (!BKC)
enum { EPI_STORE = 0, EPI_ATOMIC = 1, EPI_AUX = 2 };

// C/D layout of the 32x32 MFMA (any input dtype): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int TM, int TN, int EPI>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[TM][TN], const GemmArgs& g, int m0, int n0,
                                         int wm, int wn, int li, int lk) {
  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool relu = (g.flags & AIT_GEMM_RELU) != 0;
  const bool bias_row = (g.flags & AIT_GEMM_BIAS_ROW) != 0;
#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++) {
      const int col = n0 + wn + b * 32 + li;
      const bool col_ok = col < g.N;
      const int colc = col_ok ? col : 0;
      size_t cbase;
      if (g.c_colblk > 0)
        cbase = (size_t)(colc / g.c_colblk) * g.c_batch + (colc % g.c_colblk);
      else
        cbase = colc;
      const int rbase = m0 + wm + a * 32 + 4 * lk;
      if (EPI == EPI_ATOMIC) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (col_ok && row < g.M) unsafeAtomicAdd(g.C + cbase + (size_t)row * g.ldc, g.alpha * acc[a][b][r]);
        }
      } else {
        float v[16];
        const float bcol = (g.bias && !bias_row) ? g.bias[colc] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          v[r] = g.alpha * acc[a][b][r] + (bias_row ? (g.bias ? g.bias[min(row, g.M - 1)] : 0.f) : bcol);
        }
        if (EPI == EPI_AUX) {
          const bool mask_pos = (g.flags & AIT_GEMM_MASK_POS) != 0;
          const bool accum = (g.flags & AIT_GEMM_ACCUMULATE) != 0;
          float x[16], y[16];
          // all loads first (clamped addresses, unconditional), then the arithmetic
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
            const size_t off = cbase + (size_t)row * g.ldc;
            x[r] = g.residual ? g.residual[off] : 0.f;
            y[r] = accum ? g.C[off] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; r++) {
            if (mask_pos) v[r] = x[r] > 0.f ? v[r] : 0.f;  // ReLU backward: gate by the saved activation
            else v[r] += x[r];
            v[r] += y[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (relu) v[r] = fmaxf(v[r], 0.f);
          if (col_ok && row < g.M) g.C[cbase + (size_t)row * g.ldc] = v[r];
        }
      }
    }
}

// EPI selects the epilogue at compile time (a run-time flag test per element makes hipcc branch
// around every load/store and wait vmcnt(0) each time):
//   EPI_STORE  C = alpha*acc (+bias) (relu)            -- no loads at all
//   EPI_ATOMIC C += alpha*acc with fp32 atomics        -- split-K partial tiles
//   EPI_AUX    the forms that read memory: +residual, ReLU-backward gate, accumulate into C
template <class C, bool AK, bool BKC, int EPI>
__global__ __launch_bounds__(C::NT, C::MINW) void gemm_f32_kernel(const GemmArgs g) {
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                              // [NBUF][BK][PA]
  float* Bs = lds + C::NBUF * BK * C::PA;       // [NBUF][BK][PB]

  // ---- XCD-aware work assignment (blocks b and b+8 share an XCD / L2) ----------------------
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int bid = blockIdx.x;
  const int xcd = bid % AIT_NXCD, j = bid / AIT_NXCD;
  int tm, tn, split;
  if (g.splits == 1) {
    // every XCD gets one contiguous chunk of the row-major tile list, so the N-tiles of an
    // M-panel run back to back on ONE XCD (its A panel stays in that L2) and all 8 XCDs are busy
    // whatever tiles_m is (bijective: ids past the end simply exit)
    const int total = tiles_m * tiles_n;
    const int chunk = (total + AIT_NXCD - 1) / AIT_NXCD;
    const int id = xcd * chunk + j;
    if (j >= chunk || id >= total) return;
    tm = id / tiles_n;
    tn = id % tiles_n;
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
  const int wm = (wave / C::WN) * (C::TM * 32), wn = (wave % C::WN) * (C::TN * 32);
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[C::TM][C::TN];
#pragma unroll
  for (int a = 0; a < C::TM; a++)
#pragma unroll
    for (int b = 0; b < C::TN; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

  float4 ra[C::VA], rb[C::VB];
  float av[C::TM], bv[C::TN];
  if constexpr ((C::OPT & 2048) != 0) {
    static_assert(BK == 32, "OPT 2048 is the 32-float-slab path");
    constexpr int SA = BM * 32, SB = BN * 32;
    constexpr int NW = C::NT / 64;
    float* Bd = lds + 2 * SA;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    auto request = [&](int slot, int k0) {
      dlds_load32<AK, BM, NW>(g.A, g.lda, m0, g.M, k0, As + slot * SA, uw, lane);
      dlds_load32<BKC, BN, NW>(g.B, g.ldb, n0, g.N, k0, Bd + slot * SB, uw, lane);
    };
    request(0, kbeg);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      if (k0 + BK < kend) request(cur ^ 1, k0 + BK);
      const float* as = As + cur * SA;
      const float* bs = Bd + cur * SB;
      float4 xa[C::TM], xb[C::TN], na[C::TM], nb[C::TN];
      fetch_group32<AK, C::TM, BM>(as, wm, li, lk, 0, xa);
      fetch_group32<BKC, C::TN, BN>(bs, wn, li, lk, 0, xb);
#pragma unroll
      for (int grp = 0; grp < 4; grp++) {
        __builtin_amdgcn_sched_barrier(0);
        if (grp < 3) {
          fetch_group32<AK, C::TM, BM>(as, wm, li, lk, grp + 1, na);
          fetch_group32<BKC, C::TN, BN>(bs, wn, li, lk, grp + 1, nb);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int a = 0; a < C::TM; a++)
#pragma unroll
            for (int b = 0; b < C::TN; b++) {
              const float fa = j == 0 ? xa[a].x : j == 1 ? xa[a].y : j == 2 ? xa[a].z : xa[a].w;
              const float fb = j == 0 ? xb[b].x : j == 1 ? xb[b].y : j == 2 ? xb[b].z : xb[b].w;
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a][b], 0, 0, 0);
            }
        }
        if (grp < 3) {
#pragma unroll
          for (int a = 0; a < C::TM; a++) xa[a] = na[a];
#pragma unroll
          for (int b = 0; b < C::TN; b++) xb[b] = nb[b];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0x0070);
      __syncthreads();
      cur ^= 1;
    }
  } else if constexpr (C::NBUF == 3 && (C::OPT & 256) != 0) {
    // ---- three-slab ring fed by direct-to-LDS loads (see dlds_load); operands fetched per group of
    // four k-steps.  Slab k+2 is requested at the top of iteration k into the slot that iteration
    // k-1 finished reading; it is awaited (vmcnt) just before the barrier that ends iteration k.
    static_assert(BK == 16 && BM % 16 == 0 && BN % 16 == 0, "direct-to-LDS path needs 16-float slabs");
    constexpr int SA = BM * 16, SB = BN * 16;          // floats per slab image
    constexpr int NW = C::NT / 64;
    float* Bd = lds + 3 * SA;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    if (C::OPT & 1024) {
      // TUNER experiment: the two workgroups of a CU are identical and start together, i.e. they
      // reach their per-slab barriers together; delay the second resident workgroup by half a slab
      if (((blockIdx.x / AIT_NXCD) / 32) & 1) __builtin_amdgcn_s_sleep(64);
    }
    auto request = [&](int slot, int k0) {
      dlds_load<AK, BM, NW>(g.A, g.lda, m0, g.M, k0, As + slot * SA, uw, lane);
      dlds_load<BKC, BN, NW>(g.B, g.ldb, n0, g.N, k0, Bd + slot * SB, uw, lane);
    };
    request(0, kbeg);
    if (kbeg + BK < kend) request(1, kbeg + BK);
    __builtin_amdgcn_s_waitcnt(0x0070);                // vmcnt(0) (lgkm/exp untouched)
    __syncthreads();
    float4 xa[C::TM], xb[C::TN], na[C::TM], nb[C::TN];
    fetch_group<AK, C::TM, BM>(As, wm, li, lk, 0, xa);
    fetch_group<BKC, C::TN, BN>(Bd, wn, li, lk, 0, xb);
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const int nxt = (cur == 2) ? 0 : cur + 1;
      const int nxt2 = (nxt == 2) ? 0 : nxt + 1;
      if (k0 + 2 * BK < kend) request(nxt2, k0 + 2 * BK);
#pragma unroll
      for (int grp = 0; grp < 2; grp++) {
        const float* an_ = As + (grp == 0 ? cur : nxt) * SA;
        const float* bn_ = Bd + (grp == 0 ? cur : nxt) * SB;
        __builtin_amdgcn_sched_barrier(0);
        fetch_group<AK, C::TM, BM>(an_, wm, li, lk, grp == 0 ? 1 : 0, na);
        fetch_group<BKC, C::TN, BN>(bn_, wn, li, lk, grp == 0 ? 1 : 0, nb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int a = 0; a < C::TM; a++)
#pragma unroll
            for (int b = 0; b < C::TN; b++) {
              const float fa = j == 0 ? xa[a].x : j == 1 ? xa[a].y : j == 2 ? xa[a].z : xa[a].w;
              const float fb = j == 0 ? xb[b].x : j == 1 ? xb[b].y : j == 2 ? xb[b].z : xb[b].w;
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a][b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int a = 0; a < C::TM; a++) xa[a] = na[a];
#pragma unroll
        for (int b = 0; b < C::TN; b++) xb[b] = nb[b];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (C::OPT & 512) {   // TUNER-ONLY ablation (may read a slab before it has landed): leave this
                            // iteration's three requests in flight across the barrier
        asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
      }
      cur = nxt;
    }
  } else if constexpr (C::NBUF == 3 && (C::OPT & 16) != 0 && (AK || BKC)) {
    // ---- three-slab ring with row images for the K-contiguous operand(s) and operands fetched per
    // group of four k-steps (see rowimg_off).  Requires BK == 16.
    static_assert(BK == 16, "row image needs 16-float slabs");
    constexpr int SA = BK * C::PA, SB = BK * C::PB;      // slab strides (the row image is smaller)
    auto put = [&](int slot) {
      if (AK) store_rowimg<BM, C::NT, C::VA>(As + slot * SA, ra);
      else store_slab<false, BM, BK, C::NT, C::VA, C::PA>(As + slot * SA, ra);
      if (BKC) store_rowimg<BN, C::NT, C::VB>(Bs + slot * SB, rb);
      else store_slab<false, BN, BK, C::NT, C::VB, C::PB>(Bs + slot * SB, rb);
    };
    load_slab<AK, BM, BK, C::NT, C::VA>(g.A, g.lda, m0, g.M, kbeg, kend, ra);
    load_slab<BKC, BN, BK, C::NT, C::VB>(g.B, g.ldb, n0, g.N, kbeg, kend, rb);
    put(0);
    load_slab<AK, BM, BK, C::NT, C::VA>(g.A, g.lda, m0, g.M, kbeg + BK, kend, ra);   // zeros past kend
    load_slab<BKC, BN, BK, C::NT, C::VB>(g.B, g.ldb, n0, g.N, kbeg + BK, kend, rb);
    put(1);
    __syncthreads();
    float4 xa[C::TM], xb[C::TN], na[C::TM], nb[C::TN];
    fetch_group<AK, C::TM, C::PA>(As, wm, li, lk, 0, xa);
    fetch_group<BKC, C::TN, C::PB>(Bs, wn, li, lk, 0, xb);
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more2 = k0 + 2 * BK < kend;
      if (more2) {
        load_slab<AK, BM, BK, C::NT, C::VA>(g.A, g.lda, m0, g.M, k0 + 2 * BK, kend, ra);
        load_slab<BKC, BN, BK, C::NT, C::VB>(g.B, g.ldb, n0, g.N, k0 + 2 * BK, kend, rb);
      }
      const int nxt = (cur == 2) ? 0 : cur + 1;
      const int nxt2 = (nxt == 2) ? 0 : nxt + 1;
#pragma unroll
      for (int grp = 0; grp < 2; grp++) {
        // next group's operands: second half of this slab, then the first half of the NEXT slab
        // (complete in LDS since the last barrier), so the MFMA stream runs across the barrier
        const float* an_ = As + (grp == 0 ? cur : nxt) * SA;
        const float* bn_ = Bs + (grp == 0 ? cur : nxt) * SB;
        __builtin_amdgcn_sched_barrier(0);
        fetch_group<AK, C::TM, C::PA>(an_, wm, li, lk, grp == 0 ? 1 : 0, na);
        fetch_group<BKC, C::TN, C::PB>(bn_, wn, li, lk, grp == 0 ? 1 : 0, nb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
          for (int a = 0; a < C::TM; a++)
#pragma unroll
            for (int b = 0; b < C::TN; b++) {
              const float fa = j == 0 ? xa[a].x : j == 1 ? xa[a].y : j == 2 ? xa[a].z : xa[a].w;
              const float fb = j == 0 ? xb[b].x : j == 1 ? xb[b].y : j == 2 ? xb[b].z : xb[b].w;
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a][b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int a = 0; a < C::TM; a++) xa[a] = na[a];
#pragma unroll
        for (int b = 0; b < C::TN; b++) xb[b] = nb[b];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more2) put(nxt2);
      __syncthreads();
      cur = nxt;
    }
  } else if constexpr (C::NBUF == 3) {
    // ---- three-slab ring: slab k+2 is fetched from global while slab k is multiplied; slab k+1 is
    // already complete in LDS, so the first operands of slab k+1 are read BEFORE the barrier that
    // ends slab k and the MFMA stream runs across the barrier without an LDS round trip.
    load_slab<AK, BM, BK, C::NT, C::VA>(g.A, g.lda, m0, g.M, kbeg, kend, ra);
    load_slab<BKC, BN, BK, C::NT, C::VB>(g.B, g.ldb, n0, g.N, kbeg, kend, rb);
    store_slab<AK, BM, BK, C::NT, C::VA, C::PA>(As, ra);
    store_slab<BKC, BN, BK, C::NT, C::VB, C::PB>(Bs, rb);
    load_slab<AK, BM, BK, C::NT, C::VA>(g.A, g.lda, m0, g.M, kbeg + BK, kend, ra);   // zeros past kend
    load_slab<BKC, BN, BK, C::NT, C::VB>(g.B, g.ldb, n0, g.N, kbeg + BK, kend, rb);
    store_slab<AK, BM, BK, C::NT, C::VA, C::PA>(As + BK * C::PA, ra);
    store_slab<BKC, BN, BK, C::NT, C::VB, C::PB>(Bs + BK * C::PB, rb);
    __syncthreads();
    {
      const float* as = As + wm + li;
      const float* bs = Bs + wn + li;
#pragma unroll
      for (int a = 0; a < C::TM; a++) av[a] = as[lk * C::PA + a * 32];
#pragma unroll
      for (int b = 0; b < C::TN; b++) bv[b] = bs[lk * C::PB + b * 32];
    }
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      // OPT bits 5/6/7 are TUNER-ONLY ablations (wrong results): 32 = no global loads, 64 = no LDS
      // stores, 128 = no barrier -- to see which part of the slab hand-over the matrix pipe waits on
      const bool more2 = k0 + 2 * BK < kend;
      if (more2 && !(C::OPT & 32)) {
        load_slab<AK, BM, BK, C::NT, C::VA>(g.A, g.lda, m0, g.M, k0 + 2 * BK, kend, ra);
        load_slab<BKC, BN, BK, C::NT, C::VB>(g.B, g.ldb, n0, g.N, k0 + 2 * BK, kend, rb);
      }
      const int nxt = (cur == 2) ? 0 : cur + 1;
      const int nxt2 = (nxt == 2) ? 0 : nxt + 1;
      const float* as = As + cur * BK * C::PA + wm + li;
      const float* bs = Bs + cur * BK * C::PB + wn + li;
      const float* asn = As + nxt * BK * C::PA + wm + li;
      const float* bsn = Bs + nxt * BK * C::PB + wn + li;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float an[C::TM], bn[C::TN];
        if (kk + 2 < BK) {
#pragma unroll
          for (int a = 0; a < C::TM; a++) an[a] = as[(kk + 2 + lk) * C::PA + a * 32];
#pragma unroll
          for (int b = 0; b < C::TN; b++) bn[b] = bs[(kk + 2 + lk) * C::PB + b * 32];
        } else {   // last k-step of the slab: first operands of the NEXT slab (complete since the last barrier)
#pragma unroll
          for (int a = 0; a < C::TM; a++) an[a] = asn[lk * C::PA + a * 32];
#pragma unroll
          for (int b = 0; b < C::TN; b++) bn[b] = bsn[lk * C::PB + b * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < C::TM; a++)
#pragma unroll
          for (int b = 0; b < C::TN; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < C::TM; a++) av[a] = an[a];
#pragma unroll
        for (int b = 0; b < C::TN; b++) bv[b] = bn[b];
        if ((C::OPT & 8) && kk == BK / 2) {
          // OPT bit 3: the slab fetched at the top of this iteration is written to the ring in
          // the MIDDLE of the MFMA stream (its loads have long landed), not in front of the barrier
          __builtin_amdgcn_sched_barrier(0);
          if (more2) {
            store_slab<AK, BM, BK, C::NT, C::VA, C::PA>(As + nxt2 * BK * C::PA, ra);
            store_slab<BKC, BN, BK, C::NT, C::VB, C::PB>(Bs + nxt2 * BK * C::PB, rb);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(C::OPT & 8) && more2 && !(C::OPT & 64)) {
        store_slab<AK, BM, BK, C::NT, C::VA, C::PA>(As + nxt2 * BK * C::PA, ra);
        store_slab<BKC, BN, BK, C::NT, C::VB, C::PB>(Bs + nxt2 * BK * C::PB, rb);
      }
      if (!(C::OPT & 128)) __syncthreads();
      cur = nxt;
    }
  } else {
  load_slab<AK, BM, BK, C::NT, C::VA>(g.A, g.lda, m0, g.M, kbeg, kend, ra);
  load_slab<BKC, BN, BK, C::NT, C::VB>(g.B, g.ldb, n0, g.N, kbeg, kend, rb);
  store_slab<AK, BM, BK, C::NT, C::VA, C::PA>(As, ra);
  store_slab<BKC, BN, BK, C::NT, C::VB, C::PB>(Bs, rb);
  __syncthreads();

  int cur = 0;
  // operands of the first k-step of the current slab (OPT bit 1: fetched before the barrier that
  // precedes the slab, so the MFMAs restart without an LDS round trip after it)
  {
    const float* as = As + wm + li;
    const float* bs = Bs + wn + li;
#pragma unroll
    for (int a = 0; a < C::TM; a++) av[a] = as[lk * C::PA + a * 32];
#pragma unroll
    for (int b = 0; b < C::TN; b++) bv[b] = bs[lk * C::PB + b * 32];
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      load_slab<AK, BM, BK, C::NT, C::VA>(g.A, g.lda, m0, g.M, k0 + BK, kend, ra);
      load_slab<BKC, BN, BK, C::NT, C::VB>(g.B, g.ldb, n0, g.N, k0 + BK, kend, rb);
    }
    const float* as = As + cur * BK * C::PA + wm + li;
    const float* bs = Bs + cur * BK * C::PB + wn + li;
    if (C::OPT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float an[C::TM], bn[C::TN];
      if (kk + 2 < BK) {   // software-pipelined operand fetch: next k-step's reads fly under these MFMAs
#pragma unroll
        for (int a = 0; a < C::TM; a++) an[a] = as[(kk + 2 + lk) * C::PA + a * 32];
#pragma unroll
        for (int b = 0; b < C::TN; b++) bn[b] = bs[(kk + 2 + lk) * C::PB + b * 32];
      }
      // keep the reads ABOVE the MFMAs (hipcc otherwise sinks them below, re-serialising the
      // LDS round trip with the matrix pipe: read -> wait -> 4 MFMA -> read -> wait ...)
      if (C::OPT & 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < C::TM; a++)
#pragma unroll
        for (int b = 0; b < C::TN; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      if (kk + 2 < BK) {
#pragma unroll
        for (int a = 0; a < C::TM; a++) av[a] = an[a];
#pragma unroll
        for (int b = 0; b < C::TN; b++) bv[b] = bn[b];
      }
    }
    if (C::OPT & 1) __builtin_amdgcn_s_setprio(0);
    if (more) {
      store_slab<AK, BM, BK, C::NT, C::VA, C::PA>(As + (cur ^ 1) * BK * C::PA, ra);
      store_slab<BKC, BN, BK, C::NT, C::VB, C::PB>(Bs + (cur ^ 1) * BK * C::PB, rb);
    }
    __syncthreads();
    cur ^= 1;
    if (more) {
      const float* as2 = As + cur * BK * C::PA + wm + li;
      const float* bs2 = Bs + cur * BK * C::PB + wn + li;
#pragma unroll
      for (int a = 0; a < C::TM; a++) av[a] = as2[lk * C::PA + a * 32];
#pragma unroll
      for (int b = 0; b < C::TN; b++) bv[b] = bs2[lk * C::PB + b * 32];
    }
  }
  }

  epilogue<C::TM, C::TN, EPI>(acc, g, m0, n0, wm, wn, li, lk);
}

template <class C, bool AK, bool BKC, int EPI>
int launch(const GemmArgs& g, hipStream_t s) {
  const int tiles_n = (g.N + C::BN - 1) / C::BN;
  const int tiles_m = (g.M + C::BM - 1) / C::BM;
  unsigned blocks;
  if (g.splits == 1) {
    const int chunk = (tiles_m * tiles_n + AIT_NXCD - 1) / AIT_NXCD;
    blocks = (unsigned)(chunk * AIT_NXCD);
  } else {
    const int per_xcd = (g.splits + AIT_NXCD - 1) / AIT_NXCD;
    blocks = (unsigned)(per_xcd * AIT_NXCD * tiles_m * tiles_n);
  }
  auto kern = gemm_f32_kernel<C, AK, BKC, EPI>;
  if (C::LDS > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(C::NT), C::LDS, s, g);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

template <class C, int EPI>
int dispatch_layout(const GemmArgs& g, bool ak, bool bk, hipStream_t s) {
  if (ak && bk) return launch<C, true, true, EPI>(g, s);
  if (ak && !bk) return launch<C, true, false, EPI>(g, s);
  if (!ak && bk) return launch<C, false, true, EPI>(g, s);
  return launch<C, false, false, EPI>(g, s);
}

template <class C>
int dispatch(const GemmArgs& g, bool ak, bool bk, hipStream_t s) {
  if (g.flags & AIT_GEMM_ATOMIC) return dispatch_layout<C, EPI_ATOMIC>(g, ak, bk, s);
  if (g.residual || (g.flags & (AIT_GEMM_ACCUMULATE | AIT_GEMM_MASK_POS)))
    return dispatch_layout<C, EPI_AUX>(g, ak, bk, s);
  return dispatch_layout<C, EPI_STORE>(g, ak, bk, s);
}

// Validate arguments and fill GemmArgs (shared by the product entry point and the tuner).
inline int make_args(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A,
                     int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                     const float* residual, int flags, int split_k, int c_colblk,
                     long long c_batch_stride, int BK, GemmArgs& g) {
  if (M < 0 || N < 0 || K < 0) return AIT_EINVAL;
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
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.residual = residual;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.c_colblk = c_colblk; g.c_batch = c_batch_stride; g.alpha = alpha; g.flags = flags;
  int kps = (K + split_k - 1) / split_k;
  kps = (kps + BK - 1) / BK * BK;
  g.k_per_split = kps;
  g.splits = (K + kps - 1) / kps;
  return AIT_OK;
}

}  // namespace ait_gemm
