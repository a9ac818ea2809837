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
#include "gemm_internal.h"

namespace {
using ait_bf16s::Conv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Tile families (every wave owns 128 x 64 outputs = 4 x 2 MFMA blocks, 128 accumulator registers):
//   Small  256 x 128 x 32, 4 waves, three 24-KB stages: two workgroups per CU -- one's epilogue under the other's MFMAs.
//          Outputs narrower than 256 columns or too few tiles for the other.
//   Big    256 x 256 x 64, 8 waves, two 64-KB stages: one workgroup per CU.  A third less operand traffic per product
//          (128 against 85 FLOP per byte fetched into LDS) in whole 128-B lines instead of 64-B halves, a barrier per 32
//          MFMAs of a wave instead of 16.  These kernels are bound by L2 -> LDS bandwidth (at the pipe's ~2000 TFLOP/s the
//          small tile would pull 24 TB/s), which is why the big tile wins even on the 512-deep products whose epilogue it
//          cannot hide: 510-560 against 410-460 TFLOP/s with an f32 result, 610-665 against 520-555 with a bf16 one
//          (profiles/r05_bf16_storage_gemm_product_kernel.txt).  K >= 512, and the weight gradients.
template <int BM_, int BN_, int BK_, int NS_, int WM_, int WN_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, NS = NS_, WM = WM_, WN = WN_;
  static constexpr int NW = WM * WN, NT = 64 * NW;
  static constexpr int ROWB = BK * 2;                  // bytes per operand row of a K-contiguous slab
  static constexpr int CH = ROWB / 16;                 // 16-B chunks per such row (4 or 8)
  static constexpr int RG = 64 / CH;                   // rows per 1-KB LDS-DMA granule
  static constexpr int STAGE = (BM + BN) * ROWB;
  static constexpr int LDS = NS * STAGE;
  static constexpr int GRANULES = STAGE / 1024;
  static constexpr int LPW = GRANULES / NW;            // LDS-DMA instructions per wave and slab
  static_assert(BM / WM == 128 && BN / WN == 64, "a wave owns 128 x 64 outputs");
  static_assert(GRANULES % NW == 0 && (CH == 4 || CH == 8) && BK % 16 == 0, "tile");
};
using Small = TileCfg<256, 128, 32, 3, 2, 2>;
using Big = TileCfg<256, 256, 64, 2, 2, 4>;
// (weight gradients of 64-column outputs -- d fc_w = df^T u of the attention blocks: two waves, 20-KB stages)
using Narrow = TileCfg<256, 64, 32, 3, 2, 1>;

struct Args {
  const unsigned short* A;
  const unsigned short* B;
  float* C32;
  unsigned short* C16;
  const float* bias;
  const float* residual;      // added (EPI_RES / EPI_RESGATE) or read as the gate (EPI_GATE: value kept where residual > 0)
  const unsigned short* res16;    // ... the addend as a bf16 tensor instead (pitch ldr)
  const unsigned short* gate16;   // EPI_GATE / EPI_RESGATE: a bf16 gate tensor (the stored ReLU output; pitch ldg)
  int M, N, K;
  long long lda, ldb, ldc32, ldc16, ldr, ldg;
  int relu;
  Conv cv;                    // CONV kernels: A is a channels-last map, the reduction runs over (tap, channel)
  // CUT launches (the last, under-filled round of tiles cut along K): pieces per leftover tile, scratch for a partial tile per piece
  int cut_parts;
  float* cut_ws;
};
enum { EPI_PLAIN = 0, EPI_RES = 1, EPI_GATE = 2, EPI_RESGATE = 3 };

// The source of one operand row of a stride-1 "same" convolution (gemm_internal.h ait_bf16s::Conv): GEMM row `row` is position
// (y, x) of a map, the slab lies inside window tap (dy, dx) -- scalars of the slab -- and reads the row of the neighbouring
// position, or a row of zeros where the window hangs over the map's edge.
__device__ __forceinline__ long long conv_row(const Conv& cv, int row, int dy, int dx) {
  const int p = row & cv.hw_mask, sy = (p >> cv.w_shift) + dy, sx = (p & cv.w_mask) + dx;
  return ((unsigned)sy < (unsigned)cv.H && (unsigned)sx < (unsigned)cv.W) ? (long long)(row + dy * cv.W + dx) : -1ll;
}

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
// wait until at most `n` of this wave's memory operations are outstanding (n: compile-time)
template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else static_assert(N == 0, "add the immediate");
}
// position swizzle of a K-contiguous slab row: chunk c of row r is stored at c ^ swz(r) (conflict-free ds_read_b128 of 32
// consecutive rows: 64-B rows need the row's bits 2-3, 128-B rows its bits 0-2)
template <int CH>
__device__ __forceinline__ int swz(int r) { return CH == 4 ? (r >> 2) & 3 : r & 7; }

// The epilogue of four consecutive columns of one output row in TWO steps: epi_load fetches what it adds / gates with, epi_store
// applies it (+ addend, gate, ReLU) and stores.  Two steps because the callers issue the loads of a whole group of outputs
// first and the stores behind them: written as load-use-store per output, every load waits for the store in front of it (the
// compiler cannot rule out that they alias: s_waitcnt vmcnt(0) on both sides, 64 dependent round trips per tile -- the
// 512-deep products with an addend ran at 13-16 % MfmaUtil, profiles/r06_cfg5_bf16s_mfma_util.txt).
struct EpiIn { uint2 r16; float4 r32; uint2 g16; float4 g32; };
template <int EPI>
__device__ __forceinline__ EpiIn epi_load(const Args& g, int row, int col) {
  EpiIn in;      // (only what epi_store<EPI> reads under the same conditions is loaded; no zero fill: writes into registers a load
                 // of the other branch targets would put a wait between the loads)
  if constexpr (EPI == EPI_RES || EPI == EPI_RESGATE) {
    if (g.res16) in.r16 = *reinterpret_cast<const uint2*>(g.res16 + (size_t)row * g.ldr + col);
    else in.r32 = *reinterpret_cast<const float4*>(g.residual + (size_t)row * g.ldr + col);
  }
  if constexpr (EPI == EPI_GATE || EPI == EPI_RESGATE) {
    if (EPI == EPI_RESGATE || g.gate16) in.g16 = *reinterpret_cast<const uint2*>(g.gate16 + (size_t)row * g.ldg + col);
    else in.g32 = *reinterpret_cast<const float4*>(g.residual + (size_t)row * g.ldr + col);
  }
  return in;
}
template <int EPI>
__device__ __forceinline__ void epi_store(const Args& g, int row, int col, float4 v, const EpiIn& in) {
  if constexpr (EPI == EPI_RES || EPI == EPI_RESGATE) {
    if (g.res16) {
      v.x += __uint_as_float(in.r16.x << 16); v.y += __uint_as_float(in.r16.x & 0xffff0000u);
      v.z += __uint_as_float(in.r16.y << 16); v.w += __uint_as_float(in.r16.y & 0xffff0000u);
    } else {
      v.x += in.r32.x; v.y += in.r32.y; v.z += in.r32.z; v.w += in.r32.w;
    }
  }
  if constexpr (EPI == EPI_GATE || EPI == EPI_RESGATE) {
    if (EPI == EPI_RESGATE || g.gate16) {
      // (a bf16 value is positive iff its 16 bits, read as a signed short, are > 0: +0 is 0, negatives and -0 < 0)
      v.x = (short)(in.g16.x & 0xffffu) > 0 ? v.x : 0.f; v.y = (short)(in.g16.x >> 16) > 0 ? v.y : 0.f;
      v.z = (short)(in.g16.y & 0xffffu) > 0 ? v.z : 0.f; v.w = (short)(in.g16.y >> 16) > 0 ? v.w : 0.f;
    } else {
      v.x = in.g32.x > 0.f ? v.x : 0.f; v.y = in.g32.y > 0.f ? v.y : 0.f;
      v.z = in.g32.z > 0.f ? v.z : 0.f; v.w = in.g32.w > 0.f ? v.w : 0.f;
    }
  }
  if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  if (g.C32) *reinterpret_cast<float4*>(g.C32 + (size_t)row * g.ldc32 + col) = v;
  if (g.C16) *reinterpret_cast<uint2*>(g.C16 + (size_t)row * g.ldc16 + col) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
}

// (CUT launches, below) the finishing launch: block (leftover tile lt, band): BM / cut_parts rows of the tile summed over its
// cut_parts pieces in piece order (bit-reproducible), bias, epilogue.  All loads of an element are independent: one round trip.
template <class T, int EPI>
__global__ __launch_bounds__(256) void cut_finish_kernel(const Args g, int tile0) {
  constexpr int BM = T::BM, BN = T::BN;
  const int tiles_n = g.N / BN;
  const int lt = (int)blockIdx.x / g.cut_parts, part = (int)blockIdx.x - lt * g.cut_parts;
  const int t = tile0 + lt, m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const int band_rows = BM / g.cut_parts, band0 = part * band_rows;
  const float* ws = g.cut_ws + (size_t)lt * g.cut_parts * (BM * BN);
  for (int e = threadIdx.x; e < band_rows * (BN / 4); e += 256) {
    const int r = e / (BN / 4), c4 = e - r * (BN / 4);
    const int row = m0 + band0 + r, col = n0 + 4 * c4;
    if (row >= g.M) continue;
    float4 t4[8];
#pragma unroll
    for (int p = 0; p < 8; p++)
      if (p < g.cut_parts) t4[p] = *reinterpret_cast<const float4*>(ws + ((size_t)(p * BM + band0 + r) * BN + 4 * c4));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < 8; p++)
      if (p < g.cut_parts) { v.x += t4[p].x; v.y += t4[p].y; v.z += t4[p].z; v.w += t4[p].w; }
    if (g.bias) {
      const float4 bb = *reinterpret_cast<const float4*>(g.bias + col);
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    }
    epi_store<EPI>(g, row, col, v, epi_load<EPI>(g, row, col));
  }
}

// CUT: the launch's tiles are `full` whole rounds of the grid plus L <= grid / 2 leftover tiles -- which would be a last round of
// L serial K loops with the rest of the chip idle.  Each leftover tile is cut along K into cut_parts (2 / 4 / 8) pieces, one per
// workgroup, computed FIRST: the workgroup stores its piece's partial tile (row-major, plain f32) into its slot of cut_ws and
// goes on to its whole tiles.  The pieces are summed and finished by a second, tiny launch behind this one (cut_finish_kernel):
// no flag, no counter, nobody waits inside the kernel.
template <class T, int EPI, bool CONV, bool CUT>
__global__ __launch_bounds__(T::NT, 2) void gemm_bf16s_kernel(const Args g) {      // (two waves per SIMD: <= 256 VGPRs)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NS = T::NS, ROWB = T::ROWB, CH = T::CH, RG = T::RG, STAGE = T::STAGE;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = g.N / BN, tiles_m = (g.M + BM - 1) / BM, tiles = tiles_m * tiles_n;
  const int wg_per_xcd = gridDim.x / AIT_NXCD;
  const int xcd = blockIdx.x % AIT_NXCD, j = blockIdx.x / AIT_NXCD;
  const int slabs = g.K / BK;
  int per, mine, total;
  int pc_tile = -1, pc_s0 = 0, pc_cnt = 0, pc_id = 0;      // CUT: this workgroup's piece of a leftover tile
  if constexpr (CUT) {
    const int full = (tiles / (int)gridDim.x) * (int)gridDim.x, left = tiles - full;
    per = full / AIT_NXCD;
    mine = full / (int)gridDim.x;
    pc_id = xcd * wg_per_xcd + j;                                      // (the pieces of a tile on one XCD: they share its operand rows)
    if (pc_id < left * g.cut_parts) {
      const int pc_lt = pc_id / g.cut_parts, pc_part = pc_id - pc_lt * g.cut_parts;
      pc_tile = full + pc_lt;
      pc_s0 = pc_part * slabs / g.cut_parts;
      pc_cnt = (pc_part + 1) * slabs / g.cut_parts - pc_s0;
    }
    total = mine * slabs + pc_cnt;
    if (total <= 0) return;
  } else {
    per = (tiles + AIT_NXCD - 1) / AIT_NXCD;
    const int chunk_end = min(per, tiles - xcd * per);
    mine = j < chunk_end ? (chunk_end - j + wg_per_xcd - 1) / wg_per_xcd : 0;
    if (mine <= 0) return;
    total = mine * slabs;
  }
  const int wm = (wave / T::WN) * 128, wn = (wave % T::WN) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int rr = lane / CH, pos = lane % CH;                          // row within a granule, chunk POSITION in LDS
  const int cfetch = pos ^ swz<CH>(rr);                               // the 16-B chunk of the row that lands there
  const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds;
  auto origin_of = [&](int t, int& m0, int& n0) __attribute__((always_inline)) {
    m0 = (t / tiles_n) * BM;
    n0 = (t % tiles_n) * BN;
  };
  auto tile_origin = [&](int i, int& m0, int& n0) __attribute__((always_inline)) { origin_of(xcd * per + j + i * wg_per_xcd, m0, n0); };
  // loader: granule q of a slab is RG operand rows (A's BM / RG granules first, then B's); wave w takes q = w, w + NW, ...
  auto issue = [&](int s, int stage) __attribute__((always_inline)) {
    int m0, n0;
    const int sf = CUT ? s - pc_cnt : s, unit = sf / slabs;           // (CUT: the piece's slabs come first)
    const bool piece = CUT && s < pc_cnt;
    const int ks = piece ? pc_s0 + s : sf - unit * slabs;
    origin_of(piece ? pc_tile : xcd * per + j + unit * wg_per_xcd, m0, n0);
    const int k0 = ks * BK;
    int tap_dy = 0, tap_dx = 0, tap_c0 = 0;                          // CONV: the slab's window tap and first channel
    if constexpr (CONV) {
      const int tap = k0 >> g.cv.cin_shift;
      tap_c0 = k0 & ((1 << g.cv.cin_shift) - 1);
      tap_dy = tap / g.cv.kw - g.cv.pad;
      tap_dx = tap % g.cv.kw - g.cv.pad;
    }
#pragma unroll
    for (int i = 0; i < T::LPW; i++) {
      const int q = wave + i * T::NW;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE + q * 1024);
      if (q < BM / RG) {
        int row = m0 + q * RG + rr;
        row = row < g.M ? row : g.M - 1;                             // (rows past M: any finite data, never stored)
        if constexpr (CONV) {
          const long long src = conv_row(g.cv, row, tap_dy, tap_dx);
          glds16(src >= 0 ? g.A + (size_t)src * g.lda + tap_c0 + cfetch * 8 : g.cv.zeros + cfetch * 8, dst);
        } else {
          glds16(g.A + (size_t)row * g.lda + k0 + cfetch * 8, dst);
        }
      } else {
        const int row = n0 + (q - BM / RG) * RG + rr;
        glds16(g.B + (size_t)row * g.ldb + k0 + cfetch * 8, dst);
      }
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
#pragma unroll
  for (int i = 0; i < NS - 1; i++)
    if (i < total) issue(i, i);
  int done = 0, stage = 0;
  int ti = (CUT && pc_cnt > 0) ? -1 : 0;                             // (unit -1: the piece)
  bool stores_pending = false;
  for (int s = 0; s < total; s++) {
    // slab s must have landed: everything but the loads of the NS - 2 slabs behind it (loads complete in order; the stores
    // of an epilogue may overtake loads in the counter, so after one -- and at the tail -- everything is awaited)
    if (stores_pending || s + NS - 1 > total) wait_vm<0>();
    else wait_vm<(NS - 2) * T::LPW>();
    stores_pending = false;
    __builtin_amdgcn_s_barrier();                                    // ... for every wave; and all are done with slab s - 1
    if (s + NS - 1 < total) issue(s + NS - 1, (stage + NS - 1) % NS);
    const unsigned char* sa = lds + stage * STAGE;
    const unsigned char* sb = sa + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      const int c = 2 * ks + lk;
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int row = wm + a * 32 + li;
        fa[a] = *reinterpret_cast<const bf16x8*>(sa + row * ROWB + ((c ^ swz<CH>(row)) << 4));
      }
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int row = wn + b * 32 + li;
        fb[b] = *reinterpret_cast<const bf16x8*>(sb + row * ROWB + ((c ^ swz<CH>(row)) << 4));
      }
      // operands swapped: D^T = B A^T, lane (li, lk) holds row m = li, columns 8 q + 4 lk + (r & 3), q = r >> 2
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
    ++done;
    if constexpr (CUT) {
      if (ti < 0 && done == pc_cnt) {        // the piece: store its partial tile (row-major), on to the whole tiles
        // (buffer stores: one lane offset, the 32 block offsets as scalars -- plain 64-bit addresses cost the loop registers)
        __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(g.cut_ws + (size_t)pc_id * (BM * BN), 0, BM * BN * 4, 0x00020000);
        const int voff = ((wm + li) * BN + wn + 4 * lk) * 4;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 2; b++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
              u32x4 v;
              v.x = __float_as_uint(acc[a][b][4 * q]); v.y = __float_as_uint(acc[a][b][4 * q + 1]);
              v.z = __float_as_uint(acc[a][b][4 * q + 2]); v.w = __float_as_uint(acc[a][b][4 * q + 3]);
              __builtin_amdgcn_raw_buffer_store_b128(v, ws, voff, (a * 32 * BN + b * 32 + 8 * q) * 4, 0);
#pragma unroll
              for (int r = 0; r < 4; r++) acc[a][b][4 * q + r] = 0.f;
            }
        done = 0;
        ti = 0;
        stores_pending = true;
      }
    }
    if ((!CUT || ti >= 0) && done == slabs) {
      int m0, n0;
      tile_origin(ti, m0, n0);
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int row = m0 + wm + a * 32 + li;
        const bool ok = row < g.M;
        // the eight outputs of this row block: every load first (bias, addend, gate), then the arithmetic and the stores.
        // (Measured and NOT done: the bias once per tile in front of this loop -- 32 more registers live across all four
        // blocks, no scratch reported, yet every product 1.4-1.7x slower: the allocator pays for them in the main loop;
        // block a + 1's loads in front of block a's stores -- spills as long as the addend / gate types are run-time choices.)
        float4 bb[2][4];
        EpiIn in[2][4];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int col = n0 + wn + b * 32 + 8 * q + 4 * lk;
            bb[b][q] = g.bias ? *reinterpret_cast<const float4*>(g.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) in[b][q] = epi_load<EPI>(g, row, col);
          }
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int col = n0 + wn + b * 32 + 8 * q + 4 * lk;
            float4 v = make_float4(acc[a][b][4 * q] + bb[b][q].x, acc[a][b][4 * q + 1] + bb[b][q].y, acc[a][b][4 * q + 2] + bb[b][q].z,
                                   acc[a][b][4 * q + 3] + bb[b][q].w);
            if (ok) epi_store<EPI>(g, row, col, v, in[b][q]);
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

// ---- weight gradients: C[Mo, No] (f32) += sum_r A[r, m] B[r, n], both operands bf16 [R, features] row-major -----------
// The reduction index is the OUTER dimension of both operands (token rows): a slab is BK rows x BM (A) / BN (B) feature
// columns, moved global -> LDS as it lies (LDS-DMA, 16-B chunks along the features), and the MFMA operand -- eight
// consecutive reduction indices of one feature per lane -- is gathered by ds_read_b64_tr_b16, gfx950's transposing LDS read
// (per 16 lanes a 4-row x 16-column block, delivered column-major): two reads per operand block, no register shuffles.
// 16-B chunk c of slab row k is stored at chunk position c ^ (4 (k & 3)): the four rows of a transposed read, 512 (256) B
// apart, land on four different quarter-sets of the 64 banks.  Split-K over the rows; partial tiles are added with f32
// atomics (like the f32-storage weight-gradient launches).
struct TnArgs {
  const unsigned short* A;      // [R, Mo] (pitch lda)
  const unsigned short* B;      // [R, No] (pitch ldb)
  float* C;                     // [Mo, No] (pitch ldc), accumulated
  float* partials;              // PARTIAL: [splits][Mo][No] scratch, every element written once (no atomics); reduced afterwards
  int Mo, No, R, splits;        // the R / BK slabs of the reduction are cut into `splits` ranges that differ by at most one slab
  long long lda, ldb, ldc;
  Conv cv;                      // CONV: B is a channels-last map gathered per window tap, No = taps * cin (column = (tap, channel))
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base, int off_lo, int off_hi) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off_hi));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

template <class T, bool PARTIAL, bool CONV>
__global__ __launch_bounds__(T::NT, 2) void gemm_bf16s_tn_kernel(const TnArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NS = T::NS, STAGE = T::STAGE;
  constexpr int ROWA = BM * 2, ROWB_ = BN * 2;                // bytes per slab row (512 / 256 or 512)
  constexpr int SLAB_A = BK * ROWA;
  constexpr int GA = SLAB_A / 1024;                            // granules of the A slab (the B slab's follow)
  constexpr int LPRA = ROWA / 16, LPRB = ROWB_ / 16;           // lanes (16-B chunks) per slab row
  // chunk swizzle of slab row k: position = chunk ^ (SW (k & 3)) -- 4 for rows of 16 chunks and more, 2 for the 8-chunk rows of
  // the 64-column tile (the swizzled position must stay inside the row)
  constexpr int SWA = LPRA >= 16 ? 4 : 2, SWB = LPRB >= 16 ? 4 : 2;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_n = g.No / BN, tiles = (g.Mo / BM) * tiles_n, items = tiles * g.splits;
  const int wm = (wave / T::WN) * 128, wn = (wave % T::WN) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds;
  const int slabs_all = g.R / BK, slabs_lo = slabs_all / g.splits, slabs_rem = slabs_all % g.splits;
  // this workgroup's items.  The item list is split-major (the tiles of one K-range are neighbours: they read the same token
  // rows) and every XCD owns a contiguous chunk of it, dealt round-robin to its workgroups (blocks b and b + 8 share an L2):
  // the tiles that run side by side on an XCD share their A column slices and B column slices of ONE range in that L2.  Dealt
  // over all workgroups instead (the other branch: round 5) the eight tiles of a 1024 x 512 gradient's range sit on eight
  // different XCDs and every operand slice comes from memory once per tile.
  constexpr bool kChunks = ait_lab::Knobs::tn_xcd_chunks;
  const int per = (items + AIT_NXCD - 1) / AIT_NXCD, wg_per_xcd = (int)gridDim.x / AIT_NXCD;
  const int xcd = (int)blockIdx.x % AIT_NXCD, j = (int)blockIdx.x / AIT_NXCD;
  const int chunk_end = min(per, items - xcd * per);
  const int mine = kChunks ? (j < chunk_end ? (chunk_end - j + wg_per_xcd - 1) / wg_per_xcd : 0)
                           : ((int)blockIdx.x < items ? (items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0);
  if (mine <= 0) return;
  // item i of this workgroup: its tile, its K-range (first row k0, `cnt` slabs: the first slabs_rem ranges hold one more)
  auto item_of = [&](int i, int& m0, int& n0, int& k0, int& cnt) __attribute__((always_inline)) -> int {
    const int it = kChunks ? xcd * per + j + i * wg_per_xcd : (int)blockIdx.x + i * (int)gridDim.x;
    const int split = it / tiles, t = it % tiles;
    m0 = (t / tiles_n) * BM;
    n0 = (t % tiles_n) * BN;
    k0 = (split * slabs_lo + min(split, slabs_rem)) * BK;
    cnt = slabs_lo + (split < slabs_rem ? 1 : 0);
    return split;
  };
  int total = 0;
  for (int i = 0; i < mine; i++) {
    int m0, n0, k0, cnt;
    item_of(i, m0, n0, k0, cnt);
    total += cnt;
  }
  // the loader walks the items' slabs in order: (item, slab within it) of the NEXT slab to request
  int iss_item = 0, iss_slab = 0, iss_m0, iss_n0, iss_k0, iss_cnt;
  item_of(0, iss_m0, iss_n0, iss_k0, iss_cnt);
  auto issue = [&](int stage) __attribute__((always_inline)) {
    const int m0 = iss_m0, n0 = iss_n0, k0 = iss_k0 + iss_slab * BK;
    if (++iss_slab == iss_cnt) {
      iss_slab = 0;
      if (++iss_item < mine) item_of(iss_item, iss_m0, iss_n0, iss_k0, iss_cnt);
    }
    int tap_dy = 0, tap_dx = 0, tap_c0 = 0;                    // CONV: the tile's columns lie inside one window tap
    if constexpr (CONV) {
      const int tap = n0 >> g.cv.cin_shift;
      tap_c0 = n0 & ((1 << g.cv.cin_shift) - 1);
      tap_dy = tap / g.cv.kw - g.cv.pad;
      tap_dx = tap % g.cv.kw - g.cv.pad;
    }
#pragma unroll
    for (int i = 0; i < T::LPW; i++) {
      const int q = wave + i * T::NW;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE + q * 1024);
      if (q < GA) {
        const int row = q * (64 / LPRA) + lane / LPRA, pos = lane % LPRA;
        glds16(g.A + (size_t)(k0 + row) * g.lda + m0 + (pos ^ (SWA * (row & 3))) * 8, dst);
      } else {
        const int row = (q - GA) * (64 / LPRB) + lane / LPRB, pos = lane % LPRB;
        const int chunk = pos ^ (SWB * (row & 3));
        if constexpr (CONV) {
          const long long src = conv_row(g.cv, k0 + row, tap_dy, tap_dx);
          glds16(src >= 0 ? g.B + (size_t)src * g.ldb + tap_c0 + chunk * 8 : g.cv.zeros + chunk * 8, dst);
        } else {
          glds16(g.B + (size_t)(k0 + row) * g.ldb + n0 + chunk * 8, dst);
        }
      }
    }
  };
  // transposed-read addresses of this lane (bytes from the slab's operand base), for k-step 0 / read 0; k-step ks adds
  // 16 ks rows, the second read 4 rows.  Group gq = lane >> 4: features +16 (gq & 1), reduction rows +8 (gq >> 1).
  const int gq = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int krow = 8 * (gq >> 1) + tq;                       // (k & 3) == tq for every read: block rows start at multiples of 4
  int offA[4], offB[2];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int c = (wm + 32 * a) / 8 + 2 * (gq & 1) + (tp >> 1);
    offA[a] = krow * ROWA + ((c ^ (SWA * tq)) << 4) + 8 * (tp & 1);
  }
#pragma unroll
  for (int b = 0; b < 2; b++) {
    const int c = (wn + 32 * b) / 8 + 2 * (gq & 1) + (tp >> 1);
    offB[b] = krow * ROWB_ + ((c ^ (SWB * tq)) << 4) + 8 * (tp & 1);
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
#pragma unroll
  for (int i = 0; i < NS - 1; i++)
    if (i < total) issue(i);
  int done = 0, ti = 0, stage = 0;
  int cur_m0, cur_n0, cur_k0, cur_cnt;
  int cur_split = item_of(0, cur_m0, cur_n0, cur_k0, cur_cnt);
  for (int s = 0; s < total; s++) {
    if (s + NS - 1 > total) wait_vm<0>();
    else wait_vm<(NS - 2) * T::LPW>();
    __builtin_amdgcn_s_barrier();
    if (s + NS - 1 < total) issue((stage + NS - 1) % NS);
    const unsigned char* sa = lds + stage * STAGE;
    const unsigned char* sb = sa + SLAB_A;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int a = 0; a < 4; a++) fa[a] = tr_frag(sa, offA[a] + 16 * ks * ROWA, offA[a] + (16 * ks + 4) * ROWA);
#pragma unroll
      for (int b = 0; b < 2; b++) fb[b] = tr_frag(sb, offB[b] + 16 * ks * ROWB_, offB[b] + (16 * ks + 4) * ROWB_);
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[a][b], 0, 0, 0);
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
    if (++done == cur_cnt) {
      const int m0 = cur_m0, n0 = cur_n0;
#pragma unroll
      for (int a = 0; a < 4; a++) {
        const int row = m0 + wm + a * 32 + li;
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int col = n0 + wn + b * 32 + 8 * q + 4 * lk;
            if constexpr (PARTIAL) {      // this K-range's tile, stored once (16-B stores); tn_reduce_kernel adds the ranges
              float* dst = g.partials + ((size_t)cur_split * g.Mo + row) * g.No + col;
              *reinterpret_cast<float4*>(dst) =
                  make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
            } else {
              float* dst = g.C + (size_t)row * g.ldc + col;
#pragma unroll
              for (int r = 0; r < 4; r++) unsafeAtomicAdd(dst + r, acc[a][b][4 * q + r]);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) acc[a][b][4 * q + r] = 0.f;
          }
      }
      done = 0;
      if (++ti < mine) cur_split = item_of(ti, cur_m0, cur_n0, cur_k0, cur_cnt);
      wait_vm<0>();       // (the atomics count like stores and may overtake the loads already requested: wait for everything once)
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
      const unsigned v = pack2(tile[r][c], tile[r + 1][c]);
      unsigned short* d = dst + (long long)(c0 + c) * ld_dst + r0 + r;
      if (r0 + r + 1 < rows && !(ld_dst & 1)) *reinterpret_cast<unsigned*>(d) = v;      // (even pitch: 4-byte aligned pairs)
      else {
        d[0] = (unsigned short)(v & 0xffffu);
        if (r0 + r + 1 < rows) d[1] = (unsigned short)(v >> 16);
      }
    }
  }
}

// out[c] += sum_r x[r, c] over a bf16 [rows, cols] matrix (the bias gradient of a layer whose output gradient is stored
// in bf16).  A 1024-thread block sums a band of rows: thread (tx, ty) takes the V columns of group tx (V = 8: 16-B loads,
// else 4: 8-B) on rows ty, ty + RY, ... of the band; the RY partial rows meet in LDS and row 0 adds the band's sums to `out`
// with one f32 atomic per column.  Few, tall bands (~512 blocks): the atomics of a column all hit one address, and a
// thousand of them in a row cost more than the read (0.29 ms for 0.54 GB with 128-row bands).
template <int V>
__global__ __launch_bounds__(1024) void colsum_bf16_kernel(const unsigned short* __restrict__ x, long long rows, int cols,
                                                           long long ld, int band, int G, float* __restrict__ out) {
  extern __shared__ float part[];      // [RY][G * V]
  const int RY = blockDim.x / G, tx = threadIdx.x % G, ty = threadIdx.x / G;
  const int cv = tx + blockIdx.y * G;
  const long long r0 = (long long)blockIdx.x * band;
  const long long r1 = r0 + band < rows ? r0 + band : rows;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; j++) acc[j] = 0.f;
  if (cv < cols / V) {
    const unsigned short* __restrict__ p = x + (r0 + ty) * ld + V * cv;
    const long long step = (long long)RY * ld;
#pragma unroll 8
    for (long long r = r0 + ty; r < r1; r += RY, p += step) {
      unsigned w[V / 2];
      if constexpr (V == 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      } else {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        w[0] = v.x; w[1] = v.y;
      }
#pragma unroll
      for (int j = 0; j < V / 2; j++) {
        acc[2 * j] += __uint_as_float(w[j] << 16);
        acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; j++) part[(ty * G + tx) * V + j] = acc[j];
  __syncthreads();
  // G * V columns of the block, summed over the RY partial rows by the first G * V threads
  for (int c = threadIdx.x; c < G * V; c += blockDim.x) {
    const int col = blockIdx.y * G * V + c;
    if (col >= cols) continue;
    float s = 0.f;
    for (int y = 0; y < RY; y++) s += part[y * G * V + c];
    unsafeAtomicAdd(out + col, s);
  }
}

// resident workgroup slots of a kernel (a constant of the code object and the device model); also raises its LDS limit
template <class T>
inline int slots_of(const void* kern, int& memo) {
  if (memo > 0) return memo;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS) != hipSuccess) return -1;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T::NT, T::LDS) != hipSuccess || per_cu <= 0) {
    (void)hipGetLastError();
    per_cu = T::LDS > 80 * 1024 ? 1 : 2;
  }
  memo = (per_cu > 2 ? 2 : per_cu) * cus;
  return memo;
}

// The last round's cut (CUT kernels + cut_finish_kernel; the 256 x 256 tile only -- one workgroup per CU): when the tiles are
// whole rounds of the grid plus at most half a round, and the caller's scheduler scratch (ait_launch_ctx::sched_ws, the one the
// f32 kernel's stream-K uses: launches ordered on one stream) holds a partial tile per piece.  4096 x 16 + 8 x 16 rows of cfg5's
// layer4 are 256.5 row tiles: without the cut every 512-column product runs a third round for two tiles.
template <class T, int EPI, bool CONV>
int launch(Args g, const ait_launch_ctx* ctx, hipStream_t s) {
  const void* kern = reinterpret_cast<const void*>(gemm_bf16s_kernel<T, EPI, CONV, false>);
  static int memo = 0;
  const int slots = slots_of<T>(kern, memo);
  if (slots <= 0) return AIT_ELAUNCH;
  const int tiles = ((g.M + T::BM - 1) / T::BM) * (g.N / T::BN);
  if constexpr (T::BN == 256) {
    // (long reductions only: at 72 slabs -- layer4's 3x3 convolutions -- the cut turns 453 us into 370, two whole rounds take 374;
    // at 32 slabs it wins 7 us with two leftover tiles and loses 47 with 32, at 8 slabs it loses: the pieces and the finishing
    // launch cost about what a lone short tile does)
    if (ait_lab::Knobs::bf16s_cut && ctx && ctx->sched_ws && (slots % AIT_NXCD) == 0 && tiles > slots && g.K / T::BK >= 64) {
      const int left = tiles % slots, slabs = g.K / T::BK;
      // pieces per leftover tile: 8 / 4 / 2 (bands of whole rows), at least two slabs each, no more pieces than workgroups
      int parts = 8;
      while (parts > 1 && (left <= 0 || parts * left > slots || slabs / parts < 2)) parts >>= 1;
      const size_t need = ait_ws::kCtlBytes + (size_t)slots * T::BM * T::BN * sizeof(float);
      if (parts >= 2 && left > 0 && left <= 256 && ctx->sched_ws_bytes >= need) {
        const void* kcut = reinterpret_cast<const void*>(gemm_bf16s_kernel<T, EPI, CONV, true>);
        static int memo_cut = 0;
        if (slots_of<T>(kcut, memo_cut) == slots) {
          g.cut_parts = parts;
          g.cut_ws = reinterpret_cast<float*>(static_cast<char*>(ctx->sched_ws) + ait_ws::kCtlBytes);
          hipLaunchKernelGGL((gemm_bf16s_kernel<T, EPI, CONV, true>), dim3(slots), dim3(T::NT), T::LDS, s, g);
          AIT_CHECK_LAUNCH();
          hipLaunchKernelGGL((cut_finish_kernel<T, EPI>), dim3(left * parts), dim3(256), 0, s, g, tiles - left);
          AIT_CHECK_LAUNCH();
          return AIT_OK;
        }
      }
    }
  }
  const int per = (tiles + AIT_NXCD - 1) / AIT_NXCD;
  int w = slots / AIT_NXCD;
  if (w > per) w = per;
  if (w < 1) w = 1;
  hipLaunchKernelGGL((gemm_bf16s_kernel<T, EPI, CONV, false>), dim3(w * AIT_NXCD), dim3(T::NT), T::LDS, s, g);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}
template <class T, bool CONV>
int launch_epi(const Args& g, bool gate, const ait_launch_ctx* ctx, hipStream_t s) {
  const bool res = g.residual || g.res16;
  if (gate) return res && g.gate16 ? launch<T, EPI_RESGATE, CONV>(g, ctx, s) : launch<T, EPI_GATE, CONV>(g, ctx, s);
  if (res) return launch<T, EPI_RES, CONV>(g, ctx, s);
  return launch<T, EPI_PLAIN, CONV>(g, ctx, s);
}
// C[m, n] += sum over the K-ranges of partials[s][m][n]  (16-B loads; in range order: reproducible)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ partials, int splits, int Mo, int No,
                                                        float* __restrict__ C, long long ldc) {
  const long long n4 = (long long)Mo * (No / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const long long m = i / (No / 4);
    const int c = (int)(i - m * (No / 4)) * 4;
    float4* dst = reinterpret_cast<float4*>(C + m * ldc + c);
    float4 acc = *dst;
    for (int sp = 0; sp < splits; sp++) {
      const float4 v = *reinterpret_cast<const float4*>(partials + ((size_t)sp * Mo + m) * No + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *dst = acc;
  }
}

template <class T, bool PARTIAL, bool CONV>
int launch_tn(const TnArgs& g, hipStream_t s) {
  const void* kern = reinterpret_cast<const void*>(gemm_bf16s_tn_kernel<T, PARTIAL, CONV>);
  static int memo = 0;
  const int slots = slots_of<T>(kern, memo);
  if (slots <= 0) return AIT_ELAUNCH;
  const long long items = (long long)(g.Mo / T::BM) * (g.No / T::BN) * g.splits;
  int grid = (int)(items < slots ? items : slots);
  if (ait_lab::Knobs::tn_xcd_chunks) {      // whole XCD sets of workgroups, no more per XCD than its chunk of the items
    const long long per = (items + AIT_NXCD - 1) / AIT_NXCD;
    long long w = slots / AIT_NXCD;
    if (w > per) w = per;
    if (w < 1) w = 1;
    grid = (int)w * AIT_NXCD;
  }
  hipLaunchKernelGGL((gemm_bf16s_tn_kernel<T, PARTIAL, CONV>), dim3(grid), dim3(T::NT), T::LDS, s, g);
  AIT_CHECK_LAUNCH();
  if (PARTIAL) {
    const long long want = ((long long)g.Mo * (g.No / 4) + 255) / 256;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(want > 4096 ? 4096 : want)), dim3(256), 0, s, g.partials, g.splits, g.Mo,
                       g.No, g.C, g.ldc);
    AIT_CHECK_LAUNCH();
  }
  return AIT_OK;
}

}  // namespace

constexpr bool kUseBig = !ait_lab::Knobs::bf16s_small_only;

namespace {
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

// the window geometry the CONV kernels serve, from the public description (include/ait_hip.h ait_conv_geom)
int make_conv(const ait_conv_geom* q, int cin, const void* zeros, size_t zeros_bytes, Conv& cv, long long& rows, int& taps) {
  if (!q || !zeros) return AIT_EINVAL;
  if (q->n < 0 || q->in_h <= 0 || q->in_w <= 0 || q->kh <= 0 || q->kw <= 0 || cin <= 0) return AIT_EINVAL;
  const int hw = q->in_h * q->in_w;
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  if (q->groups > 1 || q->stride != 1 || q->out_h != q->in_h || q->out_w != q->in_w || q->kh != q->kw || !(q->kh & 1) ||
      q->pad != q->kh / 2 || !pow2(q->in_w) || !pow2(hw) || !pow2(cin) || cin < 64 || zeros_bytes < 512 || !al16(zeros))
    return AIT_EUNSUPPORTED;
  cv.on = 1;
  cv.hw_mask = hw - 1;
  cv.w_shift = __builtin_ctz((unsigned)q->in_w);
  cv.w_mask = q->in_w - 1;
  cv.H = q->in_h; cv.W = q->in_w; cv.kw = q->kw; cv.pad = q->pad;
  cv.cin_shift = __builtin_ctz((unsigned)cin);
  cv.zeros = static_cast<const unsigned short*>(zeros);
  rows = (long long)q->n * hw;
  taps = q->kh * q->kw;
  return AIT_OK;
}
}  // namespace

int ait_bf16s::gemm(const Gemm& p, const ait_launch_ctx* ctx, void* stream) {
  const int M = p.M, N = p.N, K = p.K;
  if (M < 0 || N < 0 || K < 0) return AIT_EINVAL;
  if (M == 0 || N == 0) return AIT_OK;
  if (!p.A || !p.B || (!p.C32 && !p.C16)) return AIT_EINVAL;
  const long long lda_min = p.cv.on ? (1ll << p.cv.cin_shift) : K;
  if (K == 0 || (K % Small::BK) || (N % Small::BN) || (p.lda % 8) || (p.ldb % 8) || p.lda < lda_min || p.ldb < K || !al16(p.A) ||
      !al16(p.B) || (p.bias && !al16(p.bias)))
    return AIT_EUNSUPPORTED;
  if ((p.C32 && ((p.ldc32 % 4) || p.ldc32 < N || !al16(p.C32))) || (p.C16 && ((p.ldc16 % 4) || p.ldc16 < N || !al8(p.C16))))
    return AIT_EUNSUPPORTED;
  if (p.gate && !p.res32 && !p.gate16) return AIT_EINVAL;
  if ((!p.gate && p.gate16) || (p.res32 && p.res16)) return AIT_EINVAL;
  if (p.gate && p.res16 && !p.gate16) return AIT_EINVAL;      // (an f32 tensor alone IS the gate; an addend and a gate need gate16)
  if ((p.res32 || p.res16) && ((p.ldr % 4) || p.ldr < N || (p.res32 ? !al16(p.res32) : !al8(p.res16)))) return AIT_EUNSUPPORTED;
  if (p.gate16 && ((p.ldg % 4) || p.ldg < N || !al8(p.gate16))) return AIT_EUNSUPPORTED;
  Args g;
  g.A = static_cast<const unsigned short*>(p.A); g.B = static_cast<const unsigned short*>(p.B);
  g.C32 = p.C32; g.C16 = static_cast<unsigned short*>(p.C16);
  g.bias = p.bias; g.residual = p.res32; g.res16 = static_cast<const unsigned short*>(p.res16);
  g.gate16 = static_cast<const unsigned short*>(p.gate16);
  g.M = M; g.N = N; g.K = K; g.lda = p.lda; g.ldb = p.ldb; g.ldc32 = p.ldc32; g.ldc16 = p.ldc16; g.ldr = p.ldr;
  g.ldg = p.gate16 ? p.ldg : p.ldr;
  g.relu = p.relu ? 1 : 0;
  g.cv = p.cv;
  g.cut_parts = 0; g.cut_ws = nullptr;
  hipStream_t s = ait_stream(stream);
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, 2.0 * M * N * K, s, M, N, K, 0, 1, 1);
  // long reductions on the 256 x 256 x 64 tile (at least a round of them), the 512-deep products on the 256 x 128 x 32 one
  const long long big_tiles = (long long)((M + Big::BM - 1) / Big::BM) * (N / Big::BN);
  const bool big = kUseBig && K >= ait_lab::Knobs::bf16s_big_min_k && (K % Big::BK) == 0 && (N % Big::BN) == 0 && big_tiles >= 192;
  if (p.cv.on) return big ? launch_epi<Big, true>(g, p.gate, ctx, s) : launch_epi<Small, true>(g, p.gate, ctx, s);
  return big ? launch_epi<Big, false>(g, p.gate, ctx, s) : launch_epi<Small, false>(g, p.gate, ctx, s);
}

AIT_API int ait_gemm_bf16s(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, float* C32,
                           long long ldc32, void* C16, long long ldc16, const float* bias, const float* residual,
                           const void* gate16, long long ldr, int flags, const ait_launch_ctx* ctx, void* stream) {
  if (flags & ~(AIT_GEMM_RELU | AIT_GEMM_MASK_POS)) return AIT_EUNSUPPORTED;
  ait_bf16s::Gemm p{};
  p.A = A; p.B = B; p.C32 = C32; p.C16 = C16; p.bias = bias;
  p.gate = (flags & AIT_GEMM_MASK_POS) != 0;
  // (this entry's gate is ONE tensor: gate16 if given, else `residual`; an addend and a gate together are library-internal)
  if (p.gate && gate16) p.gate16 = gate16; else p.res32 = residual;
  if (!p.gate && gate16) return AIT_EINVAL;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc32 = ldc32; p.ldc16 = ldc16; p.ldr = ldr; p.ldg = ldr;
  p.relu = (flags & AIT_GEMM_RELU) != 0;
  return ait_bf16s::gemm(p, ctx, stream);
}

int ait_bf16s::wgrad(const Wgrad& p, const ait_launch_ctx* ctx, void* stream) {
  const int Mo = p.Mo, No = p.No, R = p.R;
  if (Mo < 0 || No < 0 || R < 0) return AIT_EINVAL;
  if (Mo == 0 || No == 0 || R == 0) return AIT_OK;
  if (!p.A || !p.B || !p.C) return AIT_EINVAL;
  const long long ldb_min = p.cv.on ? (1ll << p.cv.cin_shift) : No;
  const bool narrow = No == Narrow::BN && !p.cv.on;              // one 64-column tile wide: d fc_w of the attention blocks
  if ((Mo % Small::BM) || ((No % Small::BN) && !narrow) || (p.lda % 8) || (p.ldb % 8) || p.lda < Mo || p.ldb < ldb_min || (p.ldc % 4) ||
      p.ldc < No || !al16(p.A) || !al16(p.B) || !al16(p.C))
    return AIT_EUNSUPPORTED;
  int split_k = p.split_k < 1 ? 1 : p.split_k;
  // whole 32-row slabs, at least one per range; the ranges need not be equal (they differ by at most one slab)
  if ((R % Small::BK) || R / Small::BK < split_k) return AIT_EUNSUPPORTED;
  TnArgs g;
  g.A = static_cast<const unsigned short*>(p.A); g.B = static_cast<const unsigned short*>(p.B); g.C = p.C;
  g.Mo = Mo; g.No = No; g.R = R; g.splits = split_k;
  g.lda = p.lda; g.ldb = p.ldb; g.ldc = p.ldc;
  g.cv = p.cv;
  // with scratch for one partial tile set per K-range the ranges are stored once and added by a second small launch
  // (in range order: reproducible); without it they are added to C with f32 atomics -- a third of the product's time at 16
  // ranges (profiles/r05_bf16_storage_ffn.txt)
  const size_t need = (size_t)split_k * Mo * No * sizeof(float);
  const bool use_partials = !ait_lab::Knobs::tn_atomics && split_k > 1 && p.partials && p.partials_bytes >= need && al16(p.partials);
  g.partials = use_partials ? static_cast<float*>(p.partials) : nullptr;
  hipStream_t s = ait_stream(stream);
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, 2.0 * Mo * No * R, s, Mo, No, R, 1, 0, split_k);
  // (CONV: a tile's columns lie inside one tap -- cin is a power of two >= 64, so 128- and 256-wide tiles need cin >= that)
  const bool fits_big = !p.cv.on || (1 << p.cv.cin_shift) >= Big::BN;
  if (p.cv.on && (1 << p.cv.cin_shift) < Small::BN) return AIT_EUNSUPPORTED;
  const bool big = kUseBig && fits_big && (No % Big::BN) == 0 && (R % Big::BK) == 0 && R / Big::BK >= split_k &&
                   (long long)(Mo / Big::BM) * (No / Big::BN) * split_k >= 192;
  if (narrow) return use_partials ? launch_tn<Narrow, true, false>(g, s) : launch_tn<Narrow, false, false>(g, s);
  if (p.cv.on) {
    if (use_partials) return big ? launch_tn<Big, true, true>(g, s) : launch_tn<Small, true, true>(g, s);
    return big ? launch_tn<Big, false, true>(g, s) : launch_tn<Small, false, true>(g, s);
  }
  if (use_partials) return big ? launch_tn<Big, true, false>(g, s) : launch_tn<Small, true, false>(g, s);
  return big ? launch_tn<Big, false, false>(g, s) : launch_tn<Small, false, false>(g, s);
}

AIT_API int ait_gemm_bf16s_tn(int Mo, int No, int R, const void* A, long long lda, const void* B, long long ldb, float* C,
                              long long ldc, int split_k, void* partials, size_t partials_bytes, const ait_launch_ctx* ctx,
                              void* stream) {
  ait_bf16s::Wgrad p{};
  p.A = A; p.B = B; p.C = C; p.Mo = Mo; p.No = No; p.R = R; p.split_k = split_k; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.partials = partials; p.partials_bytes = partials_bytes;
  return ait_bf16s::wgrad(p, ctx, stream);
}

// ---- convolutions over bf16 channels-last maps (include/ait_hip.h "bf16-storage convolutions") --------------------------------
AIT_API int ait_conv_fwd_bf16s(const void* x, long long ldx, const void* w, const ait_conv_geom* geom, int cin, int cout,
                               const float* bias, const void* res16, const void* gate16, long long ldr, int flags, float* y32,
                               long long ldy32, void* y16, long long ldy16, const void* zeros, size_t zeros_bytes,
                               const ait_launch_ctx* ctx, void* stream) {
  if (flags & ~(AIT_GEMM_RELU | AIT_GEMM_MASK_POS)) return AIT_EUNSUPPORTED;
  ait_bf16s::Gemm p{};
  long long rows = 0;
  int taps = 0;
  AIT_TRY_RC(make_conv(geom, cin, zeros, zeros_bytes, p.cv, rows, taps));
  if (rows > 0x7fffffffLL) return AIT_EUNSUPPORTED;
  p.A = x; p.B = w; p.C32 = y32; p.C16 = y16; p.bias = bias; p.res16 = res16;
  p.gate = (flags & AIT_GEMM_MASK_POS) != 0;
  p.gate16 = gate16;
  p.M = (int)rows; p.N = cout; p.K = taps * cin;
  p.lda = ldx; p.ldb = (long long)taps * cin; p.ldc32 = ldy32; p.ldc16 = ldy16; p.ldr = ldr; p.ldg = ldr;
  p.relu = (flags & AIT_GEMM_RELU) != 0;
  return ait_bf16s::gemm(p, ctx, stream);
}

AIT_API int ait_conv_bwd_weight_bf16s(const void* dy, long long lddy, const void* x, long long ldx, const ait_conv_geom* geom,
                                      int cin, int cout, float* dw, int split_k, const void* zeros, size_t zeros_bytes,
                                      void* partials, size_t partials_bytes, const ait_launch_ctx* ctx, void* stream) {
  ait_bf16s::Wgrad p{};
  long long rows = 0;
  int taps = 0;
  AIT_TRY_RC(make_conv(geom, cin, zeros, zeros_bytes, p.cv, rows, taps));
  if (rows > 0x7fffffffLL) return AIT_EUNSUPPORTED;
  p.A = dy; p.B = x; p.C = dw; p.Mo = cout; p.No = taps * cin; p.R = (int)rows; p.split_k = split_k;
  p.lda = lddy; p.ldb = ldx; p.ldc = (long long)taps * cin;
  p.partials = partials; p.partials_bytes = partials_bytes;
  return ait_bf16s::wgrad(p, ctx, stream);
}

// ---- weights: f32 [n_out][taps][cin] (x scale[n_out]) -> bf16, as they lie (the forward's B operand) and / or as the
// data gradient's B operand [cin][taps][n_out] with the window flipped (tap t of the gradient = tap taps - 1 - t of the forward)
namespace {
struct WJob { const float* src; const float* scale; unsigned short* dst; int n_out, taps, cin, flipped; };
struct WJobs { WJob j[24]; int n; };
__global__ __launch_bounds__(256) void fold_bf16_kernel(const WJobs jobs) {
  __shared__ float tile[64][65];
  const WJob j = jobs.j[blockIdx.y];
  const long long cols = (long long)j.taps * j.cin;
  if (!j.flipped) {
    const long long n4 = (long long)j.n_out * (cols / 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      const float sc = j.scale ? j.scale[i / (cols / 4)] : 1.f;
      const float4 v = reinterpret_cast<const float4*>(j.src)[i];
      reinterpret_cast<uint2*>(j.dst)[i] = make_uint2(pack2(v.x * sc, v.y * sc), pack2(v.z * sc, v.w * sc));
    }
    return;
  }
  const int ot = j.n_out / 64, ct = j.cin / 64, tiles = j.taps * ot * ct;
  const long long ld_dst = (long long)j.taps * j.n_out;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int tap = t / (ot * ct), rem = t - tap * (ot * ct), o0 = (rem / ct) * 64, c0 = (rem % ct) * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      tile[r][c] = j.src[(long long)(o0 + r) * cols + (long long)tap * j.cin + c0 + c] * (j.scale ? j.scale[o0 + r] : 1.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
      const int c = i >> 5, r = (i & 31) * 2;
      *reinterpret_cast<unsigned*>(j.dst + (long long)(c0 + c) * ld_dst + (long long)(j.taps - 1 - tap) * j.n_out + o0 + r) =
          pack2(tile[r][c], tile[r + 1][c]);
    }
    __syncthreads();
  }
}
}  // namespace

int ait_bf16s::fold_weights(const WeightJob* jobs, int n, void* stream) {
  if (n < 0 || n > 24) return AIT_EINVAL;
  if (n == 0) return AIT_OK;
  WJobs b;
  b.n = n;
  for (int i = 0; i < n; i++) {
    const WeightJob& w = jobs[i];
    if (!w.src || !w.dst || w.n_out <= 0 || w.taps <= 0 || w.cin <= 0) return AIT_EINVAL;
    if ((w.n_out % 64) || (w.cin % 64) || !al16(w.src) || !al16(w.dst)) return AIT_EUNSUPPORTED;
    b.j[i] = WJob{w.src, w.scale, static_cast<unsigned short*>(w.dst), w.n_out, w.taps, w.cin, w.flipped};
  }
  hipLaunchKernelGGL(fold_bf16_kernel, dim3(256, (unsigned)n), dim3(256), 0, ait_stream(stream), b);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_conv_weight_to_bf16(const float* w, const float* row_scale, int cout, int taps, int cin, void* w16,
                                    void* w16_dgrad, void* stream) {
  ait_bf16s::WeightJob jobs[2];
  int n = 0;
  if (w16) jobs[n++] = ait_bf16s::WeightJob{w, row_scale, w16, cout, taps, cin, 0};
  if (w16_dgrad) jobs[n++] = ait_bf16s::WeightJob{w, row_scale, w16_dgrad, cout, taps, cin, 1};
  return ait_bf16s::fold_weights(jobs, n, stream);
}

AIT_API int ait_colsum_bf16(const void* x, long long rows, int cols, long long ld, float* out, void* stream) {
  if (rows < 0 || cols < 0 || (cols & 3) || ld < cols || (ld & 3)) return AIT_EINVAL;
  if (rows == 0 || cols == 0) return AIT_OK;
  if (!x || !out || (reinterpret_cast<uintptr_t>(x) & 7)) return AIT_EINVAL;
  const bool wide = (cols % 8) == 0 && (ld % 8) == 0 && !(reinterpret_cast<uintptr_t>(x) & 15);
  const int V = wide ? 8 : 4, groups = cols / V;
  int G = 1;
  while (G < groups && G < 256) G *= 2;                 // column groups per block: a power of two <= 256 (divides 1024)
  const unsigned gy = (unsigned)((groups + G - 1) / G);
  const int RY = 1024 / G;
  long long want_blocks = 512 / gy > 0 ? 512 / gy : 1;
  long long band = (rows + want_blocks - 1) / want_blocks;
  const long long unit = (long long)RY * 8;             // whole unrolled passes of the block's RY row lanes
  band = (band + unit - 1) / unit * unit;
  const long long bx = (rows + band - 1) / band;
  if (band > 0x7fffffffLL || bx > 0x7fffffffLL) return AIT_EUNSUPPORTED;
  const dim3 grid((unsigned)bx, gy);
  const size_t lds = (size_t)1024 * V * sizeof(float);
  if (wide)
    hipLaunchKernelGGL(colsum_bf16_kernel<8>, grid, dim3(1024), lds, ait_stream(stream), static_cast<const unsigned short*>(x), rows,
                       cols, ld, (int)band, G, out);
  else
    hipLaunchKernelGGL(colsum_bf16_kernel<4>, grid, dim3(1024), lds, ait_stream(stream), static_cast<const unsigned short*>(x), rows,
                       cols, ld, (int)band, G, out);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_f32_to_bf16(const float* src, long long rows, int cols, long long ld_src, void* dst, long long ld_dst,
                            int transpose, void* stream) {
  if (rows < 0 || cols < 0) return AIT_EINVAL;
  if (rows == 0 || cols == 0) return AIT_OK;
  if (!src || !dst || ld_src < cols) return AIT_EINVAL;
  hipStream_t s = ait_stream(stream);
  if (transpose) {
    if (ld_dst < rows || (reinterpret_cast<uintptr_t>(dst) & 3)) return AIT_EUNSUPPORTED;
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
