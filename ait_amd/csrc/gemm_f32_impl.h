// ait_amd/csrc/gemm_f32_impl.h -- the fp32 MFMA GEMM kernels, templated on their tile configuration.
// See gemm_f32.hip for the design notes.  Included by gemm_f32.hip (product instantiations) and by
// scripts/gemm_lab.hip (the measuring harness, which instantiates the same kernels with a probe).
//
// Three main loops, selected by Cfg::MODE:
//   MODE_DB    register-staged double buffer (any K, ragged edges): small / narrow problems
//   MODE_RING  register-staged three-slab ring: the 256x128 tile when K % 16 != 0
//   MODE_DLDS  slabs moved global -> LDS directly into a three-slab ring (K % 16 == 0).  This one is
//              a PERSISTENT kernel: a workgroup walks a list of output tiles and the slab ring never
//              drains -- the first slabs of the next tile are requested while the last slabs of the
//              current one are multiplied, so a tile's prologue latency and its epilogue stores hide
//              under matrix work (gemm_f32_stream_kernel).
// (The tuner-only ablation paths of round 1 -- no-load / no-store / no-barrier variants, the
// relaxed-wait and delayed-workgroup experiments, 32-float two-stage slabs -- were removed from this
// file; they live in the history at 81edb57 and their results in DESIGN.md section 3.1.)
#pragma once
#include "common.h"
#include "gemm_internal.h"
#include "split_planes.h"
#include <atomic>

namespace ait_gemm {


// Implicit-GEMM view of a convolution over channels-last maps (gemm_f32_stream_kernel, CONV != 0).
// The GEMM's row index r runs over the positions (image, y, x) of the "row map" (rows_h x rows_w, both
// powers of two); tap t = (ty, tx) of a kh x kw window pairs row position (y, x) with position
//     (y * a + c + ty * b,  x * a + c + tx * b) / 2^div_shift        (exact division, inside src_h x src_w)
// of the gathered tensor, or with nothing (a row of zeros):
//     forward / weight gradient: rows = output map, gathered = input:  a = stride, b = 1,  c = -pad, div 1
//     data gradient:             rows = input map,  gathered = dy:     a = 1,      b = -1, c = +pad, div = stride
//   CONV_A (1): A[r, t*seg + ch] = src[gather(r, t), ch]; B is the weight [cout][kh*kw][seg] read either
//               K-contiguous (forward: B[n = cout][k]) or K-outer (data gradient: row k = (t, cout) of tap t
//               at B + t*b_tap_stride + cout*ldb);
//   CONV_B (2): weight gradient, C[cout, t*seg + ch] += sum_r A[r, cout] * src[gather(r, t), ch]: the K-outer
//               operand B is gathered row by row, the tap is fixed per output column tile (seg % BN == 0).
struct ConvGeom {
  int rows_hw_shift, rows_w_shift;      // log2(rows_h * rows_w), log2(rows_w); rows_hw_shift < 0: maps of any size, below
  int rows_hw, rows_w;                  // rows_h * rows_w, rows_w (general maps: rows < 2^24)
  float inv_hw, inv_w;                  // their reciprocals (a quotient estimate, corrected exactly)
  int n_rows;                           // GEMM rows that exist (CONV_B: reduction rows beyond it read the row of zeros)
  int src_h, src_w;
  int kw;                               // taps per window row
  int a, b, c, div_shift;
  int seg;                              // reduction floats per tap (channels of the gathered tensor)
  long long b_tap_stride;               // CONV_A with K-outer B
  const float* zero;                    // >= max(seg, BN) + 16 floats of zeros
  // grouped convolutions (0 = dense): the tile's columns (CONV_A) or rows (CONV_B) select the group, n_group
  // of them per group (a tile never straddles two), and the gathered operand's channels of that group start
  // a_group floats further per group
  int a_group, n_group;
  // ---- POSITION-MAJOR rows (pm_maps > 0; csrc/tail.hip's layer4, maps of 2^rows_hw_shift positions): GEMM row
  // r = position * pm_maps + map instead of map * positions + position, in the gathered tensor as well.  The rows of ONE
  // window position are then a contiguous block of pm_maps rows: a tap that falls outside the map for a position is
  // outside for a whole block (the weight gradient skips such blocks, below).
  int pm_maps;
  float inv_pm;
  // ... and the WEIGHT GRADIENT over position blocks (pm_wgrad != 0, CONV_B, stride 1): the reduction of a column tile (its
  // tap fixed) runs over the rows of ONE position at a time, and only over positions the tap reaches -- the work items are
  // the valid (position, tap) pairs x the tiles of a tap, every one pm_maps rows long (WorkMap below); 100 of the 144
  // pairs of a 3x3 window on 4 x 4 maps.  pm_kh = window rows.
  int pm_wgrad, pm_kh;
  // ... and the FORWARD / DATA GRADIENT with the out-of-map taps left out (pm_skip != 0, CONV_A, stride 1, b = +-1): a row
  // tile lies inside ONE position's block of rows (the last tile of a block is ragged: m0 = position * pm_maps + t * BM, rows
  // past the block masked), its reduction runs over the ntaps(position) taps that reach the map, in window order -- 4, 6 or
  // 9 of a 3x3 window's taps on 4 x 4 maps.  Items therefore differ in length; the stream-K cut handles an XCD's chunk as up
  // to two groups of equal-length items (kernel).  pm_order: the positions in launch order, a nibble each -- an order in which
  // every XCD's share of the tile list carries about the same number of taps.
  // A launch covers the row tiles pm_t0 .. pm_t0 + pm_tcnt - 1 of EVERY position's block (all of them, or -- more maps than
  // an XCD's workgroups can cut in one go -- a slice per launch, csrc/conv_f32.hip).
  int pm_skip;
  int pm_t0, pm_tcnt;
  unsigned long long pm_order;
  // ---- the data gradient of a STRIDE-2 convolution by parity class (py, px) of the input positions (ROWMAP kernels):
  // a class holds the positions (2ya + py, 2xa + px); only the window taps ty = (py + pad) mod 2 (+ 2 ...) reach it
  // (1, 2, 2 or 4 of a 3x3 window's 9), the source position of class tap (t'y, t'x) is (ya + cy - t'y, xa + cx - t'x),
  // its weights are those of window tap (wt_y0 + 2 t'y, wt_x0 + 2 t'x), and the result lands on row
  // (img << out_img_shift) + (ya << out_y_shift) + (xa << 1) + out_base of dx.  Nothing multiplies a structural zero.
  // The GEMM rows interleave the classes TILE by tile (row tile tm holds class tm & 3, its rows tm >> 2 of that class),
  // so that every XCD's share of the tile list mixes light and heavy classes; a tile's reduction length is its class's.
  // One class (a2_class, or -1) may carry one MORE tap whose operand is a second gradient tensor at the same
  // positions (A2) with its own K-outer weights (B2): the 1x1 branch of the SK block, whose stride-2 data gradient
  // only reaches class (0, 0) -- both branches' input gradient in one launch.
  struct ParityClass { int cy, cx, kw, ntaps, wt_y0, wt_x0, out_base, k_end; };
  ParityClass cls[4];
  int bm_shift;                         // log2(BM) of the launching tile
  int wt_kw;                            // taps per row of the WINDOW (weights)
  int out_img_shift, out_y_shift;
  const float* A2;
  const float* B2;
  int lda2, ldb2, a2_class;
};

// class k's parameters by constant-index selects (a run-time index into the kernel-argument struct would send the whole
// struct to scratch memory)
__device__ __forceinline__ ConvGeom::ParityClass pick_class(const ConvGeom& c, int k) {
#define AIT_PICK(f) (k == 0 ? c.cls[0].f : k == 1 ? c.cls[1].f : k == 2 ? c.cls[2].f : c.cls[3].f)
  ConvGeom::ParityClass r;
  r.cy = AIT_PICK(cy); r.cx = AIT_PICK(cx); r.kw = AIT_PICK(kw); r.ntaps = AIT_PICK(ntaps);
  r.wt_y0 = AIT_PICK(wt_y0); r.wt_x0 = AIT_PICK(wt_x0); r.out_base = AIT_PICK(out_base); r.k_end = AIT_PICK(k_end);
#undef AIT_PICK
  return r;
}
// (class, row within the class) of GEMM row r under the tile-interleaved order
__device__ __forceinline__ void parity_row(const ConvGeom& c, int r, int& cls, int& rr) {
  const int tm = r >> c.bm_shift;
  cls = tm & 3;
  rr = ((tm >> 2) << c.bm_shift) | (r & ((1 << c.bm_shift) - 1));
}
// source row of class tap `tap` for class row rr (or -1), and the window tap whose weights it multiplies
__device__ __forceinline__ int parity_src_row(const ConvGeom& c, const ConvGeom::ParityClass& pc, int rr, int tap, int& wtap) {
  const int img = rr >> c.rows_hw_shift, rem = rr & ((1 << c.rows_hw_shift) - 1);
  const int y = rem >> c.rows_w_shift, x = rem & ((1 << c.rows_w_shift) - 1);
  const int ty = tap / pc.kw, tx = tap - ty * pc.kw;
  const int sy = y + pc.cy - ty, sx = x + pc.cx - tx;
  wtap = (pc.wt_y0 + 2 * ty) * c.wt_kw + pc.wt_x0 + 2 * tx;
  const bool ok = sy >= 0 && sx >= 0 && sy < c.src_h && sx < c.src_w;
  return ok ? (img * c.src_h + sy) * c.src_w + sx : -1;
}
enum { CONV_NONE = 0, CONV_A = 1, CONV_B = 2 };

// n / d for 0 <= n < 2^24 (exact in f32): the f32 quotient is off by at most one, corrected with the remainder
__device__ __forceinline__ int div_small(int n, int d, float inv, int& rem) {
  int q = (int)((float)n * inv);
  rem = n - q * d;
  if (rem < 0) { q--; rem += d; }
  if (rem >= d) { q++; rem -= d; }
  return q;
}

// position-major rows: (map, y, x) of row r, and the source row of position (sy, sx)
__device__ __forceinline__ void pm_decode(const ConvGeom& c, int r, int& map, int& y, int& x) {
  const int p = div_small(r, c.pm_maps, c.inv_pm, map);
  y = p >> c.rows_w_shift;
  x = p & ((1 << c.rows_w_shift) - 1);
}

// (PM: only the kernels of the 256 x 256 cooperative tile carry the position-major decode -- every launch over
// position-major rows runs on it; in the 256 x 128 kernels its registers cost ~800 bytes of scratch per lane)
template <bool PM = false>
__device__ __forceinline__ int conv_src_row(const ConvGeom& c, int r, int tap) {
  int img, y, x;
  if (PM && c.pm_maps > 0) {
    pm_decode(c, r, img, y, x);
  } else if (c.rows_hw_shift >= 0) {
    img = r >> c.rows_hw_shift;
    const int rem = r & ((1 << c.rows_hw_shift) - 1);
    y = rem >> c.rows_w_shift;
    x = rem & ((1 << c.rows_w_shift) - 1);
  } else {
    if (r >= c.n_rows) return -1;
    int rem;
    img = div_small(r, c.rows_hw, c.inv_hw, rem);
    y = div_small(rem, c.rows_w, c.inv_w, x);
  }
  const int ty = tap / c.kw, tx = tap - ty * c.kw;
  const int ny = y * c.a + c.c + ty * c.b, nx = x * c.a + c.c + tx * c.b;
  const int sy = ny >> c.div_shift, sx = nx >> c.div_shift;
  const bool ok = ny >= 0 && nx >= 0 && ((ny | nx) & ((1 << c.div_shift) - 1)) == 0 && sy < c.src_h && sx < c.src_w;
  if (PM && c.pm_maps > 0) return ok ? (sy * c.src_w + sx) * c.pm_maps + img : -1;
  return ok ? (img * c.src_h + sy) * c.src_w + sx : -1;
}
// the same with the tap already decomposed (ty, tx): the weight gradient's tap is fixed per column tile, and tap / kw is
// a ~35-instruction runtime division the transfers of every slab would repeat
template <bool PM = false>
__device__ __forceinline__ int conv_src_row_t(const ConvGeom& c, int r, int ty, int tx) {
  int img, y, x;
  if (PM && c.pm_maps > 0) {
    pm_decode(c, r, img, y, x);
  } else if (c.rows_hw_shift >= 0) {
    img = r >> c.rows_hw_shift;
    const int rem = r & ((1 << c.rows_hw_shift) - 1);
    y = rem >> c.rows_w_shift;
    x = rem & ((1 << c.rows_w_shift) - 1);
  } else {
    if (r >= c.n_rows) return -1;
    int rem;
    img = div_small(r, c.rows_hw, c.inv_hw, rem);
    y = div_small(rem, c.rows_w, c.inv_w, x);
  }
  const int ny = y * c.a + c.c + ty * c.b, nx = x * c.a + c.c + tx * c.b;
  const int sy = ny >> c.div_shift, sx = nx >> c.div_shift;
  const bool ok = ny >= 0 && nx >= 0 && ((ny | nx) & ((1 << c.div_shift) - 1)) == 0 && sy < c.src_h && sx < c.src_w;
  if (PM && c.pm_maps > 0) return ok ? (sy * c.src_w + sx) * c.pm_maps + img : -1;
  return ok ? (img * c.src_h + sy) * c.src_w + sx : -1;
}

// C / residual row of GEMM row r of a parity-class data gradient
__device__ __forceinline__ unsigned conv_out_row(const ConvGeom& c, int r) {
  int cls, rr;
  parity_row(c, r, cls, rr);
  const int img = rr >> c.rows_hw_shift, rem = rr & ((1 << c.rows_hw_shift) - 1);
  const int y = rem >> c.rows_w_shift, x = rem & ((1 << c.rows_w_shift) - 1);
  const int ob = cls == 0 ? c.cls[0].out_base : cls == 1 ? c.cls[1].out_base : cls == 2 ? c.cls[2].out_base : c.cls[3].out_base;
  return (unsigned)((img << c.out_img_shift) + (y << c.out_y_shift) + (x << 1) + ob);
}

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;      // [N] (or [M] with AIT_GEMM_BIAS_ROW)
  const float* residual;  // same addressing as C
  const float* gate;      // EPI_RES only, same addressing as C: the value is zeroed where gate <= 0 (after "+ residual");
                          // NULL = no gate.  The ReLU-backward mask of the layer whose input gradient this product forms
  int M, N, K;
  int lda, ldb, ldc;
  int c_colblk;           // 0: plain row-major C.  >0: C(i,j) at (j/colblk)*c_batch + i*ldc + j%colblk
  long long c_batch;
  float alpha;
  int flags;
  int k_per_split;
  int splits;
  unsigned long long* probe;   // diagnostic stamps (scripts/gemm_lab.hip); NULL in the product
  ConvGeom conv;               // CONV != 0 kernels only
  int batch, batch2;           // register-staged kernels: independent problems along gridDim.y (x gridDim.z) ...
  long long sA, sB, sC;        // ... whose operands are this many floats apart
  long long sA2, sB2, sC2;     // (second batch level: the heads of an attention product)
  int sk_on;                   // stream-K work list for the last round (set by launch())
  float* sk_ws;                // stream-K: one BM x BN partial tile per workgroup (not needed by EPI_ATOMIC launches)
  unsigned* sk_flags;          // stream-K: one "partial published" word per workgroup (zero between launches)
  unsigned* sched;             // dynamic tile hand-out: per XCD a ticket counter at [xcd*32] and an exit counter at
                               // [xcd*32 + 1] (zero between launches); NULL = static lists (item j, j+W, ...)
};

enum { MODE_DB = 0, MODE_RING = 1, MODE_DLDS = 2 };

// Tile configuration: BM x BN output tile, K-slabs of BK, WM x WN wavefronts each owning
// (BM/WM/32) x (BN/WN/32) MFMA tiles of 32x32.
// MODE_DLDS only: NS_ = slots of the slab ring (3: the slab requested during an iteration is awaited at
// its end; 4: one more slab stays in flight across the barrier, counted vmcnt), KNOBS_ = scheduling
// choices that do not change results (measured with scripts/gemm_lab.hip):
//   KNOB_BURST  the transfers of a slab are issued back to back at the top of the iteration instead of one
//               behind each of the first MFMA steps
//   KNOB_PRIO   the second-dispatched half of the workgroup's waves runs at s_setprio 1
//   KNOB_SPREAD the LDS reads of the next k-step group are issued one 32-row tile at a time ahead of each
//               MFMA step instead of all together in front of the group
//   KNOB_STAGGER the workgroup in an odd threadgroup slot of its CU (HW_ID.tg_id) starts half a tile late: the two
//               workgroups of a CU run the same program on tiles of the same length, so without it they reach their
//               epilogues together and the matrix pipes idle through both
enum { KNOB_BURST = 1, KNOB_PRIO = 2, KNOB_SPREAD = 4, KNOB_STAGGER = 8, KNOB_SPLIT = 16 /* products on the bf16 matrix pipe, see split8 */,
       KNOB_BF16 = 64 /* with KNOB_SPLIT: operands ROUNDED to bf16 (nearest even), one MFMA per block: bf16 products, f32 accumulate */,
       KNOB_RNE = 128 /* with KNOB_SPLIT: the three planes by round-to-nearest (dropped terms <= 2^-23 |a b|) instead of by
                         truncation (<= 2^-21): same instruction count, v_cvt_pk_bf16_f32 instead of v_perm_b32 / v_and */,
       KNOB_SPLIT_SIMPLE = 32 /* lab: KNOB_SPLIT with every split in front of its tile's MFMAs instead of under the previous tile's */,
       KNOB_COOP = 1024 /* with KNOB_SPLIT: every operand value is split ONCE per workgroup -- the raw slab lands in a two-slot ring,
                         each of the NT threads splits one operand row of the NEXT slab (16 values) and writes its planes into an
                         LDS image in the P3 row format, and the MFMAs of the current slab fetch ready planes (no vector work on
                         the matrix side).  One main loop for every operand layout: only the split stage looks at the raw image */,
       KNOB_NOTICKET = 512 /* no ticket ring in LDS: static work lists only (a ring of four 40-KB slabs is all of the CU's 160 KB) */,
       KNOB_AP3 = 2048 /* with KNOB_BP3: operand A arrives pre-split as well (an activation whose producer wrote the P3 form
                          beside the f32 one): no vector work at all on the matrix side, planes of both operands by LDS-DMA */,
       KNOB_BP3 = 256 /* with KNOB_SPLIT: operand B arrives PRE-SPLIT ("P3": the three bf16 planes of every value, interleaved in
                         groups of eight along the reduction dimension, see p3_split_kernel); only A is split in registers */ };
template <int BM_, int BN_, int BK_, int WM_, int WN_, int MINW_, int MODE_ = MODE_DB, int NS_ = 3, int KNOBS_ = 0>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, MINW = MINW_, MODE = MODE_;
  static constexpr int NS = NS_, KNOBS = KNOBS_;
  static constexpr int NT = 64 * WM * WN;          // threads
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr int PA = BM + 4, PB = BN + 4;   // LDS pitches (floats), 16-B aligned rows
  static constexpr int VA = BM * BK / 4 / NT;      // float4 per thread per A slab
  static constexpr int VB = BN * BK / 4 / NT;
  static constexpr int NBUF = (MODE_ == MODE_DB) ? 2 : 3;
  static constexpr int BKB = (KNOBS_ & 256 /* KNOB_BP3 */) ? 24 : BK_;   // floats per row of a B slab (P3: 16 values x 6 B)
  static constexpr int BKA = (KNOBS_ & 2048 /* KNOB_AP3 */) ? 24 : BK_;  // ... of an A slab
  static constexpr size_t LDS = (MODE_ == MODE_DLDS && (KNOBS_ & 1024 /* KNOB_COOP */))
                                    ? sizeof(float) * 2 * (BK + 24) * (BM + BN) + ((KNOBS_ & 512) ? 0 : 64)   // raw ring of two + two plane images
                                : (MODE_ == MODE_DLDS) ? sizeof(float) * NS_ * (BKA * BM + BKB * BN) + ((KNOBS_ & 512) ? 0 : 64) /* ticket ring */
                                                     : sizeof(float) * NBUF * BK * (PA + PB);
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "tile / wave mismatch");
  static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "slab / thread mismatch");
};

// diagnostic probe: NoProbe compiles to nothing
struct NoProbe { static constexpr bool on = false; };
struct StampProbe { static constexpr bool on = true; };
#define AIT_PROBE_WORDS 40   // u64 words per workgroup in GemmArgs::probe

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

// ---- LDS images of the direct-to-LDS path (gfx950 global_load_lds_dwordx4) ------------------------
// One wave-wide instruction moves 64 x 16 B = 1 KB: every lane supplies its own global address, the
// data lands at (wave-uniform LDS base) + lane * 16 B.  No staging registers, no ds_write.  Unpadded
// images made of 1-KB granules:
//   reduction dim contiguous in memory: "row image" [row][16 floats] = 64-B rows of four 16-B chunks,
//       chunk index XOR ((row >> 2) & 3) (conflict-free ds_read_b128: a lane fetches FOUR k-steps of
//       one operand tile at once).  Granule q = rows 16q..16q+15; lane L fills slot (row 16q + L/4,
//       position L%4) and therefore FETCHES the chunk that the swizzle assigns to that slot.  The k
//       order inside a slab is then: lane half lk works through k = 8*lk + s for MFMA step s = 0..7
//       (any order is legal as long as A and B agree);
//   reduction dim outermost: K-major [16][ROWS], granule q = 256 consecutive elements of it.
// Out-of-range rows are clamped to the last valid row (their products are never stored); the caller
// guarantees whole slabs (K range a multiple of 16) and ROWS-dim % 4 == 0 for K-outer operands.
__device__ __forceinline__ int rowimg_off(int row, int chunk) {     // float offset of a 16-B chunk
  return row * 16 + ((chunk ^ ((row >> 2) & 3)) << 2);
}

typedef __attribute__((address_space(3))) void lds_void;

// Issued as inline assembly on purpose: through the builtin the compiler treats the transfer as a
// store to LDS that may alias every later ds_read and puts s_waitcnt vmcnt(0) in front of the next
// operand fetch, i.e. it serialises the slab's memory latency with the MFMA stream.  Ordering is
// explicit instead: a slot is requested only after the barrier that retired its last reader, and
// awaited (vmcnt) before the barrier that publishes it.
__device__ __forceinline__ void glds16_at(const float* src, unsigned dst) {      // dst: wave-uniform LDS byte address
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "m0", "memory");
}
template <bool FORCE_UNIFORM = false>
__device__ __forceinline__ void glds16(const float* src, float* lds_dst) {
  // wave-uniform LDS address.  (FORCE_UNIFORM: the grouped-convolution instantiations, where hipcc loses the proof
  // and would hand the asm a VGPR; the readfirstlane drags the address arithmetic into vector registers, which is why
  // it is not used for the other kernels)
  const unsigned dst = FORCE_UNIFORM ? __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)lds_dst)
                                     : (unsigned)(size_t)(lds_void*)lds_dst;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "m0", "memory");
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

template <bool ROWIMG, int PITCH>
__device__ __forceinline__ float4 fetch_tile(const float* __restrict__ slab, int row0, int li, int lk, int g, int t) {
  if (ROWIMG) return *reinterpret_cast<const float4*>(slab + rowimg_off(row0 + t * 32 + li, 2 * lk + g));
  const float* p = slab + (8 * lk + 4 * g) * PITCH + row0 + t * 32 + li;
  return make_float4(p[0], p[PITCH], p[2 * PITCH], p[3 * PITCH]);
}

// AK / BKC: true when that operand is stored with the reduction dimension contiguous.
//   forward  y = x W^T   : A = x [M,K] (AK), B = W [N,K] (BKC)
//   dgrad    dx = dy W   : A = dy [M,K'] (AK), B = W [K',N] (!BKC)
//   wgrad    dW = dy^T x : A = dy [K',M] (!AK), B = x [K',N] (!BKC)
enum { EPI_STORE = 0, EPI_ATOMIC = 1, EPI_AUX = 2, EPI_RES = 3, EPI_RESG = 4 /* EPI_RES + gate (GemmArgs::gate) */ };

// C/D layout of the 32x32 MFMA (any input dtype): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// EPI selects the epilogue at compile time (a run-time flag test per element makes hipcc branch
// around every load/store and wait vmcnt(0) each time):
//   EPI_STORE  C = alpha*acc (+bias) (relu)            -- no loads between the stores
//   EPI_ATOMIC C += alpha*acc with fp32 atomics        -- split-K partial tiles
//   EPI_RES / EPI_AUX  the forms that read memory (see below)
// Addressing: element offsets are 32-bit (make_args rejects outputs of 2^31 elements or more), formed
// as (uniform base pointer) + (per-lane unsigned offset) so that hipcc emits the saddr + voffset form
// of the global instructions: one VGPR per address instead of a 64-bit pair per store.
//
// The epilogue must not put a LOAD between its stores: vmcnt retires in order, so waiting for a load
// issued after a store waits for that store's full round trip to memory (~2 us under load), once per
// group of stores -- measured at ~20 us per 256x128 tile in the first version of this file, where the
// per-element `bias ? bias[..] : 0` selects compiled to branches around loads.  Hence:
//   EPI_STORE   takes only the column-bias form (dispatch() sends row biases to EPI_AUX); the TN bias
//               values of a lane are loaded ONCE, before any store;
//   EPI_RES     "+ residual" and the ReLU-backward gate (the two forms the training path uses on large
//               products): software-pipelined, the loads of the next half tile go out before the stores
//               of the current one;
//   EPI_AUX     everything else (accumulate into C, row bias, combinations): loads in uniform-branch
//               blocks before each half tile's stores (small / rare launches).
template <int TM, int TN, int EPI, bool ROWMAP = false>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[TM][TN], const GemmArgs& g, int m0, int n0,
                                         int wm, int wn, int li, int lk, int m_lim = -1) {
  // rows of the result that exist for THIS tile: g.M, or less for a tile that ends inside the matrix (tap skipping:
  // ConvGeom::pm_skip -- a scalar, not a copy of the argument struct, which would go to scratch memory)
  const int gM = m_lim >= 0 ? m_lim : g.M;
  const bool relu = (g.flags & AIT_GEMM_RELU) != 0;
  const unsigned ldc = (unsigned)g.ldc;
  const bool interior = (m0 + wm + TM * 32 <= gM) && (n0 + wn + TN * 32 <= g.N);   // wave-uniform
  // Element offset of the first column of the row that holds accumulator register r of an MFMA tile whose first row
  // (for this lane) is `rb`: tile_off(rb) + reg_off(r).  Plain: rows rb + (r&3) + 8*(r>>2).  ROWMAP (parity-class
  // data gradient on 4 x 4 class grids): rb is a multiple of 4 with y = lane half, so register r sits at x = r & 3,
  // y + 2 * ((r >> 2) & 1), image + (r >> 3) -- constant row offsets again, one row-map evaluation per tile.
  auto tile_off = [&](int rb) -> unsigned {
    if constexpr (ROWMAP) return conv_out_row(g.conv, rb) * ldc;
    else return (unsigned)rb * ldc;
  };
  auto reg_off = [&](int r) -> unsigned {
    if constexpr (ROWMAP)
      return (unsigned)(((r >> 3) << g.conv.out_img_shift) + (((r >> 2) & 1) << (g.conv.out_y_shift + 1)) + ((r & 3) << 1)) * ldc;
    else return (unsigned)(((r) & 3) + 8 * ((r) >> 2)) * ldc;
  };
  // row r of an MFMA tile sits (r&3) + 8*(r>>2) rows below its first row
#define AIT_ROW(r) (((r) & 3) + 8 * ((r) >> 2))
  const bool colsum = EPI != EPI_ATOMIC && (g.flags & AIT_GEMM_COLSUM) != 0;    // g.bias is then the OUTPUT
  float bcol[TN], cs[TN];
#pragma unroll
  for (int b = 0; b < TN; b++) { bcol[b] = 0.f; cs[b] = 0.f; }
  if (EPI != EPI_ATOMIC && g.bias && !(g.flags & AIT_GEMM_BIAS_ROW) && !colsum) {
#pragma unroll
    for (int b = 0; b < TN; b++) bcol[b] = g.bias[min(n0 + wn + b * 32 + li, g.N - 1)];
  }
  // column part of the element offset, per tile column b
  unsigned cb[TN];
  bool cok[TN];
#pragma unroll
  for (int b = 0; b < TN; b++) {
    const int col = n0 + wn + b * 32 + li;
    cok[b] = col < g.N;
    const int colc = cok[b] ? col : 0;
    cb[b] = g.c_colblk > 0 ? (unsigned)(colc / g.c_colblk) * (unsigned)g.c_batch + (unsigned)(colc % g.c_colblk)
                           : (unsigned)colc;
  }

  // column sums of what this wave stores, added to g.bias[col] at the end (lanes l and l+32 hold the same column)
  auto flush_colsum = [&]() {
    if (!colsum) return;
#pragma unroll
    for (int b = 0; b < TN; b++) {
      const float s = cs[b] + __shfl_xor(cs[b], 32, 64);
      if (lk == 0 && cok[b]) unsafeAtomicAdd(const_cast<float*>(g.bias) + (n0 + wn + b * 32 + li), s);
    }
  };

  if (EPI == EPI_RES || EPI == EPI_RESG) {
    // ---- residual add / ReLU-backward gate, software-pipelined over the TM*TN MFMA tiles: the 16 loads
    // of tile i+1 are issued BEFORE the 16 stores of tile i, so the wait for them is a counted vmcnt that
    // leaves those stores in flight, and a tile costs one load round trip (a quarter-tile pipeline, four
    // round trips per MFMA tile, measured 13 % slower on the ReLU-gated dgrad shape)
    const bool mask_pos = (g.flags & AIT_GEMM_MASK_POS) != 0;
    constexpr bool gated = EPI == EPI_RESG;        // (its own instantiation: 32 more registers in flight)
    constexpr int NT_ = TM * TN;
    float x[16], xn[16], gt[gated ? 16 : 1], gn[gated ? 16 : 1];
    auto row_of = [&](int i, int r) { return m0 + wm + (i / TN) * 32 + 4 * lk + AIT_ROW(r); };
    // offset of register r of MFMA tile i (loads clamp to the last row at a ragged bottom edge; the parity row map
    // never has one: its row count is a multiple of the tile)
    auto off_of = [&](int i, int r, bool clamp) -> unsigned {
      if constexpr (ROWMAP) return cb[i % TN] + tile_off(m0 + wm + (i / TN) * 32 + 4 * lk) + reg_off(r);
      else return cb[i % TN] + (unsigned)(clamp ? min(row_of(i, r), gM - 1) : row_of(i, r)) * ldc;
    };
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const unsigned o = off_of(0, r, true);
      x[r] = g.residual[o];
      if constexpr (gated) gt[r] = g.gate[o];
    }
#pragma unroll
    for (int i = 0; i < NT_; i++) {
      const int a = i / TN, b = i % TN;
      if (i + 1 < NT_) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const unsigned o = off_of(i + 1, r, true);
          xn[r] = g.residual[o];
          if constexpr (gated) gn[r] = g.gate[o];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = g.alpha * acc[a][b][r] + bcol[b];
        v = mask_pos ? (x[r] > 0.f ? v : 0.f) : v + x[r];
        if constexpr (gated) v = gt[r] > 0.f ? v : 0.f;
        if (relu) v = fmaxf(v, 0.f);
        const int row = row_of(i, r);
        if (interior || (cok[b] && row < gM)) { g.C[off_of(i, r, false)] = v; cs[b] += v; }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; r++) {
        x[r] = xn[r];
        if constexpr (gated) gt[r] = gn[r];
      }
    }
    flush_colsum();
    return;
  }

#pragma unroll
  for (int a = 0; a < TM; a++)
#pragma unroll
    for (int b = 0; b < TN; b++) {
      // one MFMA tile at a time (the fence keeps hipcc from interleaving the address arithmetic of all
      // tiles, which does not fit the 128-register budget of the two-workgroups-per-CU kernels)
      __builtin_amdgcn_sched_barrier(0);
      const int rbase = m0 + wm + a * 32 + 4 * lk;
      const unsigned obase = cb[b] + (unsigned)rbase * ldc;     // element offset of (rbase, col)
      if (EPI == EPI_ATOMIC) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          if (interior || (cok[b] && rbase + AIT_ROW(r) < gM))
            unsafeAtomicAdd(g.C + (obase + (unsigned)AIT_ROW(r) * ldc), g.alpha * acc[a][b][r]);
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; q++) v[q] = g.alpha * acc[a][b][h * 8 + q] + bcol[b];
          if (EPI == EPI_AUX) {
            // the general form (accumulate into C, row bias, any combination): loads in uniform-branch
            // blocks, then the arithmetic.  (Every such load waits for the stores before it: this path
            // serves small / rare launches only.)
            unsigned off[8];
#pragma unroll
            for (int q = 0; q < 8; q++) off[q] = cb[b] + (unsigned)min(rbase + AIT_ROW(h * 8 + q), gM - 1) * ldc;
            if (g.bias && (g.flags & AIT_GEMM_BIAS_ROW)) {
#pragma unroll
              for (int q = 0; q < 8; q++) v[q] += g.bias[min(rbase + AIT_ROW(h * 8 + q), gM - 1)];
            }
            if (g.residual) {
              float x[8];
#pragma unroll
              for (int q = 0; q < 8; q++) x[q] = g.residual[off[q]];
              if (g.flags & AIT_GEMM_MASK_POS) {
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = x[q] > 0.f ? v[q] : 0.f;
              } else {
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] += x[q];
              }
            }
            if (g.flags & AIT_GEMM_ACCUMULATE) {
              float y[8];
#pragma unroll
              for (int q = 0; q < 8; q++) y[q] = g.C[off[q]];
#pragma unroll
              for (int q = 0; q < 8; q++) v[q] += y[q];
            }
          }
          if (relu) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = fmaxf(v[q], 0.f);
          }
          if constexpr (ROWMAP) {
            const unsigned tb = cb[b] + tile_off(rbase);
#pragma unroll
            for (int q = 0; q < 8; q++)
              if (interior || cok[b]) {
                g.C[tb + reg_off(h * 8 + q)] = v[q];
                cs[b] += v[q];
              }
          } else if (interior) {
#pragma unroll
            for (int q = 0; q < 8; q++) { g.C[obase + (unsigned)AIT_ROW(h * 8 + q) * ldc] = v[q]; cs[b] += v[q]; }
          } else {
#pragma unroll
            for (int q = 0; q < 8; q++)
              if (cok[b] && rbase + AIT_ROW(h * 8 + q) < gM) {
                g.C[obase + (unsigned)AIT_ROW(h * 8 + q) * ldc] = v[q];
                cs[b] += v[q];
              }
          }
        }
      }
    }
  flush_colsum();
#undef AIT_ROW
}

// ---- f32 products on the bf16 matrix pipe (KNOB_SPLIT) --------------------------------------------------
// v_mfma_f32_32x32x16_bf16 retires 16x the FLOP per cycle of v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH: 32 cycles
// for 32x32x16 against 64 for 32x32x2).  An f32 value is EXACTLY the sum of three bf16 values (24 significant bits =
// 8 + 8 + 8: h = x truncated to bf16, m = (x - h) truncated, l = x - h - m), a product of two bf16 values is exact in
// f32, and the MFMA accumulates in f32.  So  a*b = (ah + am + al)(bh + bm + bl)  is formed from the six terms that
// are >= 2^-16 |a b|  --  ah bh, ah bm, am bh, ah bl, al bh, am bm  --  and the three dropped ones are <= 2^-21 |a b|
// together (truncation; 2^-23 with KNOB_RNE): a few roundings of an f32 multiply-add chain (2^-24 of the running sum per step).
// PRODUCT FORM SINCE ROUND 4: KNOB_RNE in every product tile.  Truncated planes all carry the sign of the value; on long
// same-signed sums the bf16 pipe's accumulator alignment chops the small ones and the result drifts low by up to 1.4e-5
// of the sum (K = 4800; profiles/r04_split_bias.txt) -- the planes rounded to nearest are zero-mean and show no drift.  Six 32-cycle
// instructions replace eight 64-cycle ones per 16 k-steps; the split costs 5.5 vector instructions per fetched value,
// issued in the shadow of the MFMAs.  The operand images in LDS, the LDS-DMA stream and the epilogues are untouched:
// a lane's eight k-values of a slab (k = 8*lk + 0..7) are exactly the 32x32x16 operand layout.
// Non-finite inputs give NaN where the f32 instruction gives an infinity (inf - inf in the remainder).
// (Planes, split2 / split8, mfma_split: split_planes.h -- shared with the attention tiles, csrc/attn.hip)

// one group of four k-steps: TM x TN MFMAs per step on the operand quads xa / xb
template <int TM, int TN>
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[TM][TN], const float4 (&xa)[TM], const float4 (&xb)[TN]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
      for (int b = 0; b < TN; b++) {
        const float fa = j == 0 ? xa[a].x : j == 1 ? xa[a].y : j == 2 ? xa[a].z : xa[a].w;
        const float fb = j == 0 ? xb[b].x : j == 1 ? xb[b].y : j == 2 ? xb[b].z : xb[b].w;
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a][b], 0, 0, 0);
      }
  }
}

// ---- work decomposition shared by both kernels --------------------------------------------------------
// Work items are (split, tile) pairs in one linear list; every XCD (blocks b and b+8 share an XCD /
// L2) owns one contiguous chunk of it:
//   splits == 1: items = tiles in row-major order, so the N-tiles of an M-panel run back to back on
//                ONE XCD (its A panel stays in that L2) and all 8 XCDs are busy whatever tiles_m is;
//   splits  > 1 (weight gradients): split-major order -- every XCD owns splits/8 K-ranges and runs ALL
//                output tiles of them concurrently, so each byte of A and B crosses the fabric once and
//                the 16-row slabs that the co-running tiles walk in step are served from that L2.
struct WorkMap {
  int tiles_m, tiles_n, tiles, items, chunk;
  // position-block weight gradient (ConvGeom::pm_wgrad): taps of the window that reach output position q along one axis
  // (n positions, window k, padding pad, stride 1): [lo, lo + cnt)
  __host__ __device__ static void pm_taps(int q, int n, int k, int pad, int& lo, int& cnt) {
    lo = pad - q > 0 ? pad - q : 0;
    const int hi = (n - 1 + pad - q) < (k - 1) ? (n - 1 + pad - q) : (k - 1);
    cnt = hi >= lo ? hi - lo + 1 : 0;
  }
  __host__ __device__ static int pm_pairs(const ConvGeom& c) {       // valid (position, tap) pairs of one map
    int tot = 0;
    for (int y = 0; y < c.src_h; y++)
      for (int x = 0; x < c.src_w; x++) {
        int l, ny, nx;
        pm_taps(y, c.src_h, c.pm_kh, -c.c, l, ny);
        pm_taps(x, c.src_w, c.kw, -c.c, l, nx);
        tot += ny * nx;
      }
    return tot;
  }
  __host__ __device__ void init(const GemmArgs& g, int BM, int BN) {
    tiles_n = (g.N + BN - 1) / BN;
    tiles_m = (g.M + BM - 1) / BM;
    tiles = tiles_m * tiles_n;
    if (g.conv.pm_wgrad) {
      // (tiles of ONE tap) x (valid pairs), position-major: neighbouring items reduce over the same block of rows
      items = tiles_m * (g.conv.seg / BN) * pm_pairs(g.conv);
      chunk = (items + AIT_NXCD - 1) / AIT_NXCD;
      return;
    }
    if (g.conv.pm_skip) {
      // every position's block in its own row tiles (this launch's slice of them)
      tiles_m = g.conv.pm_tcnt << g.conv.rows_hw_shift;
      tiles = tiles_m * tiles_n;
      items = tiles;
      chunk = (items + AIT_NXCD - 1) / AIT_NXCD;
      return;
    }
    items = tiles * g.splits;
    chunk = (g.splits == 1) ? (tiles + AIT_NXCD - 1) / AIT_NXCD
                            : ((g.splits + AIT_NXCD - 1) / AIT_NXCD) * tiles;
  }
  // taps along one axis that reach the map from position q of n: source coordinate q + c + b * t (b = +1 forward, -1 data
  // gradient), window k: [lo, lo + cnt)
  __host__ __device__ static void pm_axis(int q, int n, int k, int b, int c, int& lo, int& cnt) {
    int l = b > 0 ? -c - q : q + c - (n - 1), h = b > 0 ? n - 1 - c - q : q + c;
    if (l < 0) l = 0;
    if (h > k - 1) h = k - 1;
    lo = l;
    cnt = h >= l ? h - l + 1 : 0;
  }
  struct PmTile { int p, ylo, ny, xlo, nx, m_end; };
  // tap skipping: the position of tile-list entry `id` (through pm_order), its tap window, the end of its block of rows
  __host__ __device__ PmTile pm_tile(const GemmArgs& g, int id, int BM) const {
    const ConvGeom& c = g.conv;
    const int tpp = c.pm_tcnt * tiles_n;
    PmTile t;
    t.p = (int)((c.pm_order >> (4 * (id / tpp))) & 15ull);
    pm_axis(t.p >> c.rows_w_shift, c.src_h, c.pm_kh, c.b, c.c, t.ylo, t.ny);
    pm_axis(t.p & ((1 << c.rows_w_shift) - 1), c.src_w, c.kw, c.b, c.c, t.xlo, t.nx);
    t.m_end = (t.p + 1) * c.pm_maps;
    return t;
  }
  // item id -> tile origin and K range (PM: the kernels that serve position-major launches, see conv_src_row)
  template <bool PM = false>
  __device__ __forceinline__ void decode(const GemmArgs& g, int id, int BM, int BN, int& m0, int& n0,
                                         int& kbeg, int& kend) const {
    if (PM && g.conv.pm_skip) {
      const int tpp = g.conv.pm_tcnt * tiles_n;
      const PmTile t = pm_tile(g, id, BM);
      const int r = id - (id / tpp) * tpp, tr = r / tiles_n;
      m0 = t.p * g.conv.pm_maps + (g.conv.pm_t0 + tr) * BM;
      n0 = (r - tr * tiles_n) * BN;
      kbeg = 0;
      kend = t.ny * t.nx * g.conv.seg;
      return;
    }
    if (PM && g.conv.pm_wgrad) {
      const ConvGeom& c = g.conv;
      const int tpt = c.seg / BN, per = tiles_m * tpt;
      int j = id / per;
      const int t = id - j * per;
      int p = 0, ylo = 0, ny = 0, xlo = 0, nx = 0;
      for (;; p++) {                                  // (at most src_h * src_w steps; scalar)
        pm_taps(p >> c.rows_w_shift, c.src_h, c.pm_kh, -c.c, ylo, ny);
        pm_taps(p & ((1 << c.rows_w_shift) - 1), c.src_w, c.kw, -c.c, xlo, nx);
        if (j < ny * nx) break;
        j -= ny * nx;
      }
      const int jy = j / nx, tap = (ylo + jy) * c.kw + xlo + (j - jy * nx);
      const int tm = t / tpt;
      m0 = tm * BM;
      n0 = (tap * tpt + (t - tm * tpt)) * BN;
      kbeg = p * c.pm_maps;
      kend = kbeg + c.pm_maps;
      return;
    }
    int split = 0, t = id;
    if (g.splits > 1) { split = id / tiles; t = id - split * tiles; }
    const int tm = t / tiles_n;
    m0 = tm * BM;
    n0 = (t - tm * tiles_n) * BN;
    kbeg = split * g.k_per_split;
    kend = min(g.K, kbeg + g.k_per_split);
  }
};

// =========================================================================================================
// Persistent direct-to-LDS kernel (MODE_DLDS).  gridDim.x = 8 * W workgroups; workgroup (xcd, j) walks
// the items j, j + W, j + 2W ... of its XCD's chunk.  The three-slab LDS ring is fed by ONE continuous
// stream of slabs that runs across tile boundaries: slab s+2 of the stream is requested at the top of
// iteration s into the slot that iteration s-1 finished reading, and awaited (vmcnt) just before the
// barrier that ends iteration s; operands are fetched per group of four k-steps, the first group of
// slab s+1 before that barrier, so the MFMA stream runs across it -- and across the epilogue of a
// finished tile, whose stores are issued while the next tile's first slabs are already in LDS.
// =========================================================================================================
template <class C, bool AK, bool BKC, int EPI, class Probe = NoProbe, int CONV = CONV_NONE, bool GRP = false, bool ROWMAP = false>
__global__ __launch_bounds__(C::NT, C::MINW) void gemm_f32_stream_kernel(const GemmArgs g) {
  static_assert(C::MODE == MODE_DLDS && C::BK == 16 && C::BM % 16 == 0 && C::BN % 16 == 0,
                "direct-to-LDS path needs 16-float slabs");
  static_assert(CONV != CONV_A || AK, "the gathered operand of CONV_A is K-contiguous");
  static_assert(CONV != CONV_B || (!AK && !BKC), "CONV_B is the weight-gradient layout");
  static_assert((C::BM * C::BKA + C::BN * C::BKB) / 256 <= 8 * (C::NT / 64), "at most 8 transfers per wave per slab");
  constexpr int BM = C::BM, BN = C::BN, BK = 16;
  constexpr bool kBp3 = (C::KNOBS & KNOB_BP3) != 0;   // B pre-split: rows of 16 values x 3 planes = six 16-B chunks
  static_assert(!kBp3 || (AK && BKC && CONV == CONV_NONE && (C::KNOBS & KNOB_SPLIT) != 0 && (C::BN * 24) % 256 == 0),
                "pre-split B: both operands K-contiguous, dense");
  constexpr bool kAp3 = (C::KNOBS & KNOB_AP3) != 0;   // A pre-split too
  static_assert(!kAp3 || (kBp3 && (C::BM * 24) % 256 == 0 && !ROWMAP && !GRP), "pre-split A: on top of pre-split B");
  constexpr int SA = BM * C::BKA, SB = BN * C::BKB;      // floats per slab image
  constexpr int NW = C::NT / 64;
  constexpr int GA = SA / 256, GB = SB / 256;        // 1-KB granules per slab
  constexpr int LA = (GA + NW - 1) / NW, LB = (GB + NW - 1) / NW;   // transfers per wave per slab
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NS = C::NS;
  constexpr int NP = LA + LB;     // transfers per wave per slab
  constexpr bool kCoop = (C::KNOBS & KNOB_COOP) != 0;
  static_assert(!kCoop || ((C::KNOBS & KNOB_SPLIT) != 0 && !kBp3 && C::NT == BM + BN), "cooperative split: one thread per operand row");
  static_assert(NS == 3 || NS == 4, "ring of 3 or 4 slabs");
  static_assert(NS == 3 || (GA % NW == 0 && GB % NW == 0), "counted waits need the same transfer count in every wave");
  constexpr int NSR = kCoop ? 2 : NS;      // slots of the raw slab ring
  float* As = lds;                // [NSR][SA]
  float* Bd = lds + NSR * SA;     // [NSR][SB]
  float* Pa = lds + NSR * (SA + SB);       // kCoop: [2][BM * 24] plane image of A (P3 rows, p3_impl.h), then of B
  float* Pb = Pa + 2 * BM * 24;

  // raw barrier: __syncthreads() is a fence + barrier, and the fence makes hipcc drain vmcnt(0) whenever it
  // has stores of its own outstanding (the epilogue's), which would also drain the slab kept in flight
  auto ring_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  WorkMap wmap;
  wmap.init(g, BM, BN);
  const int W = gridDim.x / AIT_NXCD;
  const int xcd = blockIdx.x % AIT_NXCD, j = blockIdx.x / AIT_NXCD;
  const int base = xcd * wmap.chunk;
  const int lim = max(0, min(wmap.items - base, wmap.chunk));     // items in this XCD's chunk

  // ---- work list of this workgroup: [stream-K pieces] [whole items dp0, dp0 + W, ...] ------------------
  // With a partial-tile workspace (g.sk_ws) the r = lim % W tiles that would form an under-filled last
  // round are not handed out whole: their r * K/16 slabs are cut into equal contiguous runs, one per
  // workgroup, done FIRST.  A run is at most one tile long and covers the tail of one tile ("piece A") and/or
  // the head of the next ("piece B"), so a workgroup has at most two pieces.  The piece that ENDS a tile
  // (kend == K) owns it: after it the owner adds the partial tiles that the workgroups BEFORE it published
  // (fixed order: results do not depend on timing) and runs the epilogue.  A piece that does not end its tile
  // is computed and published FIRST, before its workgroup ever waits.  Waits therefore point only at LOWER
  // workgroup ids, and what they wait for is the first thing those workgroups do: with the dispatcher handing
  // out workgroups in id order, a resident workgroup only ever waits for workgroups that are resident or done
  // -- no cycle and no dependence on the whole grid being co-resident (a second stream's kernel, an RCCL
  // kernel or a CU mask may hold part of the chip).
  constexpr int SK_MIN = 4;          // slabs per run at least
  int sk_r = 0, sk_total = 0, sk_w = 1, n_sk = 0;
  int skA_tile = 0, skA_kb = 0, skA_ke = 0, skB_ke = 0;
  // slabs per item: the whole reduction, or one K-split of it (split-K launches combine with atomics: their pieces
  // need no hand-off at all -- every piece simply adds its partial tile; launch() enables the list only when all
  // splits have the same length)
  int ns = (g.splits > 1 ? g.k_per_split : g.K) / BK;
  // Tap skipping (ConvGeom::pm_skip) makes the items of a chunk differ in length -- by position, and a chunk holds the tiles
  // of whole positions in order, so its items form runs of equal length.  The cut treats (up to) TWO such runs as two
  // independent groups side by side: group `grp` = items [g_i0, g_i0 + sk_r) of the chunk, each g-local `ns` slabs long, cut
  // over the workgroups [g_j0, g_j0 + g_W) in proportion to the groups' work.  Inside a group everything is the uniform
  // scheme above with (j - g_j0, g_W) for (j, W).  (One group, all of W: every other launch.)
  int g_i0 = 0, g_j0 = 0, g_W = W;
  if (g.sk_on) {
    sk_r = lim % W;
    if constexpr (CONV == CONV_A && kCoop) {
      if (g.conv.pm_skip) {
        auto len_of = [&](int i) -> int { const WorkMap::PmTile t = wmap.pm_tile(g, base + i, BM); return t.ny * t.nx * (g.conv.seg / BK); };
        sk_r = lim;                                  // (launch() made sure that lim <= W: no whole items beside the cut)
        const int L0 = len_of(0);
        int na = 1;
        while (na < lim && len_of(na) == L0) na++;
        ns = L0;
        if (na < lim) {
          const int L1 = len_of(na);
          const long long wa = (long long)na * L0, wb = (long long)(lim - na) * L1;
          int Wa = (int)(((long long)W * wa + (wa + wb) / 2) / (wa + wb));
          Wa = max(1, min(W - 1, Wa));
          if (j < Wa) { sk_r = na; g_W = Wa; }
          else { g_i0 = na; sk_r = lim - na; g_j0 = Wa; g_W = W - Wa; ns = L1; }
        }
      }
    }
    const int jl = j - g_j0;
    sk_total = sk_r * ns;
    sk_w = max(1, min(g_W, sk_total / SK_MIN));
    if (sk_r > 0 && jl < sk_w) {
      const int lo = (int)((long long)jl * sk_total / sk_w), hi = (int)((long long)(jl + 1) * sk_total / sk_w);
      skA_tile = lo / ns;
      const int e0 = min(hi, (skA_tile + 1) * ns);
      skA_kb = (lo - skA_tile * ns) * BK;
      skA_ke = (e0 - skA_tile * ns) * BK;
      n_sk = 1;
      if (hi > e0) { skB_ke = (hi - e0) * BK; n_sk = 2; }
    }
  }
  const int sk_all = (CONV == CONV_A && kCoop && g.conv.pm_skip && g.sk_on) ? lim : sk_r;      // items of the chunk that are cut, all groups
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // The whole items (sk_r .. lim-1 of the chunk) are handed out DYNAMICALLY when the launch has a ticket
  // counter (g.sched): a workgroup draws its next item when it needs one, so a workgroup that becomes
  // resident late -- the grid is sized for an empty chip; beside an RCCL kernel of a data-parallel step, or
  // any other concurrent kernel, some workgroups start only when others exit -- finds what is left instead
  // of a fixed share of the list (a static list would nearly double the kernel's duration then), and CUs of
  // unequal speed balance.  Tickets are drawn in order, so neighbouring tiles still run together (operand
  // panels shared in the XCD's L2).  One lane draws at the top of a slab
  // iteration (for the item after the load cursor's); the ticket lands in a small LDS ring before the barrier that ends the iteration (whose
  // vmcnt(0) wait covers it), so no wave ever waits for the counter.  The last workgroup of an XCD to exit
  // zeroes its two counters.
  const bool dyn = g.sched != nullptr;
  const int lim_dp = lim - sk_all;                                 // whole items of this XCD
  int* idq = reinterpret_cast<int*>(lds + NSR * (SA + SB) + (kCoop ? 2 * 24 * (BM + BN) : 0));         // [8] ticket ring
  unsigned* ticket = g.sched + xcd * 32;
  int n_fetched = 0;            // tickets in the ring so far (identical in every wave)
  bool ended = false;           // a drawn ticket was past the end
  if (dyn) {
    if (threadIdx.x == 0) {
      const unsigned t0 = atomicAdd(ticket, 1u);
      idq[0] = (int)t0 < lim_dp ? (int)t0 : -1;
    }
    __syncthreads();
    n_fetched = 1;
    ended = idq[0] < 0;
  }
  // whole item k of this workgroup -> id within the chunk's whole items, or -1 past the end
  auto dp_id = [&](int k) __attribute__((always_inline)) -> int {
    if (dyn) return idq[k & 7];
    const int id = j + k * W;
    return id < lim_dp ? id : -1;
  };
  bool any = n_sk > 0 || dp_id(0) >= 0;
  // item it -> tile origin, K range; false past the end of this workgroup's work
  auto get_item = [&](int it, int& m0, int& n0, int& kb, int& ke) __attribute__((always_inline)) -> bool {
    int id;
    const bool pieceB = n_sk == 2 && it == 0;        // (two pieces: the non-owned head of the next tile goes first)
    if (it < n_sk) id = g_i0 + skA_tile + (pieceB ? 1 : 0);
    else {
      id = dp_id(it - n_sk);
      if (id < 0) return false;
      id += sk_all;
    }
    wmap.template decode<kCoop>(g, base + id, BM, BN, m0, n0, kb, ke);
    if constexpr (ROWMAP) { kb = 0; ke = pick_class(g.conv, (m0 >> g.conv.bm_shift) & 3).k_end; }     // (never with stream-K pieces)
    if (it < n_sk) {                      // a piece: its sub-range of the item's K range
      ke = kb + (pieceB ? skB_ke : skA_ke);
      kb = kb + (pieceB ? 0 : skA_kb);
    }
    return true;
  };
  auto leave = [&]() __attribute__((always_inline)) {          // exit protocol of the dynamic hand-out
    if (dyn && threadIdx.x == 0) {
      if (atomicAdd(ticket + 1, 1u) == (unsigned)W - 1u) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if (!any) { leave(); return; }
  if constexpr ((C::KNOBS & KNOB_STAGGER) != 0) {
    const unsigned hw = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID
    if ((hw >> 16) & 1u) {
      // one tile of `ns` slabs takes a wave ns * 64 MFMAs * 64 cycles when it has the SIMD's matrix pipe to itself,
      // twice that beside its partner: half a shared tile = ns * 4096 cycles
      const unsigned long long wait = (unsigned long long)ns * 4096ull, t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(64);
    }
  }
  const int wm = (wave / C::WN) * (C::TM * 32), wn = (wave % C::WN) * (C::TN * 32);
  const int li = lane & 31, lk = lane >> 5;

  unsigned long long t_start = 0, c_start = 0, c_loop = 0, c_wait = 0, c_bar = 0, n_slab = 0, n_tile = 0;
  if constexpr (Probe::on) { t_start = __builtin_amdgcn_s_memrealtime(); c_start = __builtin_amdgcn_s_memtime(); }

  // ---- load cursor: runs two slabs ahead of the multiply ----------------------------------------
  const float* pa[LA];
  const float* pb[LB];
  const size_t step_a = kAp3 ? 24 : AK ? 16 : (size_t)16 * g.lda;
  const size_t step_b = kBp3 ? 24 : BKC ? 16 : (size_t)16 * g.ldb;
  size_t step_b_cur = step_b;       // ROWMAP: the extra tap's weights have their own row pitch
  int l_item = 0, l_k = 0, l_kend = 0;
  int l_n0 = 0, l_m0 = 0;           // origin of the load cursor's tile (CONV kernels)
  int l_kin = 0;                    // CONV_A: reduction index within the current tap (retap() sets it; no per-slab modulo)
  int l_ty = 0, l_tx = 0, l_ch0 = 0;   // CONV_B: the column tile's tap (row, column of the window) and first source channel (+ group offset)
  int l_pm_shift = 0;                  // ... and the source rows' constant offset for the item's tap
  int l_blk_end = 0x7fffffff;          // CONV_B over position blocks: first row behind the item's block (a piece of an item ends on a slab, not on the block)
  int arow[LA];                     // CONV_A: this lane's (clamped) GEMM row per A transfer
  bool l_valid = true;
  // CONV_A: operand pointers at reduction index l_k = tap * seg + kin
  auto retap = [&]() __attribute__((always_inline)) {
    int tap = l_k / g.conv.seg;
    const int kin = l_k - tap * g.conv.seg;
    l_kin = kin;
    // tap skipping (ConvGeom::pm_skip): the reduction index counts only the taps that reach the map from the tile's position;
    // their sources lie a constant number of rows away (position-major rows), inside the map by construction
    bool pmk = false;
    int pm_shift = 0;
    if constexpr (CONV == CONV_A && !ROWMAP && !GRP && kCoop) {
      if (g.conv.pm_skip) {
        const ConvGeom& c = g.conv;
        const int p = l_m0 / c.pm_maps;
        int ylo, ny, xlo, nx;
        WorkMap::pm_axis(p >> c.rows_w_shift, c.src_h, c.pm_kh, c.b, c.c, ylo, ny);
        WorkMap::pm_axis(p & ((1 << c.rows_w_shift) - 1), c.src_w, c.kw, c.b, c.c, xlo, nx);
        const int jy = tap / nx, ty = ylo + jy, tx = xlo + (tap - jy * nx);
        tap = ty * c.kw + tx;
        pm_shift = ((c.c + c.b * ty) * c.src_w + (c.c + c.b * tx)) * c.pm_maps;
        pmk = true;
      }
    }
    int grp = 0, gch = 0;                          // GRP: the tile's group, its first channel in the gathered operand
    if constexpr (GRP) { grp = l_n0 / g.conv.n_group; gch = grp * g.conv.a_group; }
    const int col0 = GRP ? l_n0 - grp * g.conv.n_group : l_n0;
    const int cols = GRP ? g.conv.n_group : g.N;
    if constexpr (ROWMAP) {
      // parity-class data gradient: the tile's class decides taps, offsets and weights; its extra tap (if any) reads
      // the second gradient tensor at the row's own position
      const ConvGeom::ParityClass pc = pick_class(g.conv, (l_m0 >> g.conv.bm_shift) & 3);
      const bool extra = tap >= pc.ntaps;           // (only reached in class a2_class: k_end says so)
      step_b_cur = extra ? (size_t)16 * g.conv.ldb2 : step_b;
      int wtap = 0;
#pragma unroll
      for (int i = 0; i < LA; i++) {
        const int row = (wave + i * NW) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        if (extra) {
          pa[i] = g.conv.A2 + (size_t)arow[i] * g.conv.lda2 + gch + kin + chunk * 4;
        } else {
          const int src = parity_src_row(g.conv, pc, arow[i], tap, wtap);
          pa[i] = (src >= 0 ? g.A + (size_t)src * g.lda + gch : g.conv.zero) + kin + chunk * 4;
        }
      }
#pragma unroll
      for (int i = 0; i < LB; i++) {
        const int e = (wave + i * NW) * 256 + lane * 4;
        if (extra) pb[i] = g.conv.B2 + (size_t)(gch + kin + e / BN) * g.conv.ldb2 + min(col0 + e % BN, cols - 4);
        else pb[i] = g.B + wtap * g.conv.b_tap_stride + (size_t)(gch + kin + e / BN) * g.ldb + min(col0 + e % BN, cols - 4);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < LA; i++) {
      const int row = (wave + i * NW) * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      const int src = pmk ? arow[i] + pm_shift : conv_src_row<kCoop>(g.conv, arow[i], tap);
      pa[i] = (src >= 0 ? g.A + (size_t)src * g.lda + gch : g.conv.zero) + kin + chunk * 4;
    }
    if (BKC && pmk) {
      // K-contiguous weights [co][tap][ci]: the next tap in work is not the next in memory
#pragma unroll
      for (int i = 0; i < LB; i++) {
        const int row = (wave + i * NW) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        pb[i] = g.B + (size_t)min(l_n0 + row, g.N - 1) * g.ldb + tap * g.conv.seg + kin + chunk * 4;
      }
    }
    if (!BKC) {
      // K-outer weights [co][tap][ci]: rows = output channels (of the tile's group), columns within the tap
#pragma unroll
      for (int i = 0; i < LB; i++) {
        const int e = (wave + i * NW) * 256 + lane * 4;
        pb[i] = g.B + tap * g.conv.b_tap_stride + (size_t)(gch + kin + e / BN) * g.ldb + min(col0 + e % BN, cols - 4);
      }
    }
  };
  auto set_tile = [&](int it) __attribute__((always_inline)) {
    int m0, n0;
    if (!get_item(it, m0, n0, l_k, l_kend)) { l_valid = false; return; }
    l_n0 = n0;
    if constexpr (GRP || ROWMAP || CONV == CONV_B || (CONV == CONV_A && kCoop)) l_m0 = m0;
    if constexpr (CONV == CONV_B) {
      const int tap = n0 / g.conv.seg;
      l_ty = tap / g.conv.kw;
      l_tx = tap - l_ty * g.conv.kw;
      if constexpr (kCoop) {
        if (g.conv.pm_wgrad) {
          // position blocks: the tap reaches the map from every row of the item's block, its sources lie a constant number
          // of rows away -- no decode, no bounds test per transfer
          l_blk_end = (l_k / g.conv.pm_maps + 1) * g.conv.pm_maps;
          l_pm_shift = ((g.conv.c + g.conv.b * l_ty) * g.conv.src_w + (g.conv.c + g.conv.b * l_tx)) * g.conv.pm_maps;
        }
      }
      l_ch0 = n0 - tap * g.conv.seg + (GRP ? (m0 / g.conv.n_group) * g.conv.a_group : 0);
    }
#pragma unroll
    for (int i = 0; i < LA; i++) {
      const int q = wave + i * NW;
      if constexpr (kAp3) {        // P3 row image, as operand B below (g.lda = floats per P3 row)
        const int pos = q * 64 + lane;
        const int row = pos / 6, slot = pos - row * 6;
        const int chunk = slot ^ ((row >> 3) & 1);
        pa[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + (l_k >> 3) * 12 + chunk * 4;
      } else if (AK) {
        const int row = q * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        if constexpr (ROWMAP) {
          int cls_, rr_;
          parity_row(g.conv, min(m0 + row, g.M - 1), cls_, rr_);
          arow[i] = rr_;
        } else if (CONV == CONV_A) {
          // (tap skipping: the tile ends with its position's block of rows)
          int m_lim = g.M;
          if constexpr (kCoop) { if (g.conv.pm_skip) m_lim = min(g.M, (m0 / g.conv.pm_maps + 1) * g.conv.pm_maps); }
          arow[i] = min(m0 + row, m_lim - 1);
        }
        else pa[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + l_k + chunk * 4;
      } else {
        const int e = q * 256 + lane * 4;            // element of the [16][BM] image
        int kr = l_k + e / BM;
        // (weight gradient over maps of any size: the reduction runs to the next multiple of 16 rows; the gathered
        // operand reads zeros there, this one re-reads its last real row)
        if constexpr (CONV == CONV_B) { if (g.conv.rows_hw_shift < 0) kr = min(kr, g.conv.n_rows - 1); }
        pa[i] = g.A + (size_t)kr * g.lda + min(m0 + e % BM, g.M - 4);
      }
    }
    if (CONV == CONV_B) return;      // the gathered operand is addressed transfer by transfer (issue)
#pragma unroll
    for (int i = 0; i < LB; i++) {
      const int q = wave + i * NW;
      if constexpr (kBp3) {
        // P3 row image [row][six 16-B chunks]: LDS position p (16-B units) of the slab = row * 6 + slot; slot holds the
        // row's chunk slot ^ ((row >> 3) & 1)  (chunk = 3 * k-half + plane: conflict-free ds_read_b128 of one plane of
        // 32 consecutive rows).  g.ldb = floats per P3 row (1.5 per value), 16 values = 24 floats.
        const int pos = q * 64 + lane;
        const int row = pos / 6, slot = pos - row * 6;
        const int chunk = slot ^ ((row >> 3) & 1);
        pb[i] = g.B + (size_t)min(n0 + row, g.N - 1) * g.ldb + (l_k >> 3) * 12 + chunk * 4;
      } else if (BKC) {
        const int row = q * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        pb[i] = g.B + (size_t)min(n0 + row, g.N - 1) * g.ldb + l_k + chunk * 4;
      } else if (CONV != CONV_A) {
        const int e = q * 256 + lane * 4;
        pb[i] = g.B + (size_t)(l_k + e / BN) * g.ldb + min(n0 + e % BN, g.N - 4);
      }
    }
    if (CONV == CONV_A) retap();
  };
  // one 1-KB transfer of the cursor's slab (piece < LA: operand A, else B) into ring slot `slot`.  The LDS-DMA lands
  // at a WAVE-UNIFORM base (+ lane * 16 B); in the grouped / parity kernels hipcc keeps the slot index in a vector
  // register and would hand the base to the asm as a VGPR, so there the two bases of a slab (this wave's first A and
  // first B granule) are made scalar ONCE per slab (slab_bases) and the per-transfer offsets are constants --
  // a readfirstlane in front of every transfer halves the kernel's rate
  struct Bases { unsigned a, b; int slot; };
  constexpr bool kSplit = (C::KNOBS & KNOB_SPLIT) != 0;
  constexpr bool kScalarBases = CONV != CONV_NONE || kSplit;      // (kernels with more scalar state than hipcc keeps scalar)
  auto slab_bases = [&](int slot) __attribute__((always_inline)) -> Bases {
    Bases r;
    r.slot = slot;
    r.a = r.b = 0;
    if constexpr (kScalarBases) {
      r.a = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)(As + slot * SA + wave * 256));
      r.b = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)(Bd + slot * SB + wave * 256));
    }
    return r;
  };
  // (the dense kernels keep the per-transfer scalar arithmetic hipcc has always proven uniform there)
  auto dst_a = [&](const Bases& lb, int piece) __attribute__((always_inline)) -> unsigned {
    if constexpr (kScalarBases) return lb.a + (unsigned)(piece * NW * 1024);
    else return (unsigned)(size_t)(lds_void*)(As + lb.slot * SA + (wave + piece * NW) * 256);
  };
  auto dst_b = [&](const Bases& lb, int piece) __attribute__((always_inline)) -> unsigned {
    if constexpr (kScalarBases) return lb.b + (unsigned)(piece * NW * 1024);
    else return (unsigned)(size_t)(lds_void*)(Bd + lb.slot * SB + (wave + piece * NW) * 256);
  };
  auto issue = [&](int piece, const Bases& lb) __attribute__((always_inline)) {
    if (piece < LA) {
      const int q = wave + piece * NW;
      if (GA % NW == 0 || q < GA) glds16_at(pa[piece], dst_a(lb, piece));
    } else {
      const int q = wave + (piece - LA) * NW;
      if (GB % NW == 0 || q < GB) {
        const unsigned dst = dst_b(lb, piece - LA);
        if (CONV == CONV_B) {
          // weight gradient: row k of the K-outer operand is the source position that tap (fixed by the
          // column tile) pairs with GEMM row k; nothing to pair with -> the row of zeros
          // (tap, channel offset and group of the column / row tile: l_ty, l_tx, l_ch0 -- set once per tile in set_tile)
          const int e = q * 256 + lane * 4;
          // (position blocks: rows past the item's block belong to the next position -- they read the row of zeros)
          const int kr_b = l_k + e / BN;
          int src;
          if (kCoop && g.conv.pm_wgrad) src = kr_b < l_blk_end ? kr_b + l_pm_shift : -1;
          else src = conv_src_row_t<kCoop>(g.conv, kr_b, l_ty, l_tx);
          glds16_at((src >= 0 ? g.B + (size_t)src * g.ldb + l_ch0 : g.conv.zero) + e % BN, dst);
        } else {
          glds16_at(pb[piece - LA], dst);
        }
      }
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {             // cursor -> next slab of the stream (possibly the next tile's first)
#pragma unroll
    for (int i = 0; i < LA; i++) pa[i] += step_a;
    if (CONV != CONV_B) {
#pragma unroll
      for (int i = 0; i < LB; i++) pb[i] += ROWMAP ? step_b_cur : step_b;
    }
    l_k += BK;
    if (l_k >= l_kend) {
      l_item += 1;
      set_tile(l_item);          // (clears l_valid past the end)
    } else if (CONV == CONV_B && !AK) {
      if (g.conv.rows_hw_shift < 0 && l_k + BK > g.conv.n_rows) {       // the slab that holds the last real rows
#pragma unroll
        for (int i = 0; i < LA; i++) {
          const int e = (wave + i * NW) * 256 + lane * 4;
          pa[i] = g.A + (size_t)min(l_k + e / BM, g.conv.n_rows - 1) * g.lda + min(l_m0 + e % BM, g.M - 4);
        }
      }
    } else if (CONV == CONV_A) {
      l_kin += BK;
      if (l_kin >= g.conv.seg) retap();        // next tap: new source rows (and, K-outer weights, new tap base)
    }
  };
  auto request = [&](int slot) __attribute__((always_inline)) {     // whole slab at once (prologue)
    const Bases lb = slab_bases(slot);
#pragma unroll
    for (int i = 0; i < LA + LB; i++) issue(i, lb);
    advance();
  };

  if constexpr ((C::KNOBS & KNOB_PRIO) != 0) {
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  }
  set_tile(l_item);
  request(0);
  if (l_valid) request(1);
  if (NS == 4 && l_valid) {
    request(2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");      // slabs 0 and 1 complete, slab 2 in flight
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  ring_barrier();
  float4 xa[C::TM], xb[C::TN], na[C::TM], nb[C::TN];
  // kSplit state.  Simple schedule: the slab's first A tile (both k halves) and all of its B tiles, raw, fetched one
  // slab ahead.  Pipelined schedule: the bf16 planes of this slab's B tiles (bp) and of the A tile in work (ap), the
  // planes being formed under the MFMAs (nap: the next A tile; nbp: the NEXT slab's B tiles) and two raw operand
  // tiles in flight from LDS (rw): every region of six MFMAs splits what the region before it fetched.
  constexpr bool kPipe = kSplit && (C::KNOBS & KNOB_SPLIT_SIMPLE) == 0 && !kBp3 && !kCoop;
  constexpr int kTerms = (C::KNOBS & KNOB_BF16) != 0 ? 1 : 6;
  constexpr bool kRne = (C::KNOBS & KNOB_RNE) != 0;
  auto split_a = [&](float4 p, float4 q) __attribute__((always_inline)) -> Planes {
    return split8<kTerms, kRne>(p, q);
  };
  auto split_b = [&](float4 p, float4 q) __attribute__((always_inline)) -> Planes {
    return split8<kTerms, kRne>(p, q);
  };
  static_assert(!kPipe || (C::TM >= 2 && C::TN >= 2 && C::TN <= C::TM && (C::TM * C::TN) % 2 == 0), "split schedule");
  float4 sa0, sa1, sb0[C::TN], sb1[C::TN];
  Planes bp[C::TN], nbp[C::TN], ap, nap;
  float4 rw[2][2];
  // (pipelined) everything the first region of a slab needs, from a slab that is complete in LDS
  auto prime = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < C::TN; b++)
      bp[b] = split_b(fetch_tile<BKC, BN>(Bd + slot * SB, wn, li, lk, 0, b), fetch_tile<BKC, BN>(Bd + slot * SB, wn, li, lk, 1, b));
    ap = split_a(fetch_tile<AK, BM>(As + slot * SA, wm, li, lk, 0, 0), fetch_tile<AK, BM>(As + slot * SA, wm, li, lk, 1, 0));
    rw[0][0] = fetch_tile<AK, BM>(As + slot * SA, wm, li, lk, 0, 1);
    rw[0][1] = fetch_tile<AK, BM>(As + slot * SA, wm, li, lk, 1, 1);
  };
  // kBp3 state: the planes of the A tile in work (ap) and of the one being formed (nap, from the raw quads rw[0]),
  // the planes of two B tiles (bq: the one in work, the one in flight from LDS)
  bf16x8 bq[2][3];
  auto fetch_bplanes = [&](const float* slab, int b, bf16x8 (&d)[3]) __attribute__((always_inline)) {
    const int row = wn + b * 32 + li, sw = (row >> 3) & 1;
#pragma unroll
    for (int p = 0; p < 3; p++) d[p] = *reinterpret_cast<const bf16x8*>(slab + row * 24 + (((lk * 3 + p) ^ sw) << 2));
  };
  // planes of 32-row tile t (rows row0 + 32 t + li) of a plane image
  auto fetch_planes = [&](const float* img, int row0, int t, bf16x8 (&d)[3]) __attribute__((always_inline)) {
    const int row = row0 + t * 32 + li, sw = (row >> 3) & 1;
#pragma unroll
    for (int p = 0; p < 3; p++) d[p] = *reinterpret_cast<const bf16x8*>(img + row * 24 + (((lk * 3 + p) ^ sw) << 2));
  };
  auto prime3 = [&](int slot) __attribute__((always_inline)) {
    if constexpr (kAp3) {
      bf16x8 t[3];
      fetch_planes(As + slot * SA, wm, 0, t);
      ap.h = t[0]; ap.m = t[1]; ap.l = t[2];
    } else {
      ap = split8<6, kRne>(fetch_tile<AK, BM>(As + slot * SA, wm, li, lk, 0, 0), fetch_tile<AK, BM>(As + slot * SA, wm, li, lk, 1, 0));
    }
    fetch_bplanes(Bd + slot * SB, 0, bq[0]);
  };
  // kCoop: thread t owns operand row t of every slab (t < BM: row t of A, else row t - BM of B): it reads the row's 16
  // raw values, splits them and writes the six 16-B plane chunks of the row into the plane image `pslot`.
  //   raw image, reduction dim contiguous: the row's four swizzled 16-B chunks (rowimg_off);
  //   raw image K-major [16][ROWS]: sixteen ds_read_b32 a row pitch apart (consecutive lanes, consecutive rows)
  auto coop_read = [&](int rslot, float4 (&q)[4]) __attribute__((always_inline)) {
    const bool isA = threadIdx.x < BM;                   // (wave-uniform: BM is a multiple of 64)
    const int r = isA ? (int)threadIdx.x : (int)threadIdx.x - BM;
    const float* raw = isA ? As + rslot * SA : Bd + rslot * SB;
    const bool kcontig = isA ? AK : BKC;
    const int rows = isA ? BM : BN;
    if (kcontig) {
#pragma unroll
      for (int c = 0; c < 4; c++) q[c] = *reinterpret_cast<const float4*>(raw + rowimg_off(r, c));
    } else {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const float* p = raw + (4 * c) * rows + r;
        q[c] = make_float4(p[0], p[rows], p[2 * rows], p[3 * rows]);
      }
    }
  };
  auto coop_write = [&](int pslot, const Planes& lo, const Planes& hi) __attribute__((always_inline)) {
    const bool isA = threadIdx.x < BM;
    const int r = isA ? (int)threadIdx.x : (int)threadIdx.x - BM;
    float* d = (isA ? Pa + pslot * (BM * 24) : Pb + pslot * (BN * 24)) + r * 24;
    const int sw = (r >> 3) & 1;
    *reinterpret_cast<bf16x8*>(d + ((0 ^ sw) << 2)) = lo.h;
    *reinterpret_cast<bf16x8*>(d + ((1 ^ sw) << 2)) = lo.m;
    *reinterpret_cast<bf16x8*>(d + ((2 ^ sw) << 2)) = lo.l;
    *reinterpret_cast<bf16x8*>(d + ((3 ^ sw) << 2)) = hi.h;
    *reinterpret_cast<bf16x8*>(d + ((4 ^ sw) << 2)) = hi.m;
    *reinterpret_cast<bf16x8*>(d + ((5 ^ sw) << 2)) = hi.l;
  };
  if constexpr (kCoop) {
    float4 q[4];
    coop_read(0, q);
    coop_write(0, split8<6, kRne>(q[0], q[1]), split8<6, kRne>(q[2], q[3]));
    ring_barrier();          // plane image 0 complete; raw slot 0 free for slab 2
  } else if constexpr (kBp3) {
    prime3(0);
  } else if constexpr (kPipe) {
    prime(0);
  } else if constexpr (kSplit) {
    sa0 = fetch_tile<AK, BM>(As, wm, li, lk, 0, 0);
    sa1 = fetch_tile<AK, BM>(As, wm, li, lk, 1, 0);
#pragma unroll
    for (int b = 0; b < C::TN; b++) {
      sb0[b] = fetch_tile<BKC, BN>(Bd, wn, li, lk, 0, b);
      sb1[b] = fetch_tile<BKC, BN>(Bd, wn, li, lk, 1, b);
    }
  } else {
    fetch_group<AK, C::TM, BM>(As, wm, li, lk, 0, xa);
    fetch_group<BKC, C::TN, BN>(Bd, wn, li, lk, 0, xb);
  }
  int cur = 0;

  for (int item = 0;; item++) {
    int m0, n0, kbeg, kend;
    if (!get_item(item, m0, n0, kbeg, kend)) break;
    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; a++)
#pragma unroll
      for (int b = 0; b < C::TN; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    unsigned long long c0 = 0;
    if constexpr (Probe::on) c0 = __builtin_amdgcn_s_memtime();

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const int nxt = (cur == NSR - 1) ? 0 : cur + 1;
      // the slot iteration s-1 finished reading.  (Grouped / parity kernels: hipcc keeps `cur` in a vector register
      // there and would hand the LDS-DMA's wave-uniform base to the asm as a VGPR; ONE readfirstlane per slab puts the
      // slot back into a scalar -- forcing every transfer's address through readfirstlane halves the kernel's rate)
      // (kCoop: slab s + 2 goes where slab s was: its raw values were split during the previous iteration)
      const int nxt2 = kCoop ? cur : (cur == 0) ? NS - 1 : cur - 1;
      const Bases lbase = slab_bases(nxt2);
      // The transfers of slab s+NS-1 are issued one at a time BETWEEN the MFMA steps, not in a burst at
      // the top of the iteration: right after the barrier every wave of the CU would be issuing them
      // at once (an LDS-DMA costs its wave 60-180 cycles of issue) with no wave left to feed the
      // matrix pipe; inside the MFMA stream the partner wave on the SIMD covers each one.
      const bool feed = l_valid;         // wave-uniform
      // the ticket of the item AFTER the one the load cursor is in: never more than one undone ticket held
      // (drawing further ahead starves the other workgroups when there are only a few tiles each); it is
      // needed when the cursor leaves its item, at least two slabs from now (launch() checks the item lengths)
      const bool draw = dyn && !ended && n_fetched - (l_item - n_sk) < 2;
      // (inline assembly: hipcc waits vmcnt(0) right behind an atomicAdd() whose value it needs -- which here would
      // also drain the slab in flight, once per tile; the value is read only behind the wait that ends the iteration)
      unsigned drawn = 1u;
      if (draw && threadIdx.x == 0)
        asm volatile("global_atomic_add %0, %1, %0, off sc0" : "+v"(drawn) : "v"(ticket) : "memory");
      if constexpr ((C::KNOBS & KNOB_BURST) != 0) {
        if (feed) {
#pragma unroll
          for (int i = 0; i < NP; i++) issue(i, lbase);
        }
      }
      if constexpr (kCoop) {
        // Region (a, b) = the six MFMAs of tile pair (a, b) on planes fetched from image `cur`.  In their shadow: the B
        // planes of the next region; this thread's row of the NEXT slab -- raw reads in region 0, the two halves of the
        // split in regions 1..4, the six plane stores in regions 5 and 6 (image `nxt`); one LDS-DMA transfer of slab
        // s + 2 behind each of the first NP regions.
        constexpr int kPairs = C::TM * C::TN;
        static_assert(kPairs >= 8 && kPairs % 2 == 0 && NP <= kPairs, "cooperative split schedule");
        const float* pa_img = Pa + cur * (BM * 24);
        const float* pb_img = Pb + cur * (BN * 24);
        bf16x8 pA[3], pB[2][3];
        float4 q[4];
        Planes slo, shi;
        fetch_planes(pb_img, wn, 0, pB[0]);
#pragma unroll
        for (int a = 0; a < C::TM; a++) {
          fetch_planes(pa_img, wm, a, pA);
#pragma unroll
          for (int b = 0; b < C::TN; b++) {
            const int r = a * C::TN + b, pr = r & 1, nx = pr ^ 1;
            __builtin_amdgcn_sched_barrier(0);
            if (r + 1 < kPairs) fetch_planes(pb_img, wn, (b + 1) % C::TN, pB[nx]);
            if (r == 0) coop_read(nxt, q);
            {
              f32x16 c = acc[a][b];
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pA[2], pB[pr][0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pA[0], pB[pr][2], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pA[1], pB[pr][1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pA[1], pB[pr][0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pA[0], pB[pr][1], c, 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pA[0], pB[pr][0], c, 0, 0, 0);
            }
            if (r == 1) slo = split8<6, kRne>(q[0], q[1]);
            if (r == 3) shi = split8<6, kRne>(q[2], q[3]);
            if (r == 5) coop_write(nxt, slo, shi);
            __builtin_amdgcn_sched_barrier(0);
            if (r < NP) {
              if (feed) issue(r, lbase);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      } else if constexpr (kBp3) {
        // Region (a, b) = the six MFMAs of tile pair (a, b).  In their shadow: the three plane reads of the NEXT region's B
        // tile (tile 0 of the next slab behind the last region: complete in LDS since the last barrier); in region (a, 0)
        // the raw quads of the next A tile (tile a + 1, or tile 0 of the next slab), split in regions (a, 1) and (a, 2):
        // 88 vector instructions per 48 MFMAs against 264 when every wave splits both operands.
        constexpr int kPairs = C::TM * C::TN;
        static_assert(C::TN >= 3, "the A split is spread over regions (a, 1) and (a, 2)");
        const float* a_cur = As + cur * SA;
        const float* a_nxt = As + nxt * SA;
        const float* b_cur = Bd + cur * SB;
        const float* b_nxt = Bd + nxt * SB;
        u32x4 nh, nm, nl;
        bf16x8 naq[3];            // kAp3: the next A tile's planes, fetched ready-made
#pragma unroll
        for (int a = 0; a < C::TM; a++) {
#pragma unroll
          for (int b = 0; b < C::TN; b++) {
            const int r = a * C::TN + b, pr = r & 1, nx = pr ^ 1;
            __builtin_amdgcn_sched_barrier(0);
            fetch_bplanes(r + 1 == kPairs ? b_nxt : b_cur, (b + 1) % C::TN, bq[nx]);
            if (b == 0) {
              const float* src = a + 1 < C::TM ? a_cur : a_nxt;
              if constexpr (kAp3) {
                fetch_planes(src, wm, a + 1 < C::TM ? a + 1 : 0, naq);
              } else {
                rw[0][0] = fetch_tile<AK, BM>(src, wm, li, lk, 0, a + 1 < C::TM ? a + 1 : 0);
                rw[0][1] = fetch_tile<AK, BM>(src, wm, li, lk, 1, a + 1 < C::TM ? a + 1 : 0);
              }
            }
            {
              f32x16 c = acc[a][b];
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap.l, bq[pr][0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap.h, bq[pr][2], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap.m, bq[pr][1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap.m, bq[pr][0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap.h, bq[pr][1], c, 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap.h, bq[pr][0], c, 0, 0, 0);
            }
            if constexpr (kAp3) {
            } else if (b == 1) {
              unsigned x, y, z;
              split2<kRne>(rw[0][0].x, rw[0][0].y, x, y, z); nh[0] = x; nm[0] = y; nl[0] = z;
              split2<kRne>(rw[0][0].z, rw[0][0].w, x, y, z); nh[1] = x; nm[1] = y; nl[1] = z;
            } else if (b == 2) {
              unsigned x, y, z;
              split2<kRne>(rw[0][1].x, rw[0][1].y, x, y, z); nh[2] = x; nm[2] = y; nl[2] = z;
              split2<kRne>(rw[0][1].z, rw[0][1].w, x, y, z); nh[3] = x; nm[3] = y; nl[3] = z;
            }
            __builtin_amdgcn_sched_barrier(0);
            // one transfer of slab s + NS - 1 behind each of the first NP regions: with one workgroup per CU nothing else
            // covers the wait for a transfer issued late in the iteration
            if constexpr ((C::KNOBS & KNOB_BURST) == 0) {
              static_assert(NP <= kPairs, "one transfer per region");
              if (r < NP) {
                if (feed) issue(r, lbase);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          if constexpr (kAp3) {
            ap.h = naq[0]; ap.m = naq[1]; ap.l = naq[2];
          } else {
            ap.h = __builtin_bit_cast(bf16x8, nh);
            ap.m = __builtin_bit_cast(bf16x8, nm);
            ap.l = __builtin_bit_cast(bf16x8, nl);
          }
        }
        // (kPairs is even: the planes fetched in the last region sit in bq[0], where the next slab starts)
        static_assert(kPairs % 2 == 0, "B plane double buffer parity");
      } else if constexpr (kPipe) {
        // Region (a, b) = the six MFMAs of tile pair (a, b), with in their shadow: the split of the raw tile the PREVIOUS region fetched, and the ds_reads of the raw
        // tile the NEXT region splits.  Splits: region (a, 0): A tile a+1 of this slab -- in the last tile: A tile 0
        // of the next slab; region (a, 1): B tile a of the next slab (a < TN).  The next slab of the stream (possibly
        // the first of the next tile) is complete in LDS since the last barrier.
        constexpr int kPairs = C::TM * C::TN;
        const float* a_cur = As + cur * SA;
        const float* a_nxt = As + nxt * SA;
        const float* b_nxt = Bd + nxt * SB;
#pragma unroll
        for (int a = 0; a < C::TM; a++) {
#pragma unroll
          for (int b = 0; b < C::TN; b++) {
            const int r = a * C::TN + b, pr = r & 1, nx = pr ^ 1;
            __builtin_amdgcn_sched_barrier(0);
            // ---- the raw operand tile region r + 1 splits (hipcc is free to place these reads and the split below
            // between the region's MFMAs: measured 1-4 % faster than fencing them in front of / behind the chain)
            if (b == 0) {                                   // next region (a, 1) splits B tile a of the next slab
              if (a < C::TN) {
                rw[nx][0] = fetch_tile<BKC, BN>(b_nxt, wn, li, lk, 0, a);
                rw[nx][1] = fetch_tile<BKC, BN>(b_nxt, wn, li, lk, 1, a);
              }
            } else if (b == 1) {                            // next region (a + 1, 0) splits an A tile
              if (a + 2 < C::TM) {                          // ... A tile a + 2 of this slab
                rw[nx][0] = fetch_tile<AK, BM>(a_cur, wm, li, lk, 0, a + 2);
                rw[nx][1] = fetch_tile<AK, BM>(a_cur, wm, li, lk, 1, a + 2);
              } else if (a + 2 == C::TM) {                  // ... A tile 0 of the next slab
                rw[nx][0] = fetch_tile<AK, BM>(a_nxt, wm, li, lk, 0, 0);
                rw[nx][1] = fetch_tile<AK, BM>(a_nxt, wm, li, lk, 1, 0);
              } else {                                      // last tile: A tile 1 of the next slab, for ITS region (0, 0)
                rw[nx][0] = fetch_tile<AK, BM>(a_nxt, wm, li, lk, 0, 1);
                rw[nx][1] = fetch_tile<AK, BM>(a_nxt, wm, li, lk, 1, 1);
              }
            }
            // ---- the region's MFMAs and the split of what the previous region fetched
            acc[a][b] = mfma_split<kTerms>(ap, bp[b], acc[a][b]);
            if (b == 0) nap = split_a(rw[pr][0], rw[pr][1]);
            else if (b == 1 && a < C::TN) nbp[a] = split_b(rw[pr][0], rw[pr][1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((C::KNOBS & KNOB_BURST) == 0) {
#pragma unroll
              for (int piece = r * NP / kPairs; piece < (r + 1) * NP / kPairs; piece++) {
                if (feed) issue(piece, lbase);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          ap = nap;
        }
#pragma unroll
        for (int b = 0; b < C::TN; b++) bp[b] = nbp[b];
      } else if constexpr (kSplit) {
        constexpr int kPairs = C::TM * C::TN;
        Planes bp[C::TN];
#pragma unroll
        for (int b = 0; b < C::TN; b++) bp[b] = split8(sb0[b], sb1[b]);
        float4 ca0 = sa0, ca1 = sa1;
#pragma unroll
        for (int a = 0; a < C::TM; a++) {
          const Planes ap = split8(ca0, ca1);
          // the operands of the next A tile -- or, behind the last one, of the next slab of the stream (complete in
          // LDS since the last barrier; possibly the first slab of the next tile) -- arrive under this tile's MFMAs
          if (a + 1 < C::TM) {
            ca0 = fetch_tile<AK, BM>(As + cur * SA, wm, li, lk, 0, a + 1);
            ca1 = fetch_tile<AK, BM>(As + cur * SA, wm, li, lk, 1, a + 1);
          } else {
            sa0 = fetch_tile<AK, BM>(As + nxt * SA, wm, li, lk, 0, 0);
            sa1 = fetch_tile<AK, BM>(As + nxt * SA, wm, li, lk, 1, 0);
#pragma unroll
            for (int b = 0; b < C::TN; b++) {
              sb0[b] = fetch_tile<BKC, BN>(Bd + nxt * SB, wn, li, lk, 0, b);
              sb1[b] = fetch_tile<BKC, BN>(Bd + nxt * SB, wn, li, lk, 1, b);
            }
          }
#pragma unroll
          for (int b = 0; b < C::TN; b++) {
            // (lab form: every split in front of its tile pair's MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            acc[a][b] = mfma_split(ap, bp[b], acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((C::KNOBS & KNOB_BURST) == 0) {
              const int pair = a * C::TN + b;      // the slab's NP transfers, spread over the tile pairs
#pragma unroll
              for (int piece = pair * NP / kPairs; piece < (pair + 1) * NP / kPairs; piece++) {
                __builtin_amdgcn_sched_barrier(0);
                if (feed) issue(piece, lbase);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        }
      } else {
#pragma unroll
      for (int grp = 0; grp < 2; grp++) {
        // next group's operands: second half of this slab, then the first half of the NEXT slab of the
        // stream (complete in LDS since the last barrier; possibly the first slab of the next tile)
        const float* an_ = As + (grp == 0 ? cur : nxt) * SA;
        const float* bn_ = Bd + (grp == 0 ? cur : nxt) * SB;
        constexpr bool kSpread = (C::KNOBS & KNOB_SPREAD) != 0;
        constexpr int kPieces = C::TM + C::TN;
        if constexpr (!kSpread) {
          __builtin_amdgcn_sched_barrier(0);
          fetch_group<AK, C::TM, BM>(an_, wm, li, lk, grp == 0 ? 1 : 0, na);
          fetch_group<BKC, C::TN, BN>(bn_, wn, li, lk, grp == 0 ? 1 : 0, nb);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if constexpr (kSpread) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pc = j * kPieces / 4; pc < (j + 1) * kPieces / 4; pc++) {
              if (pc < C::TM) na[pc] = fetch_tile<AK, BM>(an_, wm, li, lk, grp == 0 ? 1 : 0, pc);
              else nb[pc - C::TM] = fetch_tile<BKC, BN>(bn_, wn, li, lk, grp == 0 ? 1 : 0, pc - C::TM);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int a = 0; a < C::TM; a++)
#pragma unroll
            for (int b = 0; b < C::TN; b++) {
              const float fa = j == 0 ? xa[a].x : j == 1 ? xa[a].y : j == 2 ? xa[a].z : xa[a].w;
              const float fb = j == 0 ? xb[b].x : j == 1 ? xb[b].y : j == 2 ? xb[b].z : xb[b].w;
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a][b], 0, 0, 0);
            }
          const int piece = grp * 4 + j;       // one transfer behind each of the first NP MFMA steps
          if ((C::KNOBS & KNOB_BURST) == 0 && piece < NP) {
            __builtin_amdgcn_sched_barrier(0);
            if (feed) issue(piece, lbase);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int a = 0; a < C::TM; a++) xa[a] = na[a];
#pragma unroll
        for (int b = 0; b < C::TN; b++) xb[b] = nb[b];
      }
      }
      if (feed) advance();
      __builtin_amdgcn_sched_barrier(0);
      // the slab the NEXT iteration prefetches from (s+2) must be complete behind the barrier: with a ring
      // of 3 it is the one requested in this iteration (wait for everything); with 4 it was requested one
      // iteration ago and the one requested now stays in flight (counted wait) -- unless nothing was
      // requested now (end of the stream)
      auto slab_wait = [&]() __attribute__((always_inline)) {
        if (NS == 4 && feed) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      };
      auto land = [&]() __attribute__((always_inline)) {              // behind the slab wait, in front of the barrier
        if (draw) {
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(drawn) : : "memory");     // (the ticket: with a ring of 4 the slab wait is counted)
          if (threadIdx.x == 0) idq[n_fetched & 7] = (int)drawn < lim_dp ? (int)drawn : -1;
        }
      };
      auto landed = [&]() __attribute__((always_inline)) {            // behind the barrier: every wave learns the ticket
        if (draw) { ended = idq[n_fetched & 7] < 0; n_fetched++; }
      };
      if constexpr (Probe::on) {
        const unsigned long long s0 = __builtin_amdgcn_s_memtime();
        slab_wait();
        land();
        const unsigned long long s1 = __builtin_amdgcn_s_memtime();
        ring_barrier();
        const unsigned long long s2 = __builtin_amdgcn_s_memtime();
        c_wait += s1 - s0;
        c_bar += s2 - s1;
        n_slab++;
      } else {
        slab_wait();
        land();
        ring_barrier();
      }
      landed();
      cur = nxt;
    }
    if constexpr (Probe::on) { c_loop += __builtin_amdgcn_s_memtime() - c0; n_tile++; }
    bool finish = true;          // this workgroup writes the tile
    // (the item's whole reduction: g.K, or with tap skipping its position's taps)
    int k_full = g.K;
    if constexpr (CONV == CONV_A && kCoop) {
      if (g.conv.pm_skip) { int a_, b_, c_; wmap.template decode<true>(g, base + g_i0 + skA_tile + ((n_sk == 2 && item == 0) ? 1 : 0), BM, BN, a_, b_, c_, k_full); }
    }
    if (EPI != EPI_ATOMIC && item < n_sk && (kbeg != 0 || kend != k_full)) {
      // Inter-workgroup hand-off in the write-through form: every byte of a partial tile is stored sc1
      // and read with sc1 loads (per-XCD L2s are not coherent), the flag is an agent-scope word.
      if (kend != k_full) {                        // does not end its tile: publish
        finish = false;
        __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(
            g.sk_ws + (size_t)blockIdx.x * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
        for (int a = 0; a < C::TM; a++)
#pragma unroll
          for (int b = 0; b < C::TN; b++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
              u32x4 v;
              v.x = __float_as_uint(acc[a][b][4 * q]); v.y = __float_as_uint(acc[a][b][4 * q + 1]);
              v.z = __float_as_uint(acc[a][b][4 * q + 2]); v.w = __float_as_uint(acc[a][b][4 * q + 3]);
              __builtin_amdgcn_raw_buffer_store_b128(v, ws, (int)threadIdx.x * 16, ((a * C::TN + b) * 4 + q) * C::NT * 16, 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains (also the slabs in flight)
        ring_barrier();
        if (threadIdx.x == 0)
          __hip_atomic_store(g.sk_flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {                                     // ends its tile: gather the runs before it, in workgroup order
        const int tile_start = skA_tile * ns;      // (an owned stream-K piece is always piece A)
        for (int jj = j - g_j0 - 1; jj >= 0; jj--) {      // (runs of this workgroup's group)
          if ((int)((long long)(jj + 1) * sk_total / sk_w) <= tile_start) break;      // run jj ends before this tile
          const int peer = (g_j0 + jj) * AIT_NXCD + xcd;
          if (threadIdx.x == 0) {
            while (__hip_atomic_load(g.sk_flags + peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
              __builtin_amdgcn_s_sleep(2);
            __hip_atomic_store(g.sk_flags + peer, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          ring_barrier();
          __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(
              g.sk_ws + (size_t)peer * (BM * BN), 0, BM * BN * 4, 0x00020000);
          // 16 loads in flight per thread (one round trip per half tile, not one per MFMA tile)
          constexpr int NQ = C::TM * C::TN * 4, QB = NQ < 16 ? NQ : 16;
          static_assert(NQ % QB == 0, "partial tile in whole batches");
#pragma unroll
          for (int q0 = 0; q0 < NQ; q0 += QB) {
            u32x4 t[QB];
#pragma unroll
            for (int i = 0; i < QB; i++)
              t[i] = __builtin_amdgcn_raw_buffer_load_b128(ws, (int)threadIdx.x * 16, (q0 + i) * C::NT * 16, 16);
#pragma unroll
            for (int i = 0; i < QB; i++) {
              const int idx = q0 + i, ab = idx >> 2, q = idx & 3;
              f32x16& d = acc[ab / C::TN][ab % C::TN];
              d[4 * q] += __uint_as_float(t[i].x); d[4 * q + 1] += __uint_as_float(t[i].y);
              d[4 * q + 2] += __uint_as_float(t[i].z); d[4 * q + 3] += __uint_as_float(t[i].w);
            }
          }
        }
      }
    }
    if (finish) {
      int m_lim = -1;
      if constexpr (CONV == CONV_A && !ROWMAP && !GRP && kCoop) {
        // (tap skipping: a tile ends with its position's block of rows -- the rows behind it are another position's)
        if (g.conv.pm_skip) m_lim = min(g.M, (m0 / g.conv.pm_maps + 1) * g.conv.pm_maps);
      }
      epilogue<C::TM, C::TN, EPI, ROWMAP>(acc, g, m0, n0, wm, wn, li, lk, m_lim);
    }
    // (the operand quads are dead across the epilogue -- its address arithmetic needs the registers --
    // and are fetched again from the next tile's first slab, complete in LDS since the last barrier)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kCoop) {
      // (operands are fetched from the plane image at the top of every iteration: nothing to re-fetch)
    } else if constexpr (kBp3) {
      prime3(cur);
    } else if constexpr (kPipe) {
      prime(cur);
    } else if constexpr (!kSplit) {
      fetch_group<AK, C::TM, BM>(As + cur * SA, wm, li, lk, 0, xa);
      fetch_group<BKC, C::TN, BN>(Bd + cur * SB, wn, li, lk, 0, xb);
    }
  }

  leave();
  if constexpr (Probe::on) {
    if (threadIdx.x == 0 && g.probe) {
      unsigned long long* p = g.probe + (size_t)blockIdx.x * AIT_PROBE_WORDS;
      p[0] = t_start;
      p[1] = __builtin_amdgcn_s_memrealtime();
      p[2] = c_loop; p[3] = c_wait; p[4] = c_bar; p[5] = n_slab; p[6] = n_tile;
      p[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32) |
             (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // XCC_ID | HW_ID
      p[8] = c_start;
      p[9] = __builtin_amdgcn_s_memtime();
    }
    if (lane == 0 && g.probe && wave < 8) {      // every wave's own split of its slab loop: loop, vmcnt wait, barrier
      unsigned long long* p = g.probe + (size_t)blockIdx.x * AIT_PROBE_WORDS + 10 + wave * 3;
      p[0] = c_loop; p[1] = c_wait; p[2] = c_bar;
    }
  }
}

// =========================================================================================================
// Register-staged kernels (MODE_DB / MODE_RING): one workgroup per work item.
// =========================================================================================================
template <class C, bool AK, bool BKC, int EPI>
__global__ __launch_bounds__(C::NT, C::MINW) void gemm_f32_kernel(const GemmArgs g_in) {
  static_assert(C::MODE != MODE_DLDS, "the direct-to-LDS tiles run gemm_f32_stream_kernel");
  GemmArgs g = g_in;
  if (g.batch > 1 || g.batch2 > 1) {     // batched launch: problem (blockIdx.y, blockIdx.z)
    g.A += (size_t)blockIdx.y * g.sA + (size_t)blockIdx.z * g.sA2;
    g.B += (size_t)blockIdx.y * g.sB + (size_t)blockIdx.z * g.sB2;
    g.C += (size_t)blockIdx.y * g.sC + (size_t)blockIdx.z * g.sC2;
    if (g.residual) g.residual += (size_t)blockIdx.y * g.sC + (size_t)blockIdx.z * g.sC2;
  }
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                              // [NBUF][BK][PA]
  float* Bs = lds + C::NBUF * BK * C::PA;       // [NBUF][BK][PB]

  WorkMap wmap;
  wmap.init(g, BM, BN);
  const int xcd = blockIdx.x % AIT_NXCD, j = blockIdx.x / AIT_NXCD;
  const int id = xcd * wmap.chunk + j;
  if (j >= wmap.chunk || id >= wmap.items) return;   // bijective map: ids past the end simply exit
  int m0, n0, kbeg, kend;
  wmap.decode(g, id, BM, BN, m0, n0, kbeg, kend);
  if (kbeg >= kend && g.K > 0) return;      // (K == 0: the epilogue of a zero accumulator)

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
  if constexpr (C::MODE == MODE_RING) {
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
      const bool more2 = k0 + 2 * BK < kend;
      if (more2) {
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
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more2) {
        store_slab<AK, BM, BK, C::NT, C::VA, C::PA>(As + nxt2 * BK * C::PA, ra);
        store_slab<BKC, BN, BK, C::NT, C::VB, C::PB>(Bs + nxt2 * BK * C::PB, rb);
      }
      __syncthreads();
      cur = nxt;
    }
  } else {
    // ---- register-staged double buffer ----------------------------------------------------------
    load_slab<AK, BM, BK, C::NT, C::VA>(g.A, g.lda, m0, g.M, kbeg, kend, ra);
    load_slab<BKC, BN, BK, C::NT, C::VB>(g.B, g.ldb, n0, g.N, kbeg, kend, rb);
    store_slab<AK, BM, BK, C::NT, C::VA, C::PA>(As, ra);
    store_slab<BKC, BN, BK, C::NT, C::VB, C::PB>(Bs, rb);
    __syncthreads();

    int cur = 0;
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
        __builtin_amdgcn_sched_barrier(0);
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

// ---- scheduler scratch: CALLER-OWNED (include/ait_hip.h, ait_launch_ctx) ------------------------------------
// Layout of the workspace: [0, kCtlBytes) control words -- per XCD a ticket counter and an exit counter a line apart
// (the dynamic hand-out of whole tiles), then one "partial published" flag per workgroup -- followed by one BM x BN
// partial tile per workgroup (stream-K).  The control words are zeroed ONCE by ait_gemm_workspace_init(); every
// word a launch sets is cleared again by the workgroup that consumes it, so launches that are ordered on one
// stream share a workspace.  No workspace (NULL): static work lists, whole tiles only -- nothing is allocated here.
constexpr size_t kSchedBytes = AIT_NXCD * 32 * sizeof(unsigned);      // 1 KiB
constexpr int kMaxSlots = 2048;                                       // workgroups of one persistent launch
constexpr size_t kCtlBytes = 16384;                                   // >= kSchedBytes + kMaxSlots * 4
static_assert(kCtlBytes == ait_ws::kCtlBytes, "gemm_internal.h ait_ws: the bf16 kernel's partial tiles start behind the control words too");
struct SchedWs {
  void* p = nullptr;
  size_t bytes = 0;
  unsigned* sched() const { return static_cast<unsigned*>(p); }
  unsigned* flags() const { return reinterpret_cast<unsigned*>(static_cast<char*>(p) + kSchedBytes); }
  float* partials() const { return reinterpret_cast<float*>(static_cast<char*>(p) + kCtlBytes); }
};
inline SchedWs sched_ws_of(const ait_launch_ctx* ctx) {
  SchedWs w;
  if (ctx && ctx->sched_ws) { w.p = ctx->sched_ws; w.bytes = ctx->sched_ws_bytes; }
  return w;
}

// Resident workgroup slots of the persistent kernel: what the occupancy query admits for THIS kernel (registers,
// LDS ring, waves), not a guess from the LDS size; memoised per instantiation (a constant of the code object and
// the device model).  A failed query falls back to the LDS-derived figure for MI355X.
template <class C>
inline int stream_slots(const void* kern) {
  static std::atomic<int> memo{0};
  int v = memo.load(std::memory_order_relaxed);
  if (v > 0) return v;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, C::NT, C::LDS) != hipSuccess || per_cu <= 0) {
    (void)hipGetLastError();
    per_cu = (int)(160 * 1024 / C::LDS);
    const int by_waves = 32 / (C::NT / 64);
    if (per_cu > by_waves) per_cu = by_waves;
  }
  if (per_cu > 4) per_cu = 4;
  v = (per_cu < 1 ? 1 : per_cu) * cus;
  if (v > kMaxSlots) v = kMaxSlots;
  memo.store(v, std::memory_order_relaxed);
  return v;
}

template <class C, bool AK, bool BKC, int EPI, class Probe = NoProbe, int CONV = CONV_NONE, bool GRP = false, bool ROWMAP = false>
int launch(const GemmArgs& g, hipStream_t s, const SchedWs& ws = SchedWs(), int slots = 0) {
  GemmArgs gl = g;
  gl.sk_on = 0;
  gl.sk_ws = nullptr;
  gl.sk_flags = nullptr;
  gl.sched = nullptr;
  WorkMap wmap;
  wmap.init(g, C::BM, C::BN);
  unsigned blocks;
  const void* kern;
  if constexpr (C::MODE == MODE_DLDS) {
    kern = reinterpret_cast<const void*>(gemm_f32_stream_kernel<C, AK, BKC, EPI, Probe, CONV, GRP, ROWMAP>);
    if (C::LDS > 64 * 1024 &&
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS) != hipSuccess)
      return AIT_ELAUNCH;
    // persistent: W workgroups per XCD, every one of them gets work (W <= items of the smallest chunk
    // is not required: a workgroup past its chunk's end exits at once)
    if (slots <= 0) slots = stream_slots<C>(kern);
    int w = max(1, min(wmap.chunk, slots / AIT_NXCD));
    // stream-K for the tiles of an under-filled last round (see the kernel): worth it unless that round is
    // nearly full anyway; needs whole 16-float slabs and a reduction long enough to cut
    // Cost model in slab-times (measured on the lab shapes, scripts/gemm_lab.hip sweep): whole tiles cost
    // the last round one tile (K/16 slabs) -- about 0.62 of one when at most half the slots are busy, a
    // workgroup alone on its CU runs faster -- stream-K costs rem/W of a tile plus ~9 slab-times of
    // publishing and gathering partial tiles.
    const int wfull = max(1, slots / AIT_NXCD), rem = wmap.chunk % wfull;
    // (the 0.62 holds for tiles of which a CU holds two or more: the lone workgroup of an under-filled round has the
    // CU to itself.  A tile that fills the CU alone -- the 256 x 256 ones, > 80 KB of LDS -- gains nothing from idle
    // neighbours: its last round costs a whole tile however few workgroups are in it)
    const double last_round = (rem * 2 <= wfull && (ait_lab::Knobs::old_last_round || C::LDS <= 80 * 1024)) ? 0.62 : 1.0;
    const int item_slabs = (g.splits > 1 ? g.k_per_split : g.K) / 16;
    if (EPI == EPI_ATOMIC) {
      // split-K launches (weight gradients): a piece of an item just ADDS its partial tile like a whole item does --
      // no scratch, no hand-off; costs one more atomic epilogue (~4 slab-times) per cut.  Needs equal splits.
      const bool equal = g.conv.pm_wgrad || (g.K % 16 == 0 && (g.splits == 1 || (g.k_per_split % 16 == 0 && g.K == g.splits * g.k_per_split)));
      if (equal && rem > 0 && item_slabs * (last_round - (double)rem / wfull) > 6.0) {
        gl.sk_on = 1;
        w = wfull;
      }
    }
    // (12 slab-times: what a stream-K cut of the last round must save to pay for its hand-off)
    const bool sk_pays = item_slabs * (last_round - (double)rem / wfull) > ait_lab::Knobs::sk_pays;
    if (g.conv.pm_skip) {
      // tap skipping: every item is cut (the kernel's two-group scheme); the caller checked pm_skip_fits()
      if (!ws.p || wmap.chunk > wfull || ws.bytes < kCtlBytes + (size_t)wfull * AIT_NXCD * C::BM * C::BN * sizeof(float))
        return AIT_EWORKSPACE;
      gl.sk_on = 1;
      gl.sk_ws = ws.partials();
      gl.sk_flags = ws.flags();
      blocks = (unsigned)(wfull * AIT_NXCD);
      hipLaunchKernelGGL((gemm_f32_stream_kernel<C, AK, BKC, EPI, Probe, CONV, GRP, ROWMAP>), dim3(blocks), dim3(C::NT), C::LDS, s, gl);
      AIT_CHECK_LAUNCH();
      return AIT_OK;
    }
    if (ws.p) {
      if (ws.bytes < kCtlBytes) return AIT_EWORKSPACE;
      // ticket counters for the dynamic hand-out of whole tiles (every launch), partial tiles + flags when this
      // launch also cuts its last round
      // (a ticket lands one iteration after it is drawn, and the prologue fills the slab ring from the first
      // item alone: every whole item must outlast both -- eight slabs is comfortably more than either)
      const int kps = g.splits > 1 ? g.k_per_split : g.K, klast = g.K - (g.splits - 1) * kps;
      int kmin = kps < klast ? kps : klast;
      if constexpr (ROWMAP) {
        for (int c = 0; c < 4; c++) kmin = g.conv.cls[c].k_end < kmin ? g.conv.cls[c].k_end : kmin;
      }
      if (kmin >= 128 && (C::KNOBS & KNOB_NOTICKET) == 0) gl.sched = ws.sched();
      if (EPI != EPI_ATOMIC && !ROWMAP && g.splits == 1 && g.K % 16 == 0 && rem > 0 && sk_pays) {
        if (ws.bytes < kCtlBytes + (size_t)wfull * AIT_NXCD * C::BM * C::BN * sizeof(float)) return AIT_EWORKSPACE;
        gl.sk_on = 1;
        gl.sk_ws = ws.partials();
        gl.sk_flags = ws.flags();
        w = wfull;
      }
    }
    blocks = (unsigned)(w * AIT_NXCD);
  } else {
    blocks = (unsigned)(wmap.chunk * AIT_NXCD);
    kern = reinterpret_cast<const void*>(gemm_f32_kernel<C, AK, BKC, EPI>);
    if (C::LDS > 64 * 1024 &&
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS) != hipSuccess)
      return AIT_ELAUNCH;
  }
  if constexpr (C::MODE == MODE_DLDS)
    hipLaunchKernelGGL((gemm_f32_stream_kernel<C, AK, BKC, EPI, Probe, CONV, GRP, ROWMAP>), dim3(blocks), dim3(C::NT), C::LDS, s, gl);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<C, AK, BKC, EPI>), dim3(blocks, g.batch > 1 ? g.batch : 1, g.batch2 > 1 ? g.batch2 : 1),
                       dim3(C::NT), C::LDS, s, g);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

template <class C, int EPI>
int dispatch_layout(const GemmArgs& g, bool ak, bool bk, hipStream_t s, const SchedWs& ws) {
  if (ak && bk) return launch<C, true, true, EPI>(g, s, ws);
  if (ak && !bk) return launch<C, true, false, EPI>(g, s, ws);
  if (!ak && bk) return launch<C, false, true, EPI>(g, s, ws);
  return launch<C, false, false, EPI>(g, s, ws);
}

template <class C>
int dispatch(const GemmArgs& g, bool ak, bool bk, hipStream_t s, const SchedWs& ws = SchedWs()) {
  if (g.flags & AIT_GEMM_ATOMIC) return dispatch_layout<C, EPI_ATOMIC>(g, ak, bk, s, ws);
  const bool row_bias = g.bias && (g.flags & AIT_GEMM_BIAS_ROW);
  if (g.gate) {        // "+ residual", then zeroed where gate <= 0 (library-internal callers: csrc/tail.hip)
    if (!g.residual || (g.flags & (AIT_GEMM_ACCUMULATE | AIT_GEMM_MASK_POS)) || row_bias) return AIT_EINVAL;
    return dispatch_layout<C, EPI_RESG>(g, ak, bk, s, ws);
  }
  if (g.residual && !(g.flags & AIT_GEMM_ACCUMULATE) && !row_bias) return dispatch_layout<C, EPI_RES>(g, ak, bk, s, ws);
  if (g.residual || (g.flags & (AIT_GEMM_ACCUMULATE | AIT_GEMM_MASK_POS)) || row_bias)
    return dispatch_layout<C, EPI_AUX>(g, ak, bk, s, ws);
  return dispatch_layout<C, EPI_STORE>(g, ak, bk, s, ws);
}

// Validate arguments and fill GemmArgs (shared by the product entry point and the lab harness).
// K == 0 is legal (an empty reduction): the product is the epilogue of a zero accumulator; it is
// run as one slab of zeros by the register-staged kernels (load_slab returns zeros past Kend).
inline int make_args(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A,
                     int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                     const float* residual, int flags, int split_k, int c_colblk,
                     long long c_batch_stride, int BK, GemmArgs& g) {
  if (M < 0 || N < 0 || K < 0) return AIT_EINVAL;
  if (!C || (K > 0 && (!A || !B))) return AIT_EINVAL;     // (K == 0 never reads A or B)
  // float4 staging: row pitches and bases 16-B aligned; K % 4 only matters for an operand whose
  // reduction dimension is the contiguous one
  // (K % 4 != 0 with a K-contiguous A and a K-outer B: the last 16-byte load of an A row runs into the
  // row's padding -- lda >= K rounded up to 4, padding FINITE -- and meets rows of B that read as zero)
  const bool k_tail_ok = !(K & 3) || (trans_a && !trans_b) || (!trans_a && !trans_b && lda >= ((K + 3) & ~3));
  if ((lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
      (reinterpret_cast<uintptr_t>(B) & 15) || !k_tail_ok)
    return AIT_EUNSUPPORTED;
  if (split_k < 1) split_k = 1;
  if (split_k > 1 && !(flags & AIT_GEMM_ATOMIC)) return AIT_EINVAL;
  if ((flags & AIT_GEMM_ATOMIC) && (bias || residual || (flags & AIT_GEMM_RELU)))
    return AIT_EINVAL;
  if ((flags & AIT_GEMM_MASK_POS) && !residual) return AIT_EINVAL;
  if ((flags & AIT_GEMM_COLSUM) && (!bias || c_colblk > 0 || (flags & (AIT_GEMM_ATOMIC | AIT_GEMM_BIAS_ROW | AIT_GEMM_ACCUMULATE))))
    return AIT_EINVAL;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.residual = residual;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.c_colblk = c_colblk; g.c_batch = c_batch_stride; g.alpha = alpha; g.flags = flags;
  g.probe = nullptr;
  g.gate = nullptr;
  g.sk_on = 0;
  g.sk_ws = nullptr;
  g.sk_flags = nullptr;
  g.sched = nullptr;
  g.conv = ConvGeom{};
  g.batch = g.batch2 = 1; g.sA = g.sB = g.sC = g.sA2 = g.sB2 = g.sC2 = 0;
  // 32-bit element offsets in the epilogue
  {
    const unsigned long long rows = (unsigned long long)(M > 0 ? M - 1 : 0) * (unsigned long long)(ldc > 0 ? ldc : 0);
    const unsigned long long cols = c_colblk > 0 ? (unsigned long long)((N - 1) / c_colblk) * (unsigned long long)c_batch_stride + c_colblk
                                                 : (unsigned long long)N;
    if (ldc < 0 || c_batch_stride < 0 || rows + cols >= (1ull << 31)) return AIT_EUNSUPPORTED;
  }
  int kps = (K + split_k - 1) / split_k;
  kps = (kps + BK - 1) / BK * BK;
  if (kps < BK) kps = BK;                 // K == 0: one (all-zero) slab, one split
  g.k_per_split = kps;
  g.splits = K > 0 ? (K + kps - 1) / kps : 1;
  return AIT_OK;
}

}  // namespace ait_gemm
