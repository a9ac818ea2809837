// ait_amd/csrc/conv_f32.hip -- channels-last convolutions as implicit GEMMs on the persistent kernel of gemm_f32_impl.h
// (a translation unit of its own: the instantiations compile beside those of gemm_f32.hip).
#include "gemm_f32_impl.h"
#include "gemm_internal.h"
#include <type_traits>

namespace {
using namespace ait_gemm;
// the persistent product tile of gemm_f32.hip (f32 products split onto the bf16 matrix pipe) and its bf16 form
using Tile256D = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_RNE>;
using Tile256B = Cfg<256, 128, 16, 2, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_BF16>;
}  // namespace

// =========================================================================================================
// Convolutions over channels-last maps as implicit GEMMs on the same persistent kernel (the
// 3x3 convolutions of RCNN_top / layer4, resnet_sys_transformer_sk_dilat.py:85-111,482-491).  No im2col
// buffer: the LDS-DMA of the gathered operand takes its per-lane source address from the window geometry
// (ConvGeom, gemm_f32_impl.h); positions outside the map read a caller-provided row of zeros.
// =========================================================================================================
namespace {
using Tile128D = Cfg<128, 128, 16, 2, 2, 2, MODE_DLDS, 4, KNOB_SPLIT | KNOB_RNE>;
// 64 x 128, two waves: grouped convolutions over a handful of rows (the query side of the SK block: 64 output rows) --
// a 256-row tile would multiply three quarters of padding there
using TileS = Cfg<64, 128, 16, 1, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_RNE>;
// the same three tiles with the operands rounded to bf16 and one MFMA per block (AIT_CTX_BF16)
// 256 x 256, every operand value split once per workgroup (gemm_f32.hip TileCoop): long reductions (3x3 windows, 2048-channel
// 1x1) -- +8..12 % there, a loss on the 512-deep ones (profiles/r04_gemm_lab_coop.txt)
using TileCoop = Cfg<256, 256, 16, 4, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_COOP | KNOB_NOTICKET | KNOB_RNE>;
struct SplitFam { using T256 = Tile256D; using T128 = Tile128D; using TS = TileS; };
struct Bf16Fam {
  using T256 = Tile256B;
  using T128 = Cfg<128, 128, 16, 2, 2, 2, MODE_DLDS, 4, KNOB_SPLIT | KNOB_BF16>;
  using TS = Cfg<64, 128, 16, 1, 2, 2, MODE_DLDS, 3, KNOB_SPLIT | KNOB_BF16>;
};
inline bool bf16_products(const ait_launch_ctx* ctx) { return ctx && (ctx->flags & AIT_CTX_BF16); }

inline int log2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) s++;
  return s;
}

// ConvGeom with the identity tap map / row map (plain convolutions); the parity-class data gradient overrides them
inline ConvGeom make_geom(int hw_shift, int w_shift, int src_h, int src_w, int kw, int a, int b, int c, int div_shift,
                          int seg, long long b_tap_stride, const float* zero, int a_group, int n_group) {
  ConvGeom m{};
  m.rows_hw_shift = hw_shift; m.rows_w_shift = w_shift; m.src_h = src_h; m.src_w = src_w; m.kw = kw;
  m.n_rows = 0x7fffffff;
  m.a = a; m.b = b; m.c = c; m.div_shift = div_shift; m.seg = seg; m.b_tap_stride = b_tap_stride;
  m.zero = zero; m.a_group = a_group; m.n_group = n_group;
  m.wt_kw = kw; m.a2_class = -1;
  return m;
}

// maps whose sides are not powers of two: the row -> (image, y, x) decomposition by corrected f32 quotients
// (gemm_f32_impl.h div_small; exact below 2^24 rows)
inline bool set_general_rows(ConvGeom& m, int rows_h, int rows_w, long long n_rows) {
  if (n_rows >= (1ll << 24) || rows_h <= 0 || rows_w <= 0) return false;
  m.rows_hw_shift = -1; m.rows_w_shift = -1;
  m.rows_hw = rows_h * rows_w; m.rows_w = rows_w;
  m.inv_hw = 1.0f / (float)m.rows_hw; m.inv_w = 1.0f / (float)rows_w;
  m.n_rows = (int)n_rows;
  return true;
}

inline int check_geom(const ait_conv_geom* q, int cin, int cout) {
  if (!q || q->n < 0 || q->in_h <= 0 || q->in_w <= 0 || q->out_h <= 0 || q->out_w <= 0 || q->kh <= 0 || q->kw <= 0 ||
      q->stride <= 0 || q->pad < 0 || cin <= 0 || cout <= 0 || q->groups < 0)
    return AIT_EINVAL;
  if (q->groups > 1) {
    // a 128-column (forward, data gradient) or 128-row (weight gradient) tile must lie inside one group
    if (cin % q->groups || cout % q->groups) return AIT_EINVAL;
    if ((cin / q->groups) % 128 || (cout / q->groups) % 128) return AIT_EUNSUPPORTED;
  }
  if (log2_exact(q->stride) < 0) return AIT_EUNSUPPORTED;
  // every output position must see the window the geometry describes
  if ((q->out_h - 1) * q->stride - q->pad + q->kh - 1 < 0 || (q->out_w - 1) * q->stride - q->pad + q->kw - 1 < 0)
    return AIT_EINVAL;
  return AIT_OK;
}

template <class T, int CONV, bool AK, bool BKC, bool GRP = false>
int conv_launch(const GemmArgs& g, hipStream_t s, const SchedWs& ws) {
  if (g.flags & AIT_GEMM_ATOMIC) return launch<T, AK, BKC, EPI_ATOMIC, NoProbe, CONV, GRP>(g, s, ws);
  if (GRP) {
    // grouped: bias / ReLU, and for the data-gradient layout "+ residual" / the ReLU-backward gate (the general
    // fallback of the parity-class launch)
    if constexpr (CONV == CONV_A && !BKC) {
      if (g.residual) return launch<T, AK, BKC, EPI_RES, NoProbe, CONV, GRP>(g, s, ws);
    }
    return launch<T, AK, BKC, EPI_STORE, NoProbe, CONV, GRP>(g, s, ws);
  }
  if (g.residual) return launch<T, AK, BKC, EPI_RES, NoProbe, CONV>(g, s, ws);
  return launch<T, AK, BKC, EPI_STORE, NoProbe, CONV>(g, s, ws);
}
// the parity-class data gradient of a stride-2 convolution (K-outer weights, row-mapped result; "+ residual" / the
// ReLU-backward gate with the same map)
template <class T, bool GRP>
int parity_launch(const GemmArgs& g, hipStream_t s, const SchedWs& ws) {
  if (g.residual) return launch<T, true, false, EPI_RES, NoProbe, CONV_A, GRP, true>(g, s, ws);
  return launch<T, true, false, EPI_STORE, NoProbe, CONV_A, GRP, true>(g, s, ws);
}
template <class F, int CONV, bool AK, bool BKC>
int conv_dispatch_f(const GemmArgs& g, hipStream_t s, const SchedWs& ws) {
  if (g.conv.a_group) {        // grouped: separate instantiations (see glds16<FORCE_UNIFORM>)
    if ((g.residual || (g.flags & ~AIT_GEMM_RELU)) && !(CONV == CONV_A && !BKC)) return AIT_EUNSUPPORTED;
    if (g.M <= 128) return conv_launch<typename F::TS, CONV, AK, BKC, true>(g, s, ws);
    return conv_launch<typename F::T256, CONV, AK, BKC, true>(g, s, ws);
  }
  if constexpr (std::is_same<F, SplitFam>::value && CONV == CONV_A) {      // (the gathered weight-gradient layout measured slower on it)
    const long long tiles_sq = (long long)((g.M + 255) / 256) * ((g.N + 255) / 256) * g.splits;
    if (g.K >= 2048 && g.M >= 256 && g.N >= 256 && (g.M % 4) == 0 && (g.N % 4) == 0 && tiles_sq >= 96 &&
        ((g.flags & AIT_GEMM_ATOMIC) || ws.p != nullptr))
      return conv_launch<TileCoop, CONV, AK, BKC>(g, s, ws);
  }
  const long long tiles256 = (long long)((g.M + 255) / 256) * ((g.N + 127) / 128) * g.splits;
  if (tiles256 >= 512 || (tiles256 >= 96 && g.K >= 512 && g.splits == 1 && !(g.flags & AIT_GEMM_ATOMIC) && ws.p != nullptr))
    return conv_launch<typename F::T256, CONV, AK, BKC>(g, s, ws);
  return conv_launch<typename F::T128, CONV, AK, BKC>(g, s, ws);     // few tiles: 128x128, three to a CU
}
template <int CONV, bool AK, bool BKC>
int conv_dispatch(const GemmArgs& g, hipStream_t s, const ait_launch_ctx* ctx) {
  if (bf16_products(ctx)) return conv_dispatch_f<Bf16Fam, CONV, AK, BKC>(g, s, sched_ws_of(ctx));
  return conv_dispatch_f<SplitFam, CONV, AK, BKC>(g, s, sched_ws_of(ctx));
}
template <class F>
int parity_dispatch(const GemmArgs& g, hipStream_t s, const SchedWs& ws, bool small, bool grouped, bool big) {
  if (small) return parity_launch<typename F::TS, true>(g, s, ws);
  if (grouped) return parity_launch<typename F::T256, true>(g, s, ws);
  if (big) return parity_launch<typename F::T256, false>(g, s, ws);
  return parity_launch<typename F::T128, false>(g, s, ws);
}
// does the weight gradient run on the one-per-CU 256 x 256 tile (every value split once per workgroup)?  Where its
// tiles x splits make about ONE round of the 256 slots: layer4's 3x3 (36 tiles x 8: 186 against 175 TFLOP/s on the
// 256 x 128 tile, lab); at 2.25 rounds (the RPN's 1024 -> 512 3x3) the 256 x 128 tile's finer units win (158 against 152).
inline bool coop_wgrad_takes(int M, int N, long long rows, int splits) {
  if (M < 256 || N < 256 || (M & 3) || (N & 3) || rows < 4096) return false;
  const long long units = (long long)((M + 255) / 256) * ((N + 255) / 256) * splits;
  return units >= 128 && units <= 320;
}
template <class F>
int wgrad_dispatch(const GemmArgs& g, hipStream_t s, const SchedWs& ws, bool grouped, bool coop) {
  // (grouped: 128-row tiles, one group of output channels per row tile)
  if (grouped) return conv_launch<typename F::T128, CONV_B, false, false, true>(g, s, ws);
  if constexpr (std::is_same<F, SplitFam>::value) {
    if (coop) return conv_launch<TileCoop, CONV_B, false, false>(g, s, ws);
  }
  return conv_launch<typename F::T256, CONV_B, false, false>(g, s, ws);
}
}  // namespace

// position-major rows (ConvGeom::pm_maps; csrc/tail.hip): power-of-two maps, stride 1, dense, fewer than 2^24 rows
// (launches over position-major rows run on the cooperative 256 x 256 tile's kernels -- the only ones that carry the
// position-major decode -- in the split product form: the callers below)
static int set_pm(ConvGeom& c, const ait_conv_geom* q, int pm, bool general, long long rows, const ait_launch_ctx* ctx) {
  if (!pm) return AIT_OK;
  if (general || q->stride != 1 || q->groups > 1 || q->in_h != q->out_h || q->in_w != q->out_w || rows >= (1ll << 24) ||
      bf16_products(ctx) || (ctx && (ctx->flags & AIT_CTX_NATIVE_F32)))
    return AIT_EUNSUPPORTED;
  c.pm_maps = q->n;
  c.inv_pm = 1.0f / (float)q->n;
  return AIT_OK;
}

// Tap skipping for the forward / data gradient of position-major 4 x 4 maps (ConvGeom::pm_skip): on, if the launch is what
// the kernel's two-group cut handles -- the 256 x 256 tile, every XCD's chunk of the tile list = the tiles of two whole
// positions and no longer than the workgroups of an XCD (32), a dense 3x3-like window with padding, the split product form,
// the partial-tile scratch in the launch context.  More maps than that (8 pairs of 300 proposals: 10 row tiles per position)
// go in several launches, each over a slice of every position's row tiles (ConvGeom::pm_t0 / pm_tcnt; pm_launches says how
// many).  Returns the executed (position, tap) pairs per map, 0 = off.
static int set_pm_skip(GemmArgs& g, const ait_conv_geom* q, int pm, int seg, const ait_launch_ctx* ctx, int& pm_launches) {
  pm_launches = 1;
  if (!pm || !ait_lab::Knobs::l4_pm_skip || bf16_products(ctx)) return 0;
  if (q->in_h != 4 || q->in_w != 4 || q->kh * q->kw < 2 || q->pad < 1 || q->groups > 1 || (seg % 16)) return 0;
  const SchedWs ws = sched_ws_of(ctx);
  const int tiles_n = (g.N + 255) / 256, tpr = (q->n + 255) / 256;
  if (g.N % 256 || g.M < 4096 || tiles_n > 16) return 0;
  const int per = 16 / tiles_n;                          // row tiles per position a launch can take: 2 positions x per x tiles_n <= 32
  if (!ws.p || ws.bytes < kCtlBytes + (size_t)256 * 256 * 256 * sizeof(float)) return 0;
  // (32 = the workgroups of an XCD on the 256-CU part, one cooperative tile per CU: a device with fewer resident workgroups keeps
  // every tap -- same results, no launch that could not be cut)
  if (stream_slots<TileCoop>(reinterpret_cast<const void*>(
          gemm_f32_stream_kernel<TileCoop, true, true, EPI_STORE, NoProbe, CONV_A, false, false>)) / AIT_NXCD < 32)
    return 0;
  pm_launches = (tpr + per - 1) / per;
  g.conv.pm_skip = 1;
  g.conv.pm_t0 = 0;
  g.conv.pm_tcnt = (tpr + pm_launches - 1) / pm_launches;
  g.conv.pm_kh = q->kh;
  // corners beside interiors, edges beside edges: 13 / 12 taps per XCD (position pairs in launch order)
  g.conv.pm_order = 0xedb87421af9c6350ull;
  int pairs = 0;
  for (int p = 0; p < 16; p++) {
    int l, ny, nx;
    WorkMap::pm_axis(p >> 2, 4, q->kh, g.conv.b, g.conv.c, l, ny);
    WorkMap::pm_axis(p & 3, 4, q->kw, g.conv.b, g.conv.c, l, nx);
    pairs += ny * nx;
  }
  return pairs;
}

// the forward / data gradient over position-major rows on the cooperative tile: one launch, or (tap skipping over more maps
// than one launch cuts) one per slice of every position's row tiles -- the slices are disjoint sets of output rows
template <bool BKC>
static int pm_launch_slices(GemmArgs& g, const ait_conv_geom* q, int n_launch, hipStream_t s, const SchedWs& ws) {
  if (!g.conv.pm_skip || n_launch <= 1) return conv_launch<TileCoop, CONV_A, true, BKC>(g, s, ws);
  const int tpr = (q->n + 255) / 256, per = g.conv.pm_tcnt;
  for (int t0 = 0; t0 < tpr; t0 += per) {
    g.conv.pm_t0 = t0;
    g.conv.pm_tcnt = t0 + per <= tpr ? per : tpr - t0;
    const int rc = conv_launch<TileCoop, CONV_A, true, BKC>(g, s, ws);
    if (rc != AIT_OK) return rc;
  }
  return AIT_OK;
}

int ait_conv_fwd_f32_pm(const float* x, int ldx, const float* w, const ait_conv_geom* q, int cin, int cout,
                        const float* bias, const float* residual, int flags, float* y, int ldy,
                        const float* zeros, size_t zeros_floats, int pm, const ait_launch_ctx* ctx, void* stream) {
  AIT_TRY_RC(check_geom(q, cin, cout));
  const int hw = log2_exact(q->out_h * q->out_w), ws = log2_exact(q->out_w);
  const bool general = hw < 0 || ws < 0;
  if ((cin & 15) || (cout & 3)) return AIT_EUNSUPPORTED;
  const long long rows = (long long)q->n * q->out_h * q->out_w;
  if (general && (rows >= (1ll << 24) || q->groups > 1)) return AIT_EUNSUPPORTED;
  if (rows == 0) return AIT_OK;
  if (rows > 0x7fffffffLL / 4 || !x || !w || !y || !zeros || zeros_floats < (size_t)cin + 144) return AIT_EINVAL;
  if (flags & ~(AIT_GEMM_RELU | AIT_GEMM_MASK_POS)) return AIT_EINVAL;
  const int taps = q->kh * q->kw, G = q->groups > 1 ? q->groups : 1, cing = cin / G;
  GemmArgs g;
  // (grouped: output channel n holds the taps * cin/G weights of its own group; the gathered rows start at the
  // group's first channel)
  AIT_TRY_RC(make_args(0, 1, (int)rows, cout, taps * cing, 1.f, x, ldx, w, taps * cing, y, ldy, bias, residual, flags, 1, 0, 0,
                       16, g));
  g.conv = make_geom(hw, ws, q->in_h, q->in_w, q->kw, q->stride, 1, -q->pad, 0, cing, 0, zeros, G > 1 ? cing : 0, cout / G);
  if (general) set_general_rows(g.conv, q->out_h, q->out_w, rows);
  AIT_TRY_RC(set_pm(g.conv, q, pm, general, rows, ctx));
  if (pm) {
    // (executed work: the taps that reach the map -- the skipped ones multiply rows of zeros)
    int n_launch = 1;
    const int pairs = set_pm_skip(g, q, pm, cing, ctx, n_launch);
    AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, pairs ? 2.0 * q->n * pairs * cout * cing : 2.0 * rows * cout * taps * cing,
                        ait_stream(stream), (int)rows, cout, pairs ? pairs * cing / 16 : taps * cing, 0, 1, 1);
    return pm_launch_slices<true>(g, q, pairs ? n_launch : 1, ait_stream(stream), sched_ws_of(ctx));
  }
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, 2.0 * rows * cout * taps * cing, ait_stream(stream), (int)rows, cout,
                      taps * cing, 0, 1, 1);
  return conv_dispatch<CONV_A, true, true>(g, ait_stream(stream), ctx);
}
AIT_API int ait_conv_fwd_f32(const float* x, int ldx, const float* w, const ait_conv_geom* q, int cin, int cout,
                             const float* bias, const float* residual, int flags, float* y, int ldy,
                             const float* zeros, size_t zeros_floats, const ait_launch_ctx* ctx, void* stream) {
  return ait_conv_fwd_f32_pm(x, ldx, w, q, cin, cout, bias, residual, flags, y, ldy, zeros, zeros_floats, 0, ctx, stream);
}

// Data gradient of a STRIDE-2 convolution by parity class of the input positions, ONE launch (ConvGeom, gemm_f32_impl.h):
// class rows (2ya + py, 2xa + px) are reached only by the window taps ty = (py + pad) mod 2 (+ 2 ...), so a 3x3 window
// has 1 / 2 / 2 / 4 taps per class instead of 9 mostly-zero ones (45 GFLOP executed instead of 181 on the SK block's
// 3x3 branch).  Optionally a SECOND convolution of the same input whose stride-2 window is 1x1 (it reaches class (0, 0)
// only) rides along as one more tap of that class: dx = conv3^T(dy) + conv1^T(dy1).  Returns AIT_EUNSUPPORTED where the
// decomposition does not apply (a class without taps, class rows not a multiple of the tile): the caller falls back
// to the gather over all taps.
int ait_conv_bwd_data_s2(const float* dy, int lddy, const float* w, const ait_conv_geom* q, const float* dy1, int lddy1,
                         const float* w1, int cin, int cout, const float* residual, int flags, float* dx, int lddx,
                         const float* zeros, const ait_launch_ctx* ctx, void* stream) {
  const int G = q->groups > 1 ? q->groups : 1, cing = cin / G, coutg = cout / G;
  if (q->stride != 2 || (q->in_h & 1) || (q->in_w & 1)) return AIT_EUNSUPPORTED;
  const int ch = q->in_h / 2, cw = q->in_w / 2;                 // the class grid
  if (ch != 4 || cw != 4) return AIT_EUNSUPPORTED;              // (the epilogue's constant row offsets: 8x8 maps, the SK block's)
  const int hw = log2_exact(ch * cw), wsft = log2_exact(cw);
  const int img_shift = log2_exact(q->in_h * q->in_w), y_shift = log2_exact(2 * q->in_w);
  if (hw < 0 || wsft < 0 || img_shift < 0 || y_shift < 0) return AIT_EUNSUPPORTED;
  const long long class_rows = (long long)q->n * ch * cw;
  if (class_rows * 4 > 0x7fffffffLL / 4) return AIT_EINVAL;
  const bool big = class_rows % 256 == 0 && (G > 1 || class_rows * 4 / 256 * ((cin + 127) / 128) >= 256);
  const bool small = G > 1 && !big && class_rows % 64 == 0;            // grouped, a few rows: 64-row tiles
  if (!big && !small && (G > 1 || class_rows % 128 != 0)) return AIT_EUNSUPPORTED;
  ConvGeom cg = make_geom(hw, wsft, q->out_h, q->out_w, 1, 1, -1, 0, 0, coutg, (long long)cing, zeros, G > 1 ? coutg : 0, cing);
  cg.bm_shift = big ? 8 : (small ? 6 : 7);
  cg.wt_kw = q->kw;
  cg.out_img_shift = img_shift; cg.out_y_shift = y_shift;
  cg.A2 = dy1; cg.B2 = w1; cg.lda2 = lddy1; cg.ldb2 = cing; cg.a2_class = -1;
  int kmax = 0;
  double flops = 0.0;
  // class order: heaviest first (odd, odd) ... lightest last (even, even): the tile list interleaves them anyway
  const int order[4][2] = {{1, 1}, {1, 0}, {0, 1}, {0, 0}};
  for (int c = 0; c < 4; c++) {
    const int py = order[c][0], px = order[c][1];
    const int ty0 = (py + q->pad) & 1, tx0 = (px + q->pad) & 1;
    const int nty = ty0 < q->kh ? (q->kh - ty0 + 1) / 2 : 0, ntx = tx0 < q->kw ? (q->kw - tx0 + 1) / 2 : 0;
    if (nty * ntx == 0) return AIT_EUNSUPPORTED;
    ConvGeom::ParityClass& pc = cg.cls[c];
    pc.cy = (py + q->pad - ty0) / 2; pc.cx = (px + q->pad - tx0) / 2;
    pc.kw = ntx; pc.ntaps = nty * ntx; pc.wt_y0 = ty0; pc.wt_x0 = tx0;
    pc.out_base = py * q->in_w + px;
    pc.k_end = pc.ntaps * coutg;
    if (dy1 && py == 0 && px == 0) { pc.k_end += coutg; cg.a2_class = c; }
    if (pc.k_end > kmax) kmax = pc.k_end;
    flops += 2.0 * class_rows * cin * pc.k_end;
  }
  if (dy1 && (!w1 || cg.a2_class < 0 || q->pad * 2 + 1 != q->kh)) return AIT_EINVAL;    // (the 1x1 sits at the window centre)
  GemmArgs g;
  AIT_TRY_RC(make_args(0, 0, (int)(class_rows * 4), cin, kmax, 1.f, dy, lddy, w, q->kh * q->kw * cing, dx, lddx, nullptr, residual,
                       flags, 1, 0, 0, 16, g));
  g.conv = cg;
  hipStream_t s = ait_stream(stream);
  SchedWs ws = sched_ws_of(ctx);
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, flops, s, (int)(class_rows * 4), cin, (int)(flops / (2.0 * class_rows * 4 * cin)),
                      0, 0, 1);
  if (bf16_products(ctx)) return parity_dispatch<Bf16Fam>(g, s, ws, small, G > 1, big);
  return parity_dispatch<SplitFam>(g, s, ws, small, G > 1, big);
}

int ait_conv_bwd_data_f32_pm(const float* dy, int lddy, const float* w, const ait_conv_geom* q, int cin, int cout,
                             const float* residual, int flags, float* dx, int lddx, const float* zeros,
                             size_t zeros_floats, int pm, const ait_launch_ctx* ctx, void* stream) {
  AIT_TRY_RC(check_geom(q, cin, cout));
  const int hw = log2_exact(q->in_h * q->in_w), ws = log2_exact(q->in_w);
  const bool general = hw < 0 || ws < 0;
  if ((cout & 15) || (cin & 3)) return AIT_EUNSUPPORTED;
  const long long rows = (long long)q->n * q->in_h * q->in_w;
  if (general && (rows >= (1ll << 24) || q->groups > 1)) return AIT_EUNSUPPORTED;
  if (rows == 0) return AIT_OK;
  if (rows > 0x7fffffffLL / 4 || !dy || !w || !dx || !zeros || zeros_floats < (size_t)cout + 144) return AIT_EINVAL;
  if (flags & ~AIT_GEMM_MASK_POS) return AIT_EINVAL;
  const int taps = q->kh * q->kw, G = q->groups > 1 ? q->groups : 1, cing = cin / G, coutg = cout / G;
  if (q->stride == 2) {        // by parity class of the input positions where that applies
    const int rc2 = ait_conv_bwd_data_s2(dy, lddy, w, q, nullptr, 0, nullptr, cin, cout, residual, flags, dx, lddx, zeros, ctx,
                                         stream);
    if (rc2 != AIT_EUNSUPPORTED) return rc2;
  }
  GemmArgs g;
  // B is addressed per tap (retap): K-outer rows (t, co) at w + t*cin/G + co*(taps*cin/G)
  AIT_TRY_RC(make_args(0, 0, (int)rows, cin, taps * coutg, 1.f, dy, lddy, w, taps * cing, dx, lddx, nullptr, residual, flags, 1,
                       0, 0, 16, g));
  g.conv = make_geom(hw, ws, q->out_h, q->out_w, q->kw, 1, -1, q->pad, log2_exact(q->stride), coutg, (long long)cing, zeros,
                     G > 1 ? coutg : 0, cing);
  if (general) set_general_rows(g.conv, q->in_h, q->in_w, rows);
  AIT_TRY_RC(set_pm(g.conv, q, pm, general, rows, ctx));
  if (pm) {
    int n_launch = 1;
    const int pairs = set_pm_skip(g, q, pm, coutg, ctx, n_launch);
    AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, pairs ? 2.0 * q->n * pairs * cin * coutg : 2.0 * rows * cin * taps * coutg,
                        ait_stream(stream), (int)rows, cin, pairs ? pairs * coutg / 16 : taps * coutg, 0, 0, 1);
    return pm_launch_slices<false>(g, q, pairs ? n_launch : 1, ait_stream(stream), sched_ws_of(ctx));
  }
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, 2.0 * rows * cin * taps * coutg, ait_stream(stream), (int)rows, cin,
                      taps * coutg, 0, 0, 1);
  return conv_dispatch<CONV_A, true, false>(g, ait_stream(stream), ctx);
}
AIT_API int ait_conv_bwd_data_f32(const float* dy, int lddy, const float* w, const ait_conv_geom* q, int cin, int cout,
                                  const float* residual, int flags, float* dx, int lddx, const float* zeros,
                                  size_t zeros_floats, const ait_launch_ctx* ctx, void* stream) {
  return ait_conv_bwd_data_f32_pm(dy, lddy, w, q, cin, cout, residual, flags, dx, lddx, zeros, zeros_floats, 0, ctx, stream);
}

int ait_conv_bwd_weight_f32_pm(const float* dy, int lddy, const float* x, int ldx, const ait_conv_geom* q, int cin,
                               int cout, float* dw, int split_k, const float* zeros, size_t zeros_floats, int pm,
                               const ait_launch_ctx* ctx, void* stream) {
  AIT_TRY_RC(check_geom(q, cin, cout));
  const int hw = log2_exact(q->out_h * q->out_w), ws = log2_exact(q->out_w);
  const long long rows = (long long)q->n * q->out_h * q->out_w;
  const bool general = hw < 0 || ws < 0 || (rows & 15);
  if (((cin / (q->groups > 1 ? q->groups : 1)) % 128) || (cout & 3)) return AIT_EUNSUPPORTED;
  if (general && (rows >= (1ll << 24) || q->groups > 1 || rows < 16)) return AIT_EUNSUPPORTED;
  if (rows == 0) return AIT_OK;
  if (rows > 0x7fffffffLL / 4 || !dy || !x || !dw || !zeros || zeros_floats < (size_t)cin + 144) return AIT_EINVAL;
  const int taps = q->kh * q->kw, G = q->groups > 1 ? q->groups : 1, cing = cin / G;
  GemmArgs g;
  // (general maps: the reduction runs over rows rounded up to a whole slab; see ConvGeom::n_rows)
  const bool sp_coop = G == 1 && !bf16_products(ctx) && coop_wgrad_takes(cout, taps * cing, rows, split_k < 1 ? 1 : split_k);
  AIT_TRY_RC(make_args(1, 0, cout, taps * cing, (int)((rows + 15) / 16 * 16), 1.f, dy, lddy, x, ldx, dw, taps * cing, nullptr,
                       nullptr, AIT_GEMM_ATOMIC, split_k < 1 ? 1 : split_k, 0, 0, 16, g));
  g.conv = make_geom(hw, ws, q->in_h, q->in_w, q->kw, q->stride, 1, -q->pad, 0, cing, 0, zeros, G > 1 ? cing : 0, cout / G);
  if (general) set_general_rows(g.conv, q->out_h, q->out_w, rows);
  AIT_TRY_RC(set_pm(g.conv, q, pm, general, rows, ctx));
  // position-major rows, a window that hangs over the map's edge: the reduction over position blocks, the (position, tap)
  // pairs that multiply nothing but zeros left out (ConvGeom::pm_wgrad).  Items are q->n rows long (rounded up to whole
  // slabs: the rows past a block read zeros); `splits` > 1 only says "several items per output tile" to the kernel.
  double flops = 2.0 * rows * cout * taps * cing;
  int probe_k = (int)rows;
  if (pm && ait_lab::Knobs::l4_pm_wgrad && !bf16_products(ctx) && q->pad > 0 && taps > 1 && (cing % 256) == 0 && cout >= 256 && (cout & 3) == 0 &&
      q->n >= 64) {
    g.conv.pm_wgrad = 1;
    g.conv.pm_kh = q->kh;
    g.splits = 2;
    g.k_per_split = (q->n + 15) / 16 * 16;
    const int pairs = WorkMap::pm_pairs(g.conv);
    flops = 2.0 * (double)q->n * pairs * cout * cing;
    probe_k = (int)((long long)q->n * pairs / taps);          // (the executed reduction length per output column, on average)
    AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, flops, ait_stream(stream), cout, taps * cing, probe_k, 1, 0, pairs);
    return conv_launch<TileCoop, CONV_B, false, false>(g, ait_stream(stream), sched_ws_of(ctx));
  }
  AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_GEMM, flops, ait_stream(stream), cout, taps * cing, probe_k, 1, 0, g.splits);
  if (pm) return conv_launch<TileCoop, CONV_B, false, false>(g, ait_stream(stream), sched_ws_of(ctx));      // (position-major rows: see set_pm)
  if (bf16_products(ctx)) return wgrad_dispatch<Bf16Fam>(g, ait_stream(stream), sched_ws_of(ctx), G > 1, false);
  return wgrad_dispatch<SplitFam>(g, ait_stream(stream), sched_ws_of(ctx), G > 1, sp_coop);
}
AIT_API int ait_conv_bwd_weight_f32(const float* dy, int lddy, const float* x, int ldx, const ait_conv_geom* q, int cin,
                                    int cout, float* dw, int split_k, const float* zeros, size_t zeros_floats,
                                    const ait_launch_ctx* ctx, void* stream) {
  return ait_conv_bwd_weight_f32_pm(dy, lddy, x, ldx, q, cin, cout, dw, split_k, zeros, zeros_floats, 0, ctx, stream);
}
