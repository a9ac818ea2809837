// ait_amd/csrc/roi_align_nhwc.hip -- RoIAlign on channels-last features, token-major output.
//
// Same operator as roi_align.hip (lib/model/csrc/cpu/ROIAlign_cpu.cpp:17-219, cuda/ROIAlign_cuda.cu
// :15-254: scale, PHxPW bins, adaptive ceil(roi/P) sampling grid, no coordinate rounding,
// out-of-range samples = 0, average), for the memory layout the neighbours of the operator use on
// MI355X: the feature map arrives as [B, H, W, C] (channels-last; the image-level co-attention
// emits token rows) and the pooled result is written as [n_rois, PH*PW, C] -- exactly the token
// rows the AIT's embedding GEMM reads (Models.py:252-256 transposes NCHW into that form; here
// nothing is transposed).
//
// In this layout the operator is a sparse-times-dense product with SCALAR weights per (bin, cell):
//     out[r, ph, pw, :] = sum_Y sum_X  Wy[r][ph][Y] * Wx[r][pw][X] * F[b_r, Y, X, :]
// because bilinear interpolation and the sample average are separable per axis.  Wy / Wx are the
// per-axis interpolation matrices (1/count folded into Wy); a small pre-pass writes them, with the
// band of non-zero entries of every row, into a caller-owned workspace.  Every feature access is a
// coalesced C-vector (16 B per lane), weights are wave-uniform.
//   forward : one workgroup per (roi, ph); each lane owns 4 channels and the PW accumulators of
//             its bin row; cells at bin borders are re-read from L1.
//   backward: feature cells GATHER from the RoIs that cover them, in RoI order -- every cell is written exactly
//             once: no atomics, no zero-fill, bitwise reproducible (the reference's atomicAdd scatter,
//             ROIAlign_cuda.cu:222-249, is not).  One wave per (4 x 4 cells, channel eighth) with the 16 cells'
//             accumulators in registers (roi_align_nhwc_bwd_tile_kernel; round 3: 0.47 -> 0.22 ms on the bench
//             shapes); one workgroup per cell for channel counts the tiled form does not take.
// Summation order differs from the reference's per-sample order, so results agree to fp32
// rounding (tests: 1e-5 relative), not bit for bit; the bit-exact NCHW kernels stay the default of
// the stand-alone operator.
#include "common.h"

namespace {

constexpr int kPW = 7;          // pooled width/height this file is specialised for (cfg.POOLING_SIZE)
constexpr int kCellsInFlight = ait_lab::Knobs::roi_cells;      // cells of a feature row a lane keeps in flight in the separable forward
// The separable two-stage forward (round 5) loads every window cell once and wins where the feature does NOT sit in L2
// (scripts/bench_roi.py's RoIs with a random image each: 0.253 against 0.29 ms); in the detector's step the RoIs come image
// by image, the sliced kernel's repeated loads are L1 / L2 hits, and its lack of barriers and LDS round trips wins: 0.199
// against 0.240 ms (same-box A/B, profiles/r05_roi_align_sep.txt).  The sliced kernel ships; the lab knob roi_fwd_separable
// (lab_knobs.h, scripts/build_variant.py) builds the other one.
constexpr bool kFwdSeparable = ait_lab::Knobs::roi_fwd_separable;
constexpr int kThreads = 256;

struct Geom {
  int b, gh, gw;
  float y0, x0, bh, bw, inv_count;
};

__device__ __forceinline__ Geom roi_geom(const float* __restrict__ r, float scale, int PH, int PW, int sr) {
  Geom g;
  g.b = (int)r[0];
  const float sw = r[1] * scale, sh = r[2] * scale, ew = r[3] * scale, eh = r[4] * scale;
  float rw = ew - sw, rh = eh - sh;
  if (rw < 1.f) rw = 1.f;
  if (rh < 1.f) rh = 1.f;
  g.y0 = sh;
  g.x0 = sw;
  g.bh = rh / (float)PH;
  g.bw = rw / (float)PW;
  g.gh = sr > 0 ? sr : (int)ceilf(rh / (float)PH);
  g.gw = sr > 0 ? sr : (int)ceilf(rw / (float)PW);
  g.inv_count = 1.f / (float)(g.gh * g.gw > 0 ? g.gh * g.gw : 1);
  return g;
}

// One row of a per-axis interpolation matrix: w[0..L) = sum over the `grid` samples of bin p of the
// two-tap linear weights (ROIAlign_cpu.cpp:36-95 along one axis).  Returns the non-zero band.
__device__ __forceinline__ void axis_row(float start, int p, float bin, int grid, int L, float mul,
                                         float* __restrict__ w, int& lo_out, int& hi_out) {
  int blo = L, bhi = -1;
  for (int i = 0; i < grid; i++) {
    float v = start + (float)p * bin + ((float)i + .5f) * bin / (float)grid;
    if (v < -1.0f || v > (float)L) continue;
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= L - 1) {
      hi = lo = L - 1;
      v = (float)lo;
    } else {
      hi = lo + 1;
    }
    const float wl = v - (float)lo;
    w[lo] += 1.f - wl;
    w[hi] += wl;
    blo = min(blo, lo);
    bhi = max(bhi, hi);
  }
  for (int i = blo; i <= bhi; i++) w[i] *= mul;
  lo_out = blo;
  hi_out = bhi;
}

// workspace layout: floats [n_rois][PH*H + PW*W]  then ints [n_rois][2*PH + 2*PW + 4]
//   ints: band_y[ph] = (lo, hi), band_x[pw] = (lo, hi), then ymin, ymax, xmin, xmax  (hi < lo: empty)
__host__ __device__ inline size_t tab_floats(int H, int W, int PH, int PW) { return (size_t)PH * H + (size_t)PW * W; }
__host__ __device__ inline size_t tab_ints(int PH, int PW) { return 2 * (size_t)PH + 2 * (size_t)PW + 4; }

__global__ __launch_bounds__(64) void roi_tables_kernel(const float* __restrict__ rois, int n_rois, int B,
                                                        int H, int W, int PH, int PW, float scale, int sr,
                                                        float* __restrict__ wf, int* __restrict__ wi) {
  const int n = blockIdx.x;
  const Geom g = roi_geom(rois + 5 * n, scale, PH, PW, sr);
  float* __restrict__ wy = wf + (size_t)n * tab_floats(H, W, PH, PW);
  float* __restrict__ wx = wy + (size_t)PH * H;
  int* __restrict__ ti = wi + (size_t)n * tab_ints(PH, PW);
  for (int i = threadIdx.x; i < PH * H + PW * W; i += 64) wy[i] = 0.f;
  __syncthreads();
  const bool ok = g.b >= 0 && g.b < B;
  const int t = threadIdx.x;
  if (t < PH) {
    int lo = H, hi = -1;
    if (ok) axis_row(g.y0, t, g.bh, g.gh, H, g.inv_count, wy + (size_t)t * H, lo, hi);
    ti[2 * t] = lo;
    ti[2 * t + 1] = hi;
  } else if (t >= 32 && t < 32 + PW) {
    const int p = t - 32;
    int lo = W, hi = -1;
    if (ok) axis_row(g.x0, p, g.bw, g.gw, W, 1.f, wx + (size_t)p * W, lo, hi);
    ti[2 * PH + 2 * p] = lo;
    ti[2 * PH + 2 * p + 1] = hi;
  }
  __syncthreads();
  if (t == 0) {
    int ymin = H, ymax = -1, xmin = W, xmax = -1;
    for (int p = 0; p < PH; p++) {
      if (ti[2 * p + 1] >= ti[2 * p]) {
        ymin = min(ymin, ti[2 * p]);
        ymax = max(ymax, ti[2 * p + 1]);
      }
    }
    for (int p = 0; p < PW; p++) {
      if (ti[2 * PH + 2 * p + 1] >= ti[2 * PH + 2 * p]) {
        xmin = min(xmin, ti[2 * PH + 2 * p]);
        xmax = max(xmax, ti[2 * PH + 2 * p + 1]);
      }
    }
    if (xmax < xmin || ymax < ymin) {   // no valid sample on one axis: the RoI touches nothing
      ymin = H; ymax = -1; xmin = W; xmax = -1;
    }
    int* lim = ti + 2 * PH + 2 * PW;
    lim[0] = ymin; lim[1] = ymax; lim[2] = xmin; lim[3] = xmax;
  }
}

__device__ __forceinline__ float4 fma4(float w, float4 f, float4 a) {
  a.x = fmaf(w, f.x, a.x);
  a.y = fmaf(w, f.y, a.y);
  a.z = fmaf(w, f.z, a.z);
  a.w = fmaf(w, f.w, a.w);
  return a;
}

// XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so give every XCD a contiguous
// chunk of the work list (the PH rows of one RoI / neighbouring cells share their inputs in one L2).
__device__ __forceinline__ long long xcd_chunk_id(long long total) {
  const long long chunk = (total + AIT_NXCD - 1) / AIT_NXCD;
  const long long id = (long long)(blockIdx.x % AIT_NXCD) * chunk + blockIdx.x / AIT_NXCD;
  return (blockIdx.x / AIT_NXCD) < chunk && id < total ? id : -1;
}

__global__ __launch_bounds__(kThreads) void roi_align_nhwc_fwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, int n_rois, int B, int C, int H, int W,
    int PH, const float* __restrict__ wf, const int* __restrict__ wi, float* __restrict__ out) {
  constexpr int PW = kPW;
  const long long id = xcd_chunk_id((long long)n_rois * PH);
  if (id < 0) return;
  const int n = (int)(id / PH), ph = (int)(id % PH);
  const int C4 = C >> 2;
  float4* __restrict__ o = reinterpret_cast<float4*>(out) + ((size_t)n * PH + ph) * PW * C4;
  const int* __restrict__ ti = wi + (size_t)n * tab_ints(PH, PW);
  const int* lim = ti + 2 * PH + 2 * PW;
  const int ylo = ti[2 * ph], yhi = ti[2 * ph + 1];
  const int xmin = lim[2], xmax = lim[3];
  const int b = (int)rois[5 * n];
  if (yhi < ylo || xmax < xmin || b < 0 || b >= B) {
    for (int i = threadIdx.x; i < PW * C4; i += kThreads) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float sm[];   // wy[H] | wx[PW][W]
  float* s_wy = sm;
  float* s_wx = sm + H;
  const float* __restrict__ gwy = wf + (size_t)n * tab_floats(H, W, PH, PW) + (size_t)ph * H;
  const float* __restrict__ gwx = wf + (size_t)n * tab_floats(H, W, PH, PW) + (size_t)PH * H;
  for (int i = threadIdx.x; i < H; i += kThreads) s_wy[i] = gwy[i];
  for (int i = threadIdx.x; i < PW * W; i += kThreads) s_wx[i] = gwx[i];
  int bx_lo[PW], bx_hi[PW];
#pragma unroll
  for (int p = 0; p < PW; p++) {
    bx_lo[p] = ti[2 * PH + 2 * p];
    bx_hi[p] = ti[2 * PH + 2 * p + 1];
  }
  __syncthreads();
  const float4* __restrict__ f0 = reinterpret_cast<const float4*>(feat) + (size_t)b * H * W * C4;
  for (int c4 = threadIdx.x; c4 < C4; c4 += kThreads) {
    float4 acc[PW];
#pragma unroll
    for (int p = 0; p < PW; p++) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int y = ylo; y <= yhi; y++) {
      const float wyv = s_wy[y];
      if (wyv == 0.f) continue;
      const float4* __restrict__ frow = f0 + (size_t)y * W * C4 + c4;
#pragma unroll
      for (int p = 0; p < PW; p++) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int x = bx_lo[p]; x <= bx_hi[p]; x++) t = fma4(s_wx[p * W + x], frow[(size_t)x * C4], t);
        acc[p] = fma4(wyv, t, acc[p]);
      }
    }
#pragma unroll
    for (int p = 0; p < PW; p++) o[(size_t)p * C4 + c4] = acc[p];
  }
}

// Forward, channel-sliced (C % 1024 == 0... any C that is a multiple of 8 * 128 floats): workgroup = (RoI,
// 128-channel slice); blocks b and b + 8 share an XCD, so slice = b % 8 pins ONE eighth of the channels to
// every XCD: its L2 sees the feature slice of one image (H*W*512 B = 1.2 MB at 38x63) while the RoIs of that
// image stream by in order -- the feature is fetched from HBM once instead of once per overlapping RoI
// (round 1: 10-21x re-fetch with whole-C workgroups spread over all XCDs).  Eight groups of 32 lanes: group
// ph < PH owns bin row ph, a lane 4 channels and the PW accumulators of its row.
__global__ __launch_bounds__(kThreads) void roi_align_nhwc_fwd_sliced_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, int n_rois, int B, int C, int H, int W,
    int PH, const float* __restrict__ wf, const int* __restrict__ wi, float* __restrict__ out) {
  constexpr int PW = kPW;
  const int slice = blockIdx.x % AIT_NXCD, n = blockIdx.x / AIT_NXCD;
  if (n >= n_rois) return;
  const int C4 = C >> 2, S4 = C4 / AIT_NXCD;             // float4 per slice (32 for C = 1024)
  const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int* __restrict__ ti = wi + (size_t)n * tab_ints(PH, PW);
  const int* lim = ti + 2 * PH + 2 * PW;
  const int xmin = lim[2], xmax = lim[3];
  const int b = (int)rois[5 * n];
  const bool dead = xmax < xmin || b < 0 || b >= B;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // wy[PH][H] | wx[PW][W]
  float* s_wy = sm;
  float* s_wx = sm + (size_t)PH * H;
  const float* __restrict__ gw = wf + (size_t)n * tab_floats(H, W, PH, PW);
  if (!dead) {
    for (int i = threadIdx.x; i < PH * H + PW * W; i += kThreads) sm[i] = gw[i];
  }
  int bx_lo[PW], bx_hi[PW];
#pragma unroll
  for (int p = 0; p < PW; p++) {
    bx_lo[p] = ti[2 * PH + 2 * p];
    bx_hi[p] = ti[2 * PH + 2 * p + 1];
  }
  __syncthreads();
  if (grp >= PH) return;
  const int ph = grp;
  const int ylo = ti[2 * ph], yhi = ti[2 * ph + 1];
  float4* __restrict__ o = reinterpret_cast<float4*>(out) + ((size_t)n * PH + ph) * PW * C4 + slice * S4;
  const float4* __restrict__ f0 = reinterpret_cast<const float4*>(feat) + (size_t)(dead ? 0 : b) * H * W * C4 + slice * S4;
  for (int c4 = l; c4 < S4; c4 += 32) {
    float4 acc[PW];
#pragma unroll
    for (int p = 0; p < PW; p++) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!dead) {
      for (int y = ylo; y <= yhi; y++) {
        const float wyv = s_wy[ph * H + y];
        if (wyv == 0.f) continue;
        const float4* __restrict__ frow = f0 + (size_t)y * W * C4 + c4;
#pragma unroll
        for (int p = 0; p < PW; p++) {
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int x = bx_lo[p]; x <= bx_hi[p]; x++) t = fma4(s_wx[p * W + x], frow[(size_t)x * C4], t);
          acc[p] = fma4(wyv, t, acc[p]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < PW; p++) o[(size_t)p * C4 + c4] = acc[p];
  }
}

// Forward, channel-sliced AND separable in two stages (round 5; LAB: built with -DAIT_ROI_FWD_SEPARABLE, see kFwdSeparable; profiles/r05_roi_align_sep.txt).  The sliced kernel
// above walks, per bin row ph, its band of feature rows, and per row every bin's band of cells: a cell under two bins is
// loaded twice, a row under two bin rows twice more (2-4x redundant loads, each behind its own address computation --
// the PMC pass of round 4 found the waves 0.83 of their cycles in s_waitcnt).  Here every cell of the RoI's window is
// loaded ONCE: workgroup = (RoI, 128-channel slice) as before; the window's rows are taken eight at a time, group g
// reduces row y0 + g along x into the PW partial sums T[g][pw] = sum_x Wx[pw][x] F[y0 + g][x] with four cells in flight
// per lane (weights wave-uniform: a zero weight is skipped on the scalar side), the partials go through LDS (28 KB),
// and group ph < PH folds them into its bin row with Wy[ph][y].  Same factorisation and the same order of additions as
// the sliced kernel: results are bit-identical to it.
__global__ __launch_bounds__(kThreads) void roi_align_nhwc_fwd_sep_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, int n_rois, int B, int C, int H, int W,
    int PH, const float* __restrict__ wf, const int* __restrict__ wi, float* __restrict__ out) {
  constexpr int PW = kPW;
  constexpr int kRows = kThreads / 32;                  // feature rows per chunk: one per group
  const int slice = blockIdx.x % AIT_NXCD, n = blockIdx.x / AIT_NXCD;
  if (n >= n_rois) return;
  const int C4 = C >> 2, S4 = C4 / AIT_NXCD;
  const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;        // (two groups per wave)
  const int* __restrict__ ti = wi + (size_t)n * tab_ints(PH, PW);
  const int* lim = ti + 2 * PH + 2 * PW;
  const int ymin = lim[0], ymax = lim[1], xmin = lim[2], xmax = lim[3];
  const int b = (int)rois[5 * n];
  const bool dead = xmax < xmin || ymax < ymin || b < 0 || b >= B;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // T[kRows][PW][32] float4 | wy[PH][H] | wx[PW][W]
  float4* s_t = reinterpret_cast<float4*>(sm);
  float* s_wy = sm + kRows * PW * 32 * 4;
  float* s_wx = s_wy + (size_t)PH * H;
  const float* __restrict__ gw = wf + (size_t)n * tab_floats(H, W, PH, PW);
  if (!dead) {
    for (int i = threadIdx.x; i < PH * H + PW * W; i += kThreads) s_wy[i] = gw[i];
  }
  const int ph = grp;
  const int ylo = ph < PH ? ti[2 * ph] : 1, yhi = ph < PH ? ti[2 * ph + 1] : 0;
  __syncthreads();
  float4* __restrict__ o = reinterpret_cast<float4*>(out) + ((size_t)n * PH + (ph < PH ? ph : 0)) * PW * C4 + slice * S4;
  const float4* __restrict__ f0 = reinterpret_cast<const float4*>(feat) + (size_t)(dead ? 0 : b) * H * W * C4 + slice * S4;
  for (int c4 = l; c4 < S4; c4 += 32) {
    float4 acc[PW];
#pragma unroll
    for (int p = 0; p < PW; p++) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!dead) {
      for (int y0 = ymin; y0 <= ymax; y0 += kRows) {
        // ---- stage 1: row y0 + grp reduced along x into PW partial sums ----
        const int y = y0 + grp;
        if (y <= ymax) {
          float4 t[PW];
#pragma unroll
          for (int p = 0; p < PW; p++) t[p] = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4* __restrict__ frow = f0 + (size_t)y * W * C4 + c4;
          for (int x = xmin; x <= xmax; x += kCellsInFlight) {
            float4 v[kCellsInFlight];
#pragma unroll
            for (int i = 0; i < kCellsInFlight; i++) v[i] = frow[(size_t)min(x + i, xmax) * C4];
#pragma unroll
            for (int i = 0; i < kCellsInFlight; i++) {
              if (x + i <= xmax) {
#pragma unroll
                for (int p = 0; p < PW; p++) {
                  const float w = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_wx[p * W + x + i])));
                  if (w != 0.f) t[p] = fma4(w, v[i], t[p]);
                }
              }
            }
          }
#pragma unroll
          for (int p = 0; p < PW; p++) s_t[(grp * PW + p) * 32 + l] = t[p];
        }
        __syncthreads();
        // ---- stage 2: bin row ph folds the rows of this chunk that lie in its band ----
        if (ph < PH) {
          const int ya = max(ylo, y0), yb = min(yhi, min(ymax, y0 + kRows - 1));
          for (int yy = ya; yy <= yb; yy++) {
            const float wyv = s_wy[ph * H + yy];
            if (wyv == 0.f) continue;
#pragma unroll
            for (int p = 0; p < PW; p++) acc[p] = fma4(wyv, s_t[((yy - y0) * PW + p) * 32 + l], acc[p]);
          }
        }
        __syncthreads();
      }
    }
    if (ph < PH) {
#pragma unroll
      for (int p = 0; p < PW; p++) o[(size_t)p * C4 + c4] = acc[p];
    }
  }
}

// Backward: workgroup = one feature cell.  The RoIs are scanned in chunks of 256 (one per lane);
// the covering ones are compacted IN ROI ORDER into LDS together with their 2*PW weights for this
// cell, then every lane accumulates its 4 channels over that list.
__global__ __launch_bounds__(kThreads) void roi_align_nhwc_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ rois, int n_rois, int B, int C, int H, int W,
    int PH, const float* __restrict__ wf, const int* __restrict__ wi, float* __restrict__ gin) {
  constexpr int PW = kPW;
  const long long id = xcd_chunk_id((long long)B * H * W);
  if (id < 0) return;
  const int b = (int)(id / ((long long)H * W));
  const int yx = (int)(id % ((long long)H * W));
  const int y = yx / W, x = yx % W;
  const int C4 = C >> 2;
  __shared__ int s_roi[kThreads];
  __shared__ float s_w[kThreads][2 * PW];
  __shared__ int s_cnt[kThreads / 64 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NACC = 4;                       // channel groups per lane: C <= 4 * 4 * 256
  float4 acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; k++) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t tf = tab_floats(H, W, PH, PW);
  const size_t tint = tab_ints(PH, PW);
  for (int base = 0; base < n_rois; base += kThreads) {
    const int r = base + threadIdx.x;
    bool cov = false;
    if (r < n_rois && (int)rois[5 * r] == b) {
      const int* lim = wi + (size_t)r * tint + 2 * PH + 2 * PW;
      cov = y >= lim[0] && y <= lim[1] && x >= lim[2] && x <= lim[3];
    }
    const unsigned long long m = __ballot(cov);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; w++) {
      if (w < wave) off += s_cnt[w];
      total += s_cnt[w];
    }
    if (cov) {
      const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
      s_roi[slot] = r;
      const float* __restrict__ t = wf + (size_t)r * tf;
#pragma unroll
      for (int p = 0; p < PW; p++) {
        s_w[slot][p] = p < PH ? t[(size_t)p * H + y] : 0.f;
        s_w[slot][PW + p] = t[(size_t)PH * H + (size_t)p * W + x];
      }
    }
    __syncthreads();
    for (int i = 0; i < total; i++) {
      const int rr = s_roi[i];
      const float4* __restrict__ g = reinterpret_cast<const float4*>(gout) + (size_t)rr * PH * PW * C4;
#pragma unroll
      for (int p = 0; p < PW; p++) {
        const float wyv = s_w[i][p];
        if (p >= PH || wyv == 0.f) continue;
#pragma unroll
        for (int q = 0; q < PW; q++) {
          const float wv = wyv * s_w[i][PW + q];
          if (wv == 0.f) continue;
          const float4* __restrict__ gp = g + ((size_t)p * PW + q) * C4;
#pragma unroll
          for (int k = 0; k < NACC; k++) {
            const int c4 = threadIdx.x + k * kThreads;
            if (c4 < C4) acc[k] = fma4(wv, gp[c4], acc[k]);
          }
        }
      }
    }
    __syncthreads();
  }
  float4* __restrict__ o = reinterpret_cast<float4*>(gin) + ((size_t)b * H * W + yx) * C4;
#pragma unroll
  for (int k = 0; k < NACC; k++) {
    const int c4 = threadIdx.x + k * kThreads;
    if (c4 < C4) o[c4] = acc[k];
  }
}

// Backward, tiled (C a multiple of 32, at most 1024): one WAVE per (4 x 4 feature cells, channel eighth).
//   * block % 8 = channel slice: blocks b and b + 8 share an XCD, so every XCD's L2 sees ONE eighth of the channels of
//     the pooled gradient -- each of its bytes is fetched from HBM once (the per-cell kernel above reads a pooled
//     element once per feature cell its bin touches, through whichever of the 8 L2s the cell's workgroup runs on);
//   * a pooled element (r, p, q, channels) is loaded once per TILE its bin touches and applied to the tile's 16 cells
//     from registers: lane = (half, 4 channels), a half owns two cell rows = 8 cells x float4 accumulators;
//   * RoIs are scanned 64 at a time (one per lane), the covering ones compacted IN ROI ORDER with the range of bin
//     rows / columns that touch the tile, so every cell still sums its contributions in RoI order: written once,
//     no atomics, bitwise reproducible.
__global__ __launch_bounds__(64) void roi_align_nhwc_bwd_tile_kernel(
    const float* __restrict__ gout, const float* __restrict__ rois, int n_rois, int B, int C, int H, int W,
    int PH, const float* __restrict__ wf, const int* __restrict__ wi, float* __restrict__ gin) {
  constexpr int PW = kPW, T = 4;
  const int slice = blockIdx.x % AIT_NXCD;
  const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
  int tile = blockIdx.x / AIT_NXCD;
  const int b = tile / (tiles_x * tiles_y);
  if (b >= B) return;
  tile -= b * tiles_x * tiles_y;
  const int y0 = (tile / tiles_x) * T, x0 = (tile % tiles_x) * T;
  const int lane = threadIdx.x, half = lane >> 5, c4 = lane & 31;
  const int C4 = C >> 2, S4 = C4 / AIT_NXCD;            // float4 per pooled element / per slice
  const bool live = c4 < S4;
  const int ya = y0 + 2 * half, yb = ya + 1;            // this half's two cell rows
  __shared__ int s_roi[64];
  __shared__ int s_rng[64];
  __shared__ __attribute__((aligned(16))) float s_w[2][64];
  float4 acc[2][T];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int k = 0; k < T; k++) acc[j][k] = make_float4(0.f, 0.f, 0.f, 0.f);
  static_assert(8 * kPW <= 64, "one weight per lane");
  const size_t tf = tab_floats(H, W, PH, PW);
  const size_t tint = tab_ints(PH, PW);
  for (int base = 0; base < n_rois; base += 64) {
    const int r = base + lane;
    bool cov = false;
    int rng = 0;
    if (r < n_rois && (int)rois[5 * r] == b) {
      const int* __restrict__ ti = wi + (size_t)r * tint;
      const int* lim = ti + 2 * PH + 2 * PW;
      cov = y0 <= lim[1] && y0 + T - 1 >= lim[0] && x0 <= lim[3] && x0 + T - 1 >= lim[2];
      if (cov) {
        int plo = PH, phi = -1, qlo = PW, qhi = -1;
        for (int p = 0; p < PH; p++)
          if (ti[2 * p] <= y0 + T - 1 && ti[2 * p + 1] >= y0) { plo = min(plo, p); phi = p; }
        for (int q = 0; q < PW; q++)
          if (ti[2 * PH + 2 * q] <= x0 + T - 1 && ti[2 * PH + 2 * q + 1] >= x0) { qlo = min(qlo, q); qhi = q; }
        cov = phi >= plo && qhi >= qlo;
        rng = plo | (phi << 8) | (qlo << 16) | (qhi << 24);
      }
    }
    const unsigned long long m = __ballot(cov);
    if (cov) {
      const int slot = __popcll(m & ((1ull << lane) - 1ull));
      s_roi[slot] = r;
      s_rng[slot] = rng;
    }
    __syncthreads();
    const int total = __popcll(m);
    // The weights of a covering RoI for this tile -- wy[p][4 rows], wx[q][4 columns]: 56 values -- are fetched by 56
    // lanes (one value each, one RoI ahead), parked in LDS, and read back as wave-uniform operands: the loop below
    // issues ONE vector-memory instruction per pooled element (its float4 of channels) and nothing else.
    auto weight_of = [&](int rr_) __attribute__((always_inline)) -> float {
      const float* __restrict__ t = wf + (size_t)rr_ * tf;
      if (lane < 4 * PW) {
        const int p = lane >> 2, y = y0 + (lane & 3);
        return (p < PH && y < H) ? t[(size_t)p * H + y] : 0.f;
      }
      if (lane < 8 * PW) {
        const int q = (lane - 4 * PW) >> 2, x = x0 + ((lane - 4 * PW) & 3);
        return x < W ? t[(size_t)PH * H + (size_t)q * W + x] : 0.f;
      }
      return 0.f;
    };
    float wnext = total > 0 ? weight_of(s_roi[0]) : 0.f;
    for (int i = 0; i < total; i++) {
      const int rr = s_roi[i], rg = s_rng[i];
      float* sw = s_w[i & 1];
      sw[lane] = wnext;
      if (i + 1 < total) wnext = weight_of(s_roi[i + 1]);
      __syncthreads();              // (one wave: orders the LDS write before the reads; the other buffer is free again)
      const int plo = rg & 255, phi = (rg >> 8) & 255, qlo = (rg >> 16) & 255, qhi = (rg >> 24) & 255;
      const float4* __restrict__ g = reinterpret_cast<const float4*>(gout) + (size_t)rr * PH * PW * C4 + slice * S4 + c4;
      // bins (p, q) of this RoI that touch the tile, four at a time: their pooled-gradient loads go out together (a
      // wave that consumes each load before issuing the next one waits a full memory round trip per bin)
      constexpr int U = 4;
      int p = plo, q = qlo;
      const int n = (phi - plo + 1) * (qhi - qlo + 1);
      for (int k = 0; k < n; k += U) {
        float4 gv[U];
        int pu[U], qu[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          pu[u] = p;
          qu[u] = q;
          gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k + u < n && live) gv[u] = g[(size_t)(p * PW + q) * C4];
          if (++q > qhi) { q = qlo; p++; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (k + u >= n) break;
          const float wya = sw[4 * pu[u] + 2 * half], wyb = sw[4 * pu[u] + 2 * half + 1];
          const float4 wx = *reinterpret_cast<const float4*>(sw + 4 * PW + 4 * qu[u]);
          const float4 ga = make_float4(wya * gv[u].x, wya * gv[u].y, wya * gv[u].z, wya * gv[u].w);
          const float4 gb = make_float4(wyb * gv[u].x, wyb * gv[u].y, wyb * gv[u].z, wyb * gv[u].w);
          acc[0][0] = fma4(wx.x, ga, acc[0][0]); acc[1][0] = fma4(wx.x, gb, acc[1][0]);
          acc[0][1] = fma4(wx.y, ga, acc[0][1]); acc[1][1] = fma4(wx.y, gb, acc[1][1]);
          acc[0][2] = fma4(wx.z, ga, acc[0][2]); acc[1][2] = fma4(wx.z, gb, acc[1][2]);
          acc[0][3] = fma4(wx.w, ga, acc[0][3]); acc[1][3] = fma4(wx.w, gb, acc[1][3]);
        }
      }
    }
    __syncthreads();
  }
  if (!live) return;
  float4* __restrict__ o = reinterpret_cast<float4*>(gin) + (size_t)b * H * W * C4 + slice * S4 + c4;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int y = ya + j;
    if (y >= H) continue;
#pragma unroll
    for (int k = 0; k < T; k++)
      if (x0 + k < W) o[((size_t)y * W + x0 + k) * C4] = acc[j][k];
  }
}

inline bool bad(int n_rois, int B, int C, int H, int W, int PH, int PW) {
  return n_rois < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0;
}

inline bool unsupported(int C, int PH, int PW) {
  return PW != kPW || PH > kPW || (C & 3) || C > 4 * 4 * kThreads;
}

int make_tables(const float* rois, int n_rois, int B, int H, int W, int PH, int PW, float scale, int sr,
                void* ws, size_t ws_bytes, float*& wf, int*& wi, hipStream_t s) {
  const size_t need = (size_t)n_rois * (tab_floats(H, W, PH, PW) * sizeof(float) + tab_ints(PH, PW) * sizeof(int));
  if (!ws || ws_bytes < need) return AIT_EWORKSPACE;
  wf = static_cast<float*>(ws);
  wi = reinterpret_cast<int*>(wf + (size_t)n_rois * tab_floats(H, W, PH, PW));
  hipLaunchKernelGGL(roi_tables_kernel, dim3(n_rois), dim3(64), 0, s, rois, n_rois, B, H, W, PH, PW, scale, sr,
                     wf, wi);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

inline unsigned chunked_grid(long long total) {
  return (unsigned)((total + AIT_NXCD - 1) / AIT_NXCD * AIT_NXCD);
}

}  // namespace

AIT_API size_t ait_roi_align_nhwc_workspace_bytes(int n_rois, int H, int W, int PH, int PW) {
  if (n_rois <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0) return 0;
  return (size_t)n_rois * (tab_floats(H, W, PH, PW) * sizeof(float) + tab_ints(PH, PW) * sizeof(int));
}

AIT_API int ait_roi_align_nhwc_fwd(const float* feat, const float* rois, int n_rois, int B, int C, int H,
                                   int W, int PH, int PW, float spatial_scale, int sampling_ratio,
                                   void* workspace, size_t workspace_bytes, float* out, const ait_launch_ctx* ctx,
                                   void* stream) {
  if (bad(n_rois, B, C, H, W, PH, PW)) return AIT_EINVAL;
  if (unsupported(C, PH, PW)) return AIT_EUNSUPPORTED;
  if (n_rois == 0) return AIT_OK;
  if (!feat || !rois || !out) return AIT_EINVAL;
  hipStream_t s = ait_stream(stream);
  float* wf;
  int* wi;
  const int rc = make_tables(rois, n_rois, B, H, W, PH, PW, spatial_scale, sampling_ratio, workspace,
                             workspace_bytes, wf, wi, s);
  if (rc != AIT_OK) return rc;
  const size_t lds = sizeof(float) * ((size_t)H + (size_t)PW * W);
  const size_t lds_sliced = sizeof(float) * ((size_t)PH * H + (size_t)PW * W);
  if (lds > 60 * 1024) return AIT_EUNSUPPORTED;
  {
    // algorithmic bytes: the feature read once, the RoIs, the pooled tensor written once
    AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_ROI_FWD, 4.0 * ((double)B * C * H * W + 5.0 * n_rois + (double)n_rois * PH * PW * C), s,
                        n_rois, B, C, H, W);
    const size_t lds_sep = lds_sliced + (size_t)(kThreads / 32) * kPW * 32 * sizeof(float4);
    if (C % (4 * AIT_NXCD * 32) == 0 && lds_sep <= 60 * 1024 && PH < kThreads / 32 && kFwdSeparable)
      hipLaunchKernelGGL(roi_align_nhwc_fwd_sep_kernel, dim3((unsigned)n_rois * AIT_NXCD), dim3(kThreads), lds_sep, s, feat, rois,
                         n_rois, B, C, H, W, PH, wf, wi, out);
    else if (C % (4 * AIT_NXCD * 4) == 0 && lds_sliced <= 60 * 1024 && PH <= kThreads / 32)
      hipLaunchKernelGGL(roi_align_nhwc_fwd_sliced_kernel, dim3((unsigned)n_rois * AIT_NXCD), dim3(kThreads),
                         lds_sliced, s, feat, rois, n_rois, B, C, H, W, PH, wf, wi, out);
    else
      hipLaunchKernelGGL(roi_align_nhwc_fwd_kernel, dim3(chunked_grid((long long)n_rois * PH)), dim3(kThreads),
                         lds, s, feat, rois, n_rois, B, C, H, W, PH, wf, wi, out);
  }
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_roi_align_nhwc_bwd(const float* grad_out, const float* rois, int n_rois, int B, int C,
                                   int H, int W, int PH, int PW, float spatial_scale, int sampling_ratio,
                                   void* workspace, size_t workspace_bytes, float* grad_in, const ait_launch_ctx* ctx,
                                   void* stream) {
  if (bad(n_rois, B, C, H, W, PH, PW)) return AIT_EINVAL;
  if (unsupported(C, PH, PW)) return AIT_EUNSUPPORTED;
  if (!grad_in) return AIT_EINVAL;
  hipStream_t s = ait_stream(stream);
  if (n_rois == 0)
    return hipMemsetAsync(grad_in, 0, sizeof(float) * (size_t)B * H * W * C, s) == hipSuccess ? AIT_OK : AIT_ELAUNCH;
  if (!grad_out || !rois) return AIT_EINVAL;
  float* wf;
  int* wi;
  const int rc = make_tables(rois, n_rois, B, H, W, PH, PW, spatial_scale, sampling_ratio, workspace,
                             workspace_bytes, wf, wi, s);
  if (rc != AIT_OK) return rc;
  {
    AitProbeScope probe(ait_probe_of(ctx), AIT_PROBE_ROI_BWD, 4.0 * ((double)B * C * H * W + 5.0 * n_rois + (double)n_rois * PH * PW * C), s,
                        n_rois, B, C, H, W);
    if (C % (4 * AIT_NXCD) == 0 && C / (4 * AIT_NXCD) <= 32 && PH < 256 && PW < 256) {
      const long long tiles = (long long)B * ((H + 3) / 4) * ((W + 3) / 4);
      hipLaunchKernelGGL(roi_align_nhwc_bwd_tile_kernel, dim3((unsigned)(tiles * AIT_NXCD)), dim3(64), 0, s, grad_out, rois,
                         n_rois, B, C, H, W, PH, wf, wi, grad_in);
    } else {
      hipLaunchKernelGGL(roi_align_nhwc_bwd_kernel, dim3(chunked_grid((long long)B * H * W)), dim3(kThreads), 0, s,
                         grad_out, rois, n_rois, B, C, H, W, PH, wf, wi, grad_in);
    }
  }
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}
