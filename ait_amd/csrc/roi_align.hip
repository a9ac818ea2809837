// ait_amd/csrc/roi_align.hip -- RoIAlign forward / backward for gfx950 (MI355X).
//
// Behaviour follows the reference operator (lib/model/csrc/cpu/ROIAlign_cpu.cpp:113-219 for
// the forward arithmetic and its order, lib/model/csrc/cuda/ROIAlign_cuda.cu:178-254 for the
// backward scatter): spatial_scale without rounding, 1x1 floor on the RoI size, adaptive
// ceil(roi/P) sampling grid, out-of-range samples contribute zero, average over the grid.
// Built with -ffp-contract=off so the sample coordinates are computed with exactly the fp32
// operations of the reference (a coordinate one ulp off can flip an (int) truncation).
//
// MI355X design (HBM-bound gather / scatter, no matrix work):
//   * one 256-thread workgroup per (RoI, 32-channel tile).  The bilinear taps of a RoI depend
//     only on the RoI, so they are computed ONCE per workgroup into LDS (per-axis tables: the
//     taps are separable) and reused for all channels of the tile.
//   * threads run over (channel, bin) with the bin fastest, so the [n,C,7,7] output is written
//     in fully coalesced 16-B-per-4-lanes runs; the feature reads of neighbouring bins fall in
//     the same 128-B lines and are served by L1/L2.
//   * XCD-aware block order: blockIdx % 8 picks the XCD (round-robin dispatch), and every
//     channel tile is owned by exactly one XCD, so one tile's feature planes (<= 32 x 9.6 KB per
//     image) stay resident in that XCD's 4 MiB L2 while all RoIs stream past: the feature map
//     is fetched from HBM once per call, the rest of the traffic is the unavoidable output
//     write (forward) / grad read (backward).
//   * backward: instead of 4 atomics per sample per bin (the reference's scatter), the
//     workgroup forms the exact adjoint  dF[y][x] = sum_ph sum_pw Wy[ph][y] g[ph][pw] Wx[pw][x]
//     on chip (Wy, Wx = per-axis interpolation matrices in LDS) and issues ONE float atomic per
//     touched feature cell, contiguous in x (row segments), which is what the memory-side
//     atomic units want (MI355X_MICROARCH.md "Global float atomics").
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kCT = 32;        // channels per workgroup
constexpr int kMaxTab = 128;   // per-axis tap-table entries kept in LDS (P * grid)

struct AxisTap {
  int lo, hi;
  float wl, wh;  // weight of the hi / lo neighbour:  value = wh*f[lo] + wl*f[hi]
  int valid;
};

struct RoiGeom {
  int b, gh, gw;
  float y0, x0, bh, bw, count;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ r, float scale, int PH,
                                            int PW, int sr) {
  RoiGeom g;
  g.b = (int)r[0];
  float sw = r[1] * scale, sh = r[2] * scale, ew = r[3] * scale, eh = r[4] * scale;
  float rw = ew - sw, rh = eh - sh;
  if (rw < 1.f) rw = 1.f;
  if (rh < 1.f) rh = 1.f;
  g.y0 = sh;
  g.x0 = sw;
  g.bh = rh / (float)PH;
  g.bw = rw / (float)PW;
  g.gh = sr > 0 ? sr : (int)ceilf(rh / (float)PH);
  g.gw = sr > 0 ? sr : (int)ceilf(rw / (float)PW);
  g.count = (float)(g.gh * g.gw);
  return g;
}

// One axis of ROIAlign_cpu.cpp:36-95: sample coordinate -> (lo, hi, weights, validity).
__device__ __forceinline__ AxisTap axis_tap(float start, int p, float bin, int i, int grid,
                                            int L) {
  float a = start + (float)p * bin;
  float b = ((float)i + .5f) * bin / (float)grid;
  float v = a + b;
  AxisTap t;
  t.valid = !(v < -1.0f || v > (float)L);
  if (v <= 0.f) v = 0.f;
  int lo = (int)v, hi;
  if (lo >= L - 1) {
    hi = lo = L - 1;
    v = (float)lo;
  } else {
    hi = lo + 1;
  }
  t.lo = lo;
  t.hi = hi;
  t.wl = v - (float)lo;
  t.wh = 1.f - t.wl;
  if (!t.valid) {  // keep indices in range; weights forced to zero
    t.lo = t.hi = 0;
    t.wl = t.wh = 0.f;
  }
  return t;
}

// blockIdx -> (roi, channel tile).  Tiles are dealt to XCD groups: tile = xcd + 8*slot.
__device__ __forceinline__ bool decode_block(int n_rois, int C, int& n, int& c0) {
  int bid = blockIdx.x;
  int xcd = bid % AIT_NXCD;
  int j = bid / AIT_NXCD;
  int slot = j / n_rois;
  n = j - slot * n_rois;
  c0 = (xcd + AIT_NXCD * slot) * kCT;
  return c0 < C;
}

constexpr int kWinFloats = 8192;  // 32 KB LDS window: (RoI footprint) x (channels of one sub-tile)

// Forward.  The RoI's footprint [ymin..ymax] x [xmin..xmax] of every channel of the tile is
// staged ONCE in LDS by coalesced row-segment loads (each 128-B line of the feature map is
// fetched once per workgroup instead of once per tap), then every (channel, bin) output gathers
// its gh*gw*4 taps from LDS in the reference's order.  Falls back to direct global gathers when
// the tap tables or one channel's footprint do not fit.
__global__ __launch_bounds__(kThreads) void roi_align_fwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, int n_rois, int B, int C,
    int H, int W, int PH, int PW, float scale, int sr, float* __restrict__ out) {
  int n, c0;
  if (!decode_block(n_rois, C, n, c0)) return;
  const int bins = PH * PW;
  const RoiGeom g = roi_geom(rois + 5 * n, scale, PH, PW, sr);
  const int nc = min(kCT, C - c0);
  float* __restrict__ o = out + ((size_t)n * C + c0) * bins;
  if (g.b < 0 || g.b >= B) {
    for (int i = threadIdx.x; i < nc * bins; i += kThreads) o[i] = 0.f;
    return;
  }
  __shared__ AxisTap ty[kMaxTab], tx[kMaxTab];
  __shared__ int lim[4];
  __shared__ __attribute__((aligned(16))) float win[kWinFloats];
  const float* __restrict__ img = feat + ((size_t)g.b * C + c0) * H * W;
  const bool tab = PH * g.gh <= kMaxTab && PW * g.gw <= kMaxTab;  // block-uniform
  if (tab) {
    if (threadIdx.x == 0) {
      lim[0] = H; lim[1] = -1; lim[2] = W; lim[3] = -1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PH * g.gh; i += kThreads) {
      const AxisTap t = axis_tap(g.y0, i / g.gh, g.bh, i % g.gh, g.gh, H);
      ty[i] = t;
      if (t.valid) {
        atomicMin(&lim[0], t.lo);
        atomicMax(&lim[1], t.hi);
      }
    }
    for (int i = threadIdx.x; i < PW * g.gw; i += kThreads) {
      const AxisTap t = axis_tap(g.x0, i / g.gw, g.bw, i % g.gw, g.gw, W);
      tx[i] = t;
      if (t.valid) {
        atomicMin(&lim[2], t.lo);
        atomicMax(&lim[3], t.hi);
      }
    }
    __syncthreads();
    const int ymin = lim[0], ymax = lim[1], xmin = lim[2], xmax = lim[3];
    if (ymax < ymin || xmax < xmin) {  // every sample out of range
      for (int i = threadIdx.x; i < nc * bins; i += kThreads) o[i] = 0.f;
      return;
    }
    const int wh = ymax - ymin + 1, ww = xmax - xmin + 1, cells = wh * ww;
    if (cells <= kWinFloats) {
      __syncthreads();
      // make the tables window-relative: y offsets in floats of the staged window
      for (int i = threadIdx.x; i < PH * g.gh; i += kThreads) {
        AxisTap t = ty[i];
        t.lo = t.valid ? (t.lo - ymin) * ww : 0;
        t.hi = t.valid ? (t.hi - ymin) * ww : 0;
        ty[i] = t;
      }
      for (int i = threadIdx.x; i < PW * g.gw; i += kThreads) {
        AxisTap t = tx[i];
        t.lo = t.valid ? t.lo - xmin : 0;
        t.hi = t.valid ? t.hi - xmin : 0;
        tx[i] = t;
      }
      const int cs = min(nc, kWinFloats / cells);  // channels per sub-tile
      const float* __restrict__ corner = img + ymin * W + xmin;
      for (int cb = 0; cb < nc; cb += cs) {
        const int cn = min(cs, nc - cb);
        __syncthreads();
        for (int i = threadIdx.x; i < cn * cells; i += kThreads) {
          const int c = i / cells, r = i - c * cells;
          const int y = r / ww, x = r - y * ww;
          win[i] = corner[(size_t)(cb + c) * H * W + y * W + x];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cn * bins; i += kThreads) {
          const int c = i / bins, bin = i - c * bins;
          const int ph = bin / PW, pw = bin - ph * PW;
          const float* __restrict__ wc = win + c * cells;
          float acc = 0.f;
          for (int iy = 0; iy < g.gh; iy++) {
            const AxisTap y = ty[ph * g.gh + iy];
            const float* __restrict__ r0 = wc + y.lo;
            const float* __restrict__ r1 = wc + y.hi;
            for (int ix = 0; ix < g.gw; ix++) {
              const AxisTap x = tx[pw * g.gw + ix];
              if (y.valid && x.valid) {
                // same association as ROIAlign_cpu.cpp:197-200
                acc += (y.wh * x.wh) * r0[x.lo] + (y.wh * x.wl) * r0[x.hi] +
                       (y.wl * x.wh) * r1[x.lo] + (y.wl * x.wl) * r1[x.hi];
              }
            }
          }
          o[(cb + c) * bins + bin] = acc / g.count;
        }
      }
      return;
    }
  }
  // ---- fallback: direct global gathers (huge sampling grids / footprints) -------------------
  for (int i = threadIdx.x; i < nc * bins; i += kThreads) {
    const int c = i / bins, bin = i - c * bins;
    const int ph = bin / PW, pw = bin - ph * PW;
    const float* __restrict__ plane = img + (size_t)c * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < g.gh; iy++) {
      const AxisTap y = tab ? ty[ph * g.gh + iy] : axis_tap(g.y0, ph, g.bh, iy, g.gh, H);
      const float* __restrict__ r0 = plane + y.lo * W;
      const float* __restrict__ r1 = plane + y.hi * W;
      for (int ix = 0; ix < g.gw; ix++) {
        const AxisTap x = tab ? tx[pw * g.gw + ix] : axis_tap(g.x0, pw, g.bw, ix, g.gw, W);
        if (y.valid && x.valid) {
          acc += (y.wh * x.wh) * r0[x.lo] + (y.wh * x.wl) * r0[x.hi] +
                 (y.wl * x.wh) * r1[x.lo] + (y.wl * x.wl) * r1[x.hi];
        }
      }
    }
    o[i] = acc / g.count;
  }
}

// Backward, on-chip adjoint form.  Dynamic LDS: Wy[PH][H] | Wx[PW][W] | g[kCT][PH*PW].
__global__ __launch_bounds__(kThreads) void roi_align_bwd_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, int n_rois, int B, int C,
    int H, int W, int PH, int PW, float scale, int sr, float* __restrict__ grad_in) {
  int n, c0;
  if (!decode_block(n_rois, C, n, c0)) return;
  const RoiGeom g = roi_geom(rois + 5 * n, scale, PH, PW, sr);
  if (g.b < 0 || g.b >= B) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wy = smem;
  float* Wx = Wy + PH * H;
  float* gs = Wx + PW * W;
  __shared__ int lim[4];  // ymin, ymax, xmin, xmax of the touched window
  const int bins = PH * PW;
  const int nc = min(kCT, C - c0);
  for (int i = threadIdx.x; i < PH * H + PW * W; i += kThreads) smem[i] = 0.f;
  if (threadIdx.x == 0) {
    lim[0] = H;
    lim[1] = -1;
    lim[2] = W;
    lim[3] = -1;
  }
  const float* __restrict__ go = grad_out + ((size_t)n * C + c0) * bins;
  for (int i = threadIdx.x; i < nc * bins; i += kThreads) gs[i] = go[i];
  __syncthreads();
  // one thread per pooled row / column accumulates its samples in a fixed order
  if ((int)threadIdx.x < PH) {
    const int ph = threadIdx.x;
    int lo = H, hi = -1;
    for (int iy = 0; iy < g.gh; iy++) {
      AxisTap t = axis_tap(g.y0, ph, g.bh, iy, g.gh, H);
      if (!t.valid) continue;
      Wy[ph * H + t.lo] += t.wh;
      Wy[ph * H + t.hi] += t.wl;
      lo = min(lo, t.lo);
      hi = max(hi, t.hi);
    }
    if (hi >= 0) {
      atomicMin(&lim[0], lo);
      atomicMax(&lim[1], hi);
    }
  } else if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + PW) {
    const int pw = threadIdx.x - 64;
    int lo = W, hi = -1;
    for (int ix = 0; ix < g.gw; ix++) {
      AxisTap t = axis_tap(g.x0, pw, g.bw, ix, g.gw, W);
      if (!t.valid) continue;
      Wx[pw * W + t.lo] += t.wh;
      Wx[pw * W + t.hi] += t.wl;
      lo = min(lo, t.lo);
      hi = max(hi, t.hi);
    }
    if (hi >= 0) {
      atomicMin(&lim[2], lo);
      atomicMax(&lim[3], hi);
    }
  }
  __syncthreads();
  const int ymin = lim[0], ymax = lim[1], xmin = lim[2], xmax = lim[3];
  if (ymax < ymin || xmax < xmin) return;
  const int wh = ymax - ymin + 1, ww = xmax - xmin + 1;
  const float inv = 1.f / g.count;
  float* __restrict__ gi = grad_in + ((size_t)g.b * C + c0) * H * W;
  for (int i = threadIdx.x; i < nc * wh * ww; i += kThreads) {
    const int c = i / (wh * ww);
    const int r = i - c * (wh * ww);
    const int y = ymin + r / ww, x = xmin + r % ww;
    const float* __restrict__ gc = gs + c * bins;
    float v = 0.f;
    for (int ph = 0; ph < PH; ph++) {
      const float wy = Wy[ph * H + y];
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int pw = 0; pw < PW; pw++) row += gc[ph * PW + pw] * Wx[pw * W + x];
      v += wy * row;
    }
    if (v != 0.f) unsafeAtomicAdd(gi + (size_t)c * H * W + y * W + x, v * inv);
  }
}

// Generic fallback for feature maps whose Wy/Wx tables do not fit LDS: the reference's
// per-sample scatter (ROIAlign_cuda.cu:222-249), one thread per pooled element.
__global__ __launch_bounds__(kThreads) void roi_align_bwd_scatter_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, int n_rois, int B, int C,
    int H, int W, int PH, int PW, float scale, int sr, float* __restrict__ grad_in) {
  int n, c0;
  if (!decode_block(n_rois, C, n, c0)) return;
  const RoiGeom g = roi_geom(rois + 5 * n, scale, PH, PW, sr);
  if (g.b < 0 || g.b >= B) return;
  const int bins = PH * PW;
  const int nc = min(kCT, C - c0);
  const float* __restrict__ go = grad_out + ((size_t)n * C + c0) * bins;
  float* __restrict__ gi = grad_in + ((size_t)g.b * C + c0) * H * W;
  for (int i = threadIdx.x; i < nc * bins; i += kThreads) {
    const int c = i / bins, bin = i - c * bins;
    const int ph = bin / PW, pw = bin - ph * PW;
    const float d = go[i];
    float* __restrict__ plane = gi + (size_t)c * H * W;
    for (int iy = 0; iy < g.gh; iy++) {
      const AxisTap y = axis_tap(g.y0, ph, g.bh, iy, g.gh, H);
      for (int ix = 0; ix < g.gw; ix++) {
        const AxisTap x = axis_tap(g.x0, pw, g.bw, ix, g.gw, W);
        if (!(y.valid && x.valid)) continue;
        unsafeAtomicAdd(plane + y.lo * W + x.lo, d * (y.wh * x.wh) / g.count);
        unsafeAtomicAdd(plane + y.lo * W + x.hi, d * (y.wh * x.wl) / g.count);
        unsafeAtomicAdd(plane + y.hi * W + x.lo, d * (y.wl * x.wh) / g.count);
        unsafeAtomicAdd(plane + y.hi * W + x.hi, d * (y.wl * x.wl) / g.count);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Plane-resident kernels (the fast path for C4-sized feature maps, H*W <= ~3300 cells).
//
// One workgroup owns 4 whole channel planes of ONE image in LDS (4 x H x W fp32 = 38 KB at
// 38 x 63) and walks every RoI of that image; wave w = channel w.  The feature map (forward) /
// the feature gradient (backward) therefore crosses HBM exactly once, the backward needs NO
// atomics and NO zero-fill pass (each plane is accumulated privately by one wave in a fixed RoI
// order, then stored once -- bitwise reproducible, unlike the reference's atomicAdd scatter),
// and the only other traffic is the unavoidable [n,C,7,7] tensor.  RoIs are processed four at a
// time: each wave builds the per-axis tables of one RoI, then every wave applies all four.
// ---------------------------------------------------------------------------------------------
constexpr int kPG = 4;          // channel planes per workgroup (= waves)
constexpr int kSlotTab = 64;    // tap-table entries per axis per RoI slot (P * grid <= 64)
constexpr int kListMax = 2048;  // RoIs of one image handled per pass

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ordered list (ascending RoI index) of the RoIs in [base, base+kListMax) that belong to image b
__device__ __forceinline__ int build_roi_list(const float* __restrict__ rois, int n_rois, int base,
                                              int b, unsigned short* list, int* count) {
  if (threadIdx.x < 64) {  // wave 0: ballot compaction keeps the order deterministic
    int cnt = 0;
    const int end = min(n_rois, base + kListMax);
    for (int i0 = base; i0 < end; i0 += 64) {
      const int i = i0 + (int)threadIdx.x;
      const bool mine = i < end && (int)rois[5 * (size_t)i] == b;
      const unsigned long long m = __ballot(mine);
      if (mine) list[cnt + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = (unsigned short)(i - base);
      cnt += __popcll(m);
    }
    if (threadIdx.x == 0) *count = cnt;
  }
  __syncthreads();
  return *count;
}

struct SlotMeta {
  int roi, gh, gw, ok;   // ok: tables fit and the RoI has at least one in-range sample
  int ymin, ymax, xmin, xmax;
  float inv_count;
};

__global__ __launch_bounds__(kPG * 64) void roi_align_fwd_plane_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, int n_rois, int B, int C,
    int H, int W, int PH, int PW, float scale, int sr, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int HW = H * W, bins = PH * PW;
  float* plane = smem;                                               // [kPG][HW]
  AxisTap* tabs = reinterpret_cast<AxisTap*>(plane + kPG * HW);      // [kPG slots][2][kSlotTab]
  SlotMeta* meta = reinterpret_cast<SlotMeta*>(tabs + kPG * 2 * kSlotTab);
  unsigned short* list = reinterpret_cast<unsigned short*>(meta + kPG);
  int* count = reinterpret_cast<int*>(list + kListMax);
  const int groups = (C + kPG - 1) / kPG;
  const int b = blockIdx.x / groups, c0 = (blockIdx.x % groups) * kPG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = c0 + wave;
  const bool chan_ok = c < C;
  if (chan_ok) {
    const float* __restrict__ src = feat + ((size_t)b * C + c) * HW;
    for (int i = lane; i < HW; i += 64) plane[wave * HW + i] = src[i];
  }
  if (b == 0 && chan_ok) {  // RoIs whose batch index is out of range pool to zero
    for (int r = 0; r < n_rois; r++) {
      const int rb = (int)rois[5 * (size_t)r];
      if (rb < 0 || rb >= B)
        for (int i = lane; i < bins; i += 64) out[((size_t)r * C + c) * bins + i] = 0.f;
    }
  }
  for (int base = 0; base < n_rois; base += kListMax) {
    __syncthreads();
    const int cnt = build_roi_list(rois, n_rois, base, b, list, count);
    // software pipeline: the 5 floats of the RoI this wave prepares NEXT are fetched one group
    // ahead, so the global-memory latency hides behind the pooling of the current group
    float rnext[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (wave < cnt) {
      const float* __restrict__ rp = rois + 5 * (size_t)(base + list[wave]);
#pragma unroll
      for (int q = 0; q < 5; q++) rnext[q] = rp[q];
    }
    for (int g0 = 0; g0 < cnt; g0 += kPG) {
      __syncthreads();
      float rcur[5];
#pragma unroll
      for (int q = 0; q < 5; q++) rcur[q] = rnext[q];
      if (g0 + kPG + wave < cnt) {
        const float* __restrict__ rp = rois + 5 * (size_t)(base + list[g0 + kPG + wave]);
#pragma unroll
        for (int q = 0; q < 5; q++) rnext[q] = rp[q];
      }
      // ---- each wave prepares one RoI slot ---------------------------------------------------
      if (g0 + wave < cnt) {
        const int roi = base + list[g0 + wave];
        const RoiGeom g = roi_geom(rcur, scale, PH, PW, sr);
        AxisTap* ty = tabs + (wave * 2 + 0) * kSlotTab;
        AxisTap* tx = tabs + (wave * 2 + 1) * kSlotTab;
        const bool fits = PH * g.gh <= kSlotTab && PW * g.gw <= kSlotTab;
        if (fits) {
          for (int i = lane; i < PH * g.gh; i += 64) {
            AxisTap t = axis_tap(g.y0, i / g.gh, g.bh, i % g.gh, g.gh, H);
            t.lo *= W;
            t.hi *= W;
            ty[i] = t;
          }
          for (int i = lane; i < PW * g.gw; i += 64) tx[i] = axis_tap(g.x0, i / g.gw, g.bw, i % g.gw, g.gw, W);
        }
        if (lane == 0) {
          SlotMeta m;
          m.roi = roi; m.gh = g.gh; m.gw = g.gw; m.ok = fits;
          m.ymin = m.ymax = m.xmin = m.xmax = 0;
          m.inv_count = g.count;
          meta[wave] = m;
        }
      }
      __syncthreads();
      // ---- every wave (= channel) pools the four RoIs ----------------------------------------
      const int nslot = min(kPG, cnt - g0);
      if (chan_ok) {
        const float* __restrict__ pc = plane + wave * HW;
        for (int s = 0; s < nslot; s++) {
          const SlotMeta m = meta[s];
          float* __restrict__ o = out + ((size_t)m.roi * C + c) * bins;
          const AxisTap* ty = tabs + (s * 2 + 0) * kSlotTab;
          const AxisTap* tx = tabs + (s * 2 + 1) * kSlotTab;
          RoiGeom g;
          if (!m.ok) g = roi_geom(rois + 5 * (size_t)m.roi, scale, PH, PW, sr);
          for (int bin = lane; bin < bins; bin += 64) {
            const int ph = bin / PW, pw = bin - ph * PW;
            float acc = 0.f;
            for (int iy = 0; iy < m.gh; iy++) {
              AxisTap y;
              if (m.ok) y = ty[ph * m.gh + iy];
              else { y = axis_tap(g.y0, ph, g.bh, iy, g.gh, H); y.lo *= W; y.hi *= W; }
              const float* __restrict__ r0 = pc + y.lo;
              const float* __restrict__ r1 = pc + y.hi;
              for (int ix = 0; ix < m.gw; ix++) {
                const AxisTap x = m.ok ? tx[pw * m.gw + ix] : axis_tap(g.x0, pw, g.bw, ix, g.gw, W);
                if (y.valid && x.valid) {
                  // same association as ROIAlign_cpu.cpp:197-200
                  acc += (y.wh * x.wh) * r0[x.lo] + (y.wh * x.wl) * r0[x.hi] +
                         (y.wl * x.wh) * r1[x.lo] + (y.wl * x.wl) * r1[x.hi];
                }
              }
            }
            o[bin] = acc / m.inv_count;   // inv_count holds `count` in the forward
          }
        }
      }
    }
  }
}

// Backward.  Per slot: dense per-axis interpolation matrices Wy[PH][H], Wx[PW][W] (sum of the
// sample weights of each pooled row / column); per (RoI, channel):
//   T[ph][x] = sum_pw g[ph][pw] Wx[pw][x];   dF[y][x] += (sum_ph Wy[ph][y] T[ph][x]) / count
__global__ __launch_bounds__(kPG * 64) void roi_align_bwd_plane_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, int n_rois, int B, int C,
    int H, int W, int PH, int PW, float scale, int sr, float* __restrict__ grad_in) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int HW = H * W, bins = PH * PW;
  float* plane = smem;                                   // [kPG][HW]
  float* Wy = plane + kPG * HW;                          // [kPG slots][PH][H]
  float* Wx = Wy + kPG * PH * H;                         // [kPG slots][PW][W]
  float* gbuf = Wx + kPG * PW * W;                       // [kPG waves][bins]
  float* Tbuf = gbuf + kPG * bins;                       // [kPG waves][PH][W]
  SlotMeta* meta = reinterpret_cast<SlotMeta*>(Tbuf + kPG * PH * W);
  unsigned short* list = reinterpret_cast<unsigned short*>(meta + kPG);
  int* count = reinterpret_cast<int*>(list + kListMax);
  const int groups = (C + kPG - 1) / kPG;
  const int b = blockIdx.x / groups, c0 = (blockIdx.x % groups) * kPG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = c0 + wave;
  const bool chan_ok = c < C;
  for (int i = lane; i < HW; i += 64) plane[wave * HW + i] = 0.f;
  for (int base = 0; base < n_rois; base += kListMax) {
    __syncthreads();
    const int cnt = build_roi_list(rois, n_rois, base, b, list, count);
    // software pipeline (one group ahead): the RoI row this wave prepares next, and this wave's
    // channel of grad_out for the next four RoIs (lane < bins holds one value per slot)
    float rnext[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float gnext[kPG] = {0.f, 0.f, 0.f, 0.f};
    if (wave < cnt) {
      const float* __restrict__ rp = rois + 5 * (size_t)(base + list[wave]);
#pragma unroll
      for (int q = 0; q < 5; q++) rnext[q] = rp[q];
    }
    if (chan_ok && lane < bins) {
#pragma unroll
      for (int s = 0; s < kPG; s++)
        if (s < cnt) gnext[s] = grad_out[((size_t)(base + list[s]) * C + c) * bins + lane];
    }
    for (int g0 = 0; g0 < cnt; g0 += kPG) {
      __syncthreads();
      float rcur[5], gcur[kPG];
#pragma unroll
      for (int q = 0; q < 5; q++) rcur[q] = rnext[q];
#pragma unroll
      for (int s = 0; s < kPG; s++) gcur[s] = gnext[s];
      if (g0 + kPG + wave < cnt) {
        const float* __restrict__ rp = rois + 5 * (size_t)(base + list[g0 + kPG + wave]);
#pragma unroll
        for (int q = 0; q < 5; q++) rnext[q] = rp[q];
      }
      if (chan_ok && lane < bins) {
#pragma unroll
        for (int s = 0; s < kPG; s++)
          if (g0 + kPG + s < cnt)
            gnext[s] = grad_out[((size_t)(base + list[g0 + kPG + s]) * C + c) * bins + lane];
      }
      if (g0 + wave < cnt) {   // ---- this wave prepares slot `wave` -----------------------------
        const int roi = base + list[g0 + wave];
        const RoiGeom g = roi_geom(rcur, scale, PH, PW, sr);
        float* wy = Wy + wave * PH * H;
        float* wx = Wx + wave * PW * W;
        for (int i = lane; i < PH * H; i += 64) wy[i] = 0.f;
        for (int i = lane; i < PW * W; i += 64) wx[i] = 0.f;
        wave_sync();
        int lo = 1 << 30, hi = -1;
        if (lane < PH) {          // one lane per pooled row, samples in a fixed order
          for (int iy = 0; iy < g.gh; iy++) {
            const AxisTap t = axis_tap(g.y0, lane, g.bh, iy, g.gh, H);
            if (!t.valid) continue;
            wy[lane * H + t.lo] += t.wh;
            wy[lane * H + t.hi] += t.wl;
            lo = min(lo, t.lo);
            hi = max(hi, t.hi);
          }
        }
        int ymin = lo, ymax = hi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          ymin = min(ymin, __shfl_xor(ymin, o, 64));
          ymax = max(ymax, __shfl_xor(ymax, o, 64));
        }
        lo = 1 << 30; hi = -1;
        if (lane < PW) {
          for (int ix = 0; ix < g.gw; ix++) {
            const AxisTap t = axis_tap(g.x0, lane, g.bw, ix, g.gw, W);
            if (!t.valid) continue;
            wx[lane * W + t.lo] += t.wh;
            wx[lane * W + t.hi] += t.wl;
            lo = min(lo, t.lo);
            hi = max(hi, t.hi);
          }
        }
        int xmin = lo, xmax = hi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          xmin = min(xmin, __shfl_xor(xmin, o, 64));
          xmax = max(xmax, __shfl_xor(xmax, o, 64));
        }
        if (lane == 0) {
          SlotMeta m;
          m.roi = roi; m.gh = g.gh; m.gw = g.gw;
          m.ok = (ymax >= ymin) && (xmax >= xmin);
          m.ymin = ymin; m.ymax = ymax; m.xmin = xmin; m.xmax = xmax;
          m.inv_count = 1.f / g.count;
          meta[wave] = m;
        }
      }
      __syncthreads();
      const int nslot = min(kPG, cnt - g0);
      if (chan_ok) {           // ---- this wave (= channel) scatters the four RoIs ---------------
        float* pc = plane + wave * HW;
        float* gb = gbuf + wave * bins;
        float* tb = Tbuf + wave * PH * W;
#pragma unroll
        for (int s = 0; s < kPG; s++) {
          if (s >= nslot) break;
          const SlotMeta m = meta[s];
          if (!m.ok) continue;
          const float* wy = Wy + s * PH * H;
          const float* wx = Wx + s * PW * W;
          wave_sync();
          if (lane < bins) gb[lane] = gcur[s];
          wave_sync();
          const int ww = m.xmax - m.xmin + 1, wh = m.ymax - m.ymin + 1;
          for (int i = lane; i < PH * ww; i += 64) {
            const int ph = i / ww, x = m.xmin + (i - ph * ww);
            float t = 0.f;
            for (int pw = 0; pw < PW; pw++) t += gb[ph * PW + pw] * wx[pw * W + x];
            tb[ph * W + x] = t;
          }
          wave_sync();
          for (int i = lane; i < wh * ww; i += 64) {
            const int yy = i / ww;
            const int y = m.ymin + yy, x = m.xmin + (i - yy * ww);
            float v = 0.f;
            for (int ph = 0; ph < PH; ph++) v += wy[ph * H + y] * tb[ph * W + x];
            pc[y * W + x] += v * m.inv_count;
          }
        }
      }
    }
  }
  __syncthreads();
  if (chan_ok) {
    float* __restrict__ dst = grad_in + ((size_t)b * C + c) * HW;
    for (int i = lane; i < HW; i += 64) dst[i] = plane[wave * HW + i];
  }
}

inline size_t plane_lds_fwd(int H, int W) {
  return sizeof(float) * kPG * H * W + sizeof(AxisTap) * kPG * 2 * kSlotTab + sizeof(SlotMeta) * kPG +
         sizeof(unsigned short) * kListMax + 16;
}
inline size_t plane_lds_bwd(int H, int W, int PH, int PW) {
  return sizeof(float) * ((size_t)kPG * H * W + kPG * PH * H + kPG * PW * W + kPG * PH * PW +
                          kPG * PH * W) +
         sizeof(SlotMeta) * kPG + sizeof(unsigned short) * kListMax + 16;
}

inline bool bad_shape(int n_rois, int B, int C, int H, int W, int PH, int PW) {
  return n_rois < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 ||
         PH > 64 || PW > 64;
}

inline unsigned grid_blocks(int n_rois, int C) {
  int tiles = (C + kCT - 1) / kCT;
  int slots = (tiles + AIT_NXCD - 1) / AIT_NXCD;
  return (unsigned)(slots * n_rois * AIT_NXCD);
}

}  // namespace

AIT_API int ait_roi_align_fwd(const float* feat, const float* rois, int n_rois, int B, int C,
                                 int H, int W, int PH, int PW, float spatial_scale,
                                 int sampling_ratio, float* out, void* stream) {
  if (bad_shape(n_rois, B, C, H, W, PH, PW)) return AIT_EINVAL;
  if (n_rois == 0) return AIT_OK;
  if (!feat || !rois || !out) return AIT_EINVAL;
  // Measured on MI355X (bs=4, P=300, C=1024, scripts/bench_roi.py): the window-staged kernel
  // (0.81 ms) beats the plane-resident forward (1.05 ms, LDS-latency-bound at 8 waves/CU), so the
  // latter is kept only for the lab (lab_knobs.h: roi_fwd_plane).
  constexpr bool fwd_plane = ait_lab::Knobs::roi_fwd_plane;
  if (fwd_plane && plane_lds_fwd(H, W) <= 64 * 1024) {
    const unsigned blocks = (unsigned)(B * ((C + kPG - 1) / kPG));
    hipLaunchKernelGGL(roi_align_fwd_plane_kernel, dim3(blocks), dim3(kPG * 64), plane_lds_fwd(H, W),
                       ait_stream(stream), feat, rois, n_rois, B, C, H, W, PH, PW, spatial_scale,
                       sampling_ratio, out);
    AIT_CHECK_LAUNCH();
    return AIT_OK;
  }
  hipLaunchKernelGGL(roi_align_fwd_kernel, dim3(grid_blocks(n_rois, C)), dim3(kThreads), 0,
                     ait_stream(stream), feat, rois, n_rois, B, C, H, W, PH, PW, spatial_scale,
                     sampling_ratio, out);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_roi_align_bwd(const float* grad_out, const float* rois, int n_rois, int B,
                                 int C, int H, int W, int PH, int PW, float spatial_scale,
                                 int sampling_ratio, float* grad_in, void* stream) {
  if (bad_shape(n_rois, B, C, H, W, PH, PW)) return AIT_EINVAL;
  if (!grad_in) return AIT_EINVAL;
  if (plane_lds_bwd(H, W, PH, PW) <= 64 * 1024 && PH * PW <= 64 && (n_rois == 0 || (grad_out && rois))) {
    // plane-resident path: every plane is written exactly once, so no zero-fill pass
    const unsigned blocks = (unsigned)(B * ((C + kPG - 1) / kPG));
    hipLaunchKernelGGL(roi_align_bwd_plane_kernel, dim3(blocks), dim3(kPG * 64),
                       plane_lds_bwd(H, W, PH, PW), ait_stream(stream), grad_out, rois, n_rois, B, C,
                       H, W, PH, PW, spatial_scale, sampling_ratio, grad_in);
    AIT_CHECK_LAUNCH();
    return AIT_OK;
  }
  if (hipMemsetAsync(grad_in, 0, sizeof(float) * (size_t)B * C * H * W, ait_stream(stream)) !=
      hipSuccess)
    return AIT_ELAUNCH;
  if (n_rois == 0) return AIT_OK;
  if (!grad_out || !rois) return AIT_EINVAL;
  const size_t lds = sizeof(float) * ((size_t)PH * H + (size_t)PW * W + (size_t)kCT * PH * PW);
  if (lds <= 60 * 1024) {
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(grid_blocks(n_rois, C)), dim3(kThreads), lds,
                       ait_stream(stream), grad_out, rois, n_rois, B, C, H, W, PH, PW,
                       spatial_scale, sampling_ratio, grad_in);
  } else {
    hipLaunchKernelGGL(roi_align_bwd_scatter_kernel, dim3(grid_blocks(n_rois, C)),
                       dim3(kThreads), 0, ait_stream(stream), grad_out, rois, n_rois, B, C, H, W,
                       PH, PW, spatial_scale, sampling_ratio, grad_in);
  }
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}
