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
  const bool tab = PH * g.gh <= kMaxTab && PW * g.gw <= kMaxTab;  // block-uniform
  if (tab) {
    for (int i = threadIdx.x; i < PH * g.gh; i += kThreads)
      ty[i] = axis_tap(g.y0, i / g.gh, g.bh, i % g.gh, g.gh, H);
    for (int i = threadIdx.x; i < PW * g.gw; i += kThreads)
      tx[i] = axis_tap(g.x0, i / g.gw, g.bw, i % g.gw, g.gw, W);
    __syncthreads();
  }
  const float* __restrict__ img = feat + ((size_t)g.b * C + c0) * H * W;
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
          // same association as ROIAlign_cpu.cpp:197-200
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
