/* oracle/native.c -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or
 * called from the product (ait_amd/).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it, and only as the checker.
 *
 * Plain-C restatement (single thread, fp32, no FMA contraction: build with
 * -ffp-contract=off) of the reference's native operators on the hot path:
 *
 *   orc_roi_align_fwd   follows lib/model/csrc/cpu/ROIAlign_cpu.cpp:17-219
 *                       (sample-point table :17-111, pooling loop :113-219)
 *   orc_roi_align_bwd   follows lib/model/csrc/cuda/ROIAlign_cuda.cu:125-254
 *                       (the reference has NO CPU backward, ROIAlign.h:44; this is the
 *                        scatter of that CUDA kernel run sequentially -- "unpinned by the
 *                        reference", pinned by finite differences in tests/)
 *   orc_nms             follows lib/model/csrc/cpu/nms_cpu.cpp:5-65 (suppress on
 *                       ovr >= thr, +1 pixel convention, ascending kept indices)
 *
 * Pinned against: oracle/_ref (the reference's own C++ compiled where it lies, when
 * /root/reference is present) and tests/golden/g4_roi_align.npz, g5_nms.npz.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int p[4];
  float w[4];
} tap_t;

/* One bilinear sample point -> 4 (offset, weight) taps.  ROIAlign_cpu.cpp:46-106. */
static void make_tap(float y, float x, int H, int W, tap_t* t) {
  if (y < -1.0 || y > H || x < -1.0 || x > W) {
    memset(t, 0, sizeof(*t));
    return;
  }
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y0 = (int)y, x0 = (int)x, y1, x1;
  if (y0 >= H - 1) {
    y1 = y0 = H - 1;
    y = (float)y0;
  } else {
    y1 = y0 + 1;
  }
  if (x0 >= W - 1) {
    x1 = x0 = W - 1;
    x = (float)x0;
  } else {
    x1 = x0 + 1;
  }
  float ly = y - y0, lx = x - x0;
  float hy = (float)(1. - ly), hx = (float)(1. - lx);
  t->p[0] = y0 * W + x0;
  t->p[1] = y0 * W + x1;
  t->p[2] = y1 * W + x0;
  t->p[3] = y1 * W + x1;
  t->w[0] = hy * hx;
  t->w[1] = hy * lx;
  t->w[2] = ly * hx;
  t->w[3] = ly * lx;
}

typedef struct {
  int b, gh, gw;
  float y0, x0, bh, bw, count;
} roi_geom_t;

/* RoI geometry.  ROIAlign_cpu.cpp:143-170 (no rounding of the scaled corners, 1x1 floor,
 * adaptive grid when sampling_ratio == 0). */
static roi_geom_t roi_geom(const float* r, float scale, int PH, int PW, int sr) {
  roi_geom_t g;
  g.b = (int)r[0];
  float sw = r[1] * scale, sh = r[2] * scale, ew = r[3] * scale, eh = r[4] * scale;
  float rw = ew - sw, rh = eh - sh;
  if (rw < 1.f) rw = 1.f; /* std::max(v, (T)1.) */
  if (rh < 1.f) rh = 1.f;
  g.y0 = sh;
  g.x0 = sw;
  g.bh = rh / (float)PH;
  g.bw = rw / (float)PW;
  g.gh = sr > 0 ? sr : (int)ceilf(rh / PH);
  g.gw = sr > 0 ? sr : (int)ceilf(rw / PW);
  g.count = (float)(g.gh * g.gw);
  return g;
}

static float sample_coord(float start, int p, float bin, int i, int grid) {
  /* roi_start + p*bin + (i + .5f)*bin/grid, evaluated left to right in fp32 */
  float a = start + p * bin;
  float b = (float)(i + .5f) * bin / (float)grid;
  return a + b;
}

int orc_roi_align_fwd(const float* feat, const float* rois, int n_rois, int B, int C, int H,
                      int W, int PH, int PW, float scale, int sr, float* out) {
  for (int n = 0; n < n_rois; n++) {
    roi_geom_t g = roi_geom(rois + 5 * n, scale, PH, PW, sr);
    if (g.b < 0 || g.b >= B) return -1;
    size_t ntap = (size_t)g.gh * g.gw * PH * PW;
    tap_t* taps = (tap_t*)malloc(sizeof(tap_t) * (ntap ? ntap : 1));
    size_t k = 0;
    for (int ph = 0; ph < PH; ph++)
      for (int pw = 0; pw < PW; pw++)
        for (int iy = 0; iy < g.gh; iy++) {
          float yy = sample_coord(g.y0, ph, g.bh, iy, g.gh);
          for (int ix = 0; ix < g.gw; ix++) {
            float xx = sample_coord(g.x0, pw, g.bw, ix, g.gw);
            make_tap(yy, xx, H, W, &taps[k++]);
          }
        }
    for (int c = 0; c < C; c++) {
      const float* plane = feat + ((size_t)g.b * C + c) * H * W;
      float* o = out + ((size_t)n * C + c) * PH * PW;
      k = 0;
      for (int bin = 0; bin < PH * PW; bin++) {
        float acc = 0.f;
        for (int s = 0; s < g.gh * g.gw; s++, k++) {
          const tap_t* t = &taps[k];
          acc += t->w[0] * plane[t->p[0]] + t->w[1] * plane[t->p[1]] +
                 t->w[2] * plane[t->p[2]] + t->w[3] * plane[t->p[3]];
        }
        o[bin] = acc / g.count;
      }
    }
    free(taps);
  }
  return 0;
}

/* grad_in [B,C,H,W] must be zeroed by the caller (ROIAlign_cuda.cu:316). */
int orc_roi_align_bwd(const float* grad_out, const float* rois, int n_rois, int B, int C,
                      int H, int W, int PH, int PW, float scale, int sr, float* grad_in) {
  for (int n = 0; n < n_rois; n++) {
    roi_geom_t g = roi_geom(rois + 5 * n, scale, PH, PW, sr);
    if (g.b < 0 || g.b >= B) return -1;
    for (int c = 0; c < C; c++) {
      float* plane = grad_in + ((size_t)g.b * C + c) * H * W;
      const float* go = grad_out + ((size_t)n * C + c) * PH * PW;
      for (int ph = 0; ph < PH; ph++)
        for (int pw = 0; pw < PW; pw++) {
          float d = go[ph * PW + pw];
          for (int iy = 0; iy < g.gh; iy++) {
            float yy = sample_coord(g.y0, ph, g.bh, iy, g.gh);
            for (int ix = 0; ix < g.gw; ix++) {
              float xx = sample_coord(g.x0, pw, g.bw, ix, g.gw);
              /* ROIAlign_cuda.cu:137-142: out-of-range sample contributes nothing */
              if (yy < -1.0 || yy > H || xx < -1.0 || xx > W) continue;
              tap_t t;
              make_tap(yy, xx, H, W, &t);
              for (int q = 0; q < 4; q++) plane[t.p[q]] += d * t.w[q] / g.count;
            }
          }
        }
    }
  }
  return 0;
}

/* Greedy NMS.  `order` = indices sorted by descending score (the caller sorts, as
 * nms_cpu.cpp:26 does with scores.sort); keep_out receives the surviving ORIGINAL indices
 * in ascending order (nms_cpu.cpp:64 nonzero(suppressed == 0)).  Returns the count. */
int64_t orc_nms(const float* dets, const int64_t* order, int64_t n, float thr,
                int64_t* keep_out) {
  if (n <= 0) return 0;
  float* area = (float*)malloc(sizeof(float) * n);
  uint8_t* dead = (uint8_t*)calloc(n, 1);
  for (int64_t i = 0; i < n; i++) {
    const float* d = dets + 4 * i;
    float w = d[2] - d[0] + 1, h = d[3] - d[1] + 1;
    area[i] = w * h;
  }
  for (int64_t a = 0; a < n; a++) {
    int64_t i = order ? order[a] : a;
    if (dead[i]) continue;
    const float* di = dets + 4 * i;
    for (int64_t b = a + 1; b < n; b++) {
      int64_t j = order ? order[b] : b;
      if (dead[j]) continue;
      const float* dj = dets + 4 * j;
      float xx1 = di[0] > dj[0] ? di[0] : dj[0];
      float yy1 = di[1] > dj[1] ? di[1] : dj[1];
      float xx2 = di[2] < dj[2] ? di[2] : dj[2];
      float yy2 = di[3] < dj[3] ? di[3] : dj[3];
      float w = xx2 - xx1 + 1, h = yy2 - yy1 + 1;
      if (w < 0) w = 0;
      if (h < 0) h = 0;
      float inter = w * h;
      float ovr = inter / (area[i] + area[j] - inter);
      if (ovr >= thr) dead[j] = 1;
    }
  }
  int64_t k = 0;
  for (int64_t i = 0; i < n; i++)
    if (!dead[i]) keep_out[k++] = i;
  free(area);
  free(dead);
  return k;
}
