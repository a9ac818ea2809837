/* oracle/native_san_driver.c -- TEST INFRASTRUCTURE ONLY.  A stand-alone driver that runs the three functions of
 * oracle/native.c on the cases of a file and writes their outputs, so that the restatement (index arithmetic over
 * caller-sized buffers restating /root/reference/lib/model/csrc/cpu/ROIAlign_cpu.cpp:17-219 and nms_cpu.cpp:5-65) can be
 * built and run under -fsanitize=address,undefined on the host (tests/test_oracle_ops.py).  Every buffer is malloc'ed at
 * EXACTLY the size the interface states, so an out-of-range tap or index is a heap-buffer-overflow the sanitizer reports.
 *
 * file format, little endian: int32 n_cases, then per case
 *   int32 kind (0 RoIAlign fwd + bwd, 1 NMS)
 *   kind 0: int32 n_rois B C H W PH PW sr; float32 scale; float32 feat[B*C*H*W], rois[n*5], grad_out[n*C*PH*PW]
 *           -> out: int32 rc_fwd, rc_bwd; float32 y[n*C*PH*PW], grad_in[B*C*H*W]
 *   kind 1: int32 n; float32 thr; float32 dets[n*4]; int64 order[n]
 *           -> out: int64 kept; int64 keep[kept]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_roi_align_fwd(const float*, const float*, int, int, int, int, int, int, int, float, int, float*);
int orc_roi_align_bwd(const float*, const float*, int, int, int, int, int, int, int, float, int, float*);
int64_t orc_nms(const float*, const int64_t*, int64_t, float, int64_t*);

static void rd(void* p, size_t n, FILE* f) {
  if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
}
static void* alloc(size_t n) {          /* exact size; zero-size requests get their own 1-byte block */
  void* p = malloc(n ? n : 1);
  if (!p) { fprintf(stderr, "out of memory\n"); exit(4); }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* in = fopen(argv[1], "rb");
  FILE* out = fopen(argv[2], "wb");
  if (!in || !out) return 2;
  int32_t n_cases;
  rd(&n_cases, 4, in);
  for (int c = 0; c < n_cases; c++) {
    int32_t kind;
    rd(&kind, 4, in);
    if (kind == 0) {
      int32_t h[8];
      float scale;
      rd(h, sizeof h, in);
      rd(&scale, 4, in);
      const size_t n = h[0], B = h[1], C = h[2], H = h[3], W = h[4], PH = h[5], PW = h[6];
      const size_t nf = B * C * H * W, nr = n * 5, ny = n * C * PH * PW;
      float* feat = alloc(nf * 4); float* rois = alloc(nr * 4); float* go = alloc(ny * 4);
      float* y = alloc(ny * 4); float* gi = alloc(nf * 4);
      rd(feat, nf * 4, in); rd(rois, nr * 4, in); rd(go, ny * 4, in);
      memset(y, 0, ny * 4);
      memset(gi, 0, nf * 4);
      int32_t rc[2];
      rc[0] = orc_roi_align_fwd(feat, rois, (int)n, (int)B, (int)C, (int)H, (int)W, (int)PH, (int)PW, scale, h[7], y);
      rc[1] = orc_roi_align_bwd(go, rois, (int)n, (int)B, (int)C, (int)H, (int)W, (int)PH, (int)PW, scale, h[7], gi);
      fwrite(rc, 4, 2, out); fwrite(y, 4, ny, out); fwrite(gi, 4, nf, out);
      free(feat); free(rois); free(go); free(y); free(gi);
    } else {
      int32_t n;
      float thr;
      rd(&n, 4, in); rd(&thr, 4, in);
      float* dets = alloc((size_t)n * 16);
      int64_t* order = alloc((size_t)n * 8);
      int64_t* keep = alloc((size_t)n * 8);
      rd(dets, (size_t)n * 16, in); rd(order, (size_t)n * 8, in);
      int64_t k = orc_nms(dets, order, n, thr, keep);
      fwrite(&k, 8, 1, out); fwrite(keep, 8, (size_t)k, out);
      free(dets); free(order); free(keep);
    }
  }
  fclose(in);
  if (fclose(out) != 0) return 5;
  puts("ok");
  return 0;
}
