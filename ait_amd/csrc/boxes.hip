// ait_amd/csrc/boxes.hip -- the box arithmetic of the proposal layer and of the proposal-target layer as
// a handful of kernels.  Each replaces a chain of 15-40 elementwise / gather
// launches of 3-5 us each that the host could only issue every 10-30 us: the stretch between the RPN head
// and RoIAlign was 70-80 % GPU-idle (profiles/r02_step_timeline.txt), and a GPU left idle for milliseconds
// drops its shader clock for the GEMMs that follow.
//
// Built with -ffp-contract=off and written in the reference's operation order, so that every threshold
// comparison (IoU >= 0.5, clipping, fg / bg class membership) sees the fp32 value the reference's tensor
// expressions produce:
//   rpn_decode_kernel        lib/model/rpn/bbox_transform.py:74-117 (bbox_transform_inv, clip_boxes) and the
//                            fg-score / delta re-layout of proposal_layer.py:66-93
//   proposals_assemble_kernel proposal_layer.py:150-160 (survivors -> [b, post_nms_topN, 5], zero rows after)
//   roi_classify_kernel      proposal_target_layer_cascade.py:49-52,128-150 (gt boxes appended as RoIs, IoU,
//                            best gt, labels, fg / bg membership) + bbox_transform.py:167-211 (IoU with the
//                            zero-area conventions); emits the class SIZES the host-side RNG needs and the
//                            class member lists in ascending RoI order
//   roi_sample_gather_kernel proposal_target_layer_cascade.py:160-213 + 86-126 (gather the sampled RoIs,
//                            labels, regression targets, inside / outside weights)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void rpn_decode_kernel(const float* __restrict__ probs, const float* __restrict__ deltas,
                                                         const float* __restrict__ anchors, const float* __restrict__ im_info,
                                                         int b, int A, int HW, float* __restrict__ boxes,
                                                         float* __restrict__ scores) {
  const long long N = (long long)HW * A;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)b * N) return;
  const int img = (int)(i / N);
  const long long n = i - (long long)img * N;
  const int hw = (int)(n / A), a = (int)(n - (long long)hw * A);
  const float4 an = reinterpret_cast<const float4*>(anchors)[n];
  const float* d = deltas + ((size_t)img * 4 * A + 4 * a) * HW + hw;
  const float dx = d[0], dy = d[HW], dw = d[2 * (size_t)HW], dh = d[3 * (size_t)HW];
  const float w = an.z - an.x + 1.0f, h = an.w - an.y + 1.0f;
  const float cx = an.x + 0.5f * w, cy = an.y + 0.5f * h;
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  const float hx = im_info[img * 3 + 1] - 1.0f, hy = im_info[img * 3 + 0] - 1.0f;
  float4 o;
  o.x = fminf(fmaxf(pcx - 0.5f * pw, 0.0f), hx);
  o.y = fminf(fmaxf(pcy - 0.5f * ph, 0.0f), hy);
  o.z = fminf(fmaxf(pcx + 0.5f * pw, 0.0f), hx);
  o.w = fminf(fmaxf(pcy + 0.5f * ph, 0.0f), hy);
  reinterpret_cast<float4*>(boxes)[i] = o;
  scores[i] = probs[((size_t)img * 2 * A + A + a) * HW + hw];
}

__global__ __launch_bounds__(256) void proposals_assemble_kernel(const float* __restrict__ cand, int n,
                                                                 const int64_t* __restrict__ keep, long long keep_stride,
                                                                 int keep_cols, const int* __restrict__ n_keep, int b,
                                                                 int post_n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * post_n) return;
  const int img = i / post_n, r = i - img * post_n;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < n_keep[img] && r < keep_cols) {
    long long j = keep[(size_t)img * keep_stride + r];
    j = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);
    v = reinterpret_cast<const float4*>(cand)[(size_t)img * n + j];
  }
  float* o = out + (size_t)i * 5;
  o[0] = (float)img; o[1] = v.x; o[2] = v.y; o[3] = v.z; o[4] = v.w;
}

constexpr int kClsThreads = 256;

// exclusive prefix sum of one int per thread across the workgroup; returns the total in `total`
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int& total) {
  const int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int o = 1; o < kClsThreads; o <<= 1) {
    const int x = t >= o ? lds[t - o] : 0;
    __syncthreads();
    lds[t] += x;
    __syncthreads();
  }
  total = lds[kClsThreads - 1];
  const int ex = lds[t] - v;
  __syncthreads();
  return ex;
}

// one workgroup per image; thread t owns the RoIs [t*per, (t+1)*per): member lists come out in ascending order
__global__ __launch_bounds__(kClsThreads) void roi_classify_kernel(
    const float* __restrict__ rois, int R0, const float* __restrict__ gt, int G, int gt_cols, float fg_thr, float bg_hi,
    float bg_lo, float* __restrict__ all_rois, int64_t* __restrict__ assign, float* __restrict__ labels,
    int64_t* __restrict__ counts, int64_t* __restrict__ fg_members, int64_t* __restrict__ bg_members,
    unsigned char* __restrict__ cls /* [b, R] scratch: bit0 fg, bit1 bg */) {
  __shared__ int scan[kClsThreads];
  extern __shared__ float gts[];          // [G][5]: x1 y1 x2 y2 area, then flags
  const int img = blockIdx.x, R = R0 + G;
  const float* g_img = gt + (size_t)img * G * gt_cols;
  float* gflag = gts + (size_t)G * 5;     // 1 when the gt box has zero area
  for (int k = threadIdx.x; k < G; k += kClsThreads) {
    const float x1 = g_img[k * gt_cols], y1 = g_img[k * gt_cols + 1], x2 = g_img[k * gt_cols + 2], y2 = g_img[k * gt_cols + 3];
    const float gw = x2 - x1 + 1.0f, gh = y2 - y1 + 1.0f;
    gts[k * 5] = x1; gts[k * 5 + 1] = y1; gts[k * 5 + 2] = x2; gts[k * 5 + 3] = y2; gts[k * 5 + 4] = gw * gh;
    gflag[k] = (gw == 1.0f && gh == 1.0f) ? 1.0f : 0.0f;
  }
  __syncthreads();
  const int per = (R + kClsThreads - 1) / kClsThreads;
  const int r_lo = min(R, (int)threadIdx.x * per), r_hi = min(R, r_lo + per);
  int n_fg = 0, n_bg = 0;
  for (int r = r_lo; r < r_hi; r++) {
    float x1, y1, x2, y2, c0;
    if (r < R0) {
      const float* p = rois + ((size_t)img * R0 + r) * 5;
      c0 = p[0]; x1 = p[1]; y1 = p[2]; x2 = p[3]; y2 = p[4];
    } else {                              // the gt boxes as RoIs (batch column 0, as the reference leaves it)
      const float* p = g_img + (size_t)(r - R0) * gt_cols;
      c0 = 0.0f; x1 = p[0]; y1 = p[1]; x2 = p[2]; y2 = p[3];
    }
    float* o = all_rois + ((size_t)img * R + r) * 5;
    o[0] = c0; o[1] = x1; o[2] = y1; o[3] = x2; o[4] = y2;
    const float aw = x2 - x1 + 1.0f, ah = y2 - y1 + 1.0f, a_area = aw * ah;
    const bool a_zero = (aw == 1.0f && ah == 1.0f);
    float best = 0.0f;
    int arg = 0;
    for (int k = 0; k < G; k++) {
      float iw = fminf(x2, gts[k * 5 + 2]) - fmaxf(x1, gts[k * 5]) + 1.0f;
      float ih = fminf(y2, gts[k * 5 + 3]) - fmaxf(y1, gts[k * 5 + 1]) + 1.0f;
      iw = fmaxf(iw, 0.0f); ih = fmaxf(ih, 0.0f);
      const float inter = iw * ih;
      float ov = inter / (a_area + gts[k * 5 + 4] - inter);
      if (gflag[k] != 0.0f) ov = 0.0f;
      if (a_zero) ov = -1.0f;
      if (k == 0 || ov > best) { best = ov; arg = k; }      // first maximum
    }
    assign[(size_t)img * R + r] = arg;
    labels[(size_t)img * R + r] = g_img[(size_t)arg * gt_cols + 4];
    const bool fg = best >= fg_thr, bg = (best < bg_hi) && (best >= bg_lo);
    cls[(size_t)img * R + r] = (unsigned char)((fg ? 1 : 0) | (bg ? 2 : 0));
    n_fg += fg; n_bg += bg;
  }
  int tot_fg, tot_bg;
  const int off_fg = block_exclusive_scan(n_fg, scan, tot_fg);
  const int off_bg = block_exclusive_scan(n_bg, scan, tot_bg);
  if (threadIdx.x == 0) { counts[img * 2] = tot_fg; counts[img * 2 + 1] = tot_bg; }
  // members first (ascending), the rest after them (ascending): what a stable sort of the class mask gives
  int f = off_fg, nf = tot_fg + (r_lo - off_fg), g2 = off_bg, ng = tot_bg + (r_lo - off_bg);
  for (int r = r_lo; r < r_hi; r++) {
    const unsigned char c = cls[(size_t)img * R + r];
    if (c & 1) fg_members[(size_t)img * R + f++] = r; else fg_members[(size_t)img * R + nf++] = r;
    if (c & 2) bg_members[(size_t)img * R + g2++] = r; else bg_members[(size_t)img * R + ng++] = r;
  }
}

struct SampleConsts { float mean[4], stdv[4], inside[4]; int normalize; };

// one workgroup per image
__global__ __launch_bounds__(256) void roi_sample_gather_kernel(
    const int64_t* __restrict__ pos, const int64_t* __restrict__ n_fg, int P, int R, const int64_t* __restrict__ fg_members,
    const int64_t* __restrict__ bg_members, const float* __restrict__ labels, const float* __restrict__ all_rois,
    const int64_t* __restrict__ assign, const float* __restrict__ gt, int G, int gt_cols, SampleConsts c,
    float* __restrict__ rois_b, float* __restrict__ labels_b, float* __restrict__ targets, float* __restrict__ inside_w,
    float* __restrict__ outside_w) {
  __shared__ int any_pos;
  const int img = blockIdx.x;
  if (threadIdx.x == 0) any_pos = 0;
  __syncthreads();
  const long long nf = n_fg[img];
  // pass 1: labels (the image-level "any foreground label" switch needs all of them)
  for (int s = threadIdx.x; s < P; s += blockDim.x) {
    const bool is_fg = s < nf;
    long long p = pos[(size_t)img * P + s];
    p = p < 0 ? 0 : (p > R - 1 ? R - 1 : p);
    const long long keep = (is_fg ? fg_members : bg_members)[(size_t)img * R + p];
    const float lab = labels[(size_t)img * R + keep] * (is_fg ? 1.0f : 0.0f);
    labels_b[(size_t)img * P + s] = lab;
    if (lab != 0.0f) atomicOr(&any_pos, 1);       // labels are class ids >= 0: sum != 0 <=> some label != 0
  }
  __syncthreads();
  const float img_on = any_pos ? 1.0f : 0.0f;
  for (int s = threadIdx.x; s < P; s += blockDim.x) {
    const bool is_fg = s < nf;
    long long p = pos[(size_t)img * P + s];
    p = p < 0 ? 0 : (p > R - 1 ? R - 1 : p);
    const long long keep = (is_fg ? fg_members : bg_members)[(size_t)img * R + p];
    const float* r = all_rois + ((size_t)img * R + keep) * 5;
    const float ex1 = r[1], ey1 = r[2], ex2 = r[3], ey2 = r[4];
    float* o = rois_b + ((size_t)img * P + s) * 5;
    o[0] = (float)img; o[1] = ex1; o[2] = ey1; o[3] = ex2; o[4] = ey2;
    const float* gb = gt + ((size_t)img * G + assign[(size_t)img * R + keep]) * gt_cols;
    const float ew = ex2 - ex1 + 1.0f, eh = ey2 - ey1 + 1.0f;
    const float ecx = ex1 + 0.5f * ew, ecy = ey1 + 0.5f * eh;
    const float gw = gb[2] - gb[0] + 1.0f, gh = gb[3] - gb[1] + 1.0f;
    const float gcx = gb[0] + 0.5f * gw, gcy = gb[1] + 0.5f * gh;
    float t[4] = {(gcx - ecx) / ew, (gcy - ecy) / eh, logf(gw / ew), logf(gh / eh)};
    const float lab = labels_b[(size_t)img * P + s];
    const float on = (lab > 0.0f ? 1.0f : 0.0f) * img_on;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (c.normalize) t[k] = (t[k] - c.mean[k]) / c.stdv[k];
      const float iw = on * c.inside[k];
      // (a select, not a product: the reference ASSIGNS targets at the foreground rows only,
      // proposal_target_layer_cascade.py:101-107 -- a non-finite t of a degenerate box must not leak as NaN * 0)
      targets[((size_t)img * P + s) * 4 + k] = on != 0.0f ? t[k] : 0.0f;
      inside_w[((size_t)img * P + s) * 4 + k] = iw;
      outside_w[((size_t)img * P + s) * 4 + k] = iw > 0.0f ? 1.0f : 0.0f;
    }
  }
}

// ---- anchor-target layer (lib/model/rpn/anchor_target_layer.py:55-187) ------------------------------------
// IoU of one inside anchor against the image's gt boxes (bbox_transform.py:119-165: zero-area gt -> 0)
__device__ __forceinline__ float anchor_iou(const float4 an, float a_area, const float* g /* x1 y1 x2 y2 area flag */) {
  float iw = fminf(an.z, g[2]) - fmaxf(an.x, g[0]) + 1.0f;
  float ih = fminf(an.w, g[3]) - fmaxf(an.y, g[1]) + 1.0f;
  iw = fmaxf(iw, 0.0f); ih = fmaxf(ih, 0.0f);
  const float inter = iw * ih;
  float ov = inter / (a_area + g[4] - inter);
  if (g[5] != 0.0f) ov = 0.0f;
  return ov;
}
__device__ __forceinline__ void load_gts(const float* g_img, int G, int gt_cols, float* gts) {
  for (int k = threadIdx.x; k < G; k += blockDim.x) {
    const float x1 = g_img[k * gt_cols], y1 = g_img[k * gt_cols + 1], x2 = g_img[k * gt_cols + 2], y2 = g_img[k * gt_cols + 3];
    const float gw = x2 - x1 + 1.0f, gh = y2 - y1 + 1.0f;
    float* g = gts + k * 6;
    g[0] = x1; g[1] = y1; g[2] = x2; g[3] = y2; g[4] = gw * gh; g[5] = (gw == 1.0f && gh == 1.0f) ? 1.0f : 0.0f;
  }
  __syncthreads();
}

// pass 1: best gt per anchor, and the best overlap per gt box over all anchors (IoU >= 0: its fp32 bits order
// like the values, so an integer atomicMax is the float maximum)
__global__ __launch_bounds__(256) void anchor_iou_kernel(const float* __restrict__ anchors, int n_in, const float* __restrict__ gt,
                                                         int G, int gt_cols, float* __restrict__ max_ov,
                                                         int64_t* __restrict__ argmax, int* __restrict__ gt_max_bits) {
  extern __shared__ float gts[];
  const int img = blockIdx.y;
  load_gts(gt + (size_t)img * G * gt_cols, G, gt_cols, gts);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  const float4 an = reinterpret_cast<const float4*>(anchors)[i];
  const float a_area = (an.z - an.x + 1.0f) * (an.w - an.y + 1.0f);
  float best = 0.0f;
  int arg = 0;
  for (int k = 0; k < G; k++) {
    const float ov = anchor_iou(an, a_area, gts + k * 6);
    if (k == 0 || ov > best) { best = ov; arg = k; }
    if (ov > 0.0f) atomicMax(gt_max_bits + img * G + k, __float_as_int(ov));
  }
  max_ov[(size_t)img * n_in + i] = best;
  argmax[(size_t)img * n_in + i] = arg;
}

// pass 2: labels (anchor_target_layer.py:106-124) and the two class sizes per image
__global__ __launch_bounds__(256) void anchor_label_kernel(const float* __restrict__ anchors, int n_in, const float* __restrict__ gt,
                                                           int G, int gt_cols, const float* __restrict__ max_ov,
                                                           const int* __restrict__ gt_max_bits, float neg_thr, float pos_thr,
                                                           int clobber, float* __restrict__ labels, int64_t* __restrict__ counts) {
  extern __shared__ float gts[];
  __shared__ int n1, n0;
  const int img = blockIdx.y;
  if (threadIdx.x == 0) { n1 = 0; n0 = 0; }
  load_gts(gt + (size_t)img * G * gt_cols, G, gt_cols, gts);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_in) {
    const float4 an = reinterpret_cast<const float4*>(anchors)[i];
    const float a_area = (an.z - an.x + 1.0f) * (an.w - an.y + 1.0f);
    const float mo = max_ov[(size_t)img * n_in + i];
    float lab = -1.0f;
    if (!clobber && mo < neg_thr) lab = 0.0f;
    bool is_best = false;
    for (int k = 0; k < G; k++) {
      float gm = __int_as_float(gt_max_bits[img * G + k]);
      if (gm == 0.0f) gm = 1e-5f;
      is_best |= (anchor_iou(an, a_area, gts + k * 6) == gm);
    }
    if (is_best) lab = 1.0f;
    if (mo >= pos_thr) lab = 1.0f;
    if (clobber && mo < neg_thr) lab = 0.0f;
    labels[(size_t)img * n_in + i] = lab;
    if (lab == 1.0f) atomicAdd(&n1, 1);
    if (lab == 0.0f) atomicAdd(&n0, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (n1) atomicAdd(reinterpret_cast<unsigned long long*>(counts + img * 2), (unsigned long long)n1);
    if (n0) atomicAdd(reinterpret_cast<unsigned long long*>(counts + img * 2 + 1), (unsigned long long)n0);
  }
}

// class member lists in ascending anchor order (members only; one workgroup per image)
__global__ __launch_bounds__(kClsThreads) void anchor_members_kernel(const float* __restrict__ labels, int n_in,
                                                                     int* __restrict__ fg_members, int* __restrict__ bg_members) {
  __shared__ int scan[kClsThreads];
  const int img = blockIdx.x;
  const float* lab = labels + (size_t)img * n_in;
  const int per = (n_in + kClsThreads - 1) / kClsThreads;
  const int lo = min(n_in, (int)threadIdx.x * per), hi = min(n_in, lo + per);
  int c1 = 0, c0 = 0;
  for (int i = lo; i < hi; i++) { c1 += lab[i] == 1.0f; c0 += lab[i] == 0.0f; }
  int t1, t0;
  int o1 = block_exclusive_scan(c1, scan, t1);
  int o0 = block_exclusive_scan(c0, scan, t0);
  for (int i = lo; i < hi; i++) {
    if (lab[i] == 1.0f) fg_members[(size_t)img * n_in + o1++] = i;
    if (lab[i] == 0.0f) bg_members[(size_t)img * n_in + o0++] = i;
  }
}

// the subsampling draws: label of the pos[j]-th member of the class becomes -1 (anchor_target_layer.py:131-150)
__global__ __launch_bounds__(256) void anchor_disable_kernel(const int64_t* __restrict__ pos, const int* __restrict__ n_pos, int m,
                                                             const int* __restrict__ members, int n_in, float* __restrict__ labels) {
  const int img = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_pos[img]) return;
  const int64_t p = pos[(size_t)img * m + j];
  if (p < 0 || p >= n_in) return;
  labels[(size_t)img * n_in + members[(size_t)img * n_in + p]] = -1.0f;
}

// regression targets, weights, and the scatter of the inside anchors back onto the [H, W, A] grid in the
// layouts the RPN losses read (anchor_target_layer.py:152-187)
__global__ __launch_bounds__(256) void anchor_targets_kernel(const float* __restrict__ anchors, const int* __restrict__ inside_pos,
                                                             int n_in, int A, int H, int W, const float* __restrict__ gt, int G,
                                                             int gt_cols, const int64_t* __restrict__ argmax,
                                                             const float* __restrict__ labels, float inside_weight,
                                                             float outside_weight, float* __restrict__ labels_out,
                                                             float* __restrict__ targets, float* __restrict__ inside_w,
                                                             float* __restrict__ outside_w) {
  const int img = blockIdx.y;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;        // (h, w, a) order
  const int HW = H * W;
  if (n >= HW * A) return;
  const int hw = n / A, a = n - hw * A, h = hw / W, w = hw - h * W;
  const int i = inside_pos[n];
  float lab = -1.0f, t[4] = {0.f, 0.f, 0.f, 0.f}, wi = 0.0f, wo = 0.0f;
  if (i >= 0) {
    lab = labels[(size_t)img * n_in + i];
    const float4 an = reinterpret_cast<const float4*>(anchors)[i];
    const float* gb = gt + ((size_t)img * G + argmax[(size_t)img * n_in + i]) * gt_cols;
    const float ew = an.z - an.x + 1.0f, eh = an.w - an.y + 1.0f;
    const float ecx = an.x + 0.5f * ew, ecy = an.y + 0.5f * eh;
    const float gw = gb[2] - gb[0] + 1.0f, gh = gb[3] - gb[1] + 1.0f;
    const float gcx = gb[0] + 0.5f * gw, gcy = gb[1] + 0.5f * gh;
    t[0] = (gcx - ecx) / ew; t[1] = (gcy - ecy) / eh; t[2] = logf(gw / ew); t[3] = logf(gh / eh);
    wi = (lab == 1.0f ? 1.0f : 0.0f) * inside_weight;
    wo = (lab >= 0.0f ? 1.0f : 0.0f) * outside_weight;
  }
  labels_out[((size_t)img * A + a) * HW + hw] = lab;           // [b, 1, A*H, W]: row a*H + h, column w
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const size_t o = ((size_t)img * 4 * A + 4 * a + c) * HW + hw;   // [b, 4A, H, W]
    targets[o] = t[c]; inside_w[o] = wi; outside_w[o] = wo;
  }
  (void)h; (void)w;
}

}  // namespace

AIT_API int ait_rpn_decode(const float* probs, const float* deltas, const float* anchors, const float* im_info, int b,
                           int A, int H, int W, float* boxes, float* scores, void* stream) {
  if (b < 0 || A <= 0 || H <= 0 || W <= 0) return AIT_EINVAL;
  if (b == 0) return AIT_OK;
  if (!probs || !deltas || !anchors || !im_info || !boxes || !scores) return AIT_EINVAL;
  const long long total = (long long)b * H * W * A;
  if (total >= (1ll << 31)) return AIT_EUNSUPPORTED;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ait_stream(stream), probs,
                     deltas, anchors, im_info, b, A, H * W, boxes, scores);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_proposals_assemble(const float* cand, int n, const int64_t* keep, long long keep_stride, int keep_cols,
                                   const int32_t* n_keep, int b, int post_n, float* out, void* stream) {
  if (b < 0 || n <= 0 || post_n <= 0 || keep_cols < 0) return AIT_EINVAL;
  if (b == 0) return AIT_OK;
  if (!cand || !keep || !n_keep || !out) return AIT_EINVAL;
  hipLaunchKernelGGL(proposals_assemble_kernel, dim3((unsigned)((b * post_n + 255) / 256)), dim3(256), 0,
                     ait_stream(stream), cand, n, keep, keep_stride, keep_cols, n_keep, b, post_n, out);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API size_t ait_roi_classify_workspace_bytes(int b, int R0, int G) {
  return b > 0 && R0 >= 0 && G > 0 ? (((size_t)b * (R0 + G) + 255) / 256) * 256 : 0;
}

AIT_API int ait_roi_classify(const float* rois, int b, int R0, const float* gt, int G, int gt_cols, float fg_thresh,
                             float bg_thresh_hi, float bg_thresh_lo, void* workspace, size_t workspace_bytes,
                             float* all_rois, int64_t* assign, float* labels, int64_t* counts,
                             int64_t* fg_members, int64_t* bg_members, void* stream) {
  if (b < 0 || R0 < 0 || G <= 0 || gt_cols < 5) return AIT_EINVAL;
  if (b == 0) return AIT_OK;
  if (!gt || (!rois && R0 > 0) || !all_rois || !assign || !labels || !counts || !fg_members || !bg_members || !workspace)
    return AIT_EINVAL;
  if (workspace_bytes < ait_roi_classify_workspace_bytes(b, R0, G)) return AIT_EWORKSPACE;
  const size_t lds = (size_t)G * 6 * sizeof(float);
  if (lds > 48 * 1024) return AIT_EUNSUPPORTED;
  hipLaunchKernelGGL(roi_classify_kernel, dim3(b), dim3(kClsThreads), lds, ait_stream(stream), rois, R0, gt, G, gt_cols,
                     fg_thresh, bg_thresh_hi, bg_thresh_lo, all_rois, assign, labels, counts, fg_members, bg_members,
                     static_cast<unsigned char*>(workspace));
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_roi_sample_gather(const int64_t* pos, const int64_t* n_fg, int b, int P, int R,
                                  const int64_t* fg_members, const int64_t* bg_members, const float* labels,
                                  const float* all_rois, const int64_t* assign, const float* gt, int G, int gt_cols,
                                  const float* means, const float* stds, const float* inside_weights, int normalize,
                                  float* rois_b, float* labels_b, float* bbox_targets, float* inside_w, float* outside_w,
                                  void* stream) {
  if (b < 0 || P <= 0 || R <= 0 || G <= 0 || gt_cols < 5) return AIT_EINVAL;
  if (b == 0) return AIT_OK;
  if (!pos || !n_fg || !fg_members || !bg_members || !labels || !all_rois || !assign || !gt || !means || !stds ||
      !inside_weights || !rois_b || !labels_b || !bbox_targets || !inside_w || !outside_w)
    return AIT_EINVAL;
  SampleConsts c;
  for (int k = 0; k < 4; k++) { c.mean[k] = means[k]; c.stdv[k] = stds[k]; c.inside[k] = inside_weights[k]; }   // host arrays
  c.normalize = normalize;
  hipLaunchKernelGGL(roi_sample_gather_kernel, dim3(b), dim3(256), 0, ait_stream(stream), pos, n_fg, P, R, fg_members,
                     bg_members, labels, all_rois, assign, gt, G, gt_cols, c, rois_b, labels_b, bbox_targets, inside_w,
                     outside_w);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_anchor_classify(const float* anchors_inside, int n_in, const float* gt, int b, int G, int gt_cols,
                                float negative_overlap, float positive_overlap, int clobber_positives, float* max_ov,
                                int64_t* argmax, int* gt_max_bits, float* labels, int64_t* counts, int* fg_members,
                                int* bg_members, void* stream) {
  if (b < 0 || n_in <= 0 || G <= 0 || gt_cols < 4) return AIT_EINVAL;
  if (b == 0) return AIT_OK;
  if (!anchors_inside || !gt || !max_ov || !argmax || !gt_max_bits || !labels || !counts || !fg_members || !bg_members)
    return AIT_EINVAL;
  const size_t lds = (size_t)G * 6 * sizeof(float);
  if (lds > 48 * 1024) return AIT_EUNSUPPORTED;
  hipStream_t s = ait_stream(stream);
  if (hipMemsetAsync(gt_max_bits, 0, (size_t)b * G * sizeof(int), s) != hipSuccess ||
      hipMemsetAsync(counts, 0, (size_t)b * 2 * sizeof(int64_t), s) != hipSuccess)
    return AIT_ELAUNCH;
  const dim3 grid((unsigned)((n_in + 255) / 256), (unsigned)b);
  hipLaunchKernelGGL(anchor_iou_kernel, grid, dim3(256), lds, s, anchors_inside, n_in, gt, G, gt_cols, max_ov, argmax,
                     gt_max_bits);
  AIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(anchor_label_kernel, grid, dim3(256), lds, s, anchors_inside, n_in, gt, G, gt_cols, max_ov,
                     gt_max_bits, negative_overlap, positive_overlap, clobber_positives, labels, counts);
  AIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(anchor_members_kernel, dim3(b), dim3(kClsThreads), 0, s, labels, n_in, fg_members, bg_members);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_anchor_targets(const float* anchors_inside, const int* inside_pos, int n_in, int A, int H, int W,
                               const float* gt, int b, int G, int gt_cols, const int64_t* argmax, float* labels,
                               const int64_t* fg_drop, const int* n_fg_drop, int m_fg, const int* fg_members,
                               const int64_t* bg_drop, const int* n_bg_drop, int m_bg, const int* bg_members,
                               float inside_weight, float outside_weight, float* labels_out, float* targets,
                               float* inside_w, float* outside_w, void* stream) {
  if (b < 0 || n_in <= 0 || A <= 0 || H <= 0 || W <= 0 || G <= 0 || gt_cols < 4 || m_fg < 0 || m_bg < 0) return AIT_EINVAL;
  if (b == 0) return AIT_OK;
  if (!anchors_inside || !inside_pos || !gt || !argmax || !labels || !labels_out || !targets || !inside_w || !outside_w ||
      (m_fg > 0 && (!fg_drop || !n_fg_drop || !fg_members)) || (m_bg > 0 && (!bg_drop || !n_bg_drop || !bg_members)))
    return AIT_EINVAL;
  hipStream_t s = ait_stream(stream);
  if (m_fg > 0) {
    hipLaunchKernelGGL(anchor_disable_kernel, dim3((unsigned)((m_fg + 255) / 256), (unsigned)b), dim3(256), 0, s, fg_drop,
                       n_fg_drop, m_fg, fg_members, n_in, labels);
    AIT_CHECK_LAUNCH();
  }
  if (m_bg > 0) {
    hipLaunchKernelGGL(anchor_disable_kernel, dim3((unsigned)((m_bg + 255) / 256), (unsigned)b), dim3(256), 0, s, bg_drop,
                       n_bg_drop, m_bg, bg_members, n_in, labels);
    AIT_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(anchor_targets_kernel, dim3((unsigned)((H * W * A + 255) / 256), (unsigned)b), dim3(256), 0, s,
                     anchors_inside, inside_pos, n_in, A, H, W, gt, G, gt_cols, argmax, labels, inside_weight,
                     outside_weight, labels_out, targets, inside_w, outside_w);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}
