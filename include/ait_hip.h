/* include/ait_hip.h -- C ABI of libait_hip.so, the MI355X (gfx950) implementation of AIT's
 * proposal-query matching hot path.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers + sizes; every pointer is DEVICE memory unless it says host.
 *   - the caller owns every buffer including workspaces and scheduler scratch (ait_launch_ctx); no entry point
 *     allocates or frees device memory (ait_probe_create allocates a HOST object and its HIP events).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call only
 *     enqueues work on that stream and returns -- no host synchronisation.
 *   - returns AIT_OK (0) or a negative AIT_E* code; no C++ exception crosses the boundary.
 *   - re-entrant; no global mutable state (what the library memoises -- the occupancy of a kernel on the
 *     device model -- is a constant); safe to call concurrently from several threads and on different devices
 *     (the kernels run on the device that is current for the calling thread).
 *   - tensors are fp32, contiguous, row-major / NCHW unless stated.
 *
 * Each function names the reference interface it replaces (paths relative to the reference
 * repository root).
 */
#ifndef AIT_HIP_H_
#define AIT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIT_OK 0
#define AIT_EINVAL (-1)   /* bad argument (NULL pointer, non-positive size, unsupported shape) */
#define AIT_EWORKSPACE (-2) /* workspace too small */
#define AIT_ELAUNCH (-3)  /* hipLaunchKernel / hipMemsetAsync reported an error */
#define AIT_EUNSUPPORTED (-4)

/* ABI version; bumped on any signature change. */
int ait_abi_version(void);
/* 0 for the shipped library; 1 for a lab variant compiled with experiment knobs (csrc/lab_knobs.h) -- never shipped */
int ait_lab_build(void);
const char* ait_strerror(int code);

/* ---------------------------------------------------------------------------------------
 * Launch context: what a caller MAY hand to the entry points that launch the matrix-core GEMM (ait_gemm_*,
 * ait_conv_*, ait_mha_block_*, ait_ffn_*, ait_transformer_*) and to the measured RoIAlign pair.  NULL, or a
 * zeroed struct, is always legal.  Replaces nothing in the reference: its ATen operators take scratch from
 * ATen's caching allocator (lib/model/csrc/cuda/ROIAlign_cuda.cu:273-299 -- outputs and temporaries are the
 * caller's tensors); here the scratch is an explicit caller-owned buffer.
 *
 *   sched_ws / sched_ws_bytes   scheduler scratch of the persistent GEMM kernel, ait_gemm_workspace_bytes()
 *       bytes of device memory, prepared ONCE with ait_gemm_workspace_init() and then reused by any number of
 *       launches that are ORDERED on one stream (one workspace per stream that runs GEMMs concurrently).  With it
 *       the kernel hands its output tiles out dynamically (per-XCD ticket counters: a workgroup that becomes
 *       resident late -- beside an RCCL kernel -- finds what is left instead of a fixed share) and, when a
 *       product's tile count leaves the last round of the persistent grid badly filled (layer4- and
 *       co-attention-sized products), cuts the slabs of those tiles evenly over all workgroups ("stream-K"): a
 *       workgroup that computes part of a tile it does not own publishes a partial tile (write-through stores +
 *       an agent-scope flag), the owner adds the partials in a fixed order and runs the epilogue -- results do
 *       not depend on timing.  Without it: static work lists, whole tiles; same mathematics, the summation
 *       order of the affected products differs (both are deterministic).  A workspace that is too small for the
 *       launch returns AIT_EWORKSPACE.
 *   probe   measurement probe (below) or NULL.
 *   flags   0, or AIT_CTX_NATIVE_F32: the dense fp32 products (ait_gemm_f32 and the products inside the
 *       ait_mha_ / ait_ffn_ / ait_transformer_ composites) are formed by v_mfma_f32_32x32x2_f32, the instruction
 *       that multiplies f32 operands.  Default (0): every f32 operand is split EXACTLY into three bf16 values
 *       h = bf16(x) rounded to nearest even, m = the top 8 bits of the exact remainder x - h, l = x - h - m
 *       (|m| <= 2^-9 |x|, |l| < 2^-16 |x|, both zero-mean whatever the sign of x) and a product is accumulated in f32 from the six partial products that
 *       are >= 2^-18 of it, on v_mfma_f32_32x32x16_bf16 -- 16x the FLOP per cycle of the f32 instruction, six
 *       instead of one; the three dropped partial products sum to <= 2^-23 |a b|.  Tested against float64 beside the
 *       f32 instruction (tests/test_gpu_gemm.py): on random operands the same error; on SAME-SIGNED operands over
 *       reductions of 512 ... 76800 terms (the adversarial case: profiles/r04_split_bias.txt) mean error <= 1e-7 of
 *       the sum, scatter below the f32 instruction's.  (Round 3 formed the planes by truncation: one-signed planes,
 *       chopped by the pipe's accumulator alignment on long same-signed sums -- up to 1.4e-5 of the sum low.)
 *       Same operand images, same epilogues, same summation order over k-blocks.  Non-finite inputs: NaN where
 *       the f32 instruction gives an infinity; finite operands whose magnitude exceeds the largest bf16 value
 *       (3.3895e38 < |x| <= 3.4028e38: the nearest-rounded high plane is an infinity) likewise -- every other finite,
 *       overflow-free product stays finite (tested at +-3e38 and at the bf16 maximum).  The convolution composites (ait_conv_*, ait_tail_*) always use
 *       the split form.
 *       AIT_CTX_BF16: the same dense products with every operand value ROUNDED to bf16 (nearest even) in registers
 *       and ONE v_mfma_f32_32x32x16_bf16 per block, f32 accumulate, f32 operands and results in memory -- what
 *       torch.autocast(bfloat16) computes for a linear layer with f32 parameters (BASELINE configs[4]); NOT f32
 *       accuracy (8 significant bits per operand).  Honoured by the products, the composites and the convolutions
 *       (ait_conv_*, ait_tail_*).  Inside ait_transformer_fwd_train / ait_transformer_bwd_part the feed-forward blocks
 *       additionally STORE their 2048-wide tensors (the ReLU output h, its gradient) in bf16 and multiply bf16 operands
 *       from memory (ait_gemm_bf16s, ait_gemm_bf16s_tn; shapes permitting): the `saved` buffer of such a forward holds
 *       bf16 where an f32 forward holds f32, so THE BACKWARD MUST BE GIVEN THE FLAGS OF ITS FORWARD.  The same holds
 *       for q / k / v of the attention blocks (bf16 projection results, widened by the attention kernels).
 *       AIT_CTX_IO_BF16 (with AIT_CTX_BF16; ait_transformer_fwd / _fwd_train / _bwd / _bwd_part only): the operator's
 *       `out` is WRITTEN as a bf16 tensor [bp*64, 1024] and `d_out` is READ as one, behind the float pointers of the
 *       signatures -- for a consumer that computes in bf16 anyway (the proposal tail on bf16 convolutions in
 *       BASELINE configs[4]): no f32 copy of the largest activation of the path, dec_trans and its two gradient
 *       products on bf16 operands from memory.  Sizes permitting: ait_transformer_io_bf16_ok(bp, bs, n_src) != 0,
 *       else AIT_EUNSUPPORTED.
 * ------------------------------------------------------------------------------------- */
#define AIT_CTX_NATIVE_F32 1u
#define AIT_CTX_BF16 2u
#define AIT_CTX_IO_BF16 4u
typedef struct {
  void* sched_ws;
  size_t sched_ws_bytes;
  void* probe;
  unsigned flags;
} ait_launch_ctx;
size_t ait_gemm_workspace_bytes(void);   /* for the device current to the calling thread; 0 on error */
/* zeroes the control words of a fresh workspace (enqueued on `stream`, capturable); call once after allocation */
int ait_gemm_workspace_init(void* sched_ws, size_t sched_ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Measurement (bench.py's live roofline; no reference counterpart).  An entry point that is handed a probe
 * through ait_launch_ctx::probe brackets its GEMM / RoIAlign launches -- however they are reached, directly or
 * from inside ait_transformer_* -- with a HIP event pair on the launch stream and notes the launch's
 * algorithmic work: kind AIT_PROBE_GEMM: flops = 2*M*N*K, dims = {M, N, K, trans_a, trans_b, splits};
 * AIT_PROBE_ROI_FWD / _BWD: algorithmic bytes (feature once + RoIs + pooled tensor once), dims =
 * {n_rois, B, C, H, W, 0}.  The probe is a caller-owned host object (created on, and only recording launches
 * of, the device current at ait_probe_create; at most `capacity` launches are recorded, later ones are counted
 * only); ait_probe_get reports a launch's elapsed milliseconds once the stream has been synchronised.  The
 * library keeps no reference to a probe between calls; the caller synchronises the device before resetting,
 * reading or destroying a probe that launches in flight were given.  Slots are claimed atomically: one probe
 * may be handed to launches from several host threads (PyTorch runs backward passes on its autograd threads).
 * ------------------------------------------------------------------------------------- */
#define AIT_PROBE_GEMM 1
#define AIT_PROBE_ROI_FWD 2
#define AIT_PROBE_ROI_BWD 3
void* ait_probe_create(int capacity);
void ait_probe_destroy(void* probe);
int ait_probe_reset(void* probe);
int ait_probe_count(void* probe);      /* launches seen since the reset; > ait_probe_capacity: overflowed */
int ait_probe_capacity(void* probe);
int ait_probe_get(void* probe, int i, int* kind, double* work, float* ms, int* dims6);

/* ---------------------------------------------------------------------------------------
 * RoIAlign.  Replaces model._C.roi_align_forward / roi_align_backward
 *   (lib/model/csrc/vision.cpp:9-10, lib/model/csrc/ROIAlign.h:11-45; kernels
 *    lib/model/csrc/cuda/ROIAlign_cuda.cu:65-122,178-254; CPU semantics
 *    lib/model/csrc/cpu/ROIAlign_cpu.cpp:113-219).
 *   feat      [B,C,H,W]
 *   rois      [n_rois,5] = (batch_index, x1, y1, x2, y2) in image pixels
 *   out       [n_rois,C,PH,PW]
 *   grad_in   [B,C,H,W], zero-filled by the call before the scatter (the reference
 *             allocates a zero tensor, ROIAlign_cuda.cu:316)
 * sampling_ratio <= 0 selects the adaptive grid ceil(roi/P) (ROIAlign_cpu.cpp:161-165).
 * RoIs whose batch index is outside [0,B) produce zeros / contribute nothing.
 * ------------------------------------------------------------------------------------- */
int ait_roi_align_fwd(const float* feat, const float* rois, int n_rois, int B, int C, int H,
                      int W, int PH, int PW, float spatial_scale, int sampling_ratio,
                      float* out, void* stream);
int ait_roi_align_bwd(const float* grad_out, const float* rois, int n_rois, int B, int C,
                      int H, int W, int PH, int PW, float spatial_scale, int sampling_ratio,
                      float* grad_in, void* stream);

/* The same operator on CHANNELS-LAST features with a token-major result, for callers that keep
 * those layouts (ait_amd.faster_rcnn: the co-attention emits token rows, the AIT embedding reads
 * token rows -- the NCHW<->token transposes of Models.py:252-256 disappear):
 *   feat [B,H,W,C], out / grad_out [n_rois, PH*PW, C], grad_in [B,H,W,C];  PW == 7, PH <= 7,
 *   C % 4 == 0 and C <= 4096 (else AIT_EUNSUPPORTED: use the NCHW entry points).
 * workspace: ait_roi_align_nhwc_workspace_bytes(...) bytes, caller-owned, scratch (per-RoI
 * per-axis interpolation matrices; rebuilt by every call, nothing is kept between calls).
 * The result equals the NCHW operator's up to fp32 summation order (separable weights instead of
 * per-sample sums); the backward gathers per feature cell in RoI order: no atomics, no zero-fill,
 * bitwise reproducible.  ctx: only its probe is used (may be NULL). */
size_t ait_roi_align_nhwc_workspace_bytes(int n_rois, int H, int W, int PH, int PW);
int ait_roi_align_nhwc_fwd(const float* feat, const float* rois, int n_rois, int B, int C, int H,
                           int W, int PH, int PW, float spatial_scale, int sampling_ratio,
                           void* workspace, size_t workspace_bytes, float* out, const ait_launch_ctx* ctx,
                           void* stream);
int ait_roi_align_nhwc_bwd(const float* grad_out, const float* rois, int n_rois, int B, int C,
                           int H, int W, int PH, int PW, float spatial_scale, int sampling_ratio,
                           void* workspace, size_t workspace_bytes, float* grad_in, const ait_launch_ctx* ctx,
                           void* stream);

/* ---------------------------------------------------------------------------------------
 * NMS.  Replaces model._C.nms (lib/model/csrc/vision.cpp:8, lib/model/csrc/nms.h:10-28)
 * with the CPU reference's semantics (lib/model/csrc/cpu/nms_cpu.cpp:5-65): +1 pixel
 * convention, a box is suppressed when IoU >= thr, survivors are reported as ORIGINAL
 * indices in ascending order.
 *   boxes     [n,4] (x1,y1,x2,y2)
 *   order     [n] int64 permutation sorting the boxes by descending score, or NULL when the
 *             boxes are already sorted (the only way rpn/proposal_layer.py:153 calls it)
 *   keep      [n] int64, first *n_keep entries valid
 *   n_keep    [1] int32 (device)
 *   max_keep  > 0: stop after that many survivors IN SCORE ORDER have been found
 *             (proposal_layer.py:156-157 keeps only the first post_nms_topN); only legal
 *             with order == NULL, where score order == index order.  <= 0: no limit.
 * Everything (IoU bitmask, greedy scan, compaction) runs on the device; unlike
 * lib/model/csrc/cuda/nms.cu:99-123 there is no device->host copy and no host scan.
 * ------------------------------------------------------------------------------------- */
size_t ait_nms_workspace_bytes(int n);
int ait_nms(const float* boxes, const int64_t* order, int n, float thr, int max_keep,
            void* workspace, size_t workspace_bytes, int64_t* keep, int32_t* n_keep,
            void* stream);
/* Batched form for the proposal layer (rpn/proposal_layer.py:134-164 loops over the images of a
 * batch): `batch` independent problems of n pre-sorted boxes each, boxes [batch,n,4]; image b's
 * survivors go to keep + b*keep_stride, its count to n_keep[b].  One launch pair for the whole
 * batch: the greedy scans of the images run concurrently on different CUs. */
size_t ait_nms_batched_workspace_bytes(int batch, int n);
int ait_nms_batched(const float* boxes, int batch, int n, float thr, int max_keep,
                    void* workspace, size_t workspace_bytes, int64_t* keep,
                    long long keep_stride, int32_t* n_keep, void* stream);

/* ---------------------------------------------------------------------------------------
 * Box arithmetic of the proposal layer and the proposal-target layer, one launch per stage (SURVEY 8 rows
 * a12 / a13 / f2).  Each entry replaces a chain of 15-40 elementwise / gather tensor expressions of the
 * reference; all are written in the reference's fp32 operation order (built with -ffp-contract=off), so
 * clipping, IoU thresholds and class membership see the same values.
 *
 * ait_rpn_decode         lib/model/rpn/proposal_layer.py:66-117 + bbox_transform.py:74-117
 *   (bbox_transform_inv, clip_boxes): probs [b,2A,H,W] (softmax output; channel A+a = fg score of anchor
 *   a), deltas [b,4A,H,W] (channel 4a+c), anchors [H*W*A,4] in (h, w, a) order, im_info [b,3] = (height,
 *   width, scale)  ->  boxes [b,H*W*A,4] clipped to the image, scores [b,H*W*A].
 * ait_proposals_assemble proposal_layer.py:150-160: survivors keep[i][0 .. n_keep[i]) of the sorted candidates
 *   cand [b,n,4]  ->  out [b,post_n,5] = (image index, box), zero boxes after the last survivor.
 * ait_roi_classify       proposal_target_layer_cascade.py:49-52,128-150 + bbox_transform.py:167-211:
 *   rois [b,R0,5] and gt [b,G,gt_cols>=5] (x1,y1,x2,y2,class)  ->  all_rois [b,R0+G,5] (the gt boxes appended
 *   as RoIs), assign [b,R] (best gt, first maximum; zero-area gt -> IoU 0, zero-area RoI -> -1), labels [b,R],
 *   counts [b,2] = (#fg: IoU >= fg_thresh, #bg: bg_lo <= IoU < bg_hi) -- the only numbers the reference's
 *   host-side RNG calls need -- and fg_members / bg_members [b,R]: the class members in ascending RoI order,
 *   followed by the non-members (what a stable sort of the class mask yields).
 * ait_roi_sample_gather  proposal_target_layer_cascade.py:86-126,160-213: sample s of image i is member
 *   pos[i][s] of the fg class if s < n_fg[i], else of the bg class  ->  rois_b [b,P,5], labels_b [b,P] (0 for
 *   bg), bbox_targets [b,P,4] ((t - mean) / std when normalize), inside / outside weights [b,P,4]; an image
 *   whose sampled labels are all 0 gets no regression targets.  means / stds / inside_weights: HOST arrays of 4.
 * ------------------------------------------------------------------------------------- */
int ait_rpn_decode(const float* probs, const float* deltas, const float* anchors, const float* im_info, int b,
                   int A, int H, int W, float* boxes, float* scores, void* stream);
int ait_proposals_assemble(const float* cand, int n, const int64_t* keep, long long keep_stride, int keep_cols,
                           const int32_t* n_keep, int b, int post_n, float* out, void* stream);
/* Anchor-target layer (lib/model/rpn/anchor_target_layer.py:55-187) around the reference's host-side RNG draws:
 * ait_anchor_classify  anchors_inside [n_in,4] (the anchors inside the image, ascending grid order) x gt
 *   [b,G,gt_cols>=4] -> max_ov / argmax [b,n_in], labels [b,n_in] (-1 / 0 / 1 by the negative / positive overlap
 *   thresholds and the best anchor of every gt box; gt_max_bits [b,G] is scratch), counts [b,2] = (#label 1,
 *   #label 0) -- what the host's permutation draws need -- and the two class member lists (ascending, members only).
 * ait_anchor_targets   disables the drawn members (labels -> -1; fg_drop / bg_drop [b,m] positions within the
 *   class lists, n_*_drop [b] how many are valid), then writes the layer's four outputs on the full [H,W,A] grid in
 *   the layouts the RPN losses read: labels_out [b,1,A*H,W], targets / inside_w / outside_w [b,4A,H,W]
 *   (inside_pos [H*W*A]: index of each grid anchor within anchors_inside, -1 outside the image). */
int ait_anchor_classify(const float* anchors_inside, int n_in, const float* gt, int b, int G, int gt_cols,
                        float negative_overlap, float positive_overlap, int clobber_positives, float* max_ov,
                        int64_t* argmax, int32_t* gt_max_bits, float* labels, int64_t* counts, int32_t* fg_members,
                        int32_t* bg_members, void* stream);
int ait_anchor_targets(const float* anchors_inside, const int32_t* inside_pos, int n_in, int A, int H, int W,
                       const float* gt, int b, int G, int gt_cols, const int64_t* argmax, float* labels,
                       const int64_t* fg_drop, const int32_t* n_fg_drop, int m_fg, const int32_t* fg_members,
                       const int64_t* bg_drop, const int32_t* n_bg_drop, int m_bg, const int32_t* bg_members,
                       float inside_weight, float outside_weight, float* labels_out, float* targets,
                       float* inside_w, float* outside_w, void* stream);
size_t ait_roi_classify_workspace_bytes(int b, int R0, int G);
int ait_roi_classify(const float* rois, int b, int R0, const float* gt, int G, int gt_cols, float fg_thresh,
                     float bg_thresh_hi, float bg_thresh_lo, void* workspace, size_t workspace_bytes,
                     float* all_rois, int64_t* assign, float* labels, int64_t* counts, int64_t* fg_members,
                     int64_t* bg_members, void* stream);
int ait_roi_sample_gather(const int64_t* pos, const int64_t* n_fg, int b, int P, int R, const int64_t* fg_members,
                          const int64_t* bg_members, const float* labels, const float* all_rois,
                          const int64_t* assign, const float* gt, int G, int gt_cols, const float* means,
                          const float* stds, const float* inside_weights, int normalize, float* rois_b,
                          float* labels_b, float* bbox_targets, float* inside_w, float* outside_w, void* stream);

/* ---------------------------------------------------------------------------------------
 * fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32
 * accumulate).  Replaces the ATen matmul / addmm / 1x1-conv calls behind
 *   nn.Linear w_qs/w_ks/w_vs/fc      lib/model/system/SubLayers.py:51-58,77-79,97
 *   nn.Linear w_1/w_2                lib/model/system/SubLayers.py:172-173,181
 *   conv2d_1x1 enc_emb/dec_emb/dec_trans  lib/model/system/Models.py:188-193,207-209,246-247,278
 * and their autograd backward products.
 *
 *   C (op)= alpha * opA(A) . opB(B) [+ bias] [+ residual], optional ReLU
 *   trans_a == 0: A is [M,K] row-major (lda >= K);  trans_a != 0: A is [K,M] (lda >= M)
 *   trans_b == 0: B is [K,N] row-major (ldb >= N);  trans_b != 0: B is [N,K] (ldb >= K)
 *                 (trans_b != 0 is the nn.Linear layout: y = x W^T)
 *   bias     [N], or [M] with AIT_GEMM_BIAS_ROW; may be NULL
 *   residual same addressing as C; may be NULL
 *   c_colblk == 0: C(i,j) at C[i*ldc + j].
 *   c_colblk  > 0: C(i,j) at C[(j / c_colblk) * c_batch_stride + i*ldc + (j % c_colblk)]
 *                  (writes a [channel, token] product straight into an NCHW tensor:
 *                   dec_trans, Models.py:276-278)
 *   flags    AIT_GEMM_RELU | AIT_GEMM_ACCUMULATE (C += ...) | AIT_GEMM_ATOMIC | AIT_GEMM_BIAS_ROW
 *            | AIT_GEMM_MASK_POS
 *   split_k  > 1 splits the reduction over gridDim.z; requires AIT_GEMM_ATOMIC (partial tiles
 *            are combined with fp32 atomics into C, which the caller has zeroed or wants
 *            accumulated into) and no bias / residual / ReLU.
 * Requirements: lda, ldb multiples of 4; A, B 16-byte aligned; K a multiple of 4 unless both
 * operands have the reduction dimension outermost (trans_a != 0 and trans_b == 0)
 * (else AIT_EUNSUPPORTED).
 * ------------------------------------------------------------------------------------- */
#define AIT_GEMM_RELU 1
#define AIT_GEMM_ACCUMULATE 2
#define AIT_GEMM_ATOMIC 4
#define AIT_GEMM_BIAS_ROW 8
#define AIT_GEMM_MASK_POS 16 /* C = (residual > 0) ? value : 0  (ReLU backward; `residual` holds
                                the saved forward activation and is NOT added) */
#define AIT_GEMM_COLSUM 32   /* `bias` is an OUTPUT: float[N] into which the column sums of the stored C are
                                ADDED (fp32 atomics, one per column per tile) -- the bias gradient of the layer
                                whose input gradient this product is (SubLayers.py:181: d b1 = sum over tokens of
                                dh), without a second pass over C.  No bias is added; not with AIT_GEMM_ATOMIC /
                                AIT_GEMM_BIAS_ROW / AIT_GEMM_ACCUMULATE / c_colblk. */
int ait_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A,
                 int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                 const float* residual, int flags, int split_k, int c_colblk,
                 long long c_batch_stride, const ait_launch_ctx* ctx, void* stream);

/* ---------------------------------------------------------------------------------------
 * Products with a PRE-SPLIT weight operand ("P3").  No reference counterpart: how this library keeps the weight
 * side of  y = x W^T  (lib/model/system/SubLayers.py:77-79,97,181-182; Models.py:246-247,278) and of its input
 * gradient  dx = dy W  off the vector pipe.  P3 of an f32 matrix X [rows][K] (K = the reduction dimension of
 * the product it enters) = its three bf16 planes h = bf16(x) (nearest even), m = top 8 bits of x - h, l = x - h - m
 * (x = h + m + l exactly), interleaved in groups of eight along K:
 *     P3[row][K / 8][h, m, l][8] bf16   -- 6 bytes per value, ait_p3_bytes(rows, K) in all, 16-byte aligned.
 * ait_p3_split: src [rows][cols] f32 (row pitch ld floats, ld % 4 == 0, 16-byte aligned) -> P3 of src (transpose
 *   == 0: reduction dimension = cols, cols % 8 == 0) or of its transpose (transpose != 0: P3 [cols][rows / 8][3][8],
 *   rows % 8 == 0).  A weight is converted once per optimizer step; the ait_transformer_* composites convert their
 *   own weights at the start of every call (33 MB, one launch).
 * ait_gemm_f32_p3:  C [M,N] = alpha * A [M,K] (f32, row pitch lda) . X^T, X [N][K] given as B_p3 = P3 of X with
 *   row pitch ldb_values VALUES (>= K, % 8 == 0; a K sub-range of wider rows is addressed by offsetting B_p3 by
 *   6 bytes per value).  bias / residual / flags / c_colblk as ait_gemm_f32 (no AIT_GEMM_ATOMIC / ACCUMULATE /
 *   BIAS_ROW); K % 16 == 0.  The same six partial products per f32 product as ait_gemm_f32's default form: A is
 *   split in registers, B arrives split, both to nearest; dropped terms <= 2^-23 |a b|.
 * ------------------------------------------------------------------------------------- */
size_t ait_p3_bytes(long long rows, long long cols);
int ait_p3_split(const float* src, int rows, int cols, int ld, int transpose, void* dst, void* stream);
int ait_gemm_f32_p3(int M, int N, int K, float alpha, const float* A, int lda, const void* B_p3, long long ldb_values,
                    float* C, int ldc, const float* bias, const float* residual, int flags, int c_colblk,
                    long long c_batch_stride, const ait_launch_ctx* ctx, void* stream);

/* Batched form: batch x batch2 independent products, operand (i, j) at base + i*stride + j*stride2 (floats),
 * one launch.  Replaces the three torch.matmul of the COCO variant's image-level co-attention
 *   (lib/model/modules/blocks_coatt_transformer_sk.py:86-110), the per-(image, head) score and P.V products of
 * the VOC variant's (lib/model/system/Modules.py:18,27 at len_q or len_k = H_i*W_i, second batch level = heads),
 * and their autograd backward.
 * flags: 0, AIT_GEMM_ACCUMULATE, or AIT_GEMM_ATOMIC with split_k > 1 (long reductions of few small products:
 * partial sums ADDED to C with fp32 atomics; the caller zero-fills C).  K % 4 != 0 is accepted for trans_a == 0 && trans_b == 0 when
 * lda >= K rounded up to 4 and A's row padding holds finite values (the 2394-token image side). */
int ait_gemm_f32_batched(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, int lda,
                         long long stride_a, long long stride_a2, const float* B, int ldb, long long stride_b,
                         long long stride_b2, float* C, int ldc, long long stride_c, long long stride_c2, int batch,
                         int batch2, int flags, int split_k, const ait_launch_ctx* ctx, void* stream);

/* Row softmax + dropout over [rows, cols] matrices with row pitch ld (Modules.py:24 for score matrices that do
 * not fit attn.hip's 64x64 tile): y = softmax(x) per row, y_drop = dropout(y) (stateless hash of (seed, r*cols + c);
 * y_drop may alias y when p_drop == 0).  Backward: dx = y * (dP - sum_c dP*y), dP = dy_drop * mask / (1-p). */
int ait_softmax_rows_fwd(const float* x, long long rows, int cols, long long ld, float p_drop, unsigned long long seed,
                         float* y, float* y_drop, void* stream);
int ait_softmax_rows_bwd(const float* dy_drop, const float* y, long long rows, int cols, long long ld, float p_drop,
                         unsigned long long seed, float* dx, void* stream);

/* Selective heads (ait_sh_fwd / ait_sh_bwd) for ANY sequence length T (H = 8, dv = 64): the co-attention's
 * 2394-token side.  workspace (backward): n_seq * (H*dv + dv) floats of scratch. */
int ait_sh_general_fwd(const float* O, const float* sk_w, const float* sk_b, int n_seq, int H, int T, int dv, float* u,
                       float* gate, float* s, void* stream);
int ait_sh_general_bwd(const float* du, const float* O, const float* gate, const float* sk_w, int n_seq, int H, int T,
                       int dv, float* dO, float* dg, float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------
 * Convolutions over CHANNELS-LAST maps as implicit GEMMs on the same matrix-core kernel (no im2col buffer).
 * Replace the ATen / cuDNN convolution calls (forward, data gradient, weight gradient) behind the 3x3
 * convolutions of RCNN_top = ResNet layer4 applied to the AIT output
 *   (lib/model/faster_rcnn/resnet_sys_transformer_sk_dilat.py:85-111 Bottleneck.conv2, :482-491 _head_to_tail).
 *   x   [n*in_h*in_w, cin]   row (image, y, x) of a channels-last map, row pitch ldx
 *   y   [n*out_h*out_w, cout], dy likewise; dx like x
 *   w   [cout][kh][kw][cin]  (= the channels-last memory of a PyTorch conv weight [cout, cin, kh, kw]); dw likewise
 *   y  = conv(x, w) (+ bias[cout]) (+ residual, same addressing as y) (ReLU with AIT_GEMM_RELU)
 *   dx = conv_transpose(dy, w) (+ residual, or gated by residual > 0 with AIT_GEMM_MASK_POS)
 *   dw += dy^T (*) x     ACCUMULATED (split-K partial sums combined with fp32 atomics; caller zero-fills)
 *   zeros: caller-owned device buffer of >= max(cin, cout) + 144 floats that holds zeros (what a window
 *          position outside the map reads).
 * Requirements (else AIT_EUNSUPPORTED): stride a power of two; cin % 16 == 0 (forward), cout % 16 == 0 (data
 * gradient), cin % 128 == 0 (weight gradient).  The map the GEMM rows run over (out_h x out_w for forward / weight
 * gradient, in_h x in_w for the data gradient) may have any size: power-of-two width and area decompose a row by
 * shifts, anything else by exactly corrected f32 quotients (dense convolutions, fewer than 2^24 rows; a weight
 * gradient over rows that are not a multiple of 16 runs its reduction to the next multiple, the gathered operand
 * reading zeros there).  Grouped convolutions and the per-parity stride-2 data gradient need the power-of-two form.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  int n, in_h, in_w, out_h, out_w, kh, kw, stride, pad;
  int groups; /* 0 or 1: dense.  > 1: grouped convolution (the SK block's branches,
                 blocks_sys_transformer_sk_dilat.py:938-947): weights [cout][kh][kw][cin/groups], cin/groups and
                 cout/groups multiples of 128 (a 128-wide tile of the implicit GEMM lies inside one group) */
} ait_conv_geom;
int ait_conv_fwd_f32(const float* x, int ldx, const float* w, const ait_conv_geom* geom, int cin, int cout,
                     const float* bias, const float* residual, int flags, float* y, int ldy, const float* zeros,
                     size_t zeros_floats, const ait_launch_ctx* ctx, void* stream);
int ait_conv_bwd_data_f32(const float* dy, int lddy, const float* w, const ait_conv_geom* geom, int cin, int cout,
                          const float* residual, int flags, float* dx, int lddx, const float* zeros,
                          size_t zeros_floats, const ait_launch_ctx* ctx, void* stream);
int ait_conv_bwd_weight_f32(const float* dy, int lddy, const float* x, int ldx, const ait_conv_geom* geom, int cin,
                            int cout, float* dw, int split_k, const float* zeros, size_t zeros_floats,
                            const ait_launch_ctx* ctx, void* stream);

/* bf16 matrix-core variant (BASELINE cfg 5): identical interface, layouts and epilogues; A and B
 * stay fp32 in memory, are rounded to bf16 (RNE) while being staged into LDS, multiplied on
 * v_mfma_f32_32x32x16_bf16 and accumulated in fp32.  Not used for the fp32 headline metric. */
int ait_gemm_bf16(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A,
                  int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                  const float* residual, int flags, int split_k, int c_colblk,
                  long long c_batch_stride, const ait_launch_ctx* ctx, void* stream);

/* bf16 STORAGE (BASELINE configs[4]; /root/reference/cfgs/res101.yml:1-18 names the configuration, the reference has no
 * bf16 path of its own): C[M, N] = A[M, K] . B[N, K]^T with A and B held in memory as bf16 (K contiguous, row pitches lda /
 * ldb in ELEMENTS, multiples of 8, bases 16-byte aligned), one v_mfma_f32_32x32x16_bf16 per block, f32 accumulate.
 * K % 32 == 0, N % 128 == 0, any M.  Epilogue: + bias[N] (f32, may be NULL); then EITHER + residual[M, N] (f32, pitch ldr)
 * OR with AIT_GEMM_MASK_POS the value is kept where the gate is > 0 and zeroed elsewhere -- the gate is `gate16` (a bf16
 * [M, N] tensor, pitch ldr: the STORED ReLU output of the forward) if given, else `residual`; then ReLU with AIT_GEMM_RELU.
 * The result is written as f32 into C32 (pitch ldc32) and / or rounded to bf16 (nearest even) into C16 (pitch ldc16):
 * either may be NULL, not both.  The linears of the AIT in the bf16-storage mode: Models.py:246-247,278,
 * SubLayers.py:77-79,97,181-183 and their input gradients (B = the transposed weight copy).
 * With ctx->sched_ws (the scheduler scratch of ait_launch_ctx; launches ordered on one stream) a reduction of K >= 4096 whose
 * 256 x 256 tiles leave a last round of at most half the chip's workgroups has those leftover tiles cut along K into 8 / 4 / 2
 * pieces, summed in piece order by a second small launch (ABI v8; also ait_conv_fwd_bf16s): the same products, another --
 * fixed, reproducible -- summation order for those tiles than without the scratch. */
int ait_gemm_bf16s(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, float* C32,
                   long long ldc32, void* C16, long long ldc16, const float* bias, const float* residual,
                   const void* gate16, long long ldr, int flags, const ait_launch_ctx* ctx, void* stream);
/* ... and the weight-gradient product of that mode: C[Mo, No] (f32, pitch ldc) += sum over the R token rows of
 * A[r, m] * B[r, n], A = bf16 [R, Mo] (the output gradient), B = bf16 [R, No] (the layer's input), both row-major with
 * the REDUCTION index outermost (pitches in elements, multiples of 8).  Mo % 256 == 0, No % 128 == 0 or (ABI v8) No == 64 -- d fc_w =
 * df^T u of the attention blocks, SubLayers.py:97, on a 256 x 64 tile --; the rows are whole 32-row
 * slabs (R % 32 == 0: pad the operands with zero rows), cut into split_k ranges of slabs that differ by at most one slab (ABI v8;
 * before: equal ranges only) -- at most one range per slab -- whose partial tiles are ADDED to C: zero C for a plain gradient
 * (SubLayers.py:181-183's weights).  `partials`
 * (optional, caller-owned, split_k * Mo * No floats, 16-byte aligned): the ranges' tiles are stored there once and added to
 * C in range order by a second small launch (bit-reproducible); NULL or too small: f32 atomics (order-dependent rounding). */
int ait_gemm_bf16s_tn(int Mo, int No, int R, const void* A, long long lda, const void* B, long long ldb, float* C,
                      long long ldc, int split_k, void* partials, size_t partials_bytes, const ait_launch_ctx* ctx,
                      void* stream);
/* out[c] += sum_r x[r * ld + c] over a bf16 matrix (cols, ld % 4 == 0): ait_colsum_f32 for a gradient stored in bf16 */
int ait_colsum_bf16(const void* x, long long rows, int cols, long long ld, float* out, void* stream);
/* f32 [rows, cols] (pitch ld_src) -> bf16, nearest even: dst[r, c] (pitch ld_dst >= cols; cols, pitches % 4 == 0), or with
 * transpose != 0 dst[c, r] (pitch ld_dst >= rows; any shape).  The per-step weight copies (and their transposes, the B operand of
 * the input-gradient products) and the activations whose producer is not one of this library's bf16-emitting kernels. */
int ait_f32_to_bf16(const float* src, long long rows, int cols, long long ld_src, void* dst, long long ld_dst,
                    int transpose, void* stream);

/* bf16-storage CONVOLUTIONS (ABI v8): the 3x3 convolutions of RCNN_top = ResNet layer4 behind the AIT in the bf16 configuration
 * (BASELINE configs[4]; lib/model/faster_rcnn/resnet_sys_transformer_sk_dilat.py:85-111 Bottleneck.conv2, :482-491) as implicit
 * GEMMs of ait_gemm_bf16s / ait_gemm_bf16s_tn: no im2col buffer, the operand rows gathered through the window by the LDS-DMA
 * loads (a tap outside the map reads `zeros`: a caller-owned, 16-byte aligned device buffer of >= 512 zero bytes).
 * Geometry (else AIT_EUNSUPPORTED): stride 1, odd square window with pad = k / 2 (out = in), in_w and in_h * in_w powers of two,
 * groups <= 1, cin a power of two >= 64 (>= 128 for the weight gradient), cout % 128 == 0 (% 256 for the weight gradient).
 *   x   bf16 [n * in_h * in_w, cin] rows (map, y, x) of a channels-last map, pitch ldx (elements, % 8)
 *   w   bf16 [cout][kh][kw][cin]: ait_conv_weight_to_bf16's `w16` of the f32 parameter (channels-last memory of a PyTorch weight)
 *   y   = conv(x, w) (+ bias[cout] f32) (+ res16: bf16 [rows, cout], pitch ldr) (kept where gate16 > 0 with AIT_GEMM_MASK_POS:
 *         bf16 [rows, cout], pitch ldr) (ReLU with AIT_GEMM_RELU) -> y32 (f32, pitch ldy32) and / or y16 (bf16, pitch ldy16)
 *   The DATA GRADIENT is the same call on the output gradient with the roles of cin / cout swapped and `w16_dgrad` as the weight:
 *         dx = ait_conv_fwd_bf16s(dy, ..., w16_dgrad, geom, cout, cin, ...)  (w16_dgrad = [cin][taps, window mirrored][cout]).
 *   dw  f32 [cout][kh][kw][cin] += dy^T (*) x over split_k equal ranges of whole 32-row slabs of the rows (zero dw first);
 *         partials as in ait_gemm_bf16s_tn.
 * ait_conv_weight_to_bf16: w f32 [cout][taps][cin] (x row_scale[cout] if given: a frozen BatchNorm's scale folded in) ->
 *   w16 (as it lies) and / or w16_dgrad; either may be NULL.  cout, cin % 64 == 0. */
int ait_conv_fwd_bf16s(const void* x, long long ldx, const void* w, const ait_conv_geom* geom, int cin, int cout,
                       const float* bias, const void* res16, const void* gate16, long long ldr, int flags, float* y32,
                       long long ldy32, void* y16, long long ldy16, const void* zeros, size_t zeros_bytes,
                       const ait_launch_ctx* ctx, void* stream);
int ait_conv_bwd_weight_bf16s(const void* dy, long long lddy, const void* x, long long ldx, const ait_conv_geom* geom,
                              int cin, int cout, float* dw, int split_k, const void* zeros, size_t zeros_bytes,
                              void* partials, size_t partials_bytes, const ait_launch_ctx* ctx, void* stream);
int ait_conv_weight_to_bf16(const float* w, const float* row_scale, int cout, int taps, int cin, void* w16,
                            void* w16_dgrad, void* stream);

/* ---------------------------------------------------------------------------------------
 * Row kernels (d_model = 512 only; other widths return AIT_EUNSUPPORTED).
 *
 * ait_ln_fwd:   y[r] = LayerNorm_eps( dropout(a[src(r)] + pos[t]) + residual[r] ) * gamma + beta
 *   Replaces, in one pass over HBM:
 *     encoder/decoder prologue  dropout(position_enc(x)); layer_norm   lib/model/system/Models.py:98-99,155-156
 *       incl. the zero padding 49->64 (:269-270) and the repeat of the query over proposals (:250)
 *     "dropout(fc(q)); q += residual; layer_norm(q)"                  lib/model/system/SubLayers.py:97-100
 *     "dropout(w_2(...)); x += residual; layer_norm(x)"               lib/model/system/SubLayers.py:182-185
 *   Output row r = (q, t) with q = r / seq_len, t = r % seq_len.  Its source row in `a` is
 *   (q / rep) * src_rows_per_seq + t when t < src_rows_per_seq, else a zero row
 *   (identity map: seq_len = src_rows_per_seq, rep = 1).  pos [seq_len, d] and residual
 *   [rows, d] may be NULL.  mean / rstd [rows] are saved for the backward (may be NULL).
 *   Dropout: stateless hash of (seed, r*d + c); p_drop = 0 disables it.
 * ait_ln_bwd:   recomputes z from (a, pos, residual, seed); writes
 *   da   = dropout-mask * dz.  rep == 1: indexed by SOURCE row (padding rows dropped);
 *          rep  > 1: indexed by output row (the caller sums over the rep copies, ait_rep_sum_f32).
 *          May be NULL.
 *   dres = dz [rows, d], may be NULL.
 *   dgamma, dbeta [d]: ACCUMULATED with atomics (caller zeroes or carries a running sum); both
 *          NULL or both non-NULL.
 *   dcolsum [d]: ACCUMULATES the column sums of da over the rows that have a source = the gradient
 *          of the bias of the linear layer that produced `a` (replaces autograd's dy.sum(0)); may
 *          be NULL.
 *   dy_rows_per_seq: dy holds only the first dy_rows_per_seq rows of every sequence (row
 *          (q, t) at dy[q*dy_rows_per_seq + t]); the other rows received no gradient (the
 *          encoder output, of which only the n_src real rows are read again).  == seq_len: all.
 * ait_colsum_f32:  out[c] += sum_r x[r*ld + c]  (bias gradients; cols % 4 == 0).
 * ait_rep_sum_f32: out[g*E + e] += sum_{j < rep} x[(g*rep + j)*E + e]  (gradient of a block of E
 *          floats repeated rep times: the query sequence over the proposals, Models.py:250).
 * ------------------------------------------------------------------------------------- */
int ait_ln_fwd(const float* a, const float* pos, const float* residual, const float* gamma,
               const float* beta, long long rows, int d, int seq_len, int src_rows_per_seq,
               int rep, float eps, float p_drop, unsigned long long seed, float* y, float* mean,
               float* rstd, void* stream);
/* the same with a COMPACTED output: of every sequence only the first out_rows_per_seq rows are written, row (q, t)
 * at y[q * out_rows_per_seq + t] -- the encoder's self-attention block, of whose 64 rows only the n_src real ones
 * are read again (Models.py:269-270 pads, the masks of :258-260 ignore the padding ever after); statistics are
 * saved for every row. */
int ait_ln_fwd_rows(const float* a, const float* pos, const float* residual, const float* gamma,
                    const float* beta, long long rows, int d, int seq_len, int src_rows_per_seq,
                    int rep, int out_rows_per_seq, float eps, float p_drop, unsigned long long seed, float* y,
                    float* mean, float* rstd, void* stream);
int ait_ln_bwd(const float* dy, const float* a, const float* pos, const float* residual,
               const float* gamma, const float* mean, const float* rstd, long long rows, int d,
               int seq_len, int src_rows_per_seq, int rep, int dy_rows_per_seq, float p_drop,
               unsigned long long seed, float* da, float* dres, float* dgamma, float* dbeta,
               float* dcolsum, void* stream);
int ait_colsum_f32(const float* x, long long rows, int cols, long long ld, float* out, void* stream);
int ait_rep_sum_f32(const float* x, int groups, int rep, long long E, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Selective heads + head sum (H = 8, T = 64, dv = 64 only).
 * Replaces SHBlock.forward (lib/model/system/SubLayers.py:22-39) and q.sum(dim=1) (:92).
 *   O     [n_seq, H, T, dv]  per-head attention outputs
 *   sk_w  [H*dv, dv], sk_b [H*dv]   (state_dict: *.sh.sk.weight / *.sh.sk.bias)
 *   u     [n_seq, T, dv] = sum_h O_h * softmax_h(sk(mean_t sum_h O_h))
 *   gate  [n_seq, H*dv]  the head softmax (saved for backward), s [n_seq, dv] the pooled vector
 * Backward: dO [n_seq,H,T,dv]; dg [n_seq, H*dv] = gradient at the sk() output, from which the
 * caller forms d sk_w = dg^T s (one GEMM) and d sk_b = column sums.
 * ------------------------------------------------------------------------------------- */
int ait_sh_fwd(const float* O, const float* sk_w, const float* sk_b, int n_seq, int H, int T,
               int dv, float* u, float* gate, float* s, void* stream);
int ait_sh_bwd(const float* du, const float* O, const float* gate, const float* sk_w, int n_seq,
               int H, int T, int dv, float* dO, float* dg, void* stream);

/* ---------------------------------------------------------------------------------------
 * Scaled dot-product attention per (sequence, head) (T = 64, d = 64 only).
 * Replaces ScaledDotProductAttention.forward (lib/model/system/Modules.py:16-29) and the
 * head split / transpose around it (lib/model/system/SubLayers.py:77-90).
 *   q      row (n*T + t) of a [n_seq*T, ldq] matrix; head h occupies columns [h*d, (h+1)*d)
 *   k,v    row (n*kv_rows + t) of [n_seq*kv_rows, ld*] matrices, kv_rows <= T: a memory of fewer
 *          than T tokens is passed UNPADDED (the 49-token proposal memory of the decoder's
 *          cross-attention); keys >= kv_rows do not exist (probability exactly 0)
 *   mask_mode 0: none; 1: keys >= n_valid_keys masked (src_mask, Models.py:258-260);
 *             2: causal, key <= query (trg_mask, Models.py:262-263)
 *   scale  1/temperature (= 1/8)
 *   P      [n_seq, H, T, T] softmax probabilities BEFORE dropout (saved for backward; NULL to skip)
 *   O      [n_seq, H, T, d]  = dropout(P) V
 * Backward writes dq/dk/dv with the same row/column addressing as q/k/v.
 * ------------------------------------------------------------------------------------- */
int ait_attn_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                 int n_seq, int H, int T, int d, int kv_rows, int mask_mode, int n_valid_keys, float scale,
                 float p_drop, unsigned long long seed, float* P, float* O, void* stream);
int ait_attn_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                 const float* P, const float* dO, int n_seq, int H, int T, int d, int kv_rows, float scale,
                 float p_drop, unsigned long long seed, float* dq, int lddq, float* dk, int lddk,
                 float* dv, int lddv, void* stream);

/* ---------------------------------------------------------------------------------------
 * Everything of MultiHeadAttention.forward behind the Q / K / V projections as ONE kernel, one workgroup per
 * sequence with its eight heads resident (lib/model/system/SubLayers.py:82-100 with Modules.py:16-29 and SHBlock,
 * SubLayers.py:22-39): attention tiles, selective heads, the head sum, fc, dropout, residual, LayerNorm
 *     y = LayerNorm_eps( dropout_{p_fc}( (sum_h gate_h * (dropout_{p_attn}(softmax(mask(q k^T * scale))) v)_h) fc_w^T ) + residual )
 * = ait_attn_fwd -> ait_sh_fwd -> ait_gemm_f32 (fc) -> ait_ln_fwd_rows in one launch, with the same dropout masks for
 * the same seeds.  H = 8, T = 64, d = 64, model width 512 only.  q / k / v / kv_rows / mask_mode / n_valid_keys /
 * scale as ait_attn_fwd; sk_w [512,64], sk_b [512] (SHBlock.sk), fc_w [512,64] (MultiHeadAttention.fc, no bias),
 * residual [n_seq*64, 512] (the block's input), ln_g / ln_b [512]; y [n_seq*out_rows, 512]: rows t >= out_rows of a
 * sequence are not written (out_rows = 64: all).  q_rep >= 1: sequence n takes its queries AND its residual from
 * sequence n / q_rep of q / residual ([n_seq / q_rep * 64, .]) -- inference, where the decoder's query side is the same
 * for the q_rep proposals of a pair (Models.py:250 repeats it; without dropout the repeats are equal).
 * Saved for the backward, each optional (NULL: not written; inference passes NULL for all of them and nothing but y
 * leaves the chip): P [n_seq,8,64,64] probabilities before dropout, O [n_seq,8,64,64], u [n_seq*64,64] the gated head
 * sum, gate [n_seq,512], s [n_seq,64], f [n_seq*64,512] fc's output before dropout, mean / rstd [n_seq*64].
 * ------------------------------------------------------------------------------------- */
int ait_mha_core_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int n_seq,
                     int kv_rows, int mask_mode, int n_valid_keys, float scale, float p_attn,
                     unsigned long long seed_attn, const float* sk_w, const float* sk_b, const float* fc_w,
                     const float* residual, const float* ln_g, const float* ln_b, float eps, float p_fc,
                     unsigned long long seed_fc, int out_rows, int q_rep, float* P, float* O, float* u, float* gate,
                     float* s, float* f, float* y, float* mean, float* rstd, void* stream);

/* ---------------------------------------------------------------------------------------
 * The backward of ait_mha_core_fwd between the closing LayerNorm and the Q / K / V projections as ONE kernel, one
 * workgroup per sequence with its eight heads resident (SubLayers.py:82-100 backwards; SHBlock :22-39; Modules.py:16-29):
 *     du = df fc_w ;  (dO, dg) = SHBlock'(du, O, gate, sk_w) ;  (dq, dk, dv) = attention'(q, k, v, P, dO)
 * = ait_gemm_f32 (fc's input gradient) -> ait_sh_bwd -> ait_attn_bwd in one launch; du and dO never leave the chip.
 * H = 8, T = 64, d = 64.  df [n_seq*64, 512]: the gradient at fc's output (ait_ln_bwd's `da`); fc_w [512,64];
 * O [n_seq,8,64,64], gate [n_seq,512], P [n_seq,8,64,64]: what ait_mha_core_fwd saved; sk_w [512,64]; q / k / v / kv_rows /
 * scale / p_attn / seed_attn as given to the forward.  Written: dq / dk / dv with the addressing of q / k / v (pitches
 * lddq / lddk / lddv; rows >= kv_rows of dk / dv are not written), dg [n_seq,512] the gradient at SHBlock.sk's output
 * (d sk_w = dg^T s, d sk_b = column sums of dg: the caller's M-deep products, like d fc_w = df^T u).
 * ------------------------------------------------------------------------------------- */
int ait_mha_core_bwd(const float* df, const float* fc_w, const float* O, const float* gate, const float* sk_w,
                     const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* P, int n_seq,
                     int kv_rows, float scale, float p_attn, unsigned long long seed_attn, float* dq, int lddq, float* dk,
                     int lddk, float* dv, int lddv, float* dg, void* stream);

/* ---------------------------------------------------------------------------------------
 * The whole AIT forward as one call: Transformer.forward in eval mode
 * (lib/model/system/Models.py:231-280, n_layers = 1, d_model = 512, 8 heads of 64, d_inner = 2048:
 * the configuration of faster_rcnn_sys_transformer_sk_dilat.py:148-158).  Composes the entry
 * points above exactly as ait_amd/system.py does; see ait_amd/csrc/transformer.hip.
 *   x_props [bp * n_src, 1024]  proposal tokens (token-major = channels-last [bp, h, w, 1024];
 *                                n_src = h*w <= 64, 49 for 7x7 RoIAlign bins)
 *   x_query [bs * 64, 1024]     query tokens; sequence i belongs to pair i / (bp / bs)
 *   out     [bp * 64, 1024]     token-major (= channels-last [bp, 8, 8, 1024])
 * Weights: plain pointers into the state_dict tensors; w_qkv is the row concatenation
 * [w_qs.weight; w_ks.weight; w_vs.weight] ([1536, 512]).  Workspace: caller-owned scratch of
 * ait_transformer_workspace_bytes(bp, bs, n_src) bytes.  Training (dropout, saved activations,
 * backward): the *_fwd_train / *_bwd entry points below.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const float *w_qkv;            /* [1536, 512]  Q | K | V row blocks, no bias */
  const float *sk_w, *sk_b;      /* SHBlock: sh.sk.weight [512, 64], sh.sk.bias [512] */
  const float *fc_w;             /* fc.weight [512, 64], no bias */
  const float *ln_g, *ln_b;      /* layer_norm.weight / bias [512] */
} ait_mha_weights;
typedef struct {
  const float *w1, *b1;          /* w_1.weight [2048, 512], w_1.bias [2048] */
  const float *w2, *b2;          /* w_2.weight [512, 2048], w_2.bias [512] */
  const float *ln_g, *ln_b;
} ait_ffn_weights;
typedef struct {
  const float *enc_emb_w, *enc_emb_b;       /* enc_emb.0.weight [512, 1024], bias [512] */
  const float *dec_emb_w, *dec_emb_b;
  const float *dec_trans_w, *dec_trans_b;   /* dec_trans.0.weight [1024, 512], bias [1024] */
  const float *enc_ln_g, *enc_ln_b;         /* encoder.layer_norm */
  const float *dec_ln_g, *dec_ln_b;         /* decoder.layer_norm */
  const float *pos_table;                   /* position_enc.pos_table [64, 512] */
  ait_mha_weights enc_slf, dec_slf, dec_enc;
  ait_ffn_weights enc_ffn, dec_ffn;
} ait_transformer_weights;
/* The two sub-layer blocks the operator is made of, in eval mode (no dropout):
 *   ait_mha_block_fwd  y = LayerNorm(fc(selective_heads(attention(x_q W_q, x_kv W_k, x_kv W_v))) + x_q)
 *                      (MultiHeadAttention.forward, SubLayers.py:68-102).  xq [n_seq*64, 512];
 *                      xkv NULL (or == xq) for self-attention, else the memory [n_seq*kv_rows, 512]
 *                      with kv_rows <= 64; mask_mode / n_valid_keys as in ait_attn_fwd.
 *   ait_ffn_fwd        y = LayerNorm(W2 relu(W1 x + b1) + b2 + x)   (SubLayers.py:177-187), x [rows, 512]. */
size_t ait_mha_block_workspace_bytes(int n_seq, int kv_rows);
int ait_mha_block_fwd(const float* xq, const float* xkv, int n_seq, int kv_rows, int mask_mode,
                      int n_valid_keys, const ait_mha_weights* w, void* workspace, size_t workspace_bytes,
                      float* y, const ait_launch_ctx* ctx, void* stream);
size_t ait_ffn_workspace_bytes(long long rows);
int ait_ffn_fwd(const float* x, long long rows, const ait_ffn_weights* w, void* workspace,
                size_t workspace_bytes, float* y, const ait_launch_ctx* ctx, void* stream);
size_t ait_transformer_workspace_bytes(int bp, int bs, int n_src);
int ait_transformer_fwd(const float* x_props, const float* x_query, int bp, int bs, int n_src,
                        const ait_transformer_weights* w, void* workspace, size_t workspace_bytes,
                        float* out, const ait_launch_ctx* ctx, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training: the same blocks with dropout, a forward that SAVES what the backward needs, and the
 * backward.  Replaces the autograd graph PyTorch builds over MultiHeadAttention.forward
 * (lib/model/system/SubLayers.py:68-102), PositionwiseFeedForward.forward (:177-187) and
 * Transformer.forward (lib/model/system/Models.py:231-280) when trainval_net_voc.py:420 calls
 * loss.backward().
 *
 *   saved      caller-owned buffer of ait_*_saved_bytes(...) bytes: written by *_fwd_train, read by
 *              *_bwd (layout private to the library); the block INPUTS (xq / xkv / x, x_props /
 *              x_query) are not copied -- the caller keeps them alive and passes them again.
 *   workspace  scratch of the backward (ait_*_bwd_workspace_bytes).
 *   p_drop     dropout rate behind fc / w_2 / the positional prologue (the module's `dropout`);
 *   p_attn_drop  rate on the attention probabilities (fixed at 0.1 by the reference's
 *              ScaledDotProductAttention constructor, Modules.py:14).  0 <= p < 1; 0 disables.
 *   seed       one 64-bit seed per call; every dropout site inside derives its own with
 *              ait_dropout_seed (masks are a stateless hash of (site seed, element index), recomputed
 *              -- never stored -- by the backward, which must be given the same seed).
 *   gradients  INPUT gradients (dxq, dxkv, dx, d_x_props, d_x_query) are WRITTEN (d_x_props /
 *              d_x_query / dxkv may be NULL: not computed).  PARAMETER gradients are ACCUMULATED into
 *              the buffers of the ait_*_grads struct (the weight-gradient products are split-K sums
 *              combined with fp32 atomics); the caller zero-fills them for a plain gradient, or keeps
 *              a running sum across micro-batches.  A NULL member skips that gradient.
 *              w_qkv's gradient has the layout of w_qkv: rows [0,512) d w_qs, [512,1024) d w_ks,
 *              [1024,1536) d w_vs.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  float *w_qkv, *sk_w, *sk_b, *fc_w, *ln_g, *ln_b;
} ait_mha_grads;
typedef struct {
  float *w1, *b1, *w2, *b2, *ln_g, *ln_b;
} ait_ffn_grads;
typedef struct {
  float *enc_emb_w, *enc_emb_b, *dec_emb_w, *dec_emb_b, *dec_trans_w, *dec_trans_b;
  float *enc_ln_g, *enc_ln_b, *dec_ln_g, *dec_ln_b;
  ait_mha_grads enc_slf, dec_slf, dec_enc;
  ait_ffn_grads enc_ffn, dec_ffn;
} ait_transformer_grads;

/* seed of dropout site `site` under base seed `base` (pure function, splitmix64 finaliser) */
unsigned long long ait_dropout_seed(unsigned long long base, int site);
/* The decisions of ONE dropout site, written out: scale[j] = what the kernels multiply element first_index + j of the
 * site's tensor by, 1 / (1 - p_drop) or 0 (nn.Dropout's train-mode factor: lib/model/system/Modules.py:24,
 * SubLayers.py:98,184, Models.py:98,155).  A mask is a stateless hash of (site seed, element index), so this entry runs
 * the same device function on the same arguments as the operators do; it exists so that a checker can hand the product's
 * masks to a CPU reference and compare VALUES at p > 0 (tests/test_gpu_ait.py), and for debugging.  Site seeds and
 * element indices of ait_transformer_fwd_train(seed): block seed b = ait_dropout_seed(seed, k) with k = 16 encoder
 * prologue, 17 encoder self-attention, 18 encoder feed-forward, 19 decoder prologue, 20 decoder self-attention,
 * 21 decoder cross-attention, 22 decoder feed-forward; a prologue's site seed is b itself, an attention block's are
 * ait_dropout_seed(b, 0) (probabilities, element index = flat index of [n_seq, 8, 64, 64]) and ait_dropout_seed(b, 1)
 * (fc output, flat index of [n_seq * 64, 512]), a feed-forward's is ait_dropout_seed(b, 0) (flat index of
 * [rows, 512], rows as the block sees them: the encoder's run on the n_src compacted rows per sequence); prologues:
 * flat index of [n_seq * 64, 512]. */
int ait_dropout_mask(unsigned long long site_seed, unsigned long long first_index, long long count, float p_drop,
                     float* scale, void* stream);

size_t ait_mha_block_saved_bytes(int n_seq, int kv_rows);
int ait_mha_block_fwd_train(const float* xq, const float* xkv, int n_seq, int kv_rows, int mask_mode,
                            int n_valid_keys, const ait_mha_weights* w, float p_drop, float p_attn_drop,
                            unsigned long long seed, void* saved, size_t saved_bytes, float* y,
                            const ait_launch_ctx* ctx, void* stream);
size_t ait_mha_block_bwd_workspace_bytes(int n_seq, int kv_rows);
int ait_mha_block_bwd(const float* dy, const float* xq, const float* xkv, int n_seq, int kv_rows,
                      int mask_mode, int n_valid_keys, const ait_mha_weights* w, float p_drop,
                      float p_attn_drop, unsigned long long seed, const void* saved, size_t saved_bytes,
                      void* workspace, size_t workspace_bytes, float* dxq, float* dxkv,
                      const ait_mha_grads* grads, const ait_launch_ctx* ctx, void* stream);

size_t ait_ffn_saved_bytes(long long rows);
int ait_ffn_fwd_train(const float* x, long long rows, const ait_ffn_weights* w, float p_drop,
                      unsigned long long seed, void* saved, size_t saved_bytes, float* y,
                      const ait_launch_ctx* ctx, void* stream);
size_t ait_ffn_bwd_workspace_bytes(long long rows);
int ait_ffn_bwd(const float* dy, const float* x, long long rows, const ait_ffn_weights* w, float p_drop,
                unsigned long long seed, const void* saved, size_t saved_bytes, void* workspace,
                size_t workspace_bytes, float* dx, const ait_ffn_grads* grads, const ait_launch_ctx* ctx,
                void* stream);

int ait_transformer_io_bf16_ok(int bp, int bs, int n_src);      /* see AIT_CTX_IO_BF16 */
size_t ait_transformer_saved_bytes(int bp, int bs, int n_src);
/* The STORAGE FORMAT of `saved` (ABI v7).  Under AIT_CTX_BF16 / AIT_CTX_IO_BF16 the training forward stores some of its
 * activations as bf16 -- which ones depends on the flags and on the sizes.  It reports its decisions in *saved_format
 * (required, host memory; written before any launch): a magic in the top 12 bits plus the AIT_SAVED_* bits below.  The
 * caller hands that word to the backward of the SAME step; a backward whose own ctx implies another format (flags changed
 * between the two calls, another ctx) returns AIT_EINVAL instead of reading bf16 bytes as f32 or the reverse.  A word that
 * did not come from ait_transformer_fwd_train (wrong magic) is AIT_EINVAL too. */
#define AIT_SAVED_ENC_QKV16 1u     /* encoder self-attention: q / k / v and the block's input copy stored as bf16 */
#define AIT_SAVED_DEC_QKV16 2u     /* decoder self-attention: likewise */
#define AIT_SAVED_X_QKV16 4u       /* decoder cross-attention: likewise */
#define AIT_SAVED_ENC_FFN16 8u     /* encoder feed-forward: hidden tensor, input copy and weight copies stored as bf16 */
#define AIT_SAVED_DEC_FFN16 16u    /* decoder feed-forward: likewise */
#define AIT_SAVED_IO16 32u         /* dec_trans: bf16 copy of its input kept, `out` / `d_out` are bf16 tensors */
#define AIT_SAVED_DT16 64u         /* (ABI v8) dec_trans: bf16 copy of its input kept, its products on bf16 operands; `out` / `d_out` f32 */
int ait_transformer_fwd_train(const float* x_props, const float* x_query, int bp, int bs, int n_src,
                              const ait_transformer_weights* w, float p_drop, float p_attn_drop,
                              unsigned long long seed, void* saved, size_t saved_bytes, unsigned* saved_format,
                              float* out, const ait_launch_ctx* ctx, void* stream);
size_t ait_transformer_bwd_workspace_bytes(int bp, int bs, int n_src);
int ait_transformer_bwd(const float* d_out, const float* x_props, const float* x_query, int bp, int bs,
                        int n_src, const ait_transformer_weights* w, float p_drop, float p_attn_drop,
                        unsigned long long seed, const void* saved, size_t saved_bytes, unsigned saved_format,
                        void* workspace, size_t workspace_bytes, float* d_x_props, float* d_x_query,
                        const ait_transformer_grads* grads, const ait_launch_ctx* ctx, void* stream);
/* The same backward in THREE calls, for a data-parallel caller that wants the parameter gradients in bursts it can
 * hand to the gradient all-reduce while the rest of the backward still runs (trainval_net_voc.py:321-326,391-395:
 * the reference's DataParallel reduces after the whole backward; SURVEY 8e: buckets launched as backward produces
 * them).  part 0: dec_trans + decoder feed-forward (reads d_out); part 1: decoder cross- and self-attention,
 * decoder prologue, dec_emb (writes d_x_query); part 2: the encoder, enc_emb (writes d_x_props).  Called in this
 * order with the SAME workspace (the gradient carriers between the parts live in it) and the same other arguments;
 * a part only touches the `grads` members of its own layers (the others may be NULL). */
int ait_transformer_bwd_part(int part, const float* d_out, const float* x_props, const float* x_query, int bp, int bs,
                             int n_src, const ait_transformer_weights* w, float p_drop, float p_attn_drop,
                             unsigned long long seed, const void* saved, size_t saved_bytes, unsigned saved_format,
                             void* workspace, size_t workspace_bytes, float* d_x_props, float* d_x_query,
                             const ait_transformer_grads* grads, const ait_launch_ctx* ctx, void* stream);

/* ---------------------------------------------------------------------------------------
 * The proposal tail behind the AIT as one call per direction: both SKBlocks and RCNN_top
 * (ResNet layer4) with the mean over positions.  Replaces, in _fasterRCNN.forward
 * (lib/model/faster_rcnn/faster_rcnn_sys_transformer_sk_dilat.py:247-253),
 *     props_feat, query_feat = self.sk(x_props = props_feat, x_query = non_qry)           blocks_sys_transformer_sk_dilat.py:915-997
 *     props_feat = self._head_to_tail(props_feat) ; query_feat = self._head_to_tail(query_feat)
 *                                                                                         resnet_sys_transformer_sk_dilat.py:482-491, :85-111
 * and the autograd graph PyTorch builds over them.
 *   x_props [bp*64, C]   the AIT output, token-major (= channels-last [bp, 8, 8, C]); x_query [bs*64, C] likewise
 *   pooled  [bp + bs, 4*planes]   rows 0 .. bp-1: the proposals' pooled features, then the queries'
 * SKBlock as the reference EXECUTES it: f_k = relu(conv_k(x)) for the 1x1 and the 3x3 grouped branch (8 groups),
 * output sum_k f_k * f_k -- the branch-attention weights are computed by the reference and never used (:974-981).
 * The block is evaluated only at the 16 positions layer4's stride-2 1x1 convolutions read (same sums; the others
 * receive exactly zero gradient in the reference).
 * Weights: plain pointers into the state_dict tensors.  Convolution weights in CHANNELS-LAST memory
 * ([cout][kh][kw][cin / groups]); every BatchNorm of RCNN_top is frozen and in eval mode (:435-441,474-480) and is
 * given as (scale, shift) = (gamma / sqrt(var + eps), beta - mean * scale).  C % 1024 == 0, planes % 128 == 0,
 * 2 <= n_blocks <= 4 (block[0] is the one with the projection shortcut); n_blocks == 1 returns AIT_EUNSUPPORTED (the
 * caller's cue to run that tail block by block), anything else outside the range AIT_EINVAL.
 *   saved      caller-owned, ait_tail_saved_bytes(...): written by ait_tail_fwd, read by ait_tail_bwd (layout
 *              private to the library; in inference it is scratch)
 *   gradients  d_x_props / d_x_query are WRITTEN (either may be NULL); parameter gradients are ACCUMULATED into
 *              the buffers of ait_tail_grads (same layouts as the weights; a NULL member skips that gradient)
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const float *w1, *b1;          /* convs.0.0: 1x1 grouped conv weight [C][C/8], bias [C] */
  const float *w3, *b3;          /* convs.1.0: 3x3 grouped conv weight [C][3][3][C/8], bias [C] */
} ait_sk_weights;
typedef struct {
  const float *conv1, *conv2, *conv3, *down;     /* [planes][cin], [planes][3][3][planes], [4 planes][planes], [4 planes][cin] or NULL */
  const float *bn1_scale, *bn1_shift, *bn2_scale, *bn2_shift, *bn3_scale, *bn3_shift, *bnd_scale, *bnd_shift;
} ait_bottleneck_weights;
typedef struct {
  ait_sk_weights sk_props, sk_query;
  ait_bottleneck_weights block[4];
} ait_tail_weights;
typedef struct { float *w1, *b1, *w3, *b3; } ait_sk_grads;
typedef struct { float *conv1, *conv2, *conv3, *down; } ait_bottleneck_grads;
typedef struct {
  ait_sk_grads sk_props, sk_query;
  ait_bottleneck_grads block[4];
} ait_tail_grads;
size_t ait_tail_saved_bytes(int bp, int bs, int channels, int planes, int n_blocks);
/* saved_format (ABI v7; as ait_transformer_fwd_train's): the LAYOUT of `saved` depends on the call's product form -- layer4's
 * activations are kept in position-major row order (row = position * maps + map: its 3x3 convolutions then skip the window's
 * out-of-map taps) in the split product form, map-major under AIT_CTX_BF16 / AIT_CTX_NATIVE_F32.  The forward reports what it
 * wrote (a magic in the top 12 bits + AIT_TAIL_SAVED_PM / AIT_TAIL_SAVED_BF16; host memory, required), the backward is handed that word and returns
 * AIT_EINVAL if its own ctx implies the other order or the word is not one the forward reported. */
#define AIT_TAIL_SAVED_PM 1u
/* ... and (ABI v8) whether layer4's tensors are held in bf16: under AIT_CTX_BF16, from 1024 rows (64 maps) and for planes a
 * power of two >= 256, every activation and gradient of layer4 is STORED in bf16 (map-major rows padded to a multiple of 256,
 * inside the same `saved` / `workspace` sizes) and its products read bf16 operands from memory (ait_gemm_bf16s / _tn, the 3x3
 * convolutions through ait_conv_*_bf16s' window gather; folded weights converted once per call).  The SK blocks keep f32 tensors
 * (operands rounded in registers).  Smaller calls under AIT_CTX_BF16: f32 tensors, operands rounded in registers. */
#define AIT_TAIL_SAVED_BF16 2u
int ait_tail_fwd(const float* x_props, const float* x_query, int bp, int bs, int channels, int planes, int n_blocks,
                 const ait_tail_weights* w, void* saved, size_t saved_bytes, unsigned* saved_format, float* pooled,
                 const ait_launch_ctx* ctx, void* stream);
size_t ait_tail_bwd_workspace_bytes(int bp, int bs, int channels, int planes, int n_blocks);
int ait_tail_bwd(const float* d_pooled, const float* x_props, const float* x_query, int bp, int bs, int channels,
                 int planes, int n_blocks, const ait_tail_weights* w, const void* saved, size_t saved_bytes,
                 unsigned saved_format, void* workspace, size_t workspace_bytes, float* d_x_props, float* d_x_query,
                 const ait_tail_grads* grads, const ait_launch_ctx* ctx, void* stream);

/* ---------------------------------------------------------------------------------------
 * The detector's two heads behind the proposal tail (lib/model/faster_rcnn/faster_rcnn_sys_transformer_sk_dilat.py:
 * 283-290; modules resnet_sys_transformer_sk_dilat.py:425-433): the box regressor nn.Linear(F, n_bbox) on the pooled
 * proposal features, and the similarity classifier nn.Sequential(nn.Linear(2F, 8), nn.Linear(8, 2)) on
 * cat(props, query repeated over the R / bs proposals of its pair) -- whose output `score` [R, 2] is the per-proposal
 * similarity logit pair (:288).  Replaces three ATen addmm calls (vendor GEMM) and the [R, 2F] concatenation.
 *   props [R, F], query [bs, F] (R % bs == 0, F in {256, 512, 1024, 2048, 4096}, n_bbox <= 8); weights in nn.Linear layout: w_bbox [n_bbox, F],
 *   w1 [8, 2F] (columns [0, F) meet props, [F, 2F) the query), w2 [2, 8]; outputs bbox_pred [R, n_bbox], hidden [R, 8]
 *   (the first layer's output: saved for the backward), score [R, 2].
 *   backward: d_bbox / d_score may be NULL (no gradient from that head); d_props / d_query are WRITTEN (either may be
 *   NULL); parameter gradients are ACCUMULATED with fp32 atomics (a NULL member skips that gradient).
 * ------------------------------------------------------------------------------------- */
int ait_heads_fwd(const float* props, const float* query, int R, int bs, int F, const float* w_bbox,
                  const float* b_bbox, int n_bbox, const float* w1, const float* b1, const float* w2, const float* b2,
                  float* bbox_pred, float* hidden, float* score, void* stream);
size_t ait_heads_bwd_workspace_bytes(int R, int bs);
int ait_heads_bwd(const float* d_bbox, const float* d_score, const float* props, const float* query, int R, int bs,
                  int F, const float* w_bbox, int n_bbox, const float* w1, const float* w2, const float* hidden,
                  void* workspace, size_t workspace_bytes, float* d_props, float* d_query, float* d_w_bbox,
                  float* d_b_bbox, float* d_w1, float* d_b1, float* d_w2, float* d_b2, void* stream);

/* ---------------------------------------------------------------------------------------
 * Frozen batch-norm + residual + ReLU, one pass (NCHW fp32, x [n,C,HW]).
 * Replaces the eval-mode BatchNorm2d / "out += residual" / ReLU chains of the ResNet bottlenecks
 * (lib/model/faster_rcnn/resnet_sys_transformer_sk_dilat.py:85-111; every BatchNorm is frozen and
 * kept in eval mode during training, :435-441,457-480):
 *   y  = [relu]( x*scale[c] + shift[c] [+ residual] )      scale = gamma/sqrt(var+eps),
 *                                                          shift = beta - mean*scale
 *   dx = g*[y>0]*scale[c],  dres = g*[y>0],  g = dy + dy2  (the BN parameters get no gradient)
 * residual / dres / dy2 may be NULL.  dy2: the gradient arrives as two addends -- the block's output fed the next block's
 * convolution path and its shortcut (resnet_sys_transformer_sk_dilat.py:89-107) -- and is summed in this pass instead of
 * by an add kernel in front of it.
 * ------------------------------------------------------------------------------------- */
int ait_bn_act_fwd(const float* x, const float* scale, const float* shift, const float* residual,
                   int relu, long long n, int C, int HW, float* y, void* stream);
int ait_bn_act_bwd(const float* dy, const float* dy2, const float* y, const float* scale, int relu, long long n,
                   int C, int HW, float* dx, float* dres, void* stream);

/* The same pass over bf16 tensors in channels-last memory ([rows, C] rows of C channels, C % 8 == 0; scale / shift f32):
 * the C4 trunk of the bf16 configuration (BASELINE configs[4]: its convolutions run on MIOpen in bf16, this is the
 * frozen-BN / residual / ReLU pass between them).  f32 arithmetic, nearest-even rounding on the way out. */
int ait_bn_act_fwd_bf16(const void* x, const float* scale, const float* shift, const void* residual, int relu,
                        long long rows, int C, void* y, void* stream);
int ait_bn_act_bwd_bf16(const void* dy, const void* dy2, const void* y, const float* scale, int relu, long long rows, int C,
                        void* dx, void* dres, void* stream);

/* SKBlock tail as the reference executes it (blocks_sys_transformer_sk_dilat.py:966-981: two
 * conv+ReLU branches, `v = f * f`, summed): y = relu(a)^2 + relu(b)^2 over n fp32 elements
 * (n % 4 == 0, 16-byte aligned), and da = 2*relu(a)*dy, db = 2*relu(b)*dy. */
int ait_sk_sqsum_fwd(const float* a, const float* b, long long n, float* y, void* stream);
int ait_sk_sqsum_bwd(const float* dy, const float* a, const float* b, long long n, float* da,
                     float* db, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AIT_HIP_H_ */
