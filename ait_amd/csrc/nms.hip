// ait_amd/csrc/nms.hip -- greedy IoU non-maximum suppression, entirely on the device.
//
// Semantics = the reference's CPU operator (lib/model/csrc/cpu/nms_cpu.cpp:5-65), which is what
// the parity contract names: area = (x2-x1+1)*(y2-y1+1), suppress when IoU >= thr, report the
// surviving ORIGINAL indices in ascending order as int64.  (The reference's CUDA kernel uses
// `>` and a host-side scan after a device->host copy, lib/model/csrc/cuda/nms.cu:60,99-123;
// neither is reproduced.)  Built with -ffp-contract=off: the IoU must be the same fp32 value
// the CPU computes, or ties at the threshold flip.
//
// Three phases, all enqueued on the caller's stream, no host round trip:
//   1. nms_mask_kernel   64x64 tiles of the upper triangle of the (sorted) IoU matrix ->
//                        one 64-bit suppression word per (row, column block)   [HBM-bound
//                        write of n*ceil(n/64)*8/2 bytes; boxes are L2-resident]
//   2. nms_scan_kernel   ONE workgroup walks the rows in score order.  The running "removed"
//                        bit-vector lives in LDS; per 64-row block wave 0 resolves the
//                        intra-block chain with scalar readlane/OR (no memory in the chain),
//                        then all waves OR the kept rows' mask words into the vector with
//                        coalesced, mutually independent loads (one latency per block, not
//                        per row).  Stops early once max_keep survivors are found.
//   3. nms_compact_kernel (only when an explicit `order` is given) re-expresses survivors as
//                        ascending original indices.
#include "common.h"

namespace {

constexpr int kTile = 64;
constexpr int kScanThreads = 512;

__device__ __forceinline__ float box_area(const float4 b) {
  return (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
}

// nms_cpu.cpp:49-61
__device__ __forceinline__ bool suppresses(const float4 a, float area_a, const float4 b,
                                           float area_b, float thr) {
  float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
  float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  float w = fmaxf(0.f, xx2 - xx1 + 1.f), h = fmaxf(0.f, yy2 - yy1 + 1.f);
  float inter = w * h;
  float ovr = inter / (area_a + area_b - inter);
  return ovr >= thr;
}

__global__ __launch_bounds__(kTile) void nms_mask_kernel(const float4* __restrict__ boxes,
                                                         const int64_t* __restrict__ order,
                                                         int n, float thr, int nb,
                                                         unsigned long long* __restrict__ mask) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;  // lower triangle is never read
  boxes += (size_t)blockIdx.z * n;            // batched call: one image per grid z-slice
  mask += (size_t)blockIdx.z * n * nb;
  __shared__ float4 cbox[kTile];
  __shared__ float carea[kTile];
  const int lane = threadIdx.x;
  const int col = cb * kTile + lane;
  if (col < n) {
    float4 b = boxes[order ? order[col] : col];
    cbox[lane] = b;
    carea[lane] = box_area(b);
  }
  __syncthreads();
  const int row = rb * kTile + lane;
  if (row >= n) return;
  const float4 a = boxes[order ? order[row] : row];
  const float area_a = box_area(a);
  const int ncol = min(kTile, n - cb * kTile);
  unsigned long long bits = 0ull;
  const int start = (rb == cb) ? lane + 1 : 0;
  for (int j = start; j < ncol; j++)
    if (suppresses(a, area_a, cbox[j], carea[j], thr)) bits |= 1ull << j;
  mask[(size_t)row * nb + cb] = bits;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, lane);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// Dynamic LDS: removed[nb] (u64).  alive[n] (u8, by sorted position) is written when
// order != NULL; otherwise survivors are written straight to keep[] in ascending order.
//
// Per 64-row block rb the only truly serial work is the intra-block greedy chain; everything that
// touches memory is taken off that path:
//   * wave 0 owns the chain.  One iteration ahead it prefetches, per lane (= row of the NEXT block),
//     that row's diagonal word and its word for the block after it ("critical column").  The chain
//     runs on scalar registers (readlane); when row i survives, its critical word is OR-ed in the
//     same scalar loop, so removed[rb+1] is complete the moment the chain ends -- no load in between.
//   * waves 1..7 meanwhile OR the PREVIOUS block's surviving rows into the words w >= rb+1 that are
//     not needed before the next iteration (coalesced row segments, four independent loads in
//     flight per thread, 64-bit LDS atomic OR), hidden behind the chain.
// One barrier per block.  Stops early once max_keep survivors are found.
__global__ __launch_bounds__(kScanThreads) void nms_scan_kernel(
    const unsigned long long* __restrict__ mask, int n, int nb, int max_keep,
    unsigned char* __restrict__ alive, long long* __restrict__ keep, int* __restrict__ n_keep,
    long long keep_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long removed[];
  mask += (size_t)blockIdx.x * n * nb;        // batched call: one workgroup per image
  keep += (size_t)blockIdx.x * keep_stride;
  n_keep += blockIdx.x;
  __shared__ unsigned long long kept_bits[2];
  __shared__ int kept_rows[2][kTile];    // the surviving rows of the current / previous block, in order
  __shared__ int kept_total;
  const int tid = threadIdx.x;
  for (int i = tid; i < nb; i += kScanThreads) removed[i] = 0ull;
  if (tid == 0) kept_total = 0;
  __syncthreads();
  unsigned long long diag = 0ull, crit = 0ull;
  if (tid < kTile && tid < n) {
    diag = mask[(size_t)tid * nb + 0];
    if (nb > 1) crit = mask[(size_t)tid * nb + 1];
  }
  for (int rb = 0; rb < nb; rb++) {
    const int base = rb * kTile;
    const int cnt = min(kTile, n - base);
    const int cur = rb & 1;
    if (tid < kTile) {  // wave 0: intra-block greedy chain, all in scalar registers
      unsigned long long word = removed[rb];
      if (cnt < kTile) word |= ~0ull << cnt;
      unsigned long long next_diag = 0ull, next_crit = 0ull;
      if (rb + 1 < nb && base + kTile + tid < n) {   // prefetch for the next block
        const unsigned long long* __restrict__ r = mask + (size_t)(base + kTile + tid) * nb + rb + 1;
        next_diag = r[0];
        if (rb + 2 < nb) next_crit = r[1];
      }
      unsigned long long cw = 0ull;      // what this block's survivors suppress in block rb + 1
      // one step per SURVIVOR, not per row (a block keeps ~10 of its 64): the lowest row still alive survives, its
      // diagonal word removes rows above it (s_ff1 on the scalar side finds the next one)
      for (unsigned long long todo = ~word; todo;) {
        const int i = __builtin_ctzll(todo);
        const unsigned long long d = readlane64(diag, i);
        word |= d;
        cw |= readlane64(crit, i);
        todo = (todo & (todo - 1ull)) & ~d;
      }
      diag = next_diag;
      crit = next_crit;
      unsigned long long k = ~word;  // survivors of this block
      int total = kept_total;
      if (max_keep > 0 && total + __popcll(k) > max_keep) {
        // keep only the first (max_keep - total) survivors in score order
        int need = max_keep - total;
        unsigned long long kk = 0ull, rest = k;
        for (int c = 0; c < need; c++) {
          unsigned long long low = rest & (~rest + 1ull);
          kk |= low;
          rest ^= low;
        }
        k = kk;
      }
      if ((k >> tid) & 1ull) kept_rows[cur][__popcll(k & ((1ull << tid) - 1ull))] = tid;
      if (alive) {
        if (tid < cnt) alive[base + tid] = (unsigned char)((k >> tid) & 1ull);
      } else if ((k >> tid) & 1ull) {
        int rank = __popcll(k & ((1ull << tid) - 1ull));
        keep[total + rank] = base + tid;
      }
      if (tid == 0) {
        kept_bits[cur] = k;
        kept_total = total + __popcll(k);
        if (rb + 1 < nb && cw) atomicOr(&removed[rb + 1], cw);
      }
    } else if (rb > 0) {
      // waves 1..7: the previous block's survivors -> removed[w] for w >= rb + 1 (its word for block
      // rb itself was the critical column wave 0 handled one iteration ago).  Thread = (row group g
      // of 7, word lane): a word lane walks the column words (coalesced 512-B row segments per wave),
      // a row group takes every 7th surviving row, ten per step with independent loads in flight.
      const int prev = cur ^ 1;
      const int kc = __popcll(kept_bits[prev]);
      const int t = tid - kTile, g = t >> 6, wl = t & 63;
      constexpr int G = kScanThreads / kTile - 1;
      const unsigned long long* __restrict__ mrow = mask + (size_t)(base - kTile) * nb;
      for (int w = rb + 1 + wl; w < nb; w += kTile) {
        unsigned long long acc = 0ull;
        // (ten loads in flight per thread: 70 rows per pass, so a block's <= 64 survivors take ONE memory round trip --
        // this phase, not wave 0's chain, is what a block costs: 441 -> see profiles/r04_categories.txt)
        for (int j = g; j < kc; j += 10 * G) {
          unsigned long long a[10];
#pragma unroll
          for (int u = 0; u < 10; u++) a[u] = mrow[(size_t)kept_rows[prev][min(j + u * G, kc - 1)] * nb + w];   // clamped duplicates OR harmlessly
#pragma unroll
          for (int u = 0; u < 10; u++) acc |= a[u];
        }
        if (acc) atomicOr(&removed[w], acc);
      }
    }
    __syncthreads();
    if (max_keep > 0 && kept_total >= max_keep) {
      if (alive)
        for (int i = base + cnt + tid; i < n; i += kScanThreads) alive[i] = 0;
      break;
    }
  }
  if (tid == 0) *n_keep = kept_total;
}

// alive[] is indexed by sorted position; survivors must come out as ascending ORIGINAL index.
__global__ __launch_bounds__(kScanThreads) void nms_compact_kernel(
    const unsigned char* __restrict__ alive, const int64_t* __restrict__ order, int n,
    unsigned char* __restrict__ flag, long long* __restrict__ keep) {
  __shared__ int wave_tot[kScanThreads / AIT_WAVE];
  __shared__ int running;
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += kScanThreads) flag[order[i]] = alive[i];
  if (tid == 0) running = 0;
  __syncthreads();
  for (int base = 0; base < n; base += kScanThreads) {
    const int i = base + tid;
    const int f = (i < n) ? flag[i] : 0;
    const unsigned long long ballot = __ballot(f);
    const int lane = tid & 63, wv = tid >> 6;
    const int within = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wv] = __popcll(ballot);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wv; w++) off += wave_tot[w];
    if (f) keep[off + within] = i;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < kScanThreads / AIT_WAVE; w++) t += wave_tot[w];
      running += t;
    }
    __syncthreads();
  }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

AIT_API size_t ait_nms_workspace_bytes(int n) {
  if (n <= 0) return 0;
  const size_t nb = (size_t)(n + kTile - 1) / kTile;
  return align_up((size_t)n * nb * 8, 256) + 2 * align_up((size_t)n, 256);
}

AIT_API int ait_nms(const float* boxes, const int64_t* order, int n, float thr, int max_keep,
                       void* workspace, size_t workspace_bytes, int64_t* keep, int32_t* n_keep,
                       void* stream) {
  if (n < 0 || !n_keep) return AIT_EINVAL;
  hipStream_t s = ait_stream(stream);
  if (n == 0) {
    if (hipMemsetAsync(n_keep, 0, sizeof(int32_t), s) != hipSuccess) return AIT_ELAUNCH;
    return AIT_OK;
  }
  if (!boxes || !keep || !workspace) return AIT_EINVAL;
  if (order && max_keep > 0) return AIT_EINVAL;
  if (workspace_bytes < ait_nms_workspace_bytes(n)) return AIT_EWORKSPACE;
  if ((reinterpret_cast<uintptr_t>(boxes) & 15) != 0) return AIT_EINVAL;
  const int nb = (n + kTile - 1) / kTile;
  if ((size_t)nb * 8 > 60 * 1024) return AIT_EUNSUPPORTED;  // > 491k boxes
  char* ws = static_cast<char*>(workspace);
  auto* mask = reinterpret_cast<unsigned long long*>(ws);
  unsigned char* alive = reinterpret_cast<unsigned char*>(ws + align_up((size_t)n * nb * 8, 256));
  unsigned char* flag = alive + align_up((size_t)n, 256);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb), dim3(kTile), 0, s,
                     reinterpret_cast<const float4*>(boxes), order, n, thr, nb, mask);
  AIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(kScanThreads), (size_t)nb * 8, s, mask, n,
                     nb, max_keep, order ? alive : nullptr,
                     reinterpret_cast<long long*>(keep), n_keep, 0ll);
  AIT_CHECK_LAUNCH();
  if (order) {
    hipLaunchKernelGGL(nms_compact_kernel, dim3(1), dim3(kScanThreads), 0, s, alive, order, n,
                       flag, reinterpret_cast<long long*>(keep));
    AIT_CHECK_LAUNCH();
  }
  return AIT_OK;
}

AIT_API size_t ait_nms_batched_workspace_bytes(int batch, int n) {
  if (n <= 0 || batch <= 0) return 0;
  const size_t nb = (size_t)(n + kTile - 1) / kTile;
  // image b's mask slab starts at word b*n*nb (8-byte accesses only: no per-image alignment needed)
  return align_up((size_t)batch * n * nb * 8, 256);
}

AIT_API int ait_nms_batched(const float* boxes, int batch, int n, float thr, int max_keep,
                            void* workspace, size_t workspace_bytes, int64_t* keep,
                            long long keep_stride, int32_t* n_keep, void* stream) {
  if (n < 0 || batch < 0 || !n_keep) return AIT_EINVAL;
  hipStream_t s = ait_stream(stream);
  if (batch == 0) return AIT_OK;
  if (n == 0) {
    if (hipMemsetAsync(n_keep, 0, sizeof(int32_t) * batch, s) != hipSuccess) return AIT_ELAUNCH;
    return AIT_OK;
  }
  if (!boxes || !keep || !workspace) return AIT_EINVAL;
  if (workspace_bytes < ait_nms_batched_workspace_bytes(batch, n)) return AIT_EWORKSPACE;
  if ((reinterpret_cast<uintptr_t>(boxes) & 15) != 0) return AIT_EINVAL;
  const int nb = (n + kTile - 1) / kTile;
  if ((size_t)nb * 8 > 60 * 1024) return AIT_EUNSUPPORTED;
  auto* mask = reinterpret_cast<unsigned long long*>(workspace);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb, batch), dim3(kTile), 0, s,
                     reinterpret_cast<const float4*>(boxes), nullptr, n, thr, nb, mask);
  AIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(kScanThreads), (size_t)nb * 8, s, mask, n,
                     nb, max_keep, nullptr, reinterpret_cast<long long*>(keep), n_keep,
                     keep_stride);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}
