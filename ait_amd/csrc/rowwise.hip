// ait_amd/csrc/rowwise.hip -- the HBM-bound row kernels of the AIT path for gfx950:
//   * ait_ln_fwd / ait_ln_bwd   y = LayerNorm_eps( dropout(a[src_row] + pos[t]) + residual )
//       covers the encoder / decoder prologue (lib/model/system/Models.py:98-99,155-156, with the
//       zero padding of :269-270 and the repeat over proposals of :250 folded into the row map)
//       and the post-attention / post-FFN "dropout, += residual, LayerNorm" tails
//       (lib/model/system/SubLayers.py:97-100,182-185).
//   * ait_sh_fwd / ait_sh_bwd   selective heads (SHBlock, SubLayers.py:22-39) + the head sum
//       of MultiHeadAttention.forward (SubLayers.py:92).
//
// Wave64 design: one wavefront per 512-float row (8 floats per lane = two 16-B loads), wave
// reductions by DPP/shuffle butterflies, no LDS on the forward path; everything is sized so a
// row is read once and written once.  Dropout is a stateless counter-based hash of
// (seed, element index), recomputed (not stored) in the backward pass.
#include "common.h"

namespace {

constexpr int kD = 512;      // d_model of the AIT path (channels / 2)
constexpr int kVec = 8;      // floats per lane
constexpr int kRowsPerBlock = 4;
constexpr int kThreads = 256;

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// keep-probability 1-p decision for element `idx` of dropout site `seed`
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned long long idx,
                                            float p, float inv_keep) {
  unsigned h = mix32((unsigned)idx ^ mix32((unsigned)(idx >> 32) + (unsigned)seed) ^
                     (unsigned)(seed >> 32) * 0x9e3779b9u);
  float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.f;
}

struct RowMap {
  int seq_len, src_rows, rep;
  int dy_rows;     // backward only: rows per sequence that `dy` holds (compacted gradient); == seq_len: all
  int out_rows;    // forward only: rows per sequence that `y` holds (row (q, t) at y[q * out_rows + t], t >= out_rows
                   // not written); == seq_len: all
};
// out row r=(q,t) -> source row of `a`, or -1 for a zero (padding) row
__device__ __forceinline__ long long src_row(const RowMap m, long long r) {
  const long long q = r / m.seq_len;
  const int t = (int)(r - q * m.seq_len);
  if (t >= m.src_rows) return -1;
  return (q / m.rep) * m.src_rows + t;
}

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[kVec]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0];
  const float4 b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(float* __restrict__ p, const float (&v)[kVec]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// z = dropout(a + pos) + residual for this lane's 8 columns of out row r
__device__ __forceinline__ void form_z(const float* __restrict__ a, const float* __restrict__ pos,
                                       const float* __restrict__ res, const RowMap m, long long r,
                                       int c0, float p, float inv_keep, unsigned long long seed,
                                       float (&z)[kVec], float (&ds)[kVec]) {
  const long long sr = src_row(m, r);
  if (sr >= 0) load8(a + sr * kD + c0, z);
  else {
#pragma unroll
    for (int i = 0; i < kVec; i++) z[i] = 0.f;
  }
  if (pos) {
    float pv[kVec];
    load8(pos + (size_t)(r % m.seq_len) * kD + c0, pv);
#pragma unroll
    for (int i = 0; i < kVec; i++) z[i] += pv[i];
  }
#pragma unroll
  for (int i = 0; i < kVec; i++) {
    ds[i] = (p > 0.f) ? drop_scale(seed, (unsigned long long)r * kD + c0 + i, p, inv_keep) : 1.f;
    z[i] *= ds[i];
  }
  if (res) {
    float rv[kVec];
    load8(res + (size_t)r * kD + c0, rv);
#pragma unroll
    for (int i = 0; i < kVec; i++) z[i] += rv[i];
  }
}

__global__ __launch_bounds__(kThreads) void ln_fwd_kernel(
    const float* __restrict__ a, const float* __restrict__ pos, const float* __restrict__ res,
    const float* __restrict__ gamma, const float* __restrict__ beta, long long rows, RowMap m,
    float eps, float p, unsigned long long seed, float* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * kVec;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float g[kVec], b[kVec];
  load8(gamma + c0, g);
  load8(beta + c0, b);
  for (long long r = (long long)blockIdx.x * kRowsPerBlock + wave; r < rows;
       r += (long long)gridDim.x * kRowsPerBlock) {
    float z[kVec], ds[kVec];
    form_z(a, pos, res, m, r, c0, p, inv_keep, seed, z, ds);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kVec; i++) s += z[i];
    const float mu = wave_sum(s) * (1.f / kD);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kVec; i++) {
      const float d = z[i] - mu;
      q += d * d;
    }
    const float var = wave_sum(q) * (1.f / kD);  // biased, like nn.LayerNorm
    const float rs = 1.f / sqrtf(var + eps);
    float o[kVec];
#pragma unroll
    for (int i = 0; i < kVec; i++) o[i] = (z[i] - mu) * rs * g[i] + b[i];
    if (m.out_rows == m.seq_len) {
      store8(y + (size_t)r * kD + c0, o);
    } else {      // compacted output: only the first out_rows rows of a sequence are read again
      const long long q = r / m.seq_len;
      const int t = (int)(r - q * m.seq_len);
      if (t < m.out_rows) store8(y + (size_t)(q * m.out_rows + t) * kD + c0, o);
    }
    if (lane == 0) {
      if (mean) mean[r] = mu;
      if (rstd) rstd[r] = rs;
    }
  }
}

// dgamma/dbeta: per-lane partials over the rows of this block, summed across the 4 waves in
// LDS, then ONE atomic per column per block.
__global__ __launch_bounds__(kThreads) void ln_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ a, const float* __restrict__ pos,
    const float* __restrict__ res, const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ rstd, long long rows, RowMap m, float p, unsigned long long seed,
    float* __restrict__ da, float* __restrict__ dres, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dcolsum, unsigned short* __restrict__ da16) {
  __shared__ float red[3][kRowsPerBlock][kD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * kVec;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float g[kVec], dg[kVec], db[kVec], dc[kVec];
  load8(gamma + c0, g);
#pragma unroll
  for (int i = 0; i < kVec; i++) dg[i] = db[i] = dc[i] = 0.f;
  for (long long r = (long long)blockIdx.x * kRowsPerBlock + wave; r < rows;
       r += (long long)gridDim.x * kRowsPerBlock) {
    float z[kVec], ds[kVec], d[kVec];
    form_z(a, pos, res, m, r, c0, p, inv_keep, seed, z, ds);
    if (m.dy_rows == m.seq_len) {
      load8(dy + (size_t)r * kD + c0, d);
    } else {   // compacted gradient: only the first dy_rows rows of a sequence received one
      const long long q = r / m.seq_len;
      const int t = (int)(r - q * m.seq_len);
      if (t < m.dy_rows) load8(dy + (size_t)(q * m.dy_rows + t) * kD + c0, d);
      else {
#pragma unroll
        for (int i = 0; i < kVec; i++) d[i] = 0.f;
      }
    }
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f, xh[kVec], wd[kVec];
#pragma unroll
    for (int i = 0; i < kVec; i++) {
      xh[i] = (z[i] - mu) * rs;
      wd[i] = d[i] * g[i];
      s1 += wd[i];
      s2 += wd[i] * xh[i];
      dg[i] += d[i] * xh[i];
      db[i] += d[i];
    }
    s1 = wave_sum(s1) * (1.f / kD);
    s2 = wave_sum(s2) * (1.f / kD);
    float dz[kVec];
#pragma unroll
    for (int i = 0; i < kVec; i++) dz[i] = (wd[i] - s1 - xh[i] * s2) * rs;
    if (dres) store8(dres + (size_t)r * kD + c0, dz);
    if (da || da16 || dcolsum) {
      const long long sr = src_row(m, r);      // padded rows have no source: no gradient, no column sum
      if (sr >= 0) {
#pragma unroll
        for (int i = 0; i < kVec; i++) {
          dz[i] *= ds[i];
          dc[i] += dz[i];
        }
        if (da) store8(da + (size_t)(m.rep == 1 ? sr : r) * kD + c0, dz);   // rep == 1: source indexing
        if (da16) {      // the same gradient rounded to bf16 (nearest even): the bf16-storage feed-forward's operand
          typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
          auto pk = [](float x0, float x1) { bf16x2_ t; t[0] = (__bf16)x0; t[1] = (__bf16)x1; return __builtin_bit_cast(unsigned, t); };
          *reinterpret_cast<uint4*>(da16 + (size_t)(m.rep == 1 ? sr : r) * kD + c0) =
              make_uint4(pk(dz[0], dz[1]), pk(dz[2], dz[3]), pk(dz[4], dz[5]), pk(dz[6], dz[7]));
        }
      }
    }
  }
  if (dgamma || dcolsum) {
#pragma unroll
    for (int i = 0; i < kVec; i++) {
      red[0][wave][c0 + i] = dg[i];
      red[1][wave][c0 + i] = db[i];
      red[2][wave][c0 + i] = dc[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < kD; c += kThreads) {
      float sg = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
      for (int w = 0; w < kRowsPerBlock; w++) {
        sg += red[0][w][c];
        sb += red[1][w][c];
        sc += red[2][w][c];
      }
      if (dgamma) {
        unsafeAtomicAdd(dgamma + c, sg);
        unsafeAtomicAdd(dbeta + c, sb);
      }
      if (dcolsum) unsafeAtomicAdd(dcolsum + c, sc);
    }
  }
}

// out[c] += sum_r x[r*ld + c]: bias gradients (column sums over token rows).  A 1024-thread block sums a band of rows:
// thread (tx, ty) takes column quad tx (16-B loads) on rows ty, ty + RY, ... of the band, the RY partial rows meet in LDS and
// the block adds its sums to `out` with one atomic per column.  Few, tall bands (~512 blocks): a column's atomics all hit
// ONE address, and a thousand of them in a row cost more than the read (round 5: the 128-row bands of the first version ran
// at 1.8-3.5 TB/s, this form at 5.7-5.9; csrc/gemm_bf16s.hip's colsum_bf16_kernel is the same kernel on bf16).
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, long long rows, int cols, long long ld, int band,
                                                      int G, float* __restrict__ out) {
  extern __shared__ float colsum_part[];      // [RY][G * 4]
  const int RY = blockDim.x / G, tx = threadIdx.x % G, ty = threadIdx.x / G;
  const int c4 = tx + blockIdx.y * G;
  const long long r0 = (long long)blockIdx.x * band;
  const long long r1 = r0 + band < rows ? r0 + band : rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < cols / 4) {
    const float* __restrict__ p = x + (r0 + ty) * ld + 4 * c4;
    const long long step = (long long)RY * ld;
#pragma unroll 8
    for (long long r = r0 + ty; r < r1; r += RY, p += step) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  *reinterpret_cast<float4*>(colsum_part + (ty * G + tx) * 4) = acc;
  __syncthreads();
  for (int c = threadIdx.x; c < G * 4; c += blockDim.x) {
    const int col = blockIdx.y * G * 4 + c;
    if (col >= cols) continue;
    float s_ = 0.f;
    for (int y = 0; y < RY; y++) s_ += colsum_part[y * G * 4 + c];
    unsafeAtomicAdd(out + col, s_);
  }
}

// out[g*E + e] += sum_{j in this block's slice of rep} x[(g*rep + j)*E + e]: the gradient of a row block
// that was repeated `rep` times (the query sequence over the proposals of its pair, Models.py:250).
constexpr int kRepSlice = 16;
__global__ __launch_bounds__(256) void rep_sum_kernel(const float* __restrict__ x, int rep, long long E,
                                                      float* __restrict__ out) {
  const long long e4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e4 * 4 >= E) return;
  const int g = blockIdx.y;
  const int j0 = blockIdx.z * kRepSlice, j1 = min(rep, j0 + kRepSlice);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* __restrict__ p = x + ((size_t)g * rep + j0) * E + 4 * e4;
#pragma unroll 8
  for (int j = j0; j < j1; j++, p += E) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  float* o = out + (size_t)g * E + 4 * e4;
  unsafeAtomicAdd(o + 0, acc.x);
  unsafeAtomicAdd(o + 1, acc.y);
  unsafeAtomicAdd(o + 2, acc.z);
  unsafeAtomicAdd(o + 3, acc.w);
}

// ---------------------------------------------------------------------------------------------
// Selective heads.  O [n_seq, 8, 64, 64] (head, token, channel).  One 256-thread workgroup per
// sequence; thread = (token group tq = tid>>6 of 16 tokens, channel c = tid&63), so every global
// access is a coalesced 256-B row of 64 channels.
// ---------------------------------------------------------------------------------------------
constexpr int kH = 8, kT = 64, kC = 64;

__global__ __launch_bounds__(kThreads) void sh_fwd_kernel(
    const float* __restrict__ O, const float* __restrict__ sk_w, const float* __restrict__ sk_b,
    float* __restrict__ u, float* __restrict__ gate_out, float* __restrict__ s_out) {
  __shared__ float part[4][kC];
  __shared__ float s[kC];
  __shared__ float gate[kH * kC];
  const int n = blockIdx.x, tid = threadIdx.x, c = tid & 63, tq = tid >> 6;
  const float* __restrict__ On = O + (size_t)n * kH * kT * kC;
  float acc = 0.f;
  for (int h = 0; h < kH; h++)
    for (int t = tq * 16; t < tq * 16 + 16; t++) acc += On[(h * kT + t) * kC + c];
  part[tq][c] = acc;
  __syncthreads();
  if (tid < kC) {
    const float v = (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]) * (1.f / kT);
    s[tid] = v;
    if (s_out) s_out[(size_t)n * kC + tid] = v;
  }
  __syncthreads();
  for (int j = tid; j < kH * kC; j += kThreads) {  // g = sk_w s + sk_b
    const float4* __restrict__ w = reinterpret_cast<const float4*>(sk_w + (size_t)j * kC);
    float d = sk_b[j];
#pragma unroll
    for (int q = 0; q < kC / 4; q++) {
      const float4 wv = w[q];
      d += wv.x * s[4 * q] + wv.y * s[4 * q + 1] + wv.z * s[4 * q + 2] + wv.w * s[4 * q + 3];
    }
    gate[j] = d;
  }
  __syncthreads();
  if (tid < kC) {  // softmax over the 8 heads, per channel
    float mx = gate[tid];
#pragma unroll
    for (int h = 1; h < kH; h++) mx = fmaxf(mx, gate[h * kC + tid]);
    float e[kH], sum = 0.f;
#pragma unroll
    for (int h = 0; h < kH; h++) {
      e[h] = expf(gate[h * kC + tid] - mx);
      sum += e[h];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int h = 0; h < kH; h++) gate[h * kC + tid] = e[h] * inv;
  }
  __syncthreads();
  if (gate_out)
    for (int j = tid; j < kH * kC; j += kThreads) gate_out[(size_t)n * kH * kC + j] = gate[j];
  float gl[kH];
#pragma unroll
  for (int h = 0; h < kH; h++) gl[h] = gate[h * kC + c];
  float* __restrict__ un = u + (size_t)n * kT * kC;
  for (int t = tq * 16; t < tq * 16 + 16; t++) {
    float v = 0.f;
#pragma unroll
    for (int h = 0; h < kH; h++) v += On[(h * kT + t) * kC + c] * gl[h];
    un[t * kC + c] = v;
  }
}

__global__ __launch_bounds__(kThreads) void sh_bwd_kernel(
    const float* __restrict__ du, const float* __restrict__ O, const float* __restrict__ gate_in,
    const float* __restrict__ sk_w, float* __restrict__ dO, float* __restrict__ dg_out) {
  __shared__ float part[4][kH * kC];
  __shared__ float dg[kH * kC];
  __shared__ float dsv[kC];
  const int n = blockIdx.x, tid = threadIdx.x, c = tid & 63, tq = tid >> 6;
  const float* __restrict__ On = O + (size_t)n * kH * kT * kC;
  const float* __restrict__ dun = du + (size_t)n * kT * kC;
  float gl[kH], dgate[kH];
#pragma unroll
  for (int h = 0; h < kH; h++) {
    gl[h] = gate_in[(size_t)n * kH * kC + h * kC + c];
    dgate[h] = 0.f;
  }
  for (int t = tq * 16; t < tq * 16 + 16; t++) {
    const float d = dun[t * kC + c];
#pragma unroll
    for (int h = 0; h < kH; h++) dgate[h] += d * On[(h * kT + t) * kC + c];
  }
#pragma unroll
  for (int h = 0; h < kH; h++) part[tq][h * kC + c] = dgate[h];
  __syncthreads();
  if (tid < kC) {  // softmax backward over heads
    float dgt[kH], dot = 0.f;
#pragma unroll
    for (int h = 0; h < kH; h++) {
      const int j = h * kC + tid;
      dgt[h] = part[0][j] + part[1][j] + part[2][j] + part[3][j];
      dot += dgt[h] * gl[h];  // tq == 0 for these threads, so gl[] is this channel's gate
    }
#pragma unroll
    for (int h = 0; h < kH; h++) {
      const float v = gl[h] * (dgt[h] - dot);
      dg[h * kC + tid] = v;
      dg_out[(size_t)n * kH * kC + h * kC + tid] = v;
    }
  }
  __syncthreads();
  if (tid < kC) {  // ds = sk_w^T dg, already divided by T (s is a mean over tokens)
    float d = 0.f;
    for (int j = 0; j < kH * kC; j++) d += sk_w[(size_t)j * kC + tid] * dg[j];
    dsv[tid] = d * (1.f / kT);
  }
  __syncthreads();
  const float dsc = dsv[c];
  float* __restrict__ dOn = dO + (size_t)n * kH * kT * kC;
  for (int t = tq * 16; t < tq * 16 + 16; t++) {
    const float d = dun[t * kC + c];
#pragma unroll
    for (int h = 0; h < kH; h++) dOn[(h * kT + t) * kC + c] = d * gl[h] + dsc;
  }
}

// ---------------------------------------------------------------------------------------------
// Row softmax with dropout over [rows, cols] score matrices (any cols): the image-level co-attention
// (faster_rcnn_sys_transformer_sk_dilat.py:31-102) runs MultiHeadAttention with 2394 image tokens on one
// side, so its score matrix does not fit the 64x64 register tile of attn.hip; scores and P.V go through
// the batched matrix-core GEMM and this kernel does Modules.py:24 (softmax, dropout) in between.
// One wave per row; y = softmax(x) (kept for the backward), yd = dropout(y) (what multiplies V).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void softmax_rows_fwd_kernel(const float* __restrict__ x, long long rows, int cols,
                                                                    long long ld, float p, unsigned long long seed,
                                                                    float* __restrict__ y, float* __restrict__ yd) {
  const int lane = threadIdx.x & 63;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long r = (long long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6); r < rows;
       r += (long long)gridDim.x * (kThreads / 64)) {
    const float* __restrict__ xr = x + r * ld;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += expf(xr[c] - m);
    const float inv = 1.f / wave_sum(sum);
    for (int c = lane; c < cols; c += 64) {
      const float v = expf(xr[c] - m) * inv;
      y[r * ld + c] = v;
      if (yd != y) yd[r * ld + c] = p > 0.f ? v * drop_scale(seed, (unsigned long long)r * cols + c, p, inv_keep) : v;
    }
  }
}

// dx = y * (dP - sum_c dP*y) with dP = dyd * mask/(1-p)
__global__ __launch_bounds__(kThreads) void softmax_rows_bwd_kernel(const float* __restrict__ dyd, const float* __restrict__ y,
                                                                    long long rows, int cols, long long ld, float p,
                                                                    unsigned long long seed, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long r = (long long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6); r < rows;
       r += (long long)gridDim.x * (kThreads / 64)) {
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) {
      const float d = dyd[r * ld + c] * (p > 0.f ? drop_scale(seed, (unsigned long long)r * cols + c, p, inv_keep) : 1.f);
      dot += d * y[r * ld + c];
    }
    dot = wave_sum(dot);
    for (int c = lane; c < cols; c += 64) {
      const float d = dyd[r * ld + c] * (p > 0.f ? drop_scale(seed, (unsigned long long)r * cols + c, p, inv_keep) : 1.f);
      dx[r * ld + c] = y[r * ld + c] * (d - dot);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Selective heads for ANY sequence length T (the co-attention's 2394-token side): the same arithmetic
// as sh_fwd_kernel / sh_bwd_kernel (SubLayers.py:22-39,92) cut into passes that scale over T:
//   pool   ssum[b,c]  = sum_{h,t} O[b,h,t,c]                       (grid over token chunks, atomics)
//   gate   s = ssum/T; g = sk_w s + sk_b; gate = softmax over heads  (one workgroup per sequence)
//   apply  u[b,t,c]   = sum_h O[b,h,t,c] * gate[b,h,c]
// backward: dgr[b,h,c] = sum_t du[b,t,c] O[b,h,t,c] (atomics); softmax backward -> dg, ds = sk_w^T dg / T;
//           dO[b,h,t,c] = du[b,t,c] gate[b,h,c] + ds[b,c]
// ---------------------------------------------------------------------------------------------
constexpr int kShTok = 64;     // tokens per workgroup in the pool / apply passes

__global__ __launch_bounds__(kThreads) void shg_pool_kernel(const float* __restrict__ O, int T, float* __restrict__ ssum) {
  __shared__ float part[4][kC];
  const int b = blockIdx.y, t0 = blockIdx.x * kShTok, c = threadIdx.x & 63, tq = threadIdx.x >> 6;
  const float* __restrict__ Ob = O + (size_t)b * kH * T * kC;
  float acc = 0.f;
  for (int h = 0; h < kH; h++)
    for (int t = t0 + tq; t < min(T, t0 + kShTok); t += 4) acc += Ob[((size_t)h * T + t) * kC + c];
  part[tq][c] = acc;
  __syncthreads();
  if (threadIdx.x < kC) unsafeAtomicAdd(ssum + (size_t)b * kC + c, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}

__global__ __launch_bounds__(kThreads) void shg_gate_kernel(const float* __restrict__ ssum, float inv_T,
                                                            const float* __restrict__ sk_w, const float* __restrict__ sk_b,
                                                            float* __restrict__ gate_out, float* __restrict__ s_out) {
  __shared__ float s[kC];
  __shared__ float gate[kH * kC];
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid < kC) {
    s[tid] = ssum[(size_t)n * kC + tid] * inv_T;
    s_out[(size_t)n * kC + tid] = s[tid];
  }
  __syncthreads();
  for (int j = tid; j < kH * kC; j += kThreads) {
    float d = sk_b[j];
    for (int q = 0; q < kC; q++) d += sk_w[(size_t)j * kC + q] * s[q];
    gate[j] = d;
  }
  __syncthreads();
  if (tid < kC) {
    float mx = gate[tid];
    for (int h = 1; h < kH; h++) mx = fmaxf(mx, gate[h * kC + tid]);
    float e[kH], sum = 0.f;
    for (int h = 0; h < kH; h++) { e[h] = expf(gate[h * kC + tid] - mx); sum += e[h]; }
    for (int h = 0; h < kH; h++) gate_out[(size_t)n * kH * kC + h * kC + tid] = e[h] / sum;
  }
}

__global__ __launch_bounds__(kThreads) void shg_apply_kernel(const float* __restrict__ O, const float* __restrict__ gate,
                                                             int T, float* __restrict__ u) {
  const int b = blockIdx.y, t0 = blockIdx.x * kShTok, c = threadIdx.x & 63, tq = threadIdx.x >> 6;
  const float* __restrict__ Ob = O + (size_t)b * kH * T * kC;
  float gl[kH];
  for (int h = 0; h < kH; h++) gl[h] = gate[(size_t)b * kH * kC + h * kC + c];
  for (int t = t0 + tq; t < min(T, t0 + kShTok); t += 4) {
    float v = 0.f;
    for (int h = 0; h < kH; h++) v += Ob[((size_t)h * T + t) * kC + c] * gl[h];
    u[((size_t)b * T + t) * kC + c] = v;
  }
}

__global__ __launch_bounds__(kThreads) void shg_bwd_reduce_kernel(const float* __restrict__ du, const float* __restrict__ O,
                                                                  int T, float* __restrict__ dgr) {
  __shared__ float part[4][kH * kC];
  const int b = blockIdx.y, t0 = blockIdx.x * kShTok, c = threadIdx.x & 63, tq = threadIdx.x >> 6;
  const float* __restrict__ Ob = O + (size_t)b * kH * T * kC;
  float acc[kH];
  for (int h = 0; h < kH; h++) acc[h] = 0.f;
  for (int t = t0 + tq; t < min(T, t0 + kShTok); t += 4) {
    const float d = du[((size_t)b * T + t) * kC + c];
    for (int h = 0; h < kH; h++) acc[h] += d * Ob[((size_t)h * T + t) * kC + c];
  }
  for (int h = 0; h < kH; h++) part[tq][h * kC + c] = acc[h];
  __syncthreads();
  for (int j = threadIdx.x; j < kH * kC; j += kThreads)
    unsafeAtomicAdd(dgr + (size_t)b * kH * kC + j, part[0][j] + part[1][j] + part[2][j] + part[3][j]);
}

// per sequence: dg = softmax backward over heads of dgr; ds = sk_w^T dg / T
__global__ __launch_bounds__(kThreads) void shg_bwd_gate_kernel(const float* __restrict__ dgr, const float* __restrict__ gate,
                                                                const float* __restrict__ sk_w, float inv_T,
                                                                float* __restrict__ dg_out, float* __restrict__ ds_out) {
  __shared__ float dg[kH * kC];
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid < kC) {
    float gl[kH], dgt[kH], dot = 0.f;
    for (int h = 0; h < kH; h++) {
      gl[h] = gate[(size_t)n * kH * kC + h * kC + tid];
      dgt[h] = dgr[(size_t)n * kH * kC + h * kC + tid];
      dot += dgt[h] * gl[h];
    }
    for (int h = 0; h < kH; h++) {
      const float v = gl[h] * (dgt[h] - dot);
      dg[h * kC + tid] = v;
      dg_out[(size_t)n * kH * kC + h * kC + tid] = v;
    }
  }
  __syncthreads();
  if (tid < kC) {
    float d = 0.f;
    for (int j = 0; j < kH * kC; j++) d += sk_w[(size_t)j * kC + tid] * dg[j];
    ds_out[(size_t)n * kC + tid] = d * inv_T;
  }
}

__global__ __launch_bounds__(kThreads) void shg_bwd_apply_kernel(const float* __restrict__ du, const float* __restrict__ gate,
                                                                 const float* __restrict__ ds, int T, float* __restrict__ dO) {
  const int b = blockIdx.y, t0 = blockIdx.x * kShTok, c = threadIdx.x & 63, tq = threadIdx.x >> 6;
  float gl[kH];
  for (int h = 0; h < kH; h++) gl[h] = gate[(size_t)b * kH * kC + h * kC + c];
  const float dsc = ds[(size_t)b * kC + c];
  float* __restrict__ dOb = dO + (size_t)b * kH * T * kC;
  for (int t = t0 + tq; t < min(T, t0 + kShTok); t += 4) {
    const float d = du[((size_t)b * T + t) * kC + c];
    for (int h = 0; h < kH; h++) dOb[((size_t)h * T + t) * kC + c] = d * gl[h] + dsc;
  }
}

inline unsigned ln_grid(long long rows) {
  long long b = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
  return (unsigned)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

// the keep / scale decisions of one dropout site, written out (ait_dropout_mask): element j of `out` is what the kernels
// multiply element first + j of that site's tensor by -- 1 / (1 - p) or 0
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned long long seed, unsigned long long first, long long count,
                                                           float p, float* __restrict__ out) {
  const float inv_keep = 1.f / (1.f - p);
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < count; j += (long long)gridDim.x * 256)
    out[j] = p > 0.f ? drop_scale(seed, first + (unsigned long long)j, p, inv_keep) : 1.f;
}

}  // namespace

AIT_API int ait_ln_fwd_rows(const float* a, const float* pos, const float* residual,
                            const float* gamma, const float* beta, long long rows, int d, int seq_len,
                            int src_rows_per_seq, int rep, int out_rows_per_seq, float eps, float p_drop,
                            unsigned long long seed, float* y, float* mean, float* rstd,
                            void* stream) {
  if (rows < 0 || seq_len <= 0 || src_rows_per_seq <= 0 || src_rows_per_seq > seq_len ||
      rep < 1 || p_drop < 0.f || p_drop >= 1.f || out_rows_per_seq <= 0 || out_rows_per_seq > seq_len)
    return AIT_EINVAL;
  if (d != kD) return AIT_EUNSUPPORTED;
  if (rows == 0) return AIT_OK;
  if (!a || !gamma || !beta || !y) return AIT_EINVAL;
  RowMap m{seq_len, src_rows_per_seq, rep, seq_len, out_rows_per_seq};
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(ln_grid(rows)), dim3(kThreads), 0, ait_stream(stream), a,
                     pos, residual, gamma, beta, rows, m, eps, p_drop, seed, y, mean, rstd);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_ln_fwd(const float* a, const float* pos, const float* residual,
                       const float* gamma, const float* beta, long long rows, int d, int seq_len,
                       int src_rows_per_seq, int rep, float eps, float p_drop,
                       unsigned long long seed, float* y, float* mean, float* rstd,
                       void* stream) {
  return ait_ln_fwd_rows(a, pos, residual, gamma, beta, rows, d, seq_len, src_rows_per_seq, rep, seq_len, eps, p_drop, seed, y,
                         mean, rstd, stream);
}

// da16 (library-internal, csrc/transformer.hip): the gradient `da` ALSO / INSTEAD written as bf16 [rows, 512]
int ait_ln_bwd_ex(const float* dy, const float* a, const float* pos, const float* residual,
                  const float* gamma, const float* mean, const float* rstd, long long rows,
                  int d, int seq_len, int src_rows_per_seq, int rep, int dy_rows_per_seq,
                  float p_drop, unsigned long long seed, float* da, float* dres, float* dgamma,
                  float* dbeta, float* dcolsum, void* da16, void* stream) {
  if (rows < 0 || seq_len <= 0 || src_rows_per_seq <= 0 || src_rows_per_seq > seq_len ||
      rep < 1 || p_drop < 0.f || p_drop >= 1.f || dy_rows_per_seq <= 0 || dy_rows_per_seq > seq_len)
    return AIT_EINVAL;
  if (d != kD) return AIT_EUNSUPPORTED;
  if (rows == 0) return AIT_OK;
  if (!dy || !a || !gamma || !mean || !rstd) return AIT_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return AIT_EINVAL;
  RowMap m{seq_len, src_rows_per_seq, rep, dy_rows_per_seq, seq_len};
  // fewer, fatter blocks: each block issues 2*512 atomics for the affine gradients
  long long b = (rows + 63) / 64;
  // (1024 blocks are resident at a time: between one and two rounds of 64-row blocks the second round is a partial one --
  // 1200 blocks at cfg2's 76800 rows ran 13 % slower than 1024 blocks striding over the same rows, 0.184 against 0.160 ms)
  unsigned grid = (unsigned)(b < 1 ? 1 : b <= 1024 ? b : b < 2048 ? 1024 : 2048);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(grid), dim3(kThreads), 0, ait_stream(stream), dy, a, pos,
                     residual, gamma, mean, rstd, rows, m, p_drop, seed, da, dres, dgamma, dbeta, dcolsum,
                     static_cast<unsigned short*>(da16));
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_ln_bwd(const float* dy, const float* a, const float* pos, const float* residual,
                       const float* gamma, const float* mean, const float* rstd, long long rows,
                       int d, int seq_len, int src_rows_per_seq, int rep, int dy_rows_per_seq,
                       float p_drop, unsigned long long seed, float* da, float* dres, float* dgamma,
                       float* dbeta, float* dcolsum, void* stream) {
  return ait_ln_bwd_ex(dy, a, pos, residual, gamma, mean, rstd, rows, d, seq_len, src_rows_per_seq, rep, dy_rows_per_seq, p_drop,
                       seed, da, dres, dgamma, dbeta, dcolsum, nullptr, stream);
}

AIT_API int ait_colsum_f32(const float* x, long long rows, int cols, long long ld, float* out, void* stream) {
  if (rows < 0 || cols < 0 || (cols & 3) || ld < cols || (ld & 3)) return AIT_EINVAL;
  if (rows == 0 || cols == 0) return AIT_OK;
  if (!x || !out || (reinterpret_cast<uintptr_t>(x) & 15)) return AIT_EINVAL;
  const int groups = cols / 4;
  int G = 1;
  while (G < groups && G < 256) G *= 2;                 // column quads per block: a power of two <= 256 (divides 1024)
  const unsigned gy = (unsigned)((groups + G - 1) / G);
  const int RY = 1024 / G;
  const long long want_blocks = 512 / gy > 0 ? 512 / gy : 1;
  long long band = (rows + want_blocks - 1) / want_blocks;
  const long long unit = (long long)RY * 8;             // whole unrolled passes of the block's RY row lanes
  band = (band + unit - 1) / unit * unit;
  const long long bx = (rows + band - 1) / band;
  if (band > 0x7fffffffLL || bx > 0x7fffffffLL) return AIT_EUNSUPPORTED;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)bx, gy), dim3(1024), (size_t)1024 * 4 * sizeof(float), ait_stream(stream), x, rows,
                     cols, ld, (int)band, G, out);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_rep_sum_f32(const float* x, int groups, int rep, long long E, float* out, void* stream) {
  if (groups < 0 || rep < 1 || E < 0 || (E & 3)) return AIT_EINVAL;
  if (groups == 0 || E == 0) return AIT_OK;
  if (!x || !out || (reinterpret_cast<uintptr_t>(x) & 15)) return AIT_EINVAL;
  hipLaunchKernelGGL(rep_sum_kernel, dim3((unsigned)((E / 4 + 255) / 256), (unsigned)groups,
                                          (unsigned)((rep + kRepSlice - 1) / kRepSlice)),
                     dim3(256), 0, ait_stream(stream), x, rep, E, out);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_sh_fwd(const float* O, const float* sk_w, const float* sk_b, int n_seq, int H,
                       int T, int dv, float* u, float* gate, float* s, void* stream) {
  if (n_seq < 0) return AIT_EINVAL;
  if (H != kH || T != kT || dv != kC) return AIT_EUNSUPPORTED;
  if (n_seq == 0) return AIT_OK;
  if (!O || !sk_w || !sk_b || !u) return AIT_EINVAL;
  hipLaunchKernelGGL(sh_fwd_kernel, dim3(n_seq), dim3(kThreads), 0, ait_stream(stream), O, sk_w,
                     sk_b, u, gate, s);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_sh_bwd(const float* du, const float* O, const float* gate, const float* sk_w,
                       int n_seq, int H, int T, int dv, float* dO, float* dg, void* stream) {
  if (n_seq < 0) return AIT_EINVAL;
  if (H != kH || T != kT || dv != kC) return AIT_EUNSUPPORTED;
  if (n_seq == 0) return AIT_OK;
  if (!du || !O || !gate || !sk_w || !dO || !dg) return AIT_EINVAL;
  hipLaunchKernelGGL(sh_bwd_kernel, dim3(n_seq), dim3(kThreads), 0, ait_stream(stream), du, O, gate,
                     sk_w, dO, dg);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_dropout_mask(unsigned long long site_seed, unsigned long long first_index, long long count, float p_drop,
                             float* scale, void* stream) {
  if (count < 0 || p_drop < 0.f || p_drop >= 1.f) return AIT_EINVAL;
  if (count == 0) return AIT_OK;
  if (!scale) return AIT_EINVAL;
  const long long b = (count + 255) / 256;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)(b > 16384 ? 16384 : b)), dim3(256), 0, ait_stream(stream), site_seed,
                     first_index, count, p_drop, scale);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_softmax_rows_fwd(const float* x, long long rows, int cols, long long ld, float p_drop,
                                 unsigned long long seed, float* y, float* y_drop, void* stream) {
  if (rows < 0 || cols <= 0 || ld < cols || p_drop < 0.f || p_drop >= 1.f) return AIT_EINVAL;
  if (rows == 0) return AIT_OK;
  if (!x || !y || !y_drop) return AIT_EINVAL;
  const long long b = (rows + 3) / 4;
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(kThreads), 0, ait_stream(stream),
                     x, rows, cols, ld, p_drop, seed, y, y_drop);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_softmax_rows_bwd(const float* dy_drop, const float* y, long long rows, int cols, long long ld, float p_drop,
                                 unsigned long long seed, float* dx, void* stream) {
  if (rows < 0 || cols <= 0 || ld < cols || p_drop < 0.f || p_drop >= 1.f) return AIT_EINVAL;
  if (rows == 0) return AIT_OK;
  if (!dy_drop || !y || !dx) return AIT_EINVAL;
  const long long b = (rows + 3) / 4;
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)(b > 8192 ? 8192 : b)), dim3(kThreads), 0, ait_stream(stream),
                     dy_drop, y, rows, cols, ld, p_drop, seed, dx);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_sh_general_fwd(const float* O, const float* sk_w, const float* sk_b, int n_seq, int H, int T, int dv,
                               float* u, float* gate, float* s, void* stream) {
  if (n_seq < 0 || T <= 0) return AIT_EINVAL;
  if (H != kH || dv != kC || n_seq > 65535) return AIT_EUNSUPPORTED;
  if (n_seq == 0) return AIT_OK;
  if (!O || !sk_w || !sk_b || !u || !gate || !s) return AIT_EINVAL;
  hipStream_t st = ait_stream(stream);
  // `s` doubles as the accumulator of the pooled sums before the gate pass rescales it in place
  if (hipMemsetAsync(s, 0, sizeof(float) * (size_t)n_seq * kC, st) != hipSuccess) return AIT_ELAUNCH;
  const dim3 grid((unsigned)((T + kShTok - 1) / kShTok), (unsigned)n_seq);
  hipLaunchKernelGGL(shg_pool_kernel, grid, dim3(kThreads), 0, st, O, T, s);
  AIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(shg_gate_kernel, dim3(n_seq), dim3(kThreads), 0, st, s, 1.f / (float)T, sk_w, sk_b, gate, s);
  AIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(shg_apply_kernel, grid, dim3(kThreads), 0, st, O, gate, T, u);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}

AIT_API int ait_sh_general_bwd(const float* du, const float* O, const float* gate, const float* sk_w, int n_seq, int H,
                               int T, int dv, float* dO, float* dg, float* workspace, void* stream) {
  if (n_seq < 0 || T <= 0) return AIT_EINVAL;
  if (H != kH || dv != kC || n_seq > 65535) return AIT_EUNSUPPORTED;
  if (n_seq == 0) return AIT_OK;
  if (!du || !O || !gate || !sk_w || !dO || !dg || !workspace) return AIT_EINVAL;
  hipStream_t st = ait_stream(stream);
  float* dgr = workspace;                            // [n_seq, H*dv]
  float* ds = workspace + (size_t)n_seq * kH * kC;   // [n_seq, dv]
  if (hipMemsetAsync(dgr, 0, sizeof(float) * (size_t)n_seq * kH * kC, st) != hipSuccess) return AIT_ELAUNCH;
  const dim3 grid((unsigned)((T + kShTok - 1) / kShTok), (unsigned)n_seq);
  hipLaunchKernelGGL(shg_bwd_reduce_kernel, grid, dim3(kThreads), 0, st, du, O, T, dgr);
  AIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(shg_bwd_gate_kernel, dim3(n_seq), dim3(kThreads), 0, st, dgr, gate, sk_w, 1.f / (float)T, dg, ds);
  AIT_CHECK_LAUNCH();
  hipLaunchKernelGGL(shg_bwd_apply_kernel, grid, dim3(kThreads), 0, st, du, gate, ds, T, dO);
  AIT_CHECK_LAUNCH();
  return AIT_OK;
}
