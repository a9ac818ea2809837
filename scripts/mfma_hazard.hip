// scripts/mfma_hazard.hip -- does a DEPENDENT v_mfma_f32_32x32x16_bf16 (SrcC = the previous one's vDst) always see
// its predecessor's complete result?  hipcc (ROCm 7.2) pads MFMA dependencies itself; this harness checks the padding
// it chooses against the hardware, for the schedules the f32-split GEMM could use (gemm_f32_impl.h, KNOB_SPLIT).
// (Written while hunting non-reproducible row segments in that GEMM: every pattern here is clean -- the cause was a
// packed v_pk_fma_f32 in the epilogue, DESIGN.md 3.1 -- and the schedules below are all safe to use.)
//   pattern 0: the chain back to back (nothing between dependent MFMAs)
//   pattern 1: one independent MFMA (another accumulator) between dependent ones      -- two chains alternating
//   pattern 2: two independent MFMAs between dependent ones                            -- three chains rotating
//   pattern 3: three independent MFMAs between                                        -- four chains rotating
//   pattern 10 + n: n independent v_add_f32 between dependent MFMAs (one chain)
//   pattern 30: one ds_read_b128 (into registers nothing else uses) between dependent MFMAs (one chain)
//   pattern 31: the same ds_read_b128s, all issued in front of the chain
// Operands are small integers (exact in bf16, sums exact in f32): every schedule must produce the same bits; a
// difference is a dependent MFMA that read part of its SrcC before the predecessor's write-back landed.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_hazard.hip -o scripts/_mfma_hazard && scripts/_mfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ bf16x8 operand(unsigned seed, int i) {
  bf16x8 v;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    seed = seed * 1664525u + 1013904223u + (unsigned)i;
    v[q] = (__bf16)(float)((int)((seed >> 24) & 7) - 3);      // -3 .. 4
  }
  return v;
}

template <int PATTERN>
__global__ __launch_bounds__(256, 2) void chain_kernel(int steps, float* __restrict__ out, float* __restrict__ sink) {
  const unsigned id = blockIdx.x * 256 + threadIdx.x;
  constexpr int NCH = PATTERN == 0 ? 1 : PATTERN <= 3 ? PATTERN + 1 : 1;
  constexpr int NFILL = (PATTERN >= 10 && PATTERN < 30) ? PATTERN - 10 : 0;
  __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 6];
  for (int i = threadIdx.x; i < 256 * 4 * 6; i += 256) lds[i] = (float)i;
  __syncthreads();
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 dsv[6];
#pragma unroll
  for (int q = 0; q < 6; q++) dsv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned dsaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + threadIdx.x * 4);
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; c++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  float f[12];
#pragma unroll
  for (int q = 0; q < 12; q++) f[q] = (float)(id & 15) + q;
  for (int s = 0; s < steps; s++) {
    // every chain receives the same operand sequence, so all chains must end with the same accumulator
    const bf16x8 a = operand(id * 2654435761u, s), b = operand(id * 40503u + 7u, s);
    SB();
    if constexpr (PATTERN == 31) {
#pragma unroll
      for (int q = 0; q < 6; q++) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dsv[q]) : "v"(dsaddr), "n"(0)); SB(); }
    }
#pragma unroll
    for (int rep = 0; rep < 6; rep++) {
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
        SB();
        if constexpr (PATTERN == 30) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dsv[rep]) : "v"(dsaddr), "n"(0)); SB(); }
#pragma unroll
        for (int q = 0; q < NFILL; q++) { asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(f[q])); SB(); }
      }
    }
  }
  float t = 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int q = 0; q < 6; q++) t += dsv[q][0] * 0.f;
#pragma unroll
  for (int q = 0; q < 12; q++) t += f[q];
  if (t == -1.f) sink[0] = t;
  // chain 0's accumulator; for the multi-chain patterns every chain must equal chain 0
  float bad = 0.f;
#pragma unroll
  for (int c = 1; c < NCH; c++)
#pragma unroll
    for (int r = 0; r < 16; r++) bad += acc[c][r] != acc[0][r] ? 1.f : 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) out[(size_t)id * 17 + r] = acc[0][r];
  out[(size_t)id * 17 + 16] = bad;
}

// WAR: the A operand's registers are overwritten by v_mov_b32 right behind the MFMA that reads them (PAD = 0) or
// after PAD x `s_nop 15` (the reference).  The chain alternates two operand values, so an MFMA that samples part of
// its SrcA after the overwrite accumulates the wrong product.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int PAD, int DEP>
__global__ __launch_bounds__(256, 2) void war_kernel(int steps, float* __restrict__ out) {
  const unsigned id = blockIdx.x * 256 + threadIdx.x;
  f32x16 acc[2];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
  const bf16x8 a0 = operand(id * 2654435761u, 1), a1 = operand(id * 2654435761u, 2), b = operand(id * 40503u + 7u, 3);
  const u32x4 u0 = __builtin_bit_cast(u32x4, a0), u1 = __builtin_bit_cast(u32x4, a1);
  u32x4 x = u0;
  for (int s = 0; s < steps; s++) {
#pragma unroll
    for (int rep = 0; rep < 6; rep++) {
      SB();
      // DEP = 1: one dependent chain (an MFMA queues behind its predecessor); DEP = 0: two accumulators alternate
      f32x16& d = acc[DEP ? 0 : (rep & 1)];
      d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), b, d, 0, 0, 0);
      SB();
#pragma unroll
      for (int p = 0; p < PAD; p++) asm volatile("s_nop 15");
      const u32x4 nx = (rep & 1) ? u0 : u1;
      unsigned t0 = x[0], t1 = x[1], t2 = x[2], t3 = x[3];
      asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                   : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(nx[0]), "v"(nx[1]), "v"(nx[2]), "v"(nx[3]));
      x[0] = t0; x[1] = t1; x[2] = t2; x[3] = t3;
      SB();
    }
  }
#pragma unroll
  for (int r = 0; r < 16; r++) out[(size_t)id * 17 + r] = acc[0][r] + acc[1][r];
  out[(size_t)id * 17 + 16] = 0.f;
}

template <int PAD, int DEP>
static void run_war(int steps, int blocks, std::vector<float>* keep, const std::vector<float>& ref, float* d_out) {
  hipLaunchKernelGGL((war_kernel<PAD, DEP>), dim3(blocks), dim3(256), 0, 0, steps, d_out);
  hipDeviceSynchronize();
  std::vector<float> h((size_t)blocks * 256 * 17);
  hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
  if (keep) { *keep = h; return; }
  long diff = 0;
  int lanes[4] = {0, 0, 0, 0};
  for (size_t t = 0; t < (size_t)blocks * 256; t++)
    for (int r = 0; r < 16; r++)
      if (h[t * 17 + r] != ref[t * 17 + r]) { diff++; lanes[(t & 63) / 16]++; }
  printf("SrcA overwritten right behind the MFMA (%s): %ld accumulator registers differ from the padded run (by lane quarter: %d %d %d %d)\n",
         DEP ? "one dependent chain" : "two chains alternating", diff, lanes[0], lanes[1], lanes[2], lanes[3]);
}

template <int P>
static void run(int steps, int blocks, const std::vector<float>& ref, std::vector<float>* keep, float* d_out, float* d_sink) {
  hipLaunchKernelGGL(chain_kernel<P>, dim3(blocks), dim3(256), 0, 0, steps, d_out, d_sink);
  hipDeviceSynchronize();
  std::vector<float> h((size_t)blocks * 256 * 17);
  hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
  if (keep) { *keep = h; printf("pattern %2d: reference\n", P); return; }
  long diff = 0, cross = 0;
  int lanes[4] = {0, 0, 0, 0};
  for (size_t t = 0; t < (size_t)blocks * 256; t++) {
    for (int r = 0; r < 16; r++)
      if (h[t * 17 + r] != ref[t * 17 + r]) { diff++; lanes[(t & 63) / 16]++; }
    cross += (long)h[t * 17 + 16];
  }
  printf("pattern %2d: %ld accumulator registers differ from the back-to-back chain (by lane quarter: %d %d %d %d); "
         "%ld differ between chains of one wave\n", P, diff, lanes[0], lanes[1], lanes[2], lanes[3], cross);
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 200, blocks = 512 * 8;
  float *d_out, *d_sink;
  hipMalloc(&d_out, (size_t)blocks * 256 * 17 * 4);
  hipMalloc(&d_sink, 4);
  std::vector<float> ref;
  run<0>(steps, blocks, ref, &ref, d_out, d_sink);
  for (int rep = 0; rep < 2; rep++) {
    run<0>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<1>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<2>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<3>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<11>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<12>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<14>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<17>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<22>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<30>(steps, blocks, ref, nullptr, d_out, d_sink);
    run<31>(steps, blocks, ref, nullptr, d_out, d_sink);
  }
  std::vector<float> wref;
  run_war<4, 1>(steps, blocks, &wref, wref, d_out);
  for (int rep = 0; rep < 3; rep++) run_war<0, 1>(steps, blocks, nullptr, wref, d_out);
  run_war<4, 0>(steps, blocks, &wref, wref, d_out);
  for (int rep = 0; rep < 3; rep++) run_war<0, 0>(steps, blocks, nullptr, wref, d_out);
  return 0;
}
