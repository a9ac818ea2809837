// Attainable fp32 matrix-pipe rate on this chip: a register-only v_mfma_f32_32x32x2_f32 loop
// (4 independent accumulators per wave, no LDS / global traffic), W waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak.hip -o /tmp/mfma_peak ; run: /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; r++) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// The bf16 pipe under load: v_mfma_f32_32x32x16_bf16 on RANDOM operand bits (the chip lowers its clock under matrix load
// and holds a higher one on zeros or constants: MI355X_MICROARCH.md "DVFS give-back"), register-only, 4 accumulators.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256) void kb(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  u32x4 ux, uy;
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  for (int j = 0; j < 4; j++) {
    // random sign and mantissa, exponents within [2^-2, 2^1]: finite sums
    ux[j] = (hash32(t * 8 + j) & 0x807f807fu) | 0x3f003f00u;
    uy[j] = (hash32(t * 8 + 4 + j) & 0x807f807fu) | 0x3f003f00u;
  }
  const bf16x8 x = __builtin_bit_cast(bf16x8, ux), y = __builtin_bit_cast(bf16x8, uy);
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; r++) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD
    const int blocks = 256 * wps, iters = 20000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1000);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)blocks * 4 /*waves*/ * iters * 32.0 * 4096.0;
      printf("waves/SIMD %d: %.1f ms  %.1f TFLOP/s (fp32 MFMA, register-only)\n", wps, ms, fl / ms / 1e9);
    }
  }
  for (int wps = 1; wps <= 2; wps *= 2) {
    const int blocks = 256 * wps, iters = 100000 / wps;
    hipLaunchKernelGGL(kb, dim3(blocks), dim3(256), 0, 0, d, 2000);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 4; rep++) {         // (the first window runs at the idle chip's clock; the later ones under load)
      hipEventRecord(e0);
      hipLaunchKernelGGL(kb, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)blocks * 4 /*waves*/ * iters * 32.0 * 32768.0;
      printf("waves/SIMD %d: %.1f ms  %.1f TFLOP/s (bf16 MFMA 32x32x16, register-only, random operands)\n", wps, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
