// scripts/mfma_round.hip -- LAB: how v_mfma_f32_32x32x16_bf16 (and v_mfma_f32_32x32x2_f32) round C + sum_k a_k b_k.
// Every lane gets the same operands: a = 2^ea in all 8 k-slots (so 16 identical products per output), C = c0.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_round.hip -o scripts/_mfma_round
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void kb(float a, float b, float c0, int nk, float* out) {
  bf16x8 x, y;
  for (int j = 0; j < 8; j++) { x[j] = (__bf16)(j < nk ? a : 0.f); y[j] = (__bf16)b; }
  f32x16 c;
  for (int r = 0; r < 16; r++) c[r] = c0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
__global__ void kf(float a, float b, float c0, float* out) {
  f32x16 c;
  for (int r = 0; r < 16; r++) c[r] = c0;
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  auto runb = [&](float a, float b, float c0, int nk) { hipLaunchKernelGGL(kb, dim3(1), dim3(64), 0, 0, a, b, c0, nk, d); float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); return h; };
  auto runf = [&](float a, float b, float c0) { hipLaunchKernelGGL(kf, dim3(1), dim3(64), 0, 0, a, b, c0, d); float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); return h; };
  const float ulp = ldexpf(1.f, -23);
  printf("C = 1.0 (ulp 2^-23); bf16 MFMA with n products of p each (result - 1 in ulps; exact sum in ulps)\n");
  struct { float p_ulps; int nk; } cases[] = {{0.125f, 8}, {0.25f, 8}, {0.09375f, 8}, {0.75f, 1}, {0.5f, 1}, {1.5f, 1}, {0.375f, 4}, {0.625f, 4}, {0.4375f, 8}, {0.3125f, 8}};
  for (auto cs : cases) {
    // product p = cs.p_ulps * ulp = a * b with a, b powers of two times small integers (bf16-exact)
    const float a = cs.p_ulps * 8.f, b = ulp / 8.f;
    const float r = runb(a, b, 1.0f, cs.nk), rn = runb(-a, b, 1.0f, cs.nk), rneg = runb(a, b, -1.0f, cs.nk);
    // lanes 0-31 hold k 0..7, lanes 32-63 k 8..15: every output sums 2 * nk products
    printf("  p = %.5f ulp x %2d products (exact %+.4f ulp): C=+1: %+.2f ulp   negative products: %+.2f ulp   C=-1, positive products: %+.2f ulp\n", cs.p_ulps, 2 * cs.nk,
           cs.p_ulps * 2 * cs.nk, (r - 1.0f) / ulp, (rn - 1.0f) / ulp, (rneg + 1.0f) / ulp);
  }
  printf("f32 MFMA 32x32x2 (2 products):\n");
  for (float pu : {0.25f, 0.375f, 0.75f, 0.3125f}) {
    const float r = runf(pu * 8.f, ulp / 8.f, 1.0f), rn = runf(-pu * 8.f, ulp / 8.f, 1.0f);
    printf("  p = %.4f ulp x 2 (exact %+.4f): %+.2f ulp   negative: %+.2f ulp\n", pu, 2 * pu, (r - 1.0f) / ulp, (rn - 1.0f) / ulp);
  }
  return 0;
}
