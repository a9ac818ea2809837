// scripts/pk_hazard.hip -- LAB: is a packed-f32 instruction that takes an operand half through op_sel safe right behind the
// VALU instruction that wrote that half?  (ait_amd/build.py compiles the library with -fno-slp-vectorize because hipcc's
// SLP vectoriser turned the GEMM's bias epilogue into exactly this sequence
//      v_mov_b32    v117, <bias value>
//      v_pk_fma_f32 v[114:115], s[44:45], v[96:97], v[116:117] op_sel:[0,0,1]        (both result halves + v117)
// and the result was, on some launches only, wrong in lanes 48-63 by the difference of two bias values.)
// Every wave repeats that pair with a new value per iteration (the register's previous content is the previous iteration's
// value, so a stale read is visible), with 0, 1 or 2 wait states between the two instructions, and counts wrong results per
// lane quarter.   hipcc -O3 --offload-arch=gfx950 scripts/pk_hazard.hip -o scripts/_pk_hazard && scripts/_pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define PAIR(NOPS)                                                                                       \
  asm volatile("v_mov_b32 v100, %[lo]\n\t"                                                              \
               "v_mov_b32 v104, %[a0]\n\t"                                                              \
               "v_mov_b32 v105, %[a1]\n\t"                                                              \
               "s_mov_b32 s20, %[al]\n\t"                                                               \
               "s_mov_b32 s21, %[al]\n\t"                                                               \
               "s_nop 4\n\t"                                                                            \
               "v_mov_b32 v101, %[b]\n\t" NOPS                                                          \
               "v_pk_fma_f32 v[102:103], s[20:21], v[104:105], v[100:101] op_sel:[0,0,1]\n\t"           \
               "s_nop 4\n\t"                                                                            \
               "v_mov_b32 %[r0], v102\n\t"                                                              \
               "v_mov_b32 %[r1], v103\n\t"                                                              \
               : [r0] "=v"(r0), [r1] "=v"(r1)                                                           \
               : [lo] "v"(lo), [a0] "v"(a0), [a1] "v"(a1), [al] "s"(alpha), [b] "v"(b)                  \
               : "v100", "v101", "v102", "v103", "v104", "v105", "s20", "s21")

template <int NOPS, bool BUSY>
__global__ void probe(const float* __restrict__ src, int iters, unsigned* __restrict__ bad, float alpha) {
  const int lane = threadIdx.x & 63;
  const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  unsigned wrong = 0;
  float acc_busy = 0.f;
  for (int it = 0; it < iters; it++) {
    // (a fresh value per lane and iteration, loaded like the epilogue's bias: the pair follows an s_waitcnt)
    const float b = src[(base * 7 + (size_t)it * 8191) & ((1u << 22) - 1)];
    const float lo = b * 3.f + 1.f, a0 = b + 0.5f, a1 = b - 0.25f;
    float r0, r1;
    if (NOPS == 0) PAIR("");
    else if (NOPS == 1) PAIR("s_nop 0\n\t");
    else PAIR("s_nop 1\n\t");
    const float e0 = __builtin_fmaf(alpha, a0, b), e1 = __builtin_fmaf(alpha, a1, b);
    if (r0 != e0 || r1 != e1) wrong++;
    if (BUSY) acc_busy += __expf(r0) * 1e-30f;      // other work on the vector pipe between the pairs
  }
  if (BUSY && acc_busy == 12345.f) wrong += 1u << 30;
  if (wrong) atomicAdd(bad + (lane >> 4), wrong);
}

int main() {
  const size_t n = 1u << 22;
  float* h = (float*)malloc(n * 4);
  srand(5);
  for (size_t i = 0; i < n; i++) h[i] = (float)(rand() % 20001 - 10000) / 64.f;
  float* d;
  unsigned* bad;
  hipMalloc(&d, n * 4);
  hipMalloc(&bad, 16);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  const int iters = 2000, blocks = 2048, threads = 512;
  auto run = [&](const char* name, auto kern) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemset(bad, 0, 16);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, bad, 1.25f);
      unsigned r[4];
      hipMemcpy(r, bad, 16, hipMemcpyDeviceToHost);
      printf("%-44s launch %d: wrong results by lane quarter (0-15 16-31 32-47 48-63): %u %u %u %u  of %.3g pairs per quarter\n", name, rep,
             r[0], r[1], r[2], r[3], (double)iters * blocks * threads / 4);
    }
  };
  run("v_mov -> v_pk_fma op_sel, back to back", probe<0, false>);
  run("v_mov -> v_pk_fma op_sel, back to back, busy", probe<0, true>);
  run("v_mov -> s_nop 0 -> v_pk_fma op_sel", probe<1, false>);
  run("v_mov -> s_nop 1 -> v_pk_fma op_sel", probe<2, false>);
  return 0;
}
