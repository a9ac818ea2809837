// ait_amd/csrc/common.h -- shared helpers for the gfx950 kernels of libait_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ait_hip.h"

#define AIT_WAVE 64
#define AIT_API extern "C" __attribute__((visibility("default")))

#define AIT_CHECK_LAUNCH()                         \
  do {                                             \
    if (hipGetLastError() != hipSuccess) return AIT_ELAUNCH; \
  } while (0)

static inline hipStream_t ait_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// MI355X dispatches workgroups round-robin over its 8 XCDs (blocks b and b+8 share an L2).
// `xcd_major` turns a linear block id into (xcd_slot, index-within-slot) so that all blocks
// that share one XCD's L2 can be given work that shares operands.  Speed only, never
// correctness (cdna_hip_programming.md T1).
#define AIT_NXCD 8

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
