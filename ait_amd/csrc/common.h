// ait_amd/csrc/common.h -- shared helpers for the gfx950 kernels of libait_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ait_hip.h"
#include "lab_knobs.h"

#define AIT_WAVE 64
#define AIT_API extern "C" __attribute__((visibility("default")))

#define AIT_CHECK_LAUNCH()                         \
  do {                                             \
    if (hipGetLastError() != hipSuccess) return AIT_ELAUNCH; \
  } while (0)

static inline hipStream_t ait_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define AIT_TRY_RC(expr)               \
  do {                                 \
    const int rc_try__ = (expr);       \
    if (rc_try__ != AIT_OK) return rc_try__; \
  } while (0)

// ---- measurement probe (include/ait_hip.h "Measurement"): an instrumented entry point that is handed a
// probe (ait_launch_ctx::probe) brackets its launch with a HIP event pair on the launch stream and notes the
// algorithmic work.  No probe (the default): one pointer test.  The library keeps no probe of its own.
struct AitProbeEntry {
  int kind;
  double work;            // flops (AIT_PROBE_GEMM) or algorithmic bytes
  int dims[6];
  hipEvent_t e0, e1;
};
struct AitProbe {
  int cap;
  int device;             // the device its events belong to: launches on another device are not recorded
  volatile int n;         // claimed with an atomic add: launches may come from several host threads; counts past
                          // `cap` too (ait_probe_count reports the overflow)
  AitProbeEntry* e;
};
static inline AitProbe* ait_probe_of(const ait_launch_ctx* ctx) {
  return ctx ? static_cast<AitProbe*>(ctx->probe) : nullptr;
}
struct AitProbeScope {
  AitProbeEntry* ent = nullptr;
  hipStream_t s;
  AitProbeScope(AitProbe* p, int kind, double work, hipStream_t stream, int d0 = 0, int d1 = 0, int d2 = 0, int d3 = 0,
                int d4 = 0, int d5 = 0)
      : s(stream) {
    if (!p) return;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != p->device) return;
    const int slot = __atomic_fetch_add(const_cast<int*>(&p->n), 1, __ATOMIC_RELAXED);
    if (slot >= p->cap) return;
    ent = &p->e[slot];
    ent->kind = kind; ent->work = work;
    ent->dims[0] = d0; ent->dims[1] = d1; ent->dims[2] = d2; ent->dims[3] = d3; ent->dims[4] = d4; ent->dims[5] = d5;
    (void)hipEventRecord(ent->e0, s);
  }
  ~AitProbeScope() {
    if (ent) (void)hipEventRecord(ent->e1, s);
  }
};

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
