// ait_amd/csrc/abi.hip -- ABI version + error strings of libait_hip.so.
#include "common.h"

#include <new>



AIT_API int ait_abi_version(void) { return 8; }
// 0 for the shipped library; 1 for a lab variant built with experiment knobs (csrc/lab_knobs.h)
AIT_API int ait_lab_build(void) { return ait_lab::kLabBuild; }

AIT_API const char* ait_strerror(int code) {
  switch (code) {
    case AIT_OK: return "ok";
    case AIT_EINVAL: return "invalid argument";
    case AIT_EWORKSPACE: return "workspace too small";
    case AIT_ELAUNCH: return "HIP launch / memset failed";
    case AIT_EUNSUPPORTED: return "unsupported shape";
    default: return "unknown error";
  }
}

// ---- measurement probe --------------------------------------------------------------------------------
// A probe is a caller-owned object handed to the instrumented entry points through ait_launch_ctx::probe; the
// library holds no pointer to it between calls.  The caller synchronises the device before it resets, reads
// or destroys a probe that launches still in flight were given.
AIT_API void* ait_probe_create(int capacity) {
  if (capacity <= 0) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  AitProbe* p = new (std::nothrow) AitProbe{capacity, dev, 0, nullptr};
  if (!p) return nullptr;
  p->e = new (std::nothrow) AitProbeEntry[capacity];
  if (!p->e) { delete p; return nullptr; }
  for (int i = 0; i < capacity; i++) {
    // timing-only events: no system-scope release/acquire fence at the record (the default flavour writes
    // back and invalidates the caches around the launch it brackets, which slows what is being measured)
    if (hipEventCreateWithFlags(&p->e[i].e0, hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&p->e[i].e1, hipEventDisableSystemFence) != hipSuccess) {
      p->cap = i;          // what was created is what can be used
      break;
    }
  }
  return p;
}

AIT_API void ait_probe_destroy(void* probe) {
  AitProbe* p = static_cast<AitProbe*>(probe);
  if (!p) return;
  for (int i = 0; i < p->cap; i++) { (void)hipEventDestroy(p->e[i].e0); (void)hipEventDestroy(p->e[i].e1); }
  delete[] p->e;
  delete p;
}

AIT_API int ait_probe_reset(void* probe) {
  AitProbe* p = static_cast<AitProbe*>(probe);
  if (!p) return AIT_EINVAL;
  p->n = 0;
  return AIT_OK;
}

// launches the probe was handed since the last reset -- MORE than its capacity when it overflowed (only the
// first `capacity` of them were recorded: ait_probe_get fails beyond)
AIT_API int ait_probe_count(void* probe) {
  if (!probe) return 0;
  return static_cast<AitProbe*>(probe)->n;
}

AIT_API int ait_probe_capacity(void* probe) {
  if (!probe) return 0;
  return static_cast<AitProbe*>(probe)->cap;
}

AIT_API int ait_probe_get(void* probe, int i, int* kind, double* work, float* ms, int* dims6) {
  AitProbe* p = static_cast<AitProbe*>(probe);
  if (!p || i < 0 || i >= p->n || i >= p->cap) return AIT_EINVAL;
  const AitProbeEntry& e = p->e[i];
  float t = 0.f;
  if (hipEventElapsedTime(&t, e.e0, e.e1) != hipSuccess) return AIT_ELAUNCH;     // not yet complete: synchronise first
  if (kind) *kind = e.kind;
  if (work) *work = e.work;
  if (ms) *ms = t;
  if (dims6) for (int k = 0; k < 6; k++) dims6[k] = e.dims[k];
  return AIT_OK;
}
