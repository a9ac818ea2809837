// ait_amd/csrc/abi.hip -- ABI version + error strings of libait_hip.so.
#include "common.h"

#include <atomic>
#include <new>



AIT_API int ait_abi_version(void) { return 2; }

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
namespace {
// process-wide (not per thread): PyTorch runs the backward on its autograd engine's threads
std::atomic<AitProbe*> g_probe{nullptr};
}
AitProbe* ait_probe_current() { return g_probe.load(std::memory_order_acquire); }

AIT_API void* ait_probe_create(int capacity) {
  if (capacity <= 0) return nullptr;
  AitProbe* p = new (std::nothrow) AitProbe{capacity, 0, nullptr};
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
  if (g_probe.load() == p) g_probe.store(nullptr);
  for (int i = 0; i < p->cap; i++) { (void)hipEventDestroy(p->e[i].e0); (void)hipEventDestroy(p->e[i].e1); }
  delete[] p->e;
  delete p;
}

AIT_API void ait_probe_attach(void* probe) { g_probe.store(static_cast<AitProbe*>(probe), std::memory_order_release); }

AIT_API int ait_probe_reset(void* probe) {
  AitProbe* p = static_cast<AitProbe*>(probe);
  if (!p) return AIT_EINVAL;
  p->n = 0;
  return AIT_OK;
}

AIT_API int ait_probe_count(void* probe) {
  if (!probe) return 0;
  const AitProbe* p = static_cast<AitProbe*>(probe);
  return p->n < p->cap ? p->n : p->cap;
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
