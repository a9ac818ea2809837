// ait_amd/csrc/abi.hip -- ABI version + error strings of libait_hip.so.
#include "common.h"



AIT_API int ait_abi_version(void) { return 1; }

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
