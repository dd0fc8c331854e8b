// ABI version + status strings.
#include "tt_common.h"

extern "C" int tt_abi_version(void) { return TT_ABI_VERSION; }

extern "C" const char* tt_status_string(int status) {
  switch (status) {
    case TT_OK: return "ok";
    case TT_ERR_NULL_POINTER: return "null pointer argument";
    case TT_ERR_BAD_SHAPE: return "bad shape / size argument (or out-of-range token id)";
    case TT_ERR_BAD_DTYPE: return "unsupported integer width (TT_I32 / TT_I64 expected)";
    case TT_ERR_ALIGNMENT: return "pointer or stride not 16-byte aligned";
    case TT_ERR_WORKSPACE: return "workspace too small / allocation failed";
    case TT_ERR_BAD_HANDLE: return "invalid engine handle";
    case TT_ERR_UNSUPPORTED: return "unsupported configuration";
    case TT_ERR_WRONG_DEVICE: return "current HIP device / workspace device is not the engine config's device";
    default: break;
  }
  if (status > 0) return hipGetErrorString(static_cast<hipError_t>(status));
  return "unknown status";
}
