// Library identity and error strings for libgsage.so.
#include "gsage_common.h"

extern "C" int gsage_version(void) { return 100; }  // 0.1.0

extern "C" const char *gsage_error_string(int status) {
  switch (status) {
    case GSAGE_OK: return "ok";
    case GSAGE_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or bad enum)";
    case GSAGE_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case GSAGE_ERR_WORKSPACE: return "workspace too small";
    case GSAGE_ERR_LAUNCH: return "HIP launch failed";
    case GSAGE_ERR_TOO_LARGE: return "size beyond the fixed scratch of a no-alloc entry point";
    default: return "unknown gsage status";
  }
}
