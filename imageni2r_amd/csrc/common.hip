// Error reporting and version for libi2r_hip.so.
#include "i2r_common.h"
#include "../../include/i2r_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void i2r_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Launch errors only (hipGetLastError does not synchronise, so this is safe under stream capture).
int i2r_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    i2r_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return I2R_ERR_HIP;
  }
  return I2R_OK;
}

extern "C" int i2r_version(void) { return 100; }
extern "C" const char* i2r_last_error(void) { return g_err; }
