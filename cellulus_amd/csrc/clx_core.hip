// Error reporting, version and device probing for libclx.
#include "clx_common.h"

static thread_local char g_err[512] = "";

void clx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* clx_last_error(void) { return g_err; }
extern "C" int clx_abi_version(void) { return 2; }
extern "C" int clx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
