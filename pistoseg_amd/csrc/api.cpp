// libpistoseg_hip.so: version / error plumbing (host only).
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/pistoseg_hip.h"

static thread_local char g_err[512] = "";

void ps_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ps_version(void) { return PS_VERSION; }
extern "C" const char* ps_last_error(void) { return g_err; }
extern "C" int ps_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
