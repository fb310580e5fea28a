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

// Compute units of the current device (grid size of the persistent kernels); cached per device id, 256 on MI355X.
int ps_num_cus(void) {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

#ifdef PS_DEBUG_HOOKS
// libpistoseg_hip_debug.so only: every `ps_debug_set_*` tunable back to its default (the lists live beside the tunables' definitions)
void ps_debug_reset_igemm(void);
void ps_debug_reset_wgrad(void);
extern "C" void ps_debug_reset(void) {
  ps_debug_reset_igemm();
  ps_debug_reset_wgrad();
}
#endif
