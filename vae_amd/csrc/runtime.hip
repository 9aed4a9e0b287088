// Library-level entry points: version / arch probe and error reporting.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int vae_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* vae_last_error(void) { return g_err; }
extern "C" int vae_version(void) { return 100; }
extern "C" const char* vae_arch(void) { return "gfx950"; }

// Number of GPUs the HIP runtime sees (0 on a CPU-only host); never raises.
extern "C" int vae_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
