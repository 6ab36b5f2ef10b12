// libmfnet_hip: error reporting + version (C ABI, see include/mfnet_hip.h).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void mfnet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mfnet_last_error(void) { return g_err; }
extern "C" int mfnet_version(void) { return 100; }  // 0.1.0
