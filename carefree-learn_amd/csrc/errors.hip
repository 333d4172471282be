// thread-local error message + version for libcfhip.so
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void cfhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int cfhip_version(void) { return CFHIP_VERSION; }
extern "C" const char* cfhip_last_error(void) { return g_err; }
