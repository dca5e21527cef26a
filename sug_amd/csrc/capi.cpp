// Error reporting and ABI version for libsug_amd.so (host-only translation unit).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/sug_amd.h"

static thread_local char g_err[512] = "";

void sug_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sug_last_error(void) { return g_err; }
extern "C" int sug_abi_version(void) { return SUG_ABI_VERSION; }
