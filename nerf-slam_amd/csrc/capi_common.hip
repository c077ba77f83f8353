// capi_common.hip -- error reporting and version entry points of the C ABI.
#include "common.h"

static thread_local char g_err[512] = "";

void ns_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ns_last_error(void) { return g_err; }
extern "C" int ns_version(void) { return 1; }
extern "C" const char* ns_arch(void) { return "gfx950"; }
