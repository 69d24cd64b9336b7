// core.hip — version + thread-local error string of libttsk_hip.
#include "common.h"

static thread_local char g_err[512] = "";

void ttsk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ttsk_version(void) { return TTSK_VERSION; }
extern "C" const char* ttsk_last_error(void) { return g_err; }
