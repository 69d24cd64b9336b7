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

#ifdef TTSK_STAMPS
// Diagnostic build only (`make stamps`; not declared in ttsk.h, not in the product library): one s_memrealtime stamp (100 MHz) into
// dst[0] from a one-thread kernel on `stream` — a replayed step's own clock at the points tools/debug/step_stamps.py marks, where the
// profiler's serialisation would change what is being measured.
__global__ void stamp_kernel(unsigned long long* dst) { dst[0] = __builtin_amdgcn_s_memrealtime(); }
extern "C" int ttsk_debug_stamp(void* dst, void* stream) {
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)dst);
  return TTSK_OK;
}
#endif
