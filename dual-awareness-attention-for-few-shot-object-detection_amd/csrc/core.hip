// Error reporting + ABI version for libdana_hip.so.
#include "common.h"
#include "../../include/dana_hip_debug.h"

static thread_local char g_err[512] = "";

void dana_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
const char* dana_last_error(void) { return g_err; }
int dana_abi_version(void) { return 1; }

// debug (include/dana_hip_debug.h): a stream that may only use the CUs of `mask`
int dana_debug_stream_create_cumask(const unsigned int* mask, int mask_words, int n_xcd, dana_stream_t* stream_out) {
  DANA_CHECK_ARG(mask && stream_out && mask_words > 0 && mask_words <= 32 && n_xcd > 0 && n_xcd <= 32,
                 "dana_debug_stream_create_cumask: bad arguments");
  // the command processor deals workgroups over ALL XCDs; an XCD whose CUs are all masked off never retires its share
  for (int x = 0; x < n_xcd; ++x) {
    int cus = 0;
    for (int i = x; i < mask_words * 32; i += n_xcd) cus += (mask[i >> 5] >> (i & 31)) & 1u;
    DANA_CHECK_ARG(cus > 0, "dana_debug_stream_create_cumask: XCD %d would have no CU (every XCD needs at least one)", x);
  }
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask_words, mask);
  if (e != hipSuccess) {
    dana_set_error("dana_debug_stream_create_cumask: %s", hipGetErrorString(e));
    return DANA_ERR_HIP;
  }
  *stream_out = (dana_stream_t)s;
  return DANA_OK;
}
int dana_debug_stream_destroy(dana_stream_t stream) {
  if (!stream) return DANA_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) {
    dana_set_error("dana_debug_stream_destroy: %s", hipGetErrorString(e));
    return DANA_ERR_HIP;
  }
  return DANA_OK;
}
}
