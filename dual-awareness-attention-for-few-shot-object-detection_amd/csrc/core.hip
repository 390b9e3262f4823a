// Error reporting + ABI version for libdana_hip.so.
#include "common.h"
#include "../../include/dana_hip.h"

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
}
