// segan_api.hip — error reporting and version of libsegan_hip.
#include "segan_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void segan_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int segan_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    segan_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SEGAN_ELAUNCH;
  }
  return SEGAN_OK;
}

extern "C" int segan_abi_version(void) { return SEGAN_ABI_VERSION; }
extern "C" const char* segan_last_error(void) { return g_err; }
