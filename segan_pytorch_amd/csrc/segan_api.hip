// segan_api.hip — error reporting and version of libsegan_hip.
#include "segan_common.h"
#include <string.h>
#include <stdlib.h>
#include <atomic>

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

// Process-wide, read by the launch planners of every thread while segan_set_reserved_slots may be
// writing it from another one (the header advertises the setter for hosts with concurrent streams
// and threads): an atomic, initialised from the environment exactly once (round-5 advice).
static std::atomic<int> g_reserved_slots{-1};      // -1: not set yet (environment default)
static int reserved_slots_env(void) {
  const char* e = getenv("SEGAN_RESERVED_SLOTS");
  const int v = e ? atoi(e) : 0;
  return v < 0 ? 0 : (v > 512 ? 512 : v);
}
int segan_reserved_slots_value(void) {
  int v = g_reserved_slots.load(std::memory_order_relaxed);
  if (v < 0) {
    int expected = -1;
    const int env = reserved_slots_env();
    // the first reader publishes the environment default; a concurrent setter's value wins
    v = g_reserved_slots.compare_exchange_strong(expected, env, std::memory_order_relaxed) ? env : expected;
  }
  return v;
}
extern "C" int segan_set_reserved_slots(int n) {
  const int prev = segan_reserved_slots_value();
  g_reserved_slots.store(n < 0 ? 0 : (n > 512 ? 512 : n), std::memory_order_relaxed);
  return prev;
}

extern "C" int segan_abi_version(void) { return SEGAN_ABI_VERSION; }
extern "C" const char* segan_last_error(void) { return g_err; }

// ---- default transform vectors (ones / zeros) for NULL segan_src members -----------------
__device__ float g_xf_ones[SEGAN_MAX_XF_CH];
__device__ float g_xf_zeros[SEGAN_MAX_XF_CH];

__global__ void xf_defaults_init_kernel() {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < SEGAN_MAX_XF_CH;
       i += gridDim.x * blockDim.x) {
    g_xf_ones[i] = 1.0f;
    g_xf_zeros[i] = 0.0f;
  }
}

int segan_src_defaults(segan_src* s, hipStream_t st, const char* what) {
  if (s->scale && s->shift && s->slope) return SEGAN_OK;
  static float* ones = nullptr;
  static float* zeros = nullptr;
  if (!ones) {
    void *po = nullptr, *pz = nullptr;
    if (hipGetSymbolAddress(&po, HIP_SYMBOL(g_xf_ones)) != hipSuccess ||
        hipGetSymbolAddress(&pz, HIP_SYMBOL(g_xf_zeros)) != hipSuccess) {
      segan_set_error("%s: cannot resolve the default transform vectors", what);
      return SEGAN_ELAUNCH;
    }
    hipLaunchKernelGGL(xf_defaults_init_kernel, dim3(16), dim3(256), 0, st);
    // make the constants visible to every stream before first use
    if (hipStreamSynchronize(st) != hipSuccess) {
      segan_set_error("%s: initialising the default transform vectors failed", what);
      return SEGAN_ELAUNCH;
    }
    ones = (float*)po;
    zeros = (float*)pz;
  }
  if (s->C0 + s->C1 > SEGAN_MAX_XF_CH) {
    segan_set_error("%s: %d channels exceed the supported %d", what, s->C0 + s->C1,
                    SEGAN_MAX_XF_CH);
    return SEGAN_EUNSUPPORTED;
  }
  if (!s->scale) s->scale = ones;
  if (!s->shift) s->shift = zeros;
  if (!s->slope) s->slope = ones;
  return SEGAN_OK;
}
