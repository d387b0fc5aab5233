// xl_common.cpp -- HIP device probing and the exported identification strings.
#include "xl_common.h"

#include <errno.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>

#include "../../include/xlating.h"
#include "../../include/xlating_batch.h"

// Reference: src/xlating.c:145,148,156,268 export one of "AVX" / "ARM NEON" / "Not detected" /
// "Manually turned off"; src/main.c:23 and test/perf_xlating.c:15 print it.
extern "C" const char *SIMD_STATUS = "HIP gfx950";

thread_local hipError_t xl_last_hip_error = hipSuccess;

int xl_errno_of_last_hip_error(void) {
  switch (xl_last_hip_error) {
    case hipErrorOutOfMemory:
      return -ENOMEM;
    case hipErrorNoDevice:
    case hipErrorInvalidDevice:
    case hipErrorNoBinaryForGpu:
    case hipErrorInvalidDeviceFunction:
    case hipErrorSharedObjectInitFailed:
      return -ENODEV;
    default:
      return -EIO;
  }
}

static std::once_flag g_exp_once;
static bool g_exp_enabled = false;

extern char **environ;

extern "C" const char *xl_exp_getenv(const char *name) {
  std::call_once(g_exp_once, [] {
#ifdef XL_TUNING
    g_exp_enabled = true;
#else
    const char *t = getenv("XL_TESTING");
    g_exp_enabled = t != nullptr && strcmp(t, "1") == 0;
    if (!g_exp_enabled)
      for (char **e = environ; e != nullptr && *e != nullptr; ++e)
        if (strncmp(*e, "XL_EXP_", 7) == 0) {
          XL_LOG_WARN("XL_EXP_* tuning variables are set but ignored (they are honoured only next to XL_TESTING=1): first one %.40s", *e);
          break;
        }
#endif
  });
  return g_exp_enabled ? getenv(name) : nullptr;
}

static std::once_flag g_probe_once;
static int g_device_count = 0;
static char g_info[256] = "HIP gfx950 (no device probed)";

static void xl_probe() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_device_count = 0;
    snprintf(g_info, sizeof(g_info), "HIP gfx950 (no usable device: %s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    (void)hipGetLastError();
    return;
  }
  g_device_count = n;
  int cur = 0;
  (void)hipGetDevice(&cur);
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, cur) == hipSuccess) {
    snprintf(g_info, sizeof(g_info), "HIP %s (%s), %d CUs, %d device(s)", p.gcnArchName, p.name, p.multiProcessorCount, n);
  }
}

int xl_hip_select_device(int requested) {
  std::call_once(g_probe_once, xl_probe);
  if (g_device_count <= 0) return -1;
  if (requested < 0) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return -1;
    return cur;
  }
  return requested < g_device_count ? requested : -1;
}

extern "C" const char *xlating_hip_device_info(void) {
  std::call_once(g_probe_once, xl_probe);
  return g_info;
}
