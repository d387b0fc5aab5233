// xl_common.h -- small host-side helpers shared by the drop-in filter and the batch engine.
#ifndef XL_COMMON_H_
#define XL_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// journald-style error line, the reference's logging convention (e.g. src/lpf.c:14, src/dsp_worker.c:69)
#define XL_LOG_ERR(...)                 \
  do {                                  \
    fprintf(stderr, "<3>xlating-hip: "); \
    fprintf(stderr, __VA_ARGS__);       \
    fprintf(stderr, "\n");              \
  } while (0)

// The HIP error of the latest failed XL_TRY of this thread, and its errno: allocation failures -> -ENOMEM, no
// usable device / no gfx950 code object -> -ENODEV, anything else -> -EIO.
extern thread_local hipError_t xl_last_hip_error;
int xl_errno_of_last_hip_error(void);

#define XL_TRY(expr)                                                                       \
  do {                                                                                     \
    hipError_t xl_e_ = (expr);                                                             \
    if (xl_e_ != hipSuccess) {                                                             \
      xl_last_hip_error = xl_e_;                                                           \
      XL_LOG_ERR("%s failed: %s (%s:%d)", #expr, hipGetErrorString(xl_e_), __FILE__, __LINE__); \
      goto fail;                                                                           \
    }                                                                                      \
  } while (0)

// journald "warning" line
#define XL_LOG_WARN(...)                \
  do {                                  \
    fprintf(stderr, "<4>xlating-hip: "); \
    fprintf(stderr, __VA_ARGS__);       \
    fprintf(stderr, "\n");              \
  } while (0)

// Tuning / test knobs from the environment (XL_EXP_*): honoured ONLY when the process also sets XL_TESTING=1 (tests/ and tools/ do)
// or in -DXL_TUNING builds.  A plain process ignores them -- its plan must not depend on stray environment (the reference's only
// switch is the config file's cpu_optimization, src/config.c:252-264) -- and says so once on stderr if any is set.
extern "C" const char *xl_exp_getenv(const char *name);

// Thread-safe lazy probe.  Returns the device ordinal to use (>= 0) or -1 when no usable HIP device exists.
// `requested` < 0 selects the calling thread's current device.
int xl_hip_select_device(int requested);
extern "C" const char *xlating_hip_device_info(void);  // include/xlating_batch.h

static inline uint32_t xl_bytes_per_sample(int fmt) { return fmt == 0 || fmt == 1 ? 2u : (fmt == 2 ? 4u : 8u); }
static inline uint32_t xl_roundup(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

#endif
