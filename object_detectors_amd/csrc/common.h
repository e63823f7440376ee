// Shared declarations for liblmi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lmi_hip.h"

// Records a printf-style message for lmi_last_error() on the calling thread.
void lmi_set_error(const char *fmt, ...);

// Checks the launch that was just enqueued; returns LMI_ERR_LAUNCH and records `what` on failure.
int lmi_check_launch(const char *what);

static inline int lmi_is_aligned(const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

#define LMI_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      lmi_set_error(__VA_ARGS__);    \
      return (code);                 \
    }                                \
  } while (0)
