// Shared helpers of libreadhip.so (gfx950 only; no CUDA compatibility paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "read_hip.h"
#ifdef READ_DEBUG_KNOBS
#include "read_hip_debug.h"   // debug build: the probes and the kernel timeline
#endif

namespace readhip {

void set_error(const char *fmt, ...);

#define READ_CHECK_ARG(cond, ...)                                  \
    do {                                                           \
        if (!(cond)) {                                             \
            ::readhip::set_error(__VA_ARGS__);                     \
            return READ_EINVAL;                                    \
        }                                                          \
    } while (0)

#define READ_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ::readhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),        \
                                 __FILE__, __LINE__);                                         \
            return READ_EHIP;                                                                 \
        }                                                                                     \
    } while (0)

// Launch-error check that does not synchronise.
#define READ_CHECK_LAUNCH() READ_CHECK_HIP(hipGetLastError())

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace readhip
