// ipsx_common.h - host-side helpers shared by the translation units of libipsx.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ipsx.h"

#define IPSX_API extern "C" __attribute__((visibility("default")))

namespace ipsx {

// thread-local message of the last failing call (ipsx_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// after a kernel launch: report a launch error without synchronising
static inline int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(IPSX_EHIP, "%s: %s", what, hipGetErrorString(e));
    return IPSX_OK;
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int conv_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

}  // namespace ipsx

#define IPSX_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ipsx::fail(IPSX_EINVAL, __VA_ARGS__); \
    } while (0)

#define IPSX_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != IPSX_OK) return rc_; \
    } while (0)
