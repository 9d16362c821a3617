// Shared helpers for the gfx950 kernels: error plumbing, launch checks, small device utilities.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <algorithm>
#include "../../include/graphsage_amd.h"

#define GS_WAVE 64  // CDNA wavefront width (hard-coded on purpose; warpSize folds to 64 on gfx950)

void gs_set_error(const char* fmt, ...);

#define GS_REQUIRE(cond, ...)                \
    do {                                     \
        if (!(cond)) {                       \
            gs_set_error(__VA_ARGS__);       \
            return GS_EINVAL;                \
        }                                    \
    } while (0)

#define GS_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            gs_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return GS_EHIP;                                                                 \
        }                                                                                   \
    } while (0)

// Kernel launches report configuration errors through hipGetLastError (async faults surface at sync).
#define GS_LAUNCH_CHECK(name)                                                               \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess) {                                                            \
            gs_set_error("launch of %s failed: %s", name, hipGetErrorString(e__));          \
            return GS_EHIP;                                                                 \
        }                                                                                   \
    } while (0)

static inline bool gs_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__host__ __device__ static inline int64_t gs_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#define GS_CHECK_MAT(ptr, ld, name)                                                        \
    GS_REQUIRE((ptr) != nullptr && gs_aligned16(ptr) && ((ld) % 4) == 0,                   \
               "%s: matrix must be non-null, 16-byte aligned, ld %% 4 == 0 (ld=%lld)", name, (long long)(ld))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
