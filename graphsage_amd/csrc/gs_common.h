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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that drives more than one GPU must
// set it on each of them (a once-per-process flag left the second device's launches of > 64 KB LDS failing).  One bit per
// device ordinal and call site; the result is checked.  Usage: GS_LDS_ATTR(bytes, kernel<template, args>);
#define GS_LDS_ATTR(bytes, ...)                                                                                             \
    do {                                                                                                                    \
        static unsigned long long done__ = 0ull;                                                                            \
        int dev__ = 0;                                                                                                      \
        GS_HIP(hipGetDevice(&dev__));                                                                                       \
        if (dev__ >= 64 || !((done__ >> dev__) & 1ull)) {                                                                   \
            GS_HIP(hipFuncSetAttribute((const void*)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            if (dev__ < 64) done__ |= 1ull << dev__;                                                                        \
        }                                                                                                                   \
    } while (0)

static inline bool gs_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__host__ __device__ static inline int64_t gs_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#define GS_CHECK_MAT(ptr, ld, name)                                                        \
    GS_REQUIRE((ptr) != nullptr && gs_aligned16(ptr) && ((ld) % 4) == 0,                   \
               "%s: matrix must be non-null, 16-byte aligned, ld %% 4 == 0 (ld=%lld)", name, (long long)(ld))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// splitmix64 finalizer: xorshift-multiply rounds.  Restated bit-for-bit in oracle/sampler_hash.py.
__device__ __forceinline__ uint64_t gs_mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

// ---- inverted dropout, tf.nn.dropout(x, keep_prob = 1 - rate) (aggregators.py:46-47,104-105; layers.py:107) ----
// The keep mask is a pure function of (seed, device step clock, call site, global row, float4 column): one mix64 per
// float4 gives 16 random bits per element, an element is dropped iff its bits < thresh16 = round(rate * 2^16).
// Forward and backward regenerate the same mask; nothing is stored.  Restated in oracle/sampler_hash.py.
struct DropArgs {
    uint64_t seed;
    const uint64_t* clock;   // device step counter (nullable = 0): new masks every step, also under hipGraph replay
    uint32_t site;
    uint32_t thresh16;       // 0 = dropout off
    float scale;             // 1 / keep_prob
    int64_t row0;            // global index of the call's first row
    const uint8_t* keep;     // parity-test hook: keep bits read from memory instead of the hash (gs_dropout.keep_bits)
    int64_t keep_ld;
};

__device__ __forceinline__ uint64_t gs_drop_key(const DropArgs& p) {
    const uint64_t st = p.clock ? *p.clock : 0ull;
    return gs_mix64(p.seed ^ (st * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)p.site << 32) ^ (0xD0ull << 56));
}

__device__ __forceinline__ f32x4 gs_drop4(f32x4 v, const DropArgs& p, uint64_t key, int64_t grow, int q) {
    if (p.keep) {                       // injected masks (uniform branch; tests only)
        const uint32_t b = *reinterpret_cast<const uint32_t*>(p.keep + grow * p.keep_ld + 4 * q);
        v.x = (b & 0xffu) ? v.x * p.scale : 0.f;
        v.y = (b & 0xff00u) ? v.y * p.scale : 0.f;
        v.z = (b & 0xff0000u) ? v.z * p.scale : 0.f;
        v.w = (b & 0xff000000u) ? v.w * p.scale : 0.f;
        return v;
    }
    const uint32_t thresh16 = p.thresh16;
    const float scale = p.scale;
    const uint64_t h = gs_mix64(key + (uint64_t)grow * 0xD1342543DE82EF95ull + (uint64_t)q);
    v.x = (((uint32_t)h) & 0xffffu) >= thresh16 ? v.x * scale : 0.f;
    v.y = (((uint32_t)(h >> 16)) & 0xffffu) >= thresh16 ? v.y * scale : 0.f;
    v.z = (((uint32_t)(h >> 32)) & 0xffffu) >= thresh16 ? v.z * scale : 0.f;
    v.w = ((uint32_t)(h >> 48)) >= thresh16 ? v.w * scale : 0.f;
    return v;
}

// host: C-ABI descriptor -> kernel argument (null / rate 0 = off)
static inline int gs_drop_args(const gs_dropout* d, DropArgs* out) {
    DropArgs a = {0ull, nullptr, 0u, 0u, 1.0f, 0, nullptr, 0};
    if (d && d->rate > 0.f) {
        if (!(d->rate < 1.f)) return -1;
        a.seed = d->seed; a.clock = d->clock_dev; a.site = d->site; a.row0 = d->row0;
        uint32_t t = (uint32_t)(d->rate * 65536.0f + 0.5f);
        a.thresh16 = t < 1u ? 1u : (t > 65535u ? 65535u : t);
        a.scale = 1.0f / (1.0f - d->rate);
        if (d->keep_bits) {
            if ((reinterpret_cast<uintptr_t>(d->keep_bits) & 3u) || d->keep_ld <= 0 || (d->keep_ld & 3)) return -1;
            a.keep = d->keep_bits; a.keep_ld = d->keep_ld;
        }
    }
    *out = a;
    return 0;
}


// A product that the compiler cannot contract into an FMA with a following add: the value passes through an empty asm
// statement.  (hipcc runs with -ffp-contract=fast, which IGNORES `#pragma clang fp contract(off)`; measured in round 5: the
// scalar Adam kernel and the fused float4 one differed in 7 % of the parameters by one ulp from the second step on.)
__device__ __forceinline__ float gs_mul_nofma(const float a, const float b) {
    float r = a * b;
    asm volatile("" : "+v"(r));
    return r;
}

// TF-1.x Adam on one element, with the elementwise clip of supervised_models.py:96-99 in front (clip <= 0: off).  ONE body for
// gs_adam_step, the fused slab-sum + Adam launch and the data-parallel step launch (gs_peer_step), every product kept apart
// from the add behind it, so that the three schedules give the same bits whatever code surrounds the call.
__device__ __forceinline__ void gs_adam_elem(float& p, float& m, float& v, float g, const float gscale, const float clip,
                                             const float b1, const float b2, const float eps, const float lr_t) {
    g = gs_mul_nofma(g, gscale);
    if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);
    m = gs_mul_nofma(b1, m) + gs_mul_nofma(1.0f - b1, g);
    v = gs_mul_nofma(b2, v) + gs_mul_nofma(gs_mul_nofma(1.0f - b2, g), g);
    float q = gs_mul_nofma(lr_t, m) / (sqrtf(v) + eps);
    asm volatile("" : "+v"(q));
    p = p - q;
}
// g + wd * p (the gradient of the weight-decay term, supervised_models.py:104-108), the product kept apart from the add
__device__ __forceinline__ f32x4 gs_wd_add(const f32x4 g, const f32x4 p, const float wd) {
    f32x4 r;
    r.x = g.x + gs_mul_nofma(p.x, wd); r.y = g.y + gs_mul_nofma(p.y, wd);
    r.z = g.z + gs_mul_nofma(p.z, wd); r.w = g.w + gs_mul_nofma(p.w, wd);
    return r;
}
__device__ __forceinline__ float gs_adam_lr_t(const float lr, const float b1, const float b2, const uint64_t* step_dev, const int step_offset) {
    const float t = (float)((step_dev ? *step_dev : 0ull) + (uint64_t)step_offset);
    return lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
}

// Step epilogue body (one 256-thread workgroup): loss_out = scale * sum(loss_rows[0:n]) (+= if accumulate), optionally
// aux_out = aux_scale * sum(aux_rows[0:n]), both in a fixed order, then the device counters advance.  Shared by
// finalize_step_kernel (gs_runtime.hip) and the launches that carry the epilogue as an extra workgroup.
struct StepEpilogue {
    const float* loss_rows; int64_t n; float scale; float* loss_out; int accumulate;
    const float* aux_rows; float aux_scale; float* aux_out;
    uint64_t* c0; uint64_t d0; uint64_t* c1; uint64_t d1; uint64_t* c2; uint64_t d2;
};
__device__ __forceinline__ void gs_step_epilogue_block(const StepEpilogue& e, float* part, float* part2) {
    if (e.aux_rows) {                          // a second mean in the same launch (unsupervised: mrr, models.py:404)
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < e.n; i += 256) s += e.aux_rows[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if ((threadIdx.x & 63) == 0) part2[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) e.aux_out[0] = ((part2[0] + part2[1]) + (part2[2] + part2[3])) * e.aux_scale;
    }
    if (e.loss_rows) {
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < e.n; i += 256) s += e.loss_rows[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float tot = ((part[0] + part[1]) + (part[2] + part[3])) * e.scale;
            e.loss_out[0] = e.accumulate ? e.loss_out[0] + tot : tot;
        }
    }
    if (threadIdx.x == 0) {
        if (e.c0) *e.c0 += e.d0;
        if (e.c1) *e.c1 += e.d1;
        if (e.c2) *e.c2 += e.d2;
    }
}
