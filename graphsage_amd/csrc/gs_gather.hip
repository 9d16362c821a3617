// K2: feature-row gather fused with the segmented mean (HBM-bound; the roofline-defining kernel).
//
// Work decomposition: one WAVE per (output row, 64-float4 column chunk).  The s neighbor ids of the
// row are loaded once by lanes 0..s-1 and broadcast through v_readlane (SGPR row base + per-lane
// 16-byte column offset), so every neighbor row is fetched as full 1-KiB wave loads = 8 whole
// 128-byte lines when the table's leading dimension is a multiple of 32 floats (F=602 -> ld=608).
// U independent 16-B loads per lane are issued before the first add (>= 8 KiB in flight per wave,
// >= 32 waves per CU), the [n*s, d] gathered tensor of models.py:299 never exists.
// Summation order is j = 0..s-1, fixed => results are deterministic run to run.
#include "gs_common.h"
#include <stdio.h>
#include <stdlib.h>
#include "gs_gather_dev.h"

template <int U, bool DROP = false>
__global__ __launch_bounds__(256) void gather_mean_kernel(const GatherArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t n_items = a.n * (int64_t)a.chunks;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_items) return;  // wave-uniform
    gather_mean_wave<U, DROP>(a, w, lane);
}

static int launch_gather_mean(const float* X, int64_t ldx, const int32_t* idx, int64_t n, int32_t s, int32_t d,
                              const float* S, int64_t ld_self, const int32_t* sidx, float* out, int64_t ldo,
                              float scale, hipStream_t st, const DropArgs* drop = nullptr) {
    const int d4 = (d + 3) / 4;
    const int chunks = (d4 + 63) / 64;
    const int64_t n_items = n * (int64_t)chunks;
    const int64_t blocks = gs_ceil_div(n_items, 4);
    GS_REQUIRE(blocks < (1ll << 31), "gather: grid too large (%lld blocks)", (long long)blocks);
    GatherArgs a = {X, ldx, idx, n, s, d, S, ld_self, sidx, out, ldo, scale, chunks, {0ull, nullptr, 0u, 0u, 1.0f, 0, nullptr, 0}};
    if (drop && drop->thresh16) {
        a.drop = *drop;
        if (s >= 8)
            hipLaunchKernelGGL((gather_mean_kernel<8, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((gather_mean_kernel<1, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        GS_LAUNCH_CHECK("gather_mean_kernel<dropout>");
        return GS_OK;
    }
    // diagnostics (benchmarks/probe_gather_occupancy.py): GS_GATHER_PROBE="U,lds_bytes" picks the loads in flight per lane
    // and pads the launch with dynamic LDS to cap the resident waves per CU
    static const char* probe = getenv("GS_GATHER_PROBE");
    if (probe && s >= 8) {
        int pu = 8, pl = 0;
        sscanf(probe, "%d,%d", &pu, &pl);
        if (pl > 64 * 1024) {
            GS_HIP(hipFuncSetAttribute((const void*)gather_mean_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            GS_HIP(hipFuncSetAttribute((const void*)gather_mean_kernel<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            GS_HIP(hipFuncSetAttribute((const void*)gather_mean_kernel<25>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        if (pu >= 25) hipLaunchKernelGGL(gather_mean_kernel<25>, dim3((unsigned)blocks), dim3(256), (size_t)pl, st, a);
        else if (pu >= 13) hipLaunchKernelGGL(gather_mean_kernel<13>, dim3((unsigned)blocks), dim3(256), (size_t)pl, st, a);
        else hipLaunchKernelGGL(gather_mean_kernel<8>, dim3((unsigned)blocks), dim3(256), (size_t)pl, st, a);
        GS_LAUNCH_CHECK("gather_mean_kernel<probe>");
        return GS_OK;
    }
    if (s >= 8)
        hipLaunchKernelGGL(gather_mean_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (s >= 4)
        hipLaunchKernelGGL(gather_mean_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(gather_mean_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    GS_LAUNCH_CHECK("gather_mean_kernel");
    return GS_OK;
}

static int gather_mean_entry(const float* X, int64_t ldx, const int32_t* idx, int64_t n, int32_t s, int32_t d,
                             const float* self_src, int64_t ld_self, const int32_t* self_idx, float* mean, int64_t ldm,
                             const DropArgs* drop, void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_CHECK_MAT(X, ldx, "gs_gather_mean_fwd X");
    GS_CHECK_MAT(mean, ldm, "gs_gather_mean_fwd mean");
    GS_REQUIRE(n >= 0 && s > 0 && d > 0, "gs_gather_mean_fwd: bad sizes n=%lld s=%d d=%d", (long long)n, s, d);
    const int d4x4 = ((d + 3) / 4) * 4;
    GS_REQUIRE(ldx >= d4x4 && ldm >= d4x4, "gs_gather_mean_fwd: ld must be >= round_up(d,4)");
    if (self_src) {
        GS_CHECK_MAT(self_src, ld_self, "gs_gather_mean_fwd self");
        GS_REQUIRE(ld_self >= d4x4, "gs_gather_mean_fwd: ld_self must be >= round_up(d,4)");
    }
    GS_REQUIRE(n * (int64_t)s < (1ll << 31) || idx, "gs_gather_mean_fwd: contiguous mode needs n*s < 2^31");
    if (n == 0) return GS_OK;
    const float scale = self_src ? 1.0f / (float)(s + 1) : 1.0f / (float)s;
    return launch_gather_mean(X, ldx, idx, n, s, d, self_src, ld_self, self_idx, mean, ldm, scale,
                              (hipStream_t)stream, drop);
}

extern "C" int gs_gather_mean_fwd(const float* X, int64_t ldx, const int32_t* idx, int64_t n, int32_t s, int32_t d,
                                  const float* self_src, int64_t ld_self, const int32_t* self_idx, float* mean,
                                  int64_t ldm, void* stream) {
    return gather_mean_entry(X, ldx, idx, n, s, d, self_src, ld_self, self_idx, mean, ldm, nullptr, stream);
}

extern "C" int gs_gather_mean_dropout_fwd(const float* X, int64_t ldx, const int32_t* idx, int64_t n, int32_t s, int32_t d,
                                          const float* self_src, int64_t ld_self, const int32_t* self_idx, float* mean,
                                          int64_t ldm, const gs_dropout* drop, void* stream) {
    DropArgs da;
    GS_REQUIRE(gs_drop_args(drop, &da) == 0, "gs_gather_mean_dropout_fwd: dropout rate must be in [0, 1)");
    return gather_mean_entry(X, ldx, idx, n, s, d, self_src, ld_self, self_idx, mean, ldm, &da, stream);
}

// ------------------------------------------------------------------ dropout of (gathered) rows; its own backward
__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* __restrict__ X, int64_t ldx,
                                                           const int32_t* __restrict__ ids, int64_t n, int32_t d,
                                                           const DropArgs p, float* __restrict__ out, int64_t ldo) {
    const int d4 = (d + 3) / 4;
    const int64_t total = n * (int64_t)d4;
    const uint64_t key = gs_drop_key(p);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / d4;
        const int q = (int)(t - r * d4);
        const int64_t src = ids ? (int64_t)ids[r] : r;
        f32x4 v = *reinterpret_cast<const f32x4*>(X + src * ldx + q * 4);
        if (p.thresh16) v = gs_drop4(v, p, key, p.row0 + r, q);
        *reinterpret_cast<f32x4*>(out + r * ldo + q * 4) = gs_mask_tail(v, q * 4, d);
    }
}

extern "C" int gs_dropout_rows(const float* X, int64_t ldx, const int32_t* ids, int64_t n, int32_t d,
                               const gs_dropout* drop, float* out, int64_t ldo, void* stream) {
    if (n == 0) return GS_OK;
    GS_CHECK_MAT(X, ldx, "gs_dropout_rows X");
    GS_CHECK_MAT(out, ldo, "gs_dropout_rows out");
    const int d4x4 = ((d + 3) / 4) * 4;
    GS_REQUIRE(n > 0 && d > 0 && ldx >= d4x4 && ldo >= d4x4, "gs_dropout_rows: bad sizes / ld");
    DropArgs da;
    GS_REQUIRE(gs_drop_args(drop, &da) == 0, "gs_dropout_rows: dropout rate must be in [0, 1)");
    const int64_t total = n * (int64_t)(d4x4 / 4);
    const int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 65536);
    hipLaunchKernelGGL(dropout_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, ids, n, d, da, out, ldo);
    GS_LAUNCH_CHECK("dropout_rows_kernel");
    return GS_OK;
}

extern "C" int gs_gather_rows(const float* X, int64_t ldx, const int32_t* ids, int64_t n, int32_t d, float* out,
                              int64_t ldo, void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_CHECK_MAT(X, ldx, "gs_gather_rows X");
    GS_CHECK_MAT(out, ldo, "gs_gather_rows out");
    GS_REQUIRE(ids && n >= 0 && d > 0, "gs_gather_rows: bad args");
    const int d4x4 = ((d + 3) / 4) * 4;
    GS_REQUIRE(ldx >= d4x4 && ldo >= d4x4, "gs_gather_rows: ld must be >= round_up(d,4)");
    if (n == 0) return GS_OK;
    return launch_gather_mean(X, ldx, ids, n, 1, d, nullptr, 0, nullptr, out, ldo, 1.0f, (hipStream_t)stream);
}

// ------------------------------------------------------------------ backward of the segmented mean
__global__ __launch_bounds__(256) void mean_bwd_kernel(const float* __restrict__ d_mean, int64_t ldd, int64_t rows,
                                                       int32_t s, int32_t d, float scale,
                                                       const float* __restrict__ mask_y, int64_t ldy,
                                                       float* __restrict__ d_neigh, int64_t ldn, int accumulate) {
    const int d4 = (d + 3) / 4;
    const int64_t total = rows * (int64_t)d4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / d4;
        const int col = (int)(t - r * d4) * 4;
        f32x4 g = *reinterpret_cast<const f32x4*>(d_mean + (r / s) * ldd + col) * scale;
        if (mask_y) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(mask_y + r * ldy + col);
            g.x = y.x > 0.f ? g.x : 0.f;
            g.y = y.y > 0.f ? g.y : 0.f;
            g.z = y.z > 0.f ? g.z : 0.f;
            g.w = y.w > 0.f ? g.w : 0.f;
        }
        f32x4* dst = reinterpret_cast<f32x4*>(d_neigh + r * ldn + col);
        if (accumulate) g += *dst;
        *dst = gs_mask_tail(g, col, d);
    }
}

extern "C" int gs_mean_bwd(const float* d_mean, int64_t ldd, int64_t n, int32_t s, int32_t d, float scale,
                           const float* mask_y, int64_t ldy, float* d_neigh, int64_t ldn, int accumulate,
                           void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_CHECK_MAT(d_mean, ldd, "gs_mean_bwd d_mean");
    GS_CHECK_MAT(d_neigh, ldn, "gs_mean_bwd d_neigh");
    if (mask_y) GS_CHECK_MAT(mask_y, ldy, "gs_mean_bwd mask_y");
    GS_REQUIRE(n >= 0 && s > 0 && d > 0, "gs_mean_bwd: bad sizes");
    if (n == 0) return GS_OK;
    const int64_t total = n * (int64_t)s * ((d + 3) / 4);
    int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 2048);
    hipLaunchKernelGGL(mean_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_mean, ldd, n * (int64_t)s,
                       s, d, scale, mask_y, ldy, d_neigh, ldn, accumulate);
    GS_LAUNCH_CHECK("mean_bwd_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------ input gradients of a layer, one launch
struct PullArgs {
    const float* d_self;
    int64_t ld_self, n_self;
    int32_t n_seg, d;
    const float* src[GS_PULL_MAX];
    int64_t ld_src[GS_PULL_MAX], row0[GS_PULL_MAX], row1[GS_PULL_MAX];
    int32_t s[GS_PULL_MAX];
    float scale[GS_PULL_MAX];
    const float* mask_y;
    int64_t ldy;
    float* out;
    int64_t ldo, rows;
};

__global__ __launch_bounds__(256) void input_grad_pull_kernel(const PullArgs a) {
    const int d4 = (a.d + 3) / 4;
    const int64_t total = a.rows * (int64_t)d4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / d4;
        const int col = (int)(t - r * d4) * 4;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        bool any = false;
        if (a.d_self && r < a.n_self) {
            g = *reinterpret_cast<const f32x4*>(a.d_self + r * a.ld_self + col);
            any = true;
        }
#pragma unroll
        for (int k = 0; k < GS_PULL_MAX; ++k) {
            if (k < a.n_seg && r >= a.row0[k] && r < a.row1[k]) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(a.src[k] + ((r - a.row0[k]) / a.s[k]) * a.ld_src[k] + col) * a.scale[k];
                g = any ? v + g : v;     // (the compiler may contract scale*x + g into an fma)
                any = true;
            }
        }
        if (a.mask_y) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(a.mask_y + r * a.ldy + col);
            g.x = y.x > 0.f ? g.x : 0.f;
            g.y = y.y > 0.f ? g.y : 0.f;
            g.z = y.z > 0.f ? g.z : 0.f;
            g.w = y.w > 0.f ? g.w : 0.f;
        }
        *reinterpret_cast<f32x4*>(a.out + r * a.ldo + col) = gs_mask_tail(g, col, a.d);
    }
}

extern "C" int gs_input_grad_pull(const gs_pull_desc* q, void* stream) {
    GS_REQUIRE(q && q->rows >= 0 && q->d > 0 && q->n_seg >= 0 && q->n_seg <= GS_PULL_MAX, "gs_input_grad_pull: bad args");
    if (q->rows == 0) return GS_OK;
    const int d4x4 = ((q->d + 3) / 4) * 4;
    GS_CHECK_MAT(q->out, q->ldo, "gs_input_grad_pull out");
    GS_REQUIRE(q->ldo >= d4x4, "gs_input_grad_pull: ldo too small");
    PullArgs a = {};
    a.d_self = q->d_self; a.ld_self = q->ld_self; a.n_self = q->n_self; a.n_seg = q->n_seg; a.d = q->d;
    if (q->d_self) {
        GS_CHECK_MAT(q->d_self, q->ld_self, "gs_input_grad_pull d_self");
        GS_REQUIRE(q->ld_self >= d4x4 && q->n_self >= 0 && q->n_self <= q->rows, "gs_input_grad_pull: bad d_self");
    }
    for (int k = 0; k < q->n_seg; ++k) {
        GS_CHECK_MAT(q->src[k], q->ld_src[k], "gs_input_grad_pull src");
        GS_REQUIRE(q->ld_src[k] >= d4x4 && q->s[k] > 0 && q->n[k] >= 0 && q->row0[k] >= 0 &&
                   q->row0[k] + q->n[k] * q->s[k] <= q->rows, "gs_input_grad_pull: bad segment %d", k);
        a.src[k] = q->src[k]; a.ld_src[k] = q->ld_src[k]; a.row0[k] = q->row0[k];
        a.row1[k] = q->row0[k] + q->n[k] * q->s[k]; a.s[k] = q->s[k]; a.scale[k] = q->scale[k];
    }
    if (q->mask_y) {
        GS_CHECK_MAT(q->mask_y, q->ldy, "gs_input_grad_pull mask_y");
        GS_REQUIRE(q->ldy >= d4x4, "gs_input_grad_pull: ldy too small");
    }
    a.mask_y = q->mask_y; a.ldy = q->ldy; a.out = q->out; a.ldo = q->ldo; a.rows = q->rows;
    const int64_t total = q->rows * (int64_t)(d4x4 / 4);
    const int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(input_grad_pull_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("input_grad_pull_kernel");
    return GS_OK;
}
