// K6: gradient finalisation and the optimizer (supervised_models.py:95-99, :104-108).
// Weight gradients arrive as split-K slabs from gs_dense_wgrad (deterministic fixed-order sums).
#include "gs_common.h"
#include "gs_sample_dev.h"
#include "gs_gather_dev.h"

#define GS_OPT_THREADS 256   // workgroup size of the fused reduce + Adam launch (and of its sampler / gather riders)

// 32 consecutive outputs x 8 slab groups per workgroup: thread (o, g) sums the slabs z = g, g + 8, ... with 16
// independent loads in flight (one memory round trip for 128 slabs), the 8 partials are added in group order through
// LDS.  Fixed order => deterministic.  (One thread per output walking all slabs in turn took 32 us for the 128 slabs of
// the link-prediction negatives' gradient.)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int32_t n_slabs,
                                                           int64_t slab_stride, int32_t rows, int32_t cols,
                                                           int64_t ld_slab, float wd, const float* __restrict__ w,
                                                           int64_t ldw, float* __restrict__ grad, int64_t ldg,
                                                           int accumulate) {
    __shared__ float part[8][32];
    const int64_t total = (int64_t)rows * cols;
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (int64_t base = (int64_t)blockIdx.x * 32; base < total; base += (int64_t)gridDim.x * 32) {
        const int64_t t = base + o;
        const bool valid = t < total;
        const int64_t tc = valid ? t : total - 1;
        const int r = (int)(tc / cols);
        const int c = (int)(tc - (int64_t)r * cols);
        const float* p = slabs + (int64_t)r * ld_slab + c;
        float s = 0.f;
        for (int z0 = g; z0 < n_slabs; z0 += 8 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)min(z0 + 8 * u, n_slabs - 1) * slab_stride];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (z0 + 8 * u < n_slabs) s += v[u];
        }
        part[g][o] = s;
        __syncthreads();
        if (g == 0 && valid) {
            float acc = part[0][o];
#pragma unroll
            for (int k = 1; k < 8; ++k) acc += part[k][o];
            if (w && wd != 0.f) acc += wd * w[(int64_t)r * ldw + c];
            float* dst = grad + (int64_t)r * ldg + c;
            *dst = accumulate ? *dst + acc : acc;
        }
        __syncthreads();
    }
}

extern "C" int gs_reduce_slabs(const float* slabs, int32_t n_slabs, int64_t slab_stride, int32_t rows, int32_t cols,
                               int64_t ld_slab, float weight_decay, const float* w, int64_t ldw, float* grad,
                               int64_t ldg, int accumulate, void* stream) {
    GS_REQUIRE(slabs && grad && n_slabs > 0 && rows > 0 && cols > 0, "gs_reduce_slabs: bad args");
    const int64_t total = (int64_t)rows * cols;
    int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 32), 4096);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs, slab_stride,
                       rows, cols, ld_slab, weight_decay, w, ldw, grad, ldg, accumulate);
    GS_LAUNCH_CHECK("reduce_slabs_kernel");
    return GS_OK;
}

// Column sums as slabs (bias gradient): grid (col tiles of 64, row slices).  Each of the 4 waves strides
// the slice's rows with 8 independent loads in flight, partials are combined in fixed order.
__global__ __launch_bounds__(256) void colsum_slabs_kernel(const float* __restrict__ Z, int64_t ldz, int64_t n,
                                                           int32_t n_cols, int64_t rows_per_slab,
                                                           float* __restrict__ slabs, int64_t ld_slab) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
    const int64_t r1 = min(n, r0 + rows_per_slab);
    float s = 0.f;
    if (col < n_cols) {
        int64_t r = r0 + wave;
        for (; r + 28 < r1; r += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Z[(r + 4 * u) * ldz + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < r1; r += 4) s += Z[r * ldz + col];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < n_cols)
        slabs[(int64_t)blockIdx.y * ld_slab + col] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

extern "C" int gs_colsum_slabs(const float* Z, int64_t ldz, int64_t n, int32_t n_cols, int32_t n_slabs, float* slabs,
                               int64_t ld_slab, void* stream) {
    GS_REQUIRE(Z && slabs && n > 0 && n_cols > 0 && n_slabs > 0 && n_slabs < 65536 && ld_slab >= n_cols,
               "gs_colsum_slabs: bad args");
    const int64_t rps = gs_ceil_div(n, n_slabs);
    hipLaunchKernelGGL(colsum_slabs_kernel, dim3((unsigned)gs_ceil_div(n_cols, 64), (unsigned)n_slabs), dim3(256), 0,
                       (hipStream_t)stream, Z, ldz, n, n_cols, rps, slabs, ld_slab);
    GS_LAUNCH_CHECK("colsum_slabs_kernel");
    return GS_OK;
}

// TF-1.x Adam with elementwise clip.  t is read from device memory so the launch is hipGraph-replayable.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t count, float lr,
                                                   float b1, float b2, float eps, float clip, float gscale,
                                                   const uint64_t* __restrict__ step_dev, int step_offset) {
    const float lr_t = gs_adam_lr_t(lr, b1, b2, step_dev, step_offset);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i];
        gs_adam_elem(pi, mi, vi, grad[i], gscale, clip, b1, b2, eps, lr_t);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
    }
}

extern "C" int gs_adam_step(float* p, const float* grad, float* m, float* v, int64_t count, float lr, float beta1,
                            float beta2, float eps, float clip, float grad_scale, const uint64_t* step_dev,
                            int32_t step_offset, void* stream) {
    GS_REQUIRE(p && grad && m && v && count >= 0, "gs_adam_step: bad args");
    if (count == 0) return GS_OK;
    int blocks = (int)std::min<int64_t>(gs_ceil_div(count, 256), 2048);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, count, lr, beta1, beta2,
                       eps, clip, grad_scale, step_dev, step_offset);
    GS_LAUNCH_CHECK("adam_kernel");
    return GS_OK;
}

// Single-block fixed-order reductions (loss scalar, weight-decay term).
template <bool SQUARE>
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, int64_t count, float scale,
                                                   float* __restrict__ out, int accumulate) {
    __shared__ float part[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < count; i += 1024) {
        const float v = x[i];
        s += SQUARE ? v * v : v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < 16; ++w) tot += part[w];
        tot *= scale;
        out[0] = accumulate ? out[0] + tot : tot;
    }
}

extern "C" int gs_sum_scaled(const float* x, int64_t count, float scale, float* out, int accumulate, void* stream) {
    GS_REQUIRE(x && out && count >= 0, "gs_sum_scaled: bad args");
    hipLaunchKernelGGL(sum_kernel<false>, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, count, scale, out, accumulate);
    GS_LAUNCH_CHECK("sum_kernel");
    return GS_OK;
}
extern "C" int gs_sumsq_scaled(const float* x, int64_t count, float scale, float* out, int accumulate, void* stream) {
    GS_REQUIRE(x && out && count >= 0, "gs_sumsq_scaled: bad args");
    hipLaunchKernelGGL(sum_kernel<true>, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, count, scale, out, accumulate);
    GS_LAUNCH_CHECK("sum_kernel_sq");
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// Flat gradient finalisation (+ optional fused clip/Adam): one launch over the whole parameter buffer.
#define GS_MAX_VARS 24
#ifndef GS_OPT_SLAB_BATCH
#define GS_OPT_SLAB_BATCH 24          // slab loads in flight per thread (engine.TILED3_MAX_SLABS: a variable's slabs are one round trip)
#endif
__device__ __forceinline__ int64_t gs_readfirstlane_i64(const int64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)x >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// One variable of the flat buffer in units of float4 (every offset / size is a multiple of 4 floats, the buffer is below 2^31
// float4): 32 bytes, ONE scalar load for a wave-uniform index.
struct FlatVar {
    float* slabs;
    int64_t size;                     // floats: the slab stride
    int32_t qoff, qsize;              // float4 units
    int32_t n_slabs;
    int16_t decay, clear;
};
struct FlatVars {
    int32_t qoff[GS_MAX_VARS];        // the offsets once more, side by side (INT32_MAX beyond n): the variable search is one batch of loads
    FlatVar v[GS_MAX_VARS];
    int32_t n;
};

#ifdef GS_TIMELINE
// Diagnostics build only (-DGS_TIMELINE, benchmarks/timeline_optim.py): wall-clock stamps (100 MHz) of the optimizer workgroups
// [entry, descriptors known, slabs landed, stores issued] and of the first rider workgroups [entry, end].
__device__ unsigned long long g_opt_timeline[1024 * 4];
extern "C" int gs_debug_opt_timeline(unsigned long long* out_host, int n) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_opt_timeline), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#define OPT_STAMP(slot, k) do { if (threadIdx.x == 0 && (slot) < 1024) g_opt_timeline[(slot) * 4 + (k)] = wall_clock64(); } while (0)
#else
#define OPT_STAMP(slot, k) do { } while (0)
#endif
__global__ __launch_bounds__(GS_OPT_THREADS) void flat_reduce_adam_kernel(const FlatVars V, float* __restrict__ params,
                                                               float* __restrict__ grads, float* __restrict__ m,
                                                               float* __restrict__ v, int64_t total4, float wd,
                                                               int fuse_adam, float lr, float b1, float b2, float eps,
                                                               float clip, float gscale,
                                                               const uint64_t* __restrict__ step_dev, int step_offset,
                                                               const float* __restrict__ loss_rows, int64_t loss_n,
                                                               float loss_scale, float* __restrict__ loss_out,
                                                               int loss_accumulate, const int opt_blocks,
                                                               const FanoutArgs F, const CoGatherS J) {
    // Workgroups beyond opt_blocks run the fan-out SAMPLER of a later mini-batch (one root each): five dependent memory
    // round trips of almost no work, hidden under this launch instead of heading a step as its own 7 us launch.
    __shared__ int32_t lvl[2][GS_FANOUT_LDS_SMALL];
    __shared__ int32_t law_cols[GS_MAX_HOPS][GS_LAW_COLS];
    if ((int)blockIdx.x >= opt_blocks) {
        int64_t r = (int64_t)blockIdx.x - opt_blocks;
        OPT_STAMP(512 + (int)r, 0);
        if (loss_rows) {
            // the step's scalar loss (supervised_models.py:111-118 reduce_mean): one wave of a workgroup of its own, fixed order
            // (as the first wave of optimizer workgroup 0 it put a round trip ahead of that workgroup's own)
            if (r == 0) {
                if (threadIdx.x < 64) {
                    float sacc = 0.f;
                    for (int64_t i = threadIdx.x; i < loss_n; i += 64) sacc += loss_rows[i];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 64);
                    if (threadIdx.x == 0) loss_out[0] = loss_accumulate ? loss_out[0] + sacc * loss_scale : sacc * loss_scale;
                }
                return;
            }
            --r;
        }
        if (r < F.B) sample_fanout_root<GS_FANOUT_LDS_SMALL>(F, r, lvl, law_cols);
        else run_gather_item<8, 25>(J, (r - F.B) * (GS_OPT_THREADS / 64) + (threadIdx.x >> 6), threadIdx.x & 63);   // ... and gather+mean waves of the next mini-batch
        OPT_STAMP(512 + (int)r + (loss_rows ? 1 : 0), 1);
        return;
    }
    OPT_STAMP((int)blockIdx.x, 0);
    // The launch is one float4 per thread and latency-bound; its dependent chain is [kernel arguments + every variable offset] ->
    // [the wave's variable descriptor + the step counter] -> [slabs, parameter, Adam state] -> stores.  (A per-lane variable index
    // made offset, size, slab pointer, clear and decay flags five dependent vector loads of the argument block; a scalar search
    // loop was as many dependent scalar loads: 2.1 us between a workgroup's entry and its slab requests, benchmarks/timeline_optim.py.)
    uint64_t t_dev = 0ull;
    if (fuse_adam && step_dev) t_dev = *step_dev;
    float lr_t = 0.f;
    bool have_lr = false;
    auto bias_correction = [&]() {
        if (fuse_adam && !have_lr) {
            const float t = (float)(t_dev + (uint64_t)step_offset);
            lr_t = lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));      // gs_adam_lr_t
            have_lr = true;
        }
    };
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (int64_t)opt_blocks * blockDim.x) {
        const int64_t i = q * 4;  // every segment offset/size is a multiple of 4 floats
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // parameter and Adam state first: their loads share the round trip of the slab loads below
        f32x4 p = *reinterpret_cast<const f32x4*>(params + i);
        f32x4 mi = {0.f, 0.f, 0.f, 0.f}, vi = {0.f, 0.f, 0.f, 0.f};
        if (fuse_adam) {
            mi = *reinterpret_cast<const f32x4*>(m + i);
            vi = *reinterpret_cast<const f32x4*>(v + i);
        }
        // the variable(s) of this WAVE's 64 float4 (wave-uniform indices: scalar loads).  The variables tile the flat buffer in
        // order (checked on the host); a wave almost always lies inside one.
        const int q_lo = __builtin_amdgcn_readfirstlane((int)q);   // the first active lane holds the wave's smallest index
        const int q_hi = q_lo + 63;
        int klo = 0, khi = 0;
#pragma unroll
        for (int j = 1; j < GS_MAX_VARS; ++j) {
            klo += q_lo >= V.qoff[j] ? 1 : 0;
            khi += q_hi >= V.qoff[j] ? 1 : 0;
        }
        bool decay = false;
        OPT_STAMP((int)blockIdx.x, 1);
        for (int k = klo; k <= khi; ++k) {
            FlatVar d;                                             // (two 16-byte scalar loads, not a lazy load per field)
            {
                typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
                const i32x4* src = reinterpret_cast<const i32x4*>(&V.v[k]);
                i32x4* dst = reinterpret_cast<i32x4*>(&d);
                dst[0] = src[0];
                dst[1] = src[1];
            }
            const int64_t sz = d.size;
            const int relq = (int)q - d.qoff;
            if (relq >= 0 && relq < d.qsize) {
                float* sp = d.slabs + 4 * (int64_t)relq;
                const int ns = d.n_slabs;
                // 24 slab loads in flight per thread (the 22 slabs of a Reddit step are ONE memory round trip; with 4 in flight
                // they were 6-8); summation order stays z = 0, 1, ...
                for (int z0 = 0; z0 < ns; z0 += GS_OPT_SLAB_BATCH) {
                    f32x4 sv[GS_OPT_SLAB_BATCH];
#pragma unroll
                    for (int u = 0; u < GS_OPT_SLAB_BATCH; ++u) sv[u] = *reinterpret_cast<const f32x4*>(sp + (int64_t)min(z0 + u, ns - 1) * sz);
#pragma unroll
                    for (int u = 0; u < GS_OPT_SLAB_BATCH; ++u)
                        if (z0 + u < ns) g += sv[u];
                }
                if (d.clear) *reinterpret_cast<f32x4*>(sp) = f32x4{0.f, 0.f, 0.f, 0.f};   // atomic accumulator: consume
                decay = d.decay != 0;
            }
        }
#ifdef GS_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        OPT_STAMP((int)blockIdx.x, 2);
        if (decay && wd != 0.f) g = gs_wd_add(g, p, wd);
        *reinterpret_cast<f32x4*>(grads + i) = g;
        if (fuse_adam) {
            bias_correction();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = mi[e], ve = vi[e];
                gs_adam_elem(pe, me, ve, g[e], gscale, clip, b1, b2, eps, lr_t);
                p[e] = pe; mi[e] = me; vi[e] = ve;
            }
            *reinterpret_cast<f32x4*>(m + i) = mi;
            *reinterpret_cast<f32x4*>(v + i) = vi;
            *reinterpret_cast<f32x4*>(params + i) = p;
        }
        OPT_STAMP((int)blockIdx.x, 3);
    }
}

static int flat_reduce_adam_impl(const gs_var_desc* vars_host, int32_t n_vars, float* params, float* grads, float* m,
                                   float* v, int64_t total, float weight_decay, int fuse_adam, float lr, float beta1,
                                   float beta2, float eps, float clip, float grad_scale, const uint64_t* step_dev,
                                   int32_t step_offset, const float* loss_rows, int64_t loss_n, float loss_scale,
                                   float* loss_out, int loss_accumulate, const FanoutArgs* sampler, const gs_gather_desc* jobs_host,
                                 int32_t n_jobs, void* stream) {
    GS_REQUIRE(!loss_rows || (loss_out && loss_n > 0), "gs_flat_reduce_adam: loss_out missing");
    GS_REQUIRE(vars_host && n_vars > 0 && n_vars <= GS_MAX_VARS, "gs_flat_reduce_adam: need 1..%d variables", GS_MAX_VARS);
    GS_REQUIRE(params && grads && total > 0 && total % 4 == 0, "gs_flat_reduce_adam: bad flat buffer");
    GS_REQUIRE(!fuse_adam || (m && v), "gs_flat_reduce_adam: Adam state missing");
    FlatVars V = {};
    V.n = n_vars;
    for (int i = 0; i < GS_MAX_VARS; ++i) V.qoff[i] = INT32_MAX;
    int64_t expect = 0;
    for (int i = 0; i < n_vars; ++i) {
        GS_REQUIRE(vars_host[i].offset == expect && vars_host[i].size > 0 && vars_host[i].size % 4 == 0,
                   "gs_flat_reduce_adam: variables must tile the flat buffer in order (var %d)", i);
        GS_REQUIRE(vars_host[i].n_slabs == 0 || (vars_host[i].slabs && gs_aligned16(vars_host[i].slabs)),
                   "gs_flat_reduce_adam: slabs of var %d missing/misaligned", i);
        GS_REQUIRE((vars_host[i].offset + vars_host[i].size) / 4 < INT32_MAX, "gs_flat_reduce_adam: flat buffer beyond 2^31 float4");
        V.qoff[i] = (int32_t)(vars_host[i].offset / 4);
        V.v[i].qoff = (int32_t)(vars_host[i].offset / 4);
        V.v[i].qsize = (int32_t)(vars_host[i].size / 4);
        V.v[i].size = vars_host[i].size;
        V.v[i].slabs = vars_host[i].slabs;
        V.v[i].n_slabs = vars_host[i].n_slabs;
        V.v[i].decay = vars_host[i].decay ? 1 : 0;
        V.v[i].clear = vars_host[i].clear ? 1 : 0;
        GS_REQUIRE(!vars_host[i].clear || vars_host[i].n_slabs == 1, "gs_flat_reduce_adam: var %d: clear needs n_slabs == 1", i);
        expect += vars_host[i].size;
    }
    GS_REQUIRE(expect <= total, "gs_flat_reduce_adam: variables exceed the flat buffer");
    const int64_t total4 = expect / 4;
    int blocks = (int)std::min<int64_t>(gs_ceil_div(total4, GS_OPT_THREADS), 4096);
    FanoutArgs F = {};
    int64_t roots = 0;
    if (sampler) { F = *sampler; roots = F.B; }
    CoGatherS J = {};
    int64_t waves = 0;
    int rc = build_cojobs_s(jobs_host, n_jobs, &J, &waves);
    if (rc != GS_OK) return rc;
    const int64_t rider_blocks = gs_ceil_div(waves, GS_OPT_THREADS / 64);
    const int loss_block = loss_rows ? 1 : 0;
    GS_REQUIRE(blocks + loss_block + roots + rider_blocks < (1ll << 31), "gs_flat_reduce_adam: grid too large");
    hipLaunchKernelGGL(flat_reduce_adam_kernel, dim3((unsigned)(blocks + loss_block + roots + rider_blocks)), dim3(GS_OPT_THREADS), 0, (hipStream_t)stream, V,
                       params, grads, m, v, total4, weight_decay, fuse_adam, lr, beta1, beta2, eps, clip, grad_scale, step_dev,
                       step_offset, loss_rows, loss_n, loss_scale, loss_out, loss_accumulate, blocks, F, J);
    GS_LAUNCH_CHECK("flat_reduce_adam_kernel");
    return GS_OK;
}

extern "C" int gs_flat_reduce_adam(const gs_var_desc* vars_host, int32_t n_vars, float* params, float* grads, float* m,
                                   float* v, int64_t total, float weight_decay, int fuse_adam, float lr, float beta1,
                                   float beta2, float eps, float clip, float grad_scale, const uint64_t* step_dev,
                                   int32_t step_offset, const float* loss_rows, int64_t loss_n, float loss_scale,
                                   float* loss_out, int loss_accumulate, void* stream) {
    return flat_reduce_adam_impl(vars_host, n_vars, params, grads, m, v, total, weight_decay, fuse_adam, lr, beta1, beta2, eps,
                                 clip, grad_scale, step_dev, step_offset, loss_rows, loss_n, loss_scale, loss_out,
                                 loss_accumulate, nullptr, nullptr, 0, stream);
}

extern "C" int gs_flat_reduce_adam_sample(const gs_var_desc* vars_host, int32_t n_vars, float* params, float* grads, float* m,
                                          float* v, int64_t total, float weight_decay, int fuse_adam, float lr, float beta1,
                                          float beta2, float eps, float clip, float grad_scale, const uint64_t* step_dev,
                                          int32_t step_offset, const float* loss_rows, int64_t loss_n, float loss_scale,
                                          float* loss_out, int loss_accumulate, const gs_fanout_desc* s,
                                          const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    if (!s)
        return flat_reduce_adam_impl(vars_host, n_vars, params, grads, m, v, total, weight_decay, fuse_adam, lr, beta1, beta2,
                                     eps, clip, grad_scale, step_dev, step_offset, loss_rows, loss_n, loss_scale, loss_out,
                                     loss_accumulate, nullptr, jobs_host, n_jobs, stream);
    FanoutArgs F;
    int64_t kmax = 0;
    int rc = gs_fanout_args_desc(s, &F, &kmax);
    if (rc != GS_OK) return rc;
    if (kmax > GS_FANOUT_LDS_SMALL) {
        gs_set_error("gs_flat_reduce_adam_sample: per-root fan-out %lld of a kept hop exceeds %d", (long long)kmax, GS_FANOUT_LDS_SMALL);
        return GS_ENOTSUP;
    }
    return flat_reduce_adam_impl(vars_host, n_vars, params, grads, m, v, total, weight_decay, fuse_adam, lr, beta1, beta2, eps,
                                 clip, grad_scale, step_dev, step_offset, loss_rows, loss_n, loss_scale, loss_out,
                                 loss_accumulate, s->B > 0 ? &F : nullptr, jobs_host, n_jobs, stream);
}
