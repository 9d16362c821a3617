// K4 (segment max) and K5 (supervised head: l2-normalise, classification losses).  All small,
// elementwise / row-reduction kernels: one wave per row for the reductions, float4 lanes elsewhere.
#include "gs_common.h"
#include <stdlib.h>

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// ------------------------------------------------------------------------------ segment max
__global__ __launch_bounds__(256) void segment_max_fwd_kernel(const float* __restrict__ H, int64_t ldh, int64_t n,
                                                              int32_t s, int32_t hidden, float* __restrict__ pooled,
                                                              int64_t ldp, int32_t* __restrict__ argmax, int64_t lda) {
    const int c4 = (hidden + 3) / 4;
    const int64_t total = n * (int64_t)c4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / c4;
        const int col = (int)(t - i * c4) * 4;
        const float* base = H + i * s * ldh + col;
        f32x4 best = *reinterpret_cast<const f32x4*>(base);
        int ax = 0, ay = 0, az = 0, aw = 0;
        for (int j = 1; j < s; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)j * ldh);
            if (v.x > best.x) { best.x = v.x; ax = j; }
            if (v.y > best.y) { best.y = v.y; ay = j; }
            if (v.z > best.z) { best.z = v.z; az = j; }
            if (v.w > best.w) { best.w = v.w; aw = j; }
        }
        if (col + 1 >= hidden) best.y = 0.f;
        if (col + 2 >= hidden) best.z = 0.f;
        if (col + 3 >= hidden) best.w = 0.f;
        *reinterpret_cast<f32x4*>(pooled + i * ldp + col) = best;
        int32_t* a = argmax + i * lda + col;
        a[0] = ax;
        if (col + 1 < hidden) a[1] = ay;
        if (col + 2 < hidden) a[2] = az;
        if (col + 3 < hidden) a[3] = aw;
    }
}

extern "C" int gs_segment_max_fwd(const float* H, int64_t ldh, int64_t n, int32_t s, int32_t hidden, float* pooled,
                                  int64_t ldp, int32_t* argmax, int64_t lda, void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_CHECK_MAT(H, ldh, "gs_segment_max_fwd H");
    GS_CHECK_MAT(pooled, ldp, "gs_segment_max_fwd pooled");
    GS_REQUIRE(argmax && n >= 0 && s > 0 && hidden > 0 && lda >= hidden, "gs_segment_max_fwd: bad args");
    if (n == 0) return GS_OK;
    const int64_t total = n * (int64_t)((hidden + 3) / 4);
    int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(segment_max_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, H, ldh, n, s, hidden,
                       pooled, ldp, argmax, lda);
    GS_LAUNCH_CHECK("segment_max_fwd_kernel");
    return GS_OK;
}

// reduce_max over rows picked through an index (the MLP ran on the step's UNIQUE ids): one wave per (group, 64-float4
// column chunk); the group's s row indices are loaded once and broadcast with v_readlane, U rows in flight; rows are
// compared in j order, so pooled / argmax are those of segment_max_fwd on the expanded [n*s, hidden] matrix, bit for bit.
template <int U>
__global__ __launch_bounds__(256) void segment_max_gather_kernel(const float* __restrict__ H, int64_t ldh,
                                                                 const int32_t* __restrict__ inv, int64_t n, int32_t s,
                                                                 int32_t hidden, float* __restrict__ pooled, int64_t ldp,
                                                                 int32_t* __restrict__ argmax, int64_t lda) {
    const int lane = threadIdx.x & 63;
    const int c4 = (hidden + 3) / 4, chunks = (c4 + 63) / 64;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n * chunks) return;
    const int64_t i = w / chunks;
    const int col = ((int)(w - i * chunks) * 64 + lane) * 4;
    const bool active = col < hidden;
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int ax = 0, ay = 0, az = 0, aw = 0;
    for (int jb = 0; jb < s; jb += 64) {
        const int cnt = min(64, s - jb);
        int32_t my = 0;
        if (lane < cnt) my = inv[i * s + jb + lane];
        if (active) {
            for (int j = 0; j < cnt; j += U) {
                f32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int32_t r = __builtin_amdgcn_readlane(my, min(j + u, cnt - 1));
                    v[u] = *reinterpret_cast<const f32x4*>(H + (int64_t)r * ldh + col);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (j + u < cnt) {                      // wave-uniform
                        const int jj = jb + j + u;
                        if (v[u].x > best.x) { best.x = v[u].x; ax = jj; }
                        if (v[u].y > best.y) { best.y = v[u].y; ay = jj; }
                        if (v[u].z > best.z) { best.z = v[u].z; az = jj; }
                        if (v[u].w > best.w) { best.w = v[u].w; aw = jj; }
                    }
                }
            }
        }
    }
    if (!active) return;
    if (col + 1 >= hidden) best.y = 0.f;
    if (col + 2 >= hidden) best.z = 0.f;
    if (col + 3 >= hidden) best.w = 0.f;
    *reinterpret_cast<f32x4*>(pooled + i * ldp + col) = best;
    int32_t* a = argmax + i * lda + col;
    a[0] = ax;
    if (col + 1 < hidden) a[1] = ay;
    if (col + 2 < hidden) a[2] = az;
    if (col + 3 < hidden) a[3] = aw;
}

extern "C" int gs_segment_max_gather_fwd(const float* H, int64_t ldh, const int32_t* inv, int64_t n, int32_t s, int32_t hidden,
                                         float* pooled, int64_t ldp, int32_t* argmax, int64_t lda, void* stream) {
    if (n == 0) return GS_OK;
    GS_CHECK_MAT(H, ldh, "gs_segment_max_gather_fwd H");
    GS_CHECK_MAT(pooled, ldp, "gs_segment_max_gather_fwd pooled");
    GS_REQUIRE(inv && argmax && n > 0 && s > 0 && hidden > 0 && lda >= hidden && ldh >= ((hidden + 3) & ~3) && ldp >= ((hidden + 3) & ~3),
               "gs_segment_max_gather_fwd: bad args");
    const int64_t waves = n * (int64_t)((((hidden + 3) / 4) + 63) / 64);
    GS_REQUIRE(gs_ceil_div(waves, 4) < (1ll << 31), "gs_segment_max_gather_fwd: grid too large");
    if (s >= 13)
        hipLaunchKernelGGL(segment_max_gather_kernel<13>, dim3((unsigned)gs_ceil_div(waves, 4)), dim3(256), 0, (hipStream_t)stream, H, ldh,
                           inv, n, s, hidden, pooled, ldp, argmax, lda);
    else
        hipLaunchKernelGGL(segment_max_gather_kernel<5>, dim3((unsigned)gs_ceil_div(waves, 4)), dim3(256), 0, (hipStream_t)stream, H, ldh,
                           inv, n, s, hidden, pooled, ldp, argmax, lda);
    GS_LAUNCH_CHECK("segment_max_gather_kernel");
    return GS_OK;
}

__global__ __launch_bounds__(256) void segment_max_bwd_kernel(const float* __restrict__ d_pooled, int64_t ldd,
                                                              const float* __restrict__ pooled, int64_t ldp,
                                                              const int32_t* __restrict__ argmax, int64_t lda,
                                                              int64_t rows, int32_t s, int32_t hidden,
                                                              float* __restrict__ dH, int64_t ldh) {
    const int c4 = (hidden + 3) / 4;
    const int64_t total = rows * (int64_t)c4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / c4;
        const int col = (int)(t - r * c4) * 4;
        const int64_t i = r / s;
        const int j = (int)(r - i * s);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (col + e < hidden) {
                const bool hit = argmax[i * lda + col + e] == j && pooled[i * ldp + col + e] > 0.f;
                g[e] = hit ? d_pooled[i * ldd + col + e] : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(dH + r * ldh + col) = g;
    }
}

extern "C" int gs_segment_max_bwd(const float* d_pooled, int64_t ldd, const float* pooled, int64_t ldp,
                                  const int32_t* argmax, int64_t lda, int64_t n, int32_t s, int32_t hidden, float* dH,
                                  int64_t ldh, void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_CHECK_MAT(dH, ldh, "gs_segment_max_bwd dH");
    GS_REQUIRE(d_pooled && pooled && argmax && n >= 0 && s > 0 && hidden > 0, "gs_segment_max_bwd: bad args");
    if (n == 0) return GS_OK;
    const int64_t total = n * (int64_t)s * ((hidden + 3) / 4);
    int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 8192);
    hipLaunchKernelGGL(segment_max_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_pooled, ldd, pooled, ldp,
                       argmax, lda, n * (int64_t)s, s, hidden, dH, ldh);
    GS_LAUNCH_CHECK("segment_max_bwd_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------------------ l2 normalise
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int32_t d,
                                                         float* __restrict__ y, int64_t ldy, float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float v = x[r * ldx + c];
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));  // tf.nn.l2_normalize epsilon
    const int dp = (d + 3) & ~3;
    for (int c = lane; c < dp; c += 64) y[r * ldy + c] = c < d ? x[r * ldx + c] * inv : 0.f;
    if (lane == 0 && inv_norm) inv_norm[r] = inv;
}

extern "C" int gs_l2norm_fwd(const float* x, int64_t ldx, int64_t n, int32_t d, float* y, int64_t ldy, float* inv_norm,
                             void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_REQUIRE(x && y && n >= 0 && d > 0 && ldx >= d && ldy >= ((d + 3) & ~3), "gs_l2norm_fwd: bad args");
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)gs_ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, n, d,
                       y, ldy, inv_norm);
    GS_LAUNCH_CHECK("l2norm_fwd_kernel");
    return GS_OK;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                         const float* __restrict__ y, int64_t ldy,
                                                         const float* __restrict__ inv_norm, int64_t n, int32_t d,
                                                         float* __restrict__ dx, int64_t lddx) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float dot = 0.f;
    for (int c = lane; c < d; c += 64) dot += dy[r * lddy + c] * y[r * ldy + c];
    dot = wave_sum(dot);
    const float inv = inv_norm[r];
    const bool clamped = inv >= 1.0e6f;  // sum(x^2) < 1e-12: y = x * 1e6, no normalisation term
    const int dp = (d + 3) & ~3;
    for (int c = lane; c < dp; c += 64) {
        float g = 0.f;
        if (c < d) {
            const float dyv = dy[r * lddy + c];
            g = clamped ? dyv * inv : inv * (dyv - y[r * ldy + c] * dot);
        }
        dx[r * lddx + c] = g;
    }
}

extern "C" int gs_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* inv_norm, int64_t n,
                             int32_t d, float* dx, int64_t lddx, void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_REQUIRE(dy && y && inv_norm && dx && n >= 0 && d > 0 && lddx >= ((d + 3) & ~3), "gs_l2norm_bwd: bad args");
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)gs_ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, dy, lddy, y,
                       ldy, inv_norm, n, d, dx, lddx);
    GS_LAUNCH_CHECK("l2norm_bwd_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------------------ classification loss
__global__ __launch_bounds__(256) void class_loss_kernel(const float* __restrict__ logits, int64_t ldl,
                                                         const float* __restrict__ labels, int64_t ldlab, int64_t n,
                                                         int32_t C, int sigmoid_loss, float* __restrict__ loss_rows,
                                                         float* __restrict__ preds, int64_t ldp,
                                                         float* __restrict__ dlogits, int64_t lddl) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float* x = logits + r * ldl;
    const float* z = labels + r * ldlab;
    const int Cp = (C + 3) & ~3;
    if (sigmoid_loss) {
        float acc = 0.f;
        const float gscale = 1.0f / ((float)n * (float)C);
        for (int c = lane; c < Cp; c += 64) {
            float p = 0.f, g = 0.f;
            if (c < C) {
                const float xv = x[c], zv = z[c];
                acc += fmaxf(xv, 0.f) - xv * zv + log1pf(expf(-fabsf(xv)));
                p = 1.0f / (1.0f + expf(-xv));
                g = (p - zv) * gscale;
            }
            if (preds) preds[r * ldp + c] = p;
            if (dlogits) dlogits[r * lddl + c] = g;
        }
        acc = wave_sum(acc);
        if (lane == 0) loss_rows[r] = acc / (float)C;
    } else {
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
        m = wave_max(m);
        float se = 0.f, zs = 0.f, zx = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float xv = x[c], zv = z[c];
            se += expf(xv - m);
            zs += zv;
            zx += zv * xv;
        }
        se = wave_sum(se);
        zs = wave_sum(zs);
        zx = wave_sum(zx);
        const float lse = m + logf(se);
        const float inv_se = 1.0f / se;
        const float gscale = 1.0f / (float)n;
        for (int c = lane; c < Cp; c += 64) {
            float p = 0.f, g = 0.f;
            if (c < C) {
                p = expf(x[c] - m) * inv_se;
                g = (p * zs - z[c]) * gscale;
            }
            if (preds) preds[r * ldp + c] = p;
            if (dlogits) dlogits[r * lddl + c] = g;
        }
        if (lane == 0) loss_rows[r] = zs * lse - zx;  // -sum_c z_c (x_c - lse)
    }
}

extern "C" int gs_class_loss(const float* logits, int64_t ldl, const float* labels, int64_t ldlab, int64_t n, int32_t C,
                             int sigmoid_loss, float* loss_rows, float* preds, int64_t ldp, float* dlogits, int64_t lddl,
                             void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_REQUIRE(logits && labels && loss_rows && n >= 0 && C > 0, "gs_class_loss: bad args");
    const int Cp = (C + 3) & ~3;
    GS_REQUIRE((!preds || ldp >= Cp) && (!dlogits || lddl >= Cp), "gs_class_loss: output ld must be >= round_up(C,4)");
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(class_loss_kernel, dim3((unsigned)gs_ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, logits, ldl,
                       labels, ldlab, n, C, sigmoid_loss, loss_rows, preds, ldp, dlogits, lddl);
    GS_LAUNCH_CHECK("class_loss_kernel");
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// Fused supervised head, forward AND backward in one launch (supervised_models.py:85-126 + their gradients):
//   y = l2_normalize(x);  logits = y·W + b;  loss/preds/dlogits;  d_y = dlogits·W^T;  d_x = l2norm_bwd(d_y)
// One wave per row; W [d, C] is staged once per block in LDS with an ODD row stride so that both access
// patterns are bank-conflict free: lanes over c at fixed k (logits) and lanes over k at fixed c (d_y).
// Templated on DJ = d/64 and CQ = ceil(C/64) and written branch-free (clamped indices + selects): the generic
// guarded version compiled to 5.3K instructions with 380 branches and SGPR spills.
template <int DJ, int CQ>
__global__ __launch_bounds__(256) void head_fwd_bwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t n,
                                                           const float* __restrict__ W, int64_t ldw,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ labels, int64_t ldlab, int32_t C,
                                                           int sigmoid_loss, int rows_per_wave,
                                                           float* __restrict__ y_out, int64_t ldy,
                                                           float* __restrict__ logits_out, int64_t ldlo,
                                                           float* __restrict__ preds, int64_t ldp,
                                                           float* __restrict__ dlogits, int64_t lddl,
                                                           float* __restrict__ loss_rows, float* __restrict__ dx,
                                                           int64_t lddx) {
    constexpr int d = DJ * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Cp = (C + 3) & ~3;
    const int Cs = Cp | 1;              // odd LDS stride
    float* Ws = lds;                    // [d][Cs]
    float* ybuf = lds + (((size_t)d * Cs + 3) & ~(size_t)3);  // [4 waves][d], 16-byte aligned
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
    const int64_t rfirst = min(row0, n - 1);
    // ---- this wave's first-row inputs are issued BEFORE staging W so they travel under the W loads
    float xr[DJ], lab[CQ], bs[CQ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) xr[j] = x[rfirst * ldx + j * 64 + lane];
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        const int cl = min(q * 64 + lane, C - 1);
        lab[q] = labels[rfirst * ldlab + cl];
        bs[q] = bias ? bias[cl] : 0.f;
    }
    {   // W -> LDS: float4 global loads, 8 in flight per thread; (k, c) advance incrementally (no division)
        const int c4n = Cp >> 2;
        const int total = d * c4n;
        int t = tid;
        int k = t / c4n, c4 = t - k * c4n;
        const int dk = 256 / c4n, dc = 256 - dk * c4n;
        while (t < total) {
            f32x4 v[8];
            int kk[8], cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                kk[u] = k; cc[u] = c4 * 4;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (t + u * 256 < total) v[u] = *reinterpret_cast<const f32x4*>(W + (int64_t)k * ldw + c4 * 4);
                k += dk; c4 += dc;
                if (c4 >= c4n) { c4 -= c4n; ++k; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t + u * 256 < total) {
                    float* dst = Ws + kk[u] * Cs + cc[u];   // odd stride -> scalar LDS stores (pad columns hold 0 or W pad)
                    dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
                }
            }
            t += 8 * 256;
        }
    }
    __syncthreads();
    float* yw = ybuf + wave * d;
    for (int rr = 0; rr < rows_per_wave; ++rr) {
        const int64_t r = row0 + rr;
        if (r >= n) break;  // wave-uniform
        if (rr > 0) {
#pragma unroll
            for (int j = 0; j < DJ; ++j) xr[j] = x[r * ldx + j * 64 + lane];
#pragma unroll
            for (int q = 0; q < CQ; ++q) lab[q] = labels[r * ldlab + min(q * 64 + lane, C - 1)];
        }
        // ---- l2 normalise
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < DJ; ++j) ss += xr[j] * xr[j];
        ss = wave_sum(ss);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        const bool clamped = ss < 1e-12f;
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            xr[j] *= inv;                                   // xr now holds y
            yw[j * 64 + lane] = xr[j];
            y_out[r * ldy + j * 64 + lane] = xr[j];
        }
        // ---- logits (lanes over classes; invalid lanes compute on a clamped column and are masked below)
        float lg[CQ];
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const int cl = min(q * 64 + lane, C - 1);
            float a0 = bs[q], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
            for (int k = 0; k < d; k += 4) {
                const f32x4 yv = *reinterpret_cast<const f32x4*>(yw + k);   // broadcast read
                a0 += yv.x * Ws[(k + 0) * Cs + cl];
                a1 += yv.y * Ws[(k + 1) * Cs + cl];
                a2 += yv.z * Ws[(k + 2) * Cs + cl];
                a3 += yv.w * Ws[(k + 3) * Cs + cl];
            }
            lg[q] = (a0 + a1) + (a2 + a3);
        }
        // ---- loss / preds / dlogits
        float m = -INFINITY, se = 0.f, zs = 0.f, zx = 0.f, loss_acc = 0.f;
#pragma unroll
        for (int q = 0; q < CQ; ++q) m = fmaxf(m, (q * 64 + lane < C) ? lg[q] : -INFINITY);
        m = wave_max(m);
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const bool ok = q * 64 + lane < C;
            se += ok ? expf(lg[q] - m) : 0.f;
            zs += ok ? lab[q] : 0.f;
            zx += ok ? lab[q] * lg[q] : 0.f;
        }
        se = wave_sum(se);
        zs = wave_sum(zs);
        zx = wave_sum(zx);
        const float inv_se = 1.0f / se;
        const float gscale = sigmoid_loss ? 1.0f / ((float)n * (float)C) : 1.0f / (float)n;
        float dl[CQ];
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const int c = q * 64 + lane;
            const bool ok = c < C;
            const float xv = lg[q], zv = lab[q];
            float p, g;
            if (sigmoid_loss) {
                p = 1.0f / (1.0f + expf(-xv));
                g = (p - zv) * gscale;
                loss_acc += ok ? fmaxf(xv, 0.f) - xv * zv + log1pf(expf(-fabsf(xv))) : 0.f;
            } else {
                p = expf(xv - m) * inv_se;
                g = (p * zs - zv) * gscale;
            }
            p = ok ? p : 0.f;
            g = ok ? g : 0.f;
            dl[q] = g;
            if (c < Cp) {
                if (logits_out) logits_out[r * ldlo + c] = ok ? xv : 0.f;
                if (preds) preds[r * ldp + c] = p;
                dlogits[r * lddl + c] = g;
            }
        }
        if (sigmoid_loss) loss_acc = wave_sum(loss_acc);
        if (lane == 0) loss_rows[r] = sigmoid_loss ? loss_acc / (float)C : zs * (m + logf(se)) - zx;
        // ---- d_y = dlogits · W^T (lanes over k) and the l2-normalise backward
        if (dx) {
            float dyk[DJ];
#pragma unroll
            for (int j = 0; j < DJ; ++j) dyk[j] = 0.f;
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                const int cmax = min(64, C - q * 64);      // wave-uniform
                const int dbits = __float_as_int(dl[q]);
#pragma unroll 4
                for (int cc = 0; cc < cmax; ++cc) {        // one class broadcast feeds DJ independent LDS reads/FMAs
                    const float g = __int_as_float(__builtin_amdgcn_readlane(dbits, cc));
                    const float* wcol = Ws + q * 64 + cc;
#pragma unroll
                    for (int j = 0; j < DJ; ++j) dyk[j] += g * wcol[(j * 64 + lane) * Cs];
                }
            }
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < DJ; ++j) dot += dyk[j] * xr[j];
            dot = wave_sum(dot);
#pragma unroll
            for (int j = 0; j < DJ; ++j)
                dx[r * lddx + j * 64 + lane] = clamped ? dyk[j] * inv : inv * (dyk[j] - xr[j] * dot);
        }
    }
}

template <int DJ, int CQ>
static int launch_head(const float* x, int64_t ldx, int64_t n, const float* W, int64_t ldw, const float* bias,
                       const float* labels, int64_t ldlab, int32_t C, int sigmoid_loss, float* y, int64_t ldy,
                       float* logits, int64_t ldlo, float* preds, int64_t ldp, float* dlogits, int64_t lddl,
                       float* loss_rows, float* dx, int64_t lddx, size_t lds_bytes, hipStream_t st) {
    GS_LDS_ATTR(160 * 1024, head_fwd_bwd_kernel<DJ, CQ>);
    const int rows_per_wave = n >= 4096 ? 4 : 1;
    const int64_t blocks = gs_ceil_div(n, 4 * rows_per_wave);
    hipLaunchKernelGGL((head_fwd_bwd_kernel<DJ, CQ>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, x, ldx, n, W, ldw,
                       bias, labels, ldlab, C, sigmoid_loss, rows_per_wave, y, ldy, logits, ldlo, preds, ldp, dlogits, lddl,
                       loss_rows, dx, lddx);
    GS_LAUNCH_CHECK("head_fwd_bwd_kernel");
    return GS_OK;
}

extern "C" int gs_head_fwd_bwd(const float* x, int64_t ldx, int64_t n, int32_t d, const float* W, int64_t ldw,
                               const float* bias, const float* labels, int64_t ldlab, int32_t C, int sigmoid_loss,
                               float* y, int64_t ldy, float* logits, int64_t ldlo, float* preds, int64_t ldp,
                               float* dlogits, int64_t lddl, float* loss_rows, float* dx, int64_t lddx, void* stream) {
    if (n == 0) return GS_OK;
    GS_REQUIRE(x && W && labels && y && dlogits && loss_rows && d > 0 && C > 0, "gs_head_fwd_bwd: bad args");
    const int Cp = (C + 3) & ~3;
    if (!(d == 64 || d == 128 || d == 256 || d == 512) || C > 128) {
        gs_set_error("gs_head_fwd_bwd: fused head supports d in {64,128,256,512} and C <= 128 (got d=%d C=%d); use the unfused kernels", d, C);
        return GS_ENOTSUP;
    }
    GS_REQUIRE(gs_aligned16(W) && ldw % 4 == 0 && ldw >= Cp, "gs_head_fwd_bwd: W must be 16-byte aligned with ld %% 4 == 0 and ld >= round_up(C,4)");
    GS_REQUIRE(ldy >= d && lddl >= Cp && (!preds || ldp >= Cp) && (!logits || ldlo >= Cp) && (!dx || lddx >= d) &&
               ldx >= d && ldlab >= C, "gs_head_fwd_bwd: ld too small");
    const size_t lds_bytes = ((((size_t)d * (Cp | 1) + 3) & ~(size_t)3) + 4 * (size_t)d) * sizeof(float);
    GS_REQUIRE(lds_bytes <= 160 * 1024, "gs_head_fwd_bwd: W does not fit LDS (%zu bytes)", lds_bytes);
    hipStream_t st = (hipStream_t)stream;
#define GS_HEAD_CASE(DJ, CQ) return launch_head<DJ, CQ>(x, ldx, n, W, ldw, bias, labels, ldlab, C, sigmoid_loss, y, ldy, logits, ldlo, \
                                                        preds, ldp, dlogits, lddl, loss_rows, dx, lddx, lds_bytes, st)
    const int cq = (C + 63) / 64;
    if (d == 64) { if (cq == 1) GS_HEAD_CASE(1, 1); else GS_HEAD_CASE(1, 2); }
    if (d == 128) { if (cq == 1) GS_HEAD_CASE(2, 1); else GS_HEAD_CASE(2, 2); }
    if (d == 256) { if (cq == 1) GS_HEAD_CASE(4, 1); else GS_HEAD_CASE(4, 2); }
    if (cq == 1) GS_HEAD_CASE(8, 1); else GS_HEAD_CASE(8, 2);
#undef GS_HEAD_CASE
}

// ---------------------------------------------------------------------------------------------------
// Sparse weight gradient of the MaxPool MLP at layer 0 (aggregators.py:176-181 backward).
// reduce_max routes the gradient of each (group g, hidden column c) to ONE of the s neighbor rows, so
//   dW_mlp[f, c] = sum_g v[g, c] * X[ids[g*s + argmax[g, c]], f],      v = d_pooled masked by (pooled > 0)
// is ~s-times sparser than the dense X^T·dH the reference executes (TF materialises dH = [n*s, hidden]).
// Workgroup = (CB = 128 columns) x (a slice of groups): the s feature rows of a group are staged in LDS once,
// thread t owns feature f = t and 128 column accumulators; the arg-max row of column c is broadcast with
// v_readlane (SGPR row base), so the inner loop is one ds_read_b32 + one v_fmac per (f, c).  dH never exists.
// Slices write split-K slabs (same layout as gs_dense_wgrad) summed by the flat reduce.
#define GS_SPW_FB 64  // features per block (lane = feature)
#define GS_SPW_PF 4   // float4 of the next group's row segments prefetched per thread
// Block (fb, slice): features [64 fb, 64 fb + 64) x ALL hidden columns; wave w owns columns [64 w, 64 w + 64).  Each
// X row is therefore gathered once per launch (10 blocks read disjoint 256 B segments of it); only the small
// arg-max / value rows are re-read per feature block.
template <int GB>   // groups staged per barrier pair
__global__ __launch_bounds__(512) void maxpool_sparse_wgrad_kernel(const float* __restrict__ X, int64_t ldx,
                                                                    const int32_t* __restrict__ ids, int64_t G, int32_t s,
                                                                    int32_t d, const int32_t* __restrict__ argmax, int64_t lda,
                                                                    const float* __restrict__ dpm, int64_t ldd,
                                                                    int32_t hidden, int64_t groups_per_slice,
                                                                    float* __restrict__ slabs, int64_t ld_slab) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [GB*s][64]
    const int dpad = (d + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x;
    const int f0 = blockIdx.x * GS_SPW_FB;
    const int c0 = blockIdx.z * 512 + wave * 64;
    const int64_t g0 = (int64_t)blockIdx.y * groups_per_slice;
    const int64_t g1 = min(G, g0 + groups_per_slice);
    const int n4 = GB * s * (GS_SPW_FB / 4);                     // float4 per staged chunk
    float acc[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) acc[c] = 0.f;
    // software pipeline: the next chunk's segments / arg-max / values load to registers while this chunk computes
    f32x4 pf[GS_SPW_PF];
    int a_nx[GB];
    float v_nx[GB];
    auto prefetch = [&](int64_t g) {
        const int64_t rows = (min(g1, g + GB) - g) * s;          // sampled rows left in this slice
#pragma unroll
        for (int u = 0; u < GS_SPW_PF; ++u) {
            const int t = tid + u * nthreads;
            if (t < n4) {
                const int r = t >> 4, q = t & 15;
                pf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < rows && f0 + q * 4 < dpad) {
                    const int64_t id = ids[g * s + r];
                    pf[u] = *reinterpret_cast<const f32x4*>(X + id * ldx + f0 + q * 4);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < GB; ++b) {
            a_nx[b] = 0; v_nx[b] = 0.f;                          // value 0 => a group past the slice end adds nothing
            if (g + b < g1 && c0 + lane < hidden) {
                a_nx[b] = argmax[(g + b) * lda + c0 + lane];
                v_nx[b] = dpm[(g + b) * ldd + c0 + lane];
            }
        }
    };
    if (g0 < g1) prefetch(g0);
    for (int64_t g = g0; g < g1; g += GB) {
        __syncthreads();  // previous chunk's segments are no longer read
#pragma unroll
        for (int u = 0; u < GS_SPW_PF; ++u) {
            const int t = tid + u * nthreads;
            if (t < n4) *reinterpret_cast<f32x4*>(&xs[t * 4]) = pf[u];
        }
        int a_cur[GB], v_cur[GB];
#pragma unroll
        for (int b = 0; b < GB; ++b) { a_cur[b] = a_nx[b]; v_cur[b] = __float_as_int(v_nx[b]); }
        __syncthreads();
        if (g + GB < g1) prefetch(g + GB);
#pragma unroll
        for (int b = 0; b < GB; ++b) {
            const float* xb = xs + b * s * GS_SPW_FB + lane;
#pragma unroll
            for (int c = 0; c < 64; ++c) {
                const int ra = __builtin_amdgcn_readlane(a_cur[b], c);
                const float va = __int_as_float(__builtin_amdgcn_readlane(v_cur[b], c));
                acc[c] += va * xb[ra * GS_SPW_FB];
            }
        }
    }
    const int f = f0 + lane;
    if (f < d) {
        float* dst = slabs + ((int64_t)blockIdx.y * d + f) * ld_slab + c0;
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
            if (c0 + c + 3 < hidden) {
                *reinterpret_cast<f32x4*>(dst + c) = f32x4{acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c0 + c + e < hidden) dst[c + e] = acc[c + e];
            }
        }
    }
}

// Column-per-lane form (the default): thread t owns hidden column c = 512 z + t and 64 feature accumulators; the s
// feature-row segments of a group are staged in LDS with a row stride of 66 floats: rows stay 8-byte aligned, and lanes
// reading DIFFERENT arg-max rows at the same feature pair hit different banks (ds_read_b64: bank pair (2 r + f) mod 64,
// distinct for r < 32 rows) while lanes reading the same row get a broadcast.  The arg-max row and the value of a column
// are per-lane registers (coalesced loads, no v_readlane), the row base address is computed once per group and every PAIR
// of FMAs is ONE ds_read_b64 with an immediate offset: 1.5 instructions per useful FMA (round 2: ds_read_b32 at an odd
// stride of 65, 2 per FMA -- the kernel is LDS-bandwidth bound and ds_read_b64 moves twice the bytes per LDS cycle; round
// 1: ~5).  512 x 64 x 4 B per (group, feature block).
#define GS_SPW_LDS_STRIDE 66
#define GS_SPW_DEPTH 4
typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
template <int PF>
__global__ __launch_bounds__(512) void maxpool_sparse_wgrad_cols_kernel(const float* __restrict__ X, int64_t ldx,
                                                                         const int32_t* __restrict__ ids, int64_t G, int32_t s,
                                                                         int32_t d, const int32_t* __restrict__ argmax,
                                                                         int64_t lda, const float* __restrict__ dpm, int64_t ldd,
                                                                         int32_t hidden, int64_t groups_per_slice,
                                                                         float* __restrict__ slabs, int64_t ld_slab) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [s][65]
    const int dpad = (d + 3) & ~3;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    // (placing the feature blocks of one group slice on one XCD, so that its arg-max rows and values are L2 hits for nine of
    // ten blocks, was measured: 180 vs 171 us for the step's two launches -- no gain, left out)
    const int slice = blockIdx.y, fb = blockIdx.x;
    const int f0 = fb * GS_SPW_FB;
    const int c = blockIdx.z * 512 + tid;
    const bool col_ok = c < hidden;
    const int64_t g0 = (int64_t)slice * groups_per_slice;
    const int64_t g1 = min(G, g0 + groups_per_slice);
    const int n4 = s * (GS_SPW_FB / 4);                          // float4 of a group's row segments
    float acc[GS_SPW_FB];
#pragma unroll
    for (int f = 0; f < GS_SPW_FB; ++f) acc[f] = 0.f;
    // A group's work is short (64 FMAs per thread) against the latency of its gathered rows (~2 us from HBM): the rows, arg-max
    // rows and values of the next GS_SPW_DEPTH groups are in flight (round 2-3: one group ahead -- 156 us for the 5120 x 25 hop,
    // i.e. ~1.5 us per group of which the arithmetic is 0.3)
    constexpr int DEPTH = GS_SPW_DEPTH;
    f32x4 pf[DEPTH][PF];
    int a_nx[DEPTH];
    float v_nx[DEPTH];
    auto prefetch = [&](const int slot, int64_t g) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = tid + u * nthreads;
            pf[slot][u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < n4) {
                const int r = t >> 4, q = t & 15;
                if (f0 + q * 4 < dpad) pf[slot][u] = *reinterpret_cast<const f32x4*>(X + (int64_t)ids[g * s + r] * ldx + f0 + q * 4);
            }
        }
        a_nx[slot] = col_ok ? argmax[g * lda + c] : 0;
        v_nx[slot] = col_ok ? dpm[g * ldd + c] : 0.f;
    };
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
        a_nx[j] = 0; v_nx[j] = 0.f;
        if (g0 + j < g1) prefetch(j, g0 + j);
    }
    for (int64_t gb = g0; gb < g1; gb += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int64_t g = gb + j;
            if (g >= g1) break;                            // uniform
            __syncthreads();  // the previous group's segments are no longer read
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = tid + u * nthreads;
                if (t < n4) {
                    float* dst = xs + (t >> 4) * GS_SPW_LDS_STRIDE + (t & 15) * 4;      // 8-byte aligned (264-byte rows)
                    *reinterpret_cast<gs_f32x2*>(dst) = gs_f32x2{pf[j][u].x, pf[j][u].y};
                    *reinterpret_cast<gs_f32x2*>(dst + 2) = gs_f32x2{pf[j][u].z, pf[j][u].w};
                }
            }
            const float v = v_nx[j];
            // LDS byte address of the lane's arg-max row.  The reads are written as ds_read_b64 by hand: left to the compiler,
            // pairs of them become ds_read2_b64, which moves HALF the bytes per LDS cycle (MI355X_MICROARCH.md, LDS table).
            const unsigned row = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)xs +
                                 (unsigned)a_nx[j] * (GS_SPW_LDS_STRIDE * 4u);
            __syncthreads();
            if (g + DEPTH < g1) prefetch(j, g + DEPTH);
#define GS_RD(i, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[i]) : "v"(row), "i"(off))
#define GS_RD8(o) GS_RD(0, o); GS_RD(1, o + 8); GS_RD(2, o + 16); GS_RD(3, o + 24); GS_RD(4, o + 32); GS_RD(5, o + 40); \
                  GS_RD(6, o + 48); GS_RD(7, o + 56)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {                  // four batches of 8 reads: 16 registers of row data at a time
                gs_f32x2 x[8];
                if (qt == 0) { GS_RD8(0); } else if (qt == 1) { GS_RD8(64); } else if (qt == 2) { GS_RD8(128); } else { GS_RD8(192); }
                // the wait is tied to the eight destination registers ("+v"): the FMAs below depend on ITS outputs, so neither
                // the compiler nor the machine scheduler can hoist them above the wait (the reads land asynchronously)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                             :: "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[16 * qt + 2 * i] += v * x[i].x;
                    acc[16 * qt + 2 * i + 1] += v * x[i].y;
                }
            }
#undef GS_RD8
#undef GS_RD
        }
    }
    if (col_ok) {
        float* dst = slabs + ((int64_t)slice * d + f0) * ld_slab + c;
#pragma unroll
        for (int f = 0; f < GS_SPW_FB; ++f)
            if (f0 + f < d) dst[(int64_t)f * ld_slab] = acc[f];
    }
}

// LDS-DMA form of the column-per-lane kernel (round 5; the default for hidden >= 512 and s <= 32).
// What was wrong with the form above (profiles/r05_maxpool_kernel_stats.md: 149 us for the 5120 x 25 hop, 1.4 us per group for
// 0.2 us of LDS time): its "prefetch" loads a row through `X + ids[g s + r] * ldx` -- the row load depends on the id load,
// so the compiler puts s_waitcnt vmcnt(0) between the two, and vmcnt is an in-order counter: that wait also drains the row
// loads of every group ahead.  Each group paid a full memory round trip; a register ring deep enough to fix it does not
// fit beside 64 accumulators in the 128 VGPRs that keep two workgroups on a CU (tried: 740 bytes of scratch per lane).
// Here nothing that is in flight lives in a VGPR:
//   * the slice's sampled ids are staged in LDS once per GS_SPD_IDS_CAP ids (one round trip per chunk, not per group);
//   * the row segments, arg-max rows and values of group g + 3 go global -> LDS with global_load_lds_dword (one wave
//     instruction = one 256-byte row segment at a wave-uniform LDS address, i.e. the padded 66-float row stride of the
//     bank-conflict-free ds_read_b64 pattern survives -- the 16-byte form would force a lane-linear layout) into a ring of
//     GS_SPD_NBUF stages; every wave issues exactly RPW + 2 of them per group, so `s_waitcnt vmcnt(2 (RPW + 2))` says
//     "my part of group g has landed", and ONE s_barrier per group makes everybody's part visible and frees the stage
//     that group g - 1 was read from (its reads were retired by the lgkmcnt(0) of the last batch);
//   * every load is unconditional (clamped group / row / column; a clamped group is multiplied by v = 0, a clamped column
//     lands in an accumulator that is never stored): straight-line code, exact counts.
// Arithmetic and summation order per (feature, column) are those of the form above: bit-identical slabs.
#define GS_SPD_NBUF 4
#define GS_SPD_IDS_CAP 3072     // sampled ids of a slice staged in LDS at a time (12 KB: 122 groups of 25)
// DIAG (benchmarks/micro_spw.py only, wrong values): bit 0 = no LDS reads / FMAs, bit 1 = no DMA inside the loop, bit 2 = no
// barrier inside the loop, bit 3 = row reads without FMAs, bit 4 = FMAs without row reads.
template <int RPW, int DIAG>              // RPW: row segments per wave and group = ceil(s / 8)
__device__ __forceinline__ void spw_dma_body(const float* __restrict__ X, int64_t ldx, const int32_t* __restrict__ ids, int64_t G,
                                             int32_t s, int32_t d, const int32_t* __restrict__ argmax, int64_t lda,
                                             const float* __restrict__ dpm, int64_t ldd, int32_t hidden, int64_t groups_per_slice,
                                             float* __restrict__ slabs, int64_t ld_slab, const int slice, const int fb) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr int NBUF = GS_SPD_NBUF;
    constexpr int ROWS = 8 * RPW;                                // row slots of a stage (>= s)
    constexpr int STAGE = ROWS * GS_SPW_LDS_STRIDE + 1024;       // floats: rows | int32 arg-max[512] | float value[512]
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [NBUF][STAGE] | int32 ids[GS_SPD_IDS_CAP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = fb * GS_SPW_FB;
    const int c = blockIdx.z * 512 + tid;
    const bool col_ok = c < hidden;
    const int cc = min(c, hidden - 1);                           // loads of a column past the end: clamped, never stored
    const int fcol = min(f0 + lane, d - 1);                      // this lane's feature of a row segment (clamped into the row)
    const int64_t g0 = (int64_t)slice * groups_per_slice;
    const int64_t g1 = min(G, g0 + groups_per_slice);
    int32_t* ids_l = reinterpret_cast<int32_t*>(xs + NBUF * STAGE);
    const int chunk_groups = max(1, GS_SPD_IDS_CAP / s);
    float acc[GS_SPW_FB];
#pragma unroll
    for (int f = 0; f < GS_SPW_FB; ++f) acc[f] = 0.f;
    const unsigned xs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)xs;

    for (int64_t gc = g0; gc < g1; gc += chunk_groups) {
        const int64_t ge = min(g1, gc + chunk_groups);
        const int ng = (int)(ge - gc);
        __syncthreads();                                         // the previous chunk's ids / stages are no longer read
        for (int i = tid; i < ng * s; i += 512) ids_l[i] = ids[gc * s + i];
        __syncthreads();
        int id_nx[RPW];                                          // source rows of the NEXT issue (read from LDS one issue ahead)
        auto read_ids = [&](const int gl) {
            const int gq = min(gl, ng - 1);
#pragma unroll
            for (int u = 0; u < RPW; ++u) id_nx[u] = __builtin_amdgcn_readfirstlane(ids_l[gq * s + min(wave + 8 * u, s - 1)]);   // wave-uniform: scalar row address
        };
        auto issue = [&](const int st, const int gl) {           // group gc + min(gl, ng - 1) -> stage st: RPW + 2 DMA per wave
            const int64_t g = gc + min(gl, ng - 1);
            float* sb = xs + st * STAGE;
#pragma unroll
            for (int u = 0; u < RPW; ++u)
                __builtin_amdgcn_global_load_lds(X + (int64_t)id_nx[u] * ldx + fcol,
                                                 (lds_ptr_t)(sb + (wave + 8 * u) * GS_SPW_LDS_STRIDE), 4, 0, 0);
            __builtin_amdgcn_global_load_lds(argmax + g * lda + cc, (lds_ptr_t)(sb + ROWS * GS_SPW_LDS_STRIDE + wave * 64), 4, 0, 0);
            __builtin_amdgcn_global_load_lds(dpm + g * ldd + cc, (lds_ptr_t)(sb + ROWS * GS_SPW_LDS_STRIDE + 512 + wave * 64), 4, 0, 0);
            read_ids(gl + 1);
        };
        read_ids(0);
#pragma unroll
        for (int j = 0; j < NBUF - 1; ++j) issue(j, j);
        const int rounds = (ng + NBUF - 1) / NBUF;
        for (int rd = 0; rd < rounds; ++rd) {
#pragma unroll
            for (int j = 0; j < NBUF; ++j) {
                const int gl = rd * NBUF + j;                    // group gc + gl is consumed from stage j
                // my part of group gl has landed: the two groups issued after it may still be in flight
                if (DIAG & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (RPW == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (RPW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (RPW == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                if (!(DIAG & 4)) __builtin_amdgcn_s_barrier();   // ... and so has everybody's; stage (j + 3) % 4 (group gl - 1) is free
                asm volatile("" ::: "memory");
                if (!(DIAG & 2)) issue((j + NBUF - 1) % NBUF, gl + NBUF - 1);
                if (DIAG & 1) continue;
                const float* sb = xs + j * STAGE;
                const int a_cur = reinterpret_cast<const int32_t*>(sb + ROWS * GS_SPW_LDS_STRIDE)[tid];
                const float v_ld = (sb + ROWS * GS_SPW_LDS_STRIDE + 512)[tid];
                const float v = gl < ng ? v_ld : 0.f;            // a clamped (repeated) group adds nothing
                // LDS byte address of the lane's arg-max row.  The reads are written as ds_read_b64 by hand: left to the compiler,
                // pairs of them become ds_read2_b64, which moves HALF the bytes per LDS cycle (MI355X_MICROARCH.md, LDS table).
                const unsigned row = xs_base + (unsigned)(j * STAGE + a_cur * GS_SPW_LDS_STRIDE) * 4u;
#define GS_RD(xv, i, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xv[i]) : "v"(row), "i"(off) : "memory")
#define GS_RD8(xv, o) GS_RD(xv, 0, o); GS_RD(xv, 1, o + 8); GS_RD(xv, 2, o + 16); GS_RD(xv, 3, o + 24); GS_RD(xv, 4, o + 32); \
                      GS_RD(xv, 5, o + 40); GS_RD(xv, 6, o + 48); GS_RD(xv, 7, o + 56)
                // the wait is tied to the eight destination registers ("+v"): the FMAs depend on ITS outputs, so neither the
                // compiler nor the machine scheduler can hoist them above the wait (the reads land asynchronously).
#define GS_WAIT(n, xv) asm volatile("s_waitcnt lgkmcnt(" #n ")" \
                             : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7]) \
                             :: "memory")
#define GS_FMA8(xv, base) _Pragma("unroll") for (int i = 0; i < 8; ++i) { acc[base + 2 * i] += v * xv[i].x; acc[base + 2 * i + 1] += v * xv[i].y; }
                // (two batches in flight need 16 more registers and spill at the 128 VGPRs that keep two workgroups on a CU -- and a
                //  spill reload is a VMEM operation the compiler waits for with vmcnt(0), which drains the DMA ring)
                gs_f32x2 xa[8];
                if (DIAG & 16) {                                 // diagnostics: the FMAs without the row reads
#pragma unroll
                    for (int i = 0; i < 8; ++i) xa[i] = gs_f32x2{v, __int_as_float(a_cur)};
                    GS_FMA8(xa, 0) GS_FMA8(xa, 16) GS_FMA8(xa, 32) GS_FMA8(xa, 48)
                    continue;
                }
                if (DIAG & 8) {                                  // diagnostics: the row reads without the FMAs
#define GS_USE8(xv) _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(xv[i]));
                    GS_RD8(xa, 0); GS_WAIT(0, xa); GS_USE8(xa)
                    GS_RD8(xa, 64); GS_WAIT(0, xa); GS_USE8(xa)
                    GS_RD8(xa, 128); GS_WAIT(0, xa); GS_USE8(xa)
                    GS_RD8(xa, 192); GS_WAIT(0, xa); GS_USE8(xa)
#undef GS_USE8
                    continue;
                }
                GS_RD8(xa, 0);
                GS_WAIT(0, xa);
                GS_FMA8(xa, 0)
                GS_RD8(xa, 64);
                GS_WAIT(0, xa);
                GS_FMA8(xa, 16)
                GS_RD8(xa, 128);
                GS_WAIT(0, xa);
                GS_FMA8(xa, 32)
                GS_RD8(xa, 192);
                GS_WAIT(0, xa);
                GS_FMA8(xa, 48)
#undef GS_FMA8
#undef GS_WAIT
#undef GS_RD8
#undef GS_RD
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped look-ahead issues of the chunk's last groups
    }
    if (col_ok) {
        float* dst = slabs + ((int64_t)slice * d + f0) * ld_slab + c;
#pragma unroll
        for (int f = 0; f < GS_SPW_FB; ++f)
            if (f0 + f < d) dst[(int64_t)f * ld_slab] = acc[f];
    }
}

// Measured on an MI355X at the Reddit shape (benchmarks/micro_spw.py, gpurun_out/r5q-r5s; 5120 groups x 25, cold rows):
//   this kernel 142 us (117 us inside the step, where the rows are warm) vs 152 (149) for the register-staged form;
//   DMA only 85 us = 577 MB at 6.8 TB/s (rows 308 MB + arg-max / values 21 MB x 10 feature blocks + slabs 59 MB);
//   row reads only 68 us, FMAs only 45 us, both 88 us, both without the barrier 88 us: LDS reads and FMAs of the
//   same CU add up instead of overlapping, the barrier costs nothing.
// Tried on top and not kept (bit-identical slabs, no gain): the feature blocks of a slice on ONE XCD (arg-max / values from L2:
// DMA only 74 us, whole kernel 143 vs 142 us); two batches of eight reads in flight at one workgroup per CU (173 us).
template <int RPW, int DIAG>
__global__ __launch_bounds__(512, 4) void maxpool_sparse_wgrad_dma_kernel(const float* __restrict__ X, int64_t ldx,
                                                                        const int32_t* __restrict__ ids, int64_t G, int32_t s,
                                                                        int32_t d, const int32_t* __restrict__ argmax,
                                                                        int64_t lda, const float* __restrict__ dpm, int64_t ldd,
                                                                        int32_t hidden, int64_t groups_per_slice,
                                                                        float* __restrict__ slabs, int64_t ld_slab) {
    spw_dma_body<RPW, DIAG>(X, ldx, ids, G, s, d, argmax, lda, dpm, ldd, hidden, groups_per_slice, slabs, ld_slab, blockIdx.y, blockIdx.x);
}

template <int RPW, int DIAG = 0>
static int launch_spw_dma(dim3 grid, hipStream_t st, const float* X, int64_t ldx, const int32_t* ids, int64_t n_groups, int32_t s,
                           int32_t d, const int32_t* argmax, int64_t lda, const float* dpm, int64_t ldd, int32_t hidden, int64_t gps,
                           float* slabs, int64_t ld_slab) {
    const size_t lds = ((size_t)GS_SPD_NBUF * (8 * RPW * GS_SPW_LDS_STRIDE + 1024) + GS_SPD_IDS_CAP) * sizeof(float);
    GS_LDS_ATTR(80 * 1024, maxpool_sparse_wgrad_dma_kernel<RPW, DIAG>);
    hipLaunchKernelGGL((maxpool_sparse_wgrad_dma_kernel<RPW, DIAG>), grid, dim3(512), lds, st, X, ldx, ids, n_groups, s, d, argmax, lda, dpm,
                       ldd, hidden, gps, slabs, ld_slab);
    return GS_OK;
}

extern "C" int gs_maxpool_sparse_wgrad(const float* X, int64_t ldx, const int32_t* ids, int64_t n_groups, int32_t s,
                                       int32_t d, const int32_t* argmax, int64_t lda, const float* d_pooled_masked,
                                       int64_t ldd, int32_t hidden, int32_t n_slabs, float* slabs, int64_t ld_slab,
                                       void* stream) {
    if (n_groups == 0) return GS_OK;
    GS_REQUIRE(X && ids && argmax && d_pooled_masked && slabs && s > 0 && d > 0 && hidden > 0 && n_slabs > 0,
               "gs_maxpool_sparse_wgrad: bad args");
    GS_REQUIRE(gs_aligned16(X) && ldx % 4 == 0 && ldx >= ((d + 3) & ~3) && gs_aligned16(slabs) && ld_slab % 4 == 0 &&
               ld_slab >= ((hidden + 3) & ~3), "gs_maxpool_sparse_wgrad: alignment / ld");
    const int threads = hidden >= 512 ? 512 : ((hidden + 63) / 64) * 64;
    if (s * (GS_SPW_FB / 4) > GS_SPW_PF * threads) {
        gs_set_error("gs_maxpool_sparse_wgrad: needs 16*s <= %d*min(512, round_up(hidden,64)) (hidden=%d, s=%d); use gs_dense_wgrad",
                     GS_SPW_PF, hidden, s);
        return GS_ENOTSUP;
    }
    const int64_t gps = gs_ceil_div(n_groups, n_slabs);
    const dim3 grid((unsigned)gs_ceil_div(d, GS_SPW_FB), (unsigned)n_slabs, (unsigned)gs_ceil_div(hidden, 512));
    static const bool feature_lanes = getenv("GS_SPW_FEATURE_LANES") != nullptr;   // the first form (lane = feature), kept for A/B
    if (feature_lanes) {
        // one group per barrier pair: staging 2 or 4 groups per pair measured slower (more VGPRs -> fewer resident blocks)
        hipLaunchKernelGGL(maxpool_sparse_wgrad_kernel<1>, grid, dim3(threads), (size_t)s * GS_SPW_FB * sizeof(float),
                           (hipStream_t)stream, X, ldx, ids, n_groups, s, d, argmax, lda, d_pooled_masked, ldd, hidden, gps,
                           slabs, ld_slab);
    } else if (threads == 512 && s <= 32 && !getenv("GS_SPW_NO_DMA")) {
        // the LDS-DMA pipeline (<= 62 KB of LDS: two workgroups per CU)
        hipStream_t st = (hipStream_t)stream;
        const int rpw = (s + 7) / 8;
        int rc = GS_OK;
#define GS_SPW_V(...) launch_spw_dma<__VA_ARGS__>(grid, st, X, ldx, ids, n_groups, s, d, argmax, lda, d_pooled_masked, ldd, hidden, gps, slabs, ld_slab)
        if (rpw == 1) rc = GS_SPW_V(1);
        else if (rpw == 2) rc = GS_SPW_V(2);
        else if (rpw == 3) rc = GS_SPW_V(3);
        else {
            static const int diag = getenv("GS_SPW_DIAG") ? atoi(getenv("GS_SPW_DIAG")) : 0;     // benchmarks/micro_spw.py
            if (diag == 1) rc = GS_SPW_V(4, 1);                  // DMA + barriers only
            else if (diag == 2) rc = GS_SPW_V(4, 2);             // compute only
            else if (diag == 6) rc = GS_SPW_V(4, 6);             // ... without the barrier
            else if (diag == 14) rc = GS_SPW_V(4, 14);           // ... row reads only
            else if (diag == 22) rc = GS_SPW_V(4, 22);           // ... FMAs only
            else rc = GS_SPW_V(4);
        }
#undef GS_SPW_V
        if (rc != GS_OK) return rc;
    } else {
        // PF: float4 of a group's row segments per thread (16 s of them over the block's threads)
        if (s * (GS_SPW_FB / 4) <= threads)
            hipLaunchKernelGGL(maxpool_sparse_wgrad_cols_kernel<1>, grid, dim3(threads), (size_t)s * GS_SPW_LDS_STRIDE * sizeof(float),
                               (hipStream_t)stream, X, ldx, ids, n_groups, s, d, argmax, lda, d_pooled_masked, ldd, hidden, gps,
                               slabs, ld_slab);
        else
            hipLaunchKernelGGL(maxpool_sparse_wgrad_cols_kernel<GS_SPW_PF>, grid, dim3(threads), (size_t)s * GS_SPW_LDS_STRIDE * sizeof(float),
                               (hipStream_t)stream, X, ldx, ids, n_groups, s, d, argmax, lda, d_pooled_masked, ldd, hidden, gps,
                               slabs, ld_slab);
    }
    GS_LAUNCH_CHECK("maxpool_sparse_wgrad_kernel");
    return GS_OK;
}
