// Device-side body of K2 (gather + segmented mean), shared by the standalone kernel (gs_gather.hip) and the
// horizontally fused "dense GEMM + next-step gather" kernel (gs_gemm.hip).
#pragma once
#include "gs_common.h"

struct GatherArgs {
    const float* X;
    int64_t ldx;
    const int32_t* idx;
    int64_t n;
    int32_t s, d;
    const float* S;      // GCN self rows (nullable)
    int64_t lds_;
    const int32_t* sidx;
    float* out;
    int64_t ldo;
    float scale;
    int32_t chunks;      // 64-float4 column chunks per row
    DropArgs drop;       // dropout of the gathered rows (thresh16 == 0: off)
};

__device__ __forceinline__ f32x4 gs_mask_tail(f32x4 v, int col, int d) {
    // zero the elements at logical column >= d (only the last float4 of a row can be partial)
    if (col + 3 >= d) {
        if (col + 0 >= d) v.x = 0.f;
        if (col + 1 >= d) v.y = 0.f;
        if (col + 2 >= d) v.z = 0.f;
        if (col + 3 >= d) v.w = 0.f;
    }
    return v;
}

// One wave = one (output row, 64-float4 column chunk) work item `w` (wave-uniform).
template <int U, bool DROP = false>
__device__ __forceinline__ void gather_mean_wave(const GatherArgs& a, const int64_t w, const int lane) {
    const float* __restrict__ X = a.X;
    const int32_t* __restrict__ idx = a.idx;
    const int64_t ldx = a.ldx;
    const int s = a.s, d = a.d, chunks = a.chunks;
    const int64_t row = w / chunks;
    const int c = (int)(w - row * chunks);
    const int col = (c * 64 + lane) * 4;
    const bool active = col < d;

    uint64_t dkey = 0;
    if (DROP) dkey = gs_drop_key(a.drop);
    const int64_t drow = a.drop.row0 + row * s;   // global index of this output row's first sampled row
    const int q = c * 64 + lane;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int jb = 0; jb < s; jb += 64) {
        const int cnt = min(64, s - jb);  // uniform
        int32_t my = 0;
        if (lane < cnt) my = idx ? idx[row * s + jb + lane] : (int32_t)(row * s + jb + lane);
        if (active) {
            int j = 0;
            for (; j + U <= cnt; j += U) {
                f32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int32_t r = __builtin_amdgcn_readlane(my, j + u);
                    v[u] = *reinterpret_cast<const f32x4*>(X + (int64_t)r * ldx + col);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (DROP) v[u] = gs_drop4(v[u], dkey, drow + jb + j + u, q, a.drop.thresh16, a.drop.scale);
                    acc += v[u];
                }
            }
            if (j < cnt) {
                // remainder batch: load everything (index clamped), select afterwards -- keeps the
                // loads unconditional so they stay in flight together.
                f32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int jj = min(j + u, cnt - 1);
                    const int32_t r = __builtin_amdgcn_readlane(my, jj);
                    v[u] = *reinterpret_cast<const f32x4*>(X + (int64_t)r * ldx + col);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float m = (j + u < cnt) ? 1.f : 0.f;
                    if (DROP) v[u] = gs_drop4(v[u], dkey, drow + jb + min(j + u, cnt - 1), q, a.drop.thresh16, a.drop.scale);
                    acc += v[u] * m;
                }
            }
        }
    }
    if (active) {
        if (a.S) {
            const int64_t sr = a.sidx ? (int64_t)a.sidx[row] : row;
            acc += *reinterpret_cast<const f32x4*>(a.S + sr * a.lds_ + col);
        }
        acc *= a.scale;
        acc = gs_mask_tail(acc, col, d);
        *reinterpret_cast<f32x4*>(a.out + row * a.ldo + col) = acc;
    }
}
