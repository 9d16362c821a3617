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
                    if (DROP) v[u] = gs_drop4(v[u], a.drop, dkey, drow + jb + j + u, q);
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
                    if (DROP) v[u] = gs_drop4(v[u], a.drop, dkey, drow + jb + min(j + u, cnt - 1), q);
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

// ---------------------------------------------------------------------------------------------------------------
// Co-scheduled gather jobs: the next step's gather+mean launches (they need no weights) ride along inside another
// launch as extra workgroups -- every launch of a training step that leaves HBM idle can carry a share of them.
#define GS_MAX_COJOBS_S 6
struct CoGatherS {
    GatherArgs job[GS_MAX_COJOBS_S];
    int64_t wave_start[GS_MAX_COJOBS_S + 1];   // prefix sums of work items (waves) per job
    int32_t n;
};

// U: independent 16-byte loads a wave keeps in flight.  8 suits launches whose gather waves reach >= 12 per CU; hosts
// that give a gather wave a large register budget but few wave slots (the tail launch: 8 waves per CU) use 13.
// UBIG: loads in flight for jobs with s >= UBIG rows per group.  A wave sums its rows batch by batch (U loads, wait, add),
// so an item is 1 + ceil(s / U) dependent memory round trips: ~10 us for s = 25 at U = 8, which only matters where the
// host launch is shorter than that (the 9 us optimizer launch: UBIG = 25 makes the item 2 round trips).
template <int U = 8, int UBIG = U>
__device__ __forceinline__ void run_gather_item(const CoGatherS& J, const int64_t w, const int lane) {
    if (w >= J.wave_start[J.n]) return;  // wave-uniform
    int k = 0;
    while (k + 1 < J.n && w >= J.wave_start[k + 1]) ++k;
    const GatherArgs& a = J.job[k];
    if (UBIG != U && a.s >= UBIG)
        gather_mean_wave<UBIG>(a, w - J.wave_start[k], lane);
    else if (a.s >= 8)
        gather_mean_wave<U>(a, w - J.wave_start[k], lane);
    else
        gather_mean_wave<1>(a, w - J.wave_start[k], lane);
}

static inline int gs_rup4(int x) { return (x + 3) & ~3; }

static inline int build_cojobs_s(const gs_gather_desc* jobs_host, int32_t n_jobs, CoGatherS* Jout, int64_t* waves_out) {
    GS_REQUIRE(n_jobs >= 0 && n_jobs <= GS_MAX_COJOBS_S && (n_jobs == 0 || jobs_host), "co-gather: 0..%d jobs", GS_MAX_COJOBS_S);
    CoGatherS& J = *Jout;
    J.n = n_jobs;
    int64_t waves = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const gs_gather_desc& q = jobs_host[i];
        GS_CHECK_MAT(q.X, q.ldx, "co-gather job X");
        GS_CHECK_MAT(q.out, q.ldo, "co-gather job out");
        GS_REQUIRE(q.n > 0 && q.s > 0 && q.d > 0 && q.ldx >= gs_rup4(q.d) && q.ldo >= gs_rup4(q.d), "co-gather: bad job %d", i);
        if (q.self_src) GS_CHECK_MAT(q.self_src, q.ld_self, "co-gather job self");
        const int chunks = ((q.d + 3) / 4 + 63) / 64;
        J.job[i] = GatherArgs{q.X, q.ldx, q.idx, q.n, q.s, q.d, q.self_src, q.ld_self, q.self_idx, q.out, q.ldo,
                              q.self_src ? 1.0f / (float)(q.s + 1) : 1.0f / (float)q.s, chunks,
                              DropArgs{0ull, nullptr, 0u, 0u, 1.0f, 0, nullptr, 0}};
        J.wave_start[i] = waves;
        waves += q.n * (int64_t)chunks;
    }
    J.wave_start[n_jobs] = waves;
    *waves_out = waves;
    return GS_OK;
}
