// Device-side body of the fused multi-hop fan-out sampler, shared by the standalone kernel (gs_sample.hip) and the
// optimizer launch that carries the NEXT-next step's sampler along (gs_optim.hip: the sampler is five dependent memory
// round trips of almost no work, so it hides entirely under another launch).
#pragma once
#include "gs_common.h"

#define GS_MAX_HOPS 3
#define GS_FANOUT_LDS 8192          // per-root ids of the kept hops, standalone kernel
#define GS_FANOUT_LDS_SMALL 512     // ... when the sampler rides in another launch (e.g. 10 for fan-out 25x10)
struct FanoutArgs {
    const int64_t* rowptr;
    const int32_t* col;
    int64_t n_nodes;
    int32_t pad_id;
    int32_t n_hops;
    int32_t fan[GS_MAX_HOPS];
    int64_t offsets[GS_MAX_HOPS + 1];  // start of each hop's ids inside ids_all (offsets[0] = roots)
    int32_t* ids_all;
    int64_t B;
    uint64_t seed, step;
    const uint64_t* step_dev;
    uint32_t hop0;
    int64_t root_offset;  // global index of this rank's first root (data-parallel invariance)
    // optional batch staging (order == nullptr -> roots are already in ids_all)
    const int32_t* order;
    int64_t n_order;
    const uint64_t* cursor;
    const float* label_table;
    int64_t ldt;
    int32_t C;
    float* labels_out;
    int64_t ldo;
};

// One workgroup (any size) per root `i`; lvl = two LDS fan-out buffers of CAP ints each.
template <int CAP>
__device__ __forceinline__ void sample_fanout_root(const FanoutArgs& a, const int64_t i, int32_t (*lvl)[CAP]) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    int32_t root;
    if (a.order) {
        const uint64_t c = a.cursor ? *a.cursor : 0ull;
        root = a.order[(int64_t)((c + (uint64_t)i) % (uint64_t)a.n_order)];
        if (tid == 0) a.ids_all[a.offsets[0] + i] = root;
        if (a.label_table) {
            const int Cp = (a.C + 3) & ~3;
            for (int k = tid; k < Cp; k += nthr)
                a.labels_out[i * a.ldo + k] = k < a.C ? a.label_table[(int64_t)root * a.ldt + k] : 0.f;
        }
    } else {
        root = a.ids_all[a.offsets[0] + i];
    }
    if (tid == 0) lvl[0][0] = root;
    __syncthreads();
    const uint64_t st = a.step + (a.step_dev ? *a.step_dev : 0ull);
    int64_t count_prev = 1;
    for (int h = 0; h < a.n_hops; ++h) {
        const int s = a.fan[h];
        const int64_t count = count_prev * s;
        const uint64_t key = gs_mix64(a.seed ^ (st * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(a.hop0 + h) << 56));
        const int32_t* prev = lvl[h & 1];
        int32_t* next = lvl[(h + 1) & 1];
        const bool keep = (h + 1 < a.n_hops);  // the last hop is only written to global memory
        for (int64_t t = tid; t < count; t += nthr) {
            const int64_t pl = t / s;  // parent slot in the previous level
            const uint32_t j = (uint32_t)(t - pl * s);
            const int32_t id = prev[pl];
            int32_t pick = a.pad_id;
            if (id >= 0 && (int64_t)id < a.n_nodes) {
                const int64_t b = a.rowptr[id];
                const int32_t deg = (int32_t)(a.rowptr[id + 1] - b);
                if (deg > 0) {
                    const int64_t grow = (a.root_offset + i) * count_prev + pl;  // global row at this hop
                    const uint64_t u = gs_mix64(key + (uint64_t)grow * 0xD1342543DE82EF95ull + j);
                    const uint32_t r = (uint32_t)(u >> 32);
                    pick = a.col[b + (int64_t)(((uint64_t)r * (uint64_t)(uint32_t)deg) >> 32)];
                }
            }
            if (keep) next[t] = pick;
            a.ids_all[a.offsets[h + 1] + i * count + t] = pick;
        }
        __syncthreads();
        count_prev = count;
    }
}

// host: C-ABI arguments -> FanoutArgs (shared validation); *kept_max = largest per-root count of a kept hop
static inline int gs_fanout_args(const int64_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t pad_id, int32_t n_hops,
                                 const int32_t* fan_host, const int64_t* offsets_host, int32_t* ids_all, int64_t B, uint64_t seed,
                                 uint64_t step, const uint64_t* step_dev, uint32_t hop0, int64_t root_offset,
                                 const int32_t* order, int64_t n_order, const uint64_t* cursor_dev, const float* label_table,
                                 int64_t ld_table, int32_t C, float* labels_out, int64_t ld_out, FanoutArgs* out,
                                 int64_t* kept_max) {
    GS_REQUIRE(rowptr && col && ids_all && fan_host && offsets_host && n_nodes > 0, "gs_sample_fanout_csr: null pointer");
    GS_REQUIRE(n_hops >= 1 && n_hops <= GS_MAX_HOPS, "gs_sample_fanout_csr: 1..%d hops", GS_MAX_HOPS);
    GS_REQUIRE(hop0 + n_hops <= 256, "gs_sample_fanout_csr: hop ids must be < 256");
    GS_REQUIRE(!order || n_order > 0, "gs_sample_fanout_csr: empty order");
    GS_REQUIRE(!label_table || (labels_out && C > 0 && ld_table >= C && ld_out >= ((C + 3) & ~3) && order),
               "gs_sample_fanout_csr: bad label staging arguments");
    FanoutArgs a = {};
    a.rowptr = rowptr; a.col = col; a.n_nodes = n_nodes; a.pad_id = pad_id; a.n_hops = n_hops;
    int64_t count = 1, kmax = 1;
    for (int h = 0; h < n_hops; ++h) {
        GS_REQUIRE(fan_host[h] > 0, "gs_sample_fanout_csr: fan-out must be positive");
        a.fan[h] = fan_host[h];
        if (h + 1 < n_hops) {
            count *= fan_host[h];
            kmax = std::max(kmax, count);
        }
    }
    for (int h = 0; h <= n_hops; ++h) a.offsets[h] = offsets_host[h];
    a.ids_all = ids_all; a.B = B; a.seed = seed; a.step = step; a.step_dev = step_dev; a.hop0 = hop0;
    a.root_offset = root_offset;
    a.order = order; a.n_order = n_order; a.cursor = cursor_dev;
    a.label_table = label_table; a.ldt = ld_table; a.C = C; a.labels_out = labels_out; a.ldo = ld_out;
    GS_REQUIRE(B < (1ll << 31), "gs_sample_fanout_csr: batch too large");
    *out = a;
    *kept_max = kmax;
    return GS_OK;
}
