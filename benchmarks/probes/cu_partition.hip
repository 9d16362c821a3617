// Probe (round 4): can the gather (K2, HBM-bound) and the fp32 MFMA work of a step share the CHIP without sharing SIMDs?
// One launch of persistent workgroups; a workgroup reads its CU id (HW_ID) and takes a role:
//   gather role  pulls (row, chunk) items of the production K2 body (gather_mean_wave<8>) from an atomic queue
//   mfma role    pulls items of `mfma_per_item` v_mfma_f32_32x32x2_f32 (register operands: pure matrix-pipe work)
// Modes:  A  gather on every CU                     B  gather only on CUs with cu_id < g (the others idle)
//         C  gather on cu_id < g, MFMA on the rest   D  MFMA on every CU            E  MFMA on cu_id >= g only
//         F  both roles on every CU (role by workgroup parity: the waves share SIMDs -- today's rider situation)
// Reports per role: first start -> last end (s_memrealtime), i.e. how long each stream took inside the shared launch.
//   hipcc -O3 --offload-arch=gfx950 -I include -I graphsage_amd/csrc benchmarks/probes/cu_partition.hip -o /tmp/cu_partition
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "gs_gather_dev.h"

struct PArgs {
    GatherArgs g;
    long long n_gather_items, n_mfma_items;
    int mfma_per_item;
    int mode;          // 0 A, 1 B, 2 C, 3 D, 4 E, 5 F
    int g_cus;         // gather CUs per shader engine: cu_id < g_cus
    unsigned long long* q;      // [0] gather queue, [1] mfma queue
    unsigned long long* t;      // [0] gather min start, [1] gather max end, [2] mfma min start, [3] mfma max end
    unsigned* cu_seen;          // [xcc * 256 + (se, sh, cu)] role marks
    float* sink;
};

__device__ __forceinline__ unsigned long long wall() { return wall_clock64(); }

__global__ __launch_bounds__(256) void part_kernel(const PArgs a) {
    const int lane = threadIdx.x & 63;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
    const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    int role;  // 0 gather, 1 mfma, -1 none
    switch (a.mode) {
        case 0: role = 0; break;
        case 1: role = (int)cu < a.g_cus ? 0 : -1; break;
        case 2: role = (int)cu < a.g_cus ? 0 : 1; break;
        case 3: role = 1; break;
        case 4: role = (int)cu < a.g_cus ? -1 : 1; break;
        default: role = (int)(blockIdx.x & 1); break;
    }
    if (role < 0) return;
    if (threadIdx.x == 0) atomicOr(&a.cu_seen[xcc * 256 + se * 32 + sh * 16 + cu], 1u << role);
    unsigned long long t0 = wall();
    // one queue per (role, XCD) on its own cache line; a WORKGROUP claims 4 items per atomic (one per wave): a device-scope
    // atomic on ONE address costs 20-50 ns serialised (first version of this probe: one claim per item on one address =
    // 15360 claims = 310 us), and claiming more per wave leaves most resident waves without work
    __shared__ unsigned claim;
    const int wave = threadIdx.x >> 6;
    bool any = false;
    if (role == 0) {
        const long long per = (a.n_gather_items + 7) / 8, lo = per * xcc, hi = lo + per < a.n_gather_items ? lo + per : a.n_gather_items;
        for (;;) {
            if (threadIdx.x == 0) claim = (unsigned)atomicAdd(&a.q[xcc * 16], 4ull);
            __syncthreads();
            const long long w = lo + claim + wave;
            __syncthreads();
            if (w - wave >= hi) break;
            if (w < hi) { any = true; gather_mean_wave<8>(a.g, (int64_t)w, lane); }
        }
        if (any && lane == 0) { atomicMin(&a.t[0], t0); atomicMax(&a.t[1], wall()); }
    } else {
        f32x16 acc[2];
        for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const float av = threadIdx.x * 0.5f, bv = threadIdx.x * 0.25f;
        const long long per = (a.n_mfma_items + 7) / 8, lo = per * xcc, hi = lo + per < a.n_mfma_items ? lo + per : a.n_mfma_items;
        for (;;) {
            if (threadIdx.x == 0) claim = (unsigned)atomicAdd(&a.q[(8 + xcc) * 16], 4ull);
            __syncthreads();
            const long long w = lo + claim + wave;
            __syncthreads();
            if (w - wave >= hi) break;
            if (w < hi) {
                any = true;
                for (int i = 0; i < a.mfma_per_item; i += 2) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[1], 0, 0, 0);
                }
            }
        }
        if (any) {
            float s = 0.f;
            for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
            a.sink[(blockIdx.x * 256 + threadIdx.x) & 0xFFFF] = s;
            if (lane == 0) { atomicMin(&a.t[2], t0); atomicMax(&a.t[3], wall()); }
        }
    }
}

__global__ void reset_kernel(unsigned long long* q, unsigned long long* t, unsigned* seen) {
    if (threadIdx.x < 16) q[threadIdx.x * 16] = 0ull;
    if (threadIdx.x < 4) t[threadIdx.x] = (threadIdx.x & 1) ? 0ull : ~0ull;
    for (int i = threadIdx.x; i < 16 * 256; i += blockDim.x) seen[i] = 0u;
}

int main(int argc, char** argv) {
    const long long N = 232965, LD = 608, n = 5120, s = 25, d = 602;
    const double mfma_gf = argc > 1 ? atof(argv[1]) : 3.64;   // fp32 MFMA work per launch (forward + weight gradients of a step)
    const int per_item = 128;
    float *X, *out, *sink; int32_t* idx; unsigned long long *q, *t; unsigned* seen;
    hipMalloc(&X, (N + 1) * LD * 4); hipMalloc(&out, n * LD * 4); hipMalloc(&sink, 65536 * 4);
    const int REPS = 40;
    hipMalloc(&idx, n * s * 4); hipMalloc(&q, 16 * 16 * 8); hipMalloc(&t, 32 * REPS); hipMalloc(&seen, 16 * 256 * 4);
    hipMemset(X, 0, (N + 1) * LD * 4);
    std::vector<int32_t> h(n * s);
    unsigned long long st = 88172645463325252ull;
    for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (int32_t)(st % (unsigned long long)N); }
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    PArgs a = {};
    const int chunks = 3;
    a.g = GatherArgs{X, LD, idx, n, (int)s, (int)d, nullptr, 0, nullptr, out, LD, 1.0f / s, chunks, DropArgs{0ull, nullptr, 0u, 0u, 1.0f, 0}};
    a.n_gather_items = n * chunks;
    a.mfma_per_item = per_item;
    a.n_mfma_items = (long long)(mfma_gf * 1e9 / (4096.0 * per_item));
    a.q = q; a.t = t; a.cu_seen = seen; a.sink = sink;
    const double alg_bytes = (double)n * s * d * 4 + n * s * 4 + n * d * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("gather: %lld items, %.1f MB algorithmic;  MFMA: %lld items x %d = %.2f GF\n", a.n_gather_items, alg_bytes / 1e6,
           a.n_mfma_items, per_item, a.n_mfma_items * 4096.0 * per_item / 1e9);
    printf("%-44s %9s %12s %9s %12s %10s %s\n", "mode", "total us", "gather us", "TB/s", "mfma us", "TF", "CUs(g/m)");
    struct Cfg { int mode, g, wgs; const char* name; };
    std::vector<Cfg> cfgs;
    for (int wgs : {2048, 4096}) {
        cfgs.push_back({0, 0, wgs, "A gather, all CUs"});
        for (int g : {6, 5, 4, 3, 2}) cfgs.push_back({1, g, wgs, "B gather on cu_id<g, rest idle"});
        cfgs.push_back({3, 0, wgs, "D mfma, all CUs"});
        for (int g : {4, 3, 2}) cfgs.push_back({4, g, wgs, "E mfma on cu_id>=g, rest idle"});
        for (int g : {6, 5, 4, 3, 2}) cfgs.push_back({2, g, wgs, "C gather cu_id<g | mfma rest"});
        cfgs.push_back({5, 0, wgs, "F both roles on every CU (shared SIMDs)"});
    }
    for (auto& c : cfgs) {
        a.mode = c.mode; a.g_cus = c.g;
        double best_total = 1e30, g_us = 0, m_us = 0; int ng = 0, nm = 0;
        // back-to-back launches (no host synchronisation in between: the clocks stay up), the fastest of the last 30 counts
        std::vector<hipEvent_t> ev0(REPS), ev1(REPS);
        for (int rep = 0; rep < REPS; ++rep) { hipEventCreate(&ev0[rep]); hipEventCreate(&ev1[rep]); }
        for (int rep = 0; rep < REPS; ++rep) {
            PArgs b = a;
            b.t = t + 4 * rep;
            hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(256), 0, 0, q, b.t, seen);
            hipEventRecord(ev0[rep]);
            hipLaunchKernelGGL(part_kernel, dim3(c.wgs), dim3(256), 0, 0, b);
            hipEventRecord(ev1[rep]);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> th(4 * REPS); hipMemcpy(th.data(), t, 32 * REPS, hipMemcpyDeviceToHost);
        std::vector<unsigned> sv(16 * 256); hipMemcpy(sv.data(), seen, sv.size() * 4, hipMemcpyDeviceToHost);
        for (unsigned v : sv) { ng += (v & 1) != 0; nm += (v & 2) != 0; }
        for (int rep = 10; rep < REPS; ++rep) {
            float ms; hipEventElapsedTime(&ms, ev0[rep], ev1[rep]);
            if (ms * 1e3 < best_total) {
                best_total = ms * 1e3;
                const unsigned long long* x = &th[4 * rep];
                g_us = x[1] > x[0] && x[0] != ~0ull ? (x[1] - x[0]) / 100.0 : 0;    // 100 MHz
                m_us = x[3] > x[2] && x[2] != ~0ull ? (x[3] - x[2]) / 100.0 : 0;
            }
        }
        for (int rep = 0; rep < REPS; ++rep) { hipEventDestroy(ev0[rep]); hipEventDestroy(ev1[rep]); }
        char nm_[96]; snprintf(nm_, sizeof nm_, "%s g=%d wgs=%d", c.name, c.g, c.wgs);
        printf("%-44s %9.1f %12.1f %9.2f %12.1f %10.1f %d/%d\n", nm_, best_total, g_us, g_us > 0 ? alg_bytes / g_us / 1e6 : 0.0, m_us,
               m_us > 0 ? a.n_mfma_items * 4096.0 * per_item / m_us / 1e6 : 0.0, ng, nm);
    }
    return 0;
}
