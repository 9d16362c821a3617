// Probe (round 4): the gather (K2, HBM-bound) and fp32 MFMA work on DISJOINT CU sets of one MI355X, through two streams
// created with hipExtStreamCreateWithCUMask -- does the chip overlap them cleanly when they do not share SIMDs?
// (benchmarks/probes/cu_partition.hip tried the same inside ONE launch with role-by-CU-id workgroups and atomic work queues:
//  a device-scope atomic on one address serialises at ~300 ns on this chip, so the queues themselves were the bottleneck.)
// Kernels write per-workgroup start/end stamps (s_memrealtime, 100 MHz) and the CU they ran on; the host reduces them.
//   hipcc -O3 --offload-arch=gfx950 -I include -I graphsage_amd/csrc benchmarks/probes/cu_mask_streams.hip -o /tmp/cu_mask
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <set>
#include "gs_gather_dev.h"

__device__ __forceinline__ unsigned hw_cu() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
    return (xcc << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
}

__global__ __launch_bounds__(256) void gather_kernel(const GatherArgs a, unsigned long long* stamps, unsigned* cus) {
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = wall_clock64();
    const int64_t n_items = a.n * (int64_t)a.chunks;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w < n_items) gather_mean_wave<8>(a, w, lane);
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); cus[blockIdx.x] = hw_cu(); }
}

__global__ __launch_bounds__(256) void mfma_kernel(int per_wave, float* sink, unsigned long long* stamps, unsigned* cus) {
    const unsigned long long t0 = wall_clock64();
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const float av = threadIdx.x * 0.5f, bv = threadIdx.x * 0.25f;
    for (int i = 0; i < per_wave; i += 2) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[1], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    sink[(blockIdx.x * 256 + threadIdx.x) & 0xFFFF] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); cus[blockIdx.x] = hw_cu(); }
}

struct Span { double us; int cus; };
static Span span(unsigned long long* d_st, unsigned* d_cu, int n) {
    std::vector<unsigned long long> st(2 * n); std::vector<unsigned> cu(n);
    (void)hipMemcpy(st.data(), d_st, 16 * n, hipMemcpyDeviceToHost); (void)hipMemcpy(cu.data(), d_cu, 4 * n, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0; std::set<unsigned> s;
    for (int i = 0; i < n; ++i) { lo = std::min(lo, st[2 * i]); hi = std::max(hi, st[2 * i + 1]); s.insert(cu[i]); }
    return {(hi - lo) / 100.0, (int)s.size()};
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const long long N = 232965, LD = 608, n = 5120, s = 25, d = 602;
    const int chunks = 3, gwgs = (int)(n * chunks / 4);
    float *X, *out, *sink; int32_t* idx; unsigned long long *gst, *mst; unsigned *gcu, *mcu;
    (void)hipMalloc(&X, (N + 1) * LD * 4); (void)hipMalloc(&out, n * LD * 4); (void)hipMalloc(&sink, 65536 * 4);
    (void)hipMalloc(&idx, n * s * 4); (void)hipMalloc(&gst, 16 * 8192); (void)hipMalloc(&mst, 16 * 8192);
    (void)hipMalloc(&gcu, 4 * 8192); (void)hipMalloc(&mcu, 4 * 8192);
    (void)hipMemset(X, 0, (N + 1) * LD * 4);
    std::vector<int32_t> h(n * s);
    unsigned long long st = 88172645463325252ull;
    for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (int32_t)(st % (unsigned long long)N); }
    (void)hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    GatherArgs g = {X, LD, idx, n, (int)s, (int)d, nullptr, 0, nullptr, out, LD, 1.0f / s, chunks, DropArgs{0ull, nullptr, 0u, 0u, 1.0f, 0}};
    const double alg_bytes = (double)n * s * d * 4 + n * s * 4 + n * d * 4;
    // MFMA work of one step's two contraction launches: 3.64 GF as 4096 waves (4 per SIMD) x 217 MFMAs
    const int mwgs = 1024, per_wave = 216;
    const double mflops = (double)mwgs * 4 * per_wave * 4096.0;
    printf("gather %.1f MB algorithmic in %d workgroups;  MFMA %.2f GF in %d workgroups\n", alg_bytes / 1e6, gwgs, mflops / 1e9, mwgs);
    printf("%-34s %10s %8s %6s %10s %8s %6s %10s\n", "configuration", "gather us", "TB/s", "CUs", "mfma us", "TF", "CUs", "both us");
    struct Cfg { int gcus; bool do_g, do_m; };
    std::vector<Cfg> cfgs = {{256, true, false}, {256, false, true}, {256, true, true}};
    for (int gc : {192, 160, 128, 96, 64}) { cfgs.push_back({gc, true, false}); cfgs.push_back({gc, false, true}); cfgs.push_back({gc, true, true}); }
    for (auto& c : cfgs) {
        hipStream_t sg, sm;
        uint32_t mg[8] = {0}, mm[8] = {0};
        // logical CU i of the mask: an interleaved pattern, so every XCD / shader engine contributes the same share
        for (int i = 0; i < 256; ++i) {
            const bool to_g = c.gcus >= 256 || ((long long)i * c.gcus / 256 != (long long)(i + 1) * c.gcus / 256);
            if (to_g) mg[i / 32] |= 1u << (i % 32);
            if (!to_g || c.gcus >= 256) mm[i / 32] |= 1u << (i % 32);
        }
        hipError_t r1 = hipExtStreamCreateWithCUMask(&sg, 8, mg);
        hipError_t r2 = hipExtStreamCreateWithCUMask(&sm, 8, mm);
        if (r1 != hipSuccess || r2 != hipSuccess) { printf("stream create failed: %s %s\n", hipGetErrorString(r1), hipGetErrorString(r2)); return 1; }
        printf("[cfg %d %d %d] ", c.gcus, (int)c.do_g, (int)c.do_m);
        double best = 1e30; Span bg = {0, 0}, bm = {0, 0};
        for (int rep = 0; rep < 30; ++rep) {
            hipEvent_t e0, e1, e2; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
            (void)hipEventRecord(e0, sg);
            if (c.do_m) (void)hipStreamWaitEvent(sm, e0, 0);
            if (c.do_g) hipLaunchKernelGGL(gather_kernel, dim3(gwgs), dim3(256), 0, sg, g, gst, gcu);
            if (c.do_m) {
                hipLaunchKernelGGL(mfma_kernel, dim3(mwgs), dim3(256), 0, sm, per_wave, sink, mst, mcu);
                (void)hipEventRecord(e2, sm); (void)hipStreamWaitEvent(sg, e2, 0);
            }
            (void)hipEventRecord(e1, sg);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 10 && ms * 1e3 < best) {
                best = ms * 1e3;
                if (c.do_g) bg = span(gst, gcu, gwgs);
                if (c.do_m) bm = span(mst, mcu, mwgs);
            }
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
        }
        char name[64];
        snprintf(name, sizeof name, "masks %d | %d CUs: %s", c.gcus, c.gcus >= 256 ? 256 : 256 - c.gcus,
                 c.do_g && c.do_m ? "both" : (c.do_g ? "gather alone" : "mfma alone"));
        printf("%-34s %10.1f %8.2f %6d %10.1f %8.1f %6d %10.1f\n", name, bg.us, bg.us > 0 ? alg_bytes / bg.us / 1e6 : 0.0, bg.cus, bm.us,
               bm.us > 0 ? mflops / bm.us / 1e6 : 0.0, bm.cus, best);
        (void)hipStreamDestroy(sg); (void)hipStreamDestroy(sm);
    }
    return 0;
}
