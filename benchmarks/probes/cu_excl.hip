// Probe (round 5): CU-EXCLUSIVE co-scheduling of the gather (K2, HBM-bound) and fp32 MFMA work inside ONE launch, with no
// atomics and no CU masks.  Every workgroup of the launch requests > 80 KB of LDS, so exactly ONE workgroup is resident
// per CU; roles go by block index (dispatch order): the first H workgroups are persistent MFMA hosts (they own H CUs for
// the whole launch), the remaining workgroups are gather workgroups that walk the items with a stride (persistent) --
// they own the other CUs.  Round 4's probes answered "same SIMD" (negative value) and "atomic work queues" (300 ns per
// claim); this one answers:
//   (1) how fast does K2 stream when it owns only N of the 256 CUs (8 waves per CU, U loads in flight per lane)?
//   (2) do MFMA hosts on their own CUs keep their rate while the gather saturates HBM from the other CUs -- with register
//       operands, and with operands streamed from L2 (the contraction kernels' situation)?
//   hipcc -O3 --offload-arch=gfx950 -I include -I graphsage_amd/csrc benchmarks/probes/cu_excl.hip -o /tmp/cu_excl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "gs_gather_dev.h"

struct XArgs {
    GatherArgs g;
    long long n_items;
    int H;              // MFMA host workgroups (block index < H)
    int mfma_iters;     // per wave: iterations of 8 MFMAs
    int feed;           // 0 register operands, 1 operands streamed from `feedbuf` (6 dword loads per 8 MFMAs)
    const float* feedbuf;
    long long feed_floats;
    unsigned long long* t;   // [0] gather min start [1] gather max end [2] mfma min start [3] mfma max end
    float* sink;
};

template <int U, int WAVES, bool MF>
__global__ __launch_bounds__(WAVES * 64) void excl_kernel(const XArgs a) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = wall_clock64();
    if (MF && (int)blockIdx.x < a.H) {
        // MFMA host: 64 x 128 tile per wave = 2 A fragments x 4 B fragments = 8 MFMAs per k-pair
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        if (wave < 8) {
            if (!a.feed) {
                const float av0 = threadIdx.x * 0.5f, av1 = threadIdx.x * 0.125f, bv = threadIdx.x * 0.25f;
                for (int i = 0; i < a.mfma_iters; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv + j, acc[j], 0, 0, 0);
                        acc[4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv + j, acc[4 + j], 0, 0, 0);
                    }
                }
            } else {
                // every iteration: 2 "A" dwords (rows of a 608-float matrix, this wave's own row range) + 4 "B" dwords (a
                // 128-column weight panel shared by everybody), ring of 4 iterations
                constexpr int P = 4;
                const long long rowsA = a.feed_floats / 2 / 608;
                const float* A = a.feedbuf;
                const float* B = a.feedbuf + a.feed_floats / 2;
                long long ra = ((long long)(blockIdx.x * 8 + wave) * 1031) % (rowsA - 2);
                int kb = 0;
                float av[P][2], bv[P][4];
                auto load = [&](int st) {
                    av[st][0] = A[ra * 608 + (lane & 31) + 608 * (lane >> 5)];
                    av[st][1] = A[ra * 608 + 32 + (lane & 31) + 608 * (lane >> 5)];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[st][j] = B[(kb + (lane >> 5)) * 128 + 32 * j + (lane & 31)];
                    ra += 2; if (ra >= rowsA - 2) ra = 0;
                    kb += 2; if (kb >= 600) kb = 0;
                };
#pragma unroll
                for (int st = 0; st < P; ++st) load(st);
                for (int i = 0; i < a.mfma_iters; i += P) {
#pragma unroll
                    for (int st = 0; st < P; ++st) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st][0], bv[st][j], acc[j], 0, 0, 0);
                            acc[4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st][1], bv[st][j], acc[4 + j], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        load(st);
                    }
                }
            }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[j][e];
            a.sink[(blockIdx.x * 512 + threadIdx.x) & 0xFFFF] = s;
            if (lane == 0) { atomicMin(&a.t[2], t0); atomicMax(&a.t[3], wall_clock64()); }
        }
        return;
    }
    // gather workgroup: persistent, strided over the items
    const long long nG = (long long)gridDim.x - a.H;
    bool any = false;
    for (long long w = ((long long)blockIdx.x - a.H) * WAVES + wave; w < a.n_items; w += nG * WAVES) {
        gather_mean_wave<U>(a.g, (int64_t)w, lane);
        any = true;
    }
    if (any && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); atomicMin(&a.t[0], t0); atomicMax(&a.t[1], wall_clock64()); }
    if (pad[threadIdx.x] == 12345.f) a.sink[0] = 1.f;     // keeps the LDS allocation
}

__global__ void reset_kernel(unsigned long long* t) {
    if (threadIdx.x < 4) t[threadIdx.x] = (threadIdx.x & 1) ? 0ull : ~0ull;
}

template <int U, int WAVES, bool MF = true>
static void run(const char* name, XArgs a, int grid, double alg_bytes, double mfma_flops) {
    const int REPS = 24;
    unsigned long long* t;
    (void)hipMalloc(&t, 32 * REPS);
    static bool attr = false;
    (void)hipFuncSetAttribute((const void*)excl_kernel<U, WAVES, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)attr;
    std::vector<hipEvent_t> e0(REPS), e1(REPS);
    for (int r = 0; r < REPS; ++r) { (void)hipEventCreate(&e0[r]); (void)hipEventCreate(&e1[r]); }
    for (int r = 0; r < REPS; ++r) {
        XArgs b = a; b.t = t + 4 * r;
        hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(64), 0, 0, b.t);
        (void)hipEventRecord(e0[r]);
        hipLaunchKernelGGL((excl_kernel<U, WAVES, MF>), dim3(grid), dim3(WAVES * 64), 100 * 1024, 0, b);
        (void)hipEventRecord(e1[r]);
    }
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> th(4 * REPS);
    (void)hipMemcpy(th.data(), t, 32 * REPS, hipMemcpyDeviceToHost);
    std::vector<double> tot, gs, ms;
    for (int r = 6; r < REPS; ++r) {
        float msx; (void)hipEventElapsedTime(&msx, e0[r], e1[r]);
        const unsigned long long* x = &th[4 * r];
        tot.push_back(msx * 1e3);
        gs.push_back(x[1] > x[0] && x[0] != ~0ull ? (x[1] - x[0]) / 100.0 : 0);
        ms.push_back(x[3] > x[2] && x[2] != ~0ull ? (x[3] - x[2]) / 100.0 : 0);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double g = med(gs), m = med(ms);
    printf("%-58s grid %4d H %3d | total %7.1f us | gather %7.1f us %5.2f TB/s | mfma %7.1f us %6.1f TF\n", name, grid, a.H, med(tot),
           g, g > 0 ? alg_bytes / g / 1e6 : 0.0, m, m > 0 ? mfma_flops / m / 1e6 : 0.0);
    for (int r = 0; r < REPS; ++r) { (void)hipEventDestroy(e0[r]); (void)hipEventDestroy(e1[r]); }
    (void)hipFree(t);
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const long long N = 232965, LD = 608, n = 5120, s = 25, d = 602;
    float *X, *out, *sink, *feed; int32_t* idx;
    (void)hipMalloc(&X, (N + 1) * LD * 4); (void)hipMalloc(&out, n * LD * 4); (void)hipMalloc(&sink, 65536 * 4);
    (void)hipMalloc(&idx, n * s * 4);
    const long long feed_floats = 2ll * 5632 * 608;          // "A" half: 5632 rows x 608 (13.7 MB); "B" half: weights
    (void)hipMalloc(&feed, feed_floats * 4);
    (void)hipMemset(X, 0, (N + 1) * LD * 4);
    (void)hipMemset(feed, 0, feed_floats * 4);
    std::vector<int32_t> h(n * s);
    unsigned long long st = 88172645463325252ull;
    for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (int32_t)(st % (unsigned long long)N); }
    (void)hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    XArgs a = {};
    const int chunks = 3;
    a.g = GatherArgs{X, LD, idx, n, (int)s, (int)d, nullptr, 0, nullptr, out, LD, 1.0f / s, chunks, DropArgs{0ull, nullptr, 0u, 0u, 1.0f, 0, nullptr, 0}};
    a.n_items = n * chunks;
    a.sink = sink; a.feedbuf = feed; a.feed_floats = feed_floats;
    const double alg = (double)n * s * d * 4 + n * s * 4 + n * d * 4;
    // MFMA work of one step's two contractions: 3.64 GF = 888k MFMAs = 111k iterations of 8
    const long long total_iters = 111000;
    char nm[128];
    printf("== (1) gather alone on N CUs (one workgroup per CU, persistent)\n");
    for (int N_cu : {256, 224, 192, 160, 128, 96, 64}) {
        a.H = 0; a.mfma_iters = 0;
        snprintf(nm, sizeof nm, "gather only, 8 waves/CU U=13"); run<13, 8>(nm, a, N_cu, alg, 0);
        snprintf(nm, sizeof nm, "gather only, 8 waves/CU U=25"); run<25, 8>(nm, a, N_cu, alg, 0);
        snprintf(nm, sizeof nm, "gather only, 16 waves/CU U=13"); run<13, 16, false>(nm, a, N_cu, alg, 0);
        snprintf(nm, sizeof nm, "gather only, 16 waves/CU U=25"); run<25, 16, false>(nm, a, N_cu, alg, 0);
    }
    printf("== (2) MFMA alone on H CUs (8 waves per CU, 64x128 tile per wave: 8 MFMAs per k-pair)\n");
    for (int feedm : {0, 1})
        for (int H : {256, 192, 160, 128, 96}) {
            a.H = H; a.feed = feedm; a.n_items = 0;
            a.mfma_iters = (int)(total_iters / (H * 8)) / 4 * 4;
            snprintf(nm, sizeof nm, "mfma only, %s", feedm ? "L2-fed (6 dword loads / 8 MFMAs)" : "register operands");
            run<13, 8>(nm, a, H, alg, (double)a.mfma_iters * 8 * 4096.0 * H * 8);
        }
    a.n_items = n * chunks;
    printf("== (3) both in one launch: MFMA on H CUs | gather on the other 256 - H (CU-exclusive)\n");
    for (int feedm : {0, 1})
        for (int H : {192, 160, 128, 96, 64}) {
            a.H = H; a.feed = feedm;
            a.mfma_iters = (int)(total_iters / (H * 8)) / 4 * 4;
            const double fl = (double)a.mfma_iters * 8 * 4096.0 * H * 8;
            snprintf(nm, sizeof nm, "both, %s, gather 8w U=25", feedm ? "L2-fed" : "reg"); run<25, 8>(nm, a, 256, alg, fl);
            snprintf(nm, sizeof nm, "both, %s, gather 8w U=13", feedm ? "L2-fed" : "reg"); run<13, 8>(nm, a, 256, alg, fl);
        }
    printf("== (4) reference: both in one launch sharing CUs is today's rider situation (see r04 probes)\n");
    return 0;
}
