// Probe: issue cost of fp32 MFMA (v_mfma_f32_32x32x2_f32) on gfx950 -- alone, and with independent VALU work and
// global loads (results never consumed inside the loop) either interleaved between the MFMAs or lumped behind a group
// of four.  1..3 waves per SIMD.  Prints SIMD cycles per MFMA = slowest wave's cycles / (MFMAs per wave x waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 benchmarks/probes/mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: interleaved (MFMA, VALU x V, load x L) x 4     MODE 1: lumped (MFMA x 4, then VALU x 4V, load x 4L)
template <int MODE, int VALU, int LOADS>
__global__ void probe(const float* __restrict__ src, float* out, long long* cycles, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = threadIdx.x * 0.5f, b = threadIdx.x * 0.25f;
    unsigned off = (threadIdx.x & 63) * 4;
    float x[4] = {1.f, 2.f, 3.f, 4.f};
    float ld[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long long t0 = clock64();
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < VALU; ++v) x[v & 3] = x[v & 3] * 1.0001f + 0.5f;
                    if (LOADS) { ld[u * 4 + j] = *(const float*)((const char*)src + off); off = (off + 256) & 0xFFFFF; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int v = 0; v < VALU; ++v) x[v & 3] = x[v & 3] * 1.0001f + 0.5f;
                    if (LOADS) { ld[u * 4 + j] = *(const float*)((const char*)src + off); off = (off + 256) & 0xFFFFF; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = clock64();
    float s = x[0] + x[1] + x[2] + x[3];
    for (int k = 0; k < 8; ++k) s += ld[k];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE, int VALU, int LOADS>
void run(const char* name, int threads, int blocks) {
    const int iters = 2000;
    float *src, *out; long long* cyc;
    (void)hipMalloc(&src, 64 << 20); (void)hipMemset(src, 0, 64 << 20);
    (void)hipMalloc(&out, blocks * threads * 4); (void)hipMalloc(&cyc, blocks * 16 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, VALU, LOADS>), dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long h[16];
    (void)hipMemcpy(h, cyc, sizeof(long long) * (threads / 64), hipMemcpyDeviceToHost);
    long long mx = 0, mn = 1ll << 62;
    for (int w = 0; w < threads / 64; ++w) { mx = h[w] > mx ? h[w] : mx; mn = h[w] < mn ? h[w] : mn; }
    const int wps = threads / 256;
    printf("%-52s waves/SIMD %d blocks %4d: SIMD cycles per MFMA %6.1f (slowest wave; fastest wave alone would give %6.1f)\n", name, wps,
           blocks, (double)mx / (iters * 4.0 * wps), (double)mn / (iters * 4.0 * wps));
    (void)hipFree(src); (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int t = 256; t <= 768; t += 256) {
        run<0, 0, 0>("mfma only", t, 1);
        run<0, 1, 0>("interleaved: 1 VALU per MFMA", t, 1);
        run<1, 1, 0>("lumped:      4 VALU per 4 MFMA", t, 1);
        run<0, 2, 0>("interleaved: 2 VALU per MFMA", t, 1);
        run<1, 2, 0>("lumped:      8 VALU per 4 MFMA", t, 1);
        run<0, 0, 1>("interleaved: 1 load (+2 VALU addr) per MFMA", t, 1);
        run<1, 0, 1>("lumped:      4 loads (+8 VALU addr) per 4 MFMA", t, 1);
        run<0, 2, 1>("interleaved: 2 VALU + 1 load per MFMA", t, 1);
        run<1, 2, 1>("lumped:      8 VALU + 4 loads per 4 MFMA", t, 1);
    }
    run<1, 2, 1>("lumped: 8 VALU + 4 loads per 4 MFMA, all CUs", 256, 256);
    run<1, 2, 1>("lumped: 8 VALU + 4 loads per 4 MFMA, all CUs", 512, 256);
    return 0;
}
