// Probe: does the fp32 MFMA issue rate depend on the number of independent accumulator chains and on where the A/B
// operands live?  One workgroup of 256 threads (one wave per SIMD), v_mfma_f32_32x32x2_f32 only.
//   hipcc -O3 --offload-arch=gfx950 benchmarks/probes/mfma_chains.hip -o benchmarks/probes/mfma_chains.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// CH accumulator chains used round-robin; NB distinct B registers (1 = the same register every time); LD: one 16-byte
// global load per 8 MFMAs through a ring of 8 (consumed 8 groups later, like the stream kernels)
template <int CH, int NB, int LD>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* out, long long* cycles, int iters) {
    f32x16 acc[CH];
    for (int j = 0; j < CH; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float b[NB];
    for (int k = 0; k < NB; ++k) b[k] = threadIdx.x * 0.25f + k;
    f32x4 a[8];
    for (int k = 0; k < 8; ++k) a[k] = f32x4{threadIdx.x * 0.5f, 1.f + k, 2.f, 3.f};
    const float* p = src + (threadIdx.x & 63) * 4;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {          // one iteration = 8 groups of 8 MFMAs
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int idx = g * 8 + m;
                acc[idx % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][m >> 1], b[idx % NB], acc[idx % CH], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (LD) { a[g] = *(const f32x4*)(p + ((i * 8 + g) & 1023) * 256); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += a[k].x;
    for (int j = 0; j < CH; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cycles[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int CH, int NB, int LD>
void run(const char* name, int blocks) {
    const int iters = 400;
    float *src, *out; long long* cyc;
    (void)hipMalloc(&src, 8 << 20); (void)hipMemset(src, 0, 8 << 20);
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&cyc, blocks * 4 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<CH, NB, LD>), dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long h[4];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < 4; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("%-70s blocks %4d: cycles per MFMA %6.1f\n", name, blocks, (double)mx / (iters * 64.0));
    (void)hipFree(src); (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<4, 1, 0>("4 chains, one B register", 1);
    run<2, 1, 0>("2 chains, one B register", 1);
    run<1, 1, 0>("1 chain,  one B register", 1);
    run<2, 16, 0>("2 chains, 16 B registers", 1);
    run<2, 64, 0>("2 chains, 64 B registers", 1);
    run<4, 64, 0>("4 chains, 64 B registers", 1);
    run<2, 64, 1>("2 chains, 64 B registers, one 16-byte load per 8 MFMAs (ring 8)", 1);
    run<4, 64, 1>("4 chains, 64 B registers, one 16-byte load per 8 MFMAs (ring 8)", 1);
    run<2, 64, 1>("2 chains, 64 B registers, one load per 8 MFMAs, all CUs", 256);
    run<4, 64, 1>("4 chains, 64 B registers, one load per 8 MFMAs, all CUs", 256);
    return 0;
}
