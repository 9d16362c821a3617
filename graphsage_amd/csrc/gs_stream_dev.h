// Device-side body of the "stream" layer-0 forward contraction (gs_stream.hip: sage_stream_fwd_kernel), shared with the
// launch that runs layer 0 AND the fused tail as one kernel (gs_tail.hip: sage_tail_kernel<..., FWD = true>).
#pragma once
#include "gs_common.h"
#include "gs_gather_dev.h"

// XCD-aware work placement: block b runs on XCD b % 8 (each XCD has its own 4 MB L2).  Consecutive LOGICAL ids go to
// the same XCD, so a contiguous range of work items (= a contiguous range of rows / reduction slices) shares one L2:
// without it every XCD streams the whole A operand (13.7 MB) through its 4 MB L2 and every operand load is a MALL hit.
__device__ __forceinline__ int stream_xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

#if defined(GS_TIMELINE) && defined(GS_STREAM_TU)     // (the stamp buffer lives in gs_stream.hip's translation unit)
// Diagnostics build only (-DGS_TIMELINE, benchmarks/timeline_wgrad.py): per-wave wall-clock stamps (100 MHz).
__device__ unsigned long long g_timeline[32768 * 8];
extern "C" int gs_debug_timeline(unsigned long long* out_host, int n) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#define GS_STAMP(slot) do { if (lane == 0 && tl_item < 32768) { g_timeline[tl_item * 8 + (slot)] = wall_clock64(); \
    g_timeline[tl_item * 8 + 4 + (slot)] = (slot) == 0 ? (((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | \
        (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4)) : clock64(); } } while (0)
#else
#define GS_STAMP(slot) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------ forward
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FwdTerm {
    const float* A;        // [*, lda]; row i of the term is A[a_idx ? a_idx[i] : i]
    const int32_t* a_idx;  // nullable row gather (layer 0: the self rows of the feature table)
    const float* W;        // [K, ldw]
    int32_t lda, ldw;
    int32_t K;             // the term's reduction length (the pooling aggregators' self / pooled terms differ: 602 | 512)
};
struct FwdArgs {
    FwdTerm t[2];
    int32_t nterms;       // 1, or 2 (concat: term i writes columns [i*N, (i+1)*N))
    int32_t M, N, K;
    float* C;
    int32_t ldc;
    const float* bias;    // indexed by output column (incl. the concat offset), nullable
    int32_t act;
    int32_t tiles_n;      // 64-column tiles per term
    int32_t n_tiles;      // contraction workgroups: tiles_m * tiles_n * nterms
};

// host side (gs_stream.hip): argument checks + the kernel argument block
int stream_fwd_args(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self, const float* agg, int64_t ld_agg,
                    int32_t d, int64_t n, const float* W_self, int64_t ldw_self, const float* W_neigh, int64_t ldw_neigh,
                    int32_t out_dim, int act, const float* bias, float* out, int64_t ldo, FwdArgs* gout);

// One WORKGROUP (4 waves) owns one 32 x 64 output tile; wave w contracts a QUARTER of K (split-K inside the workgroup,
// summed in a fixed order through LDS, so the result does not depend on scheduling).
//   * Why split K: 5632 x 128 x 2 terms are only 704 32x64 tiles for 1024 SIMDs, and a whole-K tile holds a SIMD for
//     17.6 us of MFMA time (75 macro steps of 8 k x 8 MFMAs x 64 cycles) while the chip-wide average is 12.1 us.
//     Quarter-K waves (2816 of them, 2.75 per SIMD) bring the makespan to 3 x 4.4 us and give every SIMD a second and
//     third wave whose loads are in flight while the first one owns the matrix pipe.
//   * Fewer non-MFMA instructions (measured on the weight-gradient kernel: every instruction between two MFMAs is
//     paid in full): per macro step of 8 k a lane issues ONE 16-byte A load (its row, 4 consecutive k) and FOUR 8-byte
//     B loads (two adjacent columns of 4 k rows) for 8 MFMAs; B is addressed with four SGPR bases (W + e ldw) and one
//     32-bit offset.  The two n-tiles of the wave interleave their columns (col = n0 + 2 (lane & 31) + j), so a lane
//     ends up with two adjacent columns of a row: 8-byte loads, 8-byte stores.
//   * A rows may be gathered (a_idx): the row pointer is per lane and fixed for the whole K loop.
//
// PUBLISH (the one-launch layer 0 + tail, gs_tail.hip): the tile's rows are consumed by OTHER workgroups of the same launch.
// Its output leaves with device-scope (write-through) stores -- as the tail's z hand-over: a release FENCE would write back the
// XCD's whole L2, the riders' dirty gather output included -- and, once every wave's stores are acknowledged, one thread adds 1
// to the monotonic counter of the tile's 32-row block (`done[tile_m]`: tiles_n * nterms arrivals per launch).  `valid` = 0: a
// padding tile of the last workgroup (clamped to a real tile, computes, stores and publishes nothing).
template <int P, bool PUBLISH = false>
__device__ __forceinline__ void stream_fwd_tile(const FwdArgs& g, const int tile, const int wave, const int lane,
                                                float (*red)[32][64], const bool valid = true, uint32_t* done = nullptr) {
    const int tl_item = tile * 4 + wave; (void)tl_item;
    GS_STAMP(0);
    // (Issue priorities -- s_setprio 3 outside the MFMA loop, 0 inside, 2 for the gather waves -- were measured: a SIMD
    // serves its oldest wave first, so its three contraction waves run one after the other; with priorities they
    // interleave instead, at the same 24-26 us for the launch.  Left out.)
    const int l31 = lane & 31, lh = lane >> 5;
    const int per_term = g.n_tiles / g.nterms;
    const int term = tile / per_term;
    const int it = tile - term * per_term;
    const int tile_m = it / g.tiles_n, tile_n = it - tile_m * g.tiles_n;
    const int m0 = tile_m * 32, n0 = tile_n * 64;
    const FwdTerm& T = g.t[term];
    const int K = T.K, N = g.N;
    const int nfull = K >> 3;                              // macro steps whose 8 k are all < K
    const int mb = (nfull * wave) >> 2, me = (nfull * (wave + 1)) >> 2;   // this wave's macro steps
    const int arow = min(m0 + l31, g.M - 1);
    const int64_t srow = T.a_idx ? (int64_t)T.a_idx[arow] : (int64_t)arow;
    const float* ap = T.A + srow * T.lda + 4 * lh + 8 * mb;
    const int cl = min(n0 + 2 * l31, N - 2);               // the lane's column pair (clamped: never stored if >= N)
    const char* __restrict__ Wb = (const char*)T.W;
    const uint32_t ldw4 = (uint32_t)T.ldw * 4u;
    const char* __restrict__ W0 = Wb;                      // four uniform bases: rows 8 m + 4 lh + e, e = 0..3
    const char* __restrict__ W1 = Wb + ldw4;
    const char* __restrict__ W2 = Wb + 2 * ldw4;
    const char* __restrict__ W3 = Wb + 3 * ldw4;
    uint32_t wo = (uint32_t)(8 * mb + 4 * lh) * ldw4 + (uint32_t)cl * 4u;
    const uint32_t wstride = 8u * ldw4;
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    f32x4 a[P];
    f32x2 b[P][4];
    // `left`: pointer advances still allowed -- a refill past the wave's last macro step re-reads the last one (always a
    // valid address; the value is never consumed), so the pipeline needs no one-at-a-time remainder.
    int left = me - mb - 1;
    auto load_stage = [&](const int st) {
#ifdef GS_DIAG_FWD_NOA      // diagnostics builds only (benchmarks/probes/build_variant.sh): operand loads compiled out
        a[st] = f32x4{__int_as_float((int)(uintptr_t)ap), 0.f, 1.f, 2.f};
#else
        a[st] = *reinterpret_cast<const f32x4*>(ap);
#endif
#ifdef GS_DIAG_FWD_NOB
        b[st][0] = b[st][1] = b[st][2] = b[st][3] = f32x2{__int_as_float((int)wo), 1.f};
#else
        b[st][0] = *reinterpret_cast<const f32x2*>(W0 + wo);
        b[st][1] = *reinterpret_cast<const f32x2*>(W1 + wo);
        b[st][2] = *reinterpret_cast<const f32x2*>(W2 + wo);
        b[st][3] = *reinterpret_cast<const f32x2*>(W3 + wo);
#endif
        const int adv = left > 0 ? 1 : 0;                  // wave-uniform
        ap += 8 * adv;
        wo += wstride * (uint32_t)adv;
        --left;
    };
    auto compute_stage = [&](const int st) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = mfma32(a[st][e], b[st][e].x, acc0);
            acc1 = mfma32(a[st][e], b[st][e].y, acc1);
        }
    };
    // branch-free steady state: stages m .. m+P-1 are in flight on entry; each is consumed and refilled with m+st+P.
    // The last (possibly partial) group is consumed under wave-uniform guards -- no loads there.  (A first version ran
    // the count % P leftover macro steps one at a time, each with its whole load latency exposed: 2-3 round trips of the
    // 8.8 us a quarter-K wave took.  Also measured and dropped: a ring of 32-k super stages that consumes whole 128-byte
    // A lines back to back, 36.5 vs 24.3 us; ring depths 6 and 8, 28 us.)
    const int cnt = me - mb;
    if (cnt > 0) {
#pragma unroll
        for (int st = 0; st < P; ++st) load_stage(st);
        GS_STAMP(1);
        int m = 0;
#pragma unroll 1
        for (; m + P < cnt; m += P) {
#pragma unroll
            for (int st = 0; st < P; ++st) {
                compute_stage(st);
                __builtin_amdgcn_sched_barrier(0);         // the refill stays below the MFMAs that free its registers
                load_stage(st);
            }
        }
#pragma unroll
        for (int st = 0; st < P; ++st)
            if (m + st < cnt) compute_stage(st);
    }
    if (wave == 3 && (K & 7) != 0) {
        // tail macro step: k = 8*nfull + 4*lh + e; elements with k >= K are zeroed on the A side, B rows are clamped
        const int kq = 8 * nfull + 4 * lh;
        f32x4 av = {0.f, 0.f, 0.f, 0.f};
        if (kq < K) {                                      // the row's pad columns [K, round_up(K, 4)) are readable
            av = *reinterpret_cast<const f32x4*>(T.A + srow * T.lda + kq);
            if (kq + 1 >= K) av.y = 0.f;
            if (kq + 2 >= K) av.z = 0.f;
            if (kq + 3 >= K) av.w = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 bv = *reinterpret_cast<const f32x2*>(T.W + (int64_t)min(kq + e, K - 1) * T.ldw + cl);
            acc0 = mfma32(av[e], bv.x, acc0);
            acc1 = mfma32(av[e], bv.y, acc1);
        }
    }
    GS_STAMP(2);
    // split-K sum in a fixed order + bias + activation + store: wave w finishes elements e = 4w .. 4w+3 of both n-tiles
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        red[wave][e][lane] = acc0[e];
        red[wave][16 + e][lane] = acc1[e];
    }
    __syncthreads();
    const int col_off = term * N;
    const int c = n0 + 2 * l31;
    f32x2 bv = {0.f, 0.f};
    if (g.bias && c < N) bv = *reinterpret_cast<const f32x2*>(g.bias + col_off + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = 4 * wave + q;
        f32x2 v;
        v.x = ((red[0][e][lane] + red[1][e][lane]) + red[2][e][lane]) + red[3][e][lane] + bv.x;
        v.y = ((red[0][16 + e][lane] + red[1][16 + e][lane]) + red[2][16 + e][lane]) + red[3][16 + e][lane] + bv.y;
        if (g.act == GS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
        // C/D layout of the 32x32 MFMA: row = (e&3) + 8*(e>>2) + 4*(lane>>5); this lane's columns are c, c+1
        const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row < g.M && c < N && valid) {
            if (PUBLISH) {
                union { f32x2 f; unsigned long long u; } cv;
                cv.f = v;
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(g.C + (int64_t)row * g.ldc + col_off + c), cv.u,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                *reinterpret_cast<f32x2*>(g.C + (int64_t)row * g.ldc + col_off + c) = v;
            }
        }
    }
    if (PUBLISH) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's stores are acknowledged ...
        __syncthreads();                                             // ... and those of the tile's other waves
        if (wave == 0 && lane == 0 && valid) __hip_atomic_fetch_add(done + tile_m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    GS_STAMP(3);
}

