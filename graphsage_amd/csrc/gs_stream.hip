// "Stream" form of the two big launches of a mean/GCN training step: the layer-0 contraction and the grouped weight
// gradients, each co-scheduled with a share of the NEXT step's gather+mean (HBM-bound, needs no weights).
//
// Why a second GEMM form (measured, profiles/r02_*): in the LDS-tiled fused kernels every workgroup -- gather
// workgroups included -- carries the GEMM's 35 KB of LDS and ~110 VGPRs, the 352 tile workgroups land 1.4 per CU, and a
// tile workgroup stalls all four waves at a barrier per 32-k stage; the fused launches ran at 20 % MFMA utilisation.
// Here the contraction waves stage nothing through LDS and meet no barrier inside the K loop:
//   * A and B fragments go straight from global memory (L2 / MALL resident rows, weights, dZ) into the MFMA operand
//     registers through a 4..8-stage register ring, addressed as 32-bit offsets against SGPR base pointers: every
//     instruction between two MFMAs costs matrix-pipe time (benchmarks/probes/mfma_issue.hip: 64 cycles per MFMA alone,
//     72-80 with one or two VALU instructions per MFMA, at 1, 2 or 3 waves per SIMD alike), so the loops carry one
//     integer add per load and nothing else;
//   * row-gathered A operands (layer 0: the self rows X[ids]) are gathered inside the A loads -- forward: the row
//     pointer is per lane and fixed for the whole K loop; weight gradient: the slice's row offsets sit in registers
//     and reach the loads through v_readlane;
//   * forward: one WORKGROUP per 32 x 64 output tile, its four waves contracting a quarter of K each (2816 waves for
//     1024 SIMDs instead of 704), partial tiles summed in a fixed order through LDS;
//   * weight gradient: one WAVE per (64 x 64 tile, reduction slice), ~900 of them, split-K slabs as before.
// fp32 MFMA 32x32x2 (exact fp32), deterministic summation order (no atomics).
#include "gs_common.h"
#include "gs_gather_dev.h"
#include <stdlib.h>

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

#ifdef GS_TIMELINE
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
template <int P>
__device__ __forceinline__ void stream_fwd_tile(const FwdArgs& g, const int tile, const int wave, const int lane,
                                                float (*red)[32][64]) {
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
        if (row < g.M && c < N) *reinterpret_cast<f32x2*>(g.C + (int64_t)row * g.ldc + col_off + c) = v;
    }
    GS_STAMP(3);
}

template <int P>
__global__ __launch_bounds__(256) void sage_stream_fwd_kernel(const FwdArgs g, const CoGatherS J) {
    __shared__ float red[4][32][64];                       // split-K partial tiles of the workgroup's four waves
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: SGPR fields
    if ((int)blockIdx.x < g.n_tiles) {
        stream_fwd_tile<P>(g, stream_xcd_swizzle(blockIdx.x, g.n_tiles), wave, lane, red);
        return;
    }
#ifdef GS_DIAG_RIDER_PRIO
    __builtin_amdgcn_s_setprio(GS_DIAG_RIDER_PRIO);
#endif
#ifdef GS_TIMELINE
    const int tl_item = g.n_tiles * 4 + ((int)blockIdx.x - g.n_tiles) * 4 + wave;     // rider waves follow the host waves
    GS_STAMP(0);
    GS_STAMP(1);
#endif
    run_gather_item(J, ((int64_t)blockIdx.x - g.n_tiles) * 4 + wave, lane);
#ifdef GS_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GS_STAMP(2);
    GS_STAMP(3);
#endif
}

// ------------------------------------------------------------------------------------------ weight gradients
#define GS_MAX_SGROUP 12
struct WgradProb {
    const float* A;       // [*, lda]: row r of the reduction is A[a_idx ? a_idx[r] : r]
    const int32_t* a_idx; // nullable row gather (layer 0: the self rows of the feature table)
    const float* dZ;      // [n, ldz], already offset by col0
    float* slabs;         // [n_slabs][d][ld_slab]
    int32_t lda, ldz, ld_slab;
    int32_t n, d, out_dim;
    int32_t tiles_m, tiles_n, n_slabs, kchunk;   // kchunk: rows per slab (even)
    int32_t item_start;   // first work item of this problem
};
struct WgradArgs {
    WgradProb p[GS_MAX_SGROUP];
    int32_t n, n_items;
};

// One wave: slab[z][f0 .. f0+63][o0 .. o0+63] = sum_{r in slice z} A[r][f] * dZ[r][o]
//
// The wave is alone on its SIMD (~900 items over 1024 SIMDs), so every instruction between two MFMAs that does not fit
// in the shadow of the last MFMA of a k-pair (64 cycles) is an MFMA bubble.  The loop therefore carries 32-bit BYTE
// offsets against wave-uniform base pointers (global_load with an SGPR base), advanced with one add per load; a
// gathered A row costs two v_readlane (SALU) + two VALU.  (The first version -- 64-bit per-lane pointers, a register
// select ladder for the gather index -- spent ~300 cycles per k-pair outside the 256 MFMA cycles.)
// GATHERED: 0 = dense A, 1 = row-gathered A addressed with 32-bit BYTE offsets (tables < 4 GB; larger tables take the tiled
// kernel of gs_gemm.hip -- a form with 16-byte-unit row offsets widened per load was measured slower on the one
// configuration that needs it, RMAT: 95.2 vs 77.3 us/step, benchmarks/variants/README.md).
template <int P, int GATHERED>
__device__ __forceinline__ void stream_wgrad_body(const WgradProb& q, const int z, const int f0, const int o0, const int lane,
                                                  const int tl_item) {
    GS_STAMP(0);
    const int l31 = lane & 31, lh = lane >> 5;
    const int rb = z * q.kchunk, re = min(rb + q.kchunk, q.n);
    const char* __restrict__ Ab = (const char*)q.A;
    const char* __restrict__ Zb = (const char*)q.dZ;
    // per-lane column byte offsets (clamped: out-of-range columns are loaded from a valid column and never stored)
    const uint32_t ca0 = (uint32_t)min(f0 + l31, q.d - 1) * 4u, ca1 = (uint32_t)min(f0 + 32 + l31, q.d - 1) * 4u;
    const uint32_t strideA = (uint32_t)q.lda * 8u, strideZ = (uint32_t)q.ldz * 8u;       // two rows, bytes
    uint32_t zo0 = (uint32_t)(rb + lh) * (uint32_t)q.ldz * 4u + (uint32_t)min(o0 + l31, q.out_dim - 1) * 4u;
    uint32_t zo1 = (uint32_t)(rb + lh) * (uint32_t)q.ldz * 4u + (uint32_t)min(o0 + 32 + l31, q.out_dim - 1) * 4u;
    uint32_t ao0 = (uint32_t)(rb + lh) * (uint32_t)q.lda * 4u + ca0, ao1 = (uint32_t)(rb + lh) * (uint32_t)q.lda * 4u + ca1;
    // Row-gathered A (layer 0 self rows): the byte offsets of the slice's source rows (<= 512) are computed ONCE
    // (lane L of rowoff[j] holds row rb + 64 j + L); the two rows of a k-pair come out with v_readlane.  rowoff[0] is
    // always the current 64-row chunk: the registers rotate down when the load cursor crosses a chunk.
    constexpr int IDXR = 8;
    uint32_t rowoff[IDXR];
    if (GATHERED) {
#pragma unroll
        for (int jx = 0; jx < IDXR; ++jx)
            rowoff[jx] = (uint32_t)q.a_idx[min(rb + lane + 64 * jx, re - 1)] * ((uint32_t)q.lda * 4u);
    }
    uint32_t hi_mask = lh ? 0xFFFFFFFFu : 0u;
    asm volatile("" : "+v"(hi_mask));                      // opaque: keeps `& hi_mask` an AND
    int cur = 0;                                           // load cursor: rows of the slice already requested (even)
    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
    float av0[P], av1[P], bv0[P], bv1[P];
    auto rotate = [&]() {
#pragma unroll
        for (int jx = 0; jx + 1 < IDXR; ++jx) rowoff[jx] = rowoff[jx + 1];
    };
    auto load_stage = [&](const int st) {                  // no arithmetic on the loaded values here: a use would
        uint32_t o0a, o1a;                                 // force a wait right behind the load and empty the ring
        const char* __restrict__ Ar = Ab;
        if (GATHERED) {
            const int l = cur & 63;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)rowoff[0], l);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)rowoff[0], l + 1);   // l even: same register
            const uint32_t t = (hi - lo) & hi_mask;        // (not a select: two SGPR sources would cost two v_mov first)
            o0a = lo + t + ca0;
            o1a = lo + t + ca1;
        } else {
            o0a = ao0; o1a = ao1;
            ao0 += strideA; ao1 += strideA;
        }
#ifdef GS_TIMELINE_NOLOAD
        av0[st] = __int_as_float(o0a); av1[st] = __int_as_float(o1a); bv0[st] = __int_as_float(zo0); bv1[st] = __int_as_float(zo1);
#else
        av0[st] = *(const float*)(Ar + o0a);
        av1[st] = *(const float*)(Ar + o1a);
        bv0[st] = *(const float*)(Zb + zo0);
        bv1[st] = *(const float*)(Zb + zo1);
#endif
        zo0 += strideZ; zo1 += strideZ;
        cur += 2;
    };
    auto compute_stage = [&](const int st) {
        acc00 = mfma32(av0[st], bv0[st], acc00);
        acc01 = mfma32(av0[st], bv1[st], acc01);
        acc10 = mfma32(av1[st], bv0[st], acc10);
        acc11 = mfma32(av1[st], bv1[st], acc11);
    };
    // Steady state: consume stage st and refill it, in a pinned issue order.  The wave issues in order and an MFMA holds
    // the matrix pipe for 64 cycles, so whatever sits between two MFMAs is free as long as it issues in < 64 cycles --
    // and an MFMA bubble if a stage's address arithmetic and its four loads are lumped behind its last MFMA (measured
    // with the loads removed altogether: 299 / 350 cycles per k-pair dense / gathered instead of 256).  Each operand
    // register is refilled right behind the last MFMA that reads it.
    auto fused_stage = [&](const int st) {
        acc00 = mfma32(av0[st], bv0[st], acc00);
        uint32_t o0a, o1a;
        const char* __restrict__ Ar = Ab;
        if (GATHERED) {
            const int l = cur & 63;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)rowoff[0], l);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)rowoff[0], l + 1);
            const uint32_t t = (hi - lo) & hi_mask;
            o0a = lo + t + ca0;
            o1a = lo + t + ca1;
        } else {
            o0a = ao0; o1a = ao1;
            ao0 += strideA; ao1 += strideA;
        }
        __builtin_amdgcn_sched_barrier(0);
        acc01 = mfma32(av0[st], bv1[st], acc01);
        __builtin_amdgcn_sched_barrier(0);
        av0[st] = *(const float*)(Ar + o0a);
        __builtin_amdgcn_sched_barrier(0);
        acc10 = mfma32(av1[st], bv0[st], acc10);
        __builtin_amdgcn_sched_barrier(0);
        bv0[st] = *(const float*)(Zb + zo0);
        zo0 += strideZ;
        __builtin_amdgcn_sched_barrier(0);
        acc11 = mfma32(av1[st], bv1[st], acc11);
        __builtin_amdgcn_sched_barrier(0);
        av1[st] = *(const float*)(Ar + o1a);
        bv1[st] = *(const float*)(Zb + zo1);
        zo1 += strideZ;
        cur += 2;
        __builtin_amdgcn_sched_barrier(0);
    };
    const int nfull = (re - rb) >> 1;                      // k-pairs whose two rows both exist
    // branch-free steady state, see stream_fwd_item.  2 P divides 64, so the cursor crosses a 64-row chunk only between
    // two iterations.
    static_assert(64 % (2 * P) == 0, "ring depth");
    int kp = 0;
    if (nfull >= P) {
#pragma unroll
        for (int st = 0; st < P; ++st) load_stage(st);
        GS_STAMP(1);
#pragma unroll 1
        for (; kp + 2 * P <= nfull; kp += P) {
            if (GATHERED && (cur & 63) == 0) rotate();
#pragma unroll
            for (int st = 0; st < P; ++st) fused_stage(st);
        }
#pragma unroll
        for (int st = 0; st < P; ++st) compute_stage(st);
        kp += P;
    }
#pragma unroll 1
    for (; kp < nfull; ++kp) {
        if (GATHERED && (cur & 63) == 0 && cur) rotate();
        load_stage(0);
        compute_stage(0);
    }
    if ((re - rb) & 1) {                                   // odd slice: its last row pairs with a zero row
        const int r = re - 1;
        const float mk = lh == 0 ? 1.f : 0.f;
        const int64_t ra = GATHERED ? q.a_idx[r] : r;
        av0[0] = *(const float*)(Ab + ra * q.lda * 4 + ca0) * mk;
        av1[0] = *(const float*)(Ab + ra * q.lda * 4 + ca1) * mk;
        bv0[0] = q.dZ[(int64_t)r * q.ldz + min(o0 + l31, q.out_dim - 1)];
        bv1[0] = q.dZ[(int64_t)r * q.ldz + min(o0 + 32 + l31, q.out_dim - 1)];
        compute_stage(0);
    }
    GS_STAMP(2);
    float* S = q.slabs + (int64_t)z * q.d * q.ld_slab;
    const int c0 = o0 + l31, c1 = o0 + 32 + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int rr = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int fa = f0 + rr, fb = f0 + 32 + rr;
        if (fa < q.d) {
            if (c0 < q.out_dim) S[fa * q.ld_slab + c0] = acc00[e];
            if (c1 < q.out_dim) S[fa * q.ld_slab + c1] = acc01[e];
        }
        if (fb < q.d) {
            if (c0 < q.out_dim) S[fb * q.ld_slab + c0] = acc10[e];
            if (c1 < q.out_dim) S[fb * q.ld_slab + c1] = acc11[e];
        }
    }
    GS_STAMP(3);
}

template <int P>
__device__ __forceinline__ void stream_wgrad_item(const WgradArgs& G, const int item, const int lane) {
    int pi = 0;
    while (pi + 1 < G.n && item >= G.p[pi + 1].item_start) ++pi;
    const WgradProb& q = G.p[pi];
    const int local = item - q.item_start;
    const int tiles = q.tiles_m * q.tiles_n;
    const int z = local / tiles;
    const int tt = local - z * tiles;
    const int tile_m = tt / q.tiles_n, tile_n = tt - tile_m * q.tiles_n;
    if (!q.a_idx) stream_wgrad_body<P, 0>(q, z, tile_m * 64, tile_n * 64, lane, item);
    else stream_wgrad_body<P, 1>(q, z, tile_m * 64, tile_n * 64, lane, item);
}

template <int P>
__global__ __launch_bounds__(256) void stream_wgrad_kernel(const WgradArgs G, const int mfma_blocks, const CoGatherS J) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: per-item fields live in SGPRs
    if ((int)blockIdx.x < mfma_blocks) {
        const int item = stream_xcd_swizzle(blockIdx.x, mfma_blocks) * 4 + wave;
        if (item < G.n_items) stream_wgrad_item<P>(G, item, lane);
        return;
    }
    run_gather_item(J, ((int64_t)blockIdx.x - mfma_blocks) * 4 + wave, lane);
}

// ------------------------------------------------------------------------------------------ host side
static inline int rup4s(int x) { return (x + 3) & ~3; }

static int sage_dense_fwd_stream_impl(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self, const float* agg,
                                      int64_t ld_agg, int32_t d, int64_t n, const float* W_self, int64_t ldw_self,
                                      const float* W_neigh, int64_t ldw_neigh, int32_t out_dim, int act, const float* bias,
                                      float* out, int64_t ldo, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);

extern "C" int gs_sage_dense_fwd_stream(const float* self, int64_t ld_self, const int32_t* self_idx, const float* agg,
                                        int64_t ld_agg, int32_t d, int64_t n, const float* W_self, int64_t ldw_self,
                                        const float* W_neigh, int64_t ldw_neigh, int32_t out_dim, int act, const float* bias,
                                        float* out, int64_t ldo, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    return sage_dense_fwd_stream_impl(self, ld_self, self_idx, d, agg, ld_agg, d, n, W_self, ldw_self, W_neigh, ldw_neigh, out_dim, act,
                                      bias, out, ldo, jobs_host, n_jobs, stream);
}

// the same launch with different reduction lengths of the two terms (pooling aggregators: self rows of d_self features, pooled rows of
// d_agg = hidden_dim; aggregators.py:183-187 / :261-265)
extern "C" int gs_sage_dense_fwd_stream2(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self, const float* agg,
                                         int64_t ld_agg, int32_t d_agg, int64_t n, const float* W_self, int64_t ldw_self,
                                         const float* W_neigh, int64_t ldw_neigh, int32_t out_dim, int act, const float* bias,
                                         float* out, int64_t ldo, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    GS_REQUIRE(self && d_self > 0, "gs_sage_dense_fwd_stream2: needs a self term");
    return sage_dense_fwd_stream_impl(self, ld_self, self_idx, d_self, agg, ld_agg, d_agg, n, W_self, ldw_self, W_neigh, ldw_neigh,
                                      out_dim, act, bias, out, ldo, jobs_host, n_jobs, stream);
}

static int sage_dense_fwd_stream_impl(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self, const float* agg,
                                      int64_t ld_agg, int32_t d, int64_t n, const float* W_self, int64_t ldw_self,
                                      const float* W_neigh, int64_t ldw_neigh, int32_t out_dim, int act, const float* bias,
                                      float* out, int64_t ldo, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    GS_REQUIRE(n > 0 && agg && W_neigh && out && d > 0 && out_dim > 0, "gs_sage_dense_fwd_stream: bad args");
    GS_REQUIRE(out_dim % 2 == 0 && ldo % 2 == 0, "gs_sage_dense_fwd_stream: out_dim and ldo must be even (8-byte column pairs)");
    GS_CHECK_MAT(agg, ld_agg, "gs_sage_dense_fwd_stream agg");
    GS_CHECK_MAT(W_neigh, ldw_neigh, "gs_sage_dense_fwd_stream W_neigh");
    GS_CHECK_MAT(out, ldo, "gs_sage_dense_fwd_stream out");
    GS_REQUIRE(ld_agg >= rup4s(d) && ldw_neigh >= out_dim, "gs_sage_dense_fwd_stream: ld too small");
    FwdArgs g = {};
    if (self) {
        GS_CHECK_MAT(self, ld_self, "gs_sage_dense_fwd_stream self");
        GS_CHECK_MAT(W_self, ldw_self, "gs_sage_dense_fwd_stream W_self");
        GS_REQUIRE(ld_self >= rup4s(d_self) && ldw_self >= out_dim, "gs_sage_dense_fwd_stream: self ld too small");
        g.t[0] = FwdTerm{self, self_idx, W_self, (int32_t)ld_self, (int32_t)ldw_self, d_self};
        g.t[1] = FwdTerm{agg, nullptr, W_neigh, (int32_t)ld_agg, (int32_t)ldw_neigh, d};
        g.nterms = 2;
    } else {
        g.t[0] = FwdTerm{agg, nullptr, W_neigh, (int32_t)ld_agg, (int32_t)ldw_neigh, d};
        g.nterms = 1;
    }
    GS_REQUIRE(ldo >= out_dim * g.nterms, "gs_sage_dense_fwd_stream: ldo too small");
    GS_REQUIRE(std::max(ld_self, ld_agg) < (1ll << 31) && ((int64_t)std::max(d, self ? d_self : 0) + 8) * std::max(ldw_self, ldw_neigh) * 4 < (1ll << 32),
               "gs_sage_dense_fwd_stream: 32-bit offsets exceeded");
    g.M = (int32_t)n; g.N = out_dim; g.K = d; g.C = out; g.ldc = (int32_t)ldo; g.bias = bias; g.act = act;
    GS_REQUIRE(n < (1ll << 31) - 64, "gs_sage_dense_fwd_stream: too many rows");
    const int tiles_m = (int)gs_ceil_div(n, 32);
    g.tiles_n = (int)gs_ceil_div(out_dim, 64);
    g.n_tiles = tiles_m * g.tiles_n * g.nterms;
    CoGatherS J = {};
    int64_t waves = 0;
    int rc = build_cojobs_s(jobs_host, n_jobs, &J, &waves);
    if (rc != GS_OK) return rc;
    const int64_t blocks = g.n_tiles + gs_ceil_div(waves, 4);
    GS_REQUIRE(blocks < (1ll << 31), "gs_sage_dense_fwd_stream: grid too large");
    hipLaunchKernelGGL(sage_stream_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, J);   // ring of 4 (6, 8: measured slower)
    GS_LAUNCH_CHECK("sage_stream_fwd_kernel");
    return GS_OK;
}

extern "C" int gs_dense_wgrad_grouped_stream(const gs_wgrad_desc* descs_host, int32_t n_desc, const gs_gather_desc* jobs_host,
                                             int32_t n_jobs, void* stream) {
    GS_REQUIRE(descs_host && n_desc > 0 && n_desc <= GS_MAX_SGROUP, "gs_dense_wgrad_grouped_stream: 1..%d problems", GS_MAX_SGROUP);
    WgradArgs G = {};
    G.n = n_desc;
    int items = 0;
    for (int i = 0; i < n_desc; ++i) {
        const gs_wgrad_desc& q = descs_host[i];
        GS_CHECK_MAT(q.A, q.lda, "gs_dense_wgrad_grouped_stream A");
        GS_CHECK_MAT(q.dZ, q.ldz, "gs_dense_wgrad_grouped_stream dZ");
        GS_CHECK_MAT(q.slabs, q.ld_slab, "gs_dense_wgrad_grouped_stream slabs");
        GS_REQUIRE(q.d > 0 && q.out_dim > 0 && q.n > 0 && q.n_slabs > 0 && q.col0 >= 0 && q.col0 % 4 == 0,
                   "gs_dense_wgrad_grouped_stream: bad sizes");
        GS_REQUIRE(q.lda >= q.d && q.ldz >= q.col0 + q.out_dim && q.ld_slab >= q.out_dim, "gs_dense_wgrad_grouped_stream: ld too small");
        // the kernel addresses A and dZ with 32-bit BYTE offsets against their base pointers
        const int64_t a_rows = q.a_idx ? q.a_rows : q.n;
        GS_REQUIRE(!q.a_idx || q.a_rows > 0, "gs_dense_wgrad_grouped_stream: a row-gathered problem must state a_rows");
        GS_REQUIRE((a_rows + 1) * q.lda * 4 < (1ll << 32) && (q.n + 1) * q.ldz * 4 < (1ll << 32) &&
                   (int64_t)q.d * q.ld_slab < (1ll << 31),
                   "gs_dense_wgrad_grouped_stream: 32-bit offsets exceeded (A %lld x %lld, dZ %lld x %lld)",
                   (long long)a_rows, (long long)q.lda, (long long)q.n, (long long)q.ldz);
        WgradProb& p = G.p[i];
        p.A = q.A; p.a_idx = q.a_idx; p.dZ = q.dZ + q.col0; p.slabs = q.slabs;
        p.lda = (int32_t)q.lda; p.ldz = (int32_t)q.ldz; p.ld_slab = (int32_t)q.ld_slab;
        p.n = (int32_t)q.n; p.d = q.d; p.out_dim = q.out_dim;
        p.tiles_m = (int)gs_ceil_div(q.d, 64);
        p.tiles_n = (int)gs_ceil_div(q.out_dim, 64);
        p.n_slabs = q.n_slabs;
        p.kchunk = (int32_t)(gs_ceil_div(gs_ceil_div(q.n, q.n_slabs), 2) * 2);
        GS_REQUIRE(!q.a_idx || p.kchunk <= 512, "gs_dense_wgrad_grouped_stream: a row-gathered problem needs slices of <= 512 rows "
                   "(n = %lld, n_slabs = %d)", (long long)q.n, q.n_slabs);
        p.item_start = items;
        items += p.tiles_m * p.tiles_n * q.n_slabs;
    }
    G.n_items = items;
    const int mfma_blocks = (int)gs_ceil_div(items, 4);
    CoGatherS J = {};
    int64_t waves = 0;
    int rc = build_cojobs_s(jobs_host, n_jobs, &J, &waves);
    if (rc != GS_OK) return rc;
    const int64_t blocks = mfma_blocks + gs_ceil_div(waves, 4);
    GS_REQUIRE(blocks < (1ll << 31), "gs_dense_wgrad_grouped_stream: grid too large");
    static const int ring = getenv("GS_STREAM_WGRAD_P") ? atoi(getenv("GS_STREAM_WGRAD_P")) : 8;   // tuning hook
    if (ring == 4)
        hipLaunchKernelGGL(stream_wgrad_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, G, mfma_blocks, J);
    else if (ring >= 32)
        hipLaunchKernelGGL(stream_wgrad_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, G, mfma_blocks, J);
    else if (ring >= 16)
        hipLaunchKernelGGL(stream_wgrad_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, G, mfma_blocks, J);
    else
        hipLaunchKernelGGL(stream_wgrad_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, G, mfma_blocks, J);
    GS_LAUNCH_CHECK("stream_wgrad_kernel");
    return GS_OK;
}
