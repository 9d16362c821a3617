// Shared device code of the fused "tail" launches (gs_tail.hip: supervised layer 1 + head; gs_unsup_tail.hip: unsupervised
// layer 1 + link-prediction head): the argument block, the z helper workgroups and their hand-over, small wave helpers.
#pragma once
#include "gs_common.h"
#include "gs_gather_dev.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define TAIL_ROWS 16
#define TAIL_THREADS 512
#define TAIL_WAVES 8

struct TailArgs {
    const float* h0; int64_t ldh; int64_t n; int32_t s; int32_t D;
    const float* Ws; int64_t ldws; const float* Wn; int64_t ldwn; int32_t O;
    const float* Wh; int64_t ldwh; const float* bh; const float* labels; int64_t ldlab; int32_t C; int32_t sigmoid;
    float* means; int64_t ldm;
    float* z; int64_t ldz;
    float* y; int64_t ldy;
    float* logits; int64_t ldlo; float* preds; int64_t ldp; float* dlogits; int64_t lddl; float* loss_rows;
    float* dz; int64_t lddz;
    float* d_h0; int64_t lddh;
    uint64_t* c0; uint64_t d0; uint64_t* c1; uint64_t d1; uint64_t* c2; uint64_t d2;
    int32_t train;
    // hand-over state of the z helpers, G = ceil(n / 16) groups.
    //   gs_linkpred_tail (counter form): [0, G) monotonic arrival counters (helpers add 1), [G, 2G) arrivals already consumed by
    //   earlier launches (written only by the group's main workgroup), [2G] error flags (bit 0: a main workgroup gave up
    //   waiting, bit 1: a group saw a number of arrivals other than HP).
    //   gs_sage_tail_fwd_bwd (granule form): [G, 2G) the group's launch EPOCH (written only by the group's main workgroup, at its
    //   end), [2G] error flags (bit 0), and from word 2G + 2 on G x 16 x 2 O eight-byte granules {z element, tag}: a helper
    //   writes every element of its slab as ONE 8-byte device-scope store tagged epoch + 1, the main workgroup polls the
    //   granules themselves -- no flag, no drained store queue, no second round trip behind a flag.
    // Nothing is ever reset, so a launch does not depend on a reset store of the previous one.
    uint32_t* sync;
    // split form: z (and the neighbor means) were written by a PREVIOUS launch (sage_tail_z_kernel): this launch has no
    // helper workgroups, waits for nothing and touches no hand-over state
    int32_t z_ready;
    // GCNAggregator form of layer 1 (aggregators.py:101-116): ONE weight matrix W [D, 2 O] given as Ws = W, Wn = W + O; both
    // column halves of z contract the SAME operand, the mean over {neighbors} U {self} = (sum_j h_neigh_j + h_self) / (s + 1),
    // which is also what `means` receives; every row of d_h0 (self and neighbor rows alike) gets relu' * (dz . W^T) / (s + 1).
    int32_t gcn;
    // Row map of the unsupervised tail (gs_unsup_tail.hip), 0 = off: the rows are [batch1 (pairB) | batch2 (pairB) | negatives];
    // groups [0, pair_groups) hold 8 PAIRS each -- local rows 0..7 = batch1 rows 8 g .. 8 g + 7, local rows 8..15 = their
    // batch2 partners pairB + 8 g .. -- so that a pair meets in ONE workgroup; groups behind them hold 16 negatives each.
    int32_t pairB, pair_groups;
    // optional: ids_copy_dst[0 : ids_copy_n) = ids_copy_src[...], written by the launch's z-helper workgroups (gs_tail_desc)
    const int32_t* ids_copy_src; int32_t* ids_copy_dst; int64_t ids_copy_n;
};

// Source row of local row r of group g (and whether it exists); rows that do not exist map to a valid row, never stored.
__device__ __forceinline__ int tail_row(const TailArgs& a, const int g, const int r, bool& valid) {
    const int n = (int)a.n;
    if (a.pairB == 0) {
        const int i = g * TAIL_ROWS + r;
        valid = i < n;
        return min(i, n - 1);
    }
    if (g < a.pair_groups) {
        const int p = 8 * g + (r & 7);
        valid = p < a.pairB;
        const int pc = min(p, a.pairB - 1);
        return r < 8 ? pc : a.pairB + pc;
    }
    const int i = 2 * a.pairB + TAIL_ROWS * (g - a.pair_groups) + r;
    valid = i < n;
    return min(i, n - 1);
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// WHAT BOUNDS THIS KERNEL: 32 workgroups, one per CU, 8 waves each -- nothing hides latency, and every global load a
// phase waits for costs a full memory round trip (~1.5-2 us: the operands were last written by another XCD).  A first
// version that loaded "just in time" spent ~24 dependent round trips (38-59 us for ~3 us of MFMA work).  None of the
// global operands depends on anything computed here (h0, the weights, the labels are all inputs), so they are
// PREFETCHED INTO REGISTERS up front, in two waves of requests, and the phases consume them from registers:
//   S0 (kernel entry): this thread's h0 rows (self + s neighbors, 2 passes) and the wave's whole W slab of the z
//                      contraction                                                        -> ~216 VGPRs in flight
//   S1 (after z):      the wave's slices of W_head (both forms), labels / bias, and its two W slabs of the input-gradient
//                      contraction                                                        -> ~200 VGPRs in flight
// The relu mask of phase 8 is kept from S0 as bit flags, so h0 is read exactly once.  Barriers between phases are
// LDS-only (fence on the "local" address space + s_barrier): they do not drain the outstanding global loads.

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Wave-wide sum / max, the result in every lane, on the DPP path: an xor butterfly over 1, 2, 4, 8 inside each row of 16 lanes
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror and row_mirror read the SAME partner value as lane ^ 1, ^ 2, ^ 4,
// ^ 8: after two steps a quad is uniform, after three an 8-lane group is), then the four row results, read with v_readlane, as
// (r0 + r1) + (r2 + r3).  A fixed order, the same in every lane and every launch (deterministic) -- but not the order of the
// __shfl_xor loop it replaces (that one started at lane ^ 32): sums differ from it in the last bit.  The __shfl_xor form
// compiles to six dependent ds_bpermute_b32 (~130 cycles each), and a main workgroup of the fused tail runs twelve reductions
// in its dependent chain: 1.9 -> 1.3 us for the loss phase, 0.9 -> 0.6 for l2-normalise' (benchmarks/timeline_tail.py).
template <int CTRL>
__device__ __forceinline__ float tail_dpp(const float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
#ifdef TAIL_REDUCE_SHFL           // diagnostics: the ds_bpermute butterfly in the SAME order (1, 2, 4, 8, then the rows)
__device__ __forceinline__ float tail_wave_sum(float v) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
    const float r0 = __shfl(v, 0, 64), r1 = __shfl(v, 16, 64), r2 = __shfl(v, 32, 64), r3 = __shfl(v, 48, 64);
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float tail_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
#else
__device__ __forceinline__ float tail_wave_sum(float v) {
    v += tail_dpp<0xB1>(v);
    v += tail_dpp<0x4E>(v);
    v += tail_dpp<0x141>(v);
    v += tail_dpp<0x140>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float tail_wave_max(float v) {
    v = fmaxf(v, tail_dpp<0xB1>(v));
    v = fmaxf(v, tail_dpp<0x4E>(v));
    v = fmaxf(v, tail_dpp<0x141>(v));
    v = fmaxf(v, tail_dpp<0x140>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
#endif

#define TAIL_NB 11   // neighbor rows per batch row held in registers (s <= TAIL_NB)
#ifndef TAIL_HELPER_STAMP
#define TAIL_HELPER_STAMP(k) do { } while (0)
#endif

// z helper (see the role comment in sage_tail_kernel): z[16 rows of group g][64 columns part*64 ..] of
//   z = [h_self . W_self | mean_j(h_neigh_j) . W_neigh]      (aggregators.py:48-58, concat, identity act)
// 8 waves = 8 K-slices of the slab's term (a 64-column slab lies in ONE term: 64 divides O), partial 16 x 64 tiles
// summed in wave order through LDS, then published: stores -> device-scope release fence -> arrival counter.
// The helper whose slab starts the neighbor term also writes the neighbor means (an input of the weight gradients).
template <int D, int O>
__device__ __forceinline__ void tail_z_helper(const TailArgs& a, const int g, const int part, const int G, const bool publish = true) {
    constexpr int ldh = D + 4;
    constexpr int D4 = D / 4;
    constexpr int PASSES = TAIL_ROWS * D4 / TAIL_THREADS;
    constexpr int KW = D / 4 / TAIL_WAVES;                       // k-steps (4 k each) per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                                             // [16][ldh]  the term's A rows
    float* Pz = lds + TAIL_ROWS * ldh;                           // [8 waves][16][64] partial tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TAIL_HELPER_STAMP(0);
#ifndef TAIL_NO_IDS_COPY          // (diagnostics switch)
    if (a.ids_copy_n > 0) {                                      // (see gs_tail_desc.ids_copy_src: a few KB, spread over the helpers)
        constexpr int HPc = 2 * O / 64;
        for (int64_t t = ((int64_t)g * HPc + part) * TAIL_THREADS + tid; t < a.ids_copy_n; t += (int64_t)G * HPc * TAIL_THREADS)
            a.ids_copy_dst[t] = a.ids_copy_src[t];
    }
#endif
    const int j = lane & 15, q = lane >> 4;
    const int n = (int)a.n, s = a.s, ldh0 = (int)a.ldh;
    const int col_base = part * 64;
    const int term = col_base >= O ? 1 : 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // this wave's weight slice: rows 4 (wave KW + u) + q, columns col_base + 32 t + 2 j (+1)
    const int ldw = (int)(term ? a.ldwn : a.ldws);
    const float* Wp = (term ? a.Wn : a.Ws) + (4 * wave * KW + q) * ldw + (col_base - term * O) + 2 * j;
    f32x2 bz[2][KW];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < KW; ++u) bz[t][u] = *reinterpret_cast<const f32x2*>(Wp + (4 * u) * ldw + 32 * t);
    const float inv_s = 1.0f / (float)s, inv_s1 = 1.0f / (float)(s + 1);
    // the tag of this launch's granules: the group's epoch + 1 (the epoch word is written by the group's main workgroup at the
    // END of a launch, i.e. after it has consumed every granule of this helper: a kernel boundary lies between that store and
    // this load)
    uint32_t tag = 0u;
    if (publish) tag = a.sync[G + g] + 1u;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int it = tid + p * TAIL_THREADS;
        const int r = it / D4, c = (it % D4) * 4;
        bool valid;
        const int i = tail_row(a, g, r, valid);
        f32x4 v;
        if (!term && !a.gcn) {
            v = *reinterpret_cast<const f32x4*>(a.h0 + i * ldh0 + c);
        } else {
            const float* nb = a.h0 + (n + i * s) * ldh0 + c;
            f32x4 hv[TAIL_NB];
            f32x4 hs = zero4;
            if (a.gcn) hs = *reinterpret_cast<const f32x4*>(a.h0 + i * ldh0 + c);
#pragma unroll
            for (int u = 0; u < TAIL_NB; ++u) hv[u] = *reinterpret_cast<const f32x4*>(nb + min(u, s - 1) * ldh0);
            v = zero4;
#pragma unroll
            for (int u = 0; u < TAIL_NB; ++u)
                if (u < s) v += hv[u];                               // summation order j = 0..s-1, as gather_mean_wave
            if (a.gcn) { v += hs; v *= inv_s1; }                      // ... then the self row, as gather_mean_wave's GCN form
            else v *= inv_s;
            if (valid && col_base == O) *reinterpret_cast<f32x4*>(a.means + i * (int)a.ldm + c) = v;
        }
        *reinterpret_cast<f32x4*>(As + r * ldh + c) = valid ? v : zero4;
    }
    lds_barrier();
    TAIL_HELPER_STAMP(1);
    {
        const float* A = As + j * ldh + 4 * wave * KW + q;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
            for (int u = 0; u < KW; ++u) {
                const float av = A[4 * u];
                acc0 = mfma16(av, bz[t][u].x, acc0);
                acc1 = mfma16(av, bz[t][u].y, acc1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x2*>(Pz + (wave * TAIL_ROWS + 4 * q + i) * 64 + 32 * t + 2 * j) = f32x2{acc0[i], acc1[i]};
        }
    }
    lds_barrier();
    TAIL_HELPER_STAMP(2);
#pragma unroll
    for (int p = 0; p < TAIL_ROWS * 32 / TAIL_THREADS; ++p) {     // (row, column pair) items: 16 x 32
        const int it = tid + p * TAIL_THREADS;
        const int r = it >> 5, c2 = (it & 31) * 2;
        f32x2 v = *reinterpret_cast<const f32x2*>(Pz + r * 64 + c2);
#pragma unroll
        for (int w = 1; w < TAIL_WAVES; ++w) v += *reinterpret_cast<const f32x2*>(Pz + (w * TAIL_ROWS + r) * 64 + c2);
        if (publish) {
            // Published as GRANULES: {element, tag} in ONE 8-byte device-scope (write-through) store each -- the data is its own
            // flag (an 8-byte store is single-copy atomic), every row of the group (the main workgroup polls all 16 x 2 O
            // granules; rows that do not exist carry zeros).  (A release FENCE would write back the XCD's whole L2, dirty gather
            // output of the riders included -- z arrived 17 us late; z stores + drained queue + arrival counter + the main
            // workgroup's poll + its z loads were 5 dependent round trips, 5-6 us behind the helpers' last MFMA.)
            unsigned long long* zg = reinterpret_cast<unsigned long long*>(a.sync + 2 * G + 2) +
                                     ((int64_t)g * TAIL_ROWS + r) * (2 * O) + col_base + c2;
            union { float f; uint32_t u; } c0v, c1v;
            c0v.f = v.x; c1v.f = v.y;
            __hip_atomic_store(zg, ((unsigned long long)tag << 32) | c0v.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(zg + 1, ((unsigned long long)tag << 32) | c1v.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            bool zvalid;                                             // split form: the kernel boundary is the hand-over
            const int zi = tail_row(a, g, r, zvalid);
            if (zvalid) *reinterpret_cast<f32x2*>(a.z + zi * (int)a.ldz + col_base + c2) = v;
        }
    }
    TAIL_HELPER_STAMP(3);
}

// Main workgroup: this launch's z rows of group g from the helpers' granules into Zs [16][ldzs] (every thread polls its own
// 16 x Z / TAIL_THREADS granules with device-scope loads until all carry the launch's tag).  BOUNDED: if a helper of this launch
// never arrives (a tool that serialises workgroups, a dispatch order that starves it) the error flag is set and the workgroup
// goes on with whatever the granules hold -- the step's numbers are then garbage, but the stream does not hang, and the host
// raises on the flag at its next fetch.  Returns the tag (the group's new epoch, stored by tail_sync_done).
template <int Z, int CH>
__device__ __forceinline__ uint32_t tail_pick_up_z(const TailArgs& a, const int G, const int grp, float* Zs, const int ldzs) {
    constexpr int NG = TAIL_ROWS * Z / TAIL_THREADS;             // granules per thread (8 for Z = 256), polled CH at a time
    static_assert(NG % CH == 0, "chunk");
    const uint32_t tag = a.sync[G + grp] + 1u;                   // (written by this workgroup's predecessor: a kernel boundary ago)
    const unsigned long long* zg = reinterpret_cast<const unsigned long long*>(a.sync + 2 * G + 2) + (int64_t)grp * TAIL_ROWS * Z;
    bool ok = true;
#pragma unroll
    for (int p0 = 0; p0 < NG; p0 += CH) {
        unsigned long long gr[CH];
        uint32_t spins = 0u;
        bool all;
        do {
#pragma unroll
            for (int p = 0; p < CH; ++p)
                gr[p] = __hip_atomic_load(zg + threadIdx.x + (p0 + p) * TAIL_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            all = true;
#pragma unroll
            for (int p = 0; p < CH; ++p) all = all && (uint32_t)(gr[p] >> 32) == tag;
            if (all) break;
            __builtin_amdgcn_s_sleep(8);
        } while (++spins < (1u << 18));
        ok = ok && all;
#pragma unroll
        for (int p = 0; p < CH; ++p) {
            const int it = threadIdx.x + (p0 + p) * TAIL_THREADS;
            union { uint32_t u; float f; } cv;
            cv.u = (uint32_t)gr[p];
            Zs[(it / Z) * ldzs + it % Z] = cv.f;
        }
    }
    if (!ok) __hip_atomic_fetch_or(a.sync + 2 * G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return tag;
}

// End of a main workgroup of the granule form: the group's epoch advances (the next launch's tag differs).  With NH main
// workgroups per group the LAST of them to finish stores it (word grp of the buffer counts finished main workgroups, NH per
// launch, never reset): each of them has read the epoch by then, whatever order they were dispatched in.
template <int NH>
__device__ __forceinline__ void tail_epoch_done(const TailArgs& a, const int G, const int grp, const uint32_t tag) {
    if (threadIdx.x == 0) {
        if (NH == 1) {
            a.sync[G + grp] = tag;
        } else {
            const uint32_t old = __hip_atomic_fetch_add(a.sync + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((old + 1u) % (uint32_t)NH == 0u) a.sync[G + grp] = tag;
        }
    }
}

// z helper of a whole TERM (gs_unsup_tail.hip): all O columns of z's self half (term 0) or neighbor-mean half (term 1) of
// group g -- the term's A rows are fetched ONCE (the 64-column helper above fetches them once per slab: the neighbor term's
// 11 rows per output row twice for O = 128) and a group needs 2 helper workgroups instead of 2 O / 64, so that the helpers of
// 66 groups AND the main workgroups are resident together on 256 CUs.  Every output element is formed exactly as by
// tail_z_helper (same K slices, same order): bit-identical z.
template <int D, int O>
__device__ __forceinline__ void tail_z_term_helper(const TailArgs& a, const int g, const int term) {
    constexpr int NS = O / 64;                                   // 64-column slabs of the term
    constexpr int ldh = D + 4;
    constexpr int D4 = D / 4;
    constexpr int PASSES = TAIL_ROWS * D4 / TAIL_THREADS;
    constexpr int KW = D / 4 / TAIL_WAVES;                       // k-steps (4 k each) per wave
    constexpr int PW = 64 * NS;                                  // row length of the partial tiles
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                                             // [16][ldh]  the term's A rows
    float* Pz = lds + TAIL_ROWS * ldh;                           // [8 waves][16][PW] partial tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int n = (int)a.n, s = a.s, ldh0 = (int)a.ldh;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int ldw = (int)(term ? a.ldwn : a.ldws);
    const float* Wp = (term ? a.Wn : a.Ws) + (4 * wave * KW + q) * ldw + 2 * j;
    f32x2 bz[NS][2][KW];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < KW; ++u) bz[ns][t][u] = *reinterpret_cast<const f32x2*>(Wp + (4 * u) * ldw + 64 * ns + 32 * t);
    const float inv_s = 1.0f / (float)s;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int it = tid + p * TAIL_THREADS;
        const int r = it / D4, c = (it % D4) * 4;
        bool valid;
        const int i = tail_row(a, g, r, valid);
        f32x4 v;
        if (!term) {
            v = *reinterpret_cast<const f32x4*>(a.h0 + i * ldh0 + c);
        } else {
            const float* nb = a.h0 + (n + i * s) * ldh0 + c;
            f32x4 hv[TAIL_NB];
#pragma unroll
            for (int u = 0; u < TAIL_NB; ++u) hv[u] = *reinterpret_cast<const f32x4*>(nb + min(u, s - 1) * ldh0);
            v = zero4;
#pragma unroll
            for (int u = 0; u < TAIL_NB; ++u)
                if (u < s) v += hv[u];                               // summation order j = 0..s-1, as gather_mean_wave
            v *= inv_s;
            if (valid) *reinterpret_cast<f32x4*>(a.means + i * (int)a.ldm + c) = v;
        }
        *reinterpret_cast<f32x4*>(As + r * ldh + c) = valid ? v : zero4;
    }
    lds_barrier();
    {
        const float* A = As + j * ldh + 4 * wave * KW + q;
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
                for (int u = 0; u < KW; ++u) {
                    const float av = A[4 * u];
                    acc0 = mfma16(av, bz[ns][t][u].x, acc0);
                    acc1 = mfma16(av, bz[ns][t][u].y, acc1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x2*>(Pz + (wave * TAIL_ROWS + 4 * q + i) * PW + 64 * ns + 32 * t + 2 * j) = f32x2{acc0[i], acc1[i]};
            }
    }
    lds_barrier();
#pragma unroll
    for (int p = 0; p < TAIL_ROWS * (PW / 2) / TAIL_THREADS; ++p) {   // (row, column pair) items: 16 x PW / 2
        const int it = tid + p * TAIL_THREADS;
        const int r = it / (PW / 2), c2 = (it % (PW / 2)) * 2;
        f32x2 v = *reinterpret_cast<const f32x2*>(Pz + r * PW + c2);
#pragma unroll
        for (int w = 1; w < TAIL_WAVES; ++w) v += *reinterpret_cast<const f32x2*>(Pz + (w * TAIL_ROWS + r) * PW + c2);
        bool zvalid;
        const int zi = tail_row(a, g, r, zvalid);
        if (zvalid) {                                                // device-scope (write-through) stores, see tail_z_helper
            union { f32x2 f; unsigned long long u; } cv;
            cv.f = v;
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.z + zi * (int)a.ldz + term * O + c2), cv.u,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.sync + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// End of a main workgroup: exactly HP helpers of THIS launch must have arrived since `base` (the helpers never wait, and
// this runs ~15 us after the hand-over); anything else -- leftovers of a launch that did not finish, a helper that
// never ran -- is flagged.  The consumed count is then published for the next launch (no reset of the counter).
template <int HP>
__device__ __forceinline__ void tail_sync_done(const TailArgs& a, const int G, const int grp, const uint32_t base) {
    if (threadIdx.x == 0) {
        const uint32_t cur = __hip_atomic_load(a.sync + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur - base != (uint32_t)HP) __hip_atomic_fetch_or(a.sync + 2 * G, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.sync + G + grp, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

