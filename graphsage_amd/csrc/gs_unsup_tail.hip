// Fused tail of the UNSUPERVISED two-layer mean model (models.py:362-405, prediction.py:68-110): everything between the
// layer-0 contraction and the weight gradients as TWO launches instead of four latency chains
// (gs_sage_tail_z | gs_linkpred_norm_fwd_bwd | linkpred_neg_bwd | gs_sage_tail_dh0, 42 us with the chip idle):
//
//   launch 1  sage_lp_tail_kernel     z helpers (layer 1: neighbor mean + both contractions, gs_tail_dev.h) -> MAIN workgroups
//             (8 PAIRS each = 8 batch1 rows + their 8 batch2 partners; the 20 negatives are every main's input):
//               y = l2_normalize(z)                                              models.py:368-370
//               aff = <y1, y2>, neg_aff = <y1, neg_j>; xent loss, MRR rank        prediction.py:102-110, models.py:393-405
//               dY of the pair rows, the negatives' partial gradient (one slab per main), dz = l2norm'(dY)
//               [d_self | d_means] = [dz[:, :O] . W_self^T | dz[:, O:] . W_neigh^T],  d_h0 = relu'(h0) * (... , / s)
//             + gather riders: like the supervised tail the launch is LONG and THIN (66 main workgroups), so the rest of the
//             chip streams a share of the NEXT step's gather at the full HBM rate while the chain's latency hides underneath.
//   launch 2  lp_neg_tail_kernel      the negatives' rows need the sum over ALL mains: slab sum in a fixed order ->
//             dz = l2norm'(.) -> their d_h0 (the dh0 body, one workgroup per 16-row group and 128-column slab, each summing
//             the slabs of its rows itself: 1 MB from L2), + the step epilogue block (loss / mrr means, device counters),
//             + the commit of the negative groups' hand-over counters.
//
// Hand-over: as in sage_tail_kernel -- helpers have the lower block indices and never wait; a main waits (bounded, error
// word) for the 2 helpers (one per term) of ITS group and of the negative groups; counters are monotonic, a group's consumed count is
// written by its main (pair groups) or by launch 2 (negative groups: every main reads them).
#include "gs_tail_dev.h"

#define LP_MAX_NEG 32        // negatives held normalised in LDS by every main workgroup

struct LpArgs {
    TailArgs t;              // h0, n = 2 B + n_neg, s, D, Ws / Wn, O, means, z, y, dz, d_h0, sync, pairB = B, pair_groups, train
    int32_t B, n_neg;
    float neg_w, scale;      // neg_sample_weights (prediction.py:108), 1 / batch_size (models.py:378)
    float* loss_rows; float* rr_rows; float* aff_all; int64_t ld_aff;
    float* neg_slabs;        // [pair_groups][n_neg][2 O]: per-main partial gradient w.r.t. the NORMALISED negatives
};

template <int D, int O>
__global__ __launch_bounds__(TAIL_THREADS) void sage_lp_tail_kernel(const LpArgs L, const CoGatherS J) {
    const TailArgs& a = L.t;
    constexpr int HP = 2;                              // helper workgroups per group: one per TERM (tail_z_term_helper)
    constexpr int Z = 2 * O, DJ = Z / 64;
    const int GP = a.pair_groups;
    const int NGH = (L.n_neg + TAIL_ROWS - 1) / TAIL_ROWS;
    const int G = GP + NGH;
    // roles by block index: helpers of the NEGATIVE groups first (every main waits for them), then the pair groups'
    // helpers, then the mains, then gather riders.  2 (64 + 2) helpers + 64 mains = 196 workgroups for B = 512: all resident
    // at once (one per CU), so the mains prefetch their operands while the helpers compute z.
    if ((int)blockIdx.x >= HP * G + GP) {
        run_gather_item<13>(J, ((int64_t)blockIdx.x - (HP * G + GP)) * TAIL_WAVES + (threadIdx.x >> 6), threadIdx.x & 63);
        return;
    }
    if ((int)blockIdx.x < HP * NGH) {
        tail_z_term_helper<D, O>(a, GP + (int)blockIdx.x / HP, (int)blockIdx.x % HP);
        return;
    }
    if ((int)blockIdx.x < HP * G) {
        const int b = (int)blockIdx.x - HP * NGH;
        tail_z_term_helper<D, O>(a, b / HP, b % HP);
        return;
    }
    const int grp = (int)blockIdx.x - HP * G;
    constexpr int ldzs = Z + 4;
    constexpr int D4 = D / 4;
    constexpr int PASSES = TAIL_ROWS * D4 / TAIL_THREADS;
    constexpr int DSLABS = 2 * D / 32, DPW = DSLABS / TAIL_WAVES, M7 = O / 16, ldi = 2 * D + 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* DIN = lds;                                  // [16][2D + 8]   [d_self | d_means]
    float* Zs = DIN + TAIL_ROWS * ldi;                 // [16][ldzs]     z, then y (rows 0..7 batch1, 8..15 batch2)
    float* DZs = Zs + TAIL_ROWS * ldzs;                // [16][ldzs]     dLoss/dz
    float* Ns = DZs + TAIL_ROWS * ldzs;                // [LP_MAX_NEG][ldzs]  normalised negatives
    float* gqs = Ns + LP_MAX_NEG * ldzs;               // [8][64]        dLoss/d(neg_aff) of pair p, negative q
    float* invs = gqs + 8 * 64;                        // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int n = (int)a.n, s = a.s, ldh0 = (int)a.ldh;
    const int B = L.B, n_neg = L.n_neg;
    const float inv_s = 1.0f / (float)s;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ================= S0: this thread's h0 rows -> relu mask bits (phase 8 reads h0 exactly once, as bit flags)
    uint32_t mself[PASSES], mnb[PASSES][2];
    {
        f32x4 hself[PASSES], hnb[PASSES][TAIL_NB];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int it = tid + p * TAIL_THREADS;
            const int r = it / D4, c = (it % D4) * 4;
            bool valid;
            const int i = tail_row(a, grp, r, valid);
            hself[p] = *reinterpret_cast<const f32x4*>(a.h0 + i * ldh0 + c);
            const float* nb = a.h0 + (n + i * s) * ldh0 + c;
#pragma unroll
            for (int u = 0; u < TAIL_NB; ++u) hnb[p][u] = *reinterpret_cast<const f32x4*>(nb + min(u, s - 1) * ldh0);
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            mnb[p][0] = mnb[p][1] = 0u;
#pragma unroll
            for (int u = 0; u < TAIL_NB; ++u) {
                const f32x4 v = hnb[p][u];
                const uint32_t bits = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
                mnb[p][u >> 3] |= bits << (4 * (u & 7));
            }
            const f32x4 hs = hself[p];
            mself[p] = (hs.x > 0.f ? 1u : 0u) | (hs.y > 0.f ? 2u : 0u) | (hs.z > 0.f ? 4u : 0u) | (hs.w > 0.f ? 8u : 0u);
            asm volatile("" : "+v"(mself[p]), "+v"(mnb[p][0]), "+v"(mnb[p][1]));      // pin the flags, free the rows
        }
    }
    // ================= S1: the weight slabs of the input-gradient contraction (they land while the helpers compute z)
    f32x4 b7[DPW][M7][2];
    if (a.train) {
#pragma unroll
        for (int sl = 0; sl < DPW; ++sl) {
            const int col0 = (wave + sl * TAIL_WAVES) * 32;          // in [0, 2D)
            const int term = col0 >= D ? 1 : 0;
            const int ldw = (int)(term ? a.ldwn : a.ldws);
            const float* B0 = (term ? a.Wn : a.Ws) + (col0 - term * D + j) * ldw + 4 * q;
#pragma unroll
            for (int m = 0; m < M7; ++m) {
                b7[sl][m][0] = *reinterpret_cast<const f32x4*>(B0 + 16 * m);
                b7[sl][m][1] = *reinterpret_cast<const f32x4*>(B0 + 16 * ldw + 16 * m);
            }
        }
    }

    // ---------------- phase 1: pick up z of the own group and of the negatives (bounded waits, see sage_tail_kernel)
    uint32_t sync_base = 0u;
    if (tid == 0) {
        sync_base = __hip_atomic_load(a.sync + G + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0u;
        bool gave_up = false;
        while (__hip_atomic_load(a.sync + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - sync_base < (uint32_t)HP) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { gave_up = true; break; }
        }
        for (int ng = 0; ng < NGH && !gave_up; ++ng) {
            const uint32_t nb_ = __hip_atomic_load(a.sync + G + GP + ng, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(a.sync + GP + ng, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - nb_ < (uint32_t)HP) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { gave_up = true; break; }
            }
        }
        if (gave_up) __hip_atomic_fetch_or(a.sync + 2 * G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    {
        // device-scope loads (they bypass this XCD's possibly stale L2 lines of z; no cache invalidation needed)
        constexpr int Z2 = Z / 2;
        for (int it = tid; it < (TAIL_ROWS + n_neg) * Z2; it += TAIL_THREADS) {
            const int r = it / Z2, c = (it % Z2) * 2;
            int src;
            float* dst;
            if (r < TAIL_ROWS) {
                bool valid;
                src = tail_row(a, grp, r, valid);
                dst = Zs + r * ldzs + c;
            } else {
                src = 2 * B + (r - TAIL_ROWS);
                dst = Ns + (r - TAIL_ROWS) * ldzs + c;
            }
            union { f32x2 f; unsigned long long u; } cv;
            cv.u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(a.z + src * (int)a.ldz + c), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<f32x2*>(dst) = cv.f;
        }
    }
    lds_barrier();

    // ---------------- phase 2: y = l2_normalize(z) (models.py:368-370): two own rows per wave, the negatives round-robin
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = wave * 2 + rr;
        bool valid;
        const int i = tail_row(a, grp, row, valid);
        float v[DJ];
        float ss = 0.f;
#pragma unroll
        for (int m = 0; m < DJ; ++m) {
            v[m] = Zs[row * ldzs + lane + 64 * m];
            ss += v[m] * v[m];
        }
        ss = tail_wave_sum(ss);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-12f));
#pragma unroll
        for (int m = 0; m < DJ; ++m) {
            const float y = v[m] * inv;
            Zs[row * ldzs + lane + 64 * m] = y;
            if (valid) a.y[i * (int)a.ldy + lane + 64 * m] = y;
        }
        if (lane == 0) invs[row] = inv;
    }
    for (int nq = wave; nq < n_neg; nq += TAIL_WAVES) {
        float v[DJ];
        float ss = 0.f;
#pragma unroll
        for (int m = 0; m < DJ; ++m) {
            v[m] = Ns[nq * ldzs + lane + 64 * m];
            ss += v[m] * v[m];
        }
        ss = tail_wave_sum(ss);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-12f));
#pragma unroll
        for (int m = 0; m < DJ; ++m) {
            const float y = v[m] * inv;
            Ns[nq * ldzs + lane + 64 * m] = y;
            if (grp == 0) a.y[(2 * B + nq) * (int)a.ldy + lane + 64 * m] = y;
        }
    }
    lds_barrier();

    // ---------------- phase 3: ONE WAVE PER PAIR: affinities, loss, MRR rank, dY -> dz of its two rows (prediction.py:102-110)
    {
        const int p = wave;
        const int pi = 8 * grp + p;
        const bool live = pi < B;
        float o1[DJ], o2[DJ], g1[DJ];
        float aff = 0.f;
#pragma unroll
        for (int m = 0; m < DJ; ++m) {
            o1[m] = Zs[p * ldzs + lane + 64 * m];
            o2[m] = Zs[(8 + p) * ldzs + lane + 64 * m];
            aff += o1[m] * o2[m];
        }
        aff = tail_wave_sum(aff);
        const float ea = __expf(-fabsf(aff));
        const float ra = __builtin_amdgcn_rcpf(1.0f + ea);
        const float sa = aff >= 0.f ? ra : ea * ra;
        const float da = (sa - 1.0f) * L.scale;
        float loss = fmaxf(aff, 0.f) - aff + __logf(1.0f + ea);
#pragma unroll
        for (int m = 0; m < DJ; ++m) g1[m] = da * o2[m];
        // the n_neg <= 32 affinities: four reductions in flight, then lane qn holds negative qn's
        float nav = 0.f;
        int qn = 0;
        for (; qn + 4 <= n_neg; qn += 4) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            const float* nr = Ns + qn * ldzs + lane;
#pragma unroll
            for (int m = 0; m < DJ; ++m) {
                p0 += o1[m] * nr[64 * m];
                p1 += o1[m] * nr[ldzs + 64 * m];
                p2 += o1[m] * nr[2 * ldzs + 64 * m];
                p3 += o1[m] * nr[3 * ldzs + 64 * m];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                p0 += __shfl_xor(p0, off, 64);
                p1 += __shfl_xor(p1, off, 64);
                p2 += __shfl_xor(p2, off, 64);
                p3 += __shfl_xor(p3, off, 64);
            }
            nav = lane == qn ? p0 : nav;
            nav = lane == qn + 1 ? p1 : nav;
            nav = lane == qn + 2 ? p2 : nav;
            nav = lane == qn + 3 ? p3 : nav;
        }
        for (; qn < n_neg; ++qn) {
            float p0 = 0.f;
#pragma unroll
            for (int m = 0; m < DJ; ++m) p0 += o1[m] * Ns[qn * ldzs + lane + 64 * m];
            p0 = tail_wave_sum(p0);
            nav = lane == qn ? p0 : nav;
        }
        const bool in = lane < n_neg;
        const float e = __expf(-fabsf(nav));
        const float r1 = __builtin_amdgcn_rcpf(1.0f + e);
        const float sg = nav >= 0.f ? r1 : e * r1;                   // sigmoid(nav)
        loss += L.neg_w * tail_wave_sum(in ? fmaxf(nav, 0.f) + __logf(1.0f + e) : 0.f);
        const int rank = __popcll(__ballot(in && nav >= aff));
        const float gqv = (in && live) ? L.neg_w * L.scale * sg : 0.f;
        gqs[p * 64 + lane] = gqv;
        if (live) {
            if (L.aff_all && in) L.aff_all[(int64_t)pi * L.ld_aff + lane] = nav;
            if (lane == 0) {
                L.loss_rows[pi] = loss;
                L.rr_rows[pi] = 1.0f / (float)(rank + 1);
                if (L.aff_all) L.aff_all[(int64_t)pi * L.ld_aff + n_neg] = aff;
            }
        }
        if (a.train) {
            for (qn = 0; qn < n_neg; ++qn) {
                const float gq = __shfl(gqv, qn, 64);
#pragma unroll
                for (int m = 0; m < DJ; ++m) g1[m] += gq * Ns[qn * ldzs + lane + 64 * m];
            }
            // back through y = z * inv:  dz = inv (g - y <g, y>);  clamped (sum z^2 < 1e-12, inv = 1e6): dz = g * inv
            float dot1 = 0.f, dot2 = 0.f;
#pragma unroll
            for (int m = 0; m < DJ; ++m) {
                dot1 += g1[m] * o1[m];
                dot2 += da * o1[m] * o2[m];
            }
            dot1 = tail_wave_sum(dot1);
            dot2 = tail_wave_sum(dot2);
            const float inv1 = invs[p], inv2 = invs[8 + p];
            const bool c1 = inv1 >= 1.0e6f, c2 = inv2 >= 1.0e6f;
#pragma unroll
            for (int m = 0; m < DJ; ++m) {
                const float ga = g1[m], gb = da * o1[m];
                const float d1 = live ? (c1 ? ga * inv1 : inv1 * (ga - o1[m] * dot1)) : 0.f;
                const float d2 = live ? (c2 ? gb * inv2 : inv2 * (gb - o2[m] * dot2)) : 0.f;
                DZs[p * ldzs + lane + 64 * m] = d1;
                DZs[(8 + p) * ldzs + lane + 64 * m] = d2;
                if (live) {
                    a.dz[(int64_t)pi * a.lddz + lane + 64 * m] = d1;
                    a.dz[(int64_t)(B + pi) * a.lddz + lane + 64 * m] = d2;
                }
            }
        }
    }
    if (!a.train) {
        tail_sync_done<HP>(a, G, grp, sync_base);
        return;
    }
    lds_barrier();

    // ---------------- phase 4: this main's slab of the negatives' gradient: slab[q][c] = sum_p gq[p][q] * y1[p][c], p = 0..7
    {
        constexpr int Z4 = Z / 4;
        float* slab = L.neg_slabs + (size_t)grp * n_neg * Z;
        for (int it = tid; it < n_neg * Z4; it += TAIL_THREADS) {
            const int qn = it / Z4, c = (it % Z4) * 4;
            f32x4 acc = zero4;
#pragma unroll
            for (int p = 0; p < 8; ++p) acc += *reinterpret_cast<const f32x4*>(Zs + p * ldzs + c) * gqs[p * 64 + qn];
            *reinterpret_cast<f32x4*>(slab + (size_t)qn * Z + c) = acc;
        }
    }

    // ---------------- phase 5: [d_self | d_means] = [d_z[:, :O] . W_self^T | d_z[:, O:] . W_neigh^T]  -> DIN
#pragma unroll
    for (int sl = 0; sl < DPW; ++sl) {
        const int col0 = (wave + sl * TAIL_WAVES) * 32;
        const int term = col0 >= D ? 1 : 0;
        const float* A = DZs + term * O + j * ldzs + 4 * q;
        f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
        for (int m = 0; m < M7; ++m) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + 16 * m);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = mfma16(a4[e], b7[sl][m][0][e], acc0);
                acc1 = mfma16(a4[e], b7[sl][m][1][e], acc1);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            DIN[(4 * q + i) * ldi + col0 + j] = acc0[i];
            DIN[(4 * q + i) * ldi + col0 + 16 + j] = acc1[i];
        }
    }
    lds_barrier();

    // ---------------- phase 6: d_h0 = relu'(h0) * (d_self on the self row, d_means / s on each of the s neighbor rows)
    {
        const int lddh0 = (int)a.lddh;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int it = tid + p * TAIL_THREADS;
            const int r = it / D4, c = (it % D4) * 4;
            bool valid;
            const int i = tail_row(a, grp, r, valid);
            if (valid) {
                const f32x4 g_self = *reinterpret_cast<const f32x4*>(DIN + r * ldi + c);
                const f32x4 g_mean = *reinterpret_cast<const f32x4*>(DIN + r * ldi + D + c) * inv_s;
                f32x4 o;
                o.x = (mself[p] & 1u) ? g_self.x : 0.f;
                o.y = (mself[p] & 2u) ? g_self.y : 0.f;
                o.z = (mself[p] & 4u) ? g_self.z : 0.f;
                o.w = (mself[p] & 8u) ? g_self.w : 0.f;
                *reinterpret_cast<f32x4*>(a.d_h0 + i * lddh0 + c) = o;
                float* dst = a.d_h0 + (n + i * s) * lddh0 + c;
#pragma unroll
                for (int u = 0; u < TAIL_NB; ++u) {
                    if (u < s) {
                        const uint32_t bits = mnb[p][u >> 3] >> (4 * (u & 7));
                        o.x = (bits & 1u) ? g_mean.x : 0.f;
                        o.y = (bits & 2u) ? g_mean.y : 0.f;
                        o.z = (bits & 4u) ? g_mean.z : 0.f;
                        o.w = (bits & 8u) ? g_mean.w : 0.f;
                        *reinterpret_cast<f32x4*>(dst + u * lddh0) = o;
                    }
                }
            }
        }
    }
    tail_sync_done<HP>(a, G, grp, sync_base);
}

// Launch 2 (see the file comment).  Blocks [0, NWG NGH): workgroup (negative group ng, 128-column slab `part` of
// [d_self | d_means]); block NWG NGH: the step epilogue + the commit of the negative groups' consumed counters.
// train == 0 (evaluation): only that last block.
template <int D, int O>
__global__ __launch_bounds__(TAIL_THREADS, 2) void lp_neg_tail_kernel(const LpArgs L, const StepEpilogue epi, const CoGatherS J) {
    const TailArgs& a = L.t;
    constexpr int HP = 2;                              // as in sage_lp_tail_kernel
    constexpr int NWG = 2 * D / 128, Z = 2 * O, Z4 = Z / 4, M7 = O / 16;
    constexpr int ldzs = Z + 4, ldi = 128 + 4;
    const int GP = a.pair_groups;
    const int NGH = (L.n_neg + TAIL_ROWS - 1) / TAIL_ROWS;
    const int G = GP + NGH;
    const int work_blocks = a.train ? NWG * NGH : 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x > work_blocks) {
        // gather riders: this launch is nine workgroups of dependent round trips -- the rest of the chip streams a share of the
        // next step's gather meanwhile (two 8-wave workgroups per CU)
        run_gather_item<13>(J, ((int64_t)blockIdx.x - work_blocks - 1) * TAIL_WAVES + wave, lane);
        return;
    }
    if ((int)blockIdx.x == work_blocks) {
        float* red = lds;
        if (tid == 0) {
            // every main of launch 1 has finished (kernel boundary): the arrivals of the negative groups are consumed; exactly
            // HP helpers per group must have arrived since the last commit
            for (int ng = 0; ng < NGH; ++ng) {
                const uint32_t cur = __hip_atomic_load(a.sync + GP + ng, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t base = __hip_atomic_load(a.sync + G + GP + ng, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur - base != (uint32_t)HP) __hip_atomic_fetch_or(a.sync + 2 * G, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.sync + G + GP + ng, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // the step epilogue (gs_step_epilogue_block's sums, in its order, on a 512-thread block: the first four waves add,
        // every wave meets the barriers)
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const float* rows = which ? epi.loss_rows : epi.aux_rows;
            if (!rows) continue;                                     // workgroup-uniform
            float sacc = 0.f;
            if (tid < 256)
                for (int64_t i = tid; i < epi.n; i += 256) sacc += rows[i];
            sacc = tail_wave_sum(sacc);
            if (lane == 0) red[8 * which + wave] = sacc;
            __syncthreads();
            if (tid == 0) {
                const float tot = ((red[8 * which] + red[8 * which + 1]) + (red[8 * which + 2] + red[8 * which + 3]));
                if (which) epi.loss_out[0] = epi.accumulate ? epi.loss_out[0] + tot * epi.scale : tot * epi.scale;
                else epi.aux_out[0] = tot * epi.aux_scale;
            }
        }
        if (tid == 0) {
            if (epi.c0) *epi.c0 += epi.d0;
            if (epi.c1) *epi.c1 += epi.d1;
            if (epi.c2) *epi.c2 += epi.d2;
        }
        return;
    }
    const int ng = (int)blockIdx.x / NWG, part = (int)blockIdx.x % NWG;
    const int colbase = part * 128;
    const int term = colbase >= D ? 1 : 0;
    const int cb = colbase - term * D;
    float* DZs = lds;                                  // [16][ldzs]   dz of the group's 16 negative rows (all Z columns)
    float* DIN = DZs + TAIL_ROWS * ldzs;               // [16][ldi]    this slab of [d_self | d_means]
    const int j = lane & 15, q = lane >> 4;
    const int n = (int)a.n, s = a.s, ldh0 = (int)a.ldh, B = L.B, n_neg = L.n_neg;
    const int r0 = 2 * B + TAIL_ROWS * ng;             // first row of the group
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // ---- inputs that depend on nothing computed here: issue them first
    f32x4 b7[M7];
    {
        const int ldw = (int)(term ? a.ldwn : a.ldws);
        const float* B0 = (term ? a.Wn : a.Ws) + (cb + 16 * wave + j) * ldw + 4 * q;
#pragma unroll
        for (int m = 0; m < M7; ++m) b7[m] = *reinterpret_cast<const f32x4*>(B0 + 16 * m);
    }
    const int hr = tid >> 5, hc = cb + 4 * (tid & 31);
    const int hi = min(r0 + hr, n - 1);
    f32x4 hv[TAIL_NB];
    if (term == 0) {
        hv[0] = *reinterpret_cast<const f32x4*>(a.h0 + hi * ldh0 + hc);
    } else {
        const float* nb = a.h0 + (n + hi * s) * ldh0 + hc;
#pragma unroll
        for (int u = 0; u < TAIL_NB; ++u) hv[u] = *reinterpret_cast<const f32x4*>(nb + min(u, s - 1) * ldh0);
    }
    // ---- the rows' gradient w.r.t. their normalised embeddings: slabs of all mains, summed in main order (16 loads in flight);
    //      wave w owns rows w and w + 8, lane = float4 column
    // (both rows' slab loads are issued together: 2 x 16 float4 in flight per lane, 4 round trips for 64 slabs instead of 8)
    f32x4 g2[2] = {zero4, zero4}, zv2[2] = {zero4, zero4};
    {
        const int qn0 = TAIL_ROWS * ng + wave, qn1 = qn0 + 8;
        const bool v0 = qn0 < n_neg && lane < Z4, v1 = qn1 < n_neg && lane < Z4;
        const size_t stride4 = (size_t)n_neg * Z4;
        const f32x4* sp0 = reinterpret_cast<const f32x4*>(L.neg_slabs + (size_t)min(qn0, n_neg - 1) * Z) + min(lane, Z4 - 1);
        const f32x4* sp1 = reinterpret_cast<const f32x4*>(L.neg_slabs + (size_t)min(qn1, n_neg - 1) * Z) + min(lane, Z4 - 1);
        if (v0) zv2[0] = *reinterpret_cast<const f32x4*>(a.z + (int64_t)(2 * B + qn0) * a.ldz + 4 * lane);
        if (v1) zv2[1] = *reinterpret_cast<const f32x4*>(a.z + (int64_t)(2 * B + qn1) * a.ldz + 4 * lane);
        int sI = 0;
        for (; sI + 16 <= GP; sI += 16) {
            f32x4 va[16], vb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                va[u] = sp0[(size_t)(sI + u) * stride4];
                vb[u] = sp1[(size_t)(sI + u) * stride4];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                g2[0] += va[u];
                g2[1] += vb[u];
            }
        }
        for (; sI < GP; ++sI) {
            g2[0] += sp0[(size_t)sI * stride4];
            g2[1] += sp1[(size_t)sI * stride4];
        }
        if (!v0) g2[0] = zero4;
        if (!v1) g2[1] = zero4;
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = wave + 8 * rr;                 // local row
        const int qn = TAIL_ROWS * ng + row;           // negative index
        const bool rvalid = qn < n_neg;                // wave-uniform
        const f32x4 g = g2[rr], zv = zv2[rr];
        float ss = (zv.x * zv.x + zv.y * zv.y) + (zv.z * zv.z + zv.w * zv.w);
        ss = tail_wave_sum(ss);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-12f));
        float dot = ((g.x * zv.x + g.y * zv.y) + (g.z * zv.z + g.w * zv.w)) * inv;
        dot = tail_wave_sum(dot);
        const bool clamped = inv >= 1.0e6f;
        if (lane < Z4) {
            const f32x4 o = rvalid ? (clamped ? g * inv : (g - zv * (inv * dot)) * inv) : zero4;
            *reinterpret_cast<f32x4*>(DZs + row * ldzs + 4 * lane) = o;
            if (rvalid && part == 0) *reinterpret_cast<f32x4*>(a.dz + (int64_t)(2 * B + qn) * a.lddz + 4 * lane) = o;
        }
    }
    lds_barrier();
    // ---- the slab of [d_self | d_means]: one 16 x 16 tile per wave
    {
        const float* A = DZs + term * O + j * ldzs + 4 * q;
        f32x4 acc = zero4;
#pragma unroll
        for (int m = 0; m < M7; ++m) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + 16 * m);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma16(a4[e], b7[m][e], acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) DIN[(4 * q + i) * ldi + 16 * wave + j] = acc[i];
    }
    lds_barrier();
    // ---- d_h0: relu mask (+ the 1/s broadcast over the row's s neighbor rows for the d_means slabs)
    if (r0 + hr < n) {
        const int lddh0 = (int)a.lddh;
        f32x4 g = *reinterpret_cast<const f32x4*>(DIN + hr * ldi + 4 * (tid & 31));
        if (term == 0) {
            f32x4 o;
            o.x = hv[0].x > 0.f ? g.x : 0.f;
            o.y = hv[0].y > 0.f ? g.y : 0.f;
            o.z = hv[0].z > 0.f ? g.z : 0.f;
            o.w = hv[0].w > 0.f ? g.w : 0.f;
            *reinterpret_cast<f32x4*>(a.d_h0 + (r0 + hr) * lddh0 + hc) = o;
        } else {
            g *= 1.0f / (float)s;
            float* dst = a.d_h0 + (n + (r0 + hr) * s) * lddh0 + hc;
#pragma unroll
            for (int u = 0; u < TAIL_NB; ++u) {
                if (u < s) {
                    f32x4 o;
                    o.x = hv[u].x > 0.f ? g.x : 0.f;
                    o.y = hv[u].y > 0.f ? g.y : 0.f;
                    o.z = hv[u].z > 0.f ? g.z : 0.f;
                    o.w = hv[u].w > 0.f ? g.w : 0.f;
                    *reinterpret_cast<f32x4*>(dst + u * lddh0) = o;
                }
            }
        }
    }
}

static size_t lp_tail_lds_bytes(int D, int O) {
    const int Z = 2 * O;
    const size_t main_f = (size_t)TAIL_ROWS * (2 * D + 8) + (size_t)2 * TAIL_ROWS * (Z + 4) + (size_t)LP_MAX_NEG * (Z + 4) + 8 * 64 + TAIL_ROWS;
    const size_t helper_f = (size_t)TAIL_ROWS * (D + 4) + (size_t)TAIL_WAVES * TAIL_ROWS * O;      // tail_z_term_helper: O columns
    return std::max(main_f, helper_f) * sizeof(float);
}

extern "C" int gs_linkpred_tail_supported(int32_t d_in, int32_t out_dim, int32_t n_neg) {
    const bool ok = (d_in == 128 || d_in == 256) && (out_dim == 64 || out_dim == 128) && n_neg >= 1 && n_neg <= LP_MAX_NEG &&
                    lp_tail_lds_bytes(d_in, out_dim) <= 160 * 1024;
    return ok ? 1 : 0;
}

template <int D, int O>
static int launch_lp_tail(const LpArgs& L, const CoGatherS& J, int64_t gather_waves, hipStream_t st) {
    GS_LDS_ATTR(160 * 1024, sage_lp_tail_kernel<D, O>);
    const int NGH = (L.n_neg + TAIL_ROWS - 1) / TAIL_ROWS;
    const int G = L.t.pair_groups + NGH;
    const int64_t blocks = (int64_t)2 * G + L.t.pair_groups + gs_ceil_div(gather_waves, TAIL_WAVES);
    GS_REQUIRE(blocks < (1ll << 31), "gs_linkpred_tail: grid too large");
    hipLaunchKernelGGL((sage_lp_tail_kernel<D, O>), dim3((unsigned)blocks), dim3(TAIL_THREADS), lp_tail_lds_bytes(D, O), st, L, J);
    GS_LAUNCH_CHECK("sage_lp_tail_kernel");
    return GS_OK;
}

template <int D, int O>
static int launch_lp_neg(const LpArgs& L, const StepEpilogue& epi, const CoGatherS& J, int64_t gather_waves, hipStream_t st) {
    const int NGH = (L.n_neg + TAIL_ROWS - 1) / TAIL_ROWS;
    const size_t lds = ((size_t)TAIL_ROWS * (2 * O + 4) + (size_t)TAIL_ROWS * (128 + 4)) * sizeof(float);
    const int64_t blocks = (L.t.train ? (2 * D / 128) * NGH : 0) + 1 + gs_ceil_div(gather_waves, TAIL_WAVES);
    GS_REQUIRE(blocks < (1ll << 31), "gs_linkpred_tail_neg: grid too large");
    hipLaunchKernelGGL((lp_neg_tail_kernel<D, O>), dim3((unsigned)blocks), dim3(TAIL_THREADS), lds, st, L, epi, J);
    GS_LAUNCH_CHECK("lp_neg_tail_kernel");
    return GS_OK;
}

static int lp_args(const gs_lp_tail_desc* q, LpArgs* out, const char* who) {
    GS_REQUIRE(q, "%s: null descriptor", who);
    GS_REQUIRE(q->B > 0 && q->n_neg > 0 && q->s > 0, "%s: bad sizes", who);
    if (q->s > TAIL_NB || !gs_linkpred_tail_supported(q->d_in, q->out_dim, q->n_neg)) {
        gs_set_error("%s: unsupported shape d_in=%d out_dim=%d n_neg=%d s=%d (d_in in {128,256}, out_dim in {64,128}, n_neg <= %d, s <= %d)",
                     who, q->d_in, q->out_dim, q->n_neg, q->s, LP_MAX_NEG, TAIL_NB);
        return GS_ENOTSUP;
    }
    const int D = q->d_in, O = q->out_dim, Z = 2 * O;
    const int64_t n = 2 * q->B + q->n_neg;
    GS_REQUIRE((n + n * (int64_t)q->s) * std::max(q->ldh, std::max(q->lddh, (int64_t)1)) < (1ll << 31),
               "%s: (n + n*s) * ld must be < 2^31 (32-bit row offsets)", who);
    GS_CHECK_MAT(q->h0, q->ldh, "gs_linkpred_tail h0");
    GS_CHECK_MAT(q->W_self, q->ldws, "gs_linkpred_tail W_self");
    GS_CHECK_MAT(q->W_neigh, q->ldwn, "gs_linkpred_tail W_neigh");
    GS_CHECK_MAT(q->means, q->ldm, "gs_linkpred_tail means");
    GS_CHECK_MAT(q->z, q->ldz, "gs_linkpred_tail z");
    GS_CHECK_MAT(q->y, q->ldy, "gs_linkpred_tail y");
    GS_REQUIRE(q->ldh >= D && q->ldws >= O && q->ldwn >= O && q->ldm >= D && q->ldz >= Z && q->ldy >= Z, "%s: leading dimension too small", who);
    GS_REQUIRE(q->loss_rows && q->rr_rows && q->sync && (!q->aff_all || q->ld_aff >= q->n_neg + 1), "%s: loss_rows / rr_rows / sync missing", who);
    if (q->train) {
        GS_CHECK_MAT(q->dz, q->lddz, "gs_linkpred_tail dz");
        GS_CHECK_MAT(q->d_h0, q->lddh, "gs_linkpred_tail d_h0");
        GS_REQUIRE(q->lddz >= Z && q->lddh >= D && q->neg_slabs && gs_aligned16(q->neg_slabs), "%s: gradient buffers missing / too small", who);
    }
    LpArgs L = {};
    TailArgs& a = L.t;
    a.h0 = q->h0; a.ldh = q->ldh; a.n = n; a.s = q->s; a.D = D;
    a.Ws = q->W_self; a.ldws = q->ldws; a.Wn = q->W_neigh; a.ldwn = q->ldwn; a.O = O;
    a.means = q->means; a.ldm = q->ldm; a.z = q->z; a.ldz = q->ldz; a.y = q->y; a.ldy = q->ldy;
    a.dz = q->dz; a.lddz = q->lddz; a.d_h0 = q->d_h0; a.lddh = q->lddh;
    a.train = q->train ? 1 : 0;
    a.sync = q->sync;
    a.pairB = (int32_t)q->B;
    a.pair_groups = (int32_t)gs_ceil_div(q->B, 8);
    L.B = (int32_t)q->B; L.n_neg = q->n_neg; L.neg_w = q->neg_weight; L.scale = q->scale;
    L.loss_rows = q->loss_rows; L.rr_rows = q->rr_rows; L.aff_all = q->aff_all; L.ld_aff = q->ld_aff;
    L.neg_slabs = q->neg_slabs;
    *out = L;
    return GS_OK;
}

extern "C" int gs_linkpred_tail(const gs_lp_tail_desc* q, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    LpArgs L;
    int rc = lp_args(q, &L, "gs_linkpred_tail");
    if (rc != GS_OK) return rc;
    CoGatherS J = {};
    int64_t gw = 0;
    rc = build_cojobs_s(jobs_host, n_jobs, &J, &gw);
    if (rc != GS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int D = q->d_in, O = q->out_dim;
    if (D == 256 && O == 128) return launch_lp_tail<256, 128>(L, J, gw, st);
    if (D == 256 && O == 64) return launch_lp_tail<256, 64>(L, J, gw, st);
    if (D == 128 && O == 128) return launch_lp_tail<128, 128>(L, J, gw, st);
    return launch_lp_tail<128, 64>(L, J, gw, st);
}

extern "C" int gs_linkpred_tail_neg(const gs_lp_tail_desc* q, float* loss_out, int accumulate, float* mrr_out, uint64_t* c0,
                                    uint64_t d0, uint64_t* c1, uint64_t d1, uint64_t* c2, uint64_t d2,
                                    const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    LpArgs L;
    int rc = lp_args(q, &L, "gs_linkpred_tail_neg");
    if (rc != GS_OK) return rc;
    CoGatherS J = {};
    int64_t gw = 0;
    rc = build_cojobs_s(jobs_host, n_jobs, &J, &gw);
    if (rc != GS_OK) return rc;
    StepEpilogue epi = {};
    if (loss_out) {
        GS_REQUIRE(mrr_out, "gs_linkpred_tail_neg: mrr_out missing");
        const float inv_b = 1.0f / (float)q->B;
        epi = StepEpilogue{q->loss_rows, q->B, inv_b, loss_out, accumulate, q->rr_rows, inv_b, mrr_out, c0, d0, c1, d1, c2, d2};
    }
    hipStream_t st = (hipStream_t)stream;
    const int D = q->d_in, O = q->out_dim;
    if (D == 256 && O == 128) return launch_lp_neg<256, 128>(L, epi, J, gw, st);
    if (D == 256 && O == 64) return launch_lp_neg<256, 64>(L, epi, J, gw, st);
    if (D == 128 && O == 128) return launch_lp_neg<128, 128>(L, epi, J, gw, st);
    return launch_lp_neg<128, 64>(L, epi, J, gw, st);
}
