// Fused "tail" of the supervised two-layer mean model: EVERYTHING between the layer-0 contraction and the weight
// gradients, forward and backward, in ONE launch:
//
//   layer 1 (last, identity act):  means = reduce_mean(h0[neighbors], 1)            aggregators.py:48
//                                  z = [h0_self . W_self | means . W_neigh]         aggregators.py:51-58
//   head:                          y = l2_normalize(z); logits = y . W + b          supervised_models.py:85-92
//                                  loss rows, preds, dlogits                        supervised_models.py:111-126
//   backward:                      d_y = dlogits . W^T;  d_z = l2norm'(d_y)
//                                  d_self = d_z[:, :O] . W_self^T,  d_means = d_z[:, O:] . W_neigh^T
//                                  d_h0 = relu'(h0) * (d_self on the self rows + d_means / s on each neighbor row)
//
// Why one kernel: on a 512-row batch these are six launches of pure launch/dependency latency (~40 us, 3 % MFMA
// utilisation).  Every batch row is independent given the weights, so a workgroup takes 16 batch rows through the
// whole chain with no inter-workgroup synchronisation at all; phases are separated by workgroup barriers only.
//
// Contractions: v_mfma_f32_16x16x4_f32 (exact fp32, M = 16 rows per workgroup).  The 16-row A operands live in LDS;
// the weight operands are read straight from global memory (L2-resident, < 0.4 MB) into the MFMA B operand registers
// -- no LDS staging, no barriers inside a K loop.  Two operand forms:
//   NN  B[k][n] n-contiguous (forward):  lane (j = l&15, q = l>>4) loads TT consecutive columns of row k+q as one
//       vector; element t of it feeds output tile t, whose columns are {n0 + TT*j + t}.
//   NT  B[n][k] k-contiguous (input gradients):  lane loads 4 consecutive k of weight row n0 + 16t + j; element e
//       feeds MFMA step e, A supplies the same 4 k from LDS (one ds_read_b128).
// 8 waves per workgroup; each wave owns a disjoint column slab of every contraction, except the tiny logits
// contraction, whose K is split across the waves and summed in a fixed order through LDS (deterministic).
#include "gs_common.h"
#include "gs_gather_dev.h"

#ifdef GS_TIMELINE
// Diagnostics build only (-DGS_TIMELINE, benchmarks/timeline_tail.py): wall-clock stamp (100 MHz) per phase boundary of the main
// workgroups (rows [0, 64)) and of the z helpers (rows [64, 64 + 256), 8 stamps each).
__device__ unsigned long long g_tail_timeline[64 * 16 + 256 * 8];
extern "C" int gs_debug_tail_timeline(unsigned long long* out_host, int n) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_tail_timeline), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#define TAIL_STAMP(k) do { if (threadIdx.x == 0 && grp < 64) g_tail_timeline[grp * 16 + (k)] = wall_clock64(); } while (0)
#define TAIL_HELPER_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_tail_timeline[64 * 16 + blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TAIL_STAMP(k) do { } while (0)
#endif
#include "gs_tail_dev.h"
// (diagnostics / A-B: GS_TAIL_HALVES=0 in the environment keeps one main workgroup per group)
static const bool g_tail_halves = [] { const char* e = getenv("GS_TAIL_HALVES"); return !(e && e[0] == '0'); }();

template <int D, int O, int CW, int NH>
__global__ __launch_bounds__(TAIL_THREADS) void sage_tail_kernel(const TailArgs a, const int tail_blocks, const CoGatherS J) {
    // Co-scheduled gather: the tail occupies n/16 CUs for ~30 us of mostly waiting; the other ~220 CUs (one 8-wave
    // workgroup each: the launch's LDS size is uniform) stream a share of the NEXT step's gather+mean from HBM meanwhile.
    // Roles by block index: [0, HP G) z HELPERS, [HP G, (HP + 1) G) the G = n/16 MAIN workgroups, then gather riders.
    // The layer-1 contraction z = [h_self . W_self | mean(h_neigh) . W_neigh] is MFMA-bound when only the G = 32 main
    // workgroups (32 CUs, 128 SIMDs) compute it (7.8 us of matrix-pipe time, plus 7 us to pull its operands through
    // 32 L2->CU ports); HP = Z / 64 helper workgroups per group compute a 16 x 64 slab of it each on their own CUs
    // (K split over the 8 waves, fixed-order sum), publish it to global memory and leave; the main workgroup meanwhile
    // prefetches every later phase's operands and then picks z up.  Helpers never wait for anything and have the lower
    // block indices (dispatched first), so a waiting main workgroup never keeps a helper off the chip for good; all
    // (HP + 1) G workgroups are resident at once for n <= 816 (one 8-wave workgroup per CU).
    // NH = 2 (training launches, D = 256): TWO main workgroups per group.  Both pick z up and run the small head phases
    // redundantly (identical values; half 0 stores them); the input-gradient contraction, the relu masks and the d_h0 stores --
    // 8 of a main workgroup's 12 us behind the pick-up, all of them bound by ONE CU's matrix pipe and store path -- are split by
    // COLUMNS: half h owns columns [128 h, 128 h + 128) of d_self and of d_means (both terms of a column stay together: the GCN
    // form adds them).
    constexpr int HP = 2 * O / 64;
    const int G = tail_blocks;
    const int hp = a.z_ready ? 0 : HP;                   // split form: no helper workgroups in this launch
    if ((int)blockIdx.x >= (hp + NH) * G) {
        // (a rider wave walking 4 consecutive items with prefetched ids, and 25 loads in flight per lane, were measured:
        // 46 us / no change against 35 us -- with one 8-wave workgroup per CU the riders stream at ~4.6 TB/s either way)
        run_gather_item<13>(J, ((int64_t)blockIdx.x - (hp + NH) * G) * TAIL_WAVES + (threadIdx.x >> 6), threadIdx.x & 63);
        return;
    }
    if ((int)blockIdx.x < hp * G) {
        tail_z_helper<D, O>(a, (int)blockIdx.x / HP, (int)blockIdx.x % HP, G);
        return;
    }
    const int mb = (int)blockIdx.x - hp * G;
    const int half = NH > 1 ? mb / G : 0;                // workgroup-uniform
    const int grp = mb - half * G;
    TAIL_STAMP(0);
    constexpr int Z = 2 * O;
    constexpr int ldh = D + 4, ldzs = Z + 4;
    constexpr int D4 = D / 4;
    static_assert(NH == 1 || (NH == 2 && D == 256), "two main workgroups per group: D = 256 only");
    constexpr int D4H = D4 / NH;                                  // float4 columns of a row this workgroup owns
    constexpr int PASSES = TAIL_ROWS * D4H / TAIL_THREADS;        // (row, float4 column) items per thread: D / 128 / NH
    constexpr int ZSLABS = Z / 32;                                // 32-column slabs of z / d_y  (<= 8: one per wave)
    constexpr int DSLABS = 2 * D / 32;                            // 32-column slabs of [d_self | d_means]
    constexpr int DPW = DSLABS / TAIL_WAVES / NH;                 // ... per wave of this workgroup (1 or 2)

    constexpr int M7 = O / 16;                                    // macro steps (16 k each) of the input-gradient contraction
    constexpr int ldi = 2 * D + 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = a.C;
    const int Cp4 = (C + 3) & ~3, Cp32 = (C + 31) & ~31;
    // logits: CW = ceil(C / 64) class groups of 64 (one class of each group per lane); 2 CW column slabs of 32 x
    // 4 / CW K-slices = the 8 waves
    constexpr int lslabs = 2 * CW, nks = 8 / lslabs, GC = 64 * CW;
    constexpr int KL = Z / nks / 4;                   // k-steps (4 k each) per K-slice
    const int ldc = Cp32 + 4;
    float* Hs = lds;                                  // [16][ldh]   self rows of h0      (later: DIN [16][2D+8])
    float* Ms = Hs + TAIL_ROWS * ldh;                 // [16][ldh]   neighbor means
    float* Zs = Ms + TAIL_ROWS * ldh;                 // [16][ldzs]  z, then y
    float* DZs = Zs + TAIL_ROWS * ldzs;               // [16][ldzs]  dLoss/dz
    float* Ps = DZs + TAIL_ROWS * ldzs;               // [nks][16][GC] logits partials   (later: DY [16][ldzs])
    const int ps_floats = max(nks * TAIL_ROWS * GC, TAIL_ROWS * ldzs);
    float* DLs = Ps + ps_floats;                      // [16][ldc]   dlogits, zero-padded to Cp32
    float* invs = DLs + TAIL_ROWS * ldc;              // [16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int r0 = grp * TAIL_ROWS;                   // (n + n*s) * ld < 2^31 is checked on the host: 32-bit offsets
    const int n = (int)a.n;
    const int s = a.s;
    const int ldh0 = (int)a.ldh;
    const float inv_s = 1.0f / (float)s;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ================= S0: issue this thread's h0 rows and the wave's z-contraction weight slab
    f32x4 hself[PASSES], hnb[PASSES][TAIL_NB];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int it = tid + p * TAIL_THREADS;
        const int r = it / D4H, c = (half * D4H + it % D4H) * 4;
        const int i = min(r0 + r, n - 1);
        hself[p] = *reinterpret_cast<const f32x4*>(a.h0 + i * ldh0 + c);
        const float* nb = a.h0 + (n + i * s) * ldh0 + c;
#pragma unroll
        for (int u = 0; u < TAIL_NB; ++u) hnb[p][u] = *reinterpret_cast<const f32x4*>(nb + min(u, s - 1) * ldh0);
    }
    // ---------------- phase 0: the relu mask bits of this thread's h0 rows (for phase 8); the rows themselves are only
    // needed by the z helpers
    uint32_t mself[PASSES], mnb[PASSES][2];            // 4 bits per row: h > 0 of the float4's elements
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
        // pin the flags here: otherwise the compiler keeps the 12 float4 rows alive until phase 8 and re-derives them
        asm volatile("" : "+v"(mself[p]), "+v"(mnb[p][0]), "+v"(mnb[p][1]));
    }
    TAIL_STAMP(1);

    // ================= S1: issue the operands of every later phase (they land while the helpers compute z)
    // phase 3 (logits, NN form): column slab g3, K-slice ks3 of Z/4 k
    const int g3 = wave % lslabs, ks3 = wave / lslabs;
    f32x2 bl[KL];
    {
        const int ldw = (int)a.ldwh;
        // lanes past the last class quad read column 0 instead (finite): their output columns are never read
        const float* B = a.Wh + (ks3 * (4 * KL) + q) * ldw + (g3 * 32 + 2 * j < Cp4 ? g3 * 32 + 2 * j : 0);
#pragma unroll
        for (int u = 0; u < KL; ++u) bl[u] = *reinterpret_cast<const f32x2*>(B + (4 * u) * ldw);
    }
    // phase 4: bias / labels of this wave's two rows (one class of every 64-class group per lane)
    int cl[CW];
    float bias_l[CW], lab2[2][CW];
#pragma unroll
    for (int m = 0; m < CW; ++m) {
        cl[m] = min(lane + 64 * m, C - 1);
        bias_l[m] = a.bh ? a.bh[cl[m]] : 0.f;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) lab2[rr][m] = a.labels[min(r0 + wave * 2 + rr, n - 1) * (int)a.ldlab + cl[m]];
    }
    // phase 7 ([d_self | d_means], NT form): DPW slabs of 32 weight rows, K = O
    f32x4 b7[DPW][M7][2];
    // first of the 32 columns of [d_self | d_means] (in [0, 2D)) of this wave's slab sl
    auto slab_col0 = [&](const int sl) -> int {
        if (NH == 1) return (wave + sl * TAIL_WAVES) * 32;
        return (wave >> 2) * D + half * (D / 2) + (wave & 3) * 32;     // waves 0-3: d_self, 4-7: d_means, this half's 128 columns
    };
    auto load_b7 = [&](const int sl) {
        const int col0 = slab_col0(sl);
        const int term = col0 >= D ? 1 : 0;
        const int ldw = (int)(term ? a.ldwn : a.ldws);
        const float* B0 = (term ? a.Wn : a.Ws) + (col0 - term * D + j) * ldw + 4 * q;
#pragma unroll
        for (int m = 0; m < M7; ++m) {
            b7[sl][m][0] = *reinterpret_cast<const f32x4*>(B0 + 16 * m);
            b7[sl][m][1] = *reinterpret_cast<const f32x4*>(B0 + 16 * ldw + 16 * m);
        }
    };
    // phase 5 (d_y, NT form): rows n0 .. n0+31 of W_head, K = Cp32 <= 64 CW -> up to 4 CW macro steps x 2 tiles
    f32x4 bh5[4 * CW][2];
    {
        const int ldw = (int)a.ldwh;
        const int n0 = (wave < ZSLABS ? wave : 0) * 32;
        const float* B0 = a.Wh + (n0 + j) * ldw;
#pragma unroll
        for (int m = 0; m < 4 * CW; ++m) {
            const int kq = min(16 * m + 4 * q, Cp4 - 4);     // clamped: dlogits are zero beyond C
            bh5[m][0] = *reinterpret_cast<const f32x4*>(B0 + kq);
            bh5[m][1] = *reinterpret_cast<const f32x4*>(B0 + 16 * ldw + kq);
        }
    }
    // (two class groups: the wider logits / d_y operands take the registers of the second input-gradient slab, which is
    // requested after the loss phase instead -- phases 5-6 cover its round trip)
    TAIL_STAMP(2);

    // ---------------- phase 1: pick up z: the helpers' granules (fused form), or the z rows an earlier launch wrote (split form)
    uint32_t z_tag = 0u;
    if (!a.z_ready) {
        // (two class groups: the wider head operands leave registers for half the granules at a time)
        z_tag = tail_pick_up_z<Z, (CW > 1 && TAIL_ROWS * Z / TAIL_THREADS >= 8) ? TAIL_ROWS * Z / TAIL_THREADS / 2 : TAIL_ROWS * Z / TAIL_THREADS>(a, G, grp, Zs, ldzs);
    } else {
        constexpr int Z2 = Z / 2;
#pragma unroll
        for (int p = 0; p < TAIL_ROWS * Z2 / TAIL_THREADS; ++p) {
            const int it = tid + p * TAIL_THREADS;
            const int r = it / Z2, c = (it % Z2) * 2;
            *reinterpret_cast<f32x2*>(Zs + r * ldzs + c) = *reinterpret_cast<const f32x2*>(a.z + min(r0 + r, n - 1) * (int)a.ldz + c);
        }
    }
    // the input-gradient weight slabs (256 KB per workgroup) are requested BEHIND the z pick-up: loads return in order, so ahead
    // of it they stood between the polls and their data (3 us of this CU's 64 B/clk fill path); phases 2-6 cover them
    load_b7(0);
    if (DPW > 1 && CW == 1) load_b7(DPW - 1);
    lds_barrier();
    TAIL_STAMP(3);

    // ---------------- phase 2: y = l2_normalize(z)   (supervised_models.py:85); two rows per wave
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = wave * 2 + rr;
        float v[Z / 64];
        float ss = 0.f;
#pragma unroll
        for (int m = 0; m < Z / 64; ++m) {
            v[m] = Zs[row * ldzs + lane + 64 * m];
            ss += v[m] * v[m];
        }
        ss = tail_wave_sum(ss);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-12f));      // v_rsq_f32 (1 ulp)
#pragma unroll
        for (int m = 0; m < Z / 64; ++m) {
            const float y = v[m] * inv;
            Zs[row * ldzs + lane + 64 * m] = y;
            if (r0 + row < n && half == 0) {
                a.y[(r0 + row) * (int)a.ldy + lane + 64 * m] = y;
                if (!a.z_ready) a.z[(r0 + row) * (int)a.ldz + lane + 64 * m] = v[m];    // (the granules are kernel-internal)
            }
        }
        if (lane == 0) invs[row] = inv;
    }
    lds_barrier();
    TAIL_STAMP(4);

    // ---------------- phase 3: logits partials: slab g, K-slice ks per wave (fixed-order sum in phase 4)
    {
        const float* A = Zs + j * ldzs + ks3 * (4 * KL) + q;
        f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
        for (int u = 0; u < KL; ++u) {
            const float av = A[4 * u];
            acc0 = mfma16(av, bl[u].x, acc0);
            acc1 = mfma16(av, bl[u].y, acc1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x2*>(Ps + (ks3 * TAIL_ROWS + 4 * q + i) * GC + g3 * 32 + 2 * j) = f32x2{acc0[i], acc1[i]};
    }
    lds_barrier();
    TAIL_STAMP(5);

    const float inv_n = 1.0f / (float)n, inv_c = 1.0f / (float)C, inv_nc = inv_n * inv_c;
    // ---------------- phase 4: logits, loss rows, preds, dlogits   (supervised_models.py:111-126); two rows per wave,
    // one class of each 64-class group per lane (C <= 64 CW).  __expf / __logf: v_exp_f32 / v_log_f32 based, ~1e-6 relative.
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = wave * 2 + rr;
        const int i = r0 + row;
        const bool valid = i < n;
        bool in[CW];
        float x[CW], zl[CW], gl[CW], pr[CW];
#pragma unroll
        for (int m = 0; m < CW; ++m) {
            in[m] = lane + 64 * m < C;
            float xv = bias_l[m];
            for (int ks = 0; ks < nks; ++ks) xv += Ps[(ks * TAIL_ROWS + row) * GC + cl[m]];
            x[m] = in[m] ? xv : 0.f;
            zl[m] = (valid && in[m]) ? lab2[rr][m] : 0.f;
        }
        float loss;
        if (a.sigmoid) {
            float term = 0.f;
#pragma unroll
            for (int m = 0; m < CW; ++m) {
                const float e = __expf(-fabsf(x[m]));
                term += in[m] ? fmaxf(x[m], 0.f) - x[m] * zl[m] + __logf(1.0f + e) : 0.f;
                const float r1 = __builtin_amdgcn_rcpf(1.0f + e);           // v_rcp_f32 (1 ulp)
                pr[m] = x[m] >= 0.f ? r1 : e * r1;
                gl[m] = (pr[m] - zl[m]) * inv_nc;
            }
            loss = tail_wave_sum(term) * inv_c;
        } else {
            float mxl = -INFINITY;
#pragma unroll
            for (int m = 0; m < CW; ++m) mxl = fmaxf(mxl, in[m] ? x[m] : -INFINITY);
            const float mx = tail_wave_max(mxl);
            float ex[CW], sel = 0.f, zsl = 0.f, zxl = 0.f;
#pragma unroll
            for (int m = 0; m < CW; ++m) {
                ex[m] = in[m] ? __expf(x[m] - mx) : 0.f;
                sel += ex[m];
                zsl += zl[m];
                zxl += zl[m] * x[m];
            }
            const float se = tail_wave_sum(sel);
            const float zs = tail_wave_sum(zsl);
            const float zx = tail_wave_sum(zxl);
            const float rse = __builtin_amdgcn_rcpf(se);
#pragma unroll
            for (int m = 0; m < CW; ++m) {
                pr[m] = ex[m] * rse;
                gl[m] = (pr[m] * zs - zl[m]) * inv_n;
            }
            loss = zs * (mx + __logf(se)) - zx;
        }
#pragma unroll
        for (int m = 0; m < CW; ++m) {
            const int c = lane + 64 * m;
            const float gv = (valid && in[m]) ? gl[m] : 0.f;
            if (c < Cp32) DLs[row * ldc + c] = gv;
            if (valid && c < Cp4 && half == 0) {
                if (a.logits) a.logits[i * (int)a.ldlo + c] = in[m] ? x[m] : 0.f;
                if (a.preds) a.preds[i * (int)a.ldp + c] = in[m] ? pr[m] : 0.f;
                a.dlogits[i * (int)a.lddl + c] = gv;
            }
        }
        if (valid && lane == 0 && half == 0) a.loss_rows[i] = loss;
    }
    if (!a.train) {
        if (!a.z_ready) tail_epoch_done<NH>(a, G, grp, z_tag);
        if (grp == 0 && half == 0 && tid == 0) {
            if (a.c0) *a.c0 += a.d0;
            if (a.c1) *a.c1 += a.d1;
            if (a.c2) *a.c2 += a.d2;
        }
        return;
    }
    if (DPW > 1 && CW > 1) load_b7(DPW - 1);
    lds_barrier();
    TAIL_STAMP(6);

    // ---------------- phase 5: d_y = dlogits . W_head^T   -> DY (aliases the logits partials)
    float* DYs = Ps;
    if (wave < ZSLABS) {
        const int n0 = wave * 32;
        const float* A = DLs + j * ldc + 4 * q;
        f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
        for (int m = 0; m < 4 * CW; ++m) {
            if (16 * m < Cp32) {                                   // wave-uniform
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + 16 * m);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = mfma16(a4[e], bh5[m][0][e], acc0);
                    acc1 = mfma16(a4[e], bh5[m][1][e], acc1);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            DYs[(4 * q + i) * ldzs + n0 + j] = acc0[i];
            DYs[(4 * q + i) * ldzs + n0 + 16 + j] = acc1[i];
        }
    }
    lds_barrier();
    TAIL_STAMP(7);

    // ---------------- phase 6: d_z = l2_normalize'(d_y)   (two rows per wave)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = wave * 2 + rr;
        float dy[Z / 64], yv[Z / 64];
        float dot = 0.f;
#pragma unroll
        for (int m = 0; m < Z / 64; ++m) {
            dy[m] = DYs[row * ldzs + lane + 64 * m];
            yv[m] = Zs[row * ldzs + lane + 64 * m];
            dot += dy[m] * yv[m];
        }
        dot = tail_wave_sum(dot);
        const float inv = invs[row];
        const bool clamped = inv >= 1.0e6f;   // sum(z^2) < 1e-12: y = z * 1e6, no normalisation term
#pragma unroll
        for (int m = 0; m < Z / 64; ++m) {
            const float g = clamped ? dy[m] * inv : inv * (dy[m] - yv[m] * dot);
            DZs[row * ldzs + lane + 64 * m] = g;
            if (r0 + row < n && half == 0) a.dz[(r0 + row) * (int)a.lddz + lane + 64 * m] = g;
        }
    }
    lds_barrier();
    TAIL_STAMP(8);

    // ---------------- phase 7: [d_self | d_means] = [d_z[:, :O] . W_self^T | d_z[:, O:] . W_neigh^T]  -> DIN
    float* DIN = Hs;                                   // [16][2D + 8]   (Hs | Ms are dead since phase 1)
#pragma unroll
    for (int sl = 0; sl < DPW; ++sl) {
        const int col0 = slab_col0(sl);
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
    TAIL_STAMP(9);

    // ---------------- phase 8: d_h0 = relu'(h0) * (d_self on the self row, d_means / s on each of the s neighbor rows);
    // the relu masks are the bit flags kept from phase 0 (h0 is read once).
    {
        const int lddh0 = (int)a.lddh;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int it = tid + p * TAIL_THREADS;
            const int r = it / D4H, c = (half * D4H + it % D4H) * 4;
            const int i = r0 + r;
            if (i < n) {
                f32x4 g_self = *reinterpret_cast<const f32x4*>(DIN + r * ldi + c);
                f32x4 g_mean = *reinterpret_cast<const f32x4*>(DIN + r * ldi + D + c);
                if (a.gcn) {                                       // one operand, the mean over {neighbors} U {self}: both column
                    g_self = (g_self + g_mean) * (1.0f / (float)(s + 1));   // halves' input gradients add up, every row gets 1/(s+1)
                    g_mean = g_self;
                } else {
                    g_mean *= inv_s;
                }
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
    TAIL_STAMP(10);
    if (!a.z_ready) tail_epoch_done<NH>(a, G, grp, z_tag);
    if (grp == 0 && half == 0 && tid == 0) {                       // device counters (sampler clock / epoch cursor / optimizer step)
        if (a.c0) *a.c0 += a.d0;
        if (a.c1) *a.c1 += a.d1;
        if (a.c2) *a.c2 += a.d2;
    }
}

// Split form, first launch: only the z helpers (+ gather riders).  A lean kernel -- no main-workgroup code path, so it
// compiles to a fraction of the fused kernel's 246 VGPRs and 49 KB of LDS: two or three workgroups share a CU and the
// riders stream at the full HBM rate from the first microsecond, also on the CUs that run helpers.
template <int D, int O>
__global__ __launch_bounds__(TAIL_THREADS, 4) void sage_tail_z_kernel(const TailArgs a, const int tail_blocks, const CoGatherS J) {
    constexpr int HP = 2 * O / 64;
    const int G = tail_blocks;
    if ((int)blockIdx.x >= HP * G) {
        run_gather_item<8>(J, ((int64_t)blockIdx.x - HP * G) * TAIL_WAVES + (threadIdx.x >> 6), threadIdx.x & 63);
        return;
    }
    tail_z_helper<D, O>(a, (int)blockIdx.x / HP, (int)blockIdx.x % HP, G, false);
}

static size_t tail_z_lds_bytes(int D) { return ((size_t)TAIL_ROWS * (D + 4) + (size_t)TAIL_WAVES * TAIL_ROWS * 64) * sizeof(float); }

template <int D, int O>
static int launch_tail_z(const TailArgs& a, const CoGatherS& J, int64_t gather_waves, hipStream_t st) {
    const int tail_blocks = (int)gs_ceil_div(a.n, TAIL_ROWS);
    const int64_t blocks = (int64_t)tail_blocks * (2 * O / 64) + gs_ceil_div(gather_waves, TAIL_WAVES);
    GS_REQUIRE(blocks < (1ll << 31), "gs_sage_tail_z: grid too large");
    hipLaunchKernelGGL((sage_tail_z_kernel<D, O>), dim3((unsigned)blocks), dim3(TAIL_THREADS), tail_z_lds_bytes(D), st, a, tail_blocks, J);
    GS_LAUNCH_CHECK("sage_tail_z_kernel");
    return GS_OK;
}

// Split-off backward half (phases 7-8 of sage_tail_kernel as a launch of their own): given dLoss/dz of a LAST mean layer,
//   [d_self | d_means] = [dz[:, :O] . W_self^T | dz[:, O:] . W_neigh^T]                       (aggregators.py:51-58 backward)
//   d_h0 = relu'(h0) * (d_self on the self row, d_means / s on each of the s neighbor rows)    (aggregators.py:48, :64)
// for models that do not take the fused tail (unsupervised, > 128 classes, ...): one launch instead of a small GEMM and
// the input-gradient pull.  LEAN like the z helpers: 2D / 128 workgroups per 16-row group, one per 128-column slab of
// [d_self | d_means] (a slab lies in ONE term: 128 divides D), each of its 8 waves one 16-column MFMA tile with its 16
// weight rows (K = O: 8 float4) in registers; ~100 VGPRs and 17 KB of LDS, so several workgroups share a CU and gather
// jobs riding in the launch stream at the full rate.  (A first version gave a whole group to one workgroup, as the fused
// kernel does: 66 workgroups for 1044 rows each pulling both weight matrices through one CU -- 20 us.)
template <int D, int O>
__global__ __launch_bounds__(TAIL_THREADS, 4) void sage_tail_dh0_kernel(const TailArgs a, const int tail_blocks, const CoGatherS J) {
    constexpr int NWG = 2 * D / 128;
    const int G = tail_blocks;
    if ((int)blockIdx.x >= NWG * G) {
        run_gather_item<8>(J, ((int64_t)blockIdx.x - NWG * G) * TAIL_WAVES + (threadIdx.x >> 6), threadIdx.x & 63);
        return;
    }
    const int grp = (int)blockIdx.x / NWG, part = (int)blockIdx.x % NWG;
    const int colbase = part * 128;                    // in [0, 2D)
    const int term = colbase >= D ? 1 : 0;             // workgroup-uniform
    const int cb = colbase - term * D;                 // first of this slab's 128 columns inside the term's D columns
    constexpr int M7 = O / 16;
    constexpr int ldzs = O + 4, ldi = 128 + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* DZs = lds;                                  // [16][ldzs]   the term's half of dz
    float* DIN = DZs + TAIL_ROWS * ldzs;               // [16][ldi]    this slab of [d_self | d_means]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int r0 = grp * TAIL_ROWS;
    const int n = (int)a.n;
    const int s = a.s;
    const int ldh0 = (int)a.ldh;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // ---- everything this workgroup reads is an input: issue it all up front
    // (1) the term's half of dz: 16 rows x O columns
    const int zr = min(tid / (O / 4), TAIL_ROWS - 1), zc = (tid % (O / 4)) * 4;
    const f32x4 dzv = *reinterpret_cast<const f32x4*>(a.dz + (int64_t)min(r0 + zr, n - 1) * a.lddz + term * O + zc);
    // (2) this wave's 16 weight rows cb + 16 wave + j, all K = O columns (NT form, as phase 7 of the fused kernel)
    f32x4 b7[M7];
    {
        const int ldw = (int)(term ? a.ldwn : a.ldws);
        const float* B0 = (term ? a.Wn : a.Ws) + (cb + 16 * wave + j) * ldw + 4 * q;
#pragma unroll
        for (int m = 0; m < M7; ++m) b7[m] = *reinterpret_cast<const f32x4*>(B0 + 16 * m);
    }
    // (3) the h0 values behind this thread's output float4s (their signs are the relu mask): thread = (row, float4 column)
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
    if (tid < TAIL_ROWS * (O / 4)) *reinterpret_cast<f32x4*>(DZs + zr * ldzs + zc) = (r0 + zr < n) ? dzv : zero4;
    lds_barrier();
    // ---- the slab of [d_self | d_means]: one 16 x 16 tile per wave
    {
        const float* A = DZs + j * ldzs + 4 * q;
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

template <int D, int O>
static int launch_tail_dh0(const TailArgs& a, const CoGatherS& J, int64_t gather_waves, hipStream_t st) {
    const size_t lds = ((size_t)TAIL_ROWS * (O + 4) + (size_t)TAIL_ROWS * (128 + 4)) * sizeof(float);
    const int tail_blocks = (int)gs_ceil_div(a.n, TAIL_ROWS);
    const int64_t blocks = (int64_t)tail_blocks * (2 * D / 128) + gs_ceil_div(gather_waves, TAIL_WAVES);
    GS_REQUIRE(blocks < (1ll << 31), "gs_sage_tail_dh0: grid too large");
    hipLaunchKernelGGL((sage_tail_dh0_kernel<D, O>), dim3((unsigned)blocks), dim3(TAIL_THREADS), lds, st, a, tail_blocks, J);
    GS_LAUNCH_CHECK("sage_tail_dh0_kernel");
    return GS_OK;
}

static size_t tail_lds_bytes(int D, int O, int C) {
    const int Z = 2 * O, ldh = D + 4, ldzs = Z + 4;
    const int Cp32 = (C + 31) & ~31;
    const int ps = std::max(4 * TAIL_ROWS * 64, TAIL_ROWS * ldzs);      // logits partials [4][16][64] | DY [16][ldzs]
    const size_t floats = (size_t)2 * TAIL_ROWS * ldh + (size_t)2 * TAIL_ROWS * ldzs + ps + (size_t)TAIL_ROWS * (Cp32 + 4) + TAIL_ROWS;
    return floats * sizeof(float);
}

extern "C" int gs_sage_tail_supported(int32_t d_in, int32_t out_dim, int32_t C) {
    const bool ok = (d_in == 128 || d_in == 256) && (out_dim == 64 || out_dim == 128) && C >= 1 && C <= 128 &&
                    tail_lds_bytes(d_in, out_dim, C) <= 160 * 1024;
    return ok ? 1 : 0;
}

template <int D, int O, int CW, int NH>
static int launch_tail_nh(const TailArgs& a, const CoGatherS& J, int64_t gather_waves, hipStream_t st) {
    const size_t lds = tail_lds_bytes(a.D, a.O, a.C);
    GS_LDS_ATTR(160 * 1024, sage_tail_kernel<D, O, CW, NH>);
    const int tail_blocks = (int)gs_ceil_div(a.n, TAIL_ROWS);       // groups of 16 rows: (2 O / 64) z helpers + NH main workgroups each
    const int64_t blocks = (int64_t)tail_blocks * ((a.z_ready ? 0 : 2 * O / 64) + NH) + gs_ceil_div(gather_waves, TAIL_WAVES);
    GS_REQUIRE(blocks < (1ll << 31), "gs_sage_tail_fwd_bwd: grid too large");
    hipLaunchKernelGGL((sage_tail_kernel<D, O, CW, NH>), dim3((unsigned)blocks), dim3(TAIL_THREADS), lds, st, a, tail_blocks, J);
    GS_LAUNCH_CHECK("sage_tail_kernel");
    return GS_OK;
}

// Training launches over 256 input columns run two main workgroups per group (see sage_tail_kernel); forward-only launches
// (the second workgroup would have nothing of its own) and D = 128 (one input-gradient slab per wave already) run one.
template <int D, int O, int CW>
static int launch_tail_cw(const TailArgs& a, const CoGatherS& J, int64_t gather_waves, hipStream_t st) {
    if constexpr (D == 256) {
        if (a.train && g_tail_halves) return launch_tail_nh<D, O, CW, 2>(a, J, gather_waves, st);
    }
    return launch_tail_nh<D, O, CW, 1>(a, J, gather_waves, st);
}

template <int D, int O>
static int launch_tail(const TailArgs& a, const CoGatherS& J, int64_t gather_waves, hipStream_t st) {
    // C <= 64: one class per lane; 64 < C <= 128 (e.g. PPI's 121 sigmoid labels, example_supervised.sh): two
    return a.C > 64 ? launch_tail_cw<D, O, 2>(a, J, gather_waves, st) : launch_tail_cw<D, O, 1>(a, J, gather_waves, st);
}

extern "C" int gs_sage_tail_fwd_bwd(const gs_tail_desc* q, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    GS_REQUIRE(q, "gs_sage_tail_fwd_bwd: null descriptor");
    if (q->n == 0) return GS_OK;
    GS_REQUIRE(q->n > 0 && q->s > 0, "gs_sage_tail_fwd_bwd: bad sizes");
    if (q->s > TAIL_NB) {
        gs_set_error("gs_sage_tail_fwd_bwd: at most %d samples per node (got %d)", TAIL_NB, q->s);
        return GS_ENOTSUP;
    }
    GS_REQUIRE((q->n + q->n * (int64_t)q->s) * std::max(q->ldh, std::max(q->lddh, (int64_t)1)) < (1ll << 31),
               "gs_sage_tail_fwd_bwd: (n + n*s) * ld must be < 2^31 (32-bit row offsets)");
    if (!gs_sage_tail_supported(q->d_in, q->out_dim, q->C)) {
        gs_set_error("gs_sage_tail_fwd_bwd: unsupported shape d_in=%d out_dim=%d C=%d (d_in in {128,256}, out_dim in {64,128}, C <= 128)",
                     q->d_in, q->out_dim, q->C);
        return GS_ENOTSUP;
    }
    const int D = q->d_in, O = q->out_dim, Z = 2 * O, Cp4 = (q->C + 3) & ~3;
    GS_CHECK_MAT(q->h0, q->ldh, "gs_sage_tail_fwd_bwd h0");
    GS_CHECK_MAT(q->W_self, q->ldws, "gs_sage_tail_fwd_bwd W_self");
    GS_CHECK_MAT(q->W_neigh, q->ldwn, "gs_sage_tail_fwd_bwd W_neigh");
    GS_CHECK_MAT(q->W_head, q->ldwh, "gs_sage_tail_fwd_bwd W_head");
    GS_CHECK_MAT(q->means, q->ldm, "gs_sage_tail_fwd_bwd means");
    GS_CHECK_MAT(q->z, q->ldz, "gs_sage_tail_fwd_bwd z");
    GS_CHECK_MAT(q->y, q->ldy, "gs_sage_tail_fwd_bwd y");
    GS_CHECK_MAT(q->dlogits, q->lddl, "gs_sage_tail_fwd_bwd dlogits");
    GS_REQUIRE(q->labels && q->loss_rows && q->ldlab >= q->C, "gs_sage_tail_fwd_bwd: labels / loss_rows missing");
    GS_REQUIRE(q->ldh >= D && q->ldws >= O && q->ldwn >= O && q->ldwh >= Cp4 && q->ldm >= D && q->ldz >= Z && q->ldy >= Z &&
               q->lddl >= Cp4 && (!q->logits || q->ldlo >= Cp4) && (!q->preds || q->ldp >= Cp4),
               "gs_sage_tail_fwd_bwd: leading dimension too small");
    if (q->train) {
        GS_CHECK_MAT(q->dz, q->lddz, "gs_sage_tail_fwd_bwd dz");
        GS_CHECK_MAT(q->d_h0, q->lddh, "gs_sage_tail_fwd_bwd d_h0");
        GS_REQUIRE(q->lddz >= Z && q->lddh >= D, "gs_sage_tail_fwd_bwd: gradient leading dimension too small");
    }
    TailArgs a = {};
    a.h0 = q->h0; a.ldh = q->ldh; a.n = q->n; a.s = q->s; a.D = D;
    a.Ws = q->W_self; a.ldws = q->ldws; a.Wn = q->W_neigh; a.ldwn = q->ldwn; a.O = O;
    a.Wh = q->W_head; a.ldwh = q->ldwh; a.bh = q->b_head; a.labels = q->labels; a.ldlab = q->ldlab; a.C = q->C;
    a.sigmoid = q->sigmoid;
    a.means = q->means; a.ldm = q->ldm; a.z = q->z; a.ldz = q->ldz; a.y = q->y; a.ldy = q->ldy;
    a.logits = q->logits; a.ldlo = q->ldlo; a.preds = q->preds; a.ldp = q->ldp; a.dlogits = q->dlogits; a.lddl = q->lddl;
    a.loss_rows = q->loss_rows; a.dz = q->dz; a.lddz = q->lddz; a.d_h0 = q->d_h0; a.lddh = q->lddh;
    a.c0 = q->c0; a.d0 = q->d0; a.c1 = q->c1; a.d1 = q->d1; a.c2 = q->c2; a.d2 = q->d2;
    a.train = q->train ? 1 : 0;
    a.z_ready = q->z_ready ? 1 : 0;
    a.gcn = q->gcn ? 1 : 0;
    GS_REQUIRE(q->ids_copy_n >= 0 && (q->ids_copy_n == 0 || (q->ids_copy_src && q->ids_copy_dst)), "gs_sage_tail_fwd_bwd: ids_copy pointers missing");
    if (!q->z_ready) { a.ids_copy_src = q->ids_copy_src; a.ids_copy_dst = q->ids_copy_dst; a.ids_copy_n = q->ids_copy_n; }
    GS_REQUIRE(!q->gcn || (q->W_neigh == q->W_self + O && q->ldwn == q->ldws),
               "gs_sage_tail_fwd_bwd: gcn form takes ONE weight matrix (W_neigh == W_self + out_dim, same ld)");
    GS_REQUIRE(q->sync || q->z_ready, "gs_sage_tail_fwd_bwd: sync (2 G + 2 + 64 G out_dim zero-initialised uint32 words, G = ceil(n / 16), "
                                      "private to the caller's stream) missing");
    GS_REQUIRE(!q->sync || (reinterpret_cast<uintptr_t>(q->sync) & 7u) == 0, "gs_sage_tail_fwd_bwd: sync must be 8-byte aligned");
    a.sync = q->sync;
    hipStream_t st = (hipStream_t)stream;
    CoGatherS J = {};
    int64_t gw = 0;
    {
        int rc = build_cojobs_s(jobs_host, n_jobs, &J, &gw);
        if (rc != GS_OK) return rc;
    }
    if (D == 256 && O == 128) return launch_tail<256, 128>(a, J, gw, st);
    if (D == 256 && O == 64) return launch_tail<256, 64>(a, J, gw, st);
    if (D == 128 && O == 128) return launch_tail<128, 128>(a, J, gw, st);
    return launch_tail<128, 64>(a, J, gw, st);
}

// Split form, first launch: z = [h_self . W_self | mean(h_neigh) . W_neigh] and the neighbor means of the descriptor (the
// head / label / gradient fields are not touched), + gather riders.  Follow with gs_sage_tail_fwd_bwd on the same
// descriptor with z_ready = 1.
extern "C" int gs_sage_tail_z(const gs_tail_desc* q, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    GS_REQUIRE(q, "gs_sage_tail_z: null descriptor");
    if (q->n == 0) return GS_OK;
    GS_REQUIRE(q->n > 0 && q->s > 0, "gs_sage_tail_z: bad sizes");
    if (q->s > TAIL_NB) {
        gs_set_error("gs_sage_tail_z: at most %d samples per node (got %d)", TAIL_NB, q->s);
        return GS_ENOTSUP;
    }
    GS_REQUIRE((q->n + q->n * (int64_t)q->s) * std::max(q->ldh, (int64_t)1) < (1ll << 31),
               "gs_sage_tail_z: (n + n*s) * ld must be < 2^31 (32-bit row offsets)");
    if (!gs_sage_tail_supported(q->d_in, q->out_dim, q->C > 0 ? q->C : 1)) {
        gs_set_error("gs_sage_tail_z: unsupported shape d_in=%d out_dim=%d", q->d_in, q->out_dim);
        return GS_ENOTSUP;
    }
    const int D = q->d_in, O = q->out_dim, Z = 2 * O;
    GS_CHECK_MAT(q->h0, q->ldh, "gs_sage_tail_z h0");
    GS_CHECK_MAT(q->W_self, q->ldws, "gs_sage_tail_z W_self");
    GS_CHECK_MAT(q->W_neigh, q->ldwn, "gs_sage_tail_z W_neigh");
    GS_CHECK_MAT(q->means, q->ldm, "gs_sage_tail_z means");
    GS_CHECK_MAT(q->z, q->ldz, "gs_sage_tail_z z");
    GS_REQUIRE(q->ldh >= D && q->ldws >= O && q->ldwn >= O && q->ldm >= D && q->ldz >= Z, "gs_sage_tail_z: leading dimension too small");
    TailArgs a = {};
    a.h0 = q->h0; a.ldh = q->ldh; a.n = q->n; a.s = q->s; a.D = D;
    a.Ws = q->W_self; a.ldws = q->ldws; a.Wn = q->W_neigh; a.ldwn = q->ldwn; a.O = O;
    a.means = q->means; a.ldm = q->ldm; a.z = q->z; a.ldz = q->ldz;
    a.z_ready = 1;
    a.gcn = q->gcn ? 1 : 0;
    GS_REQUIRE(q->ids_copy_n >= 0 && (q->ids_copy_n == 0 || (q->ids_copy_src && q->ids_copy_dst)), "gs_sage_tail_z: ids_copy pointers missing");
    a.ids_copy_src = q->ids_copy_src; a.ids_copy_dst = q->ids_copy_dst; a.ids_copy_n = q->ids_copy_n;
    hipStream_t st = (hipStream_t)stream;
    CoGatherS J = {};
    int64_t gw = 0;
    {
        int rc = build_cojobs_s(jobs_host, n_jobs, &J, &gw);
        if (rc != GS_OK) return rc;
    }
    if (D == 256 && O == 128) return launch_tail_z<256, 128>(a, J, gw, st);
    if (D == 256 && O == 64) return launch_tail_z<256, 64>(a, J, gw, st);
    if (D == 128 && O == 128) return launch_tail_z<128, 128>(a, J, gw, st);
    return launch_tail_z<128, 64>(a, J, gw, st);
}

// The input gradients of a LAST mean layer from dLoss/dz (see sage_tail_dh0_kernel): desc fields used: h0, W_self, W_neigh,
// dz (in), d_h0 (out), n, s, d_in, out_dim.
extern "C" int gs_sage_tail_dh0(const gs_tail_desc* q, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    GS_REQUIRE(q, "gs_sage_tail_dh0: null descriptor");
    if (q->n == 0) return GS_OK;
    GS_REQUIRE(q->n > 0 && q->s > 0, "gs_sage_tail_dh0: bad sizes");
    if (q->s > TAIL_NB) {
        gs_set_error("gs_sage_tail_dh0: at most %d samples per node (got %d)", TAIL_NB, q->s);
        return GS_ENOTSUP;
    }
    GS_REQUIRE((q->n + q->n * (int64_t)q->s) * std::max(q->ldh, std::max(q->lddh, (int64_t)1)) < (1ll << 31),
               "gs_sage_tail_dh0: (n + n*s) * ld must be < 2^31 (32-bit row offsets)");
    if (!gs_sage_tail_supported(q->d_in, q->out_dim, 1)) {
        gs_set_error("gs_sage_tail_dh0: unsupported shape d_in=%d out_dim=%d", q->d_in, q->out_dim);
        return GS_ENOTSUP;
    }
    const int D = q->d_in, O = q->out_dim, Z = 2 * O;
    GS_CHECK_MAT(q->h0, q->ldh, "gs_sage_tail_dh0 h0");
    GS_CHECK_MAT(q->W_self, q->ldws, "gs_sage_tail_dh0 W_self");
    GS_CHECK_MAT(q->W_neigh, q->ldwn, "gs_sage_tail_dh0 W_neigh");
    GS_CHECK_MAT(q->dz, q->lddz, "gs_sage_tail_dh0 dz");
    GS_CHECK_MAT(q->d_h0, q->lddh, "gs_sage_tail_dh0 d_h0");
    GS_REQUIRE(q->ldh >= D && q->ldws >= O && q->ldwn >= O && q->lddz >= Z && q->lddh >= D, "gs_sage_tail_dh0: leading dimension too small");
    if (q->gcn) {
        gs_set_error("gs_sage_tail_dh0: the gcn form is only built into gs_sage_tail_fwd_bwd / gs_sage_tail_z");
        return GS_ENOTSUP;
    }
    TailArgs a = {};
    a.h0 = q->h0; a.ldh = q->ldh; a.n = q->n; a.s = q->s; a.D = D;
    a.Ws = q->W_self; a.ldws = q->ldws; a.Wn = q->W_neigh; a.ldwn = q->ldwn; a.O = O;
    a.dz = q->dz; a.lddz = q->lddz; a.d_h0 = q->d_h0; a.lddh = q->lddh;
    a.z_ready = 1;
    hipStream_t st = (hipStream_t)stream;
    CoGatherS J = {};
    int64_t gw = 0;
    {
        int rc = build_cojobs_s(jobs_host, n_jobs, &J, &gw);
        if (rc != GS_OK) return rc;
    }
    if (D == 256 && O == 128) return launch_tail_dh0<256, 128>(a, J, gw, st);
    if (D == 256 && O == 64) return launch_tail_dh0<256, 64>(a, J, gw, st);
    if (D == 128 && O == 128) return launch_tail_dh0<128, 128>(a, J, gw, st);
    return launch_tail_dh0<128, 64>(a, J, gw, st);
}
