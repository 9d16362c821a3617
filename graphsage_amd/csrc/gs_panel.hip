// "Panel" form of the layer-0 contraction of a mean / GCN step (aggregators.py:51-58, :110):
//     out[:, t N : (t+1) N] = act(A_t . W_t + bias),   t = self | neighbor-mean term (concat), or one term (GCN)
// built for what round 4 measured on the "stream" form (gs_stream.hip): that launch is bound by the L2 -> CU operand
// traffic of its re-reads (218 MB per launch: every 32-row tile walks a whole weight panel; no-loads variant 14.3 us
// against 24.7), not by MFMA issue.  Here ONE workgroup per CU owns a 48-row x 128-column output panel for the WHOLE K:
//   * M = 5632 rows / 48 = 118 row tiles x 2 terms = 236 workgroups for 256 CUs: one round, every CU busy, each weight
//     panel (602 x 128 x 4 B = 308 KB) pulled ONCE per workgroup -- 236 x 308 KB + the A rows = 100 MB per launch
//     instead of 218;
//   * its 8 waves split K (two per SIMD); a wave's tile is the full 48 x 128 panel: 3 x 8 accumulator tiles of
//     v_mfma_f32_16x16x4_f32 (exact fp32).  Per macro step of 16 k a lane issues 3 A loads (16 B: its row of each 16-row
//     block, 4 consecutive k) and 8 B loads (16 B: 4 adjacent columns of one k row -- element c feeds column tile
//     4 h + c, whose columns are {64 h + 4 j + c}) for 96 MFMAs: 0.057 B per lane and MFMA cycle against 0.094, and 11
//     load instructions per 3072 matrix-pipe cycles against 5 per 512 -- every instruction between two MFMAs costs
//     matrix-pipe time on this chip (benchmarks/probes/mfma_issue.hip);
//   * a ring of P = 3 macro steps covers 3 x 3072 cycles = 4.4 us of memory latency per wave (the stream form: 1 us);
//   * the 8 partial panels are summed in wave order through LDS in two passes of 64 columns (96 KB), then bias +
//     activation + 16-byte stores: deterministic, no atomics.
// A rows may be gathered (layer 0: the self rows X[ids]); K tails (K % 16) are one extra masked macro step of wave 0.
// Gather jobs of the next step ride as extra workgroups behind the contraction workgroups (one 8-wave workgroup per CU:
// the riders stream at the full HBM rate on the CUs the contraction leaves free, and on all of them once it is done).
#include "gs_common.h"
#include "gs_gather_dev.h"
#include <stdlib.h>

#define PANEL_ROWS 48
#define PANEL_COLS 128
#define PANEL_WAVES 8
#define PANEL_THREADS 512

struct PanelTerm {
    const float* A;        // [*, lda]; row i of the term is A[a_idx ? a_idx[i] : i]
    const int32_t* a_idx;  // nullable row gather (layer 0: the self rows of the feature table)
    const float* W;        // [K, ldw]
    int64_t lda;
    int32_t ldw;
};
struct PanelArgs {
    PanelTerm t[2];
    int32_t nterms;        // 1, or 2 (concat: term i writes columns [i N, (i + 1) N))
    int32_t M, N, K;
    float* C;
    int64_t ldc;
    const float* bias;     // indexed by output column (incl. the concat offset), nullable
    int32_t act;
    int32_t tiles_m;       // 48-row tiles
    int32_t panels_n;      // 128-column panels per term
    int32_t blocks_per_term;   // ceil(tiles_m / 8) * 8 * panels_n: the panels of one row tile run on ONE XCD (block b -> XCD b % 8)
    int32_t n_blocks;      // contraction workgroups: blocks_per_term * nterms
};

__device__ __forceinline__ f32x4 panel_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int P>
__global__ __launch_bounds__(PANEL_THREADS) void sage_panel_fwd_kernel(const PanelArgs g, const CoGatherS J) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [8 waves][48 rows][64 columns]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // uniform: per-wave fields live in SGPRs
    if ((int)blockIdx.x >= g.n_blocks) {
        run_gather_item<13, 25>(J, ((int64_t)blockIdx.x - g.n_blocks) * PANEL_WAVES + wave, lane);
        return;
    }
    const int term = (int)blockIdx.x / g.blocks_per_term;
    const int r_ = (int)blockIdx.x - term * g.blocks_per_term;
    const int per_grp = 8 * g.panels_n;
    const int grp = r_ / per_grp, rr = r_ - grp * per_grp;
    const int panel = rr >> 3, tile = grp * 8 + (rr & 7);
    if (tile >= g.tiles_m) return;                                   // workgroup-uniform (padding of the last group of 8 tiles)
    const int j = lane & 15, q = lane >> 4;
    const PanelTerm& T = g.t[term];
    const int K = g.K, N = g.N, M = g.M;
    const int m0 = tile * PANEL_ROWS, n0 = panel * PANEL_COLS;
    const int nfull = K >> 4;                                        // macro steps whose 16 k are all < K
    const int mb = (nfull * wave) >> 3, me = (nfull * (wave + 1)) >> 3;   // this wave's macro steps
    // ---- A: the lane's row of each 16-row block (gathered through a_idx at layer 0), 4 consecutive k per load
    const float* ap[3];
    const float* arow0[3];
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) {
        const int arow = min(m0 + 16 * rb + j, M - 1);               // rows past M: a valid row, never stored
        const int64_t srow = T.a_idx ? (int64_t)T.a_idx[arow] : (int64_t)arow;
        arow0[rb] = T.A + srow * T.lda;
        ap[rb] = arow0[rb] + 4 * q + 16 * mb;
    }
    // ---- B: four uniform bases (k rows 16 m + 4 q + e, e = 0..3) and one 32-bit byte offset per 64-column half
    const int N4 = (N + 3) & ~3;
    const uint32_t ldw4 = (uint32_t)T.ldw * 4u;
    const char* __restrict__ W0 = (const char*)T.W;
    const char* __restrict__ W1 = W0 + ldw4;
    const char* __restrict__ W2 = W0 + 2 * ldw4;
    const char* __restrict__ W3 = W0 + 3 * ldw4;
    uint32_t wo[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)                                      // columns past N: clamped to a valid quad, never stored
        wo[h] = (uint32_t)(16 * mb + 4 * q) * ldw4 + (uint32_t)min(n0 + 64 * h + 4 * j, N4 - 4) * 4u;
    const uint32_t wstride = 16u * ldw4;

    f32x4 acc[3][8];
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[rb][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[P][3];
    f32x4 b[P][4][2];
    // `left`: pointer advances still allowed -- a refill past the wave's last macro step re-reads the last one (always a
    // valid address; the value is never consumed), so the pipeline needs no one-at-a-time remainder.
    int left = me - mb - 1;
    auto load_stage = [&](const int st) {
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) a[st][rb] = *reinterpret_cast<const f32x4*>(ap[rb]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            b[st][0][h] = *reinterpret_cast<const f32x4*>(W0 + wo[h]);
            b[st][1][h] = *reinterpret_cast<const f32x4*>(W1 + wo[h]);
            b[st][2][h] = *reinterpret_cast<const f32x4*>(W2 + wo[h]);
            b[st][3][h] = *reinterpret_cast<const f32x4*>(W3 + wo[h]);
        }
        const int adv = left > 0 ? 1 : 0;                            // wave-uniform
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) ap[rb] += 16 * adv;
        wo[0] += wstride * (uint32_t)adv;
        wo[1] += wstride * (uint32_t)adv;
        --left;
    };
    auto compute_stage = [&](const int st) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[rb][4 * h + c] = panel_mfma(a[st][rb][e], b[st][e][h][c], acc[rb][4 * h + c]);
    };
    const int cnt = me - mb;
    if (cnt > 0) {
#pragma unroll
        for (int st = 0; st < P; ++st) load_stage(st);
        int m = 0;
#pragma unroll 1
        for (; m + P < cnt; m += P) {
#pragma unroll
            for (int st = 0; st < P; ++st) {
                compute_stage(st);
                __builtin_amdgcn_sched_barrier(0);                   // the refill stays below the MFMAs that free its registers
                load_stage(st);
            }
        }
#pragma unroll
        for (int st = 0; st < P; ++st)
            if (m + st < cnt) compute_stage(st);
    }
    if (wave == 0 && (K & 15) != 0) {
        // tail macro step of wave 0 (it holds the fewest full steps): k = 16 nfull + 4 q + e; A elements with k >= K are
        // zeroed, B rows are clamped to K - 1 (finite values times an exact zero)
        const int kq = 16 * nfull + 4 * q;
        const int K4 = (K + 3) & ~3;
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            f32x4 av = *reinterpret_cast<const f32x4*>(arow0[rb] + min(kq, K4 - 4));   // the row's pad columns [K, K4) are readable
            if (kq + 0 >= K) av.x = 0.f;
            if (kq + 1 >= K) av.y = 0.f;
            if (kq + 2 >= K) av.z = 0.f;
            if (kq + 3 >= K) av.w = 0.f;
            a[0][rb] = av;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* wr = T.W + (int64_t)min(kq + e, K - 1) * T.ldw;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                b[0][e][h] = *reinterpret_cast<const f32x4*>(wr + min(n0 + 64 * h + 4 * j, N4 - 4));
        }
        compute_stage(0);
    }
    // ---- split-K sum in wave order + bias + activation + store, 64 columns per pass
    const int col_off = term * N;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();                                      // the previous pass's partials have been read
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i)                              // D layout of the 16x16 MFMA: row 4 q + i, column j
                *reinterpret_cast<f32x4*>(red + ((wave * PANEL_ROWS + 16 * rb + 4 * q + i) * 64 + 4 * j)) =
                    f32x4{acc[rb][4 * h + 0][i], acc[rb][4 * h + 1][i], acc[rb][4 * h + 2][i], acc[rb][4 * h + 3][i]};
        __syncthreads();
#pragma unroll
        for (int it0 = 0; it0 < PANEL_ROWS * 16; it0 += PANEL_THREADS) {
            const int it = it0 + tid;
            if (it < PANEL_ROWS * 16) {
                const int row = it >> 4, j4 = it & 15;
                f32x4 v = *reinterpret_cast<const f32x4*>(red + (row * 64 + 4 * j4));
#pragma unroll
                for (int w = 1; w < PANEL_WAVES; ++w) v += *reinterpret_cast<const f32x4*>(red + ((w * PANEL_ROWS + row) * 64 + 4 * j4));
                const int col = n0 + 64 * h + 4 * j4, grow = m0 + row;
                if (grow < M && col < N) {                           // N % 4 == 0: a quad is all in or all out
                    if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col_off + col);
                    if (g.act == GS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<f32x4*>(g.C + (int64_t)grow * g.ldc + col_off + col) = v;
                }
            }
        }
    }
}

static inline int rup4p(int x) { return (x + 3) & ~3; }

extern "C" int gs_sage_dense_fwd_panel(const float* self, int64_t ld_self, const int32_t* self_idx, const float* agg,
                                       int64_t ld_agg, int32_t d, int64_t n, const float* W_self, int64_t ldw_self,
                                       const float* W_neigh, int64_t ldw_neigh, int32_t out_dim, int act, const float* bias,
                                       float* out, int64_t ldo, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    GS_REQUIRE(n > 0 && agg && W_neigh && out && d > 0 && out_dim > 0, "gs_sage_dense_fwd_panel: bad args");
    GS_REQUIRE(out_dim % 4 == 0 && ldo % 4 == 0, "gs_sage_dense_fwd_panel: out_dim and ldo must be multiples of 4 (16-byte column quads)");
    GS_CHECK_MAT(agg, ld_agg, "gs_sage_dense_fwd_panel agg");
    GS_CHECK_MAT(W_neigh, ldw_neigh, "gs_sage_dense_fwd_panel W_neigh");
    GS_CHECK_MAT(out, ldo, "gs_sage_dense_fwd_panel out");
    GS_REQUIRE(ld_agg >= rup4p(d) && ldw_neigh >= out_dim, "gs_sage_dense_fwd_panel: ld too small");
    GS_REQUIRE(!bias || gs_aligned16(bias), "gs_sage_dense_fwd_panel: bias must be 16-byte aligned");
    PanelArgs g = {};
    if (self) {
        GS_CHECK_MAT(self, ld_self, "gs_sage_dense_fwd_panel self");
        GS_CHECK_MAT(W_self, ldw_self, "gs_sage_dense_fwd_panel W_self");
        GS_REQUIRE(ld_self >= rup4p(d) && ldw_self >= out_dim, "gs_sage_dense_fwd_panel: self ld too small");
        g.t[0] = PanelTerm{self, self_idx, W_self, ld_self, (int32_t)ldw_self};
        g.t[1] = PanelTerm{agg, nullptr, W_neigh, ld_agg, (int32_t)ldw_neigh};
        g.nterms = 2;
    } else {
        g.t[0] = PanelTerm{agg, nullptr, W_neigh, ld_agg, (int32_t)ldw_neigh};
        g.nterms = 1;
    }
    GS_REQUIRE(ldo >= (int64_t)out_dim * g.nterms, "gs_sage_dense_fwd_panel: ldo too small");
    // the weights are addressed with 32-bit BYTE offsets against their base pointers
    GS_REQUIRE(((int64_t)d + 32) * std::max(ldw_self, ldw_neigh) * 4 < (1ll << 32), "gs_sage_dense_fwd_panel: 32-bit weight offsets exceeded");
    GS_REQUIRE(n < (1ll << 31) - 64, "gs_sage_dense_fwd_panel: too many rows");
    g.M = (int32_t)n; g.N = out_dim; g.K = d; g.C = out; g.ldc = ldo; g.bias = bias; g.act = act;
    g.tiles_m = (int)gs_ceil_div(n, PANEL_ROWS);
    g.panels_n = (int)gs_ceil_div(out_dim, PANEL_COLS);
    g.blocks_per_term = (int)gs_ceil_div(g.tiles_m, 8) * 8 * g.panels_n;
    g.n_blocks = g.blocks_per_term * g.nterms;
    CoGatherS J = {};
    int64_t waves = 0;
    int rc = build_cojobs_s(jobs_host, n_jobs, &J, &waves);
    if (rc != GS_OK) return rc;
    const int64_t blocks = g.n_blocks + gs_ceil_div(waves, PANEL_WAVES);
    GS_REQUIRE(blocks < (1ll << 31), "gs_sage_dense_fwd_panel: grid too large");
    const size_t lds = (size_t)PANEL_WAVES * PANEL_ROWS * 64 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        GS_HIP(hipFuncSetAttribute((const void*)sage_panel_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(sage_panel_fwd_kernel<3>, dim3((unsigned)blocks), dim3(PANEL_THREADS), lds, (hipStream_t)stream, g, J);
    GS_LAUNCH_CHECK("sage_panel_fwd_kernel");
    return GS_OK;
}
