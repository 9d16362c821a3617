// K3 / K4 / K6: fp32 dense contractions on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32:
// a k-ordered fmaf chain, so results stay inside the 1e-4 parity budget without any bf16 split).
//
// One kernel template covers every contraction on the hot path:
//   NN  out = self·W            forward            A k-contiguous,  B n-contiguous
//   TN  dW  = A^T·dZ            weight gradient    A m-contiguous (row-gathered over k),  B n-contiguous
//   NT  dX  = dZ·W^T            input gradient     A k-contiguous,  B k-contiguous
// Both operands are staged through LDS as DIRECT float4 copies of the global tile (no transposes):
//   k-contiguous source  -> LDS [row][32+4]  read back with one ds_read_b128 per 4 k-steps
//   row-contiguous source-> LDS [32][rows]   read back with ds_read_b32 (lanes = consecutive rows)
// The MFMA consumes k in the order (k0 + 4*(lane>>5) + i): A and B use the same permutation of the
// 8-wide k group, which only re-orders the fp32 summation.
// Block = 256 threads = 4 waves (2x2), block tile BM x BN (64x64 or 128x128), BK = 32, register
// prefetch of the next tile overlaps the MFMAs.  blockIdx -> tile mapping is XCD-aware: the tiles that
// share an A row panel run on the same XCD (same L2).  Up to two (A, B, K) terms per launch give the
// SAGE "[self·W_self || mean·W_neigh]" (concat) or "self·W_self + mean·W_neigh" (add) in ONE kernel,
// with bias + relu fused in the epilogue.  Split-K writes per-slice slabs (deterministic, no atomics).
#include "gs_common.h"
#include "gs_gather_dev.h"
#include <stdlib.h>

struct GemmTerm {
    const float* A;
    const int32_t* a_idx;  // gathers source rows of A (nullable)
    const float* B;
    int64_t lda, ldb;
    int32_t K;
};

struct GemmArgs {
    GemmTerm t[2];
    int32_t nterms;  // 1 or 2
    int32_t concat;  // nterms == 2: 1 -> term i writes columns [i*N, (i+1)*N), 0 -> terms are summed
    int64_t M;
    int32_t N;  // columns per term
    float* C;
    int64_t ldc;
    int64_t slab_stride;  // elements between split-K slabs
    int32_t kchunk;       // K range per blockIdx.z slice (multiple of 32); 0 -> whole K
    const float* bias;    // indexed by output column (global column incl. concat offset)
    int32_t act;
    int32_t accumulate;  // C += result (before act; only with act == identity)
    int32_t tiles_m, tiles_n;  // tiles_n is per term
    // max-pool epilogue (pool_s > 0): the rows are groups of pool_s consecutive rows; a tile holds floor(BM / pool_s)
    // WHOLE groups (its row origin advances by pool_rows = that many rows), and instead of C the epilogue writes
    // pool_out[group, col] = max_j act(row j of the group) and pool_arg = the first j that attains it.
    int32_t pool_s, pool_rows;
    float* pool_out;
    int64_t pool_ld;
    int32_t* pool_arg;
    int64_t pool_lda;
    const int32_t* m_dev;  // nullable: the row count lives on the device (rows >= *m_dev do not exist); M is its upper bound
};

#define GS_IDXCAP 1024  // gather indices cached in LDS per k-chunk (row-gathered TN operand)

template <int BM, int BN, bool A_KC>
struct GemmSmem {
    static constexpr int KP = 36;
    static constexpr int AS = A_KC ? BM * KP : 32 * BM;
};

// XCD-aware tile id: consecutive tile ids (same A row panel) stay on one XCD (block b runs on XCD b % 8)
__device__ __forceinline__ int gs_xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

template <int BM, int BN, bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int wgid, const int zslice, float* smem, const int64_t gM) {
    constexpr int BK = 32;
    constexpr int KP = BK + 4;  // padded k stride for k-contiguous tiles (conflict-free ds_read_b128)
    constexpr int PA = BM / 32;  // float4 loads per thread per stage
    constexpr int PB = BN / 32;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AS_FLOATS = A_KC ? BM * KP : BK * BM;
    constexpr int BS_FLOATS = B_KC ? BN * KP : BK * BN;
    constexpr int STAGE_FLOATS = AS_FLOATS + BS_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tiles_n_total = g.tiles_n * ((g.nterms == 2 && g.concat) ? 2 : 1);
    const int tile_m = wgid / tiles_n_total;
    int tile_n = wgid - tile_m * tiles_n_total;
    int term0 = 0, term1 = g.nterms;
    int col_off = 0;
    if (g.nterms == 2 && g.concat) {
        term0 = tile_n / g.tiles_n;
        term1 = term0 + 1;
        tile_n -= term0 * g.tiles_n;
        col_off = term0 * g.N;
    }
    const int64_t m0 = (int64_t)tile_m * (g.pool_s > 0 ? g.pool_rows : BM);
    const int n0 = tile_n * BN;
    if (m0 >= gM) return;                       // (only with a device-side row count: the grid covers its upper bound)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int term = term0; term < term1; ++term) {
        const GemmTerm T = g.t[term];
        int k_begin = 0, k_end = T.K;
        if (g.kchunk > 0) {
            k_begin = min((int64_t)zslice * g.kchunk, (int64_t)T.K);
            k_end = min(k_begin + g.kchunk, T.K);
        }
        if (k_begin >= k_end) continue;
        int32_t* idx_lds = reinterpret_cast<int32_t*>(smem + 2 * STAGE_FLOATS);
        int idx_base = k_begin;

        // ---- per-thread source pointers that do not depend on k (k-contiguous operands)
        const float* a_row[PA];
        bool a_ok[PA];
        if constexpr (A_KC) {
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int64_t r = m0 + p * 32 + (tid >> 3);
                a_ok[p] = r < gM;
                const int64_t src = a_ok[p] ? (T.a_idx ? (int64_t)T.a_idx[r] : r) : 0;
                a_row[p] = T.A + src * T.lda + (tid & 7) * 4;
            }
        }
        const float* b_row[PB];
        bool b_ok[PB];
        if constexpr (B_KC) {
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int r = n0 + p * 32 + (tid >> 3);
                b_ok[p] = r < g.N;
                b_row[p] = T.B + (int64_t)(b_ok[p] ? r : 0) * T.ldb + (tid & 7) * 4;
            }
        }

        f32x4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
        auto load_tiles = [&](int k0, f32x4 (&ra)[PA], f32x4 (&rb)[PB]) {
            // ---------------- A
            if constexpr (A_KC) {
                const int k = k0 + (tid & 7) * 4;
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (a_ok[p] && k < k_end) {
                        v = *reinterpret_cast<const f32x4*>(a_row[p] + k0);
                        if (k + 3 >= k_end) {
                            if (k + 1 >= k_end) v.y = 0.f;
                            if (k + 2 >= k_end) v.z = 0.f;
                            if (k + 3 >= k_end) v.w = 0.f;
                        }
                    }
                    ra[p] = v;
                }
            } else {
                constexpr int UPR = BM / 4;  // float4 units per k-row
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const int u = p * 256 + tid;
                    const int kk = u / UPR;
                    const int64_t r = m0 + (u - kk * UPR) * 4;
                    const int k = k0 + kk;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (k < k_end && r < gM) {
                        const int64_t src = T.a_idx ? (int64_t)idx_lds[k - idx_base] : (int64_t)k;
                        v = *reinterpret_cast<const f32x4*>(T.A + src * T.lda + r);
                        if (r + 3 >= gM) {
                            if (r + 1 >= gM) v.y = 0.f;
                            if (r + 2 >= gM) v.z = 0.f;
                            if (r + 3 >= gM) v.w = 0.f;
                        }
                    }
                    ra[p] = v;
                }
            }
            // ---------------- B
            if constexpr (B_KC) {
                const int k = k0 + (tid & 7) * 4;
#pragma unroll
                for (int p = 0; p < PB; ++p) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (b_ok[p] && k < k_end) {
                        v = *reinterpret_cast<const f32x4*>(b_row[p] + k0);
                        if (k + 3 >= k_end) {
                            if (k + 1 >= k_end) v.y = 0.f;
                            if (k + 2 >= k_end) v.z = 0.f;
                            if (k + 3 >= k_end) v.w = 0.f;
                        }
                    }
                    rb[p] = v;
                }
            } else {
                constexpr int UPR = BN / 4;
#pragma unroll
                for (int p = 0; p < PB; ++p) {
                    const int u = p * 256 + tid;
                    const int kk = u / UPR;
                    const int r = n0 + (u - kk * UPR) * 4;
                    const int k = k0 + kk;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (k < k_end && r < g.N) {
                        v = *reinterpret_cast<const f32x4*>(T.B + (int64_t)k * T.ldb + r);
                        if (r + 3 >= g.N) {
                            if (r + 1 >= g.N) v.y = 0.f;
                            if (r + 2 >= g.N) v.z = 0.f;
                            if (r + 3 >= g.N) v.w = 0.f;
                        }
                    }
                    rb[p] = v;
                }
            }
        };
        auto store_tiles = [&](const f32x4 (&ra)[PA], const f32x4 (&rb)[PB], float* As, float* Bs) {
            if constexpr (A_KC) {
#pragma unroll
                for (int p = 0; p < PA; ++p)
                    *reinterpret_cast<f32x4*>(&As[(p * 32 + (tid >> 3)) * KP + (tid & 7) * 4]) = ra[p];
            } else {
                constexpr int UPR = BM / 4;
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const int u = p * 256 + tid;
                    const int kk = u / UPR;
                    *reinterpret_cast<f32x4*>(&As[kk * BM + (u - kk * UPR) * 4]) = ra[p];
                }
            }
            if constexpr (B_KC) {
#pragma unroll
                for (int p = 0; p < PB; ++p)
                    *reinterpret_cast<f32x4*>(&Bs[(p * 32 + (tid >> 3)) * KP + (tid & 7) * 4]) = rb[p];
            } else {
                constexpr int UPR = BN / 4;
#pragma unroll
                for (int p = 0; p < PB; ++p) {
                    const int u = p * 256 + tid;
                    const int kk = u / UPR;
                    *reinterpret_cast<f32x4*>(&Bs[kk * BN + (u - kk * UPR) * 4]) = rb[p];
                }
            }
        };

        auto compute = [&](const float* As, const float* Bs) {
#pragma unroll
            for (int gk = 0; gk < BK / 8; ++gk) {
                float a[TM][4], b[TN][4];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int row = wm * (BM / 2) + tm * 32 + l31;
                    if constexpr (A_KC) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&As[row * KP + gk * 8 + lh * 4]);
                        a[tm][0] = v.x; a[tm][1] = v.y; a[tm][2] = v.z; a[tm][3] = v.w;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[tm][i] = As[(gk * 8 + lh * 4 + i) * BM + row];
                    }
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int col = wn * (BN / 2) + tn * 32 + l31;
                    if constexpr (B_KC) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[col * KP + gk * 8 + lh * 4]);
                        b[tn][0] = v.x; b[tn][1] = v.y; b[tn][2] = v.z; b[tn][3] = v.w;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) b[tn][i] = Bs[(gk * 8 + lh * 4 + i) * BN + col];
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][i], b[tn][i], acc[tm][tn], 0, 0, 0);
            }
        };

        // Software pipeline: two LDS stages, two register stages.  The global loads of tile t+2 are issued
        // while tile t is consumed, so each load has two MFMA stages to land; one barrier per stage.  A stage
        // buffer is rewritten only after the barrier of the following stage, which every wave passes after it
        // finished reading that buffer.
        float* As0 = smem;
        float* Bs0 = smem + AS_FLOATS;
        float* As1 = smem + STAGE_FLOATS;
        float* Bs1 = smem + STAGE_FLOATS + AS_FLOATS;
        const bool cache_idx = !A_KC && T.a_idx != nullptr;
        const int k_end_all = k_end;
        for (int kc = k_begin; kc < k_end_all; kc += (cache_idx ? GS_IDXCAP : (1 << 30))) {
            // row-gathered reduction operand: the gather indices of this k-chunk go to LDS once, so each stage
            // has ONE global latency (the row) instead of two dependent ones (index, then row)
            k_end = cache_idx ? min(kc + GS_IDXCAP, k_end_all) : k_end_all;
            if (cache_idx) {
                __syncthreads();
                idx_base = kc;
                for (int t = tid; t < k_end - kc; t += 256) idx_lds[t] = T.a_idx[kc + t];
                __syncthreads();
            }
            load_tiles(kc, ra0, rb0);
            if (kc + BK < k_end) load_tiles(kc + BK, ra1, rb1);
            for (int k0 = kc; k0 < k_end; k0 += 2 * BK) {
                store_tiles(ra0, rb0, As0, Bs0);
                __syncthreads();
                if (k0 + 2 * BK < k_end) load_tiles(k0 + 2 * BK, ra0, rb0);
                compute(As0, Bs0);
                if (k0 + BK >= k_end) break;
                store_tiles(ra1, rb1, As1, Bs1);
                __syncthreads();
                if (k0 + 3 * BK < k_end) load_tiles(k0 + 3 * BK, ra1, rb1);
                compute(As1, Bs1);
            }
            if (!cache_idx) break;
        }
        __syncthreads();  // LDS is reused by the next term
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    if (g.pool_s > 0) {
        // max-pool over groups of pool_s rows (aggregators.py:176-181: Dense(relu) then reduce_max over the neighbor
        // axis) without writing the [rows, hidden] activations: the activated tile goes to LDS (the K loop's buffers
        // are free after its last barrier), then one thread per (group, column) scans its pool_s rows.
        static_assert(2 * STAGE_FLOATS >= BM * BN, "pool epilogue needs the tile in LDS");
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int lc = wn * (BN / 2) + tn * 32 + l31;
                const float bv = (g.bias && n0 + lc < g.N) ? g.bias[n0 + lc] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = wm * (BM / 2) + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    float v = acc[tm][tn][e] + bv;
                    if (g.act == GS_ACT_RELU) v = fmaxf(v, 0.f);
                    smem[lr * BN + lc] = v;
                }
            }
        __syncthreads();
        const int s = g.pool_s;
        const int groups = g.pool_rows / s;
        const int64_t n_groups = gM / s;
        for (int item = tid; item < groups * BN; item += 256) {
            const int gi = item / BN, lc = item - gi * BN;
            const int64_t node = (int64_t)tile_m * groups + gi;
            if (node >= n_groups || n0 + lc >= g.N) continue;
            const float* col = smem + (gi * s) * BN + lc;
            float best = col[0];
            int arg = 0;
            for (int j = 1; j < s; ++j) {
                const float v = col[j * BN];
                if (v > best) { best = v; arg = j; }
            }
            g.pool_out[node * g.pool_ld + n0 + lc] = best;
            g.pool_arg[node * g.pool_lda + n0 + lc] = arg;
        }
        return;
    }
    float* C = g.C + (g.kchunk > 0 ? (int64_t)zslice * g.slab_stride : 0);
    const int n_total = g.N * ((g.nterms == 2 && g.concat) ? 2 : 1);
    const int n_pad = (n_total + 3) & ~3;
    if (BM * BN <= 2 * STAGE_FLOATS && !g.accumulate && g.kchunk == 0 && (g.N & 3) == 0 && (g.ldc & 3) == 0) {
        // plain output, whole float4 columns: the tile goes through LDS (the stage buffers are free now) and leaves as
        // 16-byte row-contiguous stores -- 16 per thread instead of 64 dword stores behind per-row branches (the unfused
        // pooling GEMM that writes H: 949 -> see DESIGN.md).  Same values, bit for bit.
        __syncthreads();
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int lc = wn * (BN / 2) + tn * 32 + l31;
                const int col = n0 + lc;
                const float bv = (g.bias && col < g.N) ? g.bias[col_off + col] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = wm * (BM / 2) + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    float v = acc[tm][tn][e] + bv;
                    if (g.act == GS_ACT_RELU) v = fmaxf(v, 0.f);
                    smem[lr * BN + lc] = v;
                }
            }
        __syncthreads();
        constexpr int C4 = BN / 4;
        for (int item = tid; item < BM * C4; item += 256) {
            const int r = item / C4, c4 = item - r * C4;
            const int64_t row = m0 + r;
            const int col = n0 + 4 * c4;
            if (row < gM && col < g.N)
                *reinterpret_cast<f32x4*>(C + row * g.ldc + col_off + col) = *reinterpret_cast<const f32x4*>(smem + r * BN + 4 * c4);
        }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = n0 + wn * (BN / 2) + tn * 32 + l31;
            const int gcol = col_off + col;
            const bool col_ok = col < g.N;
            const bool pad_col = !col_ok && (term1 == g.nterms) && gcol >= n_total && gcol < n_pad;
            const float bv = (g.bias && col_ok) ? g.bias[gcol] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + wm * (BM / 2) + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (row < gM) {
                    if (col_ok) {
                        float v = acc[tm][tn][e] + bv;
                        float* dst = C + row * g.ldc + gcol;
                        if (g.accumulate) v += *dst;
                        if (g.act == GS_ACT_RELU) v = fmaxf(v, 0.f);
                        *dst = v;
                    } else if (pad_col && g.kchunk == 0) {
                        C[row * g.ldc + gcol] = 0.f;
                    }
                }
            }
        }
}

__device__ __forceinline__ int64_t gs_ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const GemmArgs g_in) {
    constexpr int AS_FLOATS = A_KC ? BM * 36 : 32 * BM;
    constexpr int BS_FLOATS = B_KC ? BN * 36 : 32 * BN;
    __shared__ __attribute__((aligned(16))) float smem[2 * (AS_FLOATS + BS_FLOATS) + (A_KC ? 0 : GS_IDXCAP)];
    if (g_in.m_dev) {
        // the row count lives on the device (rows >= *m_dev do not exist; M is the grid's upper bound): the XCD swizzle runs
        // over the tiles that exist -- swizzled over the whole grid, the first XCDs would own all the work (measured: 30 % of
        // the rows took 88 % of the full time)
        const int64_t gM = min(g_in.M, (int64_t)max(*g_in.m_dev, 0));
        const int tiles_n_total = g_in.tiles_n * ((g_in.nterms == 2 && g_in.concat) ? 2 : 1);
        const int nwg = (int)gs_ceil_div_dev(gM, (int64_t)(g_in.pool_s > 0 ? g_in.pool_rows : BM)) * tiles_n_total;
        if ((int)blockIdx.x >= nwg) return;
        gemm_tile<BM, BN, A_KC, B_KC>(g_in, gs_xcd_swizzle(blockIdx.x, nwg), blockIdx.z, smem, gM);
        return;
    }
    gemm_tile<BM, BN, A_KC, B_KC>(g_in, gs_xcd_swizzle(blockIdx.x, gridDim.x), blockIdx.z, smem, g_in.M);
}

// Grouped weight-gradient launch: every dW = A^T·dZ of one backward pass (all layers, all variables, the bias
// gradients as ones^T·dZ) in ONE kernel.  blockIdx.x -> (problem, split-K slice, tile) through a prefix table.
#define GS_MAX_GROUP 12
struct GroupedArgs {
    GemmArgs p[GS_MAX_GROUP];
    int32_t block_start[GS_MAX_GROUP + 1];
    int32_t n;
};

__global__ __launch_bounds__(256) void gemm_grouped_tn_kernel(const GroupedArgs G) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (32 * 64 + 32 * 64) + GS_IDXCAP];
    const int bid = blockIdx.x;
    int p = 0;
    while (p + 1 < G.n && bid >= G.block_start[p + 1]) ++p;
    const GemmArgs& g = G.p[p];
    const int local = bid - G.block_start[p];
    const int tiles = g.tiles_m * g.tiles_n;
    const int z = local / tiles;
    gemm_tile<64, 64, false, false>(g, local - z * tiles, z, smem, g.M);
}

// Small-M contraction (M <= 2048: the layer-1 / head-sized GEMMs, which are pure latency with 64x64 tiles because
// only a few dozen workgroups exist).  One 32x32 output tile per workgroup; the 4 waves split K four ways (wave w
// owns k-stages w, w+4, ...) and stage their operand slices in WAVE-PRIVATE LDS regions, so the K loop has no
// barriers at all; the four partial tiles are summed in fixed order through LDS at the end.  A is k-contiguous
// (NN and NT forms); the summation order differs from the 64x64 kernel only in the final 4-way combine.
template <bool B_KC>
__global__ __launch_bounds__(256) void gemm_small_kernel(const GemmArgs g) {
    constexpr int BK = 32, KP = 36;
    constexpr int A_FLOATS = 32 * KP;
    constexpr int B_FLOATS = B_KC ? 32 * KP : BK * 32;
    constexpr int WAVE_FLOATS = A_FLOATS + B_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[4 * WAVE_FLOATS > 4 * 32 * 33 ? 4 * WAVE_FLOATS : 4 * 32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    float* As = smem + wave * WAVE_FLOATS;
    float* Bs = As + A_FLOATS;

    const int tiles_n_total = g.tiles_n * ((g.nterms == 2 && g.concat) ? 2 : 1);
    const int wgid = blockIdx.x;
    const int tile_m = wgid / tiles_n_total;
    int tile_n = wgid - tile_m * tiles_n_total;
    int term0 = 0, term1 = g.nterms, col_off = 0;
    if (g.nterms == 2 && g.concat) {
        term0 = tile_n / g.tiles_n;
        term1 = term0 + 1;
        tile_n -= term0 * g.tiles_n;
        col_off = term0 * g.N;
    }
    const int64_t m0 = (int64_t)tile_m * 32;
    const int n0 = tile_n * 32;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    for (int term = term0; term < term1; ++term) {
        const GemmTerm T = g.t[term];
        const int K = T.K;
        // per-lane source rows: lane -> (row = p*8 + lane/8, k-quad = lane%8) for p < 4
        const float* a_row[4];
        bool a_ok[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t r = m0 + p * 8 + (lane >> 3);
            a_ok[p] = r < g.M;
            const int64_t src = a_ok[p] ? (T.a_idx ? (int64_t)T.a_idx[r] : r) : 0;
            a_row[p] = T.A + src * T.lda + (lane & 7) * 4;
        }
        auto load = [&](int k0, f32x4 (&ra)[4], f32x4 (&rb)[4]) {
            const int k = k0 + (lane & 7) * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (a_ok[p] && k < K) {
                    v = *reinterpret_cast<const f32x4*>(a_row[p] + k0);
                    if (k + 3 >= K) {
                        if (k + 1 >= K) v.y = 0.f;
                        if (k + 2 >= K) v.z = 0.f;
                        if (k + 3 >= K) v.w = 0.f;
                    }
                }
                ra[p] = v;
            }
            if constexpr (B_KC) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int r = n0 + p * 8 + (lane >> 3);
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (r < g.N && k < K) {
                        v = *reinterpret_cast<const f32x4*>(T.B + (int64_t)r * T.ldb + k);
                        if (k + 3 >= K) {
                            if (k + 1 >= K) v.y = 0.f;
                            if (k + 2 >= K) v.z = 0.f;
                            if (k + 3 >= K) v.w = 0.f;
                        }
                    }
                    rb[p] = v;
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int kk = k0 + p * 8 + (lane >> 3);
                    const int r = n0 + (lane & 7) * 4;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (kk < K && r < g.N) {
                        v = *reinterpret_cast<const f32x4*>(T.B + (int64_t)kk * T.ldb + r);
                        if (r + 3 >= g.N) {
                            if (r + 1 >= g.N) v.y = 0.f;
                            if (r + 2 >= g.N) v.z = 0.f;
                            if (r + 3 >= g.N) v.w = 0.f;
                        }
                    }
                    rb[p] = v;
                }
            }
        };
        auto store = [&](const f32x4 (&ra)[4], const f32x4 (&rb)[4]) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<f32x4*>(&As[(p * 8 + (lane >> 3)) * KP + (lane & 7) * 4]) = ra[p];
            if constexpr (B_KC) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    *reinterpret_cast<f32x4*>(&Bs[(p * 8 + (lane >> 3)) * KP + (lane & 7) * 4]) = rb[p];
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    *reinterpret_cast<f32x4*>(&Bs[(p * 8 + (lane >> 3)) * 32 + (lane & 7) * 4]) = rb[p];
            }
        };
        f32x4 ra0[4], rb0[4], ra1[4], rb1[4];
        const int kstride = 4 * BK;                      // this wave's stages: k0 = wave*32, +128, ...
        int k0 = wave * BK;
        if (k0 < K) load(k0, ra0, rb0);
        if (k0 + kstride < K) load(k0 + kstride, ra1, rb1);
        int which = 0;
        for (; k0 < K; k0 += kstride) {
            if (which == 0) store(ra0, rb0); else store(ra1, rb1);
            if (k0 + 2 * kstride < K) {
                if (which == 0) load(k0 + 2 * kstride, ra0, rb0); else load(k0 + 2 * kstride, ra1, rb1);
            }
            which ^= 1;
#pragma unroll
            for (int gk = 0; gk < BK / 8; ++gk) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(&As[l31 * KP + gk * 8 + lh * 4]);
                float b[4];
                if constexpr (B_KC) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(&Bs[l31 * KP + gk * 8 + lh * 4]);
                    b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) b[i] = Bs[(gk * 8 + lh * 4 + i) * 32 + l31];
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[3], acc, 0, 0, 0);
            }
        }
    }
    // ---- combine the four K-partials (fixed order 0,1,2,3) and run the epilogue with coalesced float4 stores
    __syncthreads();
    float* part = smem;  // [4][32][33]
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        part[wave * (32 * 33) + row * 33 + l31] = acc[e];
    }
    __syncthreads();
    const int row = tid >> 3, c4 = (tid & 7) * 4;
    const int64_t grow = m0 + row;
    const int n_total = g.N * ((g.nterms == 2 && g.concat) ? 2 : 1);
    const int n_pad = (n_total + 3) & ~3;
    if (grow < g.M) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = n0 + c4 + i;
            const int gcol = col_off + col;
            const float* pp = part + row * 33 + c4 + i;
            float v = ((pp[0] + pp[32 * 33]) + pp[2 * 32 * 33]) + pp[3 * 32 * 33];
            if (col < g.N) {
                if (g.bias) v += g.bias[gcol];
                float* dst = g.C + grow * g.ldc + gcol;
                if (g.accumulate) v += *dst;
                if (g.act == GS_ACT_RELU) v = fmaxf(v, 0.f);
                *dst = v;
            } else if (term1 == g.nterms && gcol >= n_total && gcol < n_pad) {
                g.C[grow * g.ldc + gcol] = 0.f;
            }
        }
    }
}

// Horizontal fusion: ONE launch whose first `gemm_blocks` workgroups are the tiles of the (MFMA-bound) SAGE dense
// contraction and whose remaining workgroups run gather+mean jobs (HBM-bound, e.g. the NEXT step's layer-0
// neighbor means, which need no weights).  The dispatcher places the GEMM tiles first and back-fills every CU's
// free wave slots with gather waves, so the two roofs overlap on the same CUs without any cross-stream
// dependency (a fork/join hipGraph costs ~20 us of queue synchronisation per step on this platform).
#define GS_MAX_COJOBS 4
struct CoGather {
    GatherArgs job[GS_MAX_COJOBS];
    int64_t wave_start[GS_MAX_COJOBS + 1];  // prefix sums of work items (waves) per job
    int32_t n;
};

__global__ __launch_bounds__(256) void sage_dense_cogather_kernel(const GemmArgs g, const int gemm_blocks, const CoGather J) {
    constexpr int AS_FLOATS = 64 * 36;
    constexpr int BS_FLOATS = 32 * 64;
    __shared__ __attribute__((aligned(16))) float smem[2 * (AS_FLOATS + BS_FLOATS)];
    if ((int)blockIdx.x < gemm_blocks) {
        gemm_tile<64, 64, true, false>(g, gs_xcd_swizzle(blockIdx.x, gemm_blocks), 0, smem, g.M);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x - gemm_blocks) * 4 + (threadIdx.x >> 6);
    if (w >= J.wave_start[J.n]) return;  // wave-uniform
    int k = 0;
    while (k + 1 < J.n && w >= J.wave_start[k + 1]) ++k;
    const GatherArgs& a = J.job[k];
    if (a.s >= 8)
        gather_mean_wave<8>(a, w - J.wave_start[k], lane);
    else
        gather_mean_wave<1>(a, w - J.wave_start[k], lane);
}

// The same horizontal fusion for the grouped weight-gradient launch: the second part of the NEXT step's gather rides
// along with the (latency-bound) split-K wgrad tiles, so the HBM-bound gather is spread over both big GEMM launches
// of a step instead of stretching only the layer-0 forward.
__global__ __launch_bounds__(256) void gemm_grouped_tn_cogather_kernel(const GroupedArgs G, const int gemm_blocks,
                                                                       const CoGather J) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (32 * 64 + 32 * 64) + GS_IDXCAP];
    const int bid = blockIdx.x;
    if (bid < gemm_blocks) {
        int p = 0;
        while (p + 1 < G.n && bid >= G.block_start[p + 1]) ++p;
        const GemmArgs& g = G.p[p];
        const int local = bid - G.block_start[p];
        const int tiles = g.tiles_m * g.tiles_n;
        const int z = local / tiles;
        gemm_tile<64, 64, false, false>(g, local - z * tiles, z, smem, g.M);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)bid - gemm_blocks) * 4 + (threadIdx.x >> 6);
    if (w >= J.wave_start[J.n]) return;  // wave-uniform
    int k = 0;
    while (k + 1 < J.n && w >= J.wave_start[k + 1]) ++k;
    const GatherArgs& a = J.job[k];
    if (a.s >= 8)
        gather_mean_wave<8>(a, w - J.wave_start[k], lane);
    else
        gather_mean_wave<1>(a, w - J.wave_start[k], lane);
}

// ------------------------------------------------------------------------------------------ host side
template <int BM, int BN, bool A_KC, bool B_KC>
static int launch_gemm(GemmArgs& g, int nz, hipStream_t st) {
    g.tiles_m = (int)gs_ceil_div(g.M, BM);
    if (g.pool_s > 0) {
        g.pool_rows = (BM / g.pool_s) * g.pool_s;
        g.tiles_m = (int)gs_ceil_div(g.M, g.pool_rows);
    }
    g.tiles_n = (int)gs_ceil_div(g.N, BN);
    const int64_t nblk = (int64_t)g.tiles_m * g.tiles_n * ((g.nterms == 2 && g.concat) ? 2 : 1);
    GS_REQUIRE(nblk > 0 && nblk < (1ll << 31), "gemm: bad grid (%lld tiles)", (long long)nblk);
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, A_KC, B_KC>), dim3((unsigned)nblk, 1, (unsigned)nz), dim3(256), 0, st, g);
    GS_LAUNCH_CHECK("gemm_f32_mfma_kernel");
    return GS_OK;
}

template <bool B_KC>
static int launch_gemm_small(GemmArgs& g, hipStream_t st) {
    g.tiles_m = (int)gs_ceil_div(g.M, 32);
    g.tiles_n = (int)gs_ceil_div(g.N, 32);
    const int64_t nblk = (int64_t)g.tiles_m * g.tiles_n * ((g.nterms == 2 && g.concat) ? 2 : 1);
    GS_REQUIRE(nblk > 0 && nblk < (1ll << 31), "gemm_small: bad grid");
    hipLaunchKernelGGL((gemm_small_kernel<B_KC>), dim3((unsigned)nblk), dim3(256), 0, st, g);
    GS_LAUNCH_CHECK("gemm_small_kernel");
    return GS_OK;
}

template <bool A_KC, bool B_KC>
static int dispatch_gemm(GemmArgs& g, int nz, hipStream_t st) {
    if constexpr (A_KC) {
        // few rows: 64x64 tiles would give a few dozen latency-bound workgroups -> K-split 32x32 variant
        static const bool no_small = getenv("GS_NO_SMALL_GEMM") != nullptr;
        if (!no_small && nz == 1 && g.kchunk == 0 && g.M <= 2048) return launch_gemm_small<B_KC>(g, st);
    }
    // Large problems (>= 1024 128x128 tiles) use the 128x128 tile (2x2 MFMA tiles per wave, 4x the
    // arithmetic intensity per LDS byte); everything else uses 64x64 to put >= 256 workgroups on the chip.
    const int64_t big_tiles = gs_ceil_div(g.M, 128) * gs_ceil_div(g.N, 128) * ((g.nterms == 2 && g.concat) ? 2 : 1) * nz;
    if (big_tiles >= 1024) return launch_gemm<128, 128, A_KC, B_KC>(g, nz, st);
    return launch_gemm<64, 64, A_KC, B_KC>(g, nz, st);
}

static inline int rup4(int x) { return (x + 3) & ~3; }

// gather+mean job descriptors (C ABI) -> kernel argument block
static int build_cojobs(const gs_gather_desc* jobs_host, int32_t n_jobs, CoGather* Jout, int64_t* waves_out) {
    GS_REQUIRE(n_jobs >= 0 && n_jobs <= GS_MAX_COJOBS && (n_jobs == 0 || jobs_host), "co-gather: 0..%d jobs", GS_MAX_COJOBS);
    CoGather& J = *Jout;
    J.n = n_jobs;
    int64_t waves = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const gs_gather_desc& q = jobs_host[i];
        GS_CHECK_MAT(q.X, q.ldx, "co-gather job X");
        GS_CHECK_MAT(q.out, q.ldo, "co-gather job out");
        GS_REQUIRE(q.n > 0 && q.s > 0 && q.d > 0 && q.ldx >= rup4(q.d) && q.ldo >= rup4(q.d), "co-gather: bad job %d", i);
        if (q.self_src) GS_CHECK_MAT(q.self_src, q.ld_self, "co-gather job self");
        const int chunks = ((q.d + 3) / 4 + 63) / 64;
        J.job[i] = GatherArgs{q.X, q.ldx, q.idx, q.n, q.s, q.d, q.self_src, q.ld_self, q.self_idx, q.out, q.ldo,
                              q.self_src ? 1.0f / (float)(q.s + 1) : 1.0f / (float)q.s, chunks,
                              DropArgs{0ull, nullptr, 0u, 0u, 1.0f, 0, nullptr, 0}};
        J.wave_start[i] = waves;
        waves += q.n * (int64_t)chunks;
    }
    J.wave_start[n_jobs] = waves;
    *waves_out = waves;
    return GS_OK;
}

extern "C" int gs_gemm_f32(int transA, int transB, int64_t M, int32_t N, int64_t K, const float* A, int64_t lda,
                           const int32_t* a_row_idx, const float* B, int64_t ldb, const float* bias, int act,
                           float* C, int64_t ldc, void* stream) {
    if (M == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_CHECK_MAT(A, lda, "gs_gemm_f32 A");
    GS_CHECK_MAT(B, ldb, "gs_gemm_f32 B");
    GS_CHECK_MAT(C, ldc, "gs_gemm_f32 C");
    GS_REQUIRE(M >= 0 && N > 0 && K > 0 && K < (1ll << 31), "gs_gemm_f32: bad sizes");
    GS_REQUIRE(lda >= rup4(transA ? (int)M : (int)K) && ldb >= rup4(transB ? (int)K : N) && ldc >= rup4(N),
               "gs_gemm_f32: leading dimensions must be >= round_up(width, 4)");
    if (M == 0) return GS_OK;
    GemmArgs g = {};
    g.t[0] = GemmTerm{A, a_row_idx, B, lda, ldb, (int32_t)K};
    g.nterms = 1;
    g.M = M; g.N = N; g.C = C; g.ldc = ldc; g.bias = bias; g.act = act;
    hipStream_t st = (hipStream_t)stream;
    if (!transA && !transB) return dispatch_gemm<true, false>(g, 1, st);
    if (transA && !transB) return dispatch_gemm<false, false>(g, 1, st);
    if (!transA && transB) return dispatch_gemm<true, true>(g, 1, st);
    return dispatch_gemm<false, true>(g, 1, st);
}

extern "C" int gs_sage_dense_fwd(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self,
                                 const float* agg, int64_t ld_agg, const int32_t* agg_idx, int32_t d_agg, int64_t n,
                                 const float* W_self, int64_t ldw_self, const float* W_neigh, int64_t ldw_neigh,
                                 int32_t out_dim, int concat, int act, const float* bias, float* out, int64_t ldo,
                                 void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_CHECK_MAT(agg, ld_agg, "gs_sage_dense_fwd agg");
    GS_CHECK_MAT(W_neigh, ldw_neigh, "gs_sage_dense_fwd W_neigh");
    GS_CHECK_MAT(out, ldo, "gs_sage_dense_fwd out");
    GS_REQUIRE(n >= 0 && d_agg > 0 && out_dim > 0, "gs_sage_dense_fwd: bad sizes");
    GS_REQUIRE(ld_agg >= rup4(d_agg) && ldw_neigh >= rup4(out_dim), "gs_sage_dense_fwd: ld too small");
    if (self) {
        GS_CHECK_MAT(self, ld_self, "gs_sage_dense_fwd self");
        GS_CHECK_MAT(W_self, ldw_self, "gs_sage_dense_fwd W_self");
        GS_REQUIRE(d_self > 0 && ld_self >= rup4(d_self) && ldw_self >= rup4(out_dim), "gs_sage_dense_fwd: self ld too small");
        if (concat) GS_REQUIRE(out_dim % 4 == 0, "gs_sage_dense_fwd: concat needs out_dim %% 4 == 0 (got %d)", out_dim);
    }
    const int n_total = out_dim * ((self && concat) ? 2 : 1);
    GS_REQUIRE(ldo >= rup4(n_total), "gs_sage_dense_fwd: ldo too small");
    if (n == 0) return GS_OK;
    GemmArgs g = {};
    if (self) {
        g.t[0] = GemmTerm{self, self_idx, W_self, ld_self, ldw_self, d_self};
        g.t[1] = GemmTerm{agg, agg_idx, W_neigh, ld_agg, ldw_neigh, d_agg};
        g.nterms = 2;
        g.concat = concat ? 1 : 0;
    } else {
        g.t[0] = GemmTerm{agg, agg_idx, W_neigh, ld_agg, ldw_neigh, d_agg};
        g.nterms = 1;
    }
    g.M = n; g.N = out_dim; g.C = out; g.ldc = ldo; g.bias = bias; g.act = act;
    return dispatch_gemm<true, false>(g, 1, (hipStream_t)stream);
}

// out[i] = act(X[idx[i]] . W + bias) for i < min(n_max, *n_dev): the row count is a device word (e.g. the number of
// distinct sampled ids of this step, gs_unique_ids), the grid covers n_max.  Rows >= *n_dev of `out` are not written.
extern "C" int gs_dense_fwd_rows_dev(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_max,
                                     const int32_t* n_dev, const float* W, int64_t ldw, int32_t out_dim, int act,
                                     const float* bias, float* out, int64_t ldo, void* stream) {
    if (n_max == 0) return GS_OK;
    GS_CHECK_MAT(X, ldx, "gs_dense_fwd_rows_dev X");
    GS_CHECK_MAT(W, ldw, "gs_dense_fwd_rows_dev W");
    GS_CHECK_MAT(out, ldo, "gs_dense_fwd_rows_dev out");
    GS_REQUIRE(idx && n_dev && n_max > 2048 && d > 0 && out_dim > 0 && ldx >= rup4(d) && ldw >= out_dim && ldo >= rup4(out_dim),
               "gs_dense_fwd_rows_dev: bad args (n_max must be > 2048: the tiled kernels)");
    GemmArgs g = {};
    g.t[0] = GemmTerm{X, idx, W, ldx, ldw, d};
    g.nterms = 1;
    g.M = n_max; g.N = out_dim; g.C = out; g.ldc = ldo; g.bias = bias; g.act = act;
    g.m_dev = n_dev;
    return dispatch_gemm<true, false>(g, 1, (hipStream_t)stream);
}

extern "C" int gs_dense_pool_max_fwd(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_groups, int32_t s,
                                     const float* W, int64_t ldw, int32_t hidden, const float* bias, float* pooled,
                                     int64_t ldp, int32_t* argmax, int64_t lda, void* stream) {
    if (n_groups == 0) return GS_OK;
    GS_CHECK_MAT(X, ldx, "gs_dense_pool_max_fwd X");
    GS_CHECK_MAT(W, ldw, "gs_dense_pool_max_fwd W");
    GS_REQUIRE(pooled && argmax && n_groups > 0 && s > 0 && s <= 64 && d > 0 && hidden > 0, "gs_dense_pool_max_fwd: bad args (1 <= s <= 64)");
    GS_REQUIRE(ldx >= rup4(d) && ldw >= rup4(hidden) && ldp >= hidden && lda >= hidden, "gs_dense_pool_max_fwd: ld too small");
    GemmArgs g = {};
    g.t[0] = GemmTerm{X, idx, W, ldx, ldw, d};
    g.nterms = 1;
    g.M = n_groups * s; g.N = hidden; g.bias = bias; g.act = GS_ACT_RELU;
    g.pool_s = s; g.pool_out = pooled; g.pool_ld = ldp; g.pool_arg = argmax; g.pool_lda = lda;
    // 128-row tiles hold floor(128 / s) whole groups (125 of 128 rows at s = 25); small problems take 64-row tiles
    const int64_t big_tiles = gs_ceil_div(g.M, (128 / s) * s) * gs_ceil_div(hidden, 128);
    if (big_tiles >= 512) return launch_gemm<128, 128, true, false>(g, 1, (hipStream_t)stream);
    return launch_gemm<64, 64, true, false>(g, 1, (hipStream_t)stream);
}

extern "C" int gs_sage_dense_fwd_cogather(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self,
                                          const float* agg, int64_t ld_agg, const int32_t* agg_idx, int32_t d_agg,
                                          int64_t n, const float* W_self, int64_t ldw_self, const float* W_neigh,
                                          int64_t ldw_neigh, int32_t out_dim, int concat, int act, const float* bias,
                                          float* out, int64_t ldo, const gs_gather_desc* jobs_host, int32_t n_jobs,
                                          void* stream) {
    GS_REQUIRE(n > 0 && agg && W_neigh && out, "gs_sage_dense_fwd_cogather: bad args");
    GS_REQUIRE(n_jobs >= 0 && n_jobs <= GS_MAX_COJOBS && (n_jobs == 0 || jobs_host), "gs_sage_dense_fwd_cogather: 0..%d jobs", GS_MAX_COJOBS);
    GS_CHECK_MAT(agg, ld_agg, "gs_sage_dense_fwd_cogather agg");
    GS_CHECK_MAT(W_neigh, ldw_neigh, "gs_sage_dense_fwd_cogather W_neigh");
    GS_CHECK_MAT(out, ldo, "gs_sage_dense_fwd_cogather out");
    GS_REQUIRE(d_agg > 0 && out_dim > 0 && ld_agg >= rup4(d_agg) && ldw_neigh >= rup4(out_dim), "gs_sage_dense_fwd_cogather: ld too small");
    GemmArgs g = {};
    if (self) {                                   // two-term SAGE form (Mean / pooling aggregators)
        GS_CHECK_MAT(self, ld_self, "gs_sage_dense_fwd_cogather self");
        GS_CHECK_MAT(W_self, ldw_self, "gs_sage_dense_fwd_cogather W_self");
        GS_REQUIRE(d_self > 0 && ld_self >= rup4(d_self) && ldw_self >= rup4(out_dim), "gs_sage_dense_fwd_cogather: self ld too small");
        if (concat) GS_REQUIRE(out_dim % 4 == 0, "gs_sage_dense_fwd_cogather: concat needs out_dim %% 4 == 0");
        g.t[0] = GemmTerm{self, self_idx, W_self, ld_self, ldw_self, d_self};
        g.t[1] = GemmTerm{agg, agg_idx, W_neigh, ld_agg, ldw_neigh, d_agg};
        g.nterms = 2;
        g.concat = concat ? 1 : 0;
    } else {                                      // single contraction (GCN: means . W)
        g.t[0] = GemmTerm{agg, agg_idx, W_neigh, ld_agg, ldw_neigh, d_agg};
        g.nterms = 1;
    }
    const bool two_halves = g.nterms == 2 && g.concat;
    GS_REQUIRE(ldo >= rup4(out_dim * (two_halves ? 2 : 1)), "gs_sage_dense_fwd_cogather: ldo too small");
    g.M = n; g.N = out_dim; g.C = out; g.ldc = ldo; g.bias = bias; g.act = act;
    if (n <= 2048) {
        // small contraction: nothing worth overlapping with -> issue the gather jobs and the (K-split) GEMM as the
        // separate launches would, so results stay identical to the unfused calls
        for (int i = 0; i < n_jobs; ++i) {
            const gs_gather_desc& q = jobs_host[i];
            int rc = gs_gather_mean_fwd(q.X, q.ldx, q.idx, q.n, q.s, q.d, q.self_src, q.ld_self, q.self_idx, q.out, q.ldo, stream);
            if (rc != GS_OK) return rc;
        }
        return dispatch_gemm<true, false>(g, 1, (hipStream_t)stream);
    }
    g.tiles_m = (int)gs_ceil_div(n, 64);
    g.tiles_n = (int)gs_ceil_div(out_dim, 64);
    const int64_t gemm_blocks = (int64_t)g.tiles_m * g.tiles_n * (two_halves ? 2 : 1);
    CoGather J = {};
    int64_t waves = 0;
    {
        int rc = build_cojobs(jobs_host, n_jobs, &J, &waves);
        if (rc != GS_OK) return rc;
    }
    const int64_t blocks = gemm_blocks + gs_ceil_div(waves, 4);
    GS_REQUIRE(blocks < (1ll << 31), "gs_sage_dense_fwd_cogather: grid too large");
    hipLaunchKernelGGL(sage_dense_cogather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g,
                       (int)gemm_blocks, J);
    GS_LAUNCH_CHECK("sage_dense_cogather_kernel");
    return GS_OK;
}

extern "C" int gs_dense_wgrad(const float* A, int64_t lda, const int32_t* a_idx, int32_t d, const float* dZ,
                              int64_t ldz, int32_t col0, int32_t out_dim, int64_t n, int32_t n_slabs, float* slabs,
                              int64_t ld_slab, void* stream) {
    GS_CHECK_MAT(A, lda, "gs_dense_wgrad A");
    GS_CHECK_MAT(dZ, ldz, "gs_dense_wgrad dZ");
    GS_CHECK_MAT(slabs, ld_slab, "gs_dense_wgrad slabs");
    GS_REQUIRE(d > 0 && out_dim > 0 && n > 0 && n < (1ll << 31) && n_slabs > 0 && n_slabs < 65536, "gs_dense_wgrad: bad sizes");
    GS_REQUIRE(col0 >= 0 && col0 % 4 == 0, "gs_dense_wgrad: col0 must be a multiple of 4");
    GS_REQUIRE(lda >= rup4(d) && ldz >= col0 + rup4(out_dim) && ld_slab >= rup4(out_dim), "gs_dense_wgrad: ld too small");
    GemmArgs g = {};
    g.t[0] = GemmTerm{A, a_idx, dZ + col0, lda, ldz, (int32_t)n};
    g.nterms = 1;
    g.M = d; g.N = out_dim; g.C = slabs; g.ldc = ld_slab;
    g.slab_stride = (int64_t)d * ld_slab;
    g.kchunk = (int32_t)(gs_ceil_div(gs_ceil_div(n, n_slabs), 32) * 32);
    g.act = GS_ACT_IDENTITY;
    // long reductions over wide outputs (the MaxPool MLP gradient: [602 x 512] <- 133,120 rows) are throughput-bound:
    // 128x128 tiles (2x2 MFMA tiles per wave) double the flops per LDS byte; short ones are latency-bound -> 64x64
    static const int64_t big_n = getenv("GS_WGRAD_BIG_N") ? atoll(getenv("GS_WGRAD_BIG_N")) : 16384;
    if (n >= big_n && d >= 128 && out_dim >= 128) return launch_gemm<128, 128, false, false>(g, n_slabs, (hipStream_t)stream);
    // experiment hook (benchmarks/micro.py): GS_WGRAD_TILE=12864 | 128128 selects a larger split-K tile
    static const char* tile_env = getenv("GS_WGRAD_TILE");
    if (tile_env && atoi(tile_env) == 12864) return launch_gemm<128, 64, false, false>(g, n_slabs, (hipStream_t)stream);
    if (tile_env && atoi(tile_env) == 128128) return launch_gemm<128, 128, false, false>(g, n_slabs, (hipStream_t)stream);
    return launch_gemm<64, 64, false, false>(g, n_slabs, (hipStream_t)stream);
}

extern "C" int gs_sage_dense_dgrad(const float* dZ, int64_t ldz, int64_t n, int32_t out_dim, int fwd_concat,
                                   const float* W_self, int64_t ldw_self, const float* W_neigh, int64_t ldw_neigh,
                                   int32_t d_in, float* dX2, int64_t ldx, void* stream) {
    if (n == 0) return GS_OK;
    GS_CHECK_MAT(dZ, ldz, "gs_sage_dense_dgrad dZ");
    GS_CHECK_MAT(W_self, ldw_self, "gs_sage_dense_dgrad W_self");
    GS_CHECK_MAT(W_neigh, ldw_neigh, "gs_sage_dense_dgrad W_neigh");
    GS_CHECK_MAT(dX2, ldx, "gs_sage_dense_dgrad dX2");
    GS_REQUIRE(n > 0 && out_dim > 0 && d_in > 0 && d_in % 4 == 0, "gs_sage_dense_dgrad: d_in must be a positive multiple of 4");
    GS_REQUIRE(!fwd_concat || out_dim % 4 == 0, "gs_sage_dense_dgrad: concat needs out_dim %% 4 == 0");
    GS_REQUIRE(ldz >= (fwd_concat ? 2 : 1) * rup4(out_dim) && ldw_self >= rup4(out_dim) && ldw_neigh >= rup4(out_dim) &&
               ldx >= 2 * d_in, "gs_sage_dense_dgrad: ld too small");
    GemmArgs g = {};
    g.t[0] = GemmTerm{dZ, nullptr, W_self, ldz, ldw_self, out_dim};
    g.t[1] = GemmTerm{dZ + (fwd_concat ? out_dim : 0), nullptr, W_neigh, ldz, ldw_neigh, out_dim};
    g.nterms = 2;
    g.concat = 1;
    g.M = n; g.N = d_in; g.C = dX2; g.ldc = ldx;
    g.act = GS_ACT_IDENTITY;
    return dispatch_gemm<true, true>(g, 1, (hipStream_t)stream);
}

static int wgrad_grouped(const gs_wgrad_desc* descs_host, int32_t n_desc, const gs_gather_desc* jobs_host, int32_t n_jobs,
                         void* stream) {
    GS_REQUIRE(descs_host && n_desc > 0, "gs_dense_wgrad_grouped: bad args");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n_desc; base += GS_MAX_GROUP) {
        GroupedArgs G = {};
        const int cnt = std::min(GS_MAX_GROUP, n_desc - base);
        G.n = cnt;
        int64_t blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const gs_wgrad_desc& q = descs_host[base + i];
            GS_CHECK_MAT(q.A, q.lda, "gs_dense_wgrad_grouped A");
            GS_CHECK_MAT(q.dZ, q.ldz, "gs_dense_wgrad_grouped dZ");
            GS_CHECK_MAT(q.slabs, q.ld_slab, "gs_dense_wgrad_grouped slabs");
            GS_REQUIRE(q.d > 0 && q.out_dim > 0 && q.n > 0 && q.n < (1ll << 31) && q.n_slabs > 0, "gs_dense_wgrad_grouped: bad sizes");
            GS_REQUIRE(q.col0 >= 0 && q.col0 % 4 == 0, "gs_dense_wgrad_grouped: col0 must be a multiple of 4");
            GS_REQUIRE(q.lda >= rup4(q.d) && q.ldz >= q.col0 + rup4(q.out_dim) && q.ld_slab >= rup4(q.out_dim),
                       "gs_dense_wgrad_grouped: ld too small");
            GemmArgs& g = G.p[i];
            g.t[0] = GemmTerm{q.A, q.a_idx, q.dZ + q.col0, q.lda, q.ldz, (int32_t)q.n};
            g.nterms = 1;
            g.M = q.d; g.N = q.out_dim; g.C = q.slabs; g.ldc = q.ld_slab;
            g.slab_stride = (int64_t)q.d * q.ld_slab;
            g.kchunk = (int32_t)(gs_ceil_div(gs_ceil_div(q.n, q.n_slabs), 32) * 32);
            g.act = GS_ACT_IDENTITY;
            g.tiles_m = (int)gs_ceil_div(q.d, 64);
            g.tiles_n = (int)gs_ceil_div(q.out_dim, 64);
            G.block_start[i] = (int32_t)blocks;
            blocks += (int64_t)g.tiles_m * g.tiles_n * q.n_slabs;
        }
        G.block_start[cnt] = (int32_t)blocks;
        GS_REQUIRE(blocks > 0 && blocks < (1ll << 31), "gs_dense_wgrad_grouped: bad grid");
        const bool last = base + GS_MAX_GROUP >= n_desc;
        if (last && n_jobs > 0) {          // the gather jobs ride along with the last group
            CoGather J = {};
            int64_t waves = 0;
            int rc = build_cojobs(jobs_host, n_jobs, &J, &waves);
            if (rc != GS_OK) return rc;
            const int64_t total = blocks + gs_ceil_div(waves, 4);
            GS_REQUIRE(total < (1ll << 31), "gs_dense_wgrad_grouped_cogather: grid too large");
            hipLaunchKernelGGL(gemm_grouped_tn_cogather_kernel, dim3((unsigned)total), dim3(256), 0, st, G, (int)blocks, J);
            GS_LAUNCH_CHECK("gemm_grouped_tn_cogather_kernel");
        } else {
            hipLaunchKernelGGL(gemm_grouped_tn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, G);
            GS_LAUNCH_CHECK("gemm_grouped_tn_kernel");
        }
    }
    return GS_OK;
}

extern "C" int gs_dense_wgrad_grouped(const gs_wgrad_desc* descs_host, int32_t n_desc, void* stream) {
    return wgrad_grouped(descs_host, n_desc, nullptr, 0, stream);
}

extern "C" int gs_dense_wgrad_grouped_cogather(const gs_wgrad_desc* descs_host, int32_t n_desc,
                                               const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    return wgrad_grouped(descs_host, n_desc, jobs_host, n_jobs, stream);
}

extern "C" int gs_dense_dgrad(const float* dZ, int64_t ldz, int32_t col0, int32_t out_dim, int64_t n, const float* W,
                              int64_t ldw, int32_t d, float* dX, int64_t ldx, int accumulate, void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_CHECK_MAT(dZ, ldz, "gs_dense_dgrad dZ");
    GS_CHECK_MAT(W, ldw, "gs_dense_dgrad W");
    GS_CHECK_MAT(dX, ldx, "gs_dense_dgrad dX");
    GS_REQUIRE(d > 0 && out_dim > 0 && n >= 0, "gs_dense_dgrad: bad sizes");
    GS_REQUIRE(col0 >= 0 && col0 % 4 == 0, "gs_dense_dgrad: col0 must be a multiple of 4");
    GS_REQUIRE(ldz >= col0 + rup4(out_dim) && ldw >= rup4(out_dim) && ldx >= rup4(d), "gs_dense_dgrad: ld too small");
    if (n == 0) return GS_OK;
    GemmArgs g = {};
    g.t[0] = GemmTerm{dZ + col0, nullptr, W, ldz, ldw, out_dim};
    g.nterms = 1;
    g.M = n; g.N = d; g.C = dX; g.ldc = ldx;
    g.act = GS_ACT_IDENTITY;
    g.accumulate = accumulate ? 1 : 0;
    return dispatch_gemm<true, true>(g, 1, (hipStream_t)stream);
}

// ------------------------------------------------------------------------- elementwise helpers
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dY, int64_t lddy,
                                                      const float* __restrict__ Y, int64_t ldy, int64_t n,
                                                      int32_t n_cols, int act, float* __restrict__ dZ, int64_t lddz) {
    const int c4 = (n_cols + 3) / 4;
    const int64_t total = n * (int64_t)c4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / c4;
        const int col = (int)(t - r * c4) * 4;
        f32x4 g = *reinterpret_cast<const f32x4*>(dY + r * lddy + col);
        if (act == GS_ACT_RELU) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(Y + r * ldy + col);
            g.x = y.x > 0.f ? g.x : 0.f;
            g.y = y.y > 0.f ? g.y : 0.f;
            g.z = y.z > 0.f ? g.z : 0.f;
            g.w = y.w > 0.f ? g.w : 0.f;
        }
        if (col + 3 >= n_cols) {
            if (col + 1 >= n_cols) g.y = 0.f;
            if (col + 2 >= n_cols) g.z = 0.f;
            if (col + 3 >= n_cols) g.w = 0.f;
        }
        *reinterpret_cast<f32x4*>(dZ + r * lddz + col) = g;
    }
}

extern "C" int gs_act_bwd(const float* dY, int64_t lddy, const float* Y, int64_t ldy, int64_t n, int32_t n_cols,
                          int act, float* dZ, int64_t lddz, void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_CHECK_MAT(dY, lddy, "gs_act_bwd dY");
    GS_CHECK_MAT(dZ, lddz, "gs_act_bwd dZ");
    if (act == GS_ACT_RELU) GS_CHECK_MAT(Y, ldy, "gs_act_bwd Y");
    GS_REQUIRE(n >= 0 && n_cols > 0, "gs_act_bwd: bad sizes");
    if (n == 0) return GS_OK;
    const int64_t total = n * (int64_t)((n_cols + 3) / 4);
    int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 2048);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dY, lddy, Y, ldy, n, n_cols, act,
                       dZ, lddz);
    GS_LAUNCH_CHECK("act_bwd_kernel");
    return GS_OK;
}
