// fp32 contractions on the bf16 matrix pipe, EXACTLY: every fp32 operand is cut into three bf16 pieces
//     x = h + m + l,   h = top 8 significant bits of x, m = the next 8, l = the last 8   (24 = 8 + 8 + 8: no bit is lost)
// and a product of two fp32 numbers is the sum of piece products, each of which the bf16 MFMA forms exactly (8 x 8 bits) and
// accumulates in fp32.  Six of the nine piece products are kept -- hh, hm, mh, mm, hl, lh; the dropped ml, lm, ll are
// <= 3 * 2^-24 |x y|, the size of ONE fp32 rounding -- so a K-term dot product carries the error of an fp32 FMA chain
// (measured against fp64: tests/test_split_gemm_gpu.py), while v_mfma_f32_32x32x16_bf16 issues every 32 cycles for 32 x 32 x
// 16 MACs against 64 cycles for 32 x 32 x 2 on the fp32 form: 6 / 16 of the matrix-pipe time.  (MI355X_MICROARCH.md: fp32 MFMA
// = the fp32 VECTOR rate, 1/16 of bf16; there is no xf32 on gfx950.)  dtype stays f32: inputs, outputs and accumulation are
// fp32, no value is rounded to bf16 anywhere.
//
//   gs_split_rows             W [K, N] fp32 -> W3 [2 ceil(K/16)][3][N][8] bf16: per group of 8 k and piece, the N columns side by
//                             side (16 bytes each), so that the 32 lanes of a B-fragment load read 512 contiguous bytes (a
//                             first version kept a column's pieces together -- 48-byte records 3.6 KB apart between lanes:
//                             64 cache lines per wave load, and the kernel ran at the texture addresser's pace, 40 us)
//   gs_sage_dense_fwd_split   the layer-0 contraction of a mean / GCN step (aggregators.py:51-58 / :110), drop-in for
//                             gs_sage_dense_fwd_stream: A operands (gathered self rows, neighbor means) are read as fp32
//                             and cut in registers (11 VALU per 2 elements, in the shadow of the MFMAs), B operands come
//                             pre-cut from W3.  One WAVE per 32 x 64 output tile over the whole K: no LDS, no barrier, no
//                             split-K epilogue; gather jobs of the next step ride as extra workgroups like before.
#include "gs_common.h"
#include "gs_gather_dev.h"
#include "gs_sample_dev.h"
#include <stdlib.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t gs_hi16_pair(const float x, const float y) {   // bf16 bits of x (low half) and y (high half)
    return __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u);
}
__device__ __forceinline__ float gs_top16(const float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }

// two fp32 values -> their three bf16 pieces, packed pairwise.  Truncation, not rounding: every subtraction is exact.
__device__ __forceinline__ void gs_split2(const float x, const float y, uint32_t& h, uint32_t& m, uint32_t& l) {
#ifdef GS_DIAG_SPLIT_NOCUT    // diagnostics builds only: wrong values, the kernel's time without the cut
    h = __float_as_uint(x); m = __float_as_uint(y); l = h;
    return;
#endif
    const float rx = x - gs_top16(x), ry = y - gs_top16(y);       // <= 16 significant bits
    const float sx = rx - gs_top16(rx), sy = ry - gs_top16(ry);   // <= 8 significant bits: exact in bf16
    h = gs_hi16_pair(x, y);
    m = gs_hi16_pair(rx, ry);
    l = gs_hi16_pair(sx, sy);
}
__device__ __forceinline__ void gs_split8(const f32x4 a0, const f32x4 a1, u32x4& h, u32x4& m, u32x4& l) {
#ifdef GS_DIAG_SPLIT_NOCUT    // diagnostics builds only (benchmarks/probes/build_variant.sh): wrong values, the kernel's time without the cut
    h = u32x4{__float_as_uint(a0.x), __float_as_uint(a0.y), __float_as_uint(a0.z), __float_as_uint(a0.w)};
    m = u32x4{__float_as_uint(a1.x), __float_as_uint(a1.y), __float_as_uint(a1.z), __float_as_uint(a1.w)};
    l = h;
    return;
#endif
    uint32_t hh[4], mm[4], ll[4];
    gs_split2(a0.x, a0.y, hh[0], mm[0], ll[0]);
    gs_split2(a0.z, a0.w, hh[1], mm[1], ll[1]);
    gs_split2(a1.x, a1.y, hh[2], mm[2], ll[2]);
    gs_split2(a1.z, a1.w, hh[3], mm[3], ll[3]);
    h = u32x4{hh[0], hh[1], hh[2], hh[3]};
    m = u32x4{mm[0], mm[1], mm[2], mm[3]};
    l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}

__device__ __forceinline__ f32x16 gs_mfma_bf16(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------ W -> W3
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ W, int64_t ldw, int32_t K, int32_t N,
                                                         u32x4* __restrict__ W3) {
    const int KG = 4 * ((((K + 31) / 32) + 1) & ~1);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // t = kg * N + n: consecutive lanes, consecutive n
    if (t >= (int64_t)KG * N) return;
    const int kg = (int)(t / N), n = (int)(t - (int64_t)kg * N);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * kg + j;
        v[j] = k < K ? W[(int64_t)k * ldw + n] : 0.f;
    }
    u32x4 h, m, l;
    gs_split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, h, m, l);
    u32x4* dst = W3 + (int64_t)kg * 3 * N + n;
    dst[0] = h; dst[(int64_t)N] = m; dst[2 * (int64_t)N] = l;
}

// groups of 8 k in W3: the k16 steps of the streaming kernel (2 ceil(K/16)), rounded up to the tiled kernel's even count of 32-k
// stages -- everything beyond K is zero, so neither kernel masks or clamps its B loads
static inline int split_groups(int32_t K) { return 4 * ((((K + 31) / 32) + 1) & ~1); }
static inline int64_t split_rows_bytes(int32_t K, int32_t N) { return (int64_t)N * split_groups(K) * 48; }

extern "C" int gs_split_rows_bytes(int32_t K, int32_t N, int64_t* bytes_out_host) {
    GS_REQUIRE(K > 0 && N > 0 && bytes_out_host, "gs_split_rows_bytes: bad args");
    *bytes_out_host = split_rows_bytes(K, N);
    return GS_OK;
}

extern "C" int gs_split_rows(const float* W, int64_t ldw, int32_t K, int32_t N, void* W3, void* stream) {
    GS_REQUIRE(W && W3 && K > 0 && N > 0 && ldw >= N, "gs_split_rows: bad args");
    GS_REQUIRE(gs_aligned16(W3), "gs_split_rows: W3 must be 16-byte aligned");
    const int64_t total = (int64_t)N * split_groups(K);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)gs_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, K, N,
                       (u32x4*)W3);
    GS_LAUNCH_CHECK("split_rows_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------------------------------------ LDS-tiled form
// out[i] = act(X[idx[i]] . W + bias), i < min(n_max, *n_dev)  -- the pooling MLP of the max-pool aggregator on the step's
// distinct sampled ids (aggregators.py:176-179 via layers.py:104-116; 51 GF per Reddit step: the one contraction of the path
// that is bound by the matrix pipe, 564 us on the fp32 MFMA = 0.58 of its peak).  Same three-piece arithmetic as above.
//   workgroup (8 waves) = 128 rows x 128 columns, wave = 32 x 64 (two MFMA tiles, an h-h and a small-terms accumulator each);
//   stage = 32 k: every thread loads 8 floats of one gathered row and 3 x 16 bytes of W3, cuts the A values ONCE per
//   workgroup (44 VALU per 24 MFMAs) and writes the pieces to LDS; the MFMA fragments are ds_read_b128.  Two LDS buffers, one
//   barrier per stage, the global loads of stage s + 2 in flight under the MFMAs of stage s.
#define ST_LDA 40                      // bf16 per LDS row of an A plane: 32 + 8 pad (80 bytes: the 32 lanes of a fragment read
                                       // 16 bytes each at a 20-dword stride -> all 64 banks in use)
struct SplitTiledArgs {
    const float* X; const int32_t* idx; const u32x4* W3; const float* bias; float* out; const int32_t* n_dev;
    int64_t ldx, ldo;
    int32_t n_max, K, N, act;
    float* ws;            // wide form: partial tiles of the split-K tail round (nullable: no split), [<= n_cu][128][256]
    int32_t n_cu;         // ... and the number of workgroup slots of the chip (one workgroup per CU: 156 KB of LDS)
};

// Tail round of the wide form.  Its workgroups run one per CU in lock step (equal work, 156 KB of LDS each), so nwg = 1300 tiles
// on 256 CUs are SIX rounds of which the last one keeps 20 CUs busy: 346 us for 5.08 rounds of work.  With a workspace the
// rem = nwg % n_cu tiles of the last round are cut along K into S = min(stage pairs, n_cu / rem) parts -- rem * S workgroups
// that write fp32 partial tiles (no bias / activation) -- and split_tiled_fixup_kernel sums the S parts of a tile in part order
// (deterministic) behind it.  Tiles [0, full) are computed as before.
struct WideSchedule { int full, rem, S; };
__device__ __host__ __forceinline__ WideSchedule wide_schedule(const int nwg, const int stages2, const int n_cu, const bool have_ws) {
    WideSchedule w = {nwg, 0, 1};
    if (have_ws && n_cu > 0) {
        const int r = nwg % n_cu;
        if (r > 0 && 2 * r <= n_cu) {
            const int pairs = stages2 >> 1;
            int sp = pairs < n_cu / r ? pairs : n_cu / r;
            if (sp > 10) sp = 10;                               // (the fix-up launch keeps a tile's parts in registers: <= 10)
            if (sp >= 2) { w.full = nwg - r; w.rem = r; w.S = sp; }
        }
    }
    return w;
}

__global__ __launch_bounds__(512) void split_tiled_fwd_kernel(const SplitTiledArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_PLANE = 128 * ST_LDA * 2;                 // bytes
    constexpr int A_BYTES = 3 * A_PLANE, B_BYTES = 12 * 128 * 16, BUF = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int count = min(g.n_max, g.n_dev ? *g.n_dev : g.n_max);
    const int tiles_n = (g.N + 127) >> 7, tiles_m = (count + 127) >> 7;
    const int nwg = tiles_m * tiles_n;
    if ((int)blockIdx.x >= nwg) return;
    // XCD-aware: block b runs on XCD b % 8; consecutive LOGICAL tiles (the column tiles of one row tile) share an XCD's L2
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int K = g.K, N = g.N;
    const int stages = (K + 31) >> 5;
    // ---- global -> register roles (512 threads)
    const int arow = tid >> 2, aq = tid & 3;                   // A: row of the tile, 8-float quarter of the stage
    const int grow = min(m0 + arow, count - 1);
    const int64_t srow = g.idx ? (int64_t)g.idx[grow] : (int64_t)grow;
    const int bcol = tid & 127, bch = wave >> 1;               // B: column of the tile, chunks bch, bch + 4, bch + 8 (12 per stage;
    const int bc = min(n0 + bcol, N - 1);                      //    bch is wave-uniform: the group arithmetic stays scalar)
    const char* __restrict__ W3b = (const char*)g.W3;
    const uint32_t bcol_off = (uint32_t)bc * 16u, plane_b = (uint32_t)N * 16u;
    f32x4 ra[2][2];                                            // two register sets: stage s + 2 is requested while stage s
    u32x4 rb[2][3];                                            // computes and stage s + 1 is cut and written to LDS
    const int stages2 = (stages + 1) & ~1;                     // an odd stage count is padded with one all-zero stage
    const int K4 = ((K + 3) >> 2) << 2;                        // readable columns of a row
    // Requests are straight-line code with clamped addresses (always valid memory); what lies beyond K / beyond W3 is zeroed
    // when the registers are USED (lds_store) -- a branch or a select at request time would make the compiler wait for every
    // outstanding load there.
    auto gload = [&](const int set, const int s) {
#ifdef GS_DIAG_TILED_NOG      // diagnostics builds only: no global loads
#pragma unroll
        for (int v = 0; v < 2; ++v) ra[set][v] = f32x4{(float)s, 1.f, 2.f, (float)tid};
#pragma unroll
        for (int j = 0; j < 3; ++j) rb[set][j] = u32x4{(unsigned)s, 1u, 2u, (unsigned)tid};
        return;
#endif
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int k = min(32 * s + 8 * aq + 4 * v, K4 - 4);
            ra[set][v] = *reinterpret_cast<const f32x4*>(g.X + srow * g.ldx + k);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = bch + 4 * j;                         // chunk = (k-group of the stage) * 3 + piece: W3's own order
            const uint32_t off = (uint32_t)(12 * min(s, stages2 - 1) + c) * plane_b + bcol_off;     // (zero beyond K: no mask)
            rb[set][j] = *reinterpret_cast<const u32x4*>(W3b + off);
        }
    };
    auto lds_store = [&](const int set, const int s, unsigned char* buf) {
        f32x4 x[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int k = 32 * s + 8 * aq + 4 * v;
            // selects, not branches (a branch would end the scheduling region the MFMAs are in): zero beyond K -- the stage
            // that holds K, and the all-zero stage behind an odd stage count
            x[v].x = k < K ? ra[set][v].x : 0.f;
            x[v].y = k + 1 < K ? ra[set][v].y : 0.f;
            x[v].z = k + 2 < K ? ra[set][v].z : 0.f;
            x[v].w = k + 3 < K ? ra[set][v].w : 0.f;
        }
        u32x4 h0, m0_, l0;
        gs_split8(x[0], x[1], h0, m0_, l0);
        unsigned char* pa = buf + (arow * ST_LDA + 8 * aq) * 2;
        *reinterpret_cast<u32x4*>(pa) = h0;
        *reinterpret_cast<u32x4*>(pa + A_PLANE) = m0_;
        *reinterpret_cast<u32x4*>(pa + 2 * A_PLANE) = l0;
        unsigned char* pb = buf + A_BYTES;
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4*>(pb + ((bch + 4 * j) * 128 + bcol) * 16) = rb[set][j];
    };
    // 8 waves = 4 (rows of 32) x 2 (columns of 64): two waves per SIMD, so that one wave's LDS round trips and barrier waits
    // are covered by the other's MFMAs (4 waves of 64 x 64 -- half the LDS reads per MFMA -- left the matrix pipe idle two
    // thirds of the time: 465 us, of which 300 without a single MFMA)
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2], sml[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; sml[j][e] = 0.f; }
    auto compute = [&](const unsigned char* buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u32x4 fa[3], fb[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                fa[p] = *reinterpret_cast<const u32x4*>(buf + p * A_PLANE + ((32 * wm + l31) * ST_LDA + 16 * q + 8 * lh) * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fb[j][p] = *reinterpret_cast<const u32x4*>(buf + A_BYTES + (((2 * q + lh) * 3 + p) * 128 + 64 * wn + 32 * j + l31) * 16);
            // piece product outermost, the two column halves innermost: an MFMA that accumulates into the result of the one
            // issued right before it waits for that result
#ifdef GS_DIAG_TILED_NOMFMA   // diagnostics builds only: the kernel's time without the matrix pipe
            _Pragma("unroll") for (int p = 0; p < 3; ++p) asm volatile("" :: "v"(fa[p]), "v"(fb[0][p]), "v"(fb[1][p]));
#else
#define GS_PP(dst, pa, pb) _Pragma("unroll") for (int j = 0; j < 2; ++j) dst[j] = gs_mfma_bf16(fa[pa], fb[j][pb], dst[j]);
            GS_PP(sml, 0, 2)     // h l
            GS_PP(sml, 2, 0)     // l h
            GS_PP(sml, 1, 1)     // m m
            GS_PP(sml, 0, 1)     // h m
            GS_PP(sml, 1, 0)     // m h
            GS_PP(acc, 0, 0)     // h h
#undef GS_PP
#endif
        }
    };
    // ---- pipeline: stage s computes from LDS buffer s & 1 while stage s + 1 (register set (s + 1) & 1) is cut and written to
    // the other buffer in the shadow of the MFMAs, and stage s + 2 is requested into the register set stage s has left.
    // (the loop body is unrolled by two and free of branches: see stages2)
    gload(0, 0);
    gload(1, 1);
    lds_store(0, 0, smem);
    __syncthreads();
    for (int s = 0; s < stages2; s += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {                    // the register sets are indexed statically
            const int ss = s + par;
            unsigned char* cur = smem + par * BUF;
            unsigned char* nxt = smem + (par ^ 1) * BUF;
            gload(par, ss + 2);                                // (past the end: clamped addresses, values zeroed or unused)
            compute(cur);
            lds_store(par ^ 1, ss + 1, nxt);
            // the first 8 MFMAs run beside LDS reads only; the cut of stage s + 1 (whose loads then had a stage and a third to
            // arrive) and its LDS writes follow behind the other 16
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA
                if (q < 8) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                // two LDS reads
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                // five VALU
                    __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);                // an LDS read or write
                }
            }
            __syncthreads();
        }
    }
    // ---- bias + activation, then through LDS (free after the last barrier) so that a lane stores 16 contiguous bytes of a row:
    // 8 dwordx4 stores per wave instead of 32 dword stores (H is 170 MB per Reddit step: the store phase was 61 of 428 us).
    // C/D layout: row = (e&3) + 8 (e>>2) + 4 (lane>>5), column = lane & 31.
    float* otile = reinterpret_cast<float*>(smem) + wave * (32 * 68);          // [32 rows][64 + 4 pad] per wave
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + 64 * wn + 32 * j + l31;
        const float bv = (g.bias && col < N) ? g.bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = (acc[j][e] + sml[j][e]) + bv;
            if (g.act == GS_ACT_RELU) v = fmaxf(v, 0.f);
            otile[((e & 3) + 8 * (e >> 2) + 4 * lh) * 68 + 32 * j + l31] = v;
        }
    }
    // (wave-private region: no barrier, the wave's own LDS writes are ordered before its reads by lgkmcnt)
    const int c4 = (lane & 15) * 4, r0 = lane >> 4;
    const int colg = n0 + 64 * wn + c4;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = 4 * it + r0;
        const int row = m0 + 32 * wm + r;
        const f32x4 v = *reinterpret_cast<const f32x4*>(otile + r * 68 + c4);
#ifdef GS_DIAG_TILED_NOSTORE
        if (row < count && colg < N && v.x == 12345.678f) g.out[(int64_t)row * g.ldo + colg] = v.x;
#else
        if (row < count) {
            float* dst = g.out + (int64_t)row * g.ldo + colg;
            if (colg + 3 < N) *reinterpret_cast<f32x4*>(dst) = v;
            else {
                if (colg < N) dst[0] = v.x;
                if (colg + 1 < N) dst[1] = v.y;
                if (colg + 2 < N) dst[2] = v.z;
            }
        }
#endif
    }
}

// Wide form (round 5): workgroup = 128 rows x 256 columns, 8 waves = 2 (rows of 64) x 4 (columns of 64), wave tile 64 x 64 = four
// MFMA tiles with an h-h and a small-terms accumulator each (128 accumulator registers).  Why: in the 128 x 128 form above a
// wave reads 18 fragments (ds_read_b128) and the workgroup writes 48 (ds_write_b128) per 24 MFMAs and wave -- the LDS pipe is
// ~78 % busy (576 read + 624 write cycles per 1536 matrix-pipe cycles of a stage) and the matrix pipe reaches 0.36 of its roof.
// Here a wave reads 24 fragments per 48 MFMAs, the A rows are cut once per 256 instead of 128 output columns (44 VALU per
// 48 MFMAs) and the workgroup's LDS traffic is 768 read + 936 write cycles per 3072 matrix-pipe cycles: ~55 %.
// Stage = 32 k as above; the B registers are single-buffered (W3 is L2-resident; a stage now lasts twice as long), the A
// registers keep their two sets.  Same piece products in the same order per accumulator: bit-identical to the 128 x 128 form.
__global__ __launch_bounds__(512) void split_tiled_fwd_wide_kernel(const SplitTiledArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_PLANE = 128 * ST_LDA * 2;                 // bytes
    constexpr int A_BYTES = 3 * A_PLANE, B_BYTES = 12 * 256 * 16, BUF = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int count = min(g.n_max, g.n_dev ? *g.n_dev : g.n_max);
    const int tiles_n = (g.N + 255) >> 8, tiles_m = (count + 127) >> 7;
    const int nwg = tiles_m * tiles_n;
    const int K = g.K, N = g.N;
    const int stages = (K + 31) >> 5;
    const WideSchedule sch = wide_schedule(nwg, (stages + 1) & ~1, g.n_cu, g.ws != nullptr);
    if ((int)blockIdx.x >= sch.full + sch.rem * sch.S) return;
    int tile, part = -1;                                       // part >= 0: a K part of a tail-round tile (partial tile -> g.ws)
    if ((int)blockIdx.x < sch.full) {
        const int q8 = sch.full >> 3, r8 = sch.full & 7, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    } else {
        const int r = (int)blockIdx.x - sch.full;              // the S parts of a tile sit on S consecutive block ids
        tile = sch.full + r / sch.S;
        part = r - (r / sch.S) * sch.S;
    }
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 256;
    const int arow = tid >> 2, aq = tid & 3;                   // A: row of the tile, 8-float quarter of the stage
    const int grow = min(m0 + arow, count - 1);
    const int64_t srow = g.idx ? (int64_t)g.idx[grow] : (int64_t)grow;
    const int bcol = tid & 255, bch = wave >> 2;               // B: column of the tile, chunks bch, bch + 2, ... (12 per stage, 6 per thread)
    const int bc = min(n0 + bcol, N - 1);
    const char* __restrict__ W3b = (const char*)g.W3;
    const uint32_t bcol_off = (uint32_t)bc * 16u, plane_b = (uint32_t)N * 16u;
    f32x4 ra[2][2];
    u32x4 rb[6];
    const int stages2 = (stages + 1) & ~1;
    const int K4 = ((K + 3) >> 2) << 2;
    auto gload_a = [&](const int set, const int s) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int k = min(32 * s + 8 * aq + 4 * v, K4 - 4);
            ra[set][v] = *reinterpret_cast<const f32x4*>(g.X + srow * g.ldx + k);
        }
    };
    auto gload_b = [&](const int s) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int c = bch + 2 * j;
            const uint32_t off = (uint32_t)(12 * min(s, stages2 - 1) + c) * plane_b + bcol_off;
            rb[j] = *reinterpret_cast<const u32x4*>(W3b + off);
        }
    };
    auto lds_store_a = [&](const int set, const int s, unsigned char* buf) {
        f32x4 x[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int k = 32 * s + 8 * aq + 4 * v;
            x[v].x = k < K ? ra[set][v].x : 0.f;
            x[v].y = k + 1 < K ? ra[set][v].y : 0.f;
            x[v].z = k + 2 < K ? ra[set][v].z : 0.f;
            x[v].w = k + 3 < K ? ra[set][v].w : 0.f;
        }
        u32x4 h0, m0_, l0;
        gs_split8(x[0], x[1], h0, m0_, l0);
        unsigned char* pa = buf + (arow * ST_LDA + 8 * aq) * 2;
        *reinterpret_cast<u32x4*>(pa) = h0;
        *reinterpret_cast<u32x4*>(pa + A_PLANE) = m0_;
        *reinterpret_cast<u32x4*>(pa + 2 * A_PLANE) = l0;
    };
    auto lds_store_b = [&](unsigned char* buf) {
        unsigned char* pb = buf + A_BYTES;
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4*>(pb + ((bch + 2 * j) * 256 + bcol) * 16) = rb[j];
    };
    const int wm = wave >> 2, wn = wave & 3;
    f32x16 acc[2][2], sml[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; sml[i][j][e] = 0.f; }
    auto compute = [&](const unsigned char* buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u32x4 fa[2][3], fb[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fa[i][p] = *reinterpret_cast<const u32x4*>(buf + p * A_PLANE + ((64 * wm + 32 * i + l31) * ST_LDA + 16 * q + 8 * lh) * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fb[j][p] = *reinterpret_cast<const u32x4*>(buf + A_BYTES + (((2 * q + lh) * 3 + p) * 256 + 64 * wn + 32 * j + l31) * 16);
            // piece product outermost, the four tiles innermost: no MFMA accumulates into the result of the one issued before it
#define GS_PP(dst, pa, pb) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
                               dst[i][j] = gs_mfma_bf16(fa[i][pa], fb[j][pb], dst[i][j]);
            GS_PP(sml, 0, 2)     // h l
            GS_PP(sml, 2, 0)     // l h
            GS_PP(sml, 1, 1)     // m m
            GS_PP(sml, 0, 1)     // h m
            GS_PP(sml, 1, 0)     // m h
            GS_PP(acc, 0, 0)     // h h
#undef GS_PP
        }
    };
    // pipeline: stage s computes from LDS buffer s & 1; the B chunks of stage s + 1 are requested at the top of stage s (BEFORE the
    // A rows of stage s + 2: vmcnt retires in order, so the wait for B must not stand behind the younger A request), stage s + 1
    // is cut / copied into the other buffer behind the MFMAs.
    // (a K part of a tail-round tile covers the stage pairs [pairs part / S, pairs (part + 1) / S): the look-ahead loads past its
    //  end are valid addresses whose values are written to LDS and never read)
    int s_begin = 0, s_end = stages2;
    if (part >= 0) {
        const int pairs = stages2 >> 1;
        s_begin = 2 * ((pairs * part) / sch.S);
        s_end = 2 * ((pairs * (part + 1)) / sch.S);
    }
    gload_a(0, s_begin);
    gload_b(s_begin);
    gload_a(1, s_begin + 1);
    lds_store_a(0, s_begin, smem);
    lds_store_b(smem);
    __syncthreads();
    for (int s = s_begin; s < s_end; s += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int ss = s + par;
            unsigned char* cur = smem + par * BUF;
            unsigned char* nxt = smem + (par ^ 1) * BUF;
            gload_b(ss + 1);
            gload_a(par, ss + 2);
            compute(cur);
            lds_store_a(par ^ 1, ss + 1, nxt);
            lds_store_b(nxt);
#pragma unroll
            for (int q = 0; q < 48; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA
                if (q < 12) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                // two LDS reads
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                // three VALU
                    __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);                // an LDS read or write
                }
            }
            __syncthreads();
        }
    }
    // bias + activation, then through a wave-private LDS region (free after the last barrier) so that a lane stores 16 contiguous
    // bytes of a row; the wave's two row halves go through the same region one after the other
    float* otile = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    const bool partial = part >= 0;
    float* wtile = partial ? g.ws + (int64_t)((int)blockIdx.x - sch.full) * (128 * 256) : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + l31;
            const float bv = (!partial && g.bias && col < N) ? g.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = (acc[i][j][e] + sml[i][j][e]) + bv;
                if (!partial && g.act == GS_ACT_RELU) v = fmaxf(v, 0.f);
                otile[((e & 3) + 8 * (e >> 2) + 4 * lh) * 68 + 32 * j + l31] = v;
            }
        }
        const int c4 = (lane & 15) * 4, r0 = lane >> 4;
        const int colg = n0 + 64 * wn + c4;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = 4 * it + r0;
            const int row = m0 + 64 * wm + 32 * i + r;
            const f32x4 v = *reinterpret_cast<const f32x4*>(otile + r * 68 + c4);
            if (partial) {                                     // the whole 128 x 256 partial tile, dense (rows / columns past the end are finite and unused)
                *reinterpret_cast<f32x4*>(wtile + (64 * wm + 32 * i + r) * 256 + 64 * wn + c4) = v;
            } else if (row < count) {
                float* dst = g.out + (int64_t)row * g.ldo + colg;
                if (colg + 3 < N) *reinterpret_cast<f32x4*>(dst) = v;
                else {
                    if (colg < N) dst[0] = v.x;
                    if (colg + 1 < N) dst[1] = v.y;
                    if (colg + 2 < N) dst[2] = v.z;
                }
            }
        }
    }
}

// out tile = act(sum of the S partial tiles of a tail-round tile, in part order, + bias): one workgroup per (tile, 32-row band)
__global__ __launch_bounds__(256) void split_tiled_fixup_kernel(const SplitTiledArgs g) {
    // one workgroup per (tail tile, four rows); a thread owns 16 bytes of ONE row: its S partial values are independent loads
    // (a first version walked eight rows and the S parts in a serial loop per thread: 24 us of load latency for 26 MB)
    const int count = min(g.n_max, g.n_dev ? *g.n_dev : g.n_max);
    const int tiles_n = (g.N + 255) >> 8, tiles_m = (count + 127) >> 7;
    const int nwg = tiles_m * tiles_n;
    const WideSchedule sch = wide_schedule(nwg, (((g.K + 31) >> 5) + 1) & ~1, g.n_cu, g.ws != nullptr);
    const int t = (int)blockIdx.x >> 5, r = 4 * ((int)blockIdx.x & 31) + ((int)threadIdx.x >> 6);
    if (t >= sch.rem) return;
    const int tile = sch.full + t;
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 256;
    const int row = m0 + r;
    if (row >= count) return;
    const int c4 = ((int)threadIdx.x & 63) * 4;                 // 64 threads x 16 bytes = one 256-column row
    const int colg = n0 + c4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
        if (colg < g.N) bv.x = g.bias[colg];
        if (colg + 1 < g.N) bv.y = g.bias[colg + 1];
        if (colg + 2 < g.N) bv.z = g.bias[colg + 2];
        if (colg + 3 < g.N) bv.w = g.bias[colg + 3];
    }
    const float* base = g.ws + (int64_t)t * sch.S * (128 * 256) + r * 256 + c4;
    f32x4 pv[10];                                               // S <= 10 parts (stage pairs of K <= 640); summed in part order
#pragma unroll
    for (int p = 0; p < 10; ++p) pv[p] = *reinterpret_cast<const f32x4*>(base + (int64_t)min(p, sch.S - 1) * (128 * 256));
    f32x4 v = pv[0];
#pragma unroll
    for (int p = 1; p < 10; ++p) if (p < sch.S) v += pv[p];
    v += bv;
    if (g.act == GS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    float* dst = g.out + (int64_t)row * g.ldo + colg;
    if (colg + 3 < g.N) *reinterpret_cast<f32x4*>(dst) = v;
    else {
        if (colg < g.N) dst[0] = v.x;
        if (colg + 1 < g.N) dst[1] = v.y;
        if (colg + 2 < g.N) dst[2] = v.z;
    }
}

static int dense_fwd_rows_split_impl(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_max, const int32_t* n_dev,
                                     const void* W3, int32_t out_dim, int act, const float* bias, float* out, int64_t ldo,
                                     float* ws, int64_t ws_bytes, void* stream);

extern "C" int gs_dense_fwd_rows_split_ws_bytes(int64_t* bytes_out_host) {
    GS_REQUIRE(bytes_out_host, "gs_dense_fwd_rows_split_ws_bytes: null out");
    int dev = 0;
    hipDeviceProp_t prop;
    GS_HIP(hipGetDevice(&dev));
    GS_HIP(hipGetDeviceProperties(&prop, dev));
    *bytes_out_host = (int64_t)prop.multiProcessorCount * 128 * 256 * sizeof(float);
    return GS_OK;
}

extern "C" int gs_dense_fwd_rows_split_ws(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_max,
                                          const int32_t* n_dev, const void* W3, int32_t out_dim, int act, const float* bias,
                                          float* out, int64_t ldo, float* ws, int64_t ws_bytes, void* stream) {
    return dense_fwd_rows_split_impl(X, ldx, idx, d, n_max, n_dev, W3, out_dim, act, bias, out, ldo, ws, ws_bytes, stream);
}

extern "C" int gs_dense_fwd_rows_split(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_max, const int32_t* n_dev,
                                       const void* W3, int32_t out_dim, int act, const float* bias, float* out, int64_t ldo,
                                       void* stream) {
    return dense_fwd_rows_split_impl(X, ldx, idx, d, n_max, n_dev, W3, out_dim, act, bias, out, ldo, nullptr, 0, stream);
}

static int dense_fwd_rows_split_impl(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_max, const int32_t* n_dev,
                                     const void* W3, int32_t out_dim, int act, const float* bias, float* out, int64_t ldo,
                                     float* ws, int64_t ws_bytes, void* stream) {
    if (n_max == 0) return GS_OK;
    GS_REQUIRE(X && W3 && out && d > 0 && out_dim > 0 && n_max > 0 && n_max < (1ll << 30), "gs_dense_fwd_rows_split: bad args");
    GS_CHECK_MAT(X, ldx, "gs_dense_fwd_rows_split X");
    GS_REQUIRE(ldx >= ((d + 3) / 4) * 4 && ldo >= out_dim && gs_aligned16(W3), "gs_dense_fwd_rows_split: bad leading dimensions");
    GS_REQUIRE(ldo % 4 == 0 && gs_aligned16(out), "gs_dense_fwd_rows_split: out must be 16-byte aligned with ldo % 4 == 0");
    GS_REQUIRE(split_rows_bytes(d, out_dim) < (1ll << 32), "gs_dense_fwd_rows_split: W3 must stay below 4 GB");
    SplitTiledArgs g = {X, idx, (const u32x4*)W3, bias, out, n_dev, ldx, ldo, (int32_t)n_max, d, out_dim, act};
    static const int wide_min = getenv("GS_SPLIT_WIDE_MIN_N") ? atoi(getenv("GS_SPLIT_WIDE_MIN_N")) : 256;       // 0 = never
    if (wide_min > 0 && out_dim >= wide_min) {
        int64_t wblocks = gs_ceil_div(n_max, 128) * gs_ceil_div(out_dim, 256);
        const size_t wlds = 2 * (3 * 128 * ST_LDA * 2 + 12 * 256 * 16);
        static bool wattr_set = false;
        static int n_cu = 0;
        GS_LDS_ATTR(wlds, split_tiled_fwd_wide_kernel);
        if (!wattr_set) {
            int dev = 0;
            hipDeviceProp_t prop;
            GS_HIP(hipGetDevice(&dev));
            GS_HIP(hipGetDeviceProperties(&prop, dev));
            n_cu = prop.multiProcessorCount;
            wattr_set = true;
        }
        // split-K tail round (see wide_schedule): needs a workspace of one partial tile per CU
        static const bool no_tail = getenv("GS_SPLIT_WIDE_TAIL") && atoi(getenv("GS_SPLIT_WIDE_TAIL")) == 0;     // A/B hook
        const bool tail = ws && !no_tail && n_cu > 0 && ws_bytes >= (int64_t)n_cu * 128 * 256 * (int64_t)sizeof(float);
        GS_REQUIRE(!ws || gs_aligned16(ws), "gs_dense_fwd_rows_split_ws: the workspace must be 16-byte aligned");
        if (tail) { g.ws = ws; g.n_cu = n_cu; wblocks += n_cu; }
        hipLaunchKernelGGL(split_tiled_fwd_wide_kernel, dim3((unsigned)wblocks), dim3(512), wlds, (hipStream_t)stream, g);
        GS_LAUNCH_CHECK("split_tiled_fwd_wide_kernel");
        if (tail) {
            // at most n_cu / 2 tail tiles, four 32-row bands each (workgroups of tiles that do not exist return at once)
            hipLaunchKernelGGL(split_tiled_fixup_kernel, dim3((unsigned)(16 * n_cu)), dim3(256), 0, (hipStream_t)stream, g);
            GS_LAUNCH_CHECK("split_tiled_fixup_kernel");
        }
        return GS_OK;
    }
    const int64_t blocks = gs_ceil_div(n_max, 128) * gs_ceil_div(out_dim, 128);
    const size_t lds = 2 * (3 * 128 * ST_LDA * 2 + 12 * 128 * 16);
    GS_LDS_ATTR(lds, split_tiled_fwd_kernel);
    hipLaunchKernelGGL(split_tiled_fwd_kernel, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, g);
    GS_LAUNCH_CHECK("split_tiled_fwd_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------------------------------------ layer-0 forward, LDS-tiled (round 6)
// out = act([self[self_idx] . W_self | agg . W_neigh] + bias)  for the rows of ALL hops of a layer       (aggregators.py:51-64, :110-116)
// -- the contraction of gs_sage_dense_fwd_stream (fp32 MFMA, register-streaming: 24.6 us for the Reddit step's 1.74 GF = 45 % of
// the fp32 matrix pipe's peak; three rounds of tile variants did not move it) on the bf16 matrix pipe in the three-piece
// arithmetic above: 6 / 16 of the matrix-pipe time, the same fp32 accuracy class, fp32 operands in and out -- NO pre-cut copy of
// the weights (a re-cut launch behind every optimizer step costs what the kernel wins).  With the pipe out of the way the design is
// about filling 256 CUs with a 5632-row problem and about who cuts what:
//   workgroup (FOUR waves) = 64 rows x 128 columns of ONE term (Reddit: 88 row tiles x 2 terms = 176 workgroups, one round);
//   stage = 16 k.
//   A (the (gathered) rows): one float4 per thread and stage through registers, cut ONCE per workgroup (22 VALU per thread) and
//     written to LDS as pieces ([piece][row][16 k] bf16, 48-byte rows: conflict-free 16-byte fragment reads), two buffers;
//   B (the weights): raw fp32, global -> LDS by DMA (global_load_lds_dwordx4: a wave instruction = two k rows of the tile's 128
//     columns; no register, no ds_write), ring of four stages; wave w reads the 8 k of ITS 32 columns as eight ds_read_b32 and
//     cuts them itself (44 VALU): every B element is cut exactly once per workgroup, an A fragment (pre-cut) is shared by the
//     four waves;
//   a wave contracts every stage for its 64 x 32 tile: 12 MFMAs per stage, one wave per SIMD;
//   software pipeline: during stage s a wave issues the MFMAs of stage s from fragments it already holds, reads + cuts the
//     fragments of stage s + 1, cuts and stores the A pieces of stage s + 2, requests A of stage s + 6 and B of stage s + 3, and
//     meets the other waves once, at the end; the issue order inside a stage is pinned (a sched_barrier behind every slice).
// Why four waves and 50 KB: the riders (next step's gather) inherit the launch's workgroup shape -- threads, registers AND dynamic
// LDS.  The first version (8 waves = 2 K halves x 4 column groups, 32-k stages, 111 KB) was 2.6 us faster ALONE (20.2 vs 22.8 us)
// and left no room for a rider workgroup on a host's CU; this one shares its CU with two (profiles/r06_tiled3_wgrad_ab.txt:
// with half of the gather riding 31.1 vs 38.9 us; unsupervised step 157.7 vs 167.2 us).
// Gather jobs of the next step ride as extra workgroups like in every launch of the step.
struct Fwd3Term {
    const float* A;        // [*, lda] fp32; row i of the term is A[a_idx ? a_idx[i] : i]
    const int32_t* a_idx;  // nullable row gather (layer 0: the self rows of the feature table)
    const float* W;        // [K, ldw] fp32
    int32_t lda, ldw, K;
};
struct Fwd3Args {
    Fwd3Term t[2];
    int32_t nterms;        // 1, or 2 (concat: term i writes columns [i*N, (i+1)*N))
    int32_t M, N;
    float* C;
    int32_t ldc;
    const float* bias;     // indexed by output column (incl. the concat offset), nullable
    int32_t act;
    int32_t tiles_m, tiles_n, n_tiles;     // n_tiles = tiles_m * tiles_n * nterms
};
// independent 16-byte loads a RIDER wave of the two tiled launches keeps in flight (gs_gather_dev.h: run_gather_item)
#ifndef T3_RIDER_U
#define T3_RIDER_U 8
#endif
#define F3_BM 64
#define F3_NA 2            // A-piece buffers (stage s + 1 being read, stage s + 2 being written)
#define F3_NB 4            // raw B stages in flight
#define F3_KS 16           // k per stage
#define F3_LDA 24          // bf16 per LDS row of an A plane: 16 + 8 pad (48 bytes: 16 lanes of a 16-byte fragment read cover all banks)
#ifdef F3_TIMELINE
// Diagnostics build only (-DF3_TIMELINE, benchmarks/timeline_tiled3.py): shader-clock and wall-clock stamps of waves 0 and 4 of the
// first 256 workgroups.  [wg][wave 0 | 4][0] entry [1] operands requested [2] prologue barrier passed [3] K loop left [4] K halves
// summed [5] stores issued [6] wall clock at entry [7] wall clock at exit; [8 + 3 s + 0..2] stage s: MFMAs and slices issued |
// counted wait over | barrier passed
__device__ unsigned long long g_f3_tl[256 * 2 * 96];
extern "C" int gs_debug_f3_timeline(unsigned long long* out_host, int n) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_f3_tl), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#define F3_STAMPV(k, v) do { if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 256 && (k) < 96) g_f3_tl[(blockIdx.x * 2 + (wave >> 2)) * 96 + (k)] = (v); } while (0)
#define F3_STAMP(k) F3_STAMPV(k, clock64())
#else
#define F3_STAMP(k) do { } while (0)
#define F3_STAMPV(k, v) do { } while (0)
#endif

__global__ __launch_bounds__(256) void sage_tiled3_fwd_kernel(const Fwd3Args g, const CoGatherS J) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_PLANE = F3_BM * F3_LDA * 2;                // bytes
    constexpr int A_BYTES = 3 * A_PLANE, B_BYTES = F3_KS * 128 * 4;
    constexpr int B_BASE = F3_NA * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= g.n_tiles) {
        run_gather_item<T3_RIDER_U>(J, ((int64_t)blockIdx.x - g.n_tiles) * 4 + wave, lane);
        return;
    }
    const int l31 = lane & 31, lh = lane >> 5;
    F3_STAMP(0);
    F3_STAMPV(6, wall_clock64());
    // XCD-aware: block b runs on XCD b % 8; consecutive LOGICAL tiles share an XCD's L2 (a term's W stays resident there)
    const int nwg = g.n_tiles;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    const int per_term = g.tiles_m * g.tiles_n;
    const int term = tile / per_term;
    const int it = tile - term * per_term;
    const int tile_m = it / g.tiles_n, tile_n = it - tile_m * g.tiles_n;
    const int m0 = tile_m * F3_BM, n0 = tile_n * 128;
    const Fwd3Term& T = g.t[term];
    const int K = T.K, N = g.N, M = g.M;
    const int stages = (K + F3_KS - 1) / F3_KS;
    const int K4 = ((K + 3) >> 2) << 2;                        // readable columns of a row
    // ---- A: (row tid >> 2, float4 tid & 3) of the 64 x 16 stage tile, through registers
    const int arow = tid >> 2, aq = tid & 3;
    const int grow = min(m0 + arow, M - 1);
    const int64_t srow = T.a_idx ? (int64_t)T.a_idx[grow] : (int64_t)grow;
    const float* __restrict__ xrow = T.A + srow * T.lda;
    const int a_wr = (arow * F3_LDA + 4 * aq) * 2;
    // ---- B: wave w moves chunks 2 w, 2 w + 1 of a stage's 8 (chunk = two k rows x 128 columns = 1 KB, lane-linear in LDS)
    const int bcol = min(n0 + 4 * l31, N - 4);                 // (columns beyond N: a valid column again, computed and never stored)
    const float* __restrict__ wcol = T.W + bcol;
    const int ldw = T.ldw;
    // A ring of FOUR staging sets: the set stage s + 2 is cut from is re-requested for stage s + 6 at once -- four stages (~2 us)
    // of cover for a gathered row's trip to HBM.  Requests are hand-written (asm): the compiler does not track them, so it adds no
    // wait of its own (it answered every use with vmcnt(0) once LDS-DMA requests shared the queue); the counted wait in front of
    // every stage barrier (F3_BARRIER) is what guarantees them -- see the accounting there.
    f32x4 ra[4];
    // (F3_DIAG_*: diagnostics builds only, benchmarks/probes/build_variant.sh -- wrong values, the kernel's time without one component)
    auto gload_a = [&](const int set, const int s) {           // straight-line request with a clamped (always valid) address
        const int k = min(F3_KS * s + 4 * aq, K4 - 4);
#ifdef F3_DIAG_NOGA
        ra[set] = f32x4{(float)k, 1.f, 2.f, (float)tid};
#else
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[set]) : "v"(xrow + k));
#endif
    };
    auto dma_b = [&](const int s, const int slot) {            // rows beyond K: the last row again (A is zero there)
#ifndef F3_DIAG_NOGB
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = 2 * wave + i;
            const int k = min(F3_KS * s + 2 * c + lh, K - 1);
            __builtin_amdgcn_global_load_lds(wcol + (int64_t)k * ldw, (lds_ptr_t)(smem + B_BASE + slot * B_BYTES + c * 1024), 16, 0, 0);
        }
#endif
    };
    const int wn = wave;                                       // 32-column group
    f32x16 acc[2], sml[2];                                     // [row tile of 32]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[i][e] = 0.f; sml[i][e] = 0.f; }
    // fragments of a stage: TWO register sets -- the MFMAs of stage s run on one while the other receives stage s + 1's
    u32x4 fa[2][2][3], fb[2][3];
    const int a_rd = (l31 * F3_LDA + 8 * lh) * 2;                                    // + 32 i rows, + piece plane
    const int b_rd = B_BASE + ((8 * lh) * 128 + 32 * wn + l31) * 4;                  // + i k rows (512 bytes each)
    // The stage barrier by hand: "my LDS writes and reads of this stage are done" (lgkmcnt(0)) + "the B stage the next stage reads
    // has landed" (a COUNTED vmcnt: the younger requests stay in flight across the barrier) + s_barrier.  __syncthreads, and also a
    // release fence on the local address space (the compiler orders the LDS-DMA writes behind it), would wait for vmcnt(0): every
    // request of the stages ahead would have to land before any wave may go on -- a stage would cost a memory round trip.
#define F3_BARRIER(cnt) do { asm volatile("s_waitcnt vmcnt(" #cnt ") lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); \
                             asm volatile("" ::: "memory"); } while (0)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;                      // LDS byte address of the dynamic segment
    auto cut_store_a = [&](const int set, const int s, unsigned char* buf) {
        const int k = F3_KS * s + 4 * aq;
        const float x0 = k < K ? ra[set].x : 0.f, x1 = k + 1 < K ? ra[set].y : 0.f;       // selects: zero beyond K (NaN pads must not leak)
        const float x2 = k + 2 < K ? ra[set].z : 0.f, x3 = k + 3 < K ? ra[set].w : 0.f;
        uint32_t h0, m0_, l0, h1, m1_, l1;
        gs_split2(x0, x1, h0, m0_, l0);
        gs_split2(x2, x3, h1, m1_, l1);
        // (hand-written stores: behind a C++ store to LDS the compiler waits for EVERY outstanding LDS-DMA -- it cannot tell that
        //  the B ring and the A-piece buffers never overlap -- i.e. vmcnt(0) once per stage; F3_BARRIER waits for lgkmcnt(0))
        const uint32_t ad = lds0 + (uint32_t)(buf - smem) + (uint32_t)a_wr;
        const u32x2 vh = u32x2{h0, h1}, vm = u32x2{m0_, m1_}, vl = u32x2{l0, l1};
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:%4\n\tds_write_b64 %0, %3 offset:%5"
                     :: "v"(ad), "v"(vh), "v"(vm), "v"(vl), "n"(A_PLANE), "n"(2 * A_PLANE));      // (no "memory": see above)
    };
    auto read_cut_frags = [&](const int set, const unsigned char* abuf, const unsigned char* bslot) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[set][i][p] = *reinterpret_cast<const u32x4*>(abuf + a_rd + i * (32 * F3_LDA * 2) + p * A_PLANE);
        float bw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bw[i] = *reinterpret_cast<const float*>(bslot + (b_rd - B_BASE) + i * 512);
        gs_split8(f32x4{bw[0], bw[1], bw[2], bw[3]}, f32x4{bw[4], bw[5], bw[6], bw[7]}, fb[set][0], fb[set][1], fb[set][2]);
    };
    // ---- prologue: ONE round trip -- A of stages 0..5 and B of stages 0..2 requested together; stages 0, 1 cut and stored
    //      (their sets re-requested for stages 6, 7), the fragments of stage 0 read and cut
    dma_b(0, 0);                                               // (the weights do not wait for the row ids)
    dma_b(1, 1);
    dma_b(2, 2);
    {
        f32x4 t0, t1;
        const int k0 = min(4 * aq, K4 - 4), k1 = min(F3_KS + 4 * aq, K4 - 4);
#ifdef F3_DIAG_NOGA
        t0 = t1 = f32x4{(float)k0, 1.f, 2.f, (float)k1};
#else
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t0) : "v"(xrow + k0));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t1) : "v"(xrow + k1));
#endif
        gload_a(2, 2); gload_a(3, 3); gload_a(0, 4); gload_a(1, 5);
        F3_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]) :: "memory");
        const f32x4 s0 = ra[0], s1 = ra[1];
        ra[0] = t0; cut_store_a(0, 0, smem);
        ra[1] = t1; cut_store_a(1, 1, smem + A_BYTES);
        ra[0] = s0; ra[1] = s1;                                // (sets 0, 1 hold stages 4, 5; sets 2, 3 stages 2, 3)
    }
    F3_BARRIER(0);
    F3_STAMP(2);
    read_cut_frags(0, smem, smem + B_BASE);
    // byte offsets of the A-piece buffers of stage s + 1 | s + 2 (= the one stage s was read from, during stage s - 1: every wave
    // has passed a barrier since); B slots are (stage) % 4
    int oa_nxt = A_BYTES, oa_nn = 0;
    const int stages4 = (stages + 3) & ~3;                     // whole rings: padded with all-zero stages (A is masked beyond K)
    for (int s = 0; s < stages4; s += 4) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {                       // the register sets are indexed statically
            const int ss = s + q4;
            constexpr int dummy_ = 0; (void)dummy_;
            const int r = q4 & 1;                              // fragment set of this stage
            const int as = (q4 + 2) & 3;                       // A staging set that holds stage ss + 2
            const unsigned char* na = smem + oa_nxt;
            const unsigned char* nb = smem + B_BASE + ((ss + 1) & 3) * B_BYTES + (b_rd - B_BASE);
            unsigned char* wa = smem + oa_nn;
#define F3_SB __builtin_amdgcn_sched_barrier(0);
#ifdef F3_DIAG_NOMFMA
#define F3_MM(dst, i, pa, pb) asm volatile("" :: "v"(fa[r][i][pa]), "v"(fb[r][pb])); F3_SB
#else
#define F3_MM(dst, i, pa, pb) dst[i] = gs_mfma_bf16(fa[r][i][pa], fb[r][pb], dst[i]); F3_SB
#endif
#ifdef F3_DIAG_NOLDSR
#define F3_RA(i, p) fa[r ^ 1][i][p] = u32x4{(unsigned)(p), (unsigned)(uintptr_t)na, 2u, (unsigned)tid}; F3_SB
#define F3_RW(i) bw[i] = __uint_as_float((unsigned)(uintptr_t)nb + i);
#else
#define F3_RA(i, p) fa[r ^ 1][i][p] = *reinterpret_cast<const u32x4*>(na + a_rd + (i) * (32 * F3_LDA * 2) + (p) * A_PLANE); F3_SB
#define F3_RW(i) bw[i] = *reinterpret_cast<const float*>(nb + (i) * 512);
#endif
            float bw[8];
            uint32_t bh[4], bm[4], bl[4];
            // One stage of a wave in a PINNED issue order: 12 MFMAs (small terms h l, l h, m m, h m, m h into sml, h h into acc;
            // the same accumulator every other MFMA), each followed by a slice of the other work.
            F3_MM(sml, 0, 0, 2)
            F3_RW(0) F3_RW(1) F3_RW(2) F3_RW(3) F3_RW(4) F3_RW(5) F3_RW(6) F3_RW(7)
            F3_SB
            F3_MM(sml, 1, 0, 2) F3_RA(0, 0)
            F3_MM(acc, 0, 0, 0) F3_RA(0, 1)
            F3_MM(acc, 1, 0, 0) F3_RA(0, 2)
            F3_MM(sml, 0, 2, 0) F3_RA(1, 0)
            gs_split2(bw[0], bw[1], bh[0], bm[0], bl[0]);
            F3_SB
            F3_MM(sml, 1, 2, 0) F3_RA(1, 1)
            gs_split2(bw[2], bw[3], bh[1], bm[1], bl[1]);
            F3_SB
            F3_MM(sml, 0, 1, 1) F3_RA(1, 2)
            gs_split2(bw[4], bw[5], bh[2], bm[2], bl[2]);
            F3_SB
            F3_MM(sml, 1, 1, 1)
            gs_split2(bw[6], bw[7], bh[3], bm[3], bl[3]);
            fb[r ^ 1][0] = u32x4{bh[0], bh[1], bh[2], bh[3]};
            fb[r ^ 1][1] = u32x4{bm[0], bm[1], bm[2], bm[3]};
            fb[r ^ 1][2] = u32x4{bl[0], bl[1], bl[2], bl[3]};
            F3_SB
            F3_MM(sml, 0, 0, 1)
            F3_SB
            F3_MM(sml, 1, 0, 1)
#ifndef F3_DIAG_NOLDSW
            cut_store_a(as, ss + 2, wa);
#else
            asm volatile("" :: "v"(ra[as]));
#endif
            F3_SB
            F3_MM(sml, 0, 1, 0)
            dma_b(ss + 3, (ss + 3) & 3);
            gload_a(as, ss + 6);
            F3_SB
            F3_MM(sml, 1, 1, 0)
            F3_STAMP(8 + 3 * ss);
            // What the NEXT stage consumes has landed: its raw B (stage ss + 2, requested during stage ss - 1) and the A set it cuts
            // (stage ss + 3, requested during stage ss - 3).  A thread's requests retire IN ORDER; per stage it issues its two B
            // requests and THEN its A request, so behind B(ss + 2) lie A(ss + 5) of the same stage and this stage's three:
            // vmcnt(4).  (With the A request in front of the B requests the wait for B(ss + 2) was also a wait for A(ss + 5),
            // issued one stage before -- the four-stage A ring covered one stage of an HBM round trip: ~500 cycles of every stage
            // were spent in this wait.)
#ifdef F3_TIMELINE
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            F3_STAMP(9 + 3 * ss);
#endif
            F3_BARRIER(4);
            F3_STAMP(10 + 3 * ss);
            F3_SB
#undef F3_MM
#undef F3_RA
#undef F3_RW
#undef F3_SB
            const int t = oa_nxt; oa_nxt = oa_nn; oa_nn = t;
        }
    }
    F3_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the look-ahead requests of the last stages
    __syncthreads();
    F3_STAMP(4);
    // ---- the wave's 64 x 32 tile: bias + activation, then through a wave-private LDS staging tile to 16-byte row segments
    float* otile = reinterpret_cast<float*>(smem) + wave * (64 * 36);     // [64 rows][32 + 4 pad] per wave
    const int col_off = term * N;
    {
        const int col = n0 + 32 * wn + l31;
        const float bv = (g.bias && col < N) ? g.bias[col_off + col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = (acc[i][e] + sml[i][e]) + bv;
                if (g.act == GS_ACT_RELU) v = fmaxf(v, 0.f);
                otile[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh) * 36 + l31] = v;     // C/D layout: row = (e&3) + 8 (e>>2) + 4 (lane>>5)
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (wave-private region: the wave's own LDS writes before its reads)
    const int c4 = (lane & 7) * 4, r0 = lane >> 3;
    const int colg = n0 + 32 * wn + c4;
#pragma unroll
    for (int itr = 0; itr < 8; ++itr) {
        const int r = 8 * itr + r0;
        const int row = m0 + r;
        const f32x4 v = *reinterpret_cast<const f32x4*>(otile + r * 36 + c4);
        if (row < M) {
            float* dst = g.C + (int64_t)row * g.ldc + col_off + colg;
            if (colg + 3 < N) *reinterpret_cast<f32x4*>(dst) = v;
            else {
                if (colg < N) dst[0] = v.x;
                if (colg + 1 < N) dst[1] = v.y;
                if (colg + 2 < N) dst[2] = v.z;
            }
        }
    }
    F3_STAMP(5);
    F3_STAMPV(7, wall_clock64());
}

extern "C" int gs_sage_dense_fwd_tiled3(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self, const float* agg,
                                        int64_t ld_agg, int32_t d_agg, int64_t n, const float* W_self, int64_t ldw_self,
                                        const float* W_neigh, int64_t ldw_neigh, int32_t out_dim, int act, const float* bias,
                                        float* out, int64_t ldo, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream) {
    if (n == 0 && n_jobs == 0) return GS_OK;
    GS_REQUIRE(agg && W_neigh && out && d_agg > 0 && out_dim > 0 && n >= 0 && n < (1ll << 30), "gs_sage_dense_fwd_tiled3: bad args");
    GS_REQUIRE(!self || (W_self && d_self > 0), "gs_sage_dense_fwd_tiled3: W_self / d_self missing");
    GS_CHECK_MAT(agg, ld_agg, "gs_sage_dense_fwd_tiled3 agg");
    GS_CHECK_MAT(W_neigh, ldw_neigh, "gs_sage_dense_fwd_tiled3 W_neigh");
    if (self) {
        GS_CHECK_MAT(self, ld_self, "gs_sage_dense_fwd_tiled3 self");
        GS_CHECK_MAT(W_self, ldw_self, "gs_sage_dense_fwd_tiled3 W_self");
    }
    GS_REQUIRE(ld_agg >= ((d_agg + 3) / 4) * 4 && (!self || ld_self >= ((d_self + 3) / 4) * 4), "gs_sage_dense_fwd_tiled3: ld must be >= round_up(d, 4)");
    GS_REQUIRE(out_dim % 4 == 0 && ldw_neigh >= out_dim && (!self || ldw_self >= out_dim), "gs_sage_dense_fwd_tiled3: out_dim must be a multiple of 4, ldw >= out_dim");
    GS_REQUIRE(ldo >= out_dim * (self ? 2 : 1) && ldo % 4 == 0 && gs_aligned16(out), "gs_sage_dense_fwd_tiled3: out must be 16-byte aligned, ldo a multiple of 4");
    GS_REQUIRE(std::max(ld_self, ld_agg) < (1ll << 31) && std::max(ldw_self, ldw_neigh) < (1ll << 31), "gs_sage_dense_fwd_tiled3: 32-bit leading dimensions");
    Fwd3Args g = {};
    g.nterms = self ? 2 : 1;
    if (self) {
        g.t[0] = Fwd3Term{self, self_idx, W_self, (int32_t)ld_self, (int32_t)ldw_self, d_self};
        g.t[1] = Fwd3Term{agg, nullptr, W_neigh, (int32_t)ld_agg, (int32_t)ldw_neigh, d_agg};
    } else {
        g.t[0] = Fwd3Term{agg, nullptr, W_neigh, (int32_t)ld_agg, (int32_t)ldw_neigh, d_agg};
    }
    g.M = (int32_t)n; g.N = out_dim; g.C = out; g.ldc = (int32_t)ldo; g.bias = bias; g.act = act;
    g.tiles_m = (int)gs_ceil_div(n, F3_BM);
    g.tiles_n = (int)gs_ceil_div(out_dim, 128);
    g.n_tiles = n > 0 ? g.tiles_m * g.tiles_n * g.nterms : 0;
    CoGatherS J = {};
    int64_t waves = 0;
    int rc = build_cojobs_s(jobs_host, n_jobs, &J, &waves);
    if (rc != GS_OK) return rc;
    const int64_t blocks = g.n_tiles + gs_ceil_div(waves, 4);
    GS_REQUIRE(blocks > 0 && blocks < (1ll << 31), "gs_sage_dense_fwd_tiled3: grid too large");
    const size_t lds = F3_NA * (3 * F3_BM * F3_LDA * 2) + F3_NB * (F3_KS * 128 * 4);
    GS_LDS_ATTR(lds, sage_tiled3_fwd_kernel);
    hipLaunchKernelGGL(sage_tiled3_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, g, J);
    GS_LAUNCH_CHECK("sage_tiled3_fwd_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradients, LDS-tiled
// gs_dense_wgrad_grouped_tiled3: every weight gradient of a backward pass, dW = A[a_idx]^T . dZ[:, col0 : col0 + out_dim], in the
// three-piece arithmetic above -- the grouped launch of gs_dense_wgrad_grouped_stream (same descriptors, same split-K slabs, same
// riders), LDS-tiled like sage_tiled3_fwd_kernel.  Both operands are row-contiguous over the REDUCTION index (k = a batch row), the
// matrix pipe wants 8 consecutive k per lane: both go global -> LDS raw (fp32, LDS-DMA, no register, no ds_write) as [16 k][64 m] and
// [16 k][128 n] stage tiles, and a wave reads the 8 k of ITS column as eight ds_read_b32 (lanes side by side: conflict-free) and cuts
// them in registers -- the transposition costs nothing but the cut itself.
//   workgroup (FOUR waves) = one (64 m x 128 n tile, reduction slice) of one problem; wave w owns the 32-column group w over both
//   32-row blocks of m: per 16-k stage and wave 24 ds_read_b32, 12 two-element cuts, 12 MFMAs;
//   ring of W3_NS stage slots (12 KB each) filled W3_NS - 1 stages ahead, ONE barrier per stage behind a counted vmcnt;
//   a slice's source rows (<= 1024: the row ids of a gathered problem) are read once into LDS; 64-bit row addresses (tables beyond
//   4 GB are fine); rows beyond the slice are masked in the B fragment (the clamped row is read, its dZ values are zeroed).
// Why four waves and 16-k stages: a launch has ONE workgroup shape -- threads, registers AND dynamic LDS -- for all its roles, so the
// riders (next step's gather, no LDS of their own) inherit the hosts' allocation.  At 8 waves / 100 KB (first version: 21.7 us
// alone, but +5 us per step) no rider workgroup fits on a CU beside a host; at 4 waves / 52 KB / < 170 VGPRs a host shares its CU
// with two rider workgroups (8 rider waves per CU -- what the stream kernel's launch gives them).
#ifndef W3_NS
#define W3_NS 4
#endif
#define W3_KS 16           // k per stage
#define W3_MAXROWS 1024
#define W3_MAXP 12
struct Wg3Prob {
    const float* A;        // [*, lda]: row r of the reduction is A[a_idx ? a_idx[r] : r]
    const int32_t* a_idx;  // nullable row gather
    const float* dZ;       // [n, ldz], already offset by col0
    float* slabs;          // [n_slabs][d][ld_slab]
    int32_t lda, ldz, ld_slab, zcols;      // zcols: readable columns of a dZ row from the offset pointer (multiple of 4)
    int32_t n, d, out_dim;
    int32_t tiles_m, tiles_n, n_slabs, kchunk;   // kchunk: rows per slice, a multiple of 32
    int32_t item_start;
};
struct Wg3Args {
    Wg3Prob p[W3_MAXP];
    int32_t n, n_items;
};

__global__ __launch_bounds__(256) void wgrad_tiled3_kernel(const Wg3Args G, const FanoutArgs F, const CoGatherS J) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_BYTES = W3_KS * 64 * 4, B_BYTES = W3_KS * 128 * 4, S_BYTES = A_BYTES + B_BYTES;
    constexpr int IDX_BASE = W3_NS * S_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= G.n_items) {
        // Riders behind the contraction workgroups: first the fan-out SAMPLER of a later mini-batch, one root per workgroup (five
        // dependent round trips of almost no work: as a rider of the optimizer launch it was what that launch waited for -- 8 us
        // against the optimizer's own 5.5 -- here it ends long before the contraction does); then gather+mean waves.
        const int64_t r = (int64_t)blockIdx.x - G.n_items;
#ifdef W3_NO_SAMPLER              // diagnostics switch
        const int64_t sblocks = 0;
        if (false) {
#else
        const int64_t sblocks = (F.B + 3) >> 2;                     // one root per WAVE: four per rider slot
        if (r < sblocks) {
#endif
            const int64_t root = 4 * r + wave;
            if (root < F.B) {
                int32_t* ints = reinterpret_cast<int32_t*>(smem) + wave * GS_FANOUT_LDS_INTS(GS_FANOUT_LDS_SMALL);   // (the wave's
                sample_fanout_root<GS_FANOUT_LDS_SMALL, true>(F, root,                                              //  share of the launch's LDS)
                                                              reinterpret_cast<int32_t (*)[GS_FANOUT_LDS_SMALL]>(ints),
                                                              reinterpret_cast<int32_t (*)[GS_LAW_COLS]>(ints + 2 * GS_FANOUT_LDS_SMALL));
            }
            return;
        }
        run_gather_item<T3_RIDER_U>(J, (r - sblocks) * 4 + wave, lane);
        return;
    }
    const int l31 = lane & 31, lh = lane >> 5;
    // XCD-aware: block b runs on XCD b % 8; consecutive LOGICAL items (the m tiles of one slice: the same dZ rows) share an L2
    const int nwg = G.n_items;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, lcl = blockIdx.x >> 3;
    const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + lcl;
    int pi = 0;
    while (pi + 1 < G.n && item >= G.p[pi + 1].item_start) ++pi;
    const Wg3Prob& q = G.p[pi];
    const int local = item - q.item_start;
    const int tiles = q.tiles_m * q.tiles_n;
    const int z = local / tiles;
    const int tt = local - z * tiles;
    const int tile_m = tt / q.tiles_n, tile_n = tt - tile_m * q.tiles_n;
    const int m0 = tile_m * 64, n0 = tile_n * 128;
    const int rb = z * q.kchunk, re = min(rb + q.kchunk, q.n);
    const int len = max(re - rb, 0);                           // (an empty slice still writes its slab: zeros)
    const int rlast = max(re, 1) - 1;                          // every row read is clamped to a valid one
    const int stages = (len + W3_KS - 1) / W3_KS;
    const int stages4 = (stages + 3) & ~3;
    // the slice's source rows (<= W3_MAXROWS), once: the row ids of a gathered problem, else the rows themselves -- one code path
    int32_t* idxs = reinterpret_cast<int32_t*>(smem + IDX_BASE);
#pragma unroll
    for (int t = tid; t < W3_MAXROWS; t += 256) {
        const int rr = min(rb + t, rlast);
        idxs[t] = q.a_idx ? q.a_idx[rr] : rr;
    }
    __syncthreads();
    // ---- requests: wave w moves A chunk w (4 k rows x 64 m = 1 KB, lane-linear in LDS) and B chunks 2 w, 2 w + 1 (2 k rows x 128 n)
    const int akk = 4 * wave + (lane >> 4);
    const float* __restrict__ acol = q.A + min(m0 + 4 * (lane & 15), q.lda - 4);       // (columns beyond the row: a valid chunk again,
    const float* __restrict__ zcol = q.dZ + min(n0 + 4 * l31, q.zcols - 4);            //  computed and never stored)
    const int lda = q.lda, ldz = q.ldz;
    auto a_row = [&](const int s) -> int64_t {                 // source row of this lane's A request of stage s
        return (int64_t)idxs[min(W3_KS * s + akk, W3_MAXROWS - 1)];
    };
    auto dma = [&](const int s, const int slot, const int64_t arow) {
        unsigned char* base = smem + slot * S_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = 2 * wave + i;
            const int r = min(rb + W3_KS * s + 2 * c + lh, rlast);
            __builtin_amdgcn_global_load_lds(zcol + (int64_t)r * ldz, (lds_ptr_t)(base + A_BYTES + c * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds(acol + arow * lda, (lds_ptr_t)(base + wave * 1024), 16, 0, 0);
    };
    const int wn = wave;                                       // 32-column group
    f32x16 acc[2], sml[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[i][e] = 0.f; sml[i][e] = 0.f; }
    u32x4 fa[2][2][3], fb[2][3];
    const int kfrag = 8 * lh;                                  // first of this lane's 8 k inside a stage
    const int a_rd = (kfrag * 64 + l31) * 4;                   // + i k rows (256 bytes each), + 128 for the second m block
    const int b_rd = A_BYTES + (kfrag * 128 + 32 * wn + l31) * 4;      // + i k rows (512 bytes each)
    // stage barrier by hand, see sage_tiled3_fwd_kernel: counted vmcnt (younger requests stay in flight) + lgkmcnt(0) + s_barrier
#if W3_NS == 4
#define W3_INFLIGHT 3
#elif W3_NS == 5
#define W3_INFLIGHT 6
#elif W3_NS == 6
#define W3_INFLIGHT 9
#else
#define W3_INFLIGHT 15
#endif
#define W3_WAIT_STR2(x) #x
#define W3_WAIT_STR(x) W3_WAIT_STR2(x)
#define W3_BARRIER() do { asm volatile("s_waitcnt vmcnt(" W3_WAIT_STR(W3_INFLIGHT) ") lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); \
                          asm volatile("" ::: "memory"); } while (0)
    auto read_cut_frags = [&](const int set, const unsigned char* slot, const int s) {
        float bw[8], aw[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bw[i] = *reinterpret_cast<const float*>(slot + b_rd + i * 512);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) aw[j][i] = *reinterpret_cast<const float*>(slot + a_rd + j * 128 + i * 256);
        const int lim = len - W3_KS * s - kfrag;
#pragma unroll
        for (int i = 0; i < 8; ++i) bw[i] = i < lim ? bw[i] : 0.f;
        gs_split8(f32x4{bw[0], bw[1], bw[2], bw[3]}, f32x4{bw[4], bw[5], bw[6], bw[7]}, fb[set][0], fb[set][1], fb[set][2]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            gs_split8(f32x4{aw[j][0], aw[j][1], aw[j][2], aw[j][3]}, f32x4{aw[j][4], aw[j][5], aw[j][6], aw[j][7]},
                      fa[set][j][0], fa[set][j][1], fa[set][j][2]);
    };
    // ---- prologue: stages 0 .. W3_NS - 2 requested together; stage 0's fragments read and cut
#pragma unroll
    for (int s = 0; s < W3_NS - 1; ++s) dma(s, s, a_row(s));
    int64_t arow_nxt = a_row(W3_NS - 1);
    W3_BARRIER();                                              // stages 0 and 1 have landed (everyone's share of them)
    read_cut_frags(0, smem, 0);
    int slot_rd = 1, slot_wr = W3_NS - 1;                      // slot of stage ss + 1 | of stage ss + W3_NS - 1
    // The stages whose look-ahead reads lie wholly inside the slice run WITHOUT the row mask (a branch inside a stage would break the
    // pinned issue order: the compiler sinks the cuts behind it); the last ones (and the zero padding of the ring) run the masked copy.
    auto run = [&](auto masked, const int s_begin, const int s_end) {
    for (int s = s_begin; s < s_end; s += 4) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int ss = s + q4;
            const int r = q4 & 1;
            const unsigned char* nb = smem + slot_rd * S_BYTES;
#define W3_SB __builtin_amdgcn_sched_barrier(0);
#define W3_MM(dst, i, pa, pb) dst[i] = gs_mfma_bf16(fa[r][i][pa], fb[r][pb], dst[i]); W3_SB
#define W3_RB(i) bw[i] = *reinterpret_cast<const float*>(nb + b_rd + (i) * 512);
#define W3_RA(j, i) aw[j][i] = *reinterpret_cast<const float*>(nb + a_rd + (j) * 128 + (i) * 256);
            float bw[8], aw[2][8];
            uint32_t ph[4], pm[4], pl[4];
            // One stage of a wave in a PINNED issue order: 12 MFMAs of stage ss on one fragment set, each followed by a slice of
            // the work for stage ss + 1 (24 reads, 12 cuts into the other set) and of the requests for stage ss + W3_NS - 1.
            W3_MM(sml, 0, 0, 2)
            W3_RB(0) W3_RB(1) W3_RB(2) W3_RB(3) W3_RB(4) W3_RB(5) W3_RB(6) W3_RB(7)
            W3_RA(0, 0) W3_RA(0, 1) W3_RA(0, 2) W3_RA(0, 3) W3_RA(0, 4) W3_RA(0, 5) W3_RA(0, 6) W3_RA(0, 7)
            W3_SB
            W3_MM(sml, 1, 0, 2)
            W3_RA(1, 0) W3_RA(1, 1) W3_RA(1, 2) W3_RA(1, 3) W3_RA(1, 4) W3_RA(1, 5) W3_RA(1, 6) W3_RA(1, 7)
            W3_SB
            W3_MM(acc, 0, 0, 0)
            if constexpr (decltype(masked)::value) {           // the slice ends inside stage ss + 1, or before it
                const int lim = len - W3_KS * (ss + 1) - kfrag;
#pragma unroll
                for (int i = 0; i < 8; ++i) bw[i] = i < lim ? bw[i] : 0.f;
            }
            gs_split2(bw[0], bw[1], ph[0], pm[0], pl[0]);
            W3_SB
            W3_MM(acc, 1, 0, 0)
            gs_split2(bw[2], bw[3], ph[1], pm[1], pl[1]);
            W3_SB
            W3_MM(sml, 0, 2, 0)
            gs_split2(bw[4], bw[5], ph[2], pm[2], pl[2]);
            W3_SB
            W3_MM(sml, 1, 2, 0)
            gs_split2(bw[6], bw[7], ph[3], pm[3], pl[3]);
            fb[r ^ 1][0] = u32x4{ph[0], ph[1], ph[2], ph[3]};
            fb[r ^ 1][1] = u32x4{pm[0], pm[1], pm[2], pm[3]};
            fb[r ^ 1][2] = u32x4{pl[0], pl[1], pl[2], pl[3]};
            W3_SB
            W3_MM(sml, 0, 1, 1)
            gs_split2(aw[0][0], aw[0][1], ph[0], pm[0], pl[0]);
            gs_split2(aw[0][2], aw[0][3], ph[1], pm[1], pl[1]);
            W3_SB
            W3_MM(sml, 1, 1, 1)
            gs_split2(aw[0][4], aw[0][5], ph[2], pm[2], pl[2]);
            gs_split2(aw[0][6], aw[0][7], ph[3], pm[3], pl[3]);
            fa[r ^ 1][0][0] = u32x4{ph[0], ph[1], ph[2], ph[3]};
            fa[r ^ 1][0][1] = u32x4{pm[0], pm[1], pm[2], pm[3]};
            fa[r ^ 1][0][2] = u32x4{pl[0], pl[1], pl[2], pl[3]};
            W3_SB
            W3_MM(sml, 0, 0, 1)
            gs_split2(aw[1][0], aw[1][1], ph[0], pm[0], pl[0]);
            gs_split2(aw[1][2], aw[1][3], ph[1], pm[1], pl[1]);
            W3_SB
            W3_MM(sml, 1, 0, 1)
            gs_split2(aw[1][4], aw[1][5], ph[2], pm[2], pl[2]);
            gs_split2(aw[1][6], aw[1][7], ph[3], pm[3], pl[3]);
            fa[r ^ 1][1][0] = u32x4{ph[0], ph[1], ph[2], ph[3]};
            fa[r ^ 1][1][1] = u32x4{pm[0], pm[1], pm[2], pm[3]};
            fa[r ^ 1][1][2] = u32x4{pl[0], pl[1], pl[2], pl[3]};
            W3_SB
            W3_MM(sml, 0, 1, 0)
            dma(ss + W3_NS - 1, slot_wr, arow_nxt);            // into the slot stage ss - 1 was read from (during stage ss - 2)
            W3_SB
            W3_MM(sml, 1, 1, 0)
            arow_nxt = a_row(ss + W3_NS);
            // stage ss + 2 (what the next stage reads) has landed: behind it only the W3_NS - 3 younger stages' requests, 3 each
            W3_BARRIER();
            W3_SB
#undef W3_MM
#undef W3_RB
#undef W3_RA
#undef W3_SB
            slot_rd = slot_rd + 1 == W3_NS ? 0 : slot_rd + 1;
            slot_wr = slot_wr + 1 == W3_NS ? 0 : slot_wr + 1;
        }
    }
    };
    const int s_plain = max(len / W3_KS - 1, 0) & ~3;          // stages ss < s_plain read a full stage ss + 1
    run(std::false_type{}, 0, s_plain);
    run(std::true_type{}, s_plain, stages4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead requests of the last stages
    __syncthreads();
    // ---- the wave's 64 x 32 tile -> its slab, through a wave-private LDS staging tile (16-byte row segments)
    float* otile = reinterpret_cast<float*>(smem) + wave * (64 * 36);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            otile[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh) * 36 + l31] = acc[i][e] + sml[i][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (wave-private region: the wave's own writes before its reads)
    float* __restrict__ S = q.slabs + (int64_t)z * q.d * q.ld_slab;
    const bool vec = (q.ld_slab & 3) == 0 && (reinterpret_cast<uintptr_t>(S) & 15u) == 0;
    const int c4 = (lane & 7) * 4, r0 = lane >> 3;
    const int colg = n0 + 32 * wn + c4;
    const int N = q.out_dim;
#pragma unroll
    for (int itr = 0; itr < 8; ++itr) {
        const int rr = 8 * itr + r0;
        const int row = m0 + rr;
        const f32x4 v = *reinterpret_cast<const f32x4*>(otile + rr * 36 + c4);
        if (row < q.d) {
            float* dst = S + (int64_t)row * q.ld_slab + colg;
            if (vec && colg + 3 < N) *reinterpret_cast<f32x4*>(dst) = v;
            else {
                if (colg < N) dst[0] = v.x;
                if (colg + 1 < N) dst[1] = v.y;
                if (colg + 2 < N) dst[2] = v.z;
                if (colg + 3 < N) dst[3] = v.w;
            }
        }
    }
#undef W3_BARRIER
#undef W3_INFLIGHT
}

static int wgrad_grouped_tiled3_impl(const gs_wgrad_desc* descs_host, int32_t n_desc, const gs_gather_desc* jobs_host,
                                     int32_t n_jobs, const FanoutArgs* sampler, void* stream) {
    GS_REQUIRE(descs_host && n_desc > 0 && n_desc <= W3_MAXP, "gs_dense_wgrad_grouped_tiled3: 1..%d problems", W3_MAXP);
    Wg3Args G = {};
    G.n = n_desc;
    int64_t items = 0;
    for (int i = 0; i < n_desc; ++i) {
        const gs_wgrad_desc& q = descs_host[i];
        GS_CHECK_MAT(q.A, q.lda, "gs_dense_wgrad_grouped_tiled3 A");
        GS_CHECK_MAT(q.dZ, q.ldz, "gs_dense_wgrad_grouped_tiled3 dZ");
        GS_REQUIRE(q.slabs && q.d > 0 && q.out_dim > 0 && q.n > 0 && q.n_slabs > 0 && q.col0 >= 0 && q.col0 % 4 == 0,
                   "gs_dense_wgrad_grouped_tiled3: bad sizes");
        GS_REQUIRE(q.lda >= q.d && q.ldz >= q.col0 + q.out_dim && q.ld_slab >= q.out_dim, "gs_dense_wgrad_grouped_tiled3: ld too small");
        GS_REQUIRE(q.n < (1ll << 31) - 64 * 1024 && q.lda < (1ll << 31) && q.ldz < (1ll << 31) && (int64_t)q.d * q.ld_slab < (1ll << 31),
                   "gs_dense_wgrad_grouped_tiled3: 32-bit sizes exceeded");
        Wg3Prob& p = G.p[i];
        p.A = q.A; p.a_idx = q.a_idx; p.dZ = q.dZ + q.col0; p.slabs = q.slabs;
        p.lda = (int32_t)q.lda; p.ldz = (int32_t)q.ldz; p.ld_slab = (int32_t)q.ld_slab; p.zcols = (int32_t)(q.ldz - q.col0);
        p.n = (int32_t)q.n; p.d = q.d; p.out_dim = q.out_dim;
        p.tiles_m = (int)gs_ceil_div(q.d, 64);
        p.tiles_n = (int)gs_ceil_div(q.out_dim, 128);
        p.n_slabs = q.n_slabs;
        p.kchunk = (int32_t)(gs_ceil_div(gs_ceil_div(q.n, q.n_slabs), 32) * 32);
        // (every problem: the kernel reads a slice's source rows -- ids or the rows themselves -- from its LDS row list)
        GS_REQUIRE(p.kchunk <= W3_MAXROWS, "gs_dense_wgrad_grouped_tiled3: a slice holds at most %d rows (n = %lld, n_slabs = %d)",
                   W3_MAXROWS, (long long)q.n, q.n_slabs);
        p.item_start = (int32_t)items;
        items += (int64_t)p.tiles_m * p.tiles_n * q.n_slabs;
    }
    GS_REQUIRE(items < (1ll << 30), "gs_dense_wgrad_grouped_tiled3: too many work items");
    G.n_items = (int32_t)items;
    CoGatherS J = {};
    int64_t waves = 0;
    int rc = build_cojobs_s(jobs_host, n_jobs, &J, &waves);
    if (rc != GS_OK) return rc;
    FanoutArgs F = {};
    if (sampler) F = *sampler;
    const int64_t blocks = items + gs_ceil_div(F.B, 4) + gs_ceil_div(waves, 4);
    GS_REQUIRE(blocks < (1ll << 31), "gs_dense_wgrad_grouped_tiled3: grid too large");
    const size_t lds = W3_NS * (W3_KS * 64 * 4 + W3_KS * 128 * 4) + W3_MAXROWS * 4;
    static_assert(4 * GS_FANOUT_LDS_INTS(GS_FANOUT_LDS_SMALL) * 4 <= W3_NS * (W3_KS * 64 * 4 + W3_KS * 128 * 4), "sampler rider LDS");
    GS_LDS_ATTR(lds, wgrad_tiled3_kernel);
    hipLaunchKernelGGL(wgrad_tiled3_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, G, F, J);
    GS_LAUNCH_CHECK("wgrad_tiled3_kernel");
    return GS_OK;
}

extern "C" int gs_dense_wgrad_grouped_tiled3(const gs_wgrad_desc* descs_host, int32_t n_desc, const gs_gather_desc* jobs_host,
                                             int32_t n_jobs, void* stream) {
    return wgrad_grouped_tiled3_impl(descs_host, n_desc, jobs_host, n_jobs, nullptr, stream);
}

// ... with the fan-out sampler of a later mini-batch riding in the launch (gs_fanout_desc as gs_sample_fanout_desc takes it).
// The caller guarantees that nothing this launch READS is written by that sampler: the row ids of a gathered problem must not be
// the id buffer the sampler fills (gs_tail_desc.ids_copy_* makes the private copy the weight gradients read instead).
extern "C" int gs_dense_wgrad_grouped_tiled3_sample(const gs_wgrad_desc* descs_host, int32_t n_desc, const gs_gather_desc* jobs_host,
                                                    int32_t n_jobs, const gs_fanout_desc* sampler_host, void* stream) {
    if (!sampler_host) return wgrad_grouped_tiled3_impl(descs_host, n_desc, jobs_host, n_jobs, nullptr, stream);
    FanoutArgs F = {};
    int64_t kmax = 0;
    if (gs_fanout_args_desc(sampler_host, &F, &kmax) != GS_OK) return GS_EINVAL;
    if (kmax > GS_FANOUT_LDS_SMALL) {
        gs_set_error("gs_dense_wgrad_grouped_tiled3_sample: per-root fan-out %lld of a kept hop exceeds %d", (long long)kmax, GS_FANOUT_LDS_SMALL);
        return GS_ENOTSUP;
    }
    int64_t last = sampler_host->B;                                // the id buffer's extent: [roots | hop 1 | ... | hop n_hops]
    for (int h = 0; h < sampler_host->n_hops; ++h) last *= sampler_host->fan[h];
    const int32_t* lo = sampler_host->ids_all;
    const int32_t* hi = sampler_host->ids_all + sampler_host->offsets[sampler_host->n_hops] + last;
    for (int i = 0; i < n_desc; ++i)
        GS_REQUIRE(!descs_host[i].a_idx || descs_host[i].a_idx + descs_host[i].n <= lo || descs_host[i].a_idx >= hi,
                   "gs_dense_wgrad_grouped_tiled3_sample: problem %d gathers through the id buffer the riding sampler writes", i);
    return wgrad_grouped_tiled3_impl(descs_host, n_desc, jobs_host, n_jobs, &F, stream);
}
