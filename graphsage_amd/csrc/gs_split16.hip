// The pooling MLP of the max-pool / mean-pool aggregator (aggregators.py:176-179 via layers.py:104-116) on the fp16 matrix pipe
// with fp32 accuracy: every fp32 operand is cut ONCE into two fp16 pieces
//     x * 2^e = h + m + r,   h = fp16(x 2^e),  m = fp16(x 2^e - h)      (round to nearest: |r| <= 2^-23 |x 2^e|, one fp32 ulp)
// -- with a power-of-two scale per feature-table ROW and per weight COLUMN that puts the row's / column's largest element at
// 2^13..2^14, so that h never overflows and m (2^-12 of h) stays a normal fp16 for every element within 2^-15 of the largest --
// and a product of two fp32 numbers is h h' + h m' + m h': three exact 11 x 11 bit products accumulated in fp32, scaled back by
// 2^-(e_row + e_col) in the epilogue (exact).  What is given up per product: the dropped m m' (|m| <= 2^-11 |x|: <= 2^-22 |x y|
// worst case, 2^-24 rms, random sign) and the operands' last bit (above).  In a K-term dot product these add up like a random
// walk -- sqrt(K) 2^-23 against the K-term sum -- one to two orders of magnitude BELOW the rounding of the fp32 accumulation
// itself, which is why the outputs measure as accurate against fp64 as the three-piece bf16 kernel's of gs_split.hip (no operand
// bit lost, six products of 8 x 8 bits) and more accurate than an fp32 FMA chain (tests/test_split_gemm_gpu.py prints all three:
// 1.3e-7 | 1.0e-7 | 1.7e-7 of |x|.|w| at K = 602) -- at HALF the matrix-pipe work of the three-piece form.
//
// Why (profiles/r05_pool_clock_probe.txt): the three-piece kernel is POWER-bound -- the chip sits at its 1400 W cap with the
// engine clock pulled down to 2.02 GHz while it runs (a lone workgroup takes 38 us, the same workgroup 64 us when all 256 CUs
// compute), and ~310 G bf16 MACs per launch are most of that energy.  Half the MFMAs is the only large term left to remove.
//
// Both operands are cut OUTSIDE the contraction: the weights behind every optimizer launch (as before), the FEATURE TABLE once --
// it is a constant input (models.py:299: features are a non-trainable tf.Variable), 2 x 2 bytes per element = the bytes of the
// fp32 table -- so the kernel's A and B tiles are plain copies global -> LDS and its VALU work is the epilogue.
#include "gs_common.h"
#include <stdlib.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 gs_mfma_f16(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// scale exponent e of a row / column whose largest magnitude is mx: mx 2^e in [2^13, 2^14); 0 for an all-zero or non-finite one
__device__ __forceinline__ int gs_scale_exp(const float mx) {
    const uint32_t b = __float_as_uint(mx);
    const int ex = (int)((b >> 23) & 0xFFu);
    if (ex == 0xFF || mx == 0.f) return 0;
    return 13 - (ex - 127);                                    // (a denormal maximum counts as 2^-127: its pieces just use fewer bits)
}
__device__ __forceinline__ void gs_cut16(const float x, const int e, _Float16& h, _Float16& m) {
    const float xs = ldexpf(x, e);
    h = (_Float16)xs;
    m = (_Float16)(xs - (float)h);
}

static inline int split16_stages2(int32_t K) { return (((K + 31) / 32) + 1) & ~1; }
static inline int split16_kp(int32_t K) { return 32 * split16_stages2(K); }          // halfs per plane of a table row: whole stages, zero beyond K

// ------------------------------------------------------------------------------------------------ W -> W2
// W [K, N] fp32 -> W2 [KP/8 groups of 8 k][2 pieces][N][8] fp16 (per group and piece the N columns side by side, 16 bytes each: the 32
// lanes of a B-fragment load read 512 contiguous bytes) followed by the N column exponents (int32).
// Workgroups of 16 waves per (64 columns, quarter of the k-groups) (lane = column: coalesced rows of W): the waves split K for the
// column maxima (LDS; every quarter repeats them -- W is L2-resident), then the quarter's groups of 8 k for the cut: ~40 + 10
// independent loads per thread.  (A first version ran ONE thread per column over the whole K twice: 130 us per call for the
// 602 x 512 weights, behind every optimizer launch; one workgroup per 64 columns: 16 us.)
#define S16_ROWS_WAVES 16
#define S16_ROWS_KSPLIT 4
__global__ __launch_bounds__(64 * S16_ROWS_WAVES) void split16_rows_kernel(const float* __restrict__ W, int64_t ldw, int32_t K, int32_t N,
                                                                           int32_t KP, _Float16* __restrict__ W2, int32_t* __restrict__ cexp) {
    __shared__ float smax[S16_ROWS_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = (blockIdx.x / S16_ROWS_KSPLIT) * 64 + lane;
    const int kq = blockIdx.x % S16_ROWS_KSPLIT;
    const int nc = min(n, N - 1);                              // (a column past the end: loads clamped, nothing stored)
    float mx = 0.f;
#pragma unroll 8
    for (int k = wave; k < K; k += S16_ROWS_WAVES) mx = fmaxf(mx, fabsf(W[(int64_t)k * ldw + nc]));
    smax[wave][lane] = mx;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < S16_ROWS_WAVES; ++w) mx = fmaxf(mx, smax[w][lane]);
    const int e = gs_scale_exp(mx);
    if (wave == 0 && kq == 0 && n < N) cexp[n] = e;
    for (int kg = kq * S16_ROWS_WAVES + wave; kg < KP / 8; kg += S16_ROWS_WAVES * S16_ROWS_KSPLIT) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = W[(int64_t)min(8 * kg + j, K - 1) * ldw + nc];
        f16x8 h, m;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 hh = (_Float16)0.f, mm = (_Float16)0.f;
            gs_cut16(8 * kg + j < K ? v[j] : 0.f, e, hh, mm);
            h[j] = hh; m[j] = mm;
        }
        if (n < N) {
            f16x8* dst = reinterpret_cast<f16x8*>(W2) + ((int64_t)kg * 2) * N + n;
            dst[0] = h;
            dst[N] = m;
        }
    }
}

extern "C" int gs_split_rows_f16_bytes(int32_t K, int32_t N, int64_t* bytes_out_host) {
    GS_REQUIRE(K > 0 && N > 0 && bytes_out_host, "gs_split_rows_f16_bytes: bad args");
    *bytes_out_host = (int64_t)(split16_kp(K) / 8) * 2 * N * 16 + (int64_t)N * 4;
    return GS_OK;
}

extern "C" int gs_split_rows_f16(const float* W, int64_t ldw, int32_t K, int32_t N, void* W2, void* stream) {
    GS_REQUIRE(W && W2 && K > 0 && N > 0 && ldw >= N, "gs_split_rows_f16: bad args");
    GS_REQUIRE(gs_aligned16(W2), "gs_split_rows_f16: W2 must be 16-byte aligned");
    const int KP = split16_kp(K);
    int32_t* cexp = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(W2) + (int64_t)(KP / 8) * 2 * N * 16);
    hipLaunchKernelGGL(split16_rows_kernel, dim3((unsigned)(gs_ceil_div(N, 64) * S16_ROWS_KSPLIT)), dim3(64 * S16_ROWS_WAVES), 0, (hipStream_t)stream, W, ldw, K, N, KP,
                       (_Float16*)W2, cexp);
    GS_LAUNCH_CHECK("split16_rows_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------------------------------------ X -> X2
// X [rows, ldx] fp32 -> X2 [rows][2 pieces][KP] fp16 (zero beyond d) + rexp [rows] (int32).  One wave per row.
__global__ __launch_bounds__(256) void split16_table_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int32_t d, int32_t KP,
                                                            _Float16* __restrict__ X2, int32_t* __restrict__ rexp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = X + row * ldx;
    float mx = 0.f;
    for (int k = lane; k < d; k += 64) mx = fmaxf(mx, fabsf(x[k]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const int e = gs_scale_exp(mx);
    if (lane == 0) rexp[row] = e;
    _Float16* h = X2 + row * 2 * (int64_t)KP;
    _Float16* m = h + KP;
    for (int k = lane; k < KP; k += 64) {
        _Float16 hh = (_Float16)0.f, mm = (_Float16)0.f;
        if (k < d) gs_cut16(x[k], e, hh, mm);
        h[k] = hh; m[k] = mm;
    }
}

extern "C" int gs_split_table_f16_bytes(int64_t rows, int32_t d, int64_t* table_bytes_out_host, int64_t* exp_bytes_out_host) {
    GS_REQUIRE(rows > 0 && d > 0 && table_bytes_out_host && exp_bytes_out_host, "gs_split_table_f16_bytes: bad args");
    *table_bytes_out_host = rows * 2 * (int64_t)split16_kp(d) * 2;
    *exp_bytes_out_host = rows * 4;
    return GS_OK;
}

extern "C" int gs_split_table_f16(const float* X, int64_t ldx, int64_t rows, int32_t d, void* X2, int32_t* rexp, void* stream) {
    GS_REQUIRE(X && X2 && rexp && rows > 0 && d > 0 && ldx >= d, "gs_split_table_f16: bad args");
    GS_REQUIRE(gs_aligned16(X2), "gs_split_table_f16: X2 must be 16-byte aligned");
    GS_REQUIRE(rows < (1ll << 31) * 4, "gs_split_table_f16: too many rows");
    hipLaunchKernelGGL(split16_table_kernel, dim3((unsigned)gs_ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, X, ldx, rows, d,
                       split16_kp(d), (_Float16*)X2, rexp);
    GS_LAUNCH_CHECK("split16_table_kernel");
    return GS_OK;
}

// ------------------------------------------------------------------------------------------------ the contraction
// out[i] = act(X[idx[i]] . W + bias), i < min(n_max, *n_dev), from the pre-cut table and weights.
//   workgroup (8 waves) = 128 rows x 256 columns, wave = 64 x 64 = four MFMA tiles with an h-h and a small-terms accumulator each
//   (as gs_split.hip's wide form; the reason for the second accumulator is there); stage = 32 k: every thread copies 2 x 16 bytes of
//   one table row (h and m of 8 k) and 4 x 16 bytes of W2 into LDS (A rows at an 80-byte stride: conflict-free ds_read_b128
//   fragments); 24 MFMAs and 16 fragment reads per wave and stage; two LDS buffers, one barrier per stage.
//   One workgroup per CU (104 KB of LDS), equal work per workgroup: the tiles of the last, incomplete round are cut along K into
//   parts (WideSchedule of gs_split.hip, restated here) whose partial tiles split16_fixup_kernel sums in part order.
#define S16_LDA 40                     // halfs per LDS row of an A plane: 32 + 8 pad
struct Split16Args {
    const _Float16* X2; const int32_t* rexp; const int32_t* idx; const u32x4* W2; const int32_t* cexp; const float* bias; float* out;
    const int32_t* n_dev;
    int64_t ldo;
    int32_t n_max, K, KP, N, act;
    float* ws;            // partial tiles of the split-K tail round (nullable: no split), [<= n_cu][128][256]
    int32_t n_cu;
    int32_t units;        // what a tail-round tile is cut into: 32-k stages (DMA form) or stage PAIRS (register-staged form)
};
struct Sched16 { int full, rem, S; };
__device__ __host__ __forceinline__ Sched16 split16_schedule(const int nwg, const int units, const int n_cu, const bool have_ws) {
    Sched16 w = {nwg, 0, 1};
    if (have_ws && n_cu > 0) {
        const int r = nwg % n_cu;
        if (r > 0 && 2 * r <= n_cu) {
            const int pairs = units;
            int sp = pairs < n_cu / r ? pairs : n_cu / r;
            if (sp > 10) sp = 10;                               // (the fix-up launch keeps a tile's parts in registers: <= 10)
            if (sp >= 2) { w.full = nwg - r; w.rem = r; w.S = sp; }
        }
    }
    return w;
}

__global__ __launch_bounds__(512) void split16_tiled_fwd_kernel(const Split16Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_PLANE = 128 * S16_LDA * 2;                // bytes
    constexpr int A_BYTES = 2 * A_PLANE, B_BYTES = 8 * 256 * 16, BUF = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int count = min(g.n_max, g.n_dev ? *g.n_dev : g.n_max);
    const int tiles_n = (g.N + 255) >> 8, tiles_m = (count + 127) >> 7;
    const int nwg = tiles_m * tiles_n;
    const int N = g.N, KP = g.KP;
    const int stages2 = KP >> 5;
    const Sched16 sch = split16_schedule(nwg, g.units, g.n_cu, g.ws != nullptr);
    if ((int)blockIdx.x >= sch.full + sch.rem * sch.S) return;
    int tile, part = -1;                                       // part >= 0: a K part of a tail-round tile (partial tile -> g.ws)
    if ((int)blockIdx.x < sch.full) {
        // XCD-aware: block b runs on XCD b % 8; consecutive LOGICAL tiles (the column tiles of one row tile) share an XCD's L2
        const int q8 = sch.full >> 3, r8 = sch.full & 7, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    } else {
        const int r = (int)blockIdx.x - sch.full;
        tile = sch.full + r / sch.S;
        part = r - (r / sch.S) * sch.S;
    }
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 256;
    // ---- global -> LDS roles (512 threads)
    const int arow = tid >> 2, aq = tid & 3;                   // A: row of the tile, 8-k quarter of the stage
    const int grow = min(m0 + arow, count - 1);
    const int64_t srow = g.idx ? (int64_t)g.idx[grow] : (int64_t)grow;
    const _Float16* __restrict__ xrow = g.X2 + srow * 2 * (int64_t)KP + 8 * aq;
    const int bcol = tid & 255, bch = wave >> 2;               // B: column of the tile, chunks bch, bch + 2, bch + 4, bch + 6 (8 per stage)
    const int bc = min(n0 + bcol, N - 1);
    const char* __restrict__ W2b = (const char*)g.W2;
    const uint32_t bcol_off = (uint32_t)bc * 16u, plane_b = (uint32_t)N * 16u;
    u32x4 ra[2][2];                                            // two register sets for A (stage s + 2 is requested while s computes)
    u32x4 rb[4];
    // (S16_DIAG_*: diagnostics builds only, benchmarks/probes/build_variant.sh -- wrong values, the kernel's time without one component)
    auto gload_a = [&](const int set, const int s) {           // (past the end: the last stage again -- valid addresses, unused values)
#ifdef S16_DIAG_NOGLOAD
        ra[set][0] = ra[set][1] = u32x4{(unsigned)s, 1u, 2u, (unsigned)tid};
        return;
#endif
        const int so = 32 * min(s, stages2 - 1);
        ra[set][0] = *reinterpret_cast<const u32x4*>(xrow + so);
        ra[set][1] = *reinterpret_cast<const u32x4*>(xrow + KP + so);
    };
    auto gload_b = [&](const int s) {
#ifdef S16_DIAG_NOGLOAD
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[j] = u32x4{(unsigned)s, 1u, 2u, (unsigned)tid};
        return;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = bch + 2 * j;                         // chunk = (k-group of the stage) * 2 + piece: W2's own order
            const uint32_t off = (uint32_t)(8 * min(s, stages2 - 1) + c) * plane_b + bcol_off;
            rb[j] = *reinterpret_cast<const u32x4*>(W2b + off);
        }
    };
    auto lds_store_a = [&](const int set, unsigned char* buf) {
#ifdef S16_DIAG_NOLDSW
        asm volatile("" :: "v"(ra[set][0]), "v"(ra[set][1]));
        return;
#endif
        unsigned char* pa = buf + (arow * S16_LDA + 8 * aq) * 2;
        *reinterpret_cast<u32x4*>(pa) = ra[set][0];
        *reinterpret_cast<u32x4*>(pa + A_PLANE) = ra[set][1];
    };
    auto lds_store_b = [&](unsigned char* buf) {
#ifdef S16_DIAG_NOLDSW
        asm volatile("" :: "v"(rb[0]), "v"(rb[1]), "v"(rb[2]), "v"(rb[3]));
        return;
#endif
        unsigned char* pb = buf + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(pb + ((bch + 2 * j) * 256 + bcol) * 16) = rb[j];
    };
    const int wm = wave >> 2, wn = wave & 3;                   // 8 waves = 2 (rows of 64) x 4 (columns of 64)
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
            u32x4 fa[2][2], fb[2][2];
#ifdef S16_DIAG_NOLDSR
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    fa[i][p] = u32x4{(unsigned)q, (unsigned)(uintptr_t)buf, 2u, (unsigned)tid};
                    fb[i][p] = u32x4{(unsigned)p, (unsigned)(uintptr_t)buf, 3u, (unsigned)tid};
                }
#else
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    fa[i][p] = *reinterpret_cast<const u32x4*>(buf + p * A_PLANE + ((64 * wm + 32 * i + l31) * S16_LDA + 16 * q + 8 * lh) * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    fb[j][p] = *reinterpret_cast<const u32x4*>(buf + A_BYTES + (((2 * q + lh) * 2 + p) * 256 + 64 * wn + 32 * j + l31) * 16);
#endif
#ifdef S16_DIAG_NOMFMA
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int p = 0; p < 2; ++p) asm volatile("" :: "v"(fa[i][p]), "v"(fb[i][p]));
            continue;
#endif
            // piece product outermost, the four tiles innermost: no MFMA accumulates into the result of the one issued before it
#define GS_PP(dst, pa, pb) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
                               dst[i][j] = gs_mfma_f16(fa[i][pa], fb[j][pb], dst[i][j]);
            GS_PP(sml, 0, 1)     // h m
            GS_PP(sml, 1, 0)     // m h
            GS_PP(acc, 0, 0)     // h h
#undef GS_PP
        }
    };
    int s_begin = 0, s_end = stages2;
    if (part >= 0) {
        const int pairs = stages2 >> 1;
        s_begin = 2 * ((pairs * part) / sch.S);
        s_end = 2 * ((pairs * (part + 1)) / sch.S);
    }
    gload_a(0, s_begin);
    gload_b(s_begin);
    gload_a(1, s_begin + 1);
    lds_store_a(0, smem);
    lds_store_b(smem);
    __syncthreads();
    for (int s = s_begin; s < s_end; s += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int ss = s + par;
            unsigned char* cur = smem + par * BUF;
            unsigned char* nxt = smem + (par ^ 1) * BUF;
            gload_b(ss + 1);                                   // (vmcnt retires in order: B of stage s + 1 before the younger A of s + 2)
            gload_a(par, ss + 2);
            compute(cur);
            lds_store_a(par ^ 1, nxt);
            lds_store_b(nxt);
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA
                if (q < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);         // two LDS reads
                else __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);               // an LDS write
            }
            __syncthreads();
        }
    }
    // scale back, bias + activation, through a wave-private LDS region (free after the last barrier) so that a lane stores 16
    // contiguous bytes of a row; a K part of a tail-round tile stores its raw partial sums instead
    float* otile = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    const bool partial = part >= 0;
    float* wtile = partial ? g.ws + (int64_t)((int)blockIdx.x - sch.full) * (128 * 256) : nullptr;
    const int c4 = (lane & 15) * 4, r0 = lane >> 4;
    const int colg = n0 + 64 * wn + c4;
    int ce[4] = {0, 0, 0, 0};
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (!partial) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ce[u] = g.cexp[min(colg + u, N - 1)];
        if (g.bias) {
            if (colg < N) bv.x = g.bias[colg];
            if (colg + 1 < N) bv.y = g.bias[colg + 1];
            if (colg + 2 < N) bv.z = g.bias[colg + 2];
            if (colg + 3 < N) bv.w = g.bias[colg + 3];
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                otile[((e & 3) + 8 * (e >> 2) + 4 * lh) * 68 + 32 * j + l31] = acc[i][j][e] + sml[i][j][e];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = 4 * it + r0;
            const int row = m0 + 64 * wm + 32 * i + r;
            f32x4 v = *reinterpret_cast<const f32x4*>(otile + r * 68 + c4);
            if (partial) {
                *reinterpret_cast<f32x4*>(wtile + (64 * wm + 32 * i + r) * 256 + 64 * wn + c4) = v;
            } else if (row < count) {
                const int re = g.rexp[g.idx ? g.idx[row] : row];
                v.x = ldexpf(v.x, -(re + ce[0])) + bv.x;
                v.y = ldexpf(v.y, -(re + ce[1])) + bv.y;
                v.z = ldexpf(v.z, -(re + ce[2])) + bv.z;
                v.w = ldexpf(v.w, -(re + ce[3])) + bv.w;
                if (g.act == GS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
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

// LDS-DMA form (the default): the same tiles, arithmetic and summation order as split16_tiled_fwd_kernel -- bit-identical outputs --
// but NOTHING on the way global -> LDS lives in a VGPR.  Measured on the register-staged form (-D variants, gpurun_out/
// s16_variants.txt, 83,000 rows): MFMAs + barriers alone 155 us, + fragment reads 169, + LDS writes 184, + global loads 268 --
// a W2 chunk requested at the top of a stage is written to LDS at its end, one stage (~0.6 us) of latency cover, and six
// ds_write_b128 per thread and stage go through a write path two SIMDs share.  Here both operand tiles are global_load_lds_dwordx4
// (one wave instruction = 1 KB at a wave-uniform LDS address + 16 lane) into a ring of THREE stage buffers: stage s + 2 is
// requested at the top of stage s, `s_waitcnt vmcnt(6)` at its end says "stage s + 1 has landed" (six DMA per thread and stage,
// retired in order) and leaves stage s + 2 in flight across the raw s_barrier (__syncthreads would drain it with vmcnt(0)).
//   A plane in LDS: [128 rows][4 chunks of 8 k] without padding (the DMA writes lane-linear); row r holds chunk c at position
//   c ^ ((r >> 2) & 3) -- the permutation is applied to the SOURCE address of the DMA and again to the fragment reads (the same
//   involution): the 16 lanes of a ds_read_b128 pass touch all 64 banks once.
#define S16_NBUF 3
#ifdef S16_TIMELINE
// Diagnostics build only (-DS16_TIMELINE, benchmarks/timeline_split16.py): shader-clock stamps of wave 0 of the first 256 workgroups.
//   [0] entry  [1] first stages requested  [2] first barrier passed  [3] K loop left  [4] epilogue done
//   [8 + 4 s + 0..3] stage s: first half issued | vmcnt / lgkmcnt wait over | barrier passed | second half issued
__device__ unsigned long long g_s16_tl[256 * 128];
extern "C" int gs_debug_s16_timeline(unsigned long long* out_host, int n) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_s16_tl), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#define S16_STAMP(k) do { if (tid == 0 && blockIdx.x < 256 && (k) < 128) g_s16_tl[blockIdx.x * 128 + (k)] = clock64(); } while (0)
#else
#define S16_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(512) void split16_dma_fwd_kernel(const Split16Args g) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_PLANE = 128 * 64;                          // bytes
    constexpr int A_BYTES = 2 * A_PLANE, B_BYTES = 8 * 256 * 16, BUF = A_BYTES + B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    S16_STAMP(0);
    const int count = min(g.n_max, g.n_dev ? *g.n_dev : g.n_max);
    const int tiles_n = (g.N + 255) >> 8, tiles_m = (count + 127) >> 7;
    const int nwg = tiles_m * tiles_n;
    const int N = g.N, KP = g.KP;
    const int stages = (g.K + 31) >> 5;                        // (the table's rows are padded to an EVEN stage count for the register-
    const Sched16 sch = split16_schedule(nwg, g.units, g.n_cu, g.ws != nullptr);   //  staged form; this one stops at the last stage that holds a k < K)
    if ((int)blockIdx.x >= sch.full + sch.rem * sch.S) return;
    int tile, part = -1;
    if ((int)blockIdx.x < sch.full) {
        const int q8 = sch.full >> 3, r8 = sch.full & 7, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
    } else {
        const int r = (int)blockIdx.x - sch.full;
        tile = sch.full + r / sch.S;
        part = r - (r / sch.S) * sch.S;
    }
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 256;
    // ---- DMA roles: thread (row tid >> 2, position tid & 3) of the A planes; column tid & 255, chunks (wave >> 2) + 2 j of B
    const int arow = tid >> 2;
    const int akc = (tid & 3) ^ ((arow >> 2) & 3);             // the 8-k chunk that lands at this thread's position
    const int grow = min(m0 + arow, count - 1);
    const int wm = wave >> 2, wn = wave & 3;
    // The scale exponents of the 16 output rows this thread will store (epilogue: rows 64 wm + 32 i + 4 it + (lane >> 4)), requested
    // HERE: their ids in one round trip with the A row's id, their exponents in the next, both long landed when the epilogue needs
    // them.  (Loaded inside the epilogue -- behind its `row < count` branches -- every one of the 16 was two dependent round trips
    // of its own: 19.8 k of a workgroup's 68.5 k cycles, benchmarks/timeline_split16.py.)
    int erow[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rr = min(m0 + 64 * wm + 32 * i + 4 * it + (lane >> 4), count - 1);
            erow[i][it] = g.idx ? g.idx[rr] : rr;
        }
    const int64_t srow = g.idx ? (int64_t)g.idx[grow] : (int64_t)grow;
    int rexp_r[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int it = 0; it < 8; ++it) rexp_r[i][it] = g.rexp[erow[i][it]];
    const _Float16* __restrict__ xrow = g.X2 + srow * 2 * (int64_t)KP + 8 * akc;
    const int bch = wave >> 2;
    const int bc = min(n0 + (tid & 255), N - 1);
    const char* __restrict__ W2b = (const char*)g.W2 + (uint32_t)bc * 16u;
    const uint32_t plane_b = (uint32_t)N * 16u;
    const int a_dst = wave * 1024, b_dst = A_BYTES + (wave & 3) * 1024;     // wave-uniform LDS offsets (+ 16 lane by the hardware)
    auto issue = [&](const int s, unsigned char* buf) {        // six DMA per thread (past the end: the last stage again, unused)
#ifdef S16_DIAG_NODMA
        if (s > 1) return;
#endif
        const int sc = min(s, stages - 1);
        __builtin_amdgcn_global_load_lds(xrow + 32 * sc, (lds_ptr_t)(buf + a_dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(xrow + KP + 32 * sc, (lds_ptr_t)(buf + A_PLANE + a_dst), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = bch + 2 * j;
            __builtin_amdgcn_global_load_lds(W2b + (uint32_t)(8 * sc + c) * plane_b, (lds_ptr_t)(buf + b_dst + c * 4096), 16, 0, 0);
        }
    };
    const int sw = (l31 >> 2) & 3;
    const int a_rd = (64 * wm + l31) * 64;                     // + 32 i rows, + plane, + ((2 q + lh) ^ sw) * 16
    const int a_q0 = ((0 + lh) ^ sw) * 16, a_q1 = ((2 + lh) ^ sw) * 16;
    const int b_rd = A_BYTES + (64 * wn + l31) * 16 + lh * 2 * 4096;        // + (2 q) groups, + piece, + 32 j columns
    f32x16 acc[2][2], sml[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; sml[i][j][e] = 0.f; }
    // One stage of a wave = 24 MFMAs in a PINNED order (a sched_barrier behind every one): left to the scheduler, the MFMAs of one
    // accumulator end up back to back (runs of four and five in the first build of this kernel) and each waits for the result of
    // the one before it.  Order per 16-k half: (h m) x 4 tiles, (h h) x 4 tiles, (m h) x 4 tiles -- the same accumulator comes up
    // again four MFMAs later at the earliest.
    // The stage's barrier sits in its MIDDLE: the fragment reads of the second half ride between the MFMAs of the first, then
    // "stage s + 1 has landed" (vmcnt) + "my reads of this stage's buffer are done" (lgkmcnt) + s_barrier, then the second half's
    // MFMAs carry the reads of stage s + 1's FIRST half -- no wave starts a stage by waiting for its fragments (with the barrier at
    // the end of the stage both waves of a SIMD did, at the same time: the matrix pipe idled through an LDS round trip per stage).
    u32x4 fa[2][2][2], fb[2][2][2];                            // [half q][i | j][piece]
    auto rd_a = [&](const unsigned char* buf, const int q, const int i, const int p) {
#ifdef S16_DIAG_NOLDSR2
        fa[q][i][p] = u32x4{(unsigned)q, (unsigned)(uintptr_t)buf, 2u, (unsigned)tid}; return;
#endif
        fa[q][i][p] = *reinterpret_cast<const u32x4*>(buf + p * A_PLANE + a_rd + i * (32 * 64) + (q ? a_q1 : a_q0));
    };
    auto rd_b = [&](const unsigned char* buf, const int q, const int j, const int p) {
#ifdef S16_DIAG_NOLDSR2
        fb[q][j][p] = u32x4{(unsigned)p, (unsigned)(uintptr_t)buf, 3u, (unsigned)tid}; return;
#endif
        fb[q][j][p] = *reinterpret_cast<const u32x4*>(buf + b_rd + (4 * q + p) * 4096 + j * (32 * 16));
    };
#define S16_SB __builtin_amdgcn_sched_barrier(0);
#ifdef S16_DIAG_NOMFMA2
#define S16_MM(dst, q, i, j, pa, pb) asm volatile("" :: "v"(fa[q][i][pa]), "v"(fb[q][j][pb])); S16_SB
#else
#define S16_MM(dst, q, i, j, pa, pb) dst[i][j] = gs_mfma_f16(fa[q][i][pa], fb[q][j][pb], dst[i][j]); S16_SB
#endif
    int s_begin = 0, s_end = stages;
    if (part >= 0) {
        s_begin = (stages * part) / sch.S;
        s_end = (stages * (part + 1)) / sch.S;
    }
    issue(s_begin, smem);
    issue(s_begin + 1, smem + BUF);                            // (a one-stage part: the stage behind it, landed and never read)
    S16_STAMP(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    S16_STAMP(2);
    rd_a(smem, 0, 0, 0); rd_b(smem, 0, 0, 1); rd_b(smem, 0, 1, 1); rd_a(smem, 0, 1, 0);
    rd_b(smem, 0, 0, 0); rd_b(smem, 0, 1, 0); rd_a(smem, 0, 0, 1); rd_a(smem, 0, 1, 1);
    int slot = 0;                                              // ring slot of stage s
#pragma unroll 1
    for (int s = s_begin; s < s_end; ++s) {
        const int slot1 = slot == 2 ? 0 : slot + 1;
        const int slot2 = slot == 0 ? 2 : slot - 1;            // (slot + 2) % 3: last read before the barrier of stage s - 1
        const unsigned char* cur = smem + slot * BUF;
        const unsigned char* nxt = smem + slot1 * BUF;
        const bool ahead = s + 2 < s_end;                      // nothing is requested past the end of the tile / part
        if (ahead) issue(s + 2, smem + slot2 * BUF);
        S16_SB
        S16_MM(sml, 0, 0, 0, 0, 1) rd_a(cur, 1, 0, 0); S16_SB
        S16_MM(sml, 0, 0, 1, 0, 1) rd_a(cur, 1, 1, 0); S16_SB
        S16_MM(sml, 0, 1, 0, 0, 1) rd_b(cur, 1, 0, 1); S16_SB
        S16_MM(sml, 0, 1, 1, 0, 1) rd_b(cur, 1, 1, 1); S16_SB
        S16_MM(acc, 0, 0, 0, 0, 0) rd_b(cur, 1, 0, 0); S16_SB
        S16_MM(acc, 0, 0, 1, 0, 0) rd_b(cur, 1, 1, 0); S16_SB
        S16_MM(acc, 0, 1, 0, 0, 0) rd_a(cur, 1, 0, 1); S16_SB
        S16_MM(acc, 0, 1, 1, 0, 0) rd_a(cur, 1, 1, 1); S16_SB
        S16_MM(sml, 0, 0, 0, 1, 0) S16_MM(sml, 0, 0, 1, 1, 0) S16_MM(sml, 0, 1, 0, 1, 0) S16_MM(sml, 0, 1, 1, 1, 0)
        S16_STAMP(8 + 4 * (s - s_begin));
        if (ahead) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // stage s + 1 has landed (mine); my reads of `cur` are done
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // (nothing younger in flight in the last two stages)
        S16_STAMP(9 + 4 * (s - s_begin));
        __builtin_amdgcn_s_barrier();                          // ... everybody's: `nxt` is readable, `cur` may be overwritten (stage s + 3)
        asm volatile("" ::: "memory");
        S16_STAMP(10 + 4 * (s - s_begin));
        S16_SB
        S16_MM(sml, 1, 0, 0, 0, 1) rd_a(nxt, 0, 0, 0); S16_SB
        S16_MM(sml, 1, 0, 1, 0, 1) rd_b(nxt, 0, 0, 1); S16_SB
        S16_MM(sml, 1, 1, 0, 0, 1) rd_b(nxt, 0, 1, 1); S16_SB
        S16_MM(sml, 1, 1, 1, 0, 1) rd_a(nxt, 0, 1, 0); S16_SB
        S16_MM(acc, 1, 0, 0, 0, 0) rd_b(nxt, 0, 0, 0); S16_SB
        S16_MM(acc, 1, 0, 1, 0, 0) rd_b(nxt, 0, 1, 0); S16_SB
        S16_MM(acc, 1, 1, 0, 0, 0) rd_a(nxt, 0, 0, 1); S16_SB
        S16_MM(acc, 1, 1, 1, 0, 0) rd_a(nxt, 0, 1, 1); S16_SB
        S16_MM(sml, 1, 0, 0, 1, 0) S16_MM(sml, 1, 0, 1, 1, 0) S16_MM(sml, 1, 1, 0, 1, 0) S16_MM(sml, 1, 1, 1, 1, 0)
        S16_STAMP(11 + 4 * (s - s_begin));
        slot = slot1;
    }
    S16_STAMP(3);
#undef S16_MM
#undef S16_SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the look-ahead requests of the last two stages
    __syncthreads();
#ifdef S16_DIAG_NOEPI
    if (acc[0][0][0] + sml[1][1][3] + acc[1][0][5] + sml[0][1][7] != 12345.678f) return;
#endif
    // ---- epilogue: as split16_tiled_fwd_kernel
    float* otile = reinterpret_cast<float*>(smem) + wave * (32 * 68);
    const bool partial = part >= 0;
    float* wtile = partial ? g.ws + (int64_t)((int)blockIdx.x - sch.full) * (128 * 256) : nullptr;
    const int c4 = (lane & 15) * 4, r0 = lane >> 4;
    const int colg = n0 + 64 * wn + c4;
    int ce[4] = {0, 0, 0, 0};
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (!partial) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ce[u] = g.cexp[min(colg + u, N - 1)];
        if (g.bias) {
            if (colg < N) bv.x = g.bias[colg];
            if (colg + 1 < N) bv.y = g.bias[colg + 1];
            if (colg + 2 < N) bv.z = g.bias[colg + 2];
            if (colg + 3 < N) bv.w = g.bias[colg + 3];
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                otile[((e & 3) + 8 * (e >> 2) + 4 * lh) * 68 + 32 * j + l31] = acc[i][j][e] + sml[i][j][e];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = 4 * it + r0;
            const int row = m0 + 64 * wm + 32 * i + r;
            f32x4 v = *reinterpret_cast<const f32x4*>(otile + r * 68 + c4);
            if (partial) {
                *reinterpret_cast<f32x4*>(wtile + (64 * wm + 32 * i + r) * 256 + 64 * wn + c4) = v;
            } else if (row < count) {
                const int re = rexp_r[i][it];
                v.x = ldexpf(v.x, -(re + ce[0])) + bv.x;
                v.y = ldexpf(v.y, -(re + ce[1])) + bv.y;
                v.z = ldexpf(v.z, -(re + ce[2])) + bv.z;
                v.w = ldexpf(v.w, -(re + ce[3])) + bv.w;
                if (g.act == GS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                float* dst = g.out + (int64_t)row * g.ldo + colg;
#ifdef S16_NT_STORE
                if (colg + 3 < N) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
#else
                if (colg + 3 < N) *reinterpret_cast<f32x4*>(dst) = v;
#endif
                else {
                    if (colg < N) dst[0] = v.x;
                    if (colg + 1 < N) dst[1] = v.y;
                    if (colg + 2 < N) dst[2] = v.z;
                }
            }
        }
    }
    S16_STAMP(4);
}

// out tile = act(2^-(e_row + e_col) * (sum of the S partial tiles of a tail-round tile, in part order) + bias): one workgroup per
// (tile, 32-row band)
__global__ __launch_bounds__(256) void split16_fixup_kernel(const Split16Args g) {
    // one workgroup per (tail tile, four rows); a thread owns 16 bytes of ONE row: its S partial values are independent loads
    // (a first version walked eight rows and the S parts in a serial loop per thread: 24 us of load latency for 26 MB)
    const int count = min(g.n_max, g.n_dev ? *g.n_dev : g.n_max);
    const int tiles_n = (g.N + 255) >> 8, tiles_m = (count + 127) >> 7;
    const int nwg = tiles_m * tiles_n;
    const Sched16 sch = split16_schedule(nwg, g.units, g.n_cu, g.ws != nullptr);
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
    int ce[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ce[u] = g.cexp[min(colg + u, g.N - 1)];
    const int re = g.rexp[g.idx ? g.idx[row] : row];
    const float* base = g.ws + (int64_t)t * sch.S * (128 * 256) + r * 256 + c4;
    f32x4 pv[10];                                               // S <= 10 parts (stage pairs of K <= 640); summed in part order
#pragma unroll
    for (int p = 0; p < 10; ++p) pv[p] = *reinterpret_cast<const f32x4*>(base + (int64_t)min(p, sch.S - 1) * (128 * 256));
    f32x4 v = pv[0];
#pragma unroll
    for (int p = 1; p < 10; ++p) if (p < sch.S) v += pv[p];
    v.x = ldexpf(v.x, -(re + ce[0])) + bv.x;
    v.y = ldexpf(v.y, -(re + ce[1])) + bv.y;
    v.z = ldexpf(v.z, -(re + ce[2])) + bv.z;
    v.w = ldexpf(v.w, -(re + ce[3])) + bv.w;
    if (g.act == GS_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    float* dst = g.out + (int64_t)row * g.ldo + colg;
    if (colg + 3 < g.N) *reinterpret_cast<f32x4*>(dst) = v;
    else {
        if (colg < g.N) dst[0] = v.x;
        if (colg + 1 < g.N) dst[1] = v.y;
        if (colg + 2 < g.N) dst[2] = v.z;
    }
}

extern "C" int gs_dense_fwd_rows_split16(const void* X2, const int32_t* rexp, const int32_t* idx, int32_t d, int64_t n_max,
                                         const int32_t* n_dev, const void* W2, int32_t out_dim, int act, const float* bias, float* out,
                                         int64_t ldo, float* ws, int64_t ws_bytes, void* stream) {
    if (n_max == 0) return GS_OK;
    GS_REQUIRE(X2 && rexp && W2 && out && d > 0 && out_dim > 0 && n_max > 0 && n_max < (1ll << 30), "gs_dense_fwd_rows_split16: bad args");
    GS_REQUIRE(gs_aligned16(X2) && gs_aligned16(W2), "gs_dense_fwd_rows_split16: X2 / W2 must be 16-byte aligned");
    GS_REQUIRE(ldo >= out_dim && ldo % 4 == 0 && gs_aligned16(out), "gs_dense_fwd_rows_split16: out must be 16-byte aligned with ldo % 4 == 0");
    const int KP = split16_kp(d);
    GS_REQUIRE((int64_t)(KP / 8) * 2 * out_dim * 16 < (1ll << 32), "gs_dense_fwd_rows_split16: W2 must stay below 4 GB");
    Split16Args g = {};
    g.X2 = (const _Float16*)X2; g.rexp = rexp; g.idx = idx; g.W2 = (const u32x4*)W2;
    g.cexp = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(W2) + (int64_t)(KP / 8) * 2 * out_dim * 16);
    g.bias = bias; g.out = out; g.n_dev = n_dev; g.ldo = ldo;
    g.n_max = (int32_t)n_max; g.K = d; g.KP = KP; g.N = out_dim; g.act = act;
    int64_t blocks = gs_ceil_div(n_max, 128) * gs_ceil_div(out_dim, 256);
    static const bool dma = !(getenv("GS_SPLIT16_DMA") && atoi(getenv("GS_SPLIT16_DMA")) == 0);       // 0: the register-staged form (A/B)
    const size_t lds = dma ? (size_t)S16_NBUF * (2 * 128 * 64 + 8 * 256 * 16) : (size_t)2 * (2 * 128 * S16_LDA * 2 + 8 * 256 * 16);
    static bool attr_set = false;
    static int n_cu = 0;
    GS_LDS_ATTR(S16_NBUF * (2 * 128 * 64 + 8 * 256 * 16), split16_dma_fwd_kernel);
    GS_LDS_ATTR(2 * (2 * 128 * S16_LDA * 2 + 8 * 256 * 16), split16_tiled_fwd_kernel);
    if (!attr_set) {
        int dev = 0;
        hipDeviceProp_t prop;
        GS_HIP(hipGetDevice(&dev));
        GS_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    static const bool no_tail = getenv("GS_SPLIT_WIDE_TAIL") && atoi(getenv("GS_SPLIT_WIDE_TAIL")) == 0;     // A/B hook
    const bool tail = ws && !no_tail && n_cu > 0 && ws_bytes >= (int64_t)n_cu * 128 * 256 * (int64_t)sizeof(float);
    GS_REQUIRE(!ws || gs_aligned16(ws), "gs_dense_fwd_rows_split16: the workspace must be 16-byte aligned");
    g.units = dma ? (d + 31) / 32 : split16_stages2(d) / 2;
    if (tail) { g.ws = ws; g.n_cu = n_cu; blocks += n_cu; }
    if (dma) hipLaunchKernelGGL(split16_dma_fwd_kernel, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(split16_tiled_fwd_kernel, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, g);
    GS_LAUNCH_CHECK("split16 forward kernel");
    if (tail) {
        hipLaunchKernelGGL(split16_fixup_kernel, dim3((unsigned)(16 * n_cu)), dim3(256), 0, (hipStream_t)stream, g);
        GS_LAUNCH_CHECK("split16_fixup_kernel");
    }
    return GS_OK;
}
