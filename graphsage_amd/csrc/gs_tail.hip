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
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int TT> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef f32x2 type; };
template <> struct VecT<4> { typedef f32x4 type; };

template <int TT>
__device__ __forceinline__ float vec_elem(const typename VecT<TT>::type& v, int t) {
    if constexpr (TT == 1) return v; else return v[t];
}

// acc[t] += A[16 x (k0..k1)] . B[(k0..k1) x cols(t)],  cols(t) = {n0 + TT*j + t, j = 0..15}.  (k1 - k0) % 4 == 0.
// Lanes whose first column n0 + TT*j is >= n_lim contribute zeros (their output columns are garbage-free zeros).
template <int TT>
__device__ __forceinline__ void mm16_nn(const float* __restrict__ A, const int lda, const float* __restrict__ B,
                                        const int64_t ldb, const int n0, const int n_lim, const int k0, const int k1,
                                        f32x4 (&acc)[TT], const int lane) {
    typedef typename VecT<TT>::type V;
    const int j = lane & 15, q = lane >> 4;
    const bool ok = n0 + TT * j < n_lim;
    const float* bp = B + (int64_t)(k0 + q) * ldb + (ok ? n0 + TT * j : 0);
    const float* ap = A + j * lda + k0 + q;
    constexpr int U = 8;
    int k = k0;
    for (; k + 4 * U <= k1; k += 4 * U) {
        V b[U];
        float a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            b[u] = *reinterpret_cast<const V*>(bp + (int64_t)(4 * u) * ldb);
            a[u] = ap[4 * u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float av = ok ? a[u] : 0.f;
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[t] = mfma16(av, vec_elem<TT>(b[u], t), acc[t]);
        }
        bp += (int64_t)(4 * U) * ldb;
        ap += 4 * U;
    }
    for (; k < k1; k += 4) {
        const V b = *reinterpret_cast<const V*>(bp);
        const float av = ok ? ap[0] : 0.f;
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[t] = mfma16(av, vec_elem<TT>(b, t), acc[t]);
        bp += 4 * ldb;
        ap += 4;
    }
}

// acc[t] += A[16 x K] . Bt,  Bt[k][col] = B[n0 + 16t + j][k]  (B row-major [n_rows x >=k_lim], k-contiguous).
// K % 16 == 0 (A zero-padded to K in LDS); weight rows >= n_rows and k-quads starting at >= k_lim read as zeros.
template <int TT>
__device__ __forceinline__ void mm16_nt(const float* __restrict__ A, const int lda, const float* __restrict__ B,
                                        const int64_t ldb, const int n0, const int n_rows, const int K, const int k_lim,
                                        f32x4 (&acc)[TT], const int lane) {
    const int j = lane & 15, q = lane >> 4;
    const float* ap = A + j * lda + 4 * q;
    const float* bp[TT];
    bool rok[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int nrow = n0 + 16 * t + j;
        rok[t] = nrow < n_rows;
        bp[t] = B + (int64_t)(rok[t] ? nrow : 0) * ldb + 4 * q;
    }
#pragma unroll 2
    for (int kb = 0; kb < K; kb += 16) {
        const bool kok = kb + 4 * q < k_lim;
        f32x4 b4[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            b4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rok[t] && kok) b4[t] = *reinterpret_cast<const f32x4*>(bp[t] + kb);
        }
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + kb);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[t] = mfma16(a4[e], b4[t][e], acc[t]);
    }
}

__device__ __forceinline__ float tail_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float tail_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// TZ = Z/128 (tiles per wave of the z / d_y contractions), TD = 2*D/128 (tiles per wave of the input gradients)
template <int TZ, int TD>
__global__ __launch_bounds__(TAIL_THREADS) void sage_tail_kernel(const TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = a.D, O = a.O, Z = 2 * a.O, C = a.C;
    const int ldh = D + 4, ldzs = Z + 4;
    const int Cp4 = (C + 3) & ~3, Cp16 = (C + 15) & ~15;
    const int groups = (C + 63) >> 6;                 // 64-column groups of the logits (1 or 2)
    const int GC = groups * 64;
    const int nks = TAIL_WAVES / groups;              // K-split of the logits contraction
    const int ldc = Cp16 + 4;
    float* Hs = lds;                                  // [16][ldh]   self rows of h0      (later: DIN [16][2D+8])
    float* Ms = Hs + TAIL_ROWS * ldh;                 // [16][ldh]   neighbor means
    float* Zs = Ms + TAIL_ROWS * ldh;                 // [16][ldzs]  z, then y
    float* DZs = Zs + TAIL_ROWS * ldzs;               // [16][ldzs]  dLoss/dz
    float* Ps = DZs + TAIL_ROWS * ldzs;               // [nks][16][GC] logits partials   (later: DY [16][ldzs])
    const int ps_floats = max(nks * TAIL_ROWS * GC, TAIL_ROWS * ldzs);
    float* DLs = Ps + ps_floats;                      // [16][ldc]   dlogits, zero-padded to Cp16
    float* invs = DLs + TAIL_ROWS * ldc;              // [16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * TAIL_ROWS;
    const int64_t n = a.n;
    const int s = a.s;

    // ---------------- phase 0: self rows -> LDS; neighbor means (aggregators.py:48) -> LDS + global
    {
        const int d4 = D >> 2;
        const float inv_s = 1.0f / (float)s;
        for (int it = tid; it < TAIL_ROWS * d4; it += TAIL_THREADS) {
            const int r = it / d4, c = (it - r * d4) * 4;
            const int64_t i = r0 + r;
            f32x4 hs = {0.f, 0.f, 0.f, 0.f}, acc = {0.f, 0.f, 0.f, 0.f};
            if (i < n) {
                hs = *reinterpret_cast<const f32x4*>(a.h0 + i * a.ldh + c);
                const float* nb = a.h0 + (n + i * s) * a.ldh + c;
                int jn = 0;
                for (; jn + 5 <= s; jn += 5) {
                    f32x4 v[5];
#pragma unroll
                    for (int u = 0; u < 5; ++u) v[u] = *reinterpret_cast<const f32x4*>(nb + (int64_t)(jn + u) * a.ldh);
#pragma unroll
                    for (int u = 0; u < 5; ++u) acc += v[u];
                }
                for (; jn < s; ++jn) acc += *reinterpret_cast<const f32x4*>(nb + (int64_t)jn * a.ldh);
                acc *= inv_s;
                *reinterpret_cast<f32x4*>(a.means + i * a.ldm + c) = acc;
            }
            *reinterpret_cast<f32x4*>(Hs + r * ldh + c) = hs;
            *reinterpret_cast<f32x4*>(Ms + r * ldh + c) = acc;
        }
    }
    __syncthreads();

    // ---------------- phase 1: z = [self . W_self | means . W_neigh]   (concat, identity act: last layer)
    {
        const int col0 = wave * 16 * TZ;              // this wave's first output column in [0, Z)
        const int term = col0 >= O ? 1 : 0;
        f32x4 acc[TZ];
#pragma unroll
        for (int t = 0; t < TZ; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        mm16_nn<TZ>(term ? Ms : Hs, ldh, term ? a.Wn : a.Ws, term ? a.ldwn : a.ldws, col0 - term * O, O, 0, D, acc, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * q + i;
#pragma unroll
            for (int t = 0; t < TZ; ++t) {
                const int col = col0 + TZ * j + t;
                Zs[row * ldzs + col] = acc[t][i];
                if (r0 + row < n) a.z[(r0 + row) * a.ldz + col] = acc[t][i];
            }
        }
    }
    __syncthreads();

    // ---------------- phase 2: y = l2_normalize(z)   (supervised_models.py:85); two rows per wave
    for (int rr = 0; rr < 2; ++rr) {
        const int row = wave * 2 + rr;
        float ss = 0.f;
        for (int c = lane; c < Z; c += 64) {
            const float v = Zs[row * ldzs + c];
            ss += v * v;
        }
        ss = tail_wave_sum(ss);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        for (int c = lane; c < Z; c += 64) {
            const float v = Zs[row * ldzs + c] * inv;
            Zs[row * ldzs + c] = v;
            if (r0 + row < n) a.y[(r0 + row) * a.ldy + c] = v;
        }
        if (lane == 0) invs[row] = inv;
    }
    __syncthreads();

    // ---------------- phase 3: logits partials, K split over the waves (fixed-order sum in phase 4)
    {
        const int g = wave % groups, ks = wave / groups;
        const int kchunk = Z / nks;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        mm16_nn<4>(Zs, ldzs, a.Wh, a.ldwh, g * 64, Cp4, ks * kchunk, (ks + 1) * kchunk, acc, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * q + i;
            *reinterpret_cast<f32x4*>(Ps + (ks * TAIL_ROWS + row) * GC + g * 64 + 4 * j) =
                f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
        }
    }
    __syncthreads();

    // ---------------- phase 4: logits, loss rows, preds, dlogits   (supervised_models.py:111-126); two rows per wave
    for (int rr = 0; rr < 2; ++rr) {
        const int row = wave * 2 + rr;
        const int64_t i = r0 + row;
        const bool valid = i < n;
        float x[2], zl[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = lane + 64 * m;
            x[m] = 0.f;
            zl[m] = 0.f;
            if (m < groups && c < C) {
                float v = a.bh ? a.bh[c] : 0.f;
                for (int ks = 0; ks < nks; ++ks) v += Ps[(ks * TAIL_ROWS + row) * GC + c];
                x[m] = v;
                if (valid) zl[m] = a.labels[i * a.ldlab + c];
            }
        }
        float gl[2] = {0.f, 0.f}, pr[2] = {0.f, 0.f};
        float loss = 0.f;
        if (a.sigmoid) {
            const float gscale = 1.0f / ((float)n * (float)C);
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int c = lane + 64 * m;
                if (m < groups && c < C) {
                    acc += fmaxf(x[m], 0.f) - x[m] * zl[m] + log1pf(expf(-fabsf(x[m])));
                    pr[m] = 1.0f / (1.0f + expf(-x[m]));
                    gl[m] = (pr[m] - zl[m]) * gscale;
                }
            }
            loss = tail_wave_sum(acc) / (float)C;
        } else {
            float mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (m < groups && lane + 64 * m < C) mx = fmaxf(mx, x[m]);
            mx = tail_wave_max(mx);
            float se = 0.f, zs = 0.f, zx = 0.f;
            float ex[2] = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (m < groups && lane + 64 * m < C) {
                    ex[m] = expf(x[m] - mx);
                    se += ex[m];
                    zs += zl[m];
                    zx += zl[m] * x[m];
                }
            se = tail_wave_sum(se);
            zs = tail_wave_sum(zs);
            zx = tail_wave_sum(zx);
            const float lse = mx + logf(se), inv_se = 1.0f / se, gscale = 1.0f / (float)n;
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (m < groups && lane + 64 * m < C) {
                    pr[m] = ex[m] * inv_se;
                    gl[m] = (pr[m] * zs - zl[m]) * gscale;
                }
            loss = zs * lse - zx;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = lane + 64 * m;
            if (m < groups) {
                const float gv = (valid && c < C) ? gl[m] : 0.f;
                if (c < Cp16) DLs[row * ldc + c] = gv;
                if (valid && c < Cp4) {
                    const bool in = c < C;
                    if (a.logits) a.logits[i * a.ldlo + c] = in ? x[m] : 0.f;
                    if (a.preds) a.preds[i * a.ldp + c] = in ? pr[m] : 0.f;
                    a.dlogits[i * a.lddl + c] = gv;
                }
            }
        }
        if (valid && lane == 0) a.loss_rows[i] = loss;
    }
    if (!a.train) {
        if (blockIdx.x == 0 && tid == 0) {
            if (a.c0) *a.c0 += a.d0;
            if (a.c1) *a.c1 += a.d1;
            if (a.c2) *a.c2 += a.d2;
        }
        return;
    }
    __syncthreads();

    // ---------------- phase 5: d_y = dlogits . W_head^T   -> DY (aliases the logits partials)
    float* DYs = Ps;
    {
        const int n0 = wave * 16 * TZ;
        f32x4 acc[TZ];
#pragma unroll
        for (int t = 0; t < TZ; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        mm16_nt<TZ>(DLs, ldc, a.Wh, a.ldwh, n0, Z, Cp16, Cp4, acc, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < TZ; ++t) DYs[(4 * q + i) * ldzs + n0 + 16 * t + j] = acc[t][i];
    }
    __syncthreads();

    // ---------------- phase 6: d_z = l2_normalize'(d_y)   (two rows per wave)
    for (int rr = 0; rr < 2; ++rr) {
        const int row = wave * 2 + rr;
        float dot = 0.f;
        for (int c = lane; c < Z; c += 64) dot += DYs[row * ldzs + c] * Zs[row * ldzs + c];
        dot = tail_wave_sum(dot);
        const float inv = invs[row];
        const bool clamped = inv >= 1.0e6f;   // sum(z^2) < 1e-12: y = z * 1e6, no normalisation term
        for (int c = lane; c < Z; c += 64) {
            const float dyv = DYs[row * ldzs + c];
            const float g = clamped ? dyv * inv : inv * (dyv - Zs[row * ldzs + c] * dot);
            DZs[row * ldzs + c] = g;
            if (r0 + row < n) a.dz[(r0 + row) * a.lddz + c] = g;
        }
    }
    __syncthreads();

    // ---------------- phase 7: [d_self | d_means] = [d_z[:, :O] . W_self^T | d_z[:, O:] . W_neigh^T]  -> DIN
    float* DIN = Hs;                                   // [16][2D + 8]   (Hs | Ms are dead since phase 1)
    const int ldi = 2 * D + 8;
    {
        const int col0 = wave * 16 * TD;               // in [0, 2D)
        const int term = col0 >= D ? 1 : 0;
        f32x4 acc[TD];
#pragma unroll
        for (int t = 0; t < TD; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        mm16_nt<TD>(DZs + term * O, ldzs, term ? a.Wn : a.Ws, term ? a.ldwn : a.ldws, col0 - term * D, D, O, O, acc, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < TD; ++t) DIN[(4 * q + i) * ldi + col0 + 16 * t + j] = acc[t][i];
    }
    __syncthreads();

    // ---------------- phase 8: d_h0 = relu'(h0) * (d_self on the self row, d_means / s on each of the s neighbor rows)
    {
        const int d4 = D >> 2;
        const float inv_s = 1.0f / (float)s;
        const int per_row = (1 + s) * d4;
        for (int it = tid; it < TAIL_ROWS * per_row; it += TAIL_THREADS) {
            const int r = it / per_row;
            const int rem = it - r * per_row;
            const int jn = rem / d4, c = (rem - jn * d4) * 4;
            const int64_t i = r0 + r;
            if (i >= n) continue;
            const int64_t tgt = jn == 0 ? i : n + i * s + (jn - 1);
            f32x4 g = *reinterpret_cast<const f32x4*>(DIN + r * ldi + (jn == 0 ? 0 : D) + c);
            if (jn != 0) g *= inv_s;
            const f32x4 h = *reinterpret_cast<const f32x4*>(a.h0 + tgt * a.ldh + c);
            g.x = h.x > 0.f ? g.x : 0.f;
            g.y = h.y > 0.f ? g.y : 0.f;
            g.z = h.z > 0.f ? g.z : 0.f;
            g.w = h.w > 0.f ? g.w : 0.f;
            *reinterpret_cast<f32x4*>(a.d_h0 + tgt * a.lddh + c) = g;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {                // device counters (sampler clock / epoch cursor / optimizer step)
        if (a.c0) *a.c0 += a.d0;
        if (a.c1) *a.c1 += a.d1;
        if (a.c2) *a.c2 += a.d2;
    }
}

static size_t tail_lds_bytes(int D, int O, int C) {
    const int Z = 2 * O, ldh = D + 4, ldzs = Z + 4;
    const int Cp16 = (C + 15) & ~15, groups = (C + 63) >> 6, GC = groups * 64, nks = TAIL_WAVES / groups;
    const int ps = std::max(nks * TAIL_ROWS * GC, TAIL_ROWS * ldzs);
    const size_t floats = (size_t)2 * TAIL_ROWS * ldh + (size_t)2 * TAIL_ROWS * ldzs + ps + (size_t)TAIL_ROWS * (Cp16 + 4) + TAIL_ROWS;
    return floats * sizeof(float);
}

extern "C" int gs_sage_tail_supported(int32_t d_in, int32_t out_dim, int32_t C) {
    const bool ok = (d_in == 128 || d_in == 256) && (out_dim == 64 || out_dim == 128) && C >= 1 && C <= 128 &&
                    tail_lds_bytes(d_in, out_dim, C) <= 160 * 1024;
    return ok ? 1 : 0;
}

template <int TZ, int TD>
static int launch_tail(const TailArgs& a, hipStream_t st) {
    const size_t lds = tail_lds_bytes(a.D, a.O, a.C);
    static bool attr_done = false;
    if (!attr_done) {
        GS_HIP(hipFuncSetAttribute((const void*)sage_tail_kernel<TZ, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const unsigned blocks = (unsigned)gs_ceil_div(a.n, TAIL_ROWS);
    hipLaunchKernelGGL((sage_tail_kernel<TZ, TD>), dim3(blocks), dim3(TAIL_THREADS), lds, st, a);
    GS_LAUNCH_CHECK("sage_tail_kernel");
    return GS_OK;
}

extern "C" int gs_sage_tail_fwd_bwd(const gs_tail_desc* q, void* stream) {
    GS_REQUIRE(q, "gs_sage_tail_fwd_bwd: null descriptor");
    if (q->n == 0) return GS_OK;
    GS_REQUIRE(q->n > 0 && q->s > 0, "gs_sage_tail_fwd_bwd: bad sizes");
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
    hipStream_t st = (hipStream_t)stream;
    // TZ = Z/128 in {1, 2};  TD = 2*D/128 in {2, 4}
    if (Z == 256 && D == 256) return launch_tail<2, 4>(a, st);
    if (Z == 256 && D == 128) return launch_tail<2, 2>(a, st);
    if (Z == 128 && D == 256) return launch_tail<1, 4>(a, st);
    return launch_tail<1, 2>(a, st);
}
