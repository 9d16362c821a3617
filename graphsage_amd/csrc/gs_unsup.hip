// N3: unsupervised objective kernels (models.py:332-405, prediction.py:68-110).
//   unsup_stage_kernel     edge-pair batch selection (minibatch.py:113-132 on the device) + the 20 negative samples
//                          of tf.nn.fixed_unigram_candidate_sampler(distortion=0.75, unique=False) (models.py:336-343)
//   linkpred_fwd_bwd_kernel  BipartiteEdgePredLayer xent loss (prediction.py:102-110), MRR ranks (models.py:393-405)
//                          and the gradients w.r.t. the three groups of (l2-normalised) embeddings, one wave per pair.
#include "gs_common.h"


// ids_out = [batch1 (B) | batch2 (B) | negatives (n_neg)].  pairs: int32 [n_pairs, 2] (may be NULL: roots already
// staged by the host).  cdf: uint32 [n_nodes], cdf[i] = floor(2^32 * P(node <= i)) with P ~ degree^0.75 (last = 2^32-1);
// a negative is the first node whose cdf exceeds a 32-bit draw (binary search) -- bit-exact vs oracle/sampler_hash.py.
__global__ __launch_bounds__(256) void unsup_stage_kernel(const int32_t* __restrict__ pairs, int64_t n_pairs,
                                                          const uint64_t* __restrict__ cursor, int64_t B,
                                                          const uint32_t* __restrict__ cdf, int64_t n_nodes, int32_t n_neg,
                                                          uint64_t seed, const uint64_t* __restrict__ clock,
                                                          int64_t slot_offset, int32_t* __restrict__ ids_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pairs && t < B) {
        const uint64_t c = cursor ? *cursor : 0ull;
        const int64_t e = (int64_t)((c + (uint64_t)t) % (uint64_t)n_pairs);
        ids_out[t] = pairs[2 * e];
        ids_out[B + t] = pairs[2 * e + 1];
    }
    if (cdf && t < n_neg) {
        const uint64_t st = clock ? *clock : 0ull;
        const uint64_t key = gs_mix64(seed ^ (st * 0x9E3779B97F4A7C15ull) ^ (0xFFull << 56));
        // keyed by the GLOBAL slot like the fused fan-out staging (gs_sample_dev.h): data-parallel ranks draw different
        // negatives and the stand-alone and the fused staging of the same step agree
        const uint32_t r = (uint32_t)(gs_mix64(key + (uint64_t)t + (uint64_t)slot_offset) >> 32);
        int64_t lo = 0, hi = n_nodes - 1;  // first index with cdf[idx] > r
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (cdf[mid] > r) hi = mid; else lo = mid + 1;
        }
        ids_out[2 * B + t] = (int32_t)lo;
    }
}

extern "C" int gs_unsup_stage(const int32_t* pairs, int64_t n_pairs, const uint64_t* cursor_dev, int64_t B,
                              const uint32_t* cdf, int64_t n_nodes, int32_t n_neg, uint64_t seed,
                              const uint64_t* clock_dev, int64_t slot_offset, int32_t* ids_out, void* stream) {
    GS_REQUIRE(ids_out && B >= 0 && n_neg >= 0, "gs_unsup_stage: bad args");
    GS_REQUIRE(!pairs || n_pairs > 0, "gs_unsup_stage: empty pair list");
    GS_REQUIRE(!cdf || n_nodes > 0, "gs_unsup_stage: empty cdf");
    const int64_t n = std::max<int64_t>(pairs ? B : 0, cdf ? n_neg : 0);
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(unsup_stage_kernel, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, pairs,
                       n_pairs, cursor_dev, B, cdf, n_nodes, n_neg, seed, clock_dev, slot_offset, ids_out);
    GS_LAUNCH_CHECK("unsup_stage_kernel");
    return GS_OK;
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One wave = one pair against all negatives.  o1 / o2: the pair's (normalised) rows, one 64-column group per register.
// The n_neg affinities are formed 64 at a time with lane q holding negative q's, so that the sigmoid / softplus of the
// whole block is ONE round of v_exp / v_log / v_rcp (the first version walked the negatives one by one with libm expf /
// log1pf: 21 us for 512 pairs x 20 negatives); __expf / __logf are ~1e-6 relative like the supervised tail's.
//   g1 += sum_q gq * neg_q;   mypart[q] = gq * o1   (the negatives' gradient contribution of this pair)
template <int DJ>
__device__ __forceinline__ void linkpred_pair(const float (&o1)[DJ], const float* __restrict__ negs, float* __restrict__ mypart,
                                              const int n_neg, const float aff, const float neg_w, const float scale,
                                              const bool live, const int lane, float (&g1)[DJ], float& loss, int& rank,
                                              float* __restrict__ aff_row) {
    constexpr int d = DJ * 64;
    for (int qb = 0; qb < n_neg; qb += 64) {
        const int nq = min(64, n_neg - qb);                       // wave-uniform
        float nav = 0.f;
        int q = 0;
        for (; q + 4 <= nq; q += 4) {                             // four independent dot products / reductions in flight
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            const float* nr = negs + (size_t)(qb + q) * d + lane;
#pragma unroll
            for (int j = 0; j < DJ; ++j) {
                p0 += o1[j] * nr[j * 64];
                p1 += o1[j] * nr[d + j * 64];
                p2 += o1[j] * nr[2 * d + j * 64];
                p3 += o1[j] * nr[3 * d + j * 64];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                p0 += __shfl_xor(p0, off, 64);
                p1 += __shfl_xor(p1, off, 64);
                p2 += __shfl_xor(p2, off, 64);
                p3 += __shfl_xor(p3, off, 64);
            }
            nav = lane == q ? p0 : nav;
            nav = lane == q + 1 ? p1 : nav;
            nav = lane == q + 2 ? p2 : nav;
            nav = lane == q + 3 ? p3 : nav;
        }
        for (; q < nq; ++q) {
            float p0 = 0.f;
#pragma unroll
            for (int j = 0; j < DJ; ++j) p0 += o1[j] * negs[(size_t)(qb + q) * d + j * 64 + lane];
            p0 = wsum(p0);
            nav = lane == q ? p0 : nav;
        }
        // lane q: negative qb + q
        const bool in = lane < nq;
        const float e = __expf(-fabsf(nav));
        const float r1 = __builtin_amdgcn_rcpf(1.0f + e);
        const float sg = nav >= 0.f ? r1 : e * r1;                // sigmoid(nav)
        loss += neg_w * wsum(in ? fmaxf(nav, 0.f) + __logf(1.0f + e) : 0.f);
        rank += __popcll(__ballot(in && nav >= aff));
        const float gqv = (in && live) ? neg_w * scale * sg : 0.f;
        if (aff_row && live && in) aff_row[qb + lane] = nav;
        for (q = 0; q < nq; ++q) {
            const float gq = __shfl(gqv, q, 64);
            const float* nr = negs + (size_t)(qb + q) * d + lane;
            float* mp = mypart + (size_t)(qb + q) * d + lane;
#pragma unroll
            for (int j = 0; j < DJ; ++j) {
                g1[j] += gq * nr[j * 64];
                mp[j * 64] = gq * o1[j];
            }
        }
    }
}

// Y [2B + n_neg, d]: rows [0,B) = outputs1, [B,2B) = outputs2, [2B, 2B+n_neg) = neg_outputs (all l2-normalised).
// Per pair i:  aff = <o1,o2>;  neg_aff_j = <o1, neg_j>;
//   loss_i = xent(1, aff) + w * sum_j xent(0, neg_aff_j)                                  (prediction.py:102-110)
//   rr_i   = 1 / (1 + #{j : neg_aff_j >= aff})                                           (models.py:399-404)
//   dY[i] = scale*((sig(aff)-1)*o2 + sum_j w*sig(neg_aff_j)*neg_j);  dY[B+i] = scale*(sig(aff)-1)*o1
//   dneg partial of this workgroup (4 pairs): slab[blk][j] = scale * sum_i w*sig(neg_aff_ij) * o1_i
template <int DJ>
__global__ __launch_bounds__(256) void linkpred_fwd_bwd_kernel(const float* __restrict__ Y, int64_t ldy, int64_t B,
                                                               int32_t n_neg, float neg_w, float scale,
                                                               float* __restrict__ loss_rows, float* __restrict__ rr_rows,
                                                               float* __restrict__ aff_all, int64_t ld_aff,
                                                               float* __restrict__ dY, int64_t lddy,
                                                               float* __restrict__ neg_slabs) {
    constexpr int d = DJ * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* negs = lds;                       // [n_neg][d]
    float* part = lds + (size_t)n_neg * d;   // [4 waves][n_neg][d] partial dneg
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < n_neg * d; t += 256) negs[t] = Y[(2 * B + t / d) * ldy + (t % d)];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    const bool live = i < B;
    const int64_t ic = live ? i : 0;
    float o1[DJ], o2[DJ], g1[DJ];
    float aff = 0.f;
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        o1[j] = Y[ic * ldy + j * 64 + lane];
        o2[j] = Y[(B + ic) * ldy + j * 64 + lane];
        aff += o1[j] * o2[j];
    }
    aff = wsum(aff);
    const float ea = __expf(-fabsf(aff));
    const float ra = __builtin_amdgcn_rcpf(1.0f + ea);
    const float sa = aff >= 0.f ? ra : ea * ra;
    const float da = (sa - 1.0f) * scale;
    float loss = fmaxf(aff, 0.f) - aff + __logf(1.0f + ea);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < DJ; ++j) g1[j] = da * o2[j];
    float* mypart = part + (size_t)wave * n_neg * d;
    linkpred_pair<DJ>(o1, negs, mypart, n_neg, aff, neg_w, scale, live, lane, g1, loss, rank,
                      (aff_all && live) ? aff_all + i * ld_aff : nullptr);
    if (live) {
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            dY[i * lddy + j * 64 + lane] = g1[j];
            dY[(B + i) * lddy + j * 64 + lane] = da * o1[j];
        }
        if (lane == 0) {
            loss_rows[i] = loss;
            rr_rows[i] = 1.0f / (float)(rank + 1);
            if (aff_all) aff_all[i * ld_aff + n_neg] = aff;
        }
    }
    __syncthreads();
    float* slab = neg_slabs + (size_t)blockIdx.x * n_neg * d;
    for (int t = tid; t < n_neg * d; t += 256)
        slab[t] = (part[t] + part[(size_t)n_neg * d + t]) + (part[2 * (size_t)n_neg * d + t] + part[3 * (size_t)n_neg * d + t]);
}

extern "C" int gs_linkpred_fwd_bwd(const float* Y, int64_t ldy, int64_t B, int32_t d, int32_t n_neg, float neg_weight,
                                   float scale, float* loss_rows, float* rr_rows, float* aff_all, int64_t ld_aff,
                                   float* dY, int64_t lddy, float* neg_slabs, int32_t* n_slabs_out, void* stream) {
    GS_REQUIRE(Y && loss_rows && rr_rows && dY && neg_slabs && B > 0 && n_neg > 0, "gs_linkpred_fwd_bwd: bad args");
    GS_REQUIRE(d == 64 || d == 128 || d == 256 || d == 512, "gs_linkpred_fwd_bwd: d must be 64/128/256/512 (got %d)", d);
    GS_REQUIRE(ldy >= d && lddy >= d && (!aff_all || ld_aff >= n_neg + 1), "gs_linkpred_fwd_bwd: ld too small");
    const size_t lds_bytes = (size_t)5 * n_neg * d * sizeof(float);
    GS_REQUIRE(lds_bytes <= 160 * 1024, "gs_linkpred_fwd_bwd: %d negatives x d=%d do not fit LDS", n_neg, d);
    const int64_t blocks = gs_ceil_div(B, 4);
    if (n_slabs_out) *n_slabs_out = (int32_t)blocks;
    hipStream_t st = (hipStream_t)stream;
#define GS_LP(DJ)                                                                                                         \
    do {                                                                                                                   \
        GS_LDS_ATTR(160 * 1024, linkpred_fwd_bwd_kernel<DJ>);                                                              \
        hipLaunchKernelGGL((linkpred_fwd_bwd_kernel<DJ>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, Y, ldy, B, n_neg, \
                           neg_weight, scale, loss_rows, rr_rows, aff_all, ld_aff, dY, lddy, neg_slabs);                  \
    } while (0)
    if (d == 64) GS_LP(1); else if (d == 128) GS_LP(2); else if (d == 256) GS_LP(4); else GS_LP(8);
#undef GS_LP
    GS_LAUNCH_CHECK("linkpred_fwd_bwd_kernel");
    return GS_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused unsupervised head (one launch instead of l2norm_fwd | linkpred | l2norm_bwd | sum):
//   Y = l2_normalize(Z) (models.py:368-370), the xent link-prediction loss + MRR of linkpred_fwd_bwd_kernel on Y, and
//   the gradient carried back THROUGH the normalisation to Z for the 2B pair rows:
//     dZ = inv * (g - y <g, y>)          (inv = rsqrt(max(sum z^2, 1e-12)); clamped rows: dZ = g * inv)
//   The negatives' gradient w.r.t. their NORMALISED rows leaves as per-workgroup slabs; linkpred_neg_bwd_kernel sums
//   them in a fixed order and applies the same normalisation backward (20 rows).  Also writes mean-reduction inputs
//   loss_rows / rr_rows (finalised by gs_finalize_step2).
template <int DJ>
__global__ __launch_bounds__(256) void linkpred_norm_fwd_bwd_kernel(const float* __restrict__ Z, int64_t ldz, int64_t B,
                                                                    int32_t n_neg, float neg_w, float scale,
                                                                    float* __restrict__ Y, int64_t ldy,
                                                                    float* __restrict__ loss_rows, float* __restrict__ rr_rows,
                                                                    float* __restrict__ aff_all, int64_t ld_aff,
                                                                    float* __restrict__ dZ, int64_t lddz,
                                                                    float* __restrict__ neg_slabs) {
    constexpr int d = DJ * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* negs = lds;                       // [n_neg][d]  normalised negative rows
    float* part = lds + (size_t)n_neg * d;   // [4 waves][n_neg][d] partial dneg (w.r.t. the normalised rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = wave; q < n_neg; q += 4) {
        float v[DJ], ss = 0.f;
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            v[j] = Z[(2 * B + q) * ldz + j * 64 + lane];
            ss += v[j] * v[j];
        }
        const float inv = __builtin_amdgcn_rsqf(fmaxf(wsum(ss), 1e-12f));
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const float y = v[j] * inv;
            negs[q * d + j * 64 + lane] = y;
            if (blockIdx.x == 0) Y[(2 * B + q) * ldy + j * 64 + lane] = y;
        }
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    const bool live = i < B;
    const int64_t ic = live ? i : 0;
    float o1[DJ], o2[DJ], g1[DJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        o1[j] = Z[ic * ldz + j * 64 + lane];
        o2[j] = Z[(B + ic) * ldz + j * 64 + lane];
        s1 += o1[j] * o1[j];
        s2 += o2[j] * o2[j];
    }
    const float inv1 = __builtin_amdgcn_rsqf(fmaxf(wsum(s1), 1e-12f)), inv2 = __builtin_amdgcn_rsqf(fmaxf(wsum(s2), 1e-12f));
    float aff = 0.f;
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        o1[j] *= inv1;
        o2[j] *= inv2;
        aff += o1[j] * o2[j];
    }
    aff = wsum(aff);
    const float ea = __expf(-fabsf(aff));
    const float ra = __builtin_amdgcn_rcpf(1.0f + ea);
    const float sa = aff >= 0.f ? ra : ea * ra;
    const float da = (sa - 1.0f) * scale;
    float loss = fmaxf(aff, 0.f) - aff + __logf(1.0f + ea);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < DJ; ++j) g1[j] = da * o2[j];
    float* mypart = part + (size_t)wave * n_neg * d;
    linkpred_pair<DJ>(o1, negs, mypart, n_neg, aff, neg_w, scale, live, lane, g1, loss, rank,
                      (aff_all && live) ? aff_all + i * ld_aff : nullptr);
    if (live) {
        // back through y = z * inv:  dz = inv (g - y <g, y>);  clamped (sum z^2 < 1e-12, inv = 1e6): dz = g * inv
        float dot1 = 0.f, dot2 = 0.f;
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            dot1 += g1[j] * o1[j];
            dot2 += da * o1[j] * o2[j];
        }
        dot1 = wsum(dot1);
        dot2 = wsum(dot2);
        const bool c1 = inv1 >= 1.0e6f, c2 = inv2 >= 1.0e6f;
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const float ga = g1[j], gb = da * o1[j];
            Y[i * ldy + j * 64 + lane] = o1[j];
            Y[(B + i) * ldy + j * 64 + lane] = o2[j];
            dZ[i * lddz + j * 64 + lane] = c1 ? ga * inv1 : inv1 * (ga - o1[j] * dot1);
            dZ[(B + i) * lddz + j * 64 + lane] = c2 ? gb * inv2 : inv2 * (gb - o2[j] * dot2);
        }
        if (lane == 0) {
            loss_rows[i] = loss;
            rr_rows[i] = 1.0f / (float)(rank + 1);
            if (aff_all) aff_all[i * ld_aff + n_neg] = aff;
        }
    }
    __syncthreads();
    float* slab = neg_slabs + (size_t)blockIdx.x * n_neg * d;
    for (int t = tid; t < n_neg * d; t += 256)
        slab[t] = (part[t] + part[(size_t)n_neg * d + t]) + (part[2 * (size_t)n_neg * d + t] + part[3 * (size_t)n_neg * d + t]);
}

// One workgroup per negative row q: g = sum of the n_slabs per-workgroup slabs in a fixed order, then the normalisation
// backward with the row's own inv (recomputed from Z) and y.  The row's d/4 float4 columns x SG = 1024/d slab groups are
// spread over the 256 threads; a group walks its slabs (sg, sg + SG, ...) with 16 independent 16-byte loads in flight
// (the first version: 8 dword loads per batch, 16 dependent round trips for 128 slabs -- 8 us of pure latency), the
// groups meet in LDS and are summed group 0, 1, ...  Block n_neg (when has_epi) is the step epilogue (mean loss / mrr +
// device counters): it only needs the rows the previous launch wrote.
__global__ __launch_bounds__(256) void linkpred_neg_bwd_kernel(const float* __restrict__ slabs, int32_t n_slabs, int32_t n_neg,
                                                               int32_t d, const float* __restrict__ Z, int64_t ldz,
                                                               int64_t row0, float* __restrict__ dZ, int64_t lddz,
                                                               const StepEpilogue epi) {
    __shared__ f32x4 gpart[256];
    __shared__ float red[2][4];
    if ((int)blockIdx.x == n_neg) {
        gs_step_epilogue_block(epi, red[0], red[1]);
        return;
    }
    const int q = blockIdx.x, tid = threadIdx.x;
    const int d4 = d >> 2, SG = 256 / d4;                  // d in {64, 128, 256, 512}: d4 in {16 .. 128}, SG in {16 .. 2}
    const int cg = tid % d4, sg = tid / d4;
    const f32x4* sp = reinterpret_cast<const f32x4*>(slabs + (size_t)q * d) + cg;
    const size_t stride4 = (size_t)n_neg * d4;             // float4 per slab
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4] = {zero4, zero4, zero4, zero4};
    int sI = sg;
    for (; sI + 15 * SG < n_slabs; sI += 16 * SG) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sp[(size_t)(sI + u * SG) * stride4];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] += v[u];
    }
    for (; sI < n_slabs; sI += SG) acc[0] += sp[(size_t)sI * stride4];
    gpart[tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    f32x4 g = zero4, z = zero4;
    float ss = 0.f;
    if (tid < d4) {
        g = gpart[tid];
        for (int k = 1; k < SG; ++k) g += gpart[k * d4 + tid];
        z = *reinterpret_cast<const f32x4*>(Z + (row0 + q) * ldz + 4 * tid);
        ss = (z.x * z.x + z.y * z.y) + (z.z * z.z + z.w * z.w);
    }
    ss = wsum(ss);
    if ((tid & 63) == 0) red[0][tid >> 6] = ss;
    __syncthreads();
    ss = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-12f));
    float dot = ((g.x * z.x + g.y * z.y) + (g.z * z.z + g.w * z.w)) * inv;
    dot = wsum(dot);
    if ((tid & 63) == 0) red[1][tid >> 6] = dot;
    __syncthreads();
    dot = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const bool clamped = inv >= 1.0e6f;
    if (tid < d4) {
        const f32x4 o = clamped ? g * inv : (g - z * (inv * dot)) * inv;
        *reinterpret_cast<f32x4*>(dZ + (row0 + q) * lddz + 4 * tid) = o;
    }
}

static int linkpred_norm_launch(const float* Z, int64_t ldz, int64_t B, int32_t d, int32_t n_neg, float neg_weight,
                                float scale, float* Y, int64_t ldy, float* loss_rows, float* rr_rows, float* aff_all,
                                int64_t ld_aff, float* dZ, int64_t lddz, float* neg_slabs, const StepEpilogue* epi, void* stream) {
    GS_REQUIRE(Z && Y && loss_rows && rr_rows && dZ && neg_slabs && B > 0 && n_neg > 0, "gs_linkpred_norm_fwd_bwd: bad args");
    GS_REQUIRE(d == 64 || d == 128 || d == 256 || d == 512, "gs_linkpred_norm_fwd_bwd: d must be 64/128/256/512 (got %d)", d);
    GS_REQUIRE(ldz >= d && ldy >= d && lddz >= d && (!aff_all || ld_aff >= n_neg + 1), "gs_linkpred_norm_fwd_bwd: ld too small");
    const size_t lds_bytes = (size_t)5 * n_neg * d * sizeof(float);
    GS_REQUIRE(lds_bytes <= 160 * 1024, "gs_linkpred_norm_fwd_bwd: %d negatives x d=%d do not fit LDS", n_neg, d);
    const int64_t blocks = gs_ceil_div(B, 4);
    hipStream_t st = (hipStream_t)stream;
#define GS_LPN(DJ)                                                                                                        \
    do {                                                                                                                   \
        GS_LDS_ATTR(160 * 1024, linkpred_norm_fwd_bwd_kernel<DJ>);                                                         \
        hipLaunchKernelGGL((linkpred_norm_fwd_bwd_kernel<DJ>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, Z, ldz, B,  \
                           n_neg, neg_weight, scale, Y, ldy, loss_rows, rr_rows, aff_all, ld_aff, dZ, lddz, neg_slabs);   \
    } while (0)
    if (d == 64) GS_LPN(1); else if (d == 128) GS_LPN(2); else if (d == 256) GS_LPN(4); else GS_LPN(8);
#undef GS_LPN
    GS_LAUNCH_CHECK("linkpred_norm_fwd_bwd_kernel");
    GS_REQUIRE(ldz % 4 == 0 && lddz % 4 == 0, "gs_linkpred_norm_fwd_bwd: ldz / lddz must be multiples of 4");
    const StepEpilogue none = {};
    hipLaunchKernelGGL(linkpred_neg_bwd_kernel, dim3((unsigned)(n_neg + (epi ? 1 : 0))), dim3(256), 0, st, neg_slabs,
                       (int32_t)blocks, n_neg, d, Z, ldz, 2 * B, dZ, lddz, epi ? *epi : none);
    GS_LAUNCH_CHECK("linkpred_neg_bwd_kernel");
    return GS_OK;
}

extern "C" int gs_linkpred_norm_fwd_bwd(const float* Z, int64_t ldz, int64_t B, int32_t d, int32_t n_neg, float neg_weight,
                                        float scale, float* Y, int64_t ldy, float* loss_rows, float* rr_rows, float* aff_all,
                                        int64_t ld_aff, float* dZ, int64_t lddz, float* neg_slabs, void* stream) {
    return linkpred_norm_launch(Z, ldz, B, d, n_neg, neg_weight, scale, Y, ldy, loss_rows, rr_rows, aff_all, ld_aff, dZ, lddz,
                                neg_slabs, nullptr, stream);
}

extern "C" int gs_linkpred_norm_fwd_bwd_step(const float* Z, int64_t ldz, int64_t B, int32_t d, int32_t n_neg, float neg_weight,
                                             float scale, float* Y, int64_t ldy, float* loss_rows, float* rr_rows,
                                             float* aff_all, int64_t ld_aff, float* dZ, int64_t lddz, float* neg_slabs,
                                             float* loss_out, int accumulate, float* mrr_out, uint64_t* c0, uint64_t d0,
                                             uint64_t* c1, uint64_t d1, uint64_t* c2, uint64_t d2, void* stream) {
    GS_REQUIRE(loss_out && mrr_out, "gs_linkpred_norm_fwd_bwd_step: loss_out / mrr_out missing");
    const float inv_b = B > 0 ? 1.0f / (float)B : 0.f;
    const StepEpilogue epi = {loss_rows, B, inv_b, loss_out, accumulate, rr_rows, inv_b, mrr_out, c0, d0, c1, d1, c2, d2};
    return linkpred_norm_launch(Z, ldz, B, d, n_neg, neg_weight, scale, Y, ldy, loss_rows, rr_rows, aff_all, ld_aff, dZ, lddz,
                                neg_slabs, &epi, stream);
}
