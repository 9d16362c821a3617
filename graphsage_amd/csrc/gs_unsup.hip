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
                                                          int32_t* __restrict__ ids_out) {
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
        const uint32_t r = (uint32_t)(gs_mix64(key + (uint64_t)t) >> 32);
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
                              const uint64_t* clock_dev, int32_t* ids_out, void* stream) {
    GS_REQUIRE(ids_out && B >= 0 && n_neg >= 0, "gs_unsup_stage: bad args");
    GS_REQUIRE(!pairs || n_pairs > 0, "gs_unsup_stage: empty pair list");
    GS_REQUIRE(!cdf || n_nodes > 0, "gs_unsup_stage: empty cdf");
    const int64_t n = std::max<int64_t>(pairs ? B : 0, cdf ? n_neg : 0);
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(unsup_stage_kernel, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, pairs,
                       n_pairs, cursor_dev, B, cdf, n_nodes, n_neg, seed, clock_dev, ids_out);
    GS_LAUNCH_CHECK("unsup_stage_kernel");
    return GS_OK;
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
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
    const float sa = 1.0f / (1.0f + expf(-aff));
    const float da = (sa - 1.0f) * scale;
    float loss = fmaxf(aff, 0.f) - aff + log1pf(expf(-fabsf(aff)));
    int rank = 0;
#pragma unroll
    for (int j = 0; j < DJ; ++j) g1[j] = da * o2[j];
    float* mypart = part + (size_t)wave * n_neg * d;
    for (int q = 0; q < n_neg; ++q) {
        float na = 0.f;
#pragma unroll
        for (int j = 0; j < DJ; ++j) na += o1[j] * negs[q * d + j * 64 + lane];
        na = wsum(na);
        loss += neg_w * (fmaxf(na, 0.f) + log1pf(expf(-fabsf(na))));
        rank += (na >= aff) ? 1 : 0;
        const float gq = live ? neg_w * scale / (1.0f + expf(-na)) : 0.f;
        if (aff_all && live && lane == 0) aff_all[i * ld_aff + q] = na;
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            g1[j] += gq * negs[q * d + j * 64 + lane];
            mypart[q * d + j * 64 + lane] = gq * o1[j];
        }
    }
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
        static bool attr = false;                                                                                          \
        if (!attr) {                                                                                                       \
            GS_HIP(hipFuncSetAttribute((const void*)linkpred_fwd_bwd_kernel<DJ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr = true;                                                                                                   \
        }                                                                                                                  \
        hipLaunchKernelGGL((linkpred_fwd_bwd_kernel<DJ>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, Y, ldy, B, n_neg, \
                           neg_weight, scale, loss_rows, rr_rows, aff_all, ld_aff, dY, lddy, neg_slabs);                  \
    } while (0)
    if (d == 64) GS_LP(1); else if (d == 128) GS_LP(2); else if (d == 256) GS_LP(4); else GS_LP(8);
#undef GS_LP
    GS_LAUNCH_CHECK("linkpred_fwd_bwd_kernel");
    return GS_OK;
}
