// Unique sampled ids of a mini-batch, on the device, with static launch shapes (hipGraph-replayable).
//
// Why: the MaxPool aggregator applies its MLP to every GATHERED neighbor row, relu(X[id] . W + b) (aggregators.py:176-179),
// and the result depends on the node id only -- not on which parent sampled it.  At Reddit's degree 37 % of a step's 133 k
// sampled ids are duplicates, so the 82 GF pooling GEMM can run on the unique rows (52 GF) and the reduce_max can pick
// its rows through an index.  ids are node ids in [0, n_values): a flag array + prefix sum gives, in ascending id order,
//   uniq[0 .. U)   the distinct ids,   inv[j] = position of ids[j] in uniq,   count = U (device word)
// deterministically (no atomics, no sort), in four small launches.
#include "gs_common.h"

#define DD_BLOCKS 256

// Workspace: rank_ws = [flags (n_values) | ranks (n_values)].  The flag half must be ZERO on first use; every call leaves it zero
// again (the rank kernel clears what it has read), so there is no clear launch, and the block offsets are summed by the rank
// blocks themselves (no scan launch): four launches instead of six (round 5: ~5 us of launch latency each).
__global__ __launch_bounds__(256) void dd_mark_kernel(const int32_t* __restrict__ ids, int64_t m, int32_t* __restrict__ flags,
                                                      int64_t nv) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = ids[t];
        if (id >= 0 && (int64_t)id < nv) flags[id] = 1;          // same value from every writer: no race that matters
    }
}

__device__ __forceinline__ int dd_block_sum(int v, int* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const int tot = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return tot;
}

__global__ __launch_bounds__(256) void dd_count_kernel(const int32_t* __restrict__ flags, int64_t nv, int64_t chunk,
                                                       int32_t* __restrict__ sums) {
    __shared__ int red[4];
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(lo + chunk, nv);
    int c = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) c += flags[i] != 0;
    const int tot = dd_block_sum(c, red);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// flags -> ranks (rank of an unflagged entry = number of flagged entries before it), uniq[rank] = id; the flags are cleared behind
// the read; block b's offset = sum of the block counts before it (DD_BLOCKS = blockDim: one count per thread); the last block
// writes the total
__global__ __launch_bounds__(DD_BLOCKS) void dd_rank_kernel(int32_t* __restrict__ flags, int32_t* __restrict__ ranks, int64_t nv,
                                                            int64_t chunk, const int32_t* __restrict__ sums,
                                                            int32_t* __restrict__ uniq, int32_t* __restrict__ count_out) {
    __shared__ int part[256];
    __shared__ int red[4];
    const int own = sums[threadIdx.x];
    const int before = dd_block_sum((int)threadIdx.x < (int)blockIdx.x ? own : 0, red);
    if (blockIdx.x == DD_BLOCKS - 1) {
        const int total = dd_block_sum(own, red);
        if (threadIdx.x == 0) count_out[0] = total;
    }
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(lo + chunk, nv);
    const int64_t per = (chunk + 255) / 256;
    const int64_t a = min(lo + (int64_t)threadIdx.x * per, hi), b = min(a + per, hi);
    int c = 0;
    for (int64_t i = a; i < b; ++i) c += flags[i] != 0;
    part[threadIdx.x] = c;
    __syncthreads();
    // exclusive scan of the 256 per-thread counts (Hillis-Steele, in place with a double read per step)
    for (int off = 1; off < 256; off <<= 1) {
        const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int r = before + part[threadIdx.x] - c;
    for (int64_t i = a; i < b; ++i) {
        const int f = flags[i];
        ranks[i] = r;
        if (f) {
            flags[i] = 0;
            uniq[r] = (int32_t)i;
            ++r;
        }
    }
}

__global__ __launch_bounds__(256) void dd_inv_kernel(const int32_t* __restrict__ ids, int64_t m, const int32_t* __restrict__ rank,
                                                     int64_t nv, int32_t* __restrict__ inv) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = ids[t];
        inv[t] = (id >= 0 && (int64_t)id < nv) ? rank[id] : 0;
    }
}

extern "C" int gs_unique_ids(const int32_t* ids, int64_t m, int64_t n_values, int32_t* rank_ws, int32_t* sums_ws,
                             int32_t* uniq_out, int32_t* inv_out, int32_t* count_out, void* stream) {
    GS_REQUIRE(ids && rank_ws && sums_ws && uniq_out && inv_out && count_out && m > 0 && n_values > 0, "gs_unique_ids: bad args");
    GS_REQUIRE(n_values < (1ll << 31), "gs_unique_ids: ids must fit int32");
    hipStream_t st = (hipStream_t)stream;
    const int64_t chunk = gs_ceil_div(n_values, DD_BLOCKS);
    const int mblocks = (int)std::min<int64_t>(gs_ceil_div(m, 256), 2048);
    int32_t* flags = rank_ws;
    int32_t* ranks = rank_ws + n_values;
    hipLaunchKernelGGL(dd_mark_kernel, dim3(mblocks), dim3(256), 0, st, ids, m, flags, n_values);
    hipLaunchKernelGGL(dd_count_kernel, dim3(DD_BLOCKS), dim3(256), 0, st, flags, n_values, chunk, sums_ws);
    hipLaunchKernelGGL(dd_rank_kernel, dim3(DD_BLOCKS), dim3(DD_BLOCKS), 0, st, flags, ranks, n_values, chunk, sums_ws, uniq_out, count_out);
    hipLaunchKernelGGL(dd_inv_kernel, dim3(mblocks), dim3(256), 0, st, ids, m, ranks, n_values, inv_out);
    GS_LAUNCH_CHECK("gs_unique_ids");
    return GS_OK;
}
