// Trainable identity features (graphsage/models.py:229-240, supervised_models.py:49-60): `node_embeddings`
// [N+1, identity_dim] is concatenated IN FRONT of the fixed feature columns.  Here the concatenation exists once, as
// the leading columns of the combined feature table the gather kernels already read (one row fetch per sampled id,
// not two), the master copy of the embedding is an ordinary variable of the flat parameter buffer, and
//   * gs_scatter_add_rows accumulates the layer-0 input gradients into the embedding gradient (the only place where
//     a gradient flows into a gathered table; tf.gradients of embedding_lookup = IndexedSlices summed per id),
//   * gs_copy_cols refreshes the table's leading columns after the optimizer step.
#include "gs_common.h"

// table[ids[i*s + j], c] += scale * d[i, c]   for i < n, j < s, c < cols
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ d, int64_t ldd, int64_t n,
                                                               int32_t s, int32_t cols, float scale,
                                                               const int32_t* __restrict__ ids, float* __restrict__ table,
                                                               int64_t ldt) {
    const int64_t total = n * s * cols;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / cols;            // sampled row
        const int c = (int)(t - r * cols);
        const int64_t i = r / s;               // the output row it was drawn for
        unsafeAtomicAdd(&table[(int64_t)ids[r] * ldt + c], scale * d[i * ldd + c]);   // global_atomic_add_f32
    }
}

extern "C" int gs_scatter_add_rows(const float* d, int64_t ldd, int64_t n, int32_t s, int32_t cols, float scale,
                                   const int32_t* ids, float* table, int64_t ldt, void* stream) {
    if (n == 0) return GS_OK;
    GS_REQUIRE(d && ids && table && n > 0 && s > 0 && cols > 0 && ldd >= cols && ldt >= cols, "gs_scatter_add_rows: bad args");
    const int64_t total = n * s * cols;
    const int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 65536);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d, ldd, n, s, cols, scale,
                       ids, table, ldt);
    GS_LAUNCH_CHECK("scatter_add_rows_kernel");
    return GS_OK;
}

__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int64_t ld_src,
                                                        float* __restrict__ dst, int64_t ld_dst, int64_t rows, int32_t cols) {
    const int64_t total = rows * cols;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / cols;
        const int c = (int)(t - r * cols);
        dst[r * ld_dst + c] = src[r * ld_src + c];
    }
}

extern "C" int gs_copy_cols(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int32_t cols,
                            void* stream) {
    if (rows == 0) return GS_OK;
    GS_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, "gs_copy_cols: bad args");
    const int blocks = (int)std::min<int64_t>(gs_ceil_div(rows * cols, 256), 65536);
    hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, rows, cols);
    GS_LAUNCH_CHECK("copy_cols_kernel");
    return GS_OK;
}
