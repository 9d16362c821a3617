// K1: neighbor sampling kernels (int32 index work, latency/HBM bound, bit-exact vs the oracle).
//
//  * sample_padded_kernel     exact reference semantics on the padded [N+1, max_deg] table
//                             (neigh_samplers.py:24-29 on the table of minibatch.py:227-259)
//  * sample_csr_kernel        MI355X-native: wavefront-local sampling over CSR.  One wave owns 64
//                             consecutive output slots; the <= 65 distinct source rows those slots
//                             belong to are loaded ONCE per wave (ids -> rowptr pair) by the low
//                             lanes and redistributed to the slots with ds_bpermute (__shfl); each
//                             lane then draws one neighbor with a counter-based xorshift-multiply
//                             hash and the wave stores its 64 picks as one coalesced 256-byte line.
#include "gs_common.h"
#include "gs_sample_dev.h"

__global__ __launch_bounds__(256) void sample_padded_kernel(const int32_t* __restrict__ adj, int32_t max_deg,
                                                            const int32_t* __restrict__ ids, int64_t n,
                                                            const int32_t* __restrict__ col_perm, int32_t s,
                                                            int32_t* __restrict__ out) {
    int64_t total = n * (int64_t)s;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = o / s;
        int32_t j = (int32_t)(o - i * s);
        out[o] = adj[(int64_t)ids[i] * max_deg + col_perm[j]];
    }
}

extern "C" int gs_sample_padded(const int32_t* adj, int64_t n_adj_rows, int32_t max_deg, const int32_t* ids,
                                int64_t n, const int32_t* col_perm, int32_t num_samples, int32_t* out,
                                void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_REQUIRE(adj && ids && col_perm && out, "gs_sample_padded: null pointer");
    GS_REQUIRE(n >= 0 && n_adj_rows > 0 && max_deg > 0, "gs_sample_padded: bad sizes");
    GS_REQUIRE(num_samples > 0 && num_samples <= max_deg,
               "gs_sample_padded: num_samples=%d must be in [1, max_degree=%d] (tf.slice would fail)",
               num_samples, max_deg);
    if (n == 0) return GS_OK;
    int64_t total = n * (int64_t)num_samples;
    int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 2048);
    hipLaunchKernelGGL(sample_padded_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, adj, max_deg, ids, n,
                       col_perm, num_samples, out);
    GS_LAUNCH_CHECK("sample_padded_kernel");
    return GS_OK;
}

// gs_mix64 (gs_common.h): splitmix64 finalizer, restated bit-for-bit in oracle/sampler_hash.py.

__global__ __launch_bounds__(256) void sample_csr_kernel(const int64_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col, int64_t n_nodes,
                                                         int32_t pad_id, const int32_t* __restrict__ ids, int64_t n,
                                                         int32_t s, uint64_t seed, uint64_t step,
                                                         const uint64_t* __restrict__ step_dev, uint32_t hop,
                                                         int64_t global_row_offset, const SampleLaw law,
                                                         int32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t total = n * (int64_t)s;
    const int64_t n_waves = gs_ceil_div(total, 64);
    const uint64_t st = step + (step_dev ? *step_dev : 0ull);
    const uint64_t key = gs_mix64(seed ^ (st * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)hop << 56));
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < n_waves; w += (int64_t)gridDim.x * 4) {
        const int64_t o0 = w * 64;
        const int64_t i_first = o0 / s;
        // ---- per-wave row table: lane l holds (begin, deg) of source row i_first + l
        int64_t beg = 0;
        int32_t deg = 0, row_id = 0;
        {
            int64_t i = i_first + lane;
            if (i < n) {
                int32_t id = ids[i];
                row_id = id;
                if (id >= 0 && (int64_t)id < n_nodes) {
                    int64_t b = rowptr[id], e = rowptr[id + 1];
                    beg = b;
                    deg = (int32_t)(e - b);
                }
            }
        }
        // slot -> row redistribution through the LDS crossbar (ds_bpermute)
        const int64_t o = o0 + lane;
        const int64_t i = (o < total) ? o / s : i_first;
        const int src_lane = (int)(i - i_first);  // < 64 whenever o < total and s >= 1 (<= 64 rows/wave)
        // s == 1 gives exactly 64 rows per wave (lanes 0..63) -> still fits the wave.
        const int32_t my_deg = __shfl(deg, src_lane, 64);
        const uint32_t beg_lo = (uint32_t)__shfl((int)(uint32_t)beg, src_lane, 64);
        const uint32_t beg_hi = (uint32_t)__shfl((int)(uint32_t)(beg >> 32), src_lane, 64);
        const int32_t my_id = __shfl(row_id, src_lane, 64);
        const int64_t my_beg = (int64_t)(((uint64_t)beg_hi << 32) | beg_lo);
        if (o < total) {
            const uint32_t j = (uint32_t)(o - i * s);
            int32_t pick = pad_id;
            if (my_deg > 0) {
                pick = col[my_beg + gs_draw(law, seed, key, global_row_offset + i, j, s, my_id, (uint32_t)my_deg)];
            }
            out[o] = pick;  // 64 lanes -> one 256-byte coalesced store
        }
    }
}

extern "C" int gs_sample_uniform_csr(const int64_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t pad_id,
                                     const int32_t* ids, int64_t n, int32_t num_samples, uint64_t seed,
                                     uint64_t step, const uint64_t* step_dev, uint32_t hop,
                                     int64_t global_row_offset, int32_t law, int32_t max_degree, int32_t* out,
                                     void* stream) {
    if (n == 0) return GS_OK;  // empty input: nothing to launch (pointers may be null)
    GS_REQUIRE(rowptr && col && ids && out, "gs_sample_uniform_csr: null pointer");
    GS_REQUIRE(n >= 0 && n_nodes > 0 && num_samples > 0, "gs_sample_uniform_csr: bad sizes");
    GS_REQUIRE(hop < 256, "gs_sample_uniform_csr: hop must be < 256");
    SampleLaw lw;
    if (gs_law_args(law, max_degree, &num_samples, 1, &lw) != GS_OK) return GS_EINVAL;
    int64_t total = n * (int64_t)num_samples;
    int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(sample_csr_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, n_nodes,
                       pad_id, ids, n, num_samples, seed, step, step_dev, hop, global_row_offset, lw, out);
    GS_LAUNCH_CHECK("sample_csr_kernel");
    return GS_OK;
}

__global__ __launch_bounds__(256) void select_batch_kernel(const int32_t* __restrict__ order, int64_t n_order,
                                                           const uint64_t* __restrict__ cursor, int64_t n,
                                                           int32_t* __restrict__ batch) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint64_t c = cursor ? *cursor : 0ull;
        batch[i] = order[(int64_t)((c + (uint64_t)i) % (uint64_t)n_order)];
    }
}

extern "C" int gs_select_batch(const int32_t* order, int64_t n_order, const uint64_t* cursor_dev, int64_t n,
                               int32_t* batch, void* stream) {
    GS_REQUIRE(order && batch && n_order > 0 && n >= 0, "gs_select_batch: bad args");
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(select_batch_kernel, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       order, n_order, cursor_dev, n, batch);
    GS_LAUNCH_CHECK("select_batch_kernel");
    return GS_OK;
}

// batch selection + label-row gather in one launch: one wave per batch row.
__global__ __launch_bounds__(256) void stage_batch_kernel(const int32_t* __restrict__ order, int64_t n_order,
                                                          const uint64_t* __restrict__ cursor, int64_t n,
                                                          int32_t* __restrict__ batch, const float* __restrict__ table,
                                                          int64_t ldt, int32_t C, float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const uint64_t c = cursor ? *cursor : 0ull;
    const int32_t id = order[(int64_t)((c + (uint64_t)i) % (uint64_t)n_order)];
    if (lane == 0) batch[i] = id;
    const int Cp = (C + 3) & ~3;
    for (int k = lane; k < Cp; k += 64) out[i * ldo + k] = k < C ? table[(int64_t)id * ldt + k] : 0.f;
}

extern "C" int gs_stage_batch(const int32_t* order, int64_t n_order, const uint64_t* cursor_dev, int64_t n,
                              int32_t* batch, const float* label_table, int64_t ld_table, int32_t C,
                              float* labels_out, int64_t ld_out, void* stream) {
    if (n == 0) return GS_OK;
    GS_REQUIRE(order && batch && label_table && labels_out && n_order > 0 && C > 0, "gs_stage_batch: bad args");
    GS_REQUIRE(ld_table >= C && ld_out >= ((C + 3) & ~3), "gs_stage_batch: ld too small");
    hipLaunchKernelGGL(stage_batch_kernel, dim3((unsigned)gs_ceil_div(n, 4)), dim3(256), 0, (hipStream_t)stream, order,
                       n_order, cursor_dev, n, batch, label_table, ld_table, C, labels_out, ld_out);
    GS_LAUNCH_CHECK("stage_batch_kernel");
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// Fused multi-hop fan-out: ONE launch samples every hop of a mini-batch (and optionally stages the batch ids
// and label rows).  One workgroup per root node; the ids of hop h are kept in an LDS fan-out buffer so hop h+1
// reads its parents from LDS, and every hop is written to the contiguous id buffer
// [roots | hop-1 | hop-2 | ...] with coalesced stores.  Draws are bit-identical to gs_sample_uniform_csr
// called hop by hop (same counter-based hash of (seed, step, hop, global row, j)).
__global__ __launch_bounds__(256) void sample_fanout_kernel(const FanoutArgs a) {
    __shared__ int32_t lvl[2][GS_FANOUT_LDS];
    __shared__ int32_t law_cols[GS_MAX_HOPS][GS_LAW_COLS];
    sample_fanout_root<GS_FANOUT_LDS>(a, blockIdx.x, lvl, law_cols);
}

extern "C" int gs_sample_fanout_csr(const int64_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t pad_id,
                                    int32_t n_hops, const int32_t* fan_host, const int64_t* offsets_host,
                                    int32_t* ids_all, int64_t B, uint64_t seed, uint64_t step,
                                    const uint64_t* step_dev, uint32_t hop0, int64_t root_offset,
                                    const int32_t* order, int64_t n_order, const uint64_t* cursor_dev,
                                    const float* label_table, int64_t ld_table, int32_t C, float* labels_out,
                                    int64_t ld_out, int32_t law, int32_t max_degree, void* stream) {
    if (B == 0) return GS_OK;
    FanoutArgs a;
    int64_t kmax = 0;
    int rc = gs_fanout_args(rowptr, col, n_nodes, pad_id, n_hops, fan_host, offsets_host, ids_all, B, seed, step, step_dev, hop0,
                            root_offset, order, n_order, cursor_dev, label_table, ld_table, C, labels_out, ld_out, law, max_degree,
                            &a, &kmax);
    if (rc != GS_OK) return rc;
    GS_REQUIRE(kmax <= GS_FANOUT_LDS, "gs_sample_fanout_csr: per-root fan-out %lld exceeds the LDS buffer", (long long)kmax);
    hipLaunchKernelGGL(sample_fanout_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("sample_fanout_kernel");
    return GS_OK;
}

// The virtual padded table of GS_LAW_REFERENCE, materialised: table[v][c] = col[rowptr[v] + gs_table_entry(key(seed, v), c,
// deg, M)] (pad_id for deg == 0 and for the extra row n_nodes) -- what minibatch.py:227-245 builds on the host, entry for
// entry the function the sampler evaluates per draw when it has no table.  One thread per entry.
__global__ __launch_bounds__(256) void build_padded_table_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                 int64_t n_nodes, int32_t pad_id, int32_t M, uint64_t seed,
                                                                 int32_t* __restrict__ table) {
    const int64_t total = (n_nodes + 1) * (int64_t)M;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = t / M;
        const uint32_t c = (uint32_t)(t - v * M);
        int32_t pick = pad_id;
        if (v < n_nodes) {
            const int64_t b = rowptr[v];
            const int32_t deg = (int32_t)(rowptr[v + 1] - b);
            if (deg > 0) pick = col[b + (int64_t)gs_table_entry(gs_table_key(seed, (int32_t)v), c, (uint32_t)deg, (uint32_t)M)];
        }
        table[t] = pick;
    }
}

extern "C" int gs_build_padded_table(const int64_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t pad_id,
                                     int32_t max_degree, uint64_t seed, int32_t* table_out, void* stream) {
    GS_REQUIRE(rowptr && col && table_out && n_nodes > 0 && max_degree > 0, "gs_build_padded_table: bad args");
    GS_REQUIRE(n_nodes < (1ll << 31) - 1, "gs_build_padded_table: node ids must fit int32");
    const int64_t total = (n_nodes + 1) * (int64_t)max_degree;
    const int blocks = (int)std::min<int64_t>(gs_ceil_div(total, 256), 1 << 16);
    hipLaunchKernelGGL(build_padded_table_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, n_nodes, pad_id,
                       max_degree, seed, table_out);
    GS_LAUNCH_CHECK("build_padded_table_kernel");
    return GS_OK;
}

// The same launch from a descriptor (what gs_flat_reduce_adam_sample takes as a rider), including the optional
// unsupervised root staging (edge-pair batch + unigram negatives, see gs_fanout_desc).
extern "C" int gs_sample_fanout_desc(const gs_fanout_desc* desc_host, void* stream) {
    GS_REQUIRE(desc_host, "gs_sample_fanout_desc: null descriptor");
    if (desc_host->B == 0) return GS_OK;
    FanoutArgs a;
    int64_t kmax = 0;
    int rc = gs_fanout_args_desc(desc_host, &a, &kmax);
    if (rc != GS_OK) return rc;
    GS_REQUIRE(kmax <= GS_FANOUT_LDS, "gs_sample_fanout_desc: per-root fan-out %lld exceeds the LDS buffer", (long long)kmax);
    hipLaunchKernelGGL(sample_fanout_kernel, dim3((unsigned)a.B), dim3(256), 0, (hipStream_t)stream, a);
    GS_LAUNCH_CHECK("sample_fanout_kernel");
    return GS_OK;
}
