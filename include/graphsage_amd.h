/*
 * graphsage_amd.h -- C ABI of the MI355X (gfx950) GraphSAGE sample-and-aggregate engine.
 *
 * The reference (williamleif/GraphSAGE, 100 % Python on TensorFlow 1.x) has no FFI of its own;
 * its hot path bottoms out in TF ops.  Each entry point below replaces the TF-op call sites of
 * one reference operator (cited as graphsage/<file>:<line>, relative to the reference tree) and
 * is what a ctypes binding of that operator binds.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - Every pointer is a DEVICE pointer owned by the caller unless the name ends in _host.
 *   - Every function enqueues work on `stream` (a hipStream_t passed as void*; must not be the
 *     legacy NULL stream if the caller wants to capture it into a hipGraph) and returns
 *     immediately.  The library allocates no device memory; scratch is caller-supplied.
 *   - Return value: 0 = ok, <0 = error (GS_E*); gs_last_error() gives a thread-local message.
 *     No C++ exception crosses the boundary.
 *   - Matrices are row-major fp32 with an explicit leading dimension `ld*` (elements).  Every
 *     fp32 matrix base pointer must be 16-byte aligned and every ld a multiple of 4 floats;
 *     the logical width d may be anything <= ld.  Columns [d, round_up(d,4)) of OUTPUT matrices
 *     are written as zeros; further pad columns are left untouched.
 *   - Index arrays are int32 (the reference feeds int32 placeholders, supervised_train.py:116);
 *     CSR row pointers are int64.
 */
#ifndef GRAPHSAGE_AMD_H
#define GRAPHSAGE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_OK 0
#define GS_EINVAL (-1)  /* bad argument (null pointer, misaligned, bad size)  */
#define GS_EHIP (-2)    /* a HIP runtime call failed                           */
#define GS_ENOTSUP (-3) /* combination not supported by the gfx950 kernels     */

#define GS_ACT_IDENTITY 0
#define GS_ACT_RELU 1

#define GS_ABI_VERSION 10

const char* gs_last_error(void);
int gs_abi_version(void);
/* sizeof() of the descriptor structs below, in declaration order (gs_gather_desc, gs_wgrad_desc, gs_var_desc,
 * gs_fanout_desc, gs_tail_desc, gs_dropout, gs_pull_desc, gs_lp_tail_desc): writes min(count, capacity) values, returns the count.  A
 * binding compares them with its own struct definitions at load time. */
int gs_abi_struct_sizes(int32_t* sizes_out_host, int32_t capacity);
/* Fills CU count, XCD count (8 on MI355X), gcnArchName (>= 64 bytes) of the current device. */
int gs_device_info(int* cu_count, int* xcd_count, char* arch_name_host, int arch_name_len);

/* ---------------------------------------------------------------------------------------------
 * K1  neighbor sampling      replaces UniformNeighborSampler._call, neigh_samplers.py:24-29
 *                            (called at models.py:272, result flattened at models.py:273)
 * ------------------------------------------------------------------------------------------- */

/* Exact reference semantics on the padded table of minibatch.py:227-259:
 *   out[i, j] = adj[ids[i], col_perm[j]],  j < num_samples
 * i.e. embedding_lookup (:26) + ONE column permutation shared by all rows (:27) + slice (:28).
 * TF's random_shuffle stream is not reproducible, so the permutation is an input.
 * adj is [n_adj_rows, max_deg] int32 (n_adj_rows = N+1, row N = all-pad).  Bit-exact. */
int gs_sample_padded(const int32_t* adj, int64_t n_adj_rows, int32_t max_deg,
                     const int32_t* ids, int64_t n,
                     const int32_t* col_perm, int32_t num_samples,
                     int32_t* out, void* stream);

/* MI355X-native sampler over a CSR adjacency (rowptr int64 [n_nodes+1], col int32 [nnz]):
 *   deg = rowptr[id+1]-rowptr[id];  out[i,j] = deg ? col[rowptr[id] + ((u(i,j) >> 32) * deg >> 32)] : pad_id
 * u(i,j) is a counter-based xorshift-multiply hash of
 *   (seed, step + *step_dev, hop, global_row_offset + i, j)
 * so the draw for a given root row does not depend on how rows are sharded over GPUs.
 * ids >= n_nodes (the pad id) yield pad_id (the all-pad row N of the reference table).
 * law = GS_LAW_IID: uniform WITH replacement over the true neighbor set (the formula above): the per-slot marginal
 *   equals the reference's (padded row resampled once + distinct columns); max_degree ignored.
 * law = GS_LAW_REFERENCE: the reference's joint law without its table.  Node v's row of the padded [N+1, max_degree]
 *   table of minibatch.py:227-245 is VIRTUAL -- entry c is a pure function of (seed, v, c): the first max_degree
 *   elements of a keyed permutation of the neighbor list when deg > max_degree (np.random.choice(replace=False),
 *   :240-241), max_degree frozen iid draws when deg < max_degree (replace=True, :242-243), the list itself when equal
 *   -- frozen for the run like the reference's table; per call, num_samples DISTINCT columns are the head of ONE keyed
 *   permutation of [0, max_degree) shared by all rows (tf.random_shuffle of the transposed rows, neigh_samplers.py:27-28).
 *   Requires num_samples <= max_degree (tf.slice fails otherwise).
 * law = GS_LAW_DISTINCT: per-row independent draws WITHOUT replacement whenever the list (capped to a frozen
 *   max_degree subset if max_degree > 0) holds >= num_samples entries, with replacement otherwise.
 * All three are pure functions of (seed, step, hop, global row, j[, node id]); restated in oracle/sampler_hash.py.
 * step_dev may be NULL. */
#define GS_LAW_IID 0
#define GS_LAW_REFERENCE 1
#define GS_LAW_DISTINCT 2
int gs_sample_uniform_csr(const int64_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t pad_id,
                          const int32_t* ids, int64_t n, int32_t num_samples,
                          uint64_t seed, uint64_t step, const uint64_t* step_dev, uint32_t hop,
                          int64_t global_row_offset, int32_t law, int32_t max_degree, int32_t* out, void* stream);

/* Fused multi-hop fan-out (replaces the K calls of models.py:268-274 with ONE launch): one workgroup per root,
 * hop h kept in an LDS fan-out buffer for hop h+1, all hops written to the contiguous buffer
 *   ids_all = [roots (B) | hop 1 (B*fan[0]) | hop 2 (B*fan[0]*fan[1]) | ...]   at offsets_host[0..n_hops].
 * fan_host[h] is the fan-out of the h-th sampler call (= layer_infos[K-1-h].num_samples).  Draws are bit-identical
 * to n_hops calls of gs_sample_uniform_csr with hop = hop0 + h and global_row_offset = root_offset*support[h].
 * If order != NULL the roots are first staged as ids_all[i] = order[(*cursor_dev + i) % n_order] and, if
 * label_table != NULL, labels_out[i] = label_table[root_i] (minibatch.py:264-274, 302-307 on the device). */
int gs_sample_fanout_csr(const int64_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t pad_id,
                         int32_t n_hops, const int32_t* fan_host, const int64_t* offsets_host,
                         int32_t* ids_all, int64_t B, uint64_t seed, uint64_t step, const uint64_t* step_dev,
                         uint32_t hop0, int64_t root_offset,
                         const int32_t* order, int64_t n_order, const uint64_t* cursor_dev,
                         const float* label_table, int64_t ld_table, int32_t C, float* labels_out, int64_t ld_out,
                         int32_t law, int32_t max_degree, void* stream);

/* batch[i] = order[(*cursor_dev + i) % n_order] for i < n  (epoch order lives on the device so the
 * whole training step can be one hipGraph).  Replaces the host slicing of minibatch.py:302-307. */
int gs_select_batch(const int32_t* order, int64_t n_order, const uint64_t* cursor_dev,
                    int64_t n, int32_t* batch, void* stream);

/* *counter_dev += delta  (advances step / cursor inside a captured graph). */
int gs_advance_counter(uint64_t* counter_dev, uint64_t delta, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2  feature gather (+ segmented mean)
 *     replaces tf.nn.embedding_lookup at models.py:299, the reshape at models.py:326-327 and
 *     reduce_mean(axis=1) at aggregators.py:48 (Mean) / :106-107 (GCN)
 * ------------------------------------------------------------------------------------------- */

/* out[i, :] = X[ids[i], :d] */
int gs_gather_rows(const float* X, int64_t ldx, const int32_t* ids, int64_t n, int32_t d,
                   float* out, int64_t ldo, void* stream);

/* mean[i, :] = scale * ( sum_{j<s} X[idx[i*s+j], :d]  [+ S[self_idx ? self_idx[i] : i, :d]] )
 *   idx == NULL       -> neighbor row is i*s+j (contiguous groups; hidden layers, models.py:327)
 *   self_src == NULL  -> MeanAggregator, scale = 1/s          (aggregators.py:48)
 *   self_src != NULL  -> GCNAggregator,  scale = 1/(s+1)      (aggregators.py:106-107)
 * The [n*s, d] gathered tensor of models.py:299 is never materialised. */
int gs_gather_mean_fwd(const float* X, int64_t ldx, const int32_t* idx, int64_t n, int32_t s, int32_t d,
                       const float* self_src, int64_t ld_self, const int32_t* self_idx,
                       float* mean, int64_t ldm, void* stream);

/* Backward of the segmented mean for hidden layers (gradient of aggregators.py:48 / :106-107):
 *   g = d_mean[r / s, :] * scale;  if (mask_y) g *= (mask_y[r, :] > 0)   (fused relu grad of the
 *   producing layer, aggregators.py:64);  d_neigh[r, :] (+)= g   for r < n*s. */
int gs_mean_bwd(const float* d_mean, int64_t ldd, int64_t n, int32_t s, int32_t d, float scale,
                const float* mask_y, int64_t ldy, float* d_neigh, int64_t ldn, int accumulate,
                void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3  dense contraction (fp32 MFMA, v_mfma_f32_32x32x2_f32)
 *     replaces tf.matmul at aggregators.py:51,53 (Mean), :110 (GCN), :183-184 (MaxPool),
 *     concat/add_n at :56-58, bias :61-62, act :64, and Dense._call layers.py:104-116
 * ------------------------------------------------------------------------------------------- */

/* out = act( [ self·W_self  ||  agg·W_neigh ] + bias )          concat != 0   (out is [n, 2*out_dim])
 * out = act(   self·W_self  +   agg·W_neigh   + bias )          concat == 0   (out is [n, out_dim])
 * self == NULL -> single term  out = act(agg·W_neigh + bias)  (GCN / Dense).
 * self_idx != NULL gathers the self rows from `self` on the fly (self row i = self[self_idx[i]]),
 * which is how layer 0 consumes X[samples[hop]] without materialising it.  Same for agg_idx. */
int gs_sage_dense_fwd(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self,
                      const float* agg, int64_t ld_agg, const int32_t* agg_idx, int32_t d_agg,
                      int64_t n,
                      const float* W_self, int64_t ldw_self, const float* W_neigh, int64_t ldw_neigh,
                      int32_t out_dim, int concat, int act, const float* bias,
                      float* out, int64_t ldo, void* stream);

/* Horizontally fused launch: gs_sage_dense_fwd (two-term form) PLUS up to 4 independent gs_gather_mean_fwd jobs in
 * the SAME kernel launch -- the GEMM tiles are the first workgroups, the gather waves back-fill the CUs.  Used to
 * overlap the MFMA-bound layer-0 contraction of step t with the HBM-bound neighbor gather of step t+1 (which needs
 * no weights) without cross-stream synchronisation.  Results are identical to the separate calls. */
typedef struct gs_gather_desc {
    const float* X;            /* [*, ldx] table                                         */
    const int32_t* idx;        /* [n*s] (NULL -> contiguous groups)                       */
    const float* self_src;     /* GCN: self rows (NULL -> MeanAggregator scale 1/s)       */
    const int32_t* self_idx;
    float* out;                /* [n, ldo]                                                */
    int64_t ldx, ld_self, ldo, n;
    int32_t s, d;
} gs_gather_desc;
int gs_sage_dense_fwd_cogather(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self,
                               const float* agg, int64_t ld_agg, const int32_t* agg_idx, int32_t d_agg,
                               int64_t n,
                               const float* W_self, int64_t ldw_self, const float* W_neigh, int64_t ldw_neigh,
                               int32_t out_dim, int concat, int act, const float* bias,
                               float* out, int64_t ldo,
                               const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);

/* Weight gradient as split-K slabs (deterministic; no atomics):
 *   slab[z][:, :] = sum_{r in slice z} A[a_idx? a_idx[r] : r, :d]^T · dZ[r, col0:col0+out_dim]
 * for z < n_slabs (row slices of equal size).  slabs is [n_slabs, d, ld_slab].  The caller sums
 * the slabs with gs_reduce_slabs.  Gradient of the tf.matmul call sites above w.r.t. the weights. */
int gs_dense_wgrad(const float* A, int64_t lda, const int32_t* a_idx, int32_t d,
                   const float* dZ, int64_t ldz, int32_t col0, int32_t out_dim, int64_t n,
                   int32_t n_slabs, float* slabs, int64_t ld_slab, void* stream);

/* Both input gradients of a SAGE dense layer in ONE launch (two-term NT GEMM, concatenated output):
 *   dX2[:, 0:d_in]       = dZ_self  · W_self^T      dZ_self  = dZ[:, 0:out_dim]
 *   dX2[:, d_in:2*d_in]  = dZ_neigh · W_neigh^T     dZ_neigh = dZ[:, out_dim:2*out_dim] if fwd_concat else dZ_self
 * d_in % 4 == 0 and (if fwd_concat) out_dim % 4 == 0.  Gradient of aggregators.py:51,53. */
int gs_sage_dense_dgrad(const float* dZ, int64_t ldz, int64_t n, int32_t out_dim, int fwd_concat,
                        const float* W_self, int64_t ldw_self, const float* W_neigh, int64_t ldw_neigh,
                        int32_t d_in, float* dX2, int64_t ldx, void* stream);

/* All weight gradients of one backward pass in ONE launch (grouped split-K GEMM).  Each descriptor is one
 * gs_dense_wgrad problem; a bias gradient is the problem A = ones[n, 1] (d = 1).  descs_host is a HOST array
 * that is consumed at call time (kernel arguments are copied), so it may be freed right after the call. */
typedef struct gs_wgrad_desc {
    const float* A;        /* [rows, lda] source of the reduction (row-gathered through a_idx if non-null) */
    const int32_t* a_idx;
    const float* dZ;       /* [n, ldz] */
    float* slabs;          /* [n_slabs, d, ld_slab] */
    int64_t lda, ldz, ld_slab, n;
    int32_t d, col0, out_dim, n_slabs;
    int64_t a_rows;        /* rows of the A table when a_idx != NULL (0 = not stated; gs_dense_wgrad_grouped_stream needs it) */
} gs_wgrad_desc;
int gs_dense_wgrad_grouped(const gs_wgrad_desc* descs_host, int32_t n_desc, void* stream);
/* The grouped weight-gradient launch + up to 4 gather+mean jobs (see gs_sage_dense_fwd_cogather) in ONE horizontally
 * fused launch: lets the next step's HBM-bound gather be split between the layer-0 forward and the weight-gradient
 * launch of the current step.  Results are those of the separate calls. */
int gs_dense_wgrad_grouped_cogather(const gs_wgrad_desc* descs_host, int32_t n_desc, const gs_gather_desc* jobs_host,
                                    int32_t n_jobs, void* stream);

/* "Stream" forms of the two launches above (same maths, same co-scheduled gather jobs, up to 6 of them): the
 * contraction waves stage no operand through LDS and meet no barrier inside the K loop -- operands go from L2 into the
 * MFMA registers through a register ring -- so that the gather waves sharing the launch keep their occupancy.
 *   gs_sage_dense_fwd_stream: out[:, 0:out_dim] = act(self[self_idx] . W_self + bias), out[:, out_dim:2*out_dim] =
 *     act(agg . W_neigh + bias)  (concat form of aggregators.py:51-58; self == NULL: the single GCN contraction,
 *     aggregators.py:110).  self_idx (nullable) gathers the self rows inside the A loads; agg is [n, d] dense; pad
 *     columns [d, round_up(d, 4)) must be readable.  out_dim and ldo even.  One workgroup per 32 x 64 output tile,
 *     K split over its four waves and summed in a fixed order (deterministic).
 *   gs_dense_wgrad_grouped_stream: as gs_dense_wgrad_grouped (<= 12 problems; a row-gathered problem needs
 *     ceil(n / n_slabs) <= 512 -- the slice's row offsets live in registers -- and a_rows). */
int gs_sage_dense_fwd_stream(const float* self, int64_t ld_self, const int32_t* self_idx, const float* agg, int64_t ld_agg,
                             int32_t d, int64_t n, const float* W_self, int64_t ldw_self, const float* W_neigh,
                             int64_t ldw_neigh, int32_t out_dim, int act, const float* bias, float* out, int64_t ldo,
                             const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);
/* gs_sage_dense_fwd_stream2 (ABI 8): the same launch with different reduction lengths of the two terms -- the pooling
 * aggregators' from_self = self[self_idx] . W_self over d_self features and from_neighs = pooled . W_neigh over d_agg = hidden_dim
 * (aggregators.py:183-187, :261-265), concat form; pad columns of both operands readable as above. */
int gs_sage_dense_fwd_stream2(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self, const float* agg,
                              int64_t ld_agg, int32_t d_agg, int64_t n, const float* W_self, int64_t ldw_self,
                              const float* W_neigh, int64_t ldw_neigh, int32_t out_dim, int act, const float* bias, float* out,
                              int64_t ldo, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);
int gs_dense_wgrad_grouped_stream(const gs_wgrad_desc* descs_host, int32_t n_desc, const gs_gather_desc* jobs_host,
                                  int32_t n_jobs, void* stream);
/* gs_dense_wgrad_grouped_tiled3 (ABI 9, round 6): gs_dense_wgrad_grouped_stream -- same descriptors, same split-K slabs, same
 * co-scheduled gather jobs -- in the three-piece bf16 arithmetic of gs_sage_dense_fwd_tiled3 (fp32 in and out, the accuracy of an
 * fp32 FMA chain), LDS-tiled: one 8-wave workgroup per (64 x 128 tile of dW, reduction slice), both operands moved HBM -> LDS raw
 * by LDS-DMA and cut by the waves that read them.  <= 12 problems; a slice is round_up(ceil(n / n_slabs), 32) rows and must not
 * exceed 1024 (any problem, gathered or dense); row addresses are 64-bit (tables beyond 4 GB are fine: a_rows is not needed).  A slab whose slice is empty is
 * written as zeros.  Replaces the TF-op group of aggregators.py:51-58's gradient (tf.gradients of the two matmuls). */
int gs_dense_wgrad_grouped_tiled3(const gs_wgrad_desc* descs_host, int32_t n_desc, const gs_gather_desc* jobs_host,
                                  int32_t n_jobs, void* stream);
/* gs_sage_dense_fwd_tiled3 (ABI 9, round 6): the contraction of gs_sage_dense_fwd_stream2 -- same arguments, fp32 operands in and
 * out -- on the bf16 matrix pipe in the three-piece arithmetic described below (every fp32 operand = three bf16 pieces, six exact
 * piece products per element pair, fp32 accumulation: the accuracy of an fp32 FMA chain), LDS-tiled: one 8-wave workgroup per
 * 64 x 128 output tile of a term, stage = 32 k; the A rows ((gathered) fp32) are cut once per workgroup in registers, the weights
 * go global -> LDS by DMA as fp32 and are cut by the wave that contracts them (no pre-cut copy, nothing to refresh after an
 * optimizer step); the two 16-k halves of a stage are contracted by different waves and summed once in the epilogue (fixed order:
 * deterministic).  Pad columns [d, round_up(d, 4)) of self / agg must be readable (any value: masked); out_dim and ldo multiples
 * of 4.  self == NULL: one term (GCN).  Gather jobs ride as in gs_sage_dense_fwd_stream. */
int gs_sage_dense_fwd_tiled3(const float* self, int64_t ld_self, const int32_t* self_idx, int32_t d_self, const float* agg,
                             int64_t ld_agg, int32_t d_agg, int64_t n, const float* W_self, int64_t ldw_self, const float* W_neigh,
                             int64_t ldw_neigh, int32_t out_dim, int act, const float* bias, float* out, int64_t ldo,
                             const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);
/* Contractions on the bf16 matrix pipe WITHOUT giving up fp32: every fp32 operand x is cut into three bf16 pieces
 * x = h + m + l (top / middle / low 8 significant bits: nothing is lost), a product is the sum of piece products -- each formed
 * exactly by v_mfma_f32_32x32x16_bf16 and accumulated in fp32 -- and six of the nine are kept (hh, hm, mh, mm, hl, lh; the
 * dropped ml, lm, ll are <= 3 * 2^-24 |x y|, one fp32 rounding).  Accuracy of an fp32 FMA chain at 6/16 of the fp32 MFMA's
 * matrix-pipe time (fp32 MFMA runs at the vector rate on gfx950, 1/16 of bf16).  Inputs, outputs, accumulation: fp32.
 *   gs_split_rows: W [K, ldw >= N] fp32 -> W3 [G][3][N][8] bf16 (per group of 8 k and piece: the N columns side by side,
 *     16 bytes each -- a B-fragment load of 32 lanes reads 512 contiguous bytes; G = groups up to an even count of 32-k
 *     stages, zero for k >= K: the kernels neither mask nor clamp their B loads); gs_split_rows_bytes gives the size.  Call it after every update of W.
 *   gs_dense_fwd_rows_split: gs_dense_fwd_rows_dev in the same arithmetic, LDS-tiled (128 x 128 per workgroup, A rows gathered
 *     through idx and cut once per workgroup): the pooling MLP of the max-pool aggregator on the step's distinct ids.
 */
int gs_split_rows_bytes(int32_t K, int32_t N, int64_t* bytes_out_host);
int gs_dense_fwd_rows_split(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_max, const int32_t* n_dev,
                            const void* W3, int32_t out_dim, int act, const float* bias, float* out, int64_t ldo, void* stream);
/* gs_dense_fwd_rows_split with a workspace (ABI 8): the wide form's workgroups run one per CU in lock step, so a tile count
 * that is not a multiple of the CU count ends in a round that keeps a few CUs busy (Reddit max-pool step: 1300 tiles on 256 CUs
 * = six rounds for 5.08 rounds of work).  With ws (>= gs_dense_fwd_rows_split_ws_bytes, 16-byte aligned, contents undefined
 * on entry and on return) the tiles of that last round are cut along K into up to ten parts whose fp32 partial tiles are summed
 * in part order by a second, small launch: deterministic, same accuracy class, NOT bit-identical to the call without ws for
 * the rows of those tiles (one more association of the same products).  ws = NULL: gs_dense_fwd_rows_split. */
int gs_dense_fwd_rows_split_ws_bytes(int64_t* bytes_out_host);
int gs_dense_fwd_rows_split_ws(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_max, const int32_t* n_dev,
                               const void* W3, int32_t out_dim, int act, const float* bias, float* out, int64_t ldo,
                               float* ws, int64_t ws_bytes, void* stream);
/* ---- The same contraction on the fp16 matrix pipe with TWO pieces per operand (ABI 8, csrc/gs_split16.hip).
 * Why: the three-piece kernel runs the chip into its power cap (1400 W, engine clock down to 2.02 GHz:
 * profiles/r05_pool_clock_probe.txt), and its ~310 G MACs are most of that energy.  Here x 2^e = h + m + r with h = fp16(x 2^e),
 * m = fp16(x 2^e - h), round to nearest: |r| <= 2^-23 |x 2^e| -- at most ONE fp32 ulp: h keeps 11 bits, the residual has <= 12 of which
 * m keeps 11 (2^-25 on average) -- under a power-of-two scale per feature-table
 * row / weight column that puts its largest element at 2^13..2^14 (no overflow; m stays a normal fp16 for every element within
 * 2^-15 of the largest; below that the absolute error is <= 2^-25 in scaled units, 2^-38 of the largest element).  A product is
 * h h' + h m' + m h' (exact 11 x 11 bit products, fp32 accumulation; the dropped m m' is <= 2^-22 |x y| worst case, 2^-24 rms),
 * the scales are taken out exactly in the epilogue.  Over a K-term dot product what is given up is a random walk of sqrt(K)
 * 2^-23 against the K-term sum, one to two orders of magnitude below the rounding of the fp32 accumulation itself: measured
 * against fp64 the outputs are as accurate as the three-piece kernel's and more accurate than an fp32 FMA chain
 * (tests/test_split_gemm_gpu.py prints all three), at HALF the matrix-pipe work.  Inputs, outputs, accumulation: fp32; the
 * three-piece form stays available (GS_POOL_F16=0) and bench.py reports the step with both.
 *   gs_split_rows_f16: W [K, ldw >= N] fp32 -> W2 [KP / 8][2][N][8] fp16 + N int32 column exponents (KP = K rounded up to an even
 *     count of 32-k stages, zero beyond K); call it after every update of W (Engine.split_of(form="f16x2") does).
 *   gs_split_table_f16: a CONSTANT table X [rows, ldx >= d] -> X2 [rows][2][KP] fp16 + rows int32 row exponents, once (the
 *     reference's features are a non-trainable tf.Variable: models.py:299); 2 x 2 bytes per element = the fp32 table's bytes.
 *   gs_dense_fwd_rows_split16: out[i] = act(X[idx[i]] . W + bias), i < min(n_max, *n_dev), from X2 / W2 (aggregators.py:176-179 via
 *     layers.py:104-116 on the step's distinct ids): 128 x 256 workgroup tiles, both operand tiles are plain copies global -> LDS,
 *     24 MFMAs (v_mfma_f32_32x32x16_f16) per 16 fragment reads; ws as for gs_dense_fwd_rows_split_ws (nullable). */
int gs_split_rows_f16_bytes(int32_t K, int32_t N, int64_t* bytes_out_host);
int gs_split_rows_f16(const float* W, int64_t ldw, int32_t K, int32_t N, void* W2, void* stream);
int gs_split_table_f16_bytes(int64_t rows, int32_t d, int64_t* table_bytes_out_host, int64_t* exp_bytes_out_host);
int gs_split_table_f16(const float* X, int64_t ldx, int64_t rows, int32_t d, void* X2, int32_t* rexp, void* stream);
int gs_dense_fwd_rows_split16(const void* X2, const int32_t* rexp, const int32_t* idx, int32_t d, int64_t n_max, const int32_t* n_dev,
                              const void* W2, int32_t out_dim, int act, const float* bias, float* out, int64_t ldo, float* ws,
                              int64_t ws_bytes, void* stream);
int gs_split_rows(const float* W, int64_t ldw, int32_t K, int32_t N, void* W3, void* stream);

/* Input gradient:  dX[n, d] (+)= dZ[:, col0:col0+out_dim] · W[d, out_dim]^T */
int gs_dense_dgrad(const float* dZ, int64_t ldz, int32_t col0, int32_t out_dim, int64_t n,
                   const float* W, int64_t ldw, int32_t d, float* dX, int64_t ldx, int accumulate,
                   void* stream);

/* dZ = dY * (Y > 0) for relu (tf relu grad, aggregators.py:64), dY for identity.  In-place allowed. */
int gs_act_bwd(const float* dY, int64_t lddy, const float* Y, int64_t ldy, int64_t n, int32_t n_cols,
               int act, float* dZ, int64_t lddz, void* stream);
/* Bias gradient as slabs: slabs[z][c] = sum_{r in row slice z} Z[r, c]; sum them with gs_reduce_slabs
 * (rows = 1).  Deterministic.  slabs is [n_slabs, ld_slab]. */
int gs_colsum_slabs(const float* Z, int64_t ldz, int64_t n, int32_t n_cols, int32_t n_slabs,
                    float* slabs, int64_t ld_slab, void* stream);

/* General fp32 GEMM on the same MFMA kernel family (used by the operators above; exported for tests):
 *   C[M,N] = act( opA(A)·opB(B) + bias ),  opA = A^T if transA (A stored [K, M]), same for B.
 * a_row_idx gathers the SOURCE rows of A (rows of A as stored). */
int gs_gemm_f32(int transA, int transB, int64_t M, int32_t N, int64_t K,
                const float* A, int64_t lda, const int32_t* a_row_idx,
                const float* B, int64_t ldb,
                const float* bias, int act, float* C, int64_t ldc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4  pooling aggregators     replaces aggregators.py:176-181 (MaxPool) / :254-259 (MeanPool)
 * ------------------------------------------------------------------------------------------- */

/* Pooling MLP + reduce_max in ONE launch (aggregators.py:176-181):
 *   pooled[i, c] = max_j relu( X[idx[i*s + j]] . W[:, c] + bias[c] ),   argmax[i, c] = first j attaining it
 * The [n*s, hidden] activations are never written: a GEMM tile holds whole groups of s rows (s <= 64) and reduces them
 * in its epilogue.  Same results as gs_sage_dense_fwd(self = NULL) followed by gs_segment_max_fwd. */
int gs_dense_pool_max_fwd(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_groups, int32_t s,
                          const float* W, int64_t ldw, int32_t hidden, const float* bias, float* pooled, int64_t ldp,
                          int32_t* argmax, int64_t lda, void* stream);

/* The pooling MLP on the step's DISTINCT sampled ids (its output depends on the node id only; at Reddit's degree 37 % of a
 * step's 133 k sampled ids are duplicates):
 *   gs_unique_ids            ids [m] in [0, n_values) -> uniq [count] (ascending), inv [m] (position of ids[j] in uniq) and the
 *                            device word count, by flag array + prefix sum (deterministic, static launch shapes);
 *                            rank_ws: 2 * n_values int32 words whose FIRST n_values are zero on first use (every call leaves them
 *                            zero again: no clear launch; ABI 8), sums_ws: 256
 *   gs_dense_fwd_rows_dev    out[i] = act(X[idx[i]] . W + bias) for i < min(n_max, *n_dev): the row count is a device word
 *   gs_segment_max_gather_fwd  pooled[i, c] = max_j H[inv[i*s + j], c], argmax = first j attaining it: the bits of
 *                            gs_dense_pool_max_fwd on the expanded rows */
int gs_unique_ids(const int32_t* ids, int64_t m, int64_t n_values, int32_t* rank_ws, int32_t* sums_ws, int32_t* uniq_out,
                  int32_t* inv_out, int32_t* count_out, void* stream);
int gs_dense_fwd_rows_dev(const float* X, int64_t ldx, const int32_t* idx, int32_t d, int64_t n_max, const int32_t* n_dev,
                          const float* W, int64_t ldw, int32_t out_dim, int act, const float* bias, float* out, int64_t ldo,
                          void* stream);
int gs_segment_max_gather_fwd(const float* H, int64_t ldh, const int32_t* inv, int64_t n, int32_t s, int32_t hidden,
                              float* pooled, int64_t ldp, int32_t* argmax, int64_t lda, void* stream);

/* H[n*s, hidden] = relu(X[idx] · W_mlp + b_mlp)  (Dense, layers.py:104-116) is produced by
 * gs_sage_dense_fwd(self=NULL, agg=X, agg_idx=idx, ...).  This reduces it:
 *   pooled[i, c] = max_j H[i*s+j, c];  argmax[i, c] = first j attaining it   (reduce_max, :181) */
int gs_segment_max_fwd(const float* H, int64_t ldh, int64_t n, int32_t s, int32_t hidden,
                       float* pooled, int64_t ldp, int32_t* argmax, int64_t lda, void* stream);

/* dH[i*s+j, c] = (j == argmax[i,c] && pooled[i,c] > 0) ? d_pooled[i,c] : 0
 * (reduce_max gradient followed by the relu gradient of the Dense; ties: see DESIGN.md). */
int gs_segment_max_bwd(const float* d_pooled, int64_t ldd, const float* pooled, int64_t ldp,
                       const int32_t* argmax, int64_t lda, int64_t n, int32_t s, int32_t hidden,
                       float* dH, int64_t ldh, void* stream);

/* Sparse weight gradient of the MaxPool MLP when its input rows need no gradient (layer 0):
 *   slab[z][f, c] = sum_{g in slice z} v[g, c] * X[ids[g*s + argmax[g, c]], f],   v = d_pooled_masked
 * (d_pooled_masked = d_pooled * (pooled > 0), e.g. from gs_act_bwd).  Equals X[ids]^T · dH of the dense path
 * (gs_segment_max_bwd + gs_dense_wgrad) without materialising dH = [n*s, hidden]; ~s times fewer flops.
 * Needs 16*s <= 4*min(512, round_up(hidden,64)); returns GS_ENOTSUP otherwise.  slabs: [n_slabs, d, ld_slab]. */
int gs_maxpool_sparse_wgrad(const float* X, int64_t ldx, const int32_t* ids, int64_t n_groups, int32_t s, int32_t d,
                            const int32_t* argmax, int64_t lda, const float* d_pooled_masked, int64_t ldd,
                            int32_t hidden, int32_t n_slabs, float* slabs, int64_t ld_slab, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5  supervised head         replaces supervised_models.py:85 (l2_normalize), :111-118 (losses),
 *                             :122-126 (predict)
 * ------------------------------------------------------------------------------------------- */

/* y = x * rsqrt(max(sum(x^2), 1e-12)) per row; inv_norm[n] saved for backward. */
int gs_l2norm_fwd(const float* x, int64_t ldx, int64_t n, int32_t d, float* y, int64_t ldy,
                  float* inv_norm, void* stream);
int gs_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* inv_norm,
                  int64_t n, int32_t d, float* dx, int64_t lddx, void* stream);

/* Fused classification loss forward+backward on logits [n, C] with float labels [n, C]
 * (one-hot or multi-hot, the reference's feed contract, minibatch.py:264-274):
 *   sigmoid_loss == 0: softmax CE, loss_rows[r] = -sum_c z log p;  dlogits = (p*sum(z) - z)/n
 *   sigmoid_loss != 0: loss_rows[r] = sum_c (max(x,0) - x z + log1p(exp(-|x|)))/C ;
 *                      dlogits = (sigmoid(x) - z)/(n*C)
 * preds = softmax(logits) or sigmoid(logits) (supervised_models.py:122-126).
 * mean(loss_rows) is the classification loss of :112-118.  preds / dlogits may be NULL. */
int gs_class_loss(const float* logits, int64_t ldl, const float* labels, int64_t ldlab,
                  int64_t n, int32_t C, int sigmoid_loss,
                  float* loss_rows, float* preds, int64_t ldp, float* dlogits, int64_t lddl,
                  void* stream);

/* Fused head, forward AND backward in one launch (one wave per row, W staged in LDS):
 *   y = l2_normalize(x) (:85);  logits = y·W + b (:88-92);  loss_rows / preds / dlogits as gs_class_loss;
 *   dx = l2norm_bwd(dlogits·W^T)   (dx may be NULL for evaluation).  Supports d in {64,128,256,512}, C <= 128 and
 *   W fitting LDS; returns GS_ENOTSUP otherwise (callers then use the unfused kernels above). */
int gs_head_fwd_bwd(const float* x, int64_t ldx, int64_t n, int32_t d, const float* W, int64_t ldw,
                    const float* bias, const float* labels, int64_t ldlab, int32_t C, int sigmoid_loss,
                    float* y, int64_t ldy, float* logits, int64_t ldlo, float* preds, int64_t ldp,
                    float* dlogits, int64_t lddl, float* loss_rows, float* dx, int64_t lddx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * N3  unsupervised objective   replaces models.py:336-343 (negative sampler), minibatch.py:113-132 (pair batches),
 *                              prediction.py:68-110 (BipartiteEdgePredLayer xent loss), models.py:393-405 (MRR)
 * ------------------------------------------------------------------------------------------- */

/* ids_out = [batch1 (B) | batch2 (B) | negatives (n_neg)]:
 *   pairs != NULL: batch1[i], batch2[i] = pairs[(*cursor_dev + i) % n_pairs]   (int32 [n_pairs, 2])
 *   cdf   != NULL: negatives drawn from the fixed unigram distribution ~ degree^0.75 with replacement
 *                  (tf.nn.fixed_unigram_candidate_sampler, unique=False): cdf is uint32 [n_nodes],
 *                  cdf[i] = floor(2^32 * P(node <= i)); draw = first i with cdf[i] > hash32(seed, *clock_dev, slot_offset + slot):
 *                  slot_offset = this rank's first global root (data-parallel ranks draw different negatives). */
int gs_unsup_stage(const int32_t* pairs, int64_t n_pairs, const uint64_t* cursor_dev, int64_t B,
                   const uint32_t* cdf, int64_t n_nodes, int32_t n_neg, uint64_t seed, const uint64_t* clock_dev,
                   int64_t slot_offset, int32_t* ids_out, void* stream);

/* Skip-gram cross-entropy head on the l2-normalised embeddings Y = [outputs1 (B) | outputs2 (B) | neg_outputs (n_neg)]:
 *   loss_rows[i] = xent(1, <o1_i,o2_i>) + neg_weight * sum_j xent(0, <o1_i,neg_j>)         (prediction.py:102-110)
 *   rr_rows[i]   = 1 / (1 + #{j : <o1_i,neg_j> >= <o1_i,o2_i>})                            (models.py:399-404)
 *   aff_all[i]   = [neg_aff_i (n_neg) | aff_i]  (optional, models.py:400)
 *   dY rows [0,2B) = scale * dLoss/dY;  the negatives' gradient arrives as ceil(B/4) slabs [n_neg, d] in neg_slabs
 *   (sum them with gs_reduce_slabs into dY rows [2B, 2B+n_neg)).  scale = 1/batch_size (models.py:378).
 * d in {64,128,256,512}; 5*n_neg*d*4 bytes must fit LDS. */
int gs_linkpred_fwd_bwd(const float* Y, int64_t ldy, int64_t B, int32_t d, int32_t n_neg, float neg_weight, float scale,
                        float* loss_rows, float* rr_rows, float* aff_all, int64_t ld_aff,
                        float* dY, int64_t lddy, float* neg_slabs, int32_t* n_slabs_out_host, void* stream);

/* The same objective FUSED with the normalisation on both sides (one launch + a 20-workgroup one instead of
 * l2norm_fwd | linkpred | reduce_slabs | l2norm_bwd):  Z [2B + n_neg, d] are the RAW aggregator outputs,
 *   Y = l2_normalize(Z) (models.py:368-370) is written (outputs1 / outputs2 / neg_outputs),
 *   loss_rows / rr_rows / aff_all as gs_linkpred_fwd_bwd on Y,
 *   dZ [2B + n_neg, d] = scale * dLoss/dZ (the gradient carried back through the normalisation; the negatives' rows are
 *   summed from ceil(B/4) per-workgroup slabs in neg_slabs [ceil(B/4), n_neg, d] in a fixed order). */
int gs_linkpred_norm_fwd_bwd(const float* Z, int64_t ldz, int64_t B, int32_t d, int32_t n_neg, float neg_weight, float scale,
                             float* Y, int64_t ldy, float* loss_rows, float* rr_rows, float* aff_all, int64_t ld_aff,
                             float* dZ, int64_t lddz, float* neg_slabs, void* stream);
/* ... and the step epilogue of gs_finalize_step2 as one more workgroup of its second launch (it only needs loss_rows /
 * rr_rows): loss_out[0] (+)= mean(loss_rows) (models.py:378), mrr_out[0] = mean(rr_rows) (models.py:404), then the device
 * counters c0..c2 (nullable) advance by d0..d2 -- one launch less per unsupervised step. */
int gs_linkpred_norm_fwd_bwd_step(const float* Z, int64_t ldz, int64_t B, int32_t d, int32_t n_neg, float neg_weight,
                                  float scale, float* Y, int64_t ldy, float* loss_rows, float* rr_rows, float* aff_all,
                                  int64_t ld_aff, float* dZ, int64_t lddz, float* neg_slabs, float* loss_out, int accumulate,
                                  float* mrr_out, uint64_t* c0, uint64_t d0, uint64_t* c1, uint64_t d1, uint64_t* c2,
                                  uint64_t d2, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K6  optimizer               replaces supervised_models.py:95-99 (clip_by_value +-5, Adam) and the
 *                             weight-decay terms :104-108
 * ------------------------------------------------------------------------------------------- */

/* grad[i] = sum_{z<n_slabs} slabs[z*slab_stride + i] + weight_decay * w[i],  i < count.
 * Rows of the slab are [rows, ld_slab] and the destination is dense [rows, cols]. */
int gs_reduce_slabs(const float* slabs, int32_t n_slabs, int64_t slab_stride, int32_t rows, int32_t cols,
                    int64_t ld_slab, float weight_decay, const float* w, int64_t ldw,
                    float* grad, int64_t ldg, int accumulate, void* stream);

/* TF-1.x Adam on a flat buffer with elementwise clip (tf.clip_by_value, supervised_models.py:96):
 *   g = clamp(grad*grad_scale, -clip, clip) (clip <= 0 disables);  t = *step_dev + 1
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
 *   p -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)          (epsilon OUTSIDE the sqrt) */
int gs_adam_step(float* p, const float* grad, float* m, float* v, int64_t count,
                 float lr, float beta1, float beta2, float eps, float clip, float grad_scale,
                 const uint64_t* step_dev, int32_t step_offset, void* stream);

/* Gradient finalisation over the whole flat parameter buffer in ONE launch:
 *   grads[i] = sum_{z < n_slabs(var)} slabs_var[z*size_var + (i - offset_var)] + (decay_var ? weight_decay*params[i] : 0)
 * and, when fuse_adam != 0, the clip + Adam update of gs_adam_step on the same element (single-GPU step).
 * vars_host: HOST array of per-variable descriptors (consumed at call time). */
typedef struct gs_var_desc {
    int64_t offset, size;  /* segment of the flat buffers, in floats (size includes ld padding) */
    float* slabs;
    int32_t n_slabs, decay;
    int32_t clear;         /* != 0: slab 0 is an atomically accumulated gradient (gs_scatter_add_rows); it is zeroed
                              after being read so that the next backward pass starts from 0 (needs n_slabs == 1) */
    int32_t reserved_;
} gs_var_desc;
/* step_offset: Adam's t = *step_dev + step_offset (1 when the counter is advanced AFTER this launch, 0 when an earlier
 * launch of the step -- gs_sage_tail_fwd_bwd -- already advanced it).  loss_rows (nullable): the step's scalar loss
 * loss_out[0] (+)= loss_scale * sum(loss_rows[0:loss_n]) is formed by workgroup 0 in a fixed order (replaces the
 * gs_finalize_step launch of a training step). */
int gs_flat_reduce_adam(const gs_var_desc* vars_host, int32_t n_vars, float* params, float* grads, float* m, float* v,
                        int64_t total, float weight_decay, int fuse_adam, float lr, float beta1, float beta2,
                        float eps, float clip, float grad_scale, const uint64_t* step_dev, int32_t step_offset,
                        const float* loss_rows, int64_t loss_n, float loss_scale, float* loss_out, int loss_accumulate,
                        void* stream);

/* gs_flat_reduce_adam + riders in ONE launch: the fan-out sampler of a LATER mini-batch (sampler_host, nullable: the
 * arguments of gs_sample_fanout_csr in a struct) and up to 6 gather+mean jobs of the NEXT mini-batch (as
 * gs_sage_dense_fwd_cogather).  The optimizer launch is a short latency-bound pass over ~1 MB of parameters; the
 * sampler (a chain of dependent round trips with almost no work) hides completely under it and the gather waves use
 * the idle wave slots.  The per-root id count of every kept hop must be <= 512 (else GS_ENOTSUP: launch
 * gs_sample_fanout_csr on its own). */
typedef struct gs_fanout_desc {
    const int64_t* rowptr; const int32_t* col; int64_t n_nodes;
    int32_t* ids_all; int64_t B;
    uint64_t seed, step; const uint64_t* step_dev;
    int64_t root_offset;
    const int32_t* order; int64_t n_order; const uint64_t* cursor_dev;
    const float* label_table; int64_t ld_table;
    float* labels_out; int64_t ld_out;
    int64_t offsets[4];
    int32_t fan[3];
    int32_t pad_id, n_hops, C;
    uint32_t hop0;
    int32_t law, max_degree;      /* GS_LAW_*, see gs_sample_uniform_csr */
    /* optional unsupervised root staging (pairs != NULL, exclusive with order): the roots are
     *   [pairs[e][0] (n_pair_roots) | pairs[e][1] (n_pair_roots) | n_neg negatives],  e = (*cursor_dev + i) % n_pairs,
     * B == 2 * n_pair_roots + n_neg; negative t = first node whose cdf exceeds the 32-bit draw of gs_unsup_stage
     * (neg_seed, *step_dev, t) -- same draws bit for bit; guide (nullable, 2^guide_bits + 1 entries: guide[b] = first
     * index with cdf > b << (32 - guide_bits)) only shortens the binary search. */
    const int32_t* pairs; int64_t n_pairs, n_pair_roots;
    const uint32_t* cdf; const int32_t* guide; int64_t n_cdf;
    int32_t n_neg, guide_bits;
    uint64_t neg_seed;
    /* GS_LAW_REFERENCE only, nullable: the law's virtual padded table materialised by gs_build_padded_table
     * ([n_nodes + 1, max_degree] int32, same seed / max_degree): a draw becomes one lookup table[id][column].  Same ids,
     * bit for bit, as without it. */
    const int32_t* padded_table;
    /* GS_LAW_REFERENCE only: the roots [0, seg_begin[0]) | [seg_begin[0], seg_begin[1]) | [seg_begin[1], B) come from
     * DIFFERENT sampler calls of the reference -- SampleAndAggregate._build runs sample(batch1), sample(batch2) and
     * sample(neg_samples) (models.py:347-357), each with its own tf.random_shuffle per hop (neigh_samplers.py:27) -- so
     * segment g uses the call ids hop0 + g * n_hops + h: six independent column permutations per unsupervised step.
     * {0, 0} = one call (supervised); with pair staging the boundaries default to {n_pair_roots, 2 * n_pair_roots}.
     * The staged negatives are keyed by root_offset + t, so data-parallel ranks draw different negatives. */
    int64_t seg_begin[2];
} gs_fanout_desc;
/* The padded adjacency table of minibatch.py:227-245 under GS_LAW_REFERENCE's keyed law, built ON THE DEVICE from the CSR:
 * table[v][c] = the c-th entry of node v's (frozen) padded row -- a keyed sample without replacement of max_degree
 * neighbors when deg > max_degree, max_degree keyed draws with replacement when deg < max_degree, the list itself when
 * equal; pad_id for degree-0 nodes and for row n_nodes.  table_out: [(n_nodes + 1) * max_degree] int32. */
int gs_build_padded_table(const int64_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t pad_id, int32_t max_degree,
                          uint64_t seed, int32_t* table_out, void* stream);
/* gs_sample_fanout_csr from a descriptor (incl. the unsupervised root staging and the materialised table). */
int gs_sample_fanout_desc(const gs_fanout_desc* desc_host, void* stream);
int gs_flat_reduce_adam_sample(const gs_var_desc* vars_host, int32_t n_vars, float* params, float* grads, float* m, float* v,
                               int64_t total, float weight_decay, int fuse_adam, float lr, float beta1, float beta2,
                               float eps, float clip, float grad_scale, const uint64_t* step_dev, int32_t step_offset,
                               const float* loss_rows, int64_t loss_n, float loss_scale, float* loss_out, int loss_accumulate,
                               const gs_fanout_desc* sampler_host, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);
/* gs_dense_wgrad_grouped_tiled3 with the fan-out sampler of a LATER mini-batch riding in the launch (ABI 10; sampler_host as gs_sample_fanout_desc takes
 * it, nullable; per-root id count of every kept hop <= 512, else GS_ENOTSUP).  The sampler is a chain of dependent round trips:
 * as a rider of the optimizer launch (gs_flat_reduce_adam_sample) it is what that launch waits for, here it ends long before
 * the contraction.  No problem may gather its rows (a_idx) through the id buffer that sampler fills -- GS_EINVAL; the private
 * copy gs_tail_desc.ids_copy_* makes is what the step's weight gradients read instead.  Same draws as the standalone launch. */
int gs_dense_wgrad_grouped_tiled3_sample(const gs_wgrad_desc* descs_host, int32_t n_desc, const gs_gather_desc* jobs_host,
                                         int32_t n_jobs, const gs_fanout_desc* sampler_host, void* stream);

/* Fused tail of the supervised two-layer GraphSAGE-mean model: layer 1 (MeanAggregator._call on the layer-0 outputs,
 * aggregators.py:43-64, identity act: last layer, models.py:307-310), l2_normalize + Dense head + loss/preds
 * (supervised_models.py:85-126) and their gradients down to dLoss/d(layer-0 pre-activations), ONE launch.  Every batch
 * row is independent given the weights: a workgroup takes 16 rows through the whole chain (fp32 MFMA 16x16x4).
 *   h0      [n + n*s, d_in]  relu outputs of layer 0: rows 0..n-1 = batch nodes, row n + i*s + j = j-th sample of node i
 *   means   [n, d_in], z [n, 2*out_dim] (pre-normalisation), y [n, 2*out_dim] (= outputs1), logits / preds / dlogits
 *           [n, C], loss_rows [n]: written (inputs of the weight-gradient launch and model outputs)
 *   train != 0: dz [n, 2*out_dim] = dLoss/dz and d_h0 [n + n*s, d_in] = relu'(h0) * dLoss/dh0 are written too
 *   c0..c2 (nullable device counters) are advanced by d0..d2 at the end of the launch.
 * Supported: concat, no aggregator bias, s <= 11, d_in in {128, 256}, out_dim in {64, 128}, C <= 128
 * (gs_sage_tail_supported);
 * anything else returns GS_ENOTSUP and the caller uses the per-operator entry points. */
typedef struct gs_tail_desc {
    const float* h0; int64_t ldh; int64_t n;
    const float* W_self; int64_t ldws;
    const float* W_neigh; int64_t ldwn;
    const float* W_head; int64_t ldwh;
    const float* b_head;
    const float* labels; int64_t ldlab;
    float* means; int64_t ldm;
    float* z; int64_t ldz;
    float* y; int64_t ldy;
    float* logits; int64_t ldlo;
    float* preds; int64_t ldp;
    float* dlogits; int64_t lddl;
    float* loss_rows;
    float* dz; int64_t lddz;
    float* d_h0; int64_t lddh;
    uint64_t* c0; uint64_t d0;
    uint64_t* c1; uint64_t d1;
    uint64_t* c2; uint64_t d2;
    int32_t s, d_in, out_dim, C, sigmoid, train;
    uint32_t* sync;        /* [2 * G + 2 + 64 * G * out_dim] device words, G = ceil(n / 16), 8-byte aligned, zero-initialised ONCE by
                              the caller, private to one stream and to ONE n (ABI 10; in-kernel hand-over of the layer-1
                              pre-activations from the helper workgroups to the row-group workgroups): words [G, 2 G) = the
                              groups' launch epochs, word 2 G = an error word the caller should check when it fetches results
                              (bit 0: a row-group workgroup gave up waiting -- the wait is bounded), and from word 2 G + 2 on
                              G x 16 x 2*out_dim eight-byte granules {z element, epoch tag}.  May be NULL when z_ready != 0. */
    int32_t z_ready;       /* != 0: z and means were written by gs_sage_tail_z on this stream (split form): the launch has no
                              helper workgroups, no in-kernel hand-over and needs no sync buffer */
    int32_t gcn;           /* != 0: GCNAggregator form of layer 1 (aggregators.py:101-116; gs_sage_tail_fwd_bwd and gs_sage_tail_z):
                              ONE weight matrix W [d_in, 2*out_dim] passed as W_self = W, W_neigh = W + out_dim (same ld); both
                              column halves of z contract the mean over {neighbors} U {self} = (sum_j h_neigh_j + h_self)/(s+1),
                              which `means` receives; d_h0 rows (self and neighbors) = relu'(h0) * (dz . W^T) / (s + 1) */
    const int32_t* ids_copy_src; /* optional (ids_copy_n > 0; ABI 10): the launch's helper workgroups also copy ids_copy_n int32 from */
    int32_t* ids_copy_dst;       /* src to dst (gs_sage_tail_fwd_bwd without z_ready, or gs_sage_tail_z): the private copy of the step's */
    int64_t ids_copy_n;          /* node ids that the weight gradients gather through when the NEXT-next step's sampler rides in their launch */
} gs_tail_desc;
int gs_sage_tail_supported(int32_t d_in, int32_t out_dim, int32_t C);

/* Fused tail of the UNSUPERVISED two-layer mean model (models.py:362-405, prediction.py:68-110), two launches:
 *   gs_linkpred_tail      layer 1 (z helpers, as gs_sage_tail_fwd_bwd) -> main workgroups of 8 PAIRS each: l2_normalize
 *                         (models.py:368-370), affinities / xent loss / MRR rank (prediction.py:102-110, models.py:393-405), dY,
 *                         dz = l2norm'(dY) of the 2 B pair rows, [d_self | d_means] = dz . W^T, d_h0 = relu'(h0) * (...) of
 *                         their rows, and one slab [n_neg, 2*out_dim] per main of the negatives' gradient w.r.t. their
 *                         normalised rows; + gather jobs riding as extra workgroups (a long, thin launch: the rest of the
 *                         chip streams the next step's gather at the full HBM rate);
 *   gs_linkpred_tail_neg  the negatives' rows: slabs summed in main order -> dz -> d_h0 of their rows; + the step epilogue
 *                         (loss_out = mean(loss_rows) (+= if accumulate), mrr_out = mean(rr_rows), counters c0..c2 advanced
 *                         by d0..d2; loss_out == NULL: no epilogue) + the commit of the hand-over state + gather jobs riding (nine
 *                         workgroups of dependent round trips leave the chip free).  MUST follow every
 *                         gs_linkpred_tail on the same stream (train == 0: only the epilogue / commit run).
 * Rows: h0 [n + n*s, d_in] with n = 2 B + n_neg roots [batch1 | batch2 | negatives] (row n + i*s + j = j-th sample of root
 * i); z / y / dz [n, 2*out_dim]; means [n, d_in]; loss_rows / rr_rows [B]; aff_all [B, n_neg + 1] = [neg_aff | aff]
 * (nullable); neg_slabs [ceil(B / 8)][n_neg][2*out_dim]; sync: 2 * (ceil(B / 8) + ceil(n_neg / 16)) + 2 device words,
 * zero-initialised once, private to one stream (word 2 G = error flags as gs_tail_desc.sync).
 * Supported (gs_linkpred_tail_supported): d_in in {128, 256}, out_dim in {64, 128}, n_neg <= 32, s <= 11; concat, no bias. */
typedef struct gs_lp_tail_desc {
    const float* h0; int64_t ldh;
    int64_t B; int32_t n_neg, s, d_in, out_dim, train;
    const float* W_self; int64_t ldws;
    const float* W_neigh; int64_t ldwn;
    float* means; int64_t ldm;
    float* z; int64_t ldz;
    float* y; int64_t ldy;
    float* dz; int64_t lddz;
    float* d_h0; int64_t lddh;
    float* loss_rows; float* rr_rows;
    float* aff_all; int64_t ld_aff;
    float* neg_slabs;
    float neg_weight, scale;
    uint32_t* sync;
} gs_lp_tail_desc;
int gs_linkpred_tail_supported(int32_t d_in, int32_t out_dim, int32_t n_neg);
int gs_linkpred_tail(const gs_lp_tail_desc* desc_host, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);
int gs_linkpred_tail_neg(const gs_lp_tail_desc* desc_host, float* loss_out, int accumulate, float* mrr_out, uint64_t* c0,
                         uint64_t d0, uint64_t* c1, uint64_t d1, uint64_t* c2, uint64_t d2, const gs_gather_desc* jobs_host,
                         int32_t n_jobs, void* stream);
/* Split form of the fused tail, first launch: the layer-1 pre-activations z = [h_self . W_self | mean(h_neigh) . W_neigh]
 * and the neighbor means of the descriptor (its head / label / gradient fields are ignored) as a LEAN kernel -- 4 x 64-column
 * helper workgroups per 16 rows, <= 128 VGPRs, 49 KB of LDS -- so that the gather jobs riding in the launch stream at the
 * full HBM rate, also on the CUs that run helpers (in the one-launch form every workgroup inherits the row-group
 * workgroups' 246 VGPRs / 88 KB and a CU holds one of them).  Follow with gs_sage_tail_fwd_bwd(desc with z_ready = 1).
 * Results are bit-identical to the one-launch form. */
int gs_sage_tail_z(const gs_tail_desc* desc_host, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);
/* The matching backward half as a launch of its own (the fused kernel's last two phases): from dz = dLoss/dz [n, 2*out_dim],
 *   [d_self | d_means] = [dz[:, :out_dim] . W_self^T | dz[:, out_dim:] . W_neigh^T]      (aggregators.py:51-58 backward)
 *   d_h0 [n + n*s, d_in] = relu'(h0) * (d_self on row i, d_means / s on rows n + i*s + j)  (aggregators.py:48, :64)
 * instead of a small GEMM + gs_input_grad_pull.  Descriptor fields used: h0, W_self, W_neigh, dz, d_h0, n, s, d_in,
 * out_dim.  gs_sage_tail_z / gs_sage_tail_dh0 serve last mean layers of models that do not take the fused tail. */
int gs_sage_tail_dh0(const gs_tail_desc* desc_host, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);
/* jobs_host / n_jobs (0..6): gather+mean jobs of the NEXT step co-scheduled in the launch (as gs_sage_dense_fwd_cogather):
 * the tail keeps n/16 CUs busy, the rest of the chip streams the gather meanwhile. */
int gs_sage_tail_fwd_bwd(const gs_tail_desc* desc_host, const gs_gather_desc* jobs_host, int32_t n_jobs, void* stream);

/* Inverted dropout tf.nn.dropout(x, keep_prob = 1 - rate) (aggregators.py:46-47,104-105; layers.py:107): kept
 * elements are scaled by 1/keep_prob.  The keep mask is a counter hash of (seed, *clock_dev, site, row0 + row,
 * column), so forward and backward regenerate it and a replayed hipGraph draws new masks every step (clock_dev is the
 * device step counter).  A NULL descriptor or rate == 0 means "off". */
typedef struct gs_dropout {
    uint64_t seed;
    const uint64_t* clock_dev; /* device pointer, nullable (= 0) */
    uint32_t site;             /* distinct per dropout call site of a step */
    float rate;                /* in [0, 1) */
    int64_t row0;              /* global index of this call's first row */
    /* Parity-test hook (NULL in production): when set, the keep mask is READ from this device buffer instead of the
     * counter hash -- element (row0 + i, c) is kept iff keep_bits[(row0 + i) * keep_ld + c] != 0 -- so that a test can
     * feed the masks the reference run's tf.nn.dropout drew (aggregators.py:46-47, layers.py:107), like the sampler's
     * permutations.  4-byte aligned base, keep_ld % 4 == 0 and >= round_up(d, 4). */
    const uint8_t* keep_bits;
    int64_t keep_ld;
} gs_dropout;
/* out[i, :] = mask(row0 + i, :) * X[ids ? ids[i] : i, :] / keep_prob     (in place allowed when ids == NULL).
 * The same call is the backward of itself (X = upstream gradient). */
int gs_dropout_rows(const float* X, int64_t ldx, const int32_t* ids, int64_t n, int32_t d, const gs_dropout* drop,
                    float* out, int64_t ldo, void* stream);
/* K2 with dropout applied to every gathered neighbor row BEFORE the mean (aggregators.py:46-48):
 *   mean[i, :] = (1/s) * sum_j mask(row0 + i*s + j, :) * X[idx[i*s + j], :] / keep_prob
 * (self_src as in gs_gather_mean_fwd; the caller passes already-dropped self rows). */
int gs_gather_mean_dropout_fwd(const float* X, int64_t ldx, const int32_t* idx, int64_t n, int32_t s, int32_t d,
                               const float* self_src, int64_t ld_self, const int32_t* self_idx, float* mean,
                               int64_t ldm, const gs_dropout* drop, void* stream);

/* Trainable identity features ("node_embeddings", models.py:229-240 / supervised_models.py:49-60): gradient of the
 * layer-0 row gathers w.r.t. the leading `cols` columns of the gathered table, accumulated with fp32 atomics
 * (tf.gradients of embedding_lookup: IndexedSlices summed per id):
 *   table[ids[i*s + j], c] += scale * d[i, c]      i < n, j < s, c < cols */
int gs_scatter_add_rows(const float* d, int64_t ldd, int64_t n, int32_t s, int32_t cols, float scale,
                        const int32_t* ids, float* table, int64_t ldt, void* stream);
/* dst[r, 0:cols] = src[r, 0:cols]: refreshes the embedding columns of the combined feature table (the
 * tf.concat([embeds, features], axis=1) of models.py:240, kept materialised) after an optimizer step. */
int gs_copy_cols(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int32_t cols, void* stream);

/* Input gradient of one aggregator layer w.r.t. the previous layer's hidden rows, in ONE launch ("pull" form: every
 * output row sums the contributions it receives, so no accumulation order is involved):
 *   out[r, :] = relu'(mask_y[r, :]) * ( [r < n_self] d_self[r, :]
 *                                       + sum_k [row0_k <= r < row0_k + n_k*s_k] scale_k * src_k[(r - row0_k) / s_k, :] )
 * Segment k is the reverse of a segmented mean over s_k rows (scale 1/s_k), of a GCN self term (s = 1, scale
 * 1/(s+1)) or of a plain row copy (s = 1, scale 1).  Replaces gs_act_bwd + one gs_mean_bwd per hop. */
#define GS_PULL_MAX 6
typedef struct gs_pull_desc {
    const float* d_self;   /* [n_self, ld_self], nullable */
    int64_t ld_self, n_self;
    int32_t n_seg, d;
    const float* src[GS_PULL_MAX];
    int64_t ld_src[GS_PULL_MAX], row0[GS_PULL_MAX], n[GS_PULL_MAX];
    int32_t s[GS_PULL_MAX];
    float scale[GS_PULL_MAX];
    const float* mask_y;   /* [rows, ldy] activations of the previous layer (relu mask), nullable */
    int64_t ldy;
    float* out;            /* [rows, ldo] */
    int64_t ldo, rows;
} gs_pull_desc;
int gs_input_grad_pull(const gs_pull_desc* desc_host, void* stream);

/* gs_finalize_step with a second mean in the same launch: aux_out[0] = aux_scale * sum(aux_rows[0:n]) (the unsupervised
 * model's mrr, models.py:404). */
int gs_finalize_step2(const float* loss_rows, int64_t n, float scale, float* loss_out, int accumulate,
                      const float* aux_rows, float aux_scale, float* aux_out, uint64_t* c0, uint64_t d0, uint64_t* c1,
                      uint64_t d1, uint64_t* c2, uint64_t d2, void* stream);

/* Up to three device counters advanced by one launch (cursor / sampler clock / optimizer step). */
int gs_advance_counters(uint64_t* c0, uint64_t d0, uint64_t* c1, uint64_t d1, uint64_t* c2, uint64_t d2, void* stream);

/* Step epilogue in one launch: loss_out[0] (+)= scale * sum(loss_rows[0:n]) (skipped if loss_rows == NULL), then
 * the (non-NULL) counters are advanced. */
int gs_finalize_step(const float* loss_rows, int64_t n, float scale, float* loss_out, int accumulate,
                     uint64_t* c0, uint64_t d0, uint64_t* c1, uint64_t d1, uint64_t* c2, uint64_t d2, void* stream);

/* batch[i] = order[(*cursor + i) % n_order];  labels_out[i, :C] = label_table[batch[i], :C]   (one launch;
 * replaces minibatch.py:264-274 + 302-307 for the device-resident epoch). */
int gs_stage_batch(const int32_t* order, int64_t n_order, const uint64_t* cursor_dev, int64_t n, int32_t* batch,
                   const float* label_table, int64_t ld_table, int32_t C, float* labels_out, int64_t ld_out,
                   void* stream);

/* out[0] = scale * sum_{i<count} x[i]   (single block, deterministic order) */
int gs_sum_scaled(const float* x, int64_t count, float scale, float* out, int accumulate, void* stream);
/* out[0] (+)= scale * sum x[i]^2  -- weight_decay * tf.nn.l2_loss(var), supervised_models.py:106 */
int gs_sumsq_scaled(const float* x, int64_t count, float scale, float* out, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * C1  gradient all-reduce over RCCL / xGMI (data-parallel training; the reference is single-device,
 * supervised_train.py:55-59).  One process per GPU; ONE in-place sum over the flat fp32 gradient buffer per step,
 * enqueued on `stream` and capturable into the step's hipGraph.  RCCL is bound at run time (dlopen) -- the copy
 * already mapped in the process is reused.
 *   gs_comm_unique_id : rank 0 fills a >= 128-byte HOST buffer which the host code ships to every rank
 *                       (torch.distributed / MPI / a file -- not this library's business)
 *   gs_comm_init_rank : collective over all ranks; uses the calling thread's current HIP device
 *   gs_comm_available : GS_OK iff RCCL can be bound in this process -- ranks agree on this over their bootstrap
 *                       transport BEFORE any of them enters the collective gs_comm_init_rank
 *   gs_comm_count     : ncclCommCount of the communicator (a run can report the size RCCL really has)
 * ------------------------------------------------------------------------------------------- */
#define GS_COMM_ID_BYTES 128
int gs_comm_available(void);
int gs_comm_count(void* comm, int32_t* n_ranks_out_host);
int gs_comm_unique_id(void* id_out_host, int32_t len);
int gs_comm_init_rank(void** comm_out, int32_t nranks, int32_t rank, const void* id_host, int32_t len);
int gs_comm_allreduce_sum_f32(void* comm, float* buf, int64_t count, void* stream);
int gs_comm_destroy(void* comm);

/* ---------------------------------------------------------------------------------------------
 * C2: the same exchange as direct peer stores over xGMI (opt-in; gs_comm_* stays the default).  The flat gradient is
 * 0.9 MB: instead of a ring (2(N-1) latency-bound hops) every rank stores slice p of its gradient straight into rank
 * p's window (reduce-scatter, N-1 links in parallel), rank p sums the N copies IN RANK ORDER and stores the sum into
 * every rank's window (all-gather): one hop out, one hop back, ONE kernel launch on `stream`, capturable into the
 * step's hipGraph.  Windows are uncached device memory owned by this library and shared with hipIpcMemHandle.
 *   gs_peer_create        : allocates this rank's window for an n_floats buffer (chunks: workgroups per peer, 0 = sized
 *                           so that a thread moves ~2 float4 per pass, <= 256 workgroups in all; spin_limit: polls before
 *                           a wait gives up, 0 = 2^24, of the order of ten seconds)
 *   gs_peer_export/attach : the 64-byte IPC handle of the own window / maps rank `peer_rank`'s window (the host code
 *                           ships handles between processes -- torch.distributed / MPI / a file)
 *   gs_peer_attach_local  : another rank of the SAME process, by object (several streams or devices in one process)
 *   gs_peer_allreduce_sum_f32 : in-place sum over ranks; every rank ends with identical bits
 *   gs_peer_status        : epochs completed and the error word (0 ok; bit 0: a peer's copies never arrived, bit 1: a
 *                           reduced slice never arrived, bits 8..: the ranks waited for in vain) -- every device-side
 *                           wait is bounded, a missing peer cannot hang the GPU; the error is sticky (later exchanges
 *                           return at once, the buffer untouched): re-create the windows
 * ------------------------------------------------------------------------------------------- */
#define GS_PEER_HANDLE_BYTES 64
int gs_peer_create(int64_t n_floats, int32_t world, int32_t rank, int32_t chunks, int64_t spin_limit, void** peer_out);
int gs_peer_export(void* peer, void* handle_out_host, int32_t len);
int gs_peer_attach(void* peer, int32_t peer_rank, const void* handle_host, int32_t len);
int gs_peer_attach_local(void* peer, void* other_peer);
int gs_peer_allreduce_sum_f32(void* peer, float* buf, int64_t count, void* stream);
int gs_peer_status(void* peer, int64_t* epoch_out_host, int32_t* error_out_host);
int gs_peer_destroy(void* peer);

/* ---------------------------------------------------------------------------------------------
 * hipGraph helpers: the per-step kernel chain is captured once and replayed (no tracing compiler).
 * ------------------------------------------------------------------------------------------- */
int gs_stream_create(void** stream_out);
/* Diagnostics: one wave sleeping `us` microseconds on `stream` (the stand-in for a latency-bound collective when the
 * data-parallel step schedule is probed on one GPU; bench.py GS_PROBE_DP_SCHEDULE). */
int gs_spin_us(float us, void* stream);
int gs_stream_destroy(void* stream);
int gs_stream_sync(void* stream);
int gs_capture_begin(void* stream);
int gs_capture_end(void* stream, void** graph_exec_out);
int gs_graph_launch(void* graph_exec, void* stream);
int gs_graph_destroy(void* graph_exec);
/* hipEvent timing on `stream` (torch.cuda.Event only sees torch's current stream). */
int gs_event_create(void** ev_out);
int gs_event_record(void* ev, void* stream);
/* Makes `stream` wait for `ev` (fork/join of a second stream; inside a capture this becomes a graph edge). */
int gs_stream_wait_event(void* stream, void* ev);
int gs_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out_host); /* synchronises ev_stop */
int gs_event_destroy(void* ev);

/* ---------------------------------------------------------------------------------------------
 * Host-side graph ingestion (C++, multithreaded): edge list -> CSR.  Replaces the networkx loops of
 * minibatch.py:227-259 for the CSR engine (N2 "next" row).  All pointers are HOST pointers.
 * keep_mask_host (nullable, per edge) drops edges (train_removed / val-test endpoints).
 * Adjacency lists come out sorted and de-duplicated (networkx.Graph semantics: one edge per node pair).
 * ------------------------------------------------------------------------------------------- */
int gs_build_csr_host(const int32_t* src_host, const int32_t* dst_host, const uint8_t* keep_mask_host,
                      int64_t n_edges, int64_t n_nodes, int symmetrize,
                      int64_t* rowptr_out_host /* [n_nodes+1] */, int32_t* col_out_host /* cap */,
                      int64_t col_capacity, int64_t* nnz_out_host);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHSAGE_AMD_H */
