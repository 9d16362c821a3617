// Device-side body of the fused multi-hop fan-out sampler, shared by the standalone kernel (gs_sample.hip) and the
// optimizer launch that carries the NEXT-next step's sampler along (gs_optim.hip: the sampler is five dependent memory
// round trips of almost no work, so it hides entirely under another launch).
#pragma once
#include "gs_common.h"

#define GS_MAX_HOPS 3

// ---- sampling laws (gs_sample_law in include/graphsage_amd.h) --------------------------------------------------------
// gs_perm_index(key, j, n): the j-th element of a keyed pseudo-random PERMUTATION of [0, n): an alternating (unbalanced)
// Feistel network on bits = ceil(log2 n) bits -- the value is split into a high part of floor(bits/2) and a low part of
// ceil(bits/2) bits; even rounds xor a 32-bit hash of the low part into the high part, odd rounds the other way round
// (every round is invertible, so the whole map is a permutation of [0, 2^bits)) -- with cycle walking: the map is
// re-applied until the value lands in [0, n) (it stays on the cycle of j, so it terminates; 2^bits < 2n, so fewer than
// two applications on average).  Rounds: 8 (12 for bits <= 6, 24 for bits <= 4: measured, pairs of outputs are
// chi-square uniform from there on).  All arithmetic is 32-bit (the sampler chain is latency-critical: 64-bit multiplies
// cost ~4x).  A pure function, so "s distinct columns" and "a frozen max_degree subset of a long neighbor list" need no
// state and no table.  Restated bit-for-bit in oracle/sampler_hash.py::perm_index.
__device__ __forceinline__ uint32_t gs_fmix32(uint32_t h) {      // murmur3 finalizer
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

__device__ __forceinline__ uint32_t gs_perm_index(uint64_t key, uint32_t j, uint32_t n) {
    if (n <= 1u) return 0u;
    int bits = 32 - __clz((int)(n - 1u));
    bits = bits < 2 ? 2 : bits;
    const int a = bits >> 1, b = bits - a;
    const uint32_t mA = (1u << a) - 1u, mB = (1u << b) - 1u;
    const int rounds = bits <= 4 ? 24 : (bits <= 6 ? 12 : 8);
    const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    uint32_t x = j;
    do {
        uint32_t L = x >> b, R = x & mB;
        for (int r = 0; r < rounds; r += 2) {
            L ^= ((gs_fmix32(R + k0 + (uint32_t)r * 0x9E3779B9u) ^ k1) >> 7) & mA;
            R ^= ((gs_fmix32(L + k0 + (uint32_t)(r + 1) * 0x9E3779B9u) ^ k1) >> 7) & mB;
        }
        x = (L << b) | R;
    } while (x >= n);
    return x;
}

struct SampleLaw {
    int32_t law;          // GS_LAW_IID | GS_LAW_REFERENCE | GS_LAW_DISTINCT
    int32_t max_degree;   // cap of the (virtual) padded table; 0 = uncapped (GS_LAW_DISTINCT only)
};

// key of node v's row of the VIRTUAL padded table (minibatch.py:227-245: built once, frozen for the run): a function
// of (seed, node id) only -- never of the step.
__device__ __forceinline__ uint64_t gs_table_key(uint64_t seed, int32_t v) {
    return gs_mix64(seed ^ 0x7AB1E5EEDull ^ ((uint64_t)(uint32_t)v * 0xD1342543DE82EF95ull));
}

// entry c (< M) of node v's virtual padded row -> position in its CSR neighbor list (deg > 0):
//   deg >  M : np.random.choice(neighbors, M, replace=False) (minibatch.py:240-241)  = first M of a keyed permutation
//   deg <  M : np.random.choice(neighbors, M, replace=True)  (:242-243)              = M frozen iid draws
//   deg == M : the list itself
__device__ __forceinline__ uint32_t gs_table_entry(uint64_t tkey, uint32_t c, uint32_t deg, uint32_t M) {
    if (deg > M) return gs_perm_index(tkey, c, deg);
    if (deg == M) return c;
    return (uint32_t)(((gs_mix64(tkey + (uint64_t)c) >> 32) * (uint64_t)deg) >> 32);
}

// column j of the call's column permutation under GS_LAW_REFERENCE (neigh_samplers.py:26-28: ONE permutation per call
// shared by all rows, first s columns).  callkey = hash of (seed, step, hop).
__device__ __forceinline__ uint32_t gs_call_column(uint64_t callkey, uint32_t j, uint32_t M) {
    return gs_perm_index(gs_mix64(callkey ^ 0xC0115ull), j, M);
}

// position in the neighbor list of slot j of global row `grow` (deg > 0).  cols: the call's columns precomputed by the
// caller (LDS), or nullptr.
__device__ __forceinline__ uint32_t gs_draw(const SampleLaw lw, uint64_t seed, uint64_t callkey, int64_t grow, uint32_t j,
                                            int32_t s, int32_t id, uint32_t deg, const int32_t* cols = nullptr) {
    if (lw.law == GS_LAW_REFERENCE) {
        const uint32_t M = (uint32_t)lw.max_degree;
        const uint32_t c = cols ? (uint32_t)cols[j] : gs_call_column(callkey, j, M);
        return gs_table_entry(gs_table_key(seed, id), c, deg, M);
    }
    const uint64_t rowkey = callkey + (uint64_t)grow * 0xD1342543DE82EF95ull;
    if (lw.law == GS_LAW_DISTINCT) {
        // without replacement whenever the (capped) list holds at least s entries, per-row independent
        const uint32_t M = (uint32_t)lw.max_degree;
        const uint32_t eff = (M > 0u && deg > M) ? M : deg;
        uint32_t c;
        if (eff >= (uint32_t)s) c = gs_perm_index(gs_mix64(rowkey), j, eff);
        else c = (uint32_t)(((gs_mix64(rowkey + j) >> 32) * (uint64_t)eff) >> 32);
        return eff != deg ? gs_perm_index(gs_table_key(seed, id), c, deg) : c;
    }
    return (uint32_t)(((gs_mix64(rowkey + j) >> 32) * (uint64_t)deg) >> 32);    // GS_LAW_IID: Lemire range map
}
#define GS_FANOUT_LDS 8192          // per-root ids of the kept hops, standalone kernel
#define GS_FANOUT_LDS_SMALL 512     // ... when the sampler rides in another launch (e.g. 10 for fan-out 25x10)
struct FanoutArgs {
    const int64_t* rowptr;
    const int32_t* col;
    int64_t n_nodes;
    int32_t pad_id;
    int32_t n_hops;
    int32_t fan[GS_MAX_HOPS];
    int64_t offsets[GS_MAX_HOPS + 1];  // start of each hop's ids inside ids_all (offsets[0] = roots)
    int32_t* ids_all;
    int64_t B;
    uint64_t seed, step;
    const uint64_t* step_dev;
    uint32_t hop0;
    int64_t root_offset;  // global index of this rank's first root (data-parallel invariance)
    SampleLaw law;
    // optional batch staging (order == nullptr -> roots are already in ids_all)
    const int32_t* order;
    int64_t n_order;
    const uint64_t* cursor;
    const float* label_table;
    int64_t ldt;
    int32_t C;
    float* labels_out;
    int64_t ldo;
    // optional unsupervised root staging (pairs != nullptr; minibatch.py:113-132 + models.py:336-343 on the device):
    // roots = [pairs[e][0] (n_pair_roots) | pairs[e][1] (n_pair_roots) | n_neg unigram negatives], e = (*cursor + i) % n_pairs;
    // negative t = first node whose cdf exceeds a 32-bit draw keyed by (neg_seed, sampler clock, t) -- the draws of
    // gs_unsup_stage, bit for bit (the guide table only narrows the binary search's starting interval)
    const int32_t* pairs;
    int64_t n_pairs, n_pair_roots;
    const uint32_t* cdf;
    const int32_t* guide;      // nullable: guide[b] = first index with cdf > (b << (32 - guide_bits)), 2^guide_bits + 1 entries
    int64_t n_cdf;
    int32_t n_neg, guide_bits;
    uint64_t neg_seed;
    // GS_LAW_REFERENCE only, nullable: the virtual padded table MATERIALISED ([n_nodes + 1, max_degree] int32, built once
    // by gs_build_padded_table from the same gs_table_entry): a draw is then ONE lookup table[id][column] -- no rowptr
    // pair, no permutation arithmetic, 4 cache lines per parent instead of ~16 scattered ones.  Same ids bit for bit.
    const int32_t* table;
    // GS_LAW_REFERENCE only: roots [0, seg1) | [seg1, seg2) | [seg2, B) belong to DIFFERENT sampler calls of the reference
    // (models.py:347-357: sample(batch1), sample(batch2), sample(neg_samples) each shuffle their own columns), so segment g
    // uses the call ids hop0 + g * n_hops + h.  seg1 == seg2 == B (one segment) otherwise.
    int64_t seg1, seg2;
};

#define GS_LAW_COLS 128      // per-call columns kept in LDS up to this fan-out (larger fan-outs compute them per slot)

// One workgroup (any size) per root `i`; lvl = two LDS fan-out buffers of CAP ints each, law_cols = LDS for the calls' columns
// (both handed in by the caller: a launch that carries the sampler as a rider gives it a piece of ITS LDS allocation).
#define GS_FANOUT_LDS_INTS(CAP) (2 * (CAP) + GS_MAX_HOPS * GS_LAW_COLS)
// PER_WAVE: one WAVE per root instead of one workgroup (a launch that carries the sampler as a rider packs four roots into a
// rider workgroup's slot: the chain is as long, the slots it holds are a quarter); lvl / law_cols are then the wave's own.
template <int CAP, bool PER_WAVE = false>
__device__ __forceinline__ void sample_fanout_root(const FanoutArgs& a, const int64_t i, int32_t (*lvl)[CAP],
                                                   int32_t (*law_cols)[GS_LAW_COLS]) {
    const int tid = PER_WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, nthr = PER_WAVE ? 64 : (int)blockDim.x;
    // workgroup barrier that orders LDS only: what crosses it here (lvl, law_cols) lives in LDS, and __syncthreads() would also
    // drain the global loads / stores in flight -- a round trip per barrier in a chain that is nothing but round trips.  (One
    // wave: its LDS operations execute in order; only the compiler has to keep them there.)
    auto lds_sync = [] {
        if (PER_WAVE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        } else {
#ifdef GS_SAMPLER_FULL_BARRIER    // diagnostics
            __syncthreads();
#else
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
        }
    };
    const uint64_t st = a.step + (a.step_dev ? *a.step_dev : 0ull);
    int32_t root = 0;
    if (a.pairs) {
        const uint64_t c = a.cursor ? *a.cursor : 0ull;
        if (i < 2 * a.n_pair_roots) {
            if (tid == 0) {
                const int side = i >= a.n_pair_roots ? 1 : 0;
                const int64_t e = (int64_t)((c + (uint64_t)(i - side * a.n_pair_roots)) % (uint64_t)a.n_pairs);
                root = a.pairs[2 * e + side];
                a.ids_all[a.offsets[0] + i] = root;
            }
        } else if (tid < 64) {
            // a negative: the first node whose cdf exceeds the draw.  The guide table leaves an interval of a few entries: the
            // first wave reads it in ONE round trip and counts the entries <= r (the cdf is non-decreasing: lo + that count is
            // where the binary search ends); longer intervals are halved first.  (One lane bisecting was one dependent round
            // trip per step: the 20 negatives' workgroups ended 4-5 us after the 1024 pair roots' and with them the launch.)
            const uint64_t t = (uint64_t)(i - 2 * a.n_pair_roots);
            const uint64_t nkey = gs_mix64(a.neg_seed ^ (st * 0x9E3779B97F4A7C15ull) ^ (0xFFull << 56));
            // keyed by the GLOBAL slot: data-parallel ranks draw different negatives
            const uint32_t r = (uint32_t)(gs_mix64(nkey + t + (uint64_t)a.root_offset) >> 32);
            int64_t lo = 0, hi = a.n_cdf - 1;  // first index with cdf[idx] > r
            if (a.guide) {
                const uint32_t b = r >> (32 - a.guide_bits);
                lo = a.guide[b];
                hi = min((int64_t)a.guide[b + 1], a.n_cdf - 1);
            }
            while (hi - lo > 64) {
                const int64_t mid = (lo + hi) >> 1;
                if (a.cdf[mid] > r) hi = mid; else lo = mid + 1;
            }
            const bool le = (int64_t)tid < hi - lo && a.cdf[lo + tid] <= r;
            root = (int32_t)(lo + (int64_t)__popcll(__ballot(le)));
            if (tid == 0) a.ids_all[a.offsets[0] + i] = root;
        }
    } else if (a.order) {
        const uint64_t c = a.cursor ? *a.cursor : 0ull;
        root = a.order[(int64_t)((c + (uint64_t)i) % (uint64_t)a.n_order)];        // (requested; first used below the columns)
    } else {
        root = a.ids_all[a.offsets[0] + i];
    }
    const uint32_t seg = a.law.law == GS_LAW_REFERENCE ? (uint32_t)(i >= a.seg1) + (uint32_t)(i >= a.seg2) : 0u;
    auto hop_key = [&](const int h) -> uint64_t {
        return gs_mix64(a.seed ^ (st * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(a.hop0 + seg * (uint32_t)a.n_hops + h) << 56));
    };
    // GS_LAW_REFERENCE: every hop's call columns once per workgroup instead of once per slot, all hops up front -- they depend on
    // the sampler clock only, so the permutation arithmetic runs under the root's round trip instead of between the hops'
    if (a.law.law == GS_LAW_REFERENCE) {
        for (int h = 0; h < a.n_hops; ++h) {
            const int s = a.fan[h];
            if (s <= GS_LAW_COLS) {
                const uint64_t key = hop_key(h);
                for (int jj = tid; jj < s; jj += nthr) law_cols[h][jj] = (int32_t)gs_call_column(key, (uint32_t)jj, (uint32_t)a.law.max_degree);
            }
        }
    }
    if (a.order && !a.pairs && tid == 0) a.ids_all[a.offsets[0] + i] = root;
    // the root's label row (stored at once: a form that only requested it here and stored it behind the hops measured a 0.3 us
    // shorter chain, and 200-step training runs with it differed from each other in 1-5 % of the runs -- never without it,
    // profiles/r06_determinism.txt; the mechanism was not found, the form is gone)
    // A workgroup gives the row to its LAST wave (when it fits one): that wave's load -> store round trip then runs beside the
    // first hop's, which only occupies the first wave, instead of in front of it.
    if (a.order && !a.pairs && a.label_table) {
        const int Cp_lab = (a.C + 3) & ~3;
        if (!PER_WAVE && nthr >= 128 && Cp_lab <= 64) {
            const int k = tid - (nthr - 64);
            if (k >= 0 && k < Cp_lab) a.labels_out[i * a.ldo + k] = k < a.C ? a.label_table[(int64_t)root * a.ldt + k] : 0.f;
        } else {
            for (int k = tid; k < Cp_lab; k += nthr)
                a.labels_out[i * a.ldo + k] = k < a.C ? a.label_table[(int64_t)root * a.ldt + k] : 0.f;
        }
    }
    if (tid == 0) lvl[0][0] = root;
    lds_sync();
    int64_t count_prev = 1;
    for (int h = 0; h < a.n_hops; ++h) {
        const int s = a.fan[h];
        const int64_t count = count_prev * s;
        const uint64_t key = hop_key(h);
        const int32_t* prev = lvl[h & 1];
        int32_t* next = lvl[(h + 1) & 1];
        const bool keep = (h + 1 < a.n_hops);  // the last hop is only written to global memory
        const int32_t* cols = (a.law.law == GS_LAW_REFERENCE && s <= GS_LAW_COLS) ? law_cols[h] : nullptr;
        // U slots per thread and pass: their lookups are requested together, the stores follow (one slot at a time, the store of
        // slot t stood between the lookups of t and t + nthr: a wave walking a 250-slot hop paid four round trips for one)
        constexpr int U = PER_WAVE ? 4 : 1;
        for (int64_t t0 = tid; t0 < count; t0 += (int64_t)U * nthr) {
            int32_t picks[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t t = t0 + (int64_t)u * nthr;
                int32_t pick = a.pad_id;
                if (t < count) {
                    const int64_t pl = t / s;  // parent slot in the previous level
                    const uint32_t j = (uint32_t)(t - pl * s);
                    const int32_t id = prev[pl];
                    if (a.table) {
                        if (id >= 0 && (int64_t)id <= a.n_nodes) {             // row n_nodes = the all-pad row
                            const uint32_t M = (uint32_t)a.law.max_degree;
                            const uint32_t c = cols ? (uint32_t)cols[j] : gs_call_column(key, j, M);
                            pick = a.table[(int64_t)id * M + c];
                        }
                    } else if (id >= 0 && (int64_t)id < a.n_nodes) {
                        const int64_t b = a.rowptr[id];
                        const int32_t deg = (int32_t)(a.rowptr[id + 1] - b);
                        if (deg > 0) {
                            const int64_t grow = (a.root_offset + i) * count_prev + pl;  // global row at this hop
                            pick = a.col[b + (int64_t)gs_draw(a.law, a.seed, key, grow, j, s, id, (uint32_t)deg, cols)];
                        }
                    }
                }
                picks[u] = pick;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t t = t0 + (int64_t)u * nthr;
                if (t < count) {
                    if (keep) next[t] = picks[u];
                    a.ids_all[a.offsets[h + 1] + i * count + t] = picks[u];
                }
            }
        }
        lds_sync();
        count_prev = count;
    }
}

// host: validate (law, max_degree) against the fan-outs of the calls it will serve
static inline int gs_law_args(int32_t law, int32_t max_degree, const int32_t* fans, int32_t n_fans, SampleLaw* out) {
    GS_REQUIRE(law == GS_LAW_IID || law == GS_LAW_REFERENCE || law == GS_LAW_DISTINCT, "sampler: unknown law %d", law);
    GS_REQUIRE(max_degree >= 0, "sampler: max_degree must be >= 0");
    if (law == GS_LAW_REFERENCE) {
        GS_REQUIRE(max_degree > 0, "sampler: GS_LAW_REFERENCE needs max_degree > 0 (the padded table width)");
        for (int h = 0; h < n_fans; ++h)
            GS_REQUIRE(fans[h] <= max_degree, "sampler: num_samples=%d must be <= max_degree=%d (tf.slice would fail)",
                       fans[h], max_degree);
    }
    out->law = law;
    out->max_degree = law == GS_LAW_IID ? 0 : max_degree;
    return GS_OK;
}

// host: C-ABI arguments -> FanoutArgs (shared validation); *kept_max = largest per-root count of a kept hop
static inline int gs_fanout_args(const int64_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t pad_id, int32_t n_hops,
                                 const int32_t* fan_host, const int64_t* offsets_host, int32_t* ids_all, int64_t B, uint64_t seed,
                                 uint64_t step, const uint64_t* step_dev, uint32_t hop0, int64_t root_offset,
                                 const int32_t* order, int64_t n_order, const uint64_t* cursor_dev, const float* label_table,
                                 int64_t ld_table, int32_t C, float* labels_out, int64_t ld_out, int32_t law,
                                 int32_t max_degree, FanoutArgs* out, int64_t* kept_max) {
    GS_REQUIRE(rowptr && col && ids_all && fan_host && offsets_host && n_nodes > 0, "gs_sample_fanout_csr: null pointer");
    GS_REQUIRE(n_hops >= 1 && n_hops <= GS_MAX_HOPS, "gs_sample_fanout_csr: 1..%d hops", GS_MAX_HOPS);
    GS_REQUIRE(hop0 + n_hops <= 256, "gs_sample_fanout_csr: hop ids must be < 256");
    GS_REQUIRE(!order || n_order > 0, "gs_sample_fanout_csr: empty order");
    GS_REQUIRE(!label_table || (labels_out && C > 0 && ld_table >= C && ld_out >= ((C + 3) & ~3) && order),
               "gs_sample_fanout_csr: bad label staging arguments");
    FanoutArgs a = {};
    if (gs_law_args(law, max_degree, fan_host, n_hops, &a.law) != GS_OK) return GS_EINVAL;
    a.rowptr = rowptr; a.col = col; a.n_nodes = n_nodes; a.pad_id = pad_id; a.n_hops = n_hops;
    int64_t count = 1, kmax = 1;
    for (int h = 0; h < n_hops; ++h) {
        GS_REQUIRE(fan_host[h] > 0, "gs_sample_fanout_csr: fan-out must be positive");
        a.fan[h] = fan_host[h];
        if (h + 1 < n_hops) {
            count *= fan_host[h];
            kmax = std::max(kmax, count);
        }
    }
    for (int h = 0; h <= n_hops; ++h) a.offsets[h] = offsets_host[h];
    a.ids_all = ids_all; a.B = B; a.seed = seed; a.step = step; a.step_dev = step_dev; a.hop0 = hop0;
    a.root_offset = root_offset;
    a.seg1 = a.seg2 = B;
    a.order = order; a.n_order = n_order; a.cursor = cursor_dev;
    a.label_table = label_table; a.ldt = ld_table; a.C = C; a.labels_out = labels_out; a.ldo = ld_out;
    GS_REQUIRE(B < (1ll << 31), "gs_sample_fanout_csr: batch too large");
    *out = a;
    *kept_max = kmax;
    return GS_OK;
}

// host: struct gs_fanout_desc -> FanoutArgs (incl. the optional unsupervised root staging)
static inline int gs_fanout_args_desc(const gs_fanout_desc* s, FanoutArgs* out, int64_t* kept_max) {
    GS_REQUIRE(s, "gs_fanout_desc: null descriptor");
    int rc = gs_fanout_args(s->rowptr, s->col, s->n_nodes, s->pad_id, s->n_hops, s->fan, s->offsets, s->ids_all, s->B, s->seed,
                            s->step, s->step_dev, s->hop0, s->root_offset, s->order, s->n_order, s->cursor_dev, s->label_table,
                            s->ld_table, s->C, s->labels_out, s->ld_out, s->law, s->max_degree, out, kept_max);
    if (rc != GS_OK) return rc;
    GS_REQUIRE(!s->padded_table || s->law == GS_LAW_REFERENCE, "gs_fanout_desc: padded_table is the GS_LAW_REFERENCE table");
    out->table = s->padded_table;
    if (s->pairs) {
        GS_REQUIRE(!s->order, "gs_fanout_desc: pairs and order are exclusive");
        GS_REQUIRE(s->n_pairs > 0 && s->n_pair_roots >= 0 && s->n_neg >= 0 && 2 * s->n_pair_roots + s->n_neg == s->B,
                   "gs_fanout_desc: pair staging needs B == 2 * n_pair_roots + n_neg");
        GS_REQUIRE(s->n_neg == 0 || (s->cdf && s->n_cdf > 0), "gs_fanout_desc: negatives need the unigram cdf");
        GS_REQUIRE(!s->guide || (s->guide_bits >= 1 && s->guide_bits <= 20), "gs_fanout_desc: guide_bits in 1..20");
        out->pairs = s->pairs; out->n_pairs = s->n_pairs; out->n_pair_roots = s->n_pair_roots;
        out->cdf = s->cdf; out->guide = s->guide; out->n_cdf = s->n_cdf; out->n_neg = s->n_neg; out->guide_bits = s->guide_bits;
        out->neg_seed = s->neg_seed;
        out->seg1 = s->n_pair_roots; out->seg2 = 2 * s->n_pair_roots;
    }
    if (s->seg_begin[0] > 0 || s->seg_begin[1] > 0) {
        GS_REQUIRE(s->seg_begin[0] >= 0 && s->seg_begin[0] <= s->seg_begin[1] && s->seg_begin[1] <= s->B,
                   "gs_fanout_desc: need 0 <= seg_begin[0] <= seg_begin[1] <= B");
        GS_REQUIRE(!s->pairs || (s->seg_begin[0] == s->n_pair_roots && s->seg_begin[1] == 2 * s->n_pair_roots),
                   "gs_fanout_desc: seg_begin must match the pair staging");
        out->seg1 = s->seg_begin[0]; out->seg2 = s->seg_begin[1];
    }
    GS_REQUIRE(s->law != GS_LAW_REFERENCE || out->seg1 >= s->B || s->hop0 + 3u * (uint32_t)s->n_hops <= 256u,
               "gs_fanout_desc: call ids must be < 256");
    return GS_OK;
}
