// Runtime plumbing of the C ABI: error string, device info, streams, hipGraph capture/replay, events,
// device-side counters, and the host-side (multithreaded C++) edge-list -> CSR builder.
#include "gs_common.h"
#include <string.h>
#include <thread>
#include <vector>
#include <algorithm>
#include <atomic>

static thread_local char g_err[512] = "";

void gs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gs_last_error(void) { return g_err; }
extern "C" int gs_abi_version(void) { return GS_ABI_VERSION; }

// sizeof() of the descriptor structs of include/graphsage_amd.h, in declaration order: a binding checks its own struct
// definitions against these at load time instead of corrupting memory on a layout mismatch.
extern "C" int gs_abi_struct_sizes(int32_t* sizes_out_host, int32_t capacity) {
    const int32_t sizes[] = {(int32_t)sizeof(gs_gather_desc), (int32_t)sizeof(gs_wgrad_desc), (int32_t)sizeof(gs_var_desc),
                             (int32_t)sizeof(gs_fanout_desc), (int32_t)sizeof(gs_tail_desc), (int32_t)sizeof(gs_dropout),
                             (int32_t)sizeof(gs_pull_desc), (int32_t)sizeof(gs_lp_tail_desc)};
    const int32_t n = (int32_t)(sizeof(sizes) / sizeof(sizes[0]));
    for (int32_t i = 0; i < n && i < capacity; ++i) sizes_out_host[i] = sizes[i];
    return n;
}

extern "C" int gs_device_info(int* cu_count, int* xcd_count, char* arch_name_host, int arch_name_len) {
    int dev = 0;
    GS_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    GS_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (xcd_count) *xcd_count = 8;  // MI355X: 8 XCDs x 32 CUs (not queryable through hipDeviceProp_t)
    if (arch_name_host && arch_name_len > 0) {
        strncpy(arch_name_host, prop.gcnArchName, (size_t)arch_name_len - 1);
        arch_name_host[arch_name_len - 1] = 0;
    }
    return GS_OK;
}

// ----------------------------------------------------------------------------- streams / graphs
extern "C" int gs_stream_create(void** stream_out) {
    GS_REQUIRE(stream_out, "gs_stream_create: null out");
    hipStream_t s;
    GS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = (void*)s;
    return GS_OK;
}
extern "C" int gs_stream_destroy(void* stream) {
    GS_HIP(hipStreamDestroy((hipStream_t)stream));
    return GS_OK;
}
extern "C" int gs_stream_sync(void* stream) {
    // GS_SYNC_SPIN=1 (diagnostics, benchmarks/r5_launch_probe.sh): poll the stream instead of sleeping on its completion signal
    static const bool spin = getenv("GS_SYNC_SPIN") && atoi(getenv("GS_SYNC_SPIN")) != 0;
    if (spin) {
        hipError_t q;
        while ((q = hipStreamQuery((hipStream_t)stream)) == hipErrorNotReady) { }
        GS_HIP(q);
        return GS_OK;
    }
    GS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return GS_OK;
}
extern "C" int gs_capture_begin(void* stream) {
    GS_REQUIRE(stream, "gs_capture_begin: the legacy NULL stream cannot be captured");
    GS_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return GS_OK;
}
extern "C" int gs_capture_end(void* stream, void** graph_exec_out) {
    GS_REQUIRE(graph_exec_out, "gs_capture_end: null out");
    hipGraph_t graph = nullptr;
    GS_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    // GS_GRAPH_INSTANTIATE_FLAGS (diagnostics): hipGraphInstantiateFlags, e.g. 2 = upload the executable graph at instantiation
    static const int inst_flags = getenv("GS_GRAPH_INSTANTIATE_FLAGS") ? atoi(getenv("GS_GRAPH_INSTANTIATE_FLAGS")) : 0;
    hipError_t e = inst_flags ? hipGraphInstantiateWithFlags(&exec, graph, (unsigned long long)inst_flags)
                              : hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);  // the executable graph keeps what it needs
    if (e != hipSuccess) {
        gs_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return GS_EHIP;
    }
    *graph_exec_out = (void*)exec;
    return GS_OK;
}
extern "C" int gs_graph_launch(void* graph_exec, void* stream) {
    GS_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return GS_OK;
}
extern "C" int gs_graph_destroy(void* graph_exec) {
    GS_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return GS_OK;
}

// ----------------------------------------------------------------------------- events
extern "C" int gs_event_create(void** ev_out) {
    GS_REQUIRE(ev_out, "gs_event_create: null out");
    hipEvent_t e;
    GS_HIP(hipEventCreate(&e));
    *ev_out = (void*)e;
    return GS_OK;
}
extern "C" int gs_event_record(void* ev, void* stream) {
    GS_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return GS_OK;
}
extern "C" int gs_stream_wait_event(void* stream, void* ev) {
    GS_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return GS_OK;
}
extern "C" int gs_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out_host) {
    GS_REQUIRE(ms_out_host, "gs_event_elapsed_ms: null out");
    GS_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    GS_HIP(hipEventElapsedTime(ms_out_host, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return GS_OK;
}
extern "C" int gs_event_destroy(void* ev) {
    GS_HIP(hipEventDestroy((hipEvent_t)ev));
    return GS_OK;
}

// ----------------------------------------------------------------------------- device counters
__global__ void advance_counter_kernel(uint64_t* c, uint64_t delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += delta;
}
// Diagnostics: ONE wave that sleeps for `us` microseconds of wall clock (100 MHz constant counter) -- the stand-in for a
// latency-bound collective when the data-parallel step schedule is probed on a single GPU (it holds one CU's wave slot
// and leaves the rest of the chip free, like an all-reduce of 0.9 MB waiting on its peers).
__global__ void spin_us_kernel(const uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int gs_spin_us(float us, void* stream) {
    GS_REQUIRE(us >= 0.f && us <= 1.0e6f, "gs_spin_us: 0 .. 1e6 us");
    hipLaunchKernelGGL(spin_us_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint64_t)(us * 100.0f));
    GS_LAUNCH_CHECK("spin_us_kernel");
    return GS_OK;
}

extern "C" int gs_advance_counter(uint64_t* counter_dev, uint64_t delta, void* stream) {
    GS_REQUIRE(counter_dev, "gs_advance_counter: null counter");
    hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter_dev, delta);
    GS_LAUNCH_CHECK("advance_counter_kernel");
    return GS_OK;
}

__global__ void advance_counters_kernel(uint64_t* c0, uint64_t d0, uint64_t* c1, uint64_t d1, uint64_t* c2, uint64_t d2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (c0) *c0 += d0;
        if (c1) *c1 += d1;
        if (c2) *c2 += d2;
    }
}
extern "C" int gs_advance_counters(uint64_t* c0, uint64_t d0, uint64_t* c1, uint64_t d1, uint64_t* c2, uint64_t d2,
                                   void* stream) {
    hipLaunchKernelGGL(advance_counters_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c0, d0, c1, d1, c2, d2);
    GS_LAUNCH_CHECK("advance_counters_kernel");
    return GS_OK;
}

// Step epilogue in ONE launch: loss_out[0] = scale * sum(loss_rows[0:n]) (+= if accumulate) in a fixed order, then
// the device counters are advanced (cursor / sampler clock / optimizer step).
__global__ __launch_bounds__(256) void finalize_step_kernel(const float* __restrict__ loss_rows, int64_t n, float scale,
                                                            float* __restrict__ loss_out, int accumulate, uint64_t* c0,
                                                            uint64_t d0, uint64_t* c1, uint64_t d1, uint64_t* c2,
                                                            uint64_t d2, const float* __restrict__ aux_rows, float aux_scale,
                                                            float* __restrict__ aux_out) {
    __shared__ float part[4];
    __shared__ float part2[4];
    const StepEpilogue e = {loss_rows, n, scale, loss_out, accumulate, aux_rows, aux_scale, aux_out, c0, d0, c1, d1, c2, d2};
    gs_step_epilogue_block(e, part, part2);
}
extern "C" int gs_finalize_step(const float* loss_rows, int64_t n, float scale, float* loss_out, int accumulate,
                                uint64_t* c0, uint64_t d0, uint64_t* c1, uint64_t d1, uint64_t* c2, uint64_t d2,
                                void* stream) {
    GS_REQUIRE(!loss_rows || (loss_out && n >= 0), "gs_finalize_step: bad args");
    hipLaunchKernelGGL(finalize_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_rows, n, scale, loss_out,
                       accumulate, c0, d0, c1, d1, c2, d2, (const float*)nullptr, 0.f, (float*)nullptr);
    GS_LAUNCH_CHECK("finalize_step_kernel");
    return GS_OK;
}

extern "C" int gs_finalize_step2(const float* loss_rows, int64_t n, float scale, float* loss_out, int accumulate,
                                 const float* aux_rows, float aux_scale, float* aux_out, uint64_t* c0, uint64_t d0,
                                 uint64_t* c1, uint64_t d1, uint64_t* c2, uint64_t d2, void* stream) {
    GS_REQUIRE(!loss_rows || (loss_out && n >= 0), "gs_finalize_step2: bad args");
    GS_REQUIRE(!aux_rows || aux_out, "gs_finalize_step2: aux_out missing");
    hipLaunchKernelGGL(finalize_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_rows, n, scale, loss_out,
                       accumulate, c0, d0, c1, d1, c2, d2, aux_rows, aux_scale, aux_out);
    GS_LAUNCH_CHECK("finalize_step_kernel");
    return GS_OK;
}

// ----------------------------------------------------------------------------- host CSR builder
// Counting sort by source node; parallel histogram + parallel fill over node ranges.
extern "C" int gs_build_csr_host(const int32_t* src, const int32_t* dst, const uint8_t* keep,
                                 int64_t n_edges, int64_t n_nodes, int symmetrize,
                                 int64_t* rowptr, int32_t* col, int64_t col_capacity, int64_t* nnz_out) {
    GS_REQUIRE(src && dst && rowptr && col && nnz_out && n_nodes > 0 && n_edges >= 0, "gs_build_csr_host: bad args");
    std::vector<int64_t> cnt((size_t)n_nodes + 1, 0);
    for (int64_t e = 0; e < n_edges; ++e) {
        if (keep && !keep[e]) continue;
        int32_t s = src[e], d = dst[e];
        if (s < 0 || s >= n_nodes || d < 0 || d >= n_nodes) {
            gs_set_error("gs_build_csr_host: edge %lld endpoint out of range", (long long)e);
            return GS_EINVAL;
        }
        cnt[(size_t)s + 1]++;
        if (symmetrize && s != d) cnt[(size_t)d + 1]++;
    }
    rowptr[0] = 0;
    for (int64_t i = 0; i < n_nodes; ++i) rowptr[i + 1] = rowptr[i] + cnt[(size_t)i + 1];
    int64_t nnz = rowptr[n_nodes];
    *nnz_out = nnz;
    if (nnz > col_capacity) {
        gs_set_error("gs_build_csr_host: col capacity %lld < nnz %lld", (long long)col_capacity, (long long)nnz);
        return GS_EINVAL;
    }
    std::vector<int64_t> cursor(rowptr, rowptr + n_nodes);
    for (int64_t e = 0; e < n_edges; ++e) {
        if (keep && !keep[e]) continue;
        int32_t s = src[e], d = dst[e];
        col[cursor[(size_t)s]++] = d;
        if (symmetrize && s != d) col[cursor[(size_t)d]++] = s;
    }
    // sort each adjacency list (parallel over node ranges) so the CSR is canonical
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    std::atomic<int64_t> next(0);
    const int64_t chunk = 4096;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&]() {
            for (;;) {
                int64_t b = next.fetch_add(chunk);
                if (b >= n_nodes) break;
                int64_t e = std::min(n_nodes, b + chunk);
                for (int64_t i = b; i < e; ++i) std::sort(col + rowptr[i], col + rowptr[i + 1]);
            }
        });
    for (auto& x : th) x.join();
    // drop parallel edges (networkx.Graph, which the reference loads into, keeps one edge per node pair)
    int64_t w = 0;
    for (int64_t i = 0; i < n_nodes; ++i) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        rowptr[i] = w;
        for (int64_t k = b; k < e; ++k)
            if (k == b || col[k] != col[k - 1]) col[w++] = col[k];
    }
    rowptr[n_nodes] = w;
    *nnz_out = w;
    return GS_OK;
}
