// C2: gradient all-reduce as DIRECT PEER STORES over xGMI (the exchange step of the data-parallel hot path;
// the reference is single-device, supervised_train.py:55-59, so there is no reference interface to mirror).
//
// Why beside RCCL (gs_comm.hip, the default): the flat gradient is 0.9 MB.  A ring all-reduce over 8 ranks is 14
// latency-bound hops for 115 KB pieces; xGMI is point-to-point (every GPU has a link to every other GPU), so the
// latency-optimal exchange for this size is ONE hop out and ONE hop back:
//     reduce-scatter : rank r stores slice p of its gradient straight into rank p's window      (world-1 links in parallel)
//     all-gather     : rank p sums the world copies of its slice IN RANK ORDER and stores the sum into every window
// Both are plain global stores through hipIpc-mapped peer pointers followed by one system-scope flag store per
// chunk; a rank never reads remote memory (remote reads are round trips, remote writes are posted).  ONE kernel launch
// per step on the caller's stream, capturable into the step's hipGraph like ncclAllReduce.
//
// Window of a rank (uncached device memory, shared by hipIpcMemHandle; L = slice length, W = chunks per slice):
//     recv[2][world][L]   the copies of MY slice, one slot per source rank, double-buffered by epoch parity
//     full[world * L]     the reduced gradient, slice p written by rank p
//     rs_flag[world][W]   epoch of the last complete copy of chunk w from rank q      } one writer per word, a release
//     ag_flag[world][W]   epoch of the last reduced chunk w of rank p's slice          } store; words 64 bytes apart
//     wg_epoch[world*W]   exchanges done, private to each workgroup of the own kernel; error
// Safety of buffer reuse: a slot of recv[parity] is overwritten two epochs later, which needs its writer to have
// finished the epoch in between, which needs MY reduced slice of that epoch, which my kernel of that epoch produced
// after its last read of the slot.  full[] is overwritten by p's next reduce, which needs my next push, which follows my
// copy-out in stream order.  Every wait is bounded (spin_limit polls): a rank whose peer never arrives sets the error
// word and leaves -- the host reads it (gs_peer_status) -- instead of hanging the device.
//
// The sum order is the rank order on every rank, so all ranks hold the same bits afterwards (replicas stay identical);
// for world = 2 it is also bit-identical to RCCL's sum (a + b).
#include "gs_common.h"
#include <string.h>
#include <new>

#define GS_PEER_MAX_WORLD 16
#define GS_PEER_THREADS 256
#define GS_PEER_STRIDE 16        // flag words sit 64 bytes apart

struct PeerWindow {             // device addresses inside ONE rank's window
    float* recv;                // [2][world][L]
    float* full;                // [world * L]
    uint32_t* rs_flag;          // [world][W] words at a 64-byte stride: epoch of the last complete copy of chunk w from rank q
    uint32_t* ag_flag;          // [world][W]: epoch of the last reduced chunk w of rank p's slice
    uint32_t* wg_epoch;         // [world * W]: exchanges done, one private word per workgroup of the own kernel
    uint32_t* error;
};

struct PeerArgs {
    PeerWindow win[GS_PEER_MAX_WORLD];     // win[me] = own window, win[p] = rank p's window as mapped here
    float* grads;
    int64_t n, L;
    int32_t world, me, W;
    uint32_t spin_limit;
};

// Memory ordering, kept to ONE cache maintenance operation per workgroup and hand-over (a system-scope release is a
// write-back of the XCD's L2, an acquire an invalidate: issued by every wave, or by every poll, they cost more than the
// 0.9 MB being moved -- measured 108 us -> see DESIGN.md):
//   writer : every wave waits for its own stores (s_waitcnt), workgroup barrier, then thread 0 release-stores the flag
//   reader : polling lanes spin on RELAXED system-scope loads; the wave that polled issues one acquire fence, then the
//            workgroup barrier; the other waves share its CU (same L1) and XCD (same L2)
__device__ __forceinline__ void peer_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// wait until *flag >= target (wrap-safe); 0 after spin_limit polls
__device__ __forceinline__ int peer_wait(const uint32_t* flag, uint32_t target, uint32_t spin_limit) {
    for (uint32_t it = 0; it < spin_limit; ++it) {
        if ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - target) >= 0) return 1;
        __builtin_amdgcn_s_sleep(16);
    }
    return 0;
}

__global__ __launch_bounds__(GS_PEER_THREADS) void peer_clear_kernel(uint32_t* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * GS_PEER_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * GS_PEER_THREADS) p[i] = 0u;
}

// grid = world * W workgroups; workgroup (p, w) owns every transfer between this rank and rank p for chunk w.  Hand-overs
// are per chunk: a flag word holds the epoch of the last complete chunk (ONE release store by ONE writer -- no contended
// read-modify-write: atomics on one address serialise at ~0.3 us each on this chip).
__global__ __launch_bounds__(GS_PEER_THREADS) void peer_allreduce_kernel(const PeerArgs a) {
    const int p = blockIdx.x / a.W, w = blockIdx.x % a.W;
    const int me = a.me, world = a.world;
    const PeerWindow mine = a.win[me];
    // the epoch is device state (kernel arguments are frozen inside a hipGraph): a private word per workgroup, read here
    // and advanced at the end by the same workgroup, so that all workgroups of all launches agree without any atomics
    const uint32_t epoch = mine.wg_epoch[blockIdx.x] + 1u;
    if (*mine.error != 0u) {
        // sticky: an earlier exchange gave up waiting.  Its flags are out of step for good, so every later launch (the rest
        // of a multi-step hipGraph, say) returns at once instead of spending spin_limit polls each; the host sees the
        // error word at its next gs_peer_status and has to re-create the windows.
        if (threadIdx.x == 0) mine.wg_epoch[blockIdx.x] = epoch;
        return;
    }
    const int par = (int)(epoch & 1u);
    // chunk w of a slice: [c0, c1) floats, whole float4s
    const int64_t per = ((a.L / 4 + a.W - 1) / a.W) * 4;
    const int64_t c0 = std::min<int64_t>(a.L, (int64_t)w * per), c1 = std::min<int64_t>(a.L, c0 + per);
    const int64_t lim = std::max<int64_t>(0, std::min<int64_t>(a.L, a.n - (int64_t)p * a.L));     // the last slice is ragged
    uint32_t err = 0;

    // ---- A. reduce-scatter push: my copy of slice p -> rank p's recv[par][me]  (p == me: a local copy)
    {
        const float* src = a.grads + (int64_t)p * a.L;
        float* dst = a.win[p].recv + ((int64_t)par * world + me) * a.L;
        for (int64_t i = c0 + (int64_t)threadIdx.x * 4; i < c1; i += GS_PEER_THREADS * 4) {
            f32x4 v;
            if (i + 4 <= lim) v = *(const f32x4*)(src + i);
            else {
                v.x = i < lim ? src[i] : 0.f; v.y = i + 1 < lim ? src[i + 1] : 0.f;
                v.z = i + 2 < lim ? src[i + 2] : 0.f; v.w = 0.f;
            }
            *(f32x4*)(dst + i) = v;
        }
        peer_stores_done();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(a.win[p].rs_flag + ((int64_t)me * a.W + w) * GS_PEER_STRIDE, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- B. chunk w of MY slice has landed from every rank: sum the copies in rank order, store the sum into rank p's full[]
    {
        int ok = 1;
        if ((int)threadIdx.x < world) {
            ok = peer_wait(mine.rs_flag + ((int64_t)threadIdx.x * a.W + w) * GS_PEER_STRIDE, epoch, a.spin_limit);
            if (!ok) atomicOr(mine.error, 1u | (256u << threadIdx.x));       // bits 8..: the ranks whose copies are missing
        }
        if (threadIdx.x < GS_WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");       // world <= 16: the polling lanes are in wave 0
        ok = __syncthreads_and(ok);
        if (!ok) err = 1u;
    }
    if (!err) {
        const float* in = mine.recv + (int64_t)par * world * a.L;
        float* dst = a.win[p].full + (int64_t)me * a.L;
        for (int64_t i = c0 + (int64_t)threadIdx.x * 4; i < c1; i += GS_PEER_THREADS * 4) {
            f32x4 s = *(const f32x4*)(in + i);
            for (int q = 1; q < world; ++q) {
                const f32x4 v = *(const f32x4*)(in + (int64_t)q * a.L + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *(f32x4*)(dst + i) = s;
        }
        peer_stores_done();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(a.win[p].ag_flag + ((int64_t)me * a.W + w) * GS_PEER_STRIDE, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // ---- C. chunk w of rank p's reduced slice has landed in my full[]: copy it out into the gradient buffer
        int ok = 1;
        if (threadIdx.x == 0) {
            ok = peer_wait(mine.ag_flag + ((int64_t)p * a.W + w) * GS_PEER_STRIDE, epoch, a.spin_limit);
            if (!ok) atomicOr(mine.error, 2u | (256u << p));
        }
        if (threadIdx.x < GS_WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        ok = __syncthreads_and(ok);
        if (!ok) err = 2u;
    }
    if (!err) {
        const float* src = mine.full + (int64_t)p * a.L;
        float* dst = a.grads + (int64_t)p * a.L;
        for (int64_t i = c0 + (int64_t)threadIdx.x * 4; i < c1; i += GS_PEER_THREADS * 4) {
            const f32x4 v = *(const f32x4*)(src + i);
            if (i + 4 <= lim) *(f32x4*)(dst + i) = v;
            else {
                if (i < lim) dst[i] = v.x;
                if (i + 1 < lim) dst[i + 1] = v.y;
                if (i + 2 < lim) dst[i + 2] = v.z;
            }
        }
    }
    if (threadIdx.x == 0) mine.wg_epoch[blockIdx.x] = epoch;
}

struct GsPeer {
    int32_t world, rank, W;
    int64_t n, L, bytes;
    uint32_t spin_limit;
    char* base[GS_PEER_MAX_WORLD];          // base[rank] = own allocation; others: hipIpcOpenMemHandle / in-process pointers
    bool ipc[GS_PEER_MAX_WORLD];
};

static int64_t peer_slice_len(int64_t n, int32_t world) {
    const int64_t L = (n + world - 1) / world;
    return (L + 63) / 64 * 64;              // whole 256-byte lines per slice
}
static int64_t peer_window_bytes(int64_t L, int32_t world, int32_t W) {
    return (int64_t)sizeof(float) * (2 * world * L + world * L) + (int64_t)sizeof(uint32_t) * (2 * (int64_t)world * W * GS_PEER_STRIDE + (int64_t)world * W + 64);
}
static PeerWindow peer_layout(char* base, int64_t L, int32_t world, int32_t W) {
    PeerWindow w;
    w.recv = (float*)base;
    w.full = w.recv + 2 * world * L;
    w.rs_flag = (uint32_t*)(w.full + world * L);
    w.ag_flag = w.rs_flag + (int64_t)world * W * GS_PEER_STRIDE;
    w.wg_epoch = w.ag_flag + (int64_t)world * W * GS_PEER_STRIDE;
    w.error = w.wg_epoch + (int64_t)world * W;
    return w;
}

extern "C" int gs_peer_create(int64_t n_floats, int32_t world, int32_t rank, int32_t chunks, int64_t spin_limit, void** peer_out) {
    GS_REQUIRE(peer_out && n_floats > 0 && world >= 1 && world <= GS_PEER_MAX_WORLD && rank >= 0 && rank < world,
               "gs_peer_create: need n > 0, 1 <= world <= %d, 0 <= rank < world", GS_PEER_MAX_WORLD);
    GS_REQUIRE(chunks >= 0 && chunks <= 256 && spin_limit >= 0 && spin_limit <= 0xffffffffll, "gs_peer_create: chunks 0..256, spin_limit 0..2^32-1");
    GsPeer* g = new (std::nothrow) GsPeer();
    GS_REQUIRE(g != nullptr, "gs_peer_create: out of host memory");
    memset(g, 0, sizeof(*g));
    g->world = world; g->rank = rank; g->n = n_floats;
    g->L = peer_slice_len(n_floats, world);
    // chunks = 0: about two float4 per thread and pass, at most 256 workgroups in all (they must be co-resident: they wait
    // for peers inside the kernel)
    // (round 5: one float4 per thread and pass where the 256-workgroup budget allows -- every pass of a chunk is a dependent
    //  memory round trip through uncached memory, so fewer, fatter workgroups only add round trips: world = 1, L = 230 k:
    //  225 chunks instead of 64)
    g->W = chunks > 0 ? chunks : (int32_t)std::max<int64_t>(1, std::min<int64_t>(256 / world, (g->L + 1023) / 1024));
    g->spin_limit = spin_limit > 0 ? (uint32_t)spin_limit : (1u << 24);
    g->bytes = peer_window_bytes(g->L, world, g->W);
    void* p = nullptr;
    // UNCACHED device memory: a flag word that a peer rewrites must never be served from an XCD's L2 (a polled line would
    // stay resident there -- measured: with two workgroups per XCD nobody invalidates it and the wait times out)
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)g->bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        gs_set_error("gs_peer_create: hipExtMallocWithFlags(uncached, %lld bytes) failed: %s", (long long)g->bytes, hipGetErrorString(e));
        delete g;
        return GS_EHIP;
    }
    // cleared by a kernel of this library (not hipMemset): the code object is then loaded before the first exchange, whose
    // bounded waits must not be spent on a peer's module loading
    peer_clear_kernel<<<256, GS_PEER_THREADS>>>((uint32_t*)p, g->bytes / 4);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        gs_set_error("gs_peer_create: clearing the window failed: %s", hipGetErrorString(e));
        (void)hipFree(p);
        delete g;
        return GS_EHIP;
    }
    g->base[rank] = (char*)p;
    *peer_out = (void*)g;
    return GS_OK;
}

extern "C" int gs_peer_export(void* peer, void* handle_out_host, int32_t len) {
    GS_REQUIRE(peer && handle_out_host && len >= (int32_t)sizeof(hipIpcMemHandle_t), "gs_peer_export: need a %d-byte host buffer",
               (int)sizeof(hipIpcMemHandle_t));
    GsPeer* g = (GsPeer*)peer;
    hipIpcMemHandle_t h;
    GS_HIP(hipIpcGetMemHandle(&h, g->base[g->rank]));
    memset(handle_out_host, 0, (size_t)len);
    memcpy(handle_out_host, &h, sizeof(h));
    return GS_OK;
}

extern "C" int gs_peer_attach(void* peer, int32_t peer_rank, const void* handle_host, int32_t len) {
    GS_REQUIRE(peer && handle_host && len >= (int32_t)sizeof(hipIpcMemHandle_t), "gs_peer_attach: bad args");
    GsPeer* g = (GsPeer*)peer;
    GS_REQUIRE(peer_rank >= 0 && peer_rank < g->world && peer_rank != g->rank && !g->base[peer_rank], "gs_peer_attach: bad or repeated rank %d", peer_rank);
    hipIpcMemHandle_t h;
    memcpy(&h, handle_host, sizeof(h));
    void* p = nullptr;
    GS_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    g->base[peer_rank] = (char*)p;
    g->ipc[peer_rank] = true;
    return GS_OK;
}

// Two ranks of ONE process (several streams / devices of the same process, and the kernel-level tests): the other
// rank's window by address, no IPC handle.
extern "C" int gs_peer_attach_local(void* peer, void* other_peer) {
    GS_REQUIRE(peer && other_peer && peer != other_peer, "gs_peer_attach_local: bad args");
    GsPeer* g = (GsPeer*)peer;
    GsPeer* o = (GsPeer*)other_peer;
    GS_REQUIRE(o->world == g->world && o->n == g->n && o->rank != g->rank && !g->base[o->rank], "gs_peer_attach_local: windows do not match");
    g->base[o->rank] = o->base[o->rank];
    g->ipc[o->rank] = false;
    return GS_OK;
}

extern "C" int gs_peer_allreduce_sum_f32(void* peer, float* buf, int64_t count, void* stream) {
    GS_REQUIRE(peer && buf && gs_aligned16(buf), "gs_peer_allreduce_sum_f32: bad args");
    GsPeer* g = (GsPeer*)peer;
    GS_REQUIRE(count == g->n, "gs_peer_allreduce_sum_f32: the window was created for %lld floats, got %lld", (long long)g->n, (long long)count);
    PeerArgs a;
    memset(&a, 0, sizeof(a));
    for (int r = 0; r < g->world; ++r) {
        GS_REQUIRE(g->base[r] != nullptr, "gs_peer_allreduce_sum_f32: rank %d's window is not attached", r);
        a.win[r] = peer_layout(g->base[r], g->L, g->world, g->W);
    }
    a.grads = buf; a.n = g->n; a.L = g->L; a.world = g->world; a.me = g->rank; a.W = g->W; a.spin_limit = g->spin_limit;
    hipStream_t s = (hipStream_t)stream;
    peer_allreduce_kernel<<<g->world * g->W, GS_PEER_THREADS, 0, s>>>(a);
    GS_LAUNCH_CHECK("peer_allreduce_kernel");
    return GS_OK;
}

// epoch = exchanges completed on this rank; error: bit 0 = a peer's slice copies never arrived, bit 1 = a reduced slice never
// arrived, bits 8.. = the ranks that were waited for in vain.  Synchronises with nothing: call after the stream has been synchronised.
extern "C" int gs_peer_status(void* peer, int64_t* epoch_out_host, int32_t* error_out_host) {
    GS_REQUIRE(peer && epoch_out_host && error_out_host, "gs_peer_status: bad args");
    GsPeer* g = (GsPeer*)peer;
    PeerWindow w = peer_layout(g->base[g->rank], g->L, g->world, g->W);
    uint32_t ep = 0, er = 0;
    GS_HIP(hipMemcpy(&ep, w.wg_epoch, sizeof(ep), hipMemcpyDeviceToHost));
    GS_HIP(hipMemcpy(&er, w.error, sizeof(er), hipMemcpyDeviceToHost));
    *epoch_out_host = (int64_t)ep;
    *error_out_host = (int32_t)er;
    return GS_OK;
}

extern "C" int gs_peer_destroy(void* peer) {
    if (!peer) return GS_OK;
    GsPeer* g = (GsPeer*)peer;
    for (int r = 0; r < g->world; ++r)
        if (r != g->rank && g->base[r] && g->ipc[r]) (void)hipIpcCloseMemHandle(g->base[r]);
    if (g->base[g->rank]) (void)hipFree(g->base[g->rank]);
    delete g;
    return GS_OK;
}
