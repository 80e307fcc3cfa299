// rt_collective.cpp — the one exchange step of the multi-GPU render: per-GPU accumulation images summed into one (rt_api.h,
// "multi-GPU").  The reference has no counterpart (one process, one thread pool, main.rs:1314-1409); north_star asks for image
// tiles or sample ranges sharded over the 8 GPUs of a node with the per-GPU buffers summed by an RCCL reduce over xGMI.
// Every GPU has a direct xGMI link to the root and nobody else needs the sum, so the collective is ONE ncclReduce(sum, root)
// per frame, in place, on the caller's stream — not an all-reduce, no ring of eight.
//
// RCCL is opened with dlopen on first use: the library stays loadable (and every single-GPU entry point usable) on a machine
// without RCCL, and a host that already carries its own copy (PyTorch does) shares it instead of loading a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <new>
#include <vector>

#include "rt_device_internal.h"

using rtk_internal::ctx_device;
using rtk_internal::ctx_fail;

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    const char *why = "not tried";
};

const Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // Which RCCL: the one that lives next to the HIP runtime THIS library is bound to.  A process may hold two ROCm stacks
        // (PyTorch ships its own HIP, HSA and RCCL next to the system's); an RCCL of the other stack would bring up a second
        // HSA runtime and find no device.  RT_RCCL_LIB names a file explicitly; the plain sonames are the last resort.
        std::string mapped;
        if (const char *e = getenv("RT_RCCL_LIB")) mapped = e;
        if (mapped.empty()) {
            Dl_info info{};
            if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
                std::string dir = info.dli_fname;
                const size_t slash = dir.rfind('/');
                if (slash != std::string::npos) {
                    dir.resize(slash + 1);
                    for (const char *name : {"librccl.so.1", "librccl.so"}) {
                        if (r.lib) break;
                        r.lib = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_GLOBAL);
                        if (r.lib) mapped = dir + name;
                    }
                }
            }
        } else {
            r.lib = dlopen(mapped.c_str(), RTLD_NOW | RTLD_GLOBAL);
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) mapped = name;
        }
        if (getenv("RT_DEBUG")) fprintf(stderr, "[rt] RCCL: %s\n", r.lib ? mapped.c_str() : "not found");
        if (!r.lib) { r.why = "librccl.so could not be loaded"; return; }
#define SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f)); if (!r.f) { r.why = "librccl.so lacks nccl" #f; r.lib = nullptr; return; }
        SYM(GetUniqueId) SYM(CommInitRank) SYM(CommInitAll) SYM(CommDestroy) SYM(Reduce) SYM(GroupStart) SYM(GroupEnd) SYM(GetErrorString)
#undef SYM
    });
    return r.lib ? &r : nullptr;
}
const char *rccl_unavailable() { rccl(); static Rccl dummy; (void)dummy; return "RCCL is not available"; }

}  // namespace

struct rt_comm {
    rt_ctx *ctx = nullptr;        // for error messages of rt_comm_reduce_to_root only (the context must be alive for that call)
    int device = 0;               // the context's device, kept here: rt_comm_destroy may run after the context is gone
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

struct rt_group {
    std::vector<rt_ctx *> ctxs;
    std::vector<ncclComm_t> comms;
};

#define RCCL_TRY(ctx, R, call)                                                                                   \
    do {                                                                                                         \
        ncclResult_t r_ = (call);                                                                                \
        if (r_ != ncclSuccess) return ctx_fail(ctx, RT_E_RCCL, "%s failed: %s", #call, (R)->GetErrorString(r_)); \
    } while (0)

extern "C" {

int rt_device_alloc(rt_ctx *ctx, size_t bytes, void **out) {
    if (!ctx) return RT_E_INVALID;
    if (!out || bytes == 0) return ctx_fail(ctx, RT_E_INVALID, "rt_device_alloc: bad arguments");
    *out = nullptr;
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_device_alloc: hipSetDevice failed");
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) return ctx_fail(ctx, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP, "rt_device_alloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    return RT_OK;
}

int rt_device_free(rt_ctx *ctx, void *p) {
    if (!ctx) return RT_E_INVALID;
    if (!p) return RT_OK;
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_device_free: hipSetDevice failed");
    hipError_t e = hipFree(p);
    if (e != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_device_free: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_device_zero(rt_ctx *ctx, void *p, size_t bytes, void *stream) {
    if (!ctx) return RT_E_INVALID;
    if (!p) return ctx_fail(ctx, RT_E_INVALID, "rt_device_zero: NULL buffer");
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_device_zero: hipSetDevice failed");
    hipError_t e = hipMemsetAsync(p, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_device_zero: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_device_read(rt_ctx *ctx, void *dst_host, const void *src_device, size_t bytes, void *stream) {
    if (!ctx) return RT_E_INVALID;
    if (!dst_host || !src_device) return ctx_fail(ctx, RT_E_INVALID, "rt_device_read: NULL buffer");
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_device_read: hipSetDevice failed");
    hipError_t e = hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_device_read: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_stream_create(rt_ctx *ctx, void **out) {
    if (!ctx) return RT_E_INVALID;
    if (!out) return ctx_fail(ctx, RT_E_INVALID, "rt_stream_create: out is NULL");
    *out = nullptr;
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_stream_create: hipSetDevice failed");
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_stream_create: %s", hipGetErrorString(e));
    *out = s;
    return RT_OK;
}

int rt_stream_destroy(rt_ctx *ctx, void *stream) {
    if (!ctx) return RT_E_INVALID;
    if (!stream) return RT_OK;
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_stream_destroy: hipSetDevice failed");
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_stream_destroy: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_stream_wait(rt_ctx *ctx, void *stream, void *after) {
    if (!ctx) return RT_E_INVALID;
    if (stream == after) return RT_OK;      // a stream is in order with itself
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_stream_wait: hipSetDevice failed");
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, (hipStream_t)after);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, ev, 0);
    if (ev) (void)hipEventDestroy(ev);      // (released by the runtime once the wait has been satisfied)
    if (e != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_stream_wait: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_comm_unique_id(char id[RT_COMM_ID_BYTES]) {
    static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rt_api.h and rccl.h disagree on the id size");
    if (!id) return ctx_fail(nullptr, RT_E_INVALID, "rt_comm_unique_id: id is NULL");
    const Rccl *R = rccl();
    if (!R) return ctx_fail(nullptr, RT_E_RCCL, "%s", rccl_unavailable());
    ncclUniqueId u;
    RCCL_TRY(nullptr, R, R->GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return RT_OK;
}

int rt_comm_create(rt_ctx *ctx, const char id[RT_COMM_ID_BYTES], int rank, int world, rt_comm **out) {
    if (!ctx) return RT_E_INVALID;
    if (!out) return ctx_fail(ctx, RT_E_INVALID, "rt_comm_create: out is NULL");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return ctx_fail(ctx, RT_E_INVALID, "rt_comm_create: rank %d of %d", rank, world);
    const Rccl *R = rccl();
    if (!R) return ctx_fail(ctx, RT_E_RCCL, "%s", rccl_unavailable());
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return ctx_fail(ctx, RT_E_HIP, "rt_comm_create: hipSetDevice(%d) failed", ctx_device(ctx));
    rt_comm *c = new (std::nothrow) rt_comm();
    if (!c) return ctx_fail(ctx, RT_E_OOM, "out of host memory");
    c->ctx = ctx; c->device = ctx_device(ctx); c->rank = rank; c->world = world;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = R->CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return ctx_fail(ctx, RT_E_RCCL, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, R->GetErrorString(r)); }
    *out = c;
    return RT_OK;
}

int rt_comm_reduce_to_root(rt_comm *c, float *rgba_sum, int64_t n_floats, int root, void *stream) {
    if (!c) return RT_E_INVALID;
    if (!rgba_sum || n_floats < 0 || root < 0 || root >= c->world) return ctx_fail(c->ctx, RT_E_INVALID, "rt_comm_reduce_to_root: bad arguments (root %d of %d)", root, c->world);
    if (n_floats == 0) return RT_OK;
    const Rccl *R = rccl();
    if (!R) return ctx_fail(c->ctx, RT_E_RCCL, "%s", rccl_unavailable());
    if (hipSetDevice(c->device) != hipSuccess) return ctx_fail(c->ctx, RT_E_HIP, "rt_comm_reduce_to_root: hipSetDevice failed");
    // in place: on the root the sums of all ranks arrive in rgba_sum, on the others it is only read
    RCCL_TRY(c->ctx, R, R->Reduce(rgba_sum, rgba_sum, (size_t)n_floats, ncclFloat, ncclSum, root, c->comm, (hipStream_t)stream));
    return RT_OK;
}

void rt_comm_destroy(rt_comm *c) {
    if (!c) return;
    const Rccl *R = rccl();
    if (R && c->comm) { (void)hipSetDevice(c->device); (void)R->CommDestroy(c->comm); }      // (touches no context: see rt_comm::device)
    delete c;
}

int rt_group_create(const int *device_ids, int n, rt_group **out) {
    if (!out) return ctx_fail(nullptr, RT_E_INVALID, "rt_group_create: out is NULL");
    *out = nullptr;
    if (!device_ids || n < 1 || n > 64) return ctx_fail(nullptr, RT_E_INVALID, "rt_group_create: %d devices", n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (device_ids[i] == device_ids[j]) return ctx_fail(nullptr, RT_E_INVALID, "rt_group_create: device %d listed twice", device_ids[i]);
    const Rccl *R = n > 1 ? rccl() : nullptr;          // a group of one needs no communicator: it works without RCCL
    if (n > 1 && !R) return ctx_fail(nullptr, RT_E_RCCL, "%s", rccl_unavailable());
    rt_group *g = new (std::nothrow) rt_group();
    if (!g) return ctx_fail(nullptr, RT_E_OOM, "out of host memory");
    for (int i = 0; i < n; ++i) {
        rt_ctx *ctx = nullptr;
        int st = rt_create(device_ids[i], &ctx);      // (leaves its message in the creation error)
        if (st != RT_OK) { rt_group_destroy(g); return st; }
        g->ctxs.push_back(ctx);
    }
    g->comms.assign((size_t)n, nullptr);
    ncclResult_t r = n > 1 ? R->CommInitAll(g->comms.data(), n, device_ids) : ncclSuccess;
    if (r != ncclSuccess) {
        g->comms.clear();
        rt_group_destroy(g);
        return ctx_fail(nullptr, RT_E_RCCL, "ncclCommInitAll over %d devices failed: %s", n, R->GetErrorString(r));
    }
    *out = g;
    return RT_OK;
}

int rt_group_size(const rt_group *g) { return g ? (int)g->ctxs.size() : 0; }

rt_ctx *rt_group_ctx(rt_group *g, int i) { return (g && i >= 0 && (size_t)i < g->ctxs.size()) ? g->ctxs[(size_t)i] : nullptr; }

int rt_group_reduce_to_root(rt_group *g, float *const *rgba_sums, int64_t n_floats, int root, void *const *streams) {
    if (!g || g->ctxs.empty()) return RT_E_INVALID;
    rt_ctx *c0 = g->ctxs[0];
    const int n = (int)g->ctxs.size();
    if (!rgba_sums || n_floats < 0 || root < 0 || root >= n) return ctx_fail(c0, RT_E_INVALID, "rt_group_reduce_to_root: bad arguments (root %d of %d)", root, n);
    for (int i = 0; i < n; ++i)
        if (!rgba_sums[i]) return ctx_fail(c0, RT_E_INVALID, "rt_group_reduce_to_root: buffer %d is NULL", i);
    if (n_floats == 0 || n == 1) return RT_OK;         // one device: its buffer is the sum
    const Rccl *R = rccl();
    if (!R) return ctx_fail(c0, RT_E_RCCL, "%s", rccl_unavailable());
    // one process drives all devices: the n calls of the one collective go between GroupStart / GroupEnd
    RCCL_TRY(c0, R, R->GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < n; ++i) {
        ncclResult_t r = R->Reduce(rgba_sums[i], rgba_sums[i], (size_t)n_floats, ncclFloat, ncclSum, root, g->comms[(size_t)i],
                                   streams ? (hipStream_t)streams[i] : nullptr);
        if (r != ncclSuccess && bad == ncclSuccess) bad = r;
    }
    ncclResult_t e = R->GroupEnd();
    if (bad != ncclSuccess || e != ncclSuccess) return ctx_fail(c0, RT_E_RCCL, "ncclReduce over %d devices failed: %s", n, R->GetErrorString(bad != ncclSuccess ? bad : e));
    return RT_OK;
}

void rt_group_destroy(rt_group *g) {
    if (!g) return;
    const Rccl *R = rccl();
    for (size_t i = 0; i < g->comms.size(); ++i)
        if (R && g->comms[i]) { (void)hipSetDevice(ctx_device(g->ctxs[i])); (void)R->CommDestroy(g->comms[i]); }
    for (rt_ctx *c : g->ctxs) rt_destroy(c);
    delete g;
}

}  // extern "C"
