// rt_device.hip — C-ABI of include/rt_api.h over the kernels of rt_kernels.h.
//
// Replaces the worker pool of RaytracerState::new (raytracer/src/main.rs:1314-1409).  One
// rt_ctx = one GPU; scene and sampler tables stay resident in HBM between rt_render calls
// (they are a few hundred KB to a few hundred MB; 288 GB leaves room for any of the
// BASELINE scenes many times over), the accumulation image is the caller's.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "rt_bvh_build.h"
#include "rt_device_internal.h"
#include "rt_kernels.h"
#include "rt_wavefront.h"

using namespace rtk;

constexpr int kMaxPipes = 4;
constexpr int kFrames = 2;     // renders in flight per context: each has a workspace of its own

// Workspace of one render in flight.  rt_render returns once a frame's last kernels are queued; the frame's workspace is
// taken again by the render after next, which first waits for `done` (settle()).
struct WfSpace {
    void *pool = nullptr;      size_t pool_entries = 0;  int pool_pipes = 0;   // 2 buffers x entries x 72 B for each of pool_pipes pipelines
    float4 *samples = nullptr; size_t samples_entries = 0;
    WfCounters *ctr = nullptr;                                  // per pipeline: ring + status words behind it
    StackEntry *ovf = nullptr; size_t ovf_entries = 0;          // traversal stack overflow (entries beyond the LDS part)
    uint32_t *host_status = nullptr;                            // pinned, per pipeline 2 x 4 words
    hipStream_t stream[kMaxPipes] = {};                         // one per pipeline
    hipEvent_t event[kMaxPipes][2] = {};
    hipEvent_t fork = nullptr, join[kMaxPipes] = {}, done = nullptr;
    // the frame in flight: `done` follows its fold on the caller's stream; the pipelines whose last iterations were queued
    // without waiting for their status leave it in host_status[8 q + 4 final_at[q]], to be looked at when the frame has settled
    bool in_flight = false;
    int final_at[kMaxPipes] = {-1, -1, -1, -1};                 // -1: the host has seen this pipeline drained
    uint32_t final_items[kMaxPipes] = {};
};

struct rt_ctx {
    int device = 0;
    std::string error;
    hipStream_t stream = nullptr;  // library-owned stream for host-buffer calls
    std::vector<void *> scene_allocs, sampler_allocs;
    DScene sc{};
    bool have_scene = false, have_sampler = false;
    uint32_t stack_depth = 1;
    uint32_t n_bvh_nodes = 0, built_on_device = 0;
    bool bvh_has_general_xform = false;   // some slot under a general Transform sits in a BVH leaf
    int features = EXT_ALL;               // EXT_* bits the uploaded scene needs (rt_kernels.h); picks the kernel set
    float bvh_build_ms = 0.0f;
    unsigned long long *d_stats = nullptr;  // RT_STATS_TOTAL_WORDS: counters + region timers of a render, the tail kernel's own timers from word 64
                                            // (development builds), the counters per kernel from word RT_KSTAT_BASE (counting variant)
    rt_stats kernel_stats[RT_K_COUNT] = {}; // ... of the last render with collect_stats (rt_last_kernel_stats)
    bool have_kernel_stats = false;
    hipDeviceProp_t prop{};
    // wavefront workspaces (grown on demand, kept until rt_destroy): one per render in flight (render_wavefront)
    WfSpace wf[kFrames];
    int wf_last = 0;                                            // the workspace the last render used
    char *noalloc_scratch = nullptr;   // = g_noalloc_ptr[device]
};

// Scratch of rt_trace_first_hit_noalloc (rt_api.h): staging of the rays and results plus the traversal-stack overflow, in a
// static device array instead of a hipMalloc'd buffer.  Every GPU the library runs on holds its own copy at its own address
// until the process ends.  In host code this name is only a handle for hipGetSymbolAddress: the address kernels and copies
// use is the one rt_create looks up per device (g_noalloc_ptr), never the name itself.
static __device__ __align__(256) unsigned char g_noalloc_scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int kMaxDevices = 64;
void *g_noalloc_ptr[kMaxDevices] = {};   // device address of g_noalloc_scratch on each device (set once by rt_create, same value every time)
std::mutex g_noalloc_mutex[kMaxDevices]; // rt_trace_first_hit_noalloc: one call at a time per device

thread_local std::string g_create_error;

int fail(rt_ctx *ctx, int status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf; else g_create_error = buf;
    return status;
}

#define HIP_TRY(ctx, call)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

void free_all(std::vector<void *> &v) {
    for (void *p : v) (void)hipFree(p);
    v.clear();
}

// Copy a host array to a fresh device allocation (at least 16 bytes so that empty tables still
// have a valid, 16-byte-aligned address).
template <class T>
int upload(rt_ctx *ctx, std::vector<void *> &owner, const void *src, size_t bytes, const T **dst) {
    void *d = nullptr;
    size_t alloc = bytes < 64 ? 64 : bytes;
    HIP_TRY(ctx, hipMalloc(&d, alloc));
    owner.push_back(d);
    HIP_TRY(ctx, hipMemsetAsync(d, 0, alloc < 4096 ? alloc : 64, ctx->stream));
    if (bytes && src) HIP_TRY(ctx, hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));   // src == null: allocate only
    *dst = static_cast<const T *>(d);
    return RT_OK;
}

int set_device(rt_ctx *ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return RT_OK;
}

// Tuning knobs (RT_WF_*, RT_NOISE_COOP, RT_KERNEL_SET: INTEGRATION.md) exist for the tests, which prove that none of them changes
// a bit of the image, and for A/B measurements.  They are read from the environment only when RT_KNOBS is set there as well, so
// a product run is what this file's defaults say, whatever else its environment holds.
const char *knob(const char *name) { return getenv("RT_KNOBS") ? getenv(name) : nullptr; }

size_t env_size(const char *name, size_t dflt, size_t lo, size_t hi) {
    if (const char *e = knob(name)) {
        long long v = atoll(e);
        if (v >= (long long)lo && v <= (long long)hi) return (size_t)v;
    }
    return dflt;
}

// Kernel sets (rt_kernels.h, EXT): the feature sets the shade / generate / tail kernels are instantiated for.  A scene runs on
// the smallest set that covers what it contains; the counting variants exist for the full set only (same paths, same bits).
constexpr int KS_LITE = EXT_NONE;                               // spheres, rectangles, Translate / RotateY, solid + checker: C1, C2
constexpr int KS_MESH = EXT_TRI | EXT_IMAGE;                    // + triangle meshes and image textures: C4
constexpr int KS_BOOK2 = EXT_NOISE | EXT_MEDIA | EXT_IMAGE;     // + Perlin noise, media, image textures: C3, C5
constexpr int KS_FULL = EXT_ALL & ~EXT_CHAIN;                   // + quadrics and general Transforms
constexpr int KS_CHAIN = EXT_ALL;                               // + wrappers nested in any order (chains of xform records)
int kernel_set_for(int need) {
    if ((need & ~KS_LITE) == 0) return KS_LITE;
    if ((need & ~KS_MESH) == 0) return KS_MESH;
    if ((need & ~KS_BOOK2) == 0) return KS_BOOK2;
    if ((need & ~KS_FULL) == 0) return KS_FULL;
    return KS_CHAIN;
}
template <bool STATS> struct WfKernels {
    decltype(&wf_generate_kernel<STATS, KS_CHAIN>) generate;
    decltype(&wf_shade_kernel<STATS, KS_CHAIN>) shade;
    decltype(&wf_tail_kernel<STATS, KS_CHAIN>) tail;
    decltype(&wf_traverse_kernel<STATS, EXT_NONE>) traverse;
    decltype(&wf_traverse_kernel<STATS, EXT_NONE, true>) traverse_top;   // ... with the top of the tree in LDS (TopLds)
};
template <bool STATS, int KS> WfKernels<STATS> wf_kernels_of(bool tree_has_xform, bool tree_has_tris) {
    constexpr int XF = EXT_XFORM | EXT_TRI | (KS & EXT_CHAIN);
    WfKernels<STATS> k;
    k.generate = wf_generate_kernel<STATS, KS>;
    k.shade = wf_shade_kernel<STATS, KS>;
    k.tail = wf_tail_kernel<STATS, KS>;
    // the walk's leaf step: spheres and rectangles, + triangles, + slots under a general Transform
    k.traverse = tree_has_xform ? wf_traverse_kernel<STATS, XF> : tree_has_tris ? wf_traverse_kernel<STATS, EXT_TRI> : wf_traverse_kernel<STATS, EXT_NONE>;
    k.traverse_top = tree_has_xform ? wf_traverse_kernel<STATS, XF, true> : tree_has_tris ? wf_traverse_kernel<STATS, EXT_TRI, true> : wf_traverse_kernel<STATS, EXT_NONE, true>;
    return k;
}
template <bool STATS> WfKernels<STATS> wf_kernels(int features, bool tree_has_xform) {
    const bool tris = (features & EXT_TRI) != 0;
    if constexpr (STATS) {
        return wf_kernels_of<STATS, KS_CHAIN>(tree_has_xform, tris);
    } else {
        switch (kernel_set_for(features)) {
        case KS_LITE: return wf_kernels_of<STATS, KS_LITE>(tree_has_xform, tris);
        case KS_MESH: return wf_kernels_of<STATS, KS_MESH>(tree_has_xform, tris);
        case KS_BOOK2: return wf_kernels_of<STATS, KS_BOOK2>(tree_has_xform, tris);
        case KS_FULL: return wf_kernels_of<STATS, KS_FULL>(tree_has_xform, tris);
        default: return wf_kernels_of<STATS, KS_CHAIN>(tree_has_xform, tris);
        }
    }
}

constexpr uint32_t kMaxStackDepth = 256;   // traversal stack entries; beyond the LDS part they live in HBM
constexpr size_t kMaxTopNodes = 1024;      // breadth-first table of the tree's top (DScene::top_order): more than any launch stages
constexpr size_t kEntryBytes = 72;   // rt_wavefront.h: 4 float4 + 1 float2 per pool entry

PathQ pool_buffer(const WfSpace &S, int which) {
    const size_t n = S.pool_entries;
    char *base = static_cast<char *>(S.pool) + (size_t)which * kEntryBytes * n;
    PathQ q;
    q.ray_o = reinterpret_cast<float4 *>(base); q.ray_d = q.ray_o + n; q.thr = q.ray_d + n; q.hit = q.thr + n;
    q.rng = reinterpret_cast<float2 *>(q.hit + n);
    return q;
}

// (m, k) with floor(n / d) == (n * m) >> k for all n < 2^26 (Granlund & Montgomery): k = 26 + ceil(log2 d).
void magic_div(uint32_t d, uint32_t *m, uint32_t *k) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    *k = 26 + l;
    *m = (uint32_t)(((1ull << *k) / d) + 1);
}

// Hand-out order of a pipeline's tiles (rt_wavefront.h, wf_generate_kernel): a stride near the golden section of their number,
// coprime to it, with (n - 1) * stride < 2^32.  RT_WF_TILE_STRIDE=1 restores raster order (A/B).
uint32_t tile_stride_for(uint32_t n) {
    if (n < 4u) return 1u;
    const size_t forced = env_size("RT_WF_TILE_STRIDE", 0, 0, 0xffffffffu);
    uint64_t s = forced ? forced % n : (uint64_t)((double)n * 0.6180339887498949);
    const uint64_t cap = 0xffffffffull / n;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
    while (s > 1 && gcd(s, n) != 1) --s;
    return (uint32_t)s;
}

// Wait for the frame that used this workspace last.  Pipelines whose last iterations were queued without waiting for their
// status (render_wavefront) must have ended with an empty pool and every item handed out: the bound on their iterations says
// so, and this is where a violation of it would be reported.
int settle(rt_ctx *ctx, WfSpace &S) {
    if (!S.in_flight) return RT_OK;
    S.in_flight = false;
    HIP_TRY(ctx, hipEventSynchronize(S.done));
    for (int q = 0; q < kMaxPipes; ++q) {
        if (S.final_at[q] < 0) continue;
        const uint32_t *hs = S.host_status + 8 * q + 4 * S.final_at[q];
        S.final_at[q] = -1;
        if (hs[0] != 0u || hs[1] != S.final_items[q])
            return fail(ctx, RT_E_HIP, "the previous render ended with %u paths alive and %u of %u items handed out (pipeline %d)", hs[0], hs[1], S.final_items[q], q);
    }
    return RT_OK;
}
int settle_all(rt_ctx *ctx) {
    int rc = RT_OK;
    for (WfSpace &S : ctx->wf) { const int r = settle(ctx, S); if (rc == RT_OK) rc = r; }
    return rc;
}

// The streaming pipeline of rt_wavefront.h for one rt_render call.  Samples are rendered in batches
// that fit the sample buffer.  A batch's items are split over RT_WF_PIPES independent pipelines (own
// pool, counters and HIP stream): traverse is bound by memory latency and LDS, shade by registers,
// so kernels of two pipelines running side by side on a CU use what the other leaves idle.
// Per pipeline the host enqueues iterations (traverse, shade, generate) eight at a time and looks at
// the pool's status two groups behind, so the GPU always has work queued while the host decides
// whether the pool has drained.
template <bool STATS>
int render_wavefront(rt_ctx *ctx, const RenderArgs &a, const rt_render_params *p, float4 *d_out, uint32_t *d_rays, hipStream_t stream) {
    const uint32_t tiles_x = (uint32_t)(a.x1 - a.x0 + 7) / 8u, tiles_y = (uint32_t)(a.y1 - a.y0 + 7) / 8u;
    const uint32_t n_tiles = tiles_x * tiles_y;
    const size_t per_sample = (size_t)n_tiles * 64u;                                    // items per sample of the window
    const size_t cap_entries = env_size("RT_WF_SAMPLES_MB", 8192, 1, 262144) * (1024 * 1024 / 16);
    if (per_sample > cap_entries || per_sample >= (1ull << 31)) return fail(ctx, RT_E_OOM, "window of %u tiles does not fit the sample buffer", n_tiles);
    size_t spp_batch_max = cap_entries / per_sample;
    if (spp_batch_max * per_sample >= (1ull << 32)) spp_batch_max = ((1ull << 32) - 1) / per_sample;
    if (spp_batch_max > (size_t)a.spp_count) spp_batch_max = (size_t)a.spp_count;
    const bool has_bvh = ctx->sc.root_ref != RT_BVH_EMPTY;
    ctx->sc.noise_coop = (uint32_t)env_size("RT_NOISE_COOP", 32, 0, 64);   // rt_kernels.h coop_turbulence (0: every noise hit by its own lane)
    const size_t shard_q = (size_t)RT_WF_SHARDS * 256;    // pool sizes are multiples of this
    const size_t need_samples = spp_batch_max * per_sample;
    // Pipelines: two, or three for a batch with enough items — how many is enough depends on what the third pipeline has to share
    // (profiles/r3/ab_pipelines_3.txt, the rounds at its end): without a tree (the Cornell box, Blocks, the small scenes) three win from
    // 2^25 items on (+6 ... 7.5 %; +1 ... 3 % on the small scenes); with a tree that is held in LDS (Book-1: 234 nodes) from 2^26
    // (Book-1 at 100 spp +2.3 %, at 1000 spp +4 %, but at 40 spp -3.6 % and at 10 spp -8 %: a short frame is three tails instead
    // of two); with a larger tree (Book-2: 1 755 nodes) only for the largest batches (+0.5 % at 1000 spp, -2 ... -3 % at 60 - 200 spp);
    // never on the 2.26 M-triangle mesh (-3 ... -6.5 %: three walks at once over a 170 MB tree).
    const size_t pipes3_from = !has_bvh ? ((size_t)1 << 25) : ctx->n_bvh_nodes <= 256u ? ((size_t)1 << 26)
                             : ctx->n_bvh_nodes < (1u << 17) ? ((size_t)1 << 28) : ~(size_t)0;
    int n_pipes = (int)env_size("RT_WF_PIPES", need_samples >= pipes3_from ? 3 : 2, 1, kMaxPipes);
    if (need_samples < 4 * shard_q || n_tiles < 2u) n_pipes = 1;
    if ((uint32_t)n_pipes > n_tiles) n_pipes = (int)n_tiles;      // every pipeline gets at least one tile (magic_div(tiles_pipe) divides by it)
    // Pool: 2^23 entries over all pipelines — what fits the caches, and the best size for every call of up to a few hundred
    // million items and for scenes without a tree at any size (the Cornell box loses 3 ... 13 % with more) — and 2^25 for a call
    // of 2^28 items or more on a scene with a tree: a launch of the traversal kernel ends with its longest walks, and four times
    // the rays per launch pay for that four times less often.  Measured (profiles/r3/ab_pool_by_call_size.txt): Book-2 from 500 spp
    // +3 %, the mesh at 256 spp +4.6 %, Book-1 at 1000 spp -0.3 %; 2^26 and 2^27 are 1 % behind 2^25 again.  (The same file
    // keeps the numbers that first led here, +20 ... +30 % for Book-2 with 2^27: those were the serial folds of
    // rt_kernels.h coop_exhaustive, which a larger pool met less often per ray.)
    const uint64_t call_items = (uint64_t)per_sample * (uint64_t)a.spp_count;
    size_t pool_log2 = env_size("RT_WF_POOL_LOG2", has_bvh && call_items >= (1ull << 28) ? 25 : 23, 14, 28);
    size_t pool_want = 0;
    auto set_pool_want = [&]() {       // per pipeline
        pool_want = (((size_t)1 << pool_log2) / (size_t)n_pipes) / shard_q * shard_q;
        if (pool_want < shard_q) pool_want = shard_q;
    };
    set_pool_want();
    // workspace: the one the last render did not use, once the frame that used it before has settled.  With two of them a
    // frame's last launches (C1 at 10 spp: 0.6 of its 2.5 ms are the tail kernel on a few hundred waves) run beside the next
    // frame's first ones when the caller orders the two on different streams; RT_WF_FRAMES=1 keeps one workspace, and a render
    // that reads counters back (collect_stats, the timers of a development build) has the device to itself.
    const int n_frames = (STATS || kTimers) ? 1 : (int)env_size("RT_WF_FRAMES", kFrames, 1, kFrames);
    if (n_frames == 1) { const int rc = settle_all(ctx); if (rc != RT_OK) return rc; }
    ctx->wf_last = n_frames == 1 ? 0 : (ctx->wf_last + 1) % n_frames;
    WfSpace &S = ctx->wf[ctx->wf_last];
    { const int rc = settle(ctx, S); if (rc != RT_OK) return rc; }
    for (int &f : S.final_at) f = -1;     // (a call that failed or was cancelled half-way may have left some set)
    if (S.samples_entries < need_samples) {
        if (S.samples) { (void)hipDeviceSynchronize(); (void)hipFree(S.samples); S.samples = nullptr; S.samples_entries = 0; }
        HIP_TRY(ctx, hipMalloc((void **)&S.samples, need_samples * 16));
        S.samples_entries = need_samples;
    }
    const size_t per_pipe_items = (need_samples + n_pipes - 1) / n_pipes;
    size_t pool = per_pipe_items < pool_want ? (per_pipe_items + shard_q - 1) / shard_q * shard_q : pool_want;
    while (S.pool_entries < pool || S.pool_pipes < n_pipes) {   // two buffers of the largest pool seen for each of the most pipelines seen
        const size_t entries = pool > S.pool_entries ? pool : S.pool_entries;
        const int pipes = n_pipes > S.pool_pipes ? n_pipes : S.pool_pipes;
        if (S.pool) { (void)hipDeviceSynchronize(); (void)hipFree(S.pool); S.pool = nullptr; S.pool_entries = 0; S.pool_pipes = 0; }
        const hipError_t e = hipMalloc(&S.pool, entries * 2 * kEntryBytes * (size_t)pipes);
        if (e == hipSuccess) { S.pool_entries = entries; S.pool_pipes = pipes; break; }
        S.pool = nullptr;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || pool_log2 <= 23) return fail(ctx, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP, "path pool of %zu entries: %s", entries * 2 * (size_t)pipes, hipGetErrorString(e));
        pool_log2 -= 2;          // a large pool is a matter of speed only: go on with a smaller one
        set_pool_want();
        pool = per_pipe_items < pool_want ? (per_pipe_items + shard_q - 1) / shard_q * shard_q : pool_want;
    }
    // counters, status words, streams and events: made when first needed and kept (what exists after a failure is used by the
    // next call or released by rt_destroy).  (Measured and dropped, profiles/r3/ab_frames_in_flight.txt: a second stream of the
    // highest priority per pipeline for the iterations after the host has seen the pool in tail mode — with two frames in flight
    // 6 % slower than without, and a hang with three pipelines once such streams shared a hardware queue.)
    {
        hipError_t e = hipSuccess;
        if (!S.ctr) e = hipMalloc((void **)&S.ctr, kMaxPipes * (RT_WF_RING + 1) * sizeof(WfCounters));   // per pipe: ring + one block for the status words
        if (e == hipSuccess && !S.host_status) e = hipHostMalloc((void **)&S.host_status, kMaxPipes * 2 * 4 * sizeof(uint32_t), hipHostMallocDefault);
        if (e == hipSuccess && !S.fork) e = hipEventCreateWithFlags(&S.fork, hipEventDisableTiming);
        if (e == hipSuccess && !S.done) e = hipEventCreateWithFlags(&S.done, hipEventDisableTiming);
        for (int q = 0; q < n_pipes && e == hipSuccess; ++q) {
            if (!S.stream[q]) e = hipStreamCreateWithFlags(&S.stream[q], hipStreamNonBlocking);
            for (hipEvent_t &ev : S.event[q]) if (e == hipSuccess && !ev) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess && !S.join[q]) e = hipEventCreateWithFlags(&S.join[q], hipEventDisableTiming);
        }
        if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP, "render workspace set-up failed: %s", hipGetErrorString(e));
    }
    // traversal stack: the first lds_depth entries of every lane in LDS, the (rarely reached) rest in HBM
    // The top of the tree (breadth-first) in LDS behind the stacks (rt_kernels.h, TopLds).  Measured (profiles/r3/
    // ab_traverse_top_nodes_lds*.txt): a tree that fits whole (Book-1: 234 nodes, 15 KB) with 12 stack levels in LDS instead of
    // 20 is worth 2.6 % of the 10-spp frame and 4 % at 100 spp; the top 44-170 nodes of a larger tree (Book-2: 1 170 nodes, the
    // mesh: 400 000) are worth nothing (-2 ... +0.3 %), so those keep the LDS for stacks and waves.
    const bool small_tree = ctx->sc.n_top != 0 && ctx->n_bvh_nodes <= 256u;
    const uint32_t top_want = (uint32_t)env_size("RT_WF_TOP_NODES", small_tree ? 256 : 0, 0, kMaxTopNodes);
    const uint32_t n_top = top_want < ctx->sc.n_top ? top_want : ctx->sc.n_top;
    const uint32_t lds_want = (uint32_t)env_size("RT_WF_LDS_DEPTH", n_top ? 12 : 20, 1, 64);
    const uint32_t lds_depth = lds_want < ctx->stack_depth ? lds_want : ctx->stack_depth;
    const size_t lds = (size_t)lds_depth * 256 * sizeof(StackEntry) + (size_t)n_top * 64;
    int per_cu = 0;
    // two instances of the traversal kernel: scenes with a general Transform inside the BVH need the code for it in the leaf step
    int features = ctx->features;
    if (const char *e = knob("RT_KERNEL_SET")) { if (!strcmp(e, "full")) features |= KS_FULL; }   // A/B: run a scene on the full set
    const WfKernels<STATS> K = wf_kernels<STATS>(features, ctx->bvh_has_general_xform);
    auto *const trav_kernel = n_top ? K.traverse_top : K.traverse;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trav_kernel, 256, lds));
    {   // the runtime's answer may assume the 64 KB per-workgroup LDS limit; gfx950 has 160 KB per CU
        hipFuncAttributes fa{};
        HIP_TRY(ctx, hipFuncGetAttributes(&fa, (const void *)trav_kernel));
        int regs = fa.numRegs > 0 ? (fa.numRegs + 7) / 8 * 8 : 128;
        int by_regs = 512 / regs > 8 ? 8 : 512 / regs;
        int by_lds = lds ? (int)((160 * 1024) / lds) : 8;
        int mine = by_regs < by_lds ? by_regs : by_lds;
        if (getenv("RT_DEBUG")) fprintf(stderr, "[rt] traverse: numRegs %d lds %zu runtime per_cu %d, computed %d, pipes %d\n", fa.numRegs, lds, per_cu, mine, n_pipes);
        if (mine > per_cu) per_cu = mine;
        per_cu = (int)env_size("RT_WF_TRAV_BLOCKS", (size_t)per_cu, 1, 16);
    }
    if (per_cu < 1) per_cu = 1;
    const unsigned trav_blocks = ((unsigned)(per_cu * ctx->prop.multiProcessorCount) + RT_WF_SHARDS - 1) / RT_WF_SHARDS * RT_WF_SHARDS;
    const size_t ovf_per_pipe = (size_t)(ctx->stack_depth - lds_depth) * trav_blocks * 256;
    if (ovf_per_pipe * kMaxPipes > S.ovf_entries) {
        if (S.ovf) { (void)hipDeviceSynchronize(); (void)hipFree(S.ovf); S.ovf = nullptr; S.ovf_entries = 0; }
        HIP_TRY(ctx, hipMalloc((void **)&S.ovf, ovf_per_pipe * kMaxPipes * sizeof(StackEntry)));
        S.ovf_entries = ovf_per_pipe * kMaxPipes;
    }
    const bool debug = getenv("RT_DEBUG") != nullptr;
    unsigned long long *d_stats = (STATS || kTimers) ? ctx->d_stats : nullptr;
    if (STATS || kTimers) HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, RT_STATS_TOTAL_WORDS * sizeof(unsigned long long), stream));
    // the tail kernel (rt_wavefront.h): once every item is out and at most RT_WF_TAIL paths live, an iteration is RT_WF_TAIL_BOUNCES
    // bounces of every path in one launch instead of traverse + shade; 0 = never.  Measured on MI355X against 0: C1 at 10 spp
    // 5.2 -> 3.1 ms, C4 at 4 spp +40 %, C2 at 16 spp +18 %, the long renders +1 %.
    auto *const tail_kernel = K.tail;
    const uint32_t tail_max = (uint32_t)env_size("RT_WF_TAIL", 131072, 0, 1u << 20);
    const int tail_every = (int)env_size("RT_WF_TAIL_EVERY", 2, 1, 16);   // while the host does not know whether every item is out: every n-th iteration may be a tail iteration
    const uint32_t tail_bounces = (uint32_t)env_size("RT_WF_TAIL_BOUNCES", 4, 1, 1u << 16);   // bounces of the first tail iteration (then the survivors are compacted)
    const uint32_t tail_thin = (uint32_t)env_size("RT_WF_TAIL_THIN", 1024, 0, 1u << 16);   // fewer paths per wave while that keeps the waves at or below this number (0: never)
    const uint32_t tail_bounces2 = (uint32_t)env_size("RT_WF_TAIL_BOUNCES2", 64, 1, 1u << 16); // ... of every later one
    const size_t tail_set_env = env_size("RT_WF_TAIL_SET", 64, 8, 64);
    const uint32_t tail_set = tail_set_env >= 64 ? 64u : tail_set_env >= 32 ? 32u : tail_set_env >= 16 ? 16u : 8u;   // paths per wave
    unsigned tail_waves = tail_max / tail_set + RT_WF_SHARDS;            // one lane per path, one wave per set (a shard's last set may be ragged)
    if (tail_waves < tail_thin + RT_WF_SHARDS) tail_waves = tail_thin + RT_WF_SHARDS;
    if (tail_waves > trav_blocks * 4u) tail_waves = trav_blocks * 4u;    // its lanes share the pipeline's stack overflow area
    tail_waves = (tail_waves + 3u) / 4u * 4u;
    const size_t tail_blk_env = env_size("RT_WF_TAIL_BLOCK", 256, 64, 256);
    const unsigned tail_blk = tail_blk_env >= 256 ? 256u : tail_blk_env >= 128 ? 128u : 64u;   // threads per workgroup (its waves are independent)
    const size_t tail_lds = (size_t)lds_depth * tail_blk * sizeof(StackEntry);
    const int tail_group = (int)env_size("RT_WF_TAIL_GROUP", 2, 1, 16);   // iterations enqueued at a time once every item is out
    const uint32_t tail_regroup = (uint32_t)env_size("RT_WF_TAIL_REGROUP", 1, 0, 1);   // survivors of a tail iteration are grouped by the slot they last hit
    const bool hand_off = !STATS && !kTimers && env_size("RT_WF_HAND_OFF", 1, 0, 1) != 0;   // return without waiting for the last status (0: wait, as before)
    const uint32_t tail_hint = (uint32_t)env_size("RT_WF_TAIL_HINT", 1, 0, 1);         // a bounce's walk starts with the sphere the path has just hit as a candidate

    struct Pipe {
        hipStream_t s;
        PathQ Q[2];
        WfCounters *ctr;
        uint32_t *d_status, *h_status;
        StackEntry *ovf;
        WfArgs w;
        uint32_t it, groups;
        unsigned chunks;          // shade / generate grid: chunks of 256 entries per shard (x), 64 shards (y)
        unsigned trav_blocks;     // traverse grid (a multiple of the 64 shards)
        uint32_t it_end;          // (once items_out) no path can be alive after this iteration
        bool items_out;           // the host has seen a status with every item handed out: the pool only shrinks from there
        bool drained;
        uint32_t known_out;       // items handed out as of the last status the host has seen
        uint32_t tail_from;       // every iteration from this one on was enqueued as a possible tail iteration (0: none yet)
    } pipe[kMaxPipes];
    for (int q = 0; q < n_pipes; ++q) {
        Pipe &P = pipe[q];
        P.s = n_pipes == 1 ? stream : S.stream[q];
        P.Q[0] = pool_buffer(S, 2 * q); P.Q[1] = pool_buffer(S, 2 * q + 1);
        P.ctr = S.ctr + (size_t)q * (RT_WF_RING + 1);
        P.d_status = reinterpret_cast<uint32_t *>(P.ctr + RT_WF_RING);
        P.h_status = S.host_status + (size_t)q * 8;
        P.ovf = ovf_per_pipe ? S.ovf + (size_t)q * ovf_per_pipe : nullptr;
        P.w = WfArgs{};
        P.w.a = a;
        P.w.tiles_x = tiles_x; P.w.n_tiles = n_tiles;
        magic_div(tiles_x, &P.w.tx_m, &P.w.tx_k);
    }
    for (int32_t s0 = 0; s0 < a.spp_count; s0 += (int32_t)spp_batch_max) {
        const uint32_t spp_batch = (uint32_t)((size_t)(a.spp_count - s0) < spp_batch_max ? (size_t)(a.spp_count - s0) : spp_batch_max);
        const bool last_batch = (size_t)s0 + spp_batch >= (size_t)a.spp_count;
        if (n_pipes > 1) {   // the pipelines start after whatever the caller queued before this call
            HIP_TRY(ctx, hipEventRecord(S.fork, stream));
            for (int q = 0; q < n_pipes; ++q) HIP_TRY(ctx, hipStreamWaitEvent(pipe[q].s, S.fork, 0));
        }
        for (int q = 0; q < n_pipes; ++q) {
            Pipe &P = pipe[q];
            WfArgs &w = P.w;
            w.spp_batch = spp_batch;
            w.sample0 = (uint32_t)(a.spp_begin + s0);
            magic_div(spp_batch, &w.spp_m, &w.spp_k);
            // pipe q renders the tiles t with t % n_pipes == q: whole tiles, so an item's index is the same with any number of pipes
            w.pipe = (uint32_t)q; w.n_pipes = (uint32_t)n_pipes;
            w.tiles_pipe = (n_tiles - (uint32_t)q + (uint32_t)n_pipes - 1u) / (uint32_t)n_pipes;
            // the balanced hand-out order (rt_wavefront.h) pays where traverse blocks wait for each other; a scene that is all
            // always-list (Cornell, Blocks) or nearly so (GeometricPrimitives: 7 quadrics and a tree of a dozen nodes) keeps raster
            // order, which is kinder to its texture and sample-buffer accesses (+6 % there)
            const bool balanced = has_bvh && ctx->n_bvh_nodes >= 64u;
            w.tile_stride = (balanced || knob("RT_WF_TILE_STRIDE")) ? tile_stride_for(w.tiles_pipe) : 1u;
            w.tail_max = tail_max;
            // the material deal of the shade kernel: not for the small feature set without a tree (Cornell-type scenes: nearly all
            // hits are one kind there); RT_WF_SORT_BLOCKS=0 / 1 forces it
            w.sort = (uint32_t)env_size("RT_WF_SORT_BLOCKS", (has_bvh || kernel_set_for(features) != KS_LITE) ? 1 : 0, 0, 1);
            w.tp_m = 0; w.tp_k = 0;
            if (env_size("RT_WF_SAMPLE_MAJOR", balanced ? 1 : 0, 0, 1)) magic_div(w.tiles_pipe, &w.tp_m, &w.tp_k);
            w.n_items = w.tiles_pipe * spp_batch * 64u;
            const size_t items_up = ((size_t)w.n_items + shard_q - 1) / shard_q * shard_q;
            w.pool = (uint32_t)(items_up < S.pool_entries ? items_up : S.pool_entries);
            if (w.pool > pool_want) w.pool = (uint32_t)pool_want;
            w.cap = w.pool / RT_WF_SHARDS;
            P.chunks = w.cap / 256u;
            // a batch that fits the pool is handed out whole by the first generate: the pool only shrinks from the start
            P.items_out = w.n_items <= w.pool;
            P.trav_blocks = trav_blocks;
            HIP_TRY(ctx, hipMemsetAsync(P.ctr, 0, (RT_WF_RING + 1) * sizeof(WfCounters), P.s));
            P.it = 0; P.groups = 0; P.drained = false; P.known_out = 0; P.tail_from = 0;
            P.it_end = P.items_out ? (uint32_t)(a.max_depth > 0 ? a.max_depth : 0) + 1u : 0xffffffffu;
            hipLaunchKernelGGL(K.generate, dim3(P.chunks, RT_WF_SHARDS), dim3(256), 0, P.s, ctx->sc, w, P.Q[1], P.ctr, P.it,
                               S.samples, P.d_status, d_stats, d_rays);
        }
        auto enqueue_iteration = [&](Pipe &P) {
            if (P.items_out && P.it >= P.it_end) return;   // every path has run out of depth by then: nothing left to enqueue
            const uint32_t it = ++P.it;
            const PathQ &cur = P.Q[it & 1], &nxt = P.Q[(it + 1) & 1];
            // may this iteration be a tail iteration (the kernels decide from the counters: WfShard::tail_mode)?  Once the host
            // knows that every item is out: each one; before that every tail_every-th, so that a batch whose status the host has
            // not seen yet does not wait for it
            // ... and not at all while more than a few pool fills of items are still to be handed out (a tail launch that finds
            // nothing to do still has to wait for registers and LDS to come free, which stalls its pipeline)
            const bool near_end = (uint64_t)(P.w.n_items - P.known_out) <= 8ull * (uint64_t)P.w.pool;
            const uint32_t with_tail = tail_max && (P.items_out || (near_end && (int)(it % (uint32_t)tail_every) == tail_every - 1)) ? 1u : 0u;
            if (with_tail && P.items_out) { if (!P.tail_from) P.tail_from = it; } else P.tail_from = 0;
            if (has_bvh) hipLaunchKernelGGL(trav_kernel, dim3(P.trav_blocks), dim3(256), lds, P.s, ctx->sc, cur, P.ctr, it, P.w.cap,
                                            lds_depth, P.ovf, d_stats, with_tail, n_top);
            hipLaunchKernelGGL(K.shade, dim3(P.chunks, RT_WF_SHARDS), dim3(256), 0, P.s, ctx->sc, P.w, cur, nxt, P.ctr, it,
                               S.samples, d_stats, d_rays, with_tail);
            if (with_tail)
                hipLaunchKernelGGL(tail_kernel, dim3(tail_waves * 64u / tail_blk), dim3(tail_blk), tail_lds, P.s, ctx->sc, P.w, cur, nxt, P.ctr, it,
                                   tail_bounces, tail_bounces2, tail_set, tail_thin, lds_depth, P.ovf, S.samples, d_stats, d_rays, tail_regroup, tail_hint);
            hipLaunchKernelGGL(K.generate, dim3(P.chunks, RT_WF_SHARDS), dim3(256), 0, P.s, ctx->sc, P.w, nxt, P.ctr, it,
                               S.samples, P.d_status, d_stats, d_rays);
        };
        // groups of 8 iterations while items are handed out; 2 in the tail, where what is enqueued after the pool has drained
        // (the host learns of it a group late) is pure waste
        auto group_size = [&](const Pipe &P) { return P.items_out ? tail_group : 8; };
        // the first group of every pipeline, iteration by iteration in turn: enqueued one pipeline after the other, the second
        // pipeline would start a whole group of launches late
        for (int k = 0; k < 8; ++k)
            for (int q = 0; q < n_pipes; ++q)
                if (k < group_size(pipe[q])) enqueue_iteration(pipe[q]);
        for (int live = n_pipes; live > 0;) {
            for (int q = 0; q < n_pipes; ++q) {
                Pipe &P = pipe[q];
                if (P.drained) continue;
                const uint32_t group = P.groups;
                if (group >= 2) {   // look at the status copied after group - 2
                    HIP_TRY(ctx, hipEventSynchronize(S.event[q][group & 1]));
                    const uint32_t *hs = P.h_status + 4 * (group & 1);
                    if (debug) fprintf(stderr, "[rt] pipe %d after iteration %u: %u paths live, %u of %u items out\n", q, hs[2], hs[0], hs[1], P.w.n_items);
                    if (hs[0] == 0u && hs[1] == P.w.n_items) { P.drained = true; --live; continue; }
                    P.known_out = hs[1];
                    if (p->cancel && *p->cancel) { (void)hipDeviceSynchronize(); return fail(ctx, RT_E_CANCELLED, "cancelled"); }
                    if (hs[3] != 0xffffffffu) {   // every item is out and the largest shard held hs[3] paths: the pool only shrinks from there,
                        // so the grids shrink with it (an empty iteration then costs a few microseconds, not 3 x 15).  A path loses one
                        // unit of depth per iteration, so max_depth iterations after this status (taken after iteration hs[2] - 1) the
                        // pool is empty whatever the scene: the host may enqueue that far ahead without waste.
                        if (!P.items_out) P.it_end = hs[2] + (uint32_t)(a.max_depth > 0 ? a.max_depth : 0) + 1u;
                        P.items_out = true;
                        const unsigned need = (hs[3] + 255u) / 256u;
                        P.chunks = need < 1u ? 1u : (need < P.w.cap / 256u ? need : P.w.cap / 256u);
                        const unsigned tb = P.chunks * RT_WF_SHARDS;
                        P.trav_blocks = tb < trav_blocks ? tb : trav_blocks;
                        if (tail_max && hs[0] <= tail_max) {   // tail mode (the pool only shrinks): iteration hs[2], which this status
                            // describes, and every one after it is a tail iteration if it was (or will be) enqueued as one.  The first
                            // of them takes at least min(tail_bounces, tail_bounces2) units of depth off every path (it may be the first
                            // tail iteration or a later one), each one after it tail_bounces2: after that many the pool is empty
                            // whatever the scene (Book-1 at depth 50: 4 + 64 bounces = 2 iterations, usually enqueued already)
                            const uint32_t depth = (uint32_t)(a.max_depth > 0 ? a.max_depth : 0);
                            const uint32_t first = tail_bounces < tail_bounces2 ? tail_bounces : tail_bounces2;
                            const uint32_t from = P.tail_from == 0u ? P.it + 1u : (hs[2] > P.tail_from ? hs[2] : P.tail_from);
                            const uint32_t end = from + (depth > first ? (depth - first + tail_bounces2 - 1u) / tail_bounces2 : 0u);
                            if (end < P.it_end) P.it_end = end;
                        }
                        // (belt and braces: everything enqueued has run and something is still alive — go on one iteration at a time)
                        if (hs[2] == P.it + 1u && P.it >= P.it_end) P.it_end = P.it + 1u;
                    }
                }
                if (group >= 1)      // (group 0 was enqueued above)
                    for (int k = 0, gsize = group_size(P); k < gsize; ++k) enqueue_iteration(P);
                HIP_TRY(ctx, hipGetLastError());
                HIP_TRY(ctx, hipMemcpyAsync(P.h_status + 4 * (group & 1), P.d_status, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, P.s));
                HIP_TRY(ctx, hipEventRecord(S.event[q][group & 1], P.s));
                ++P.groups;
                // everything this pipeline can need is queued (every item is out and no path has depth left after iteration
                // it_end): the host does not wait for the status that says so — settle() reads it when the workspace is next used
                if (hand_off && last_batch && P.items_out && P.it >= P.it_end) {
                    S.final_at[q] = (int)(group & 1); S.final_items[q] = P.w.n_items;
                    P.drained = true; --live;
                }
            }
        }
        if (n_pipes > 1)     // the fold (and whatever the caller queues next) waits for every pipeline
            for (int q = 0; q < n_pipes; ++q) {
                HIP_TRY(ctx, hipEventRecord(S.join[q], pipe[q].s));
                HIP_TRY(ctx, hipStreamWaitEvent(stream, S.join[q], 0));
            }
        WfArgs wf = pipe[0].w;
        hipLaunchKernelGGL(wf_fold_kernel, dim3((n_tiles * 64u + 255u) / 256u), dim3(256), 0, stream, wf, S.samples, d_out);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(S.done, stream));
    S.in_flight = true;
    if (kTimers && !STATS) {   // development build: print where the waves' cycles went
        unsigned long long h[128];
        HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        fprintf(stderr, "[rt-turns] %llu\n", h[RT_STAT_WORDS]);
        fprintf(stderr, "[rt-regions]");
        for (int i = 0; i < RT_N_REGIONS; ++i) fprintf(stderr, " %llu", h[RT_STAT_WORDS + 1 + i]);
        fprintf(stderr, "\n");
        if (h[64 + RT_STAT_WORDS]) {   // the tail kernel ran: its slowest wave (cycles), and the sums over its waves by region
            fprintf(stderr, "[rt-tail-slowest-wave-cycles] %llu\n[rt-tail-regions]", h[64 + RT_STAT_WORDS]);
            for (int i = 0; i < RT_N_REGIONS; ++i) fprintf(stderr, " %llu", h[64 + RT_STAT_WORDS + 1 + i]);
            fprintf(stderr, "\n");
        }
    }
    return RT_OK;
}

}  // namespace

namespace rtk_internal {
int ctx_device(const rt_ctx *ctx) { return ctx->device; }
int ctx_fail(rt_ctx *ctx, int status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return fail(ctx, status, "%s", buf);
}
}  // namespace rtk_internal

extern "C" {

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_create(int device_id, rt_ctx **out) {
    if (!out) return fail(nullptr, RT_E_INVALID, "rt_create: out is NULL");
    *out = nullptr;
    int n = rt_device_count();
    if (n <= 0) return fail(nullptr, RT_E_NO_DEVICE, "no HIP device is visible");
    if (device_id < 0 || device_id >= n) return fail(nullptr, RT_E_INVALID, "device %d out of range (have %d)", device_id, n);
    rt_ctx *ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return fail(nullptr, RT_E_OOM, "out of host memory");
    ctx->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->prop, device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_stats, RT_STATS_TOTAL_WORDS * sizeof(unsigned long long));
    if (e == hipSuccess && device_id < kMaxDevices) {   // this device's copy of the static scratch array (device_id is current)
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch));
        if (e == hipSuccess) { g_noalloc_ptr[device_id] = p; ctx->noalloc_scratch = static_cast<char *>(p); }
    }
    if (e != hipSuccess) {
        int st = fail(nullptr, RT_E_HIP, "rt_create: %s", hipGetErrorString(e));
        delete ctx;
        return st;
    }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();   // renders may still be in flight on the caller's streams and the pipelines'
    free_all(ctx->scene_allocs);
    free_all(ctx->sampler_allocs);
    if (ctx->d_stats) (void)hipFree(ctx->d_stats);
    for (WfSpace &S : ctx->wf) {
        if (S.in_flight && S.done) (void)hipEventSynchronize(S.done);
        for (hipStream_t st : S.stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        if (S.pool) (void)hipFree(S.pool);
        if (S.samples) (void)hipFree(S.samples);
        if (S.ctr) (void)hipFree(S.ctr);
        if (S.ovf) (void)hipFree(S.ovf);
        if (S.host_status) (void)hipHostFree(S.host_status);
        for (auto &evs : S.event) for (hipEvent_t ev : evs) if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : S.join) if (ev) (void)hipEventDestroy(ev);
        if (S.fork) (void)hipEventDestroy(S.fork);
        if (S.done) (void)hipEventDestroy(S.done);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *s) {
    if (!ctx) return RT_E_INVALID;
    if (!s || s->struct_size != sizeof(rt_scene_desc)) return fail(ctx, RT_E_INVALID, "rt_scene_upload: bad descriptor size");
    // shape checks: everything the kernels index must be consistent BEFORE anything is launched
    if (s->n_slots && !s->slots) return fail(ctx, RT_E_INVALID, "slots missing");
    if (s->n_bvh_nodes && !s->bvh_nodes) return fail(ctx, RT_E_INVALID, "bvh_nodes missing");
    if (s->n_media > RT_MAX_MEDIA) return fail(ctx, RT_E_UNSUPPORTED_PRIMITIVE, "at most %d ConstantMedium objects are supported", RT_MAX_MEDIA);
    if ((s->n_xforms && !s->xforms) || (s->n_media && !s->media) || (s->n_lights && !s->lights) || (s->n_materials && !s->materials) ||
        (s->n_textures && !s->textures))
        return fail(ctx, RT_E_INVALID, "a record array (xforms / media / lights / materials / textures) is missing");
    if (s->n_lights == 0) return fail(ctx, RT_E_INVALID, "`lights` is empty (the reference indexes objects[random_int(0,-1)], hittable_list.rs:71-75)");
    if ((uint64_t)s->always_begin + s->always_count > s->n_slots) return fail(ctx, RT_E_INVALID, "always-list outside the slot array");
    if (s->n_slots >= (1u << 28)) return fail(ctx, RT_E_INVALID, "too many primitives");
    const bool device_build = s->build_bvh_on_device != 0;
    if (device_build) {
        if (s->always_begin + s->always_count != s->n_slots) return fail(ctx, RT_E_INVALID, "device BVH build: the always-list must be the tail of the slot array");
        if (s->always_begin && !s->slot_boxes) return fail(ctx, RT_E_INVALID, "device BVH build: slot_boxes missing");
        if (s->bvh_leaf_max > 2) return fail(ctx, RT_E_INVALID, "bvh_leaf_max %u: 0, 1 or 2", s->bvh_leaf_max);
        if (s->build_bvh_on_device > RT_BVH_DEVICE_LBVH) return fail(ctx, RT_E_INVALID, "build_bvh_on_device %u: unknown builder", s->build_bvh_on_device);
    } else if (s->bvh_stack_depth == 0 || s->bvh_stack_depth > kMaxStackDepth)
        return fail(ctx, RT_E_INVALID, "bvh_stack_depth %u out of range", s->bvh_stack_depth);
    auto ref_ok = [&](int32_t ref) {
        if (ref == RT_BVH_EMPTY) return true;   // a 4-wide node may have fewer than four children
        if (ref >= 0) return (uint32_t)ref < s->n_bvh_nodes;
        uint32_t code = (uint32_t)~ref;
        return (uint64_t)(code >> 3) + (code & 7u) + 1u <= s->n_slots;
    };
    if (!device_build && s->bvh_root_ref != RT_BVH_EMPTY && !ref_ok(s->bvh_root_ref)) return fail(ctx, RT_E_INVALID, "bvh_root_ref out of range");
    for (uint32_t i = 0; !device_build && i < s->n_bvh_nodes; ++i) {
        int32_t refs[4];
        memcpy(refs, &s->bvh_nodes[(size_t)i * RT_NODE_FLOATS + 12], 16);
        for (int32_t r : refs)
            if (!ref_ok(r)) return fail(ctx, RT_E_INVALID, "bvh node %u has a child out of range", i);
    }
    // The walk pushes without a bound check and its persistent waves only end when every stack has drained: a node reachable
    // twice (a cycle or a shared subtree) or a stack_depth below what the tree needs would be a GPU fault or a hang.  Walk the
    // tree from the root: every inner node is reached at most once, and a visit at inner depth d (root = 1) can leave
    // 3 (d - 1) siblings on the stack and push 3 more, so the stack must hold 3 d entries (the host builder declares 3 d + 2).
    if (!device_build && s->bvh_root_ref != RT_BVH_EMPTY && s->bvh_root_ref >= 0) {
        std::vector<uint8_t> seen(s->n_bvh_nodes, 0);
        std::vector<std::pair<int32_t, uint32_t>> todo;   // (node, inner depth)
        todo.push_back({s->bvh_root_ref, 1u});
        uint32_t max_depth = 0;
        while (!todo.empty()) {
            const auto [node, depth] = todo.back();
            todo.pop_back();
            if (seen[(size_t)node]) return fail(ctx, RT_E_INVALID, "bvh node %d is reachable twice (cycle or shared subtree)", node);
            seen[(size_t)node] = 1;
            if (depth > max_depth) max_depth = depth;
            int32_t refs[4];
            memcpy(refs, &s->bvh_nodes[(size_t)node * RT_NODE_FLOATS + 12], 16);
            for (int32_t r : refs)
                if (r >= 0 && r != RT_BVH_EMPTY) todo.push_back({r, depth + 1u});
        }
        if (3u * max_depth > s->bvh_stack_depth)
            return fail(ctx, RT_E_INVALID, "bvh_stack_depth %u is too small for this tree: %u inner levels need %u entries", s->bvh_stack_depth, max_depth, 3u * max_depth);
    }
    bool tris = false, general_in_bvh = false;
    int need = s->n_media ? EXT_MEDIA : EXT_NONE;     // EXT_* features the scene needs (kernel set, rt_kernels.h)
    float r_min = INFINITY;      // smallest sphere radius (ray_is_far, rt_kernels.h)
    for (uint32_t i = 0; i < s->n_slots; ++i) {
        int32_t w4, mat;
        memcpy(&w4, &s->slots[i * RT_SLOT_FLOATS + 4], 4);
        memcpy(&mat, &s->slots[i * RT_SLOT_FLOATS + 5], 4);
        int type = w4 & 15, xf = (w4 >> 8) - 1;
        if (type == RT_PRIM_SPHERE || type == RT_PRIM_MOVING_SPHERE) {
            const float rad = fabsf(s->slots[i * RT_SLOT_FLOATS + 3]);
            if (rad > 0.0f && rad < r_min) r_min = rad;
        }
        if (type > RT_PRIM_HYPERBOLOID) return fail(ctx, RT_E_UNSUPPORTED_PRIMITIVE, "slot %u: primitive type %d", i, type);
        if (xf >= (int)s->n_xforms) return fail(ctx, RT_E_INVALID, "slot %u: xform out of range", i);
        bool general = false, chain = false;
        if (xf >= 0) {
            int32_t flags, m;
            memcpy(&flags, &s->xforms[(size_t)xf * RT_REC_FLOATS], 4);
            memcpy(&m, &s->xforms[(size_t)xf * RT_REC_FLOATS + 1], 4);
            general = (flags & 12) != 0;     // a general Transform or a chain: kernels with EXT_XFORM
            auto matrix_ok = [&](int32_t mi) { return mi >= 0 && (uint32_t)mi < s->n_matrices && s->matrices; };
            if ((flags & 4) && !matrix_ok(m)) return fail(ctx, RT_E_INVALID, "slot %u: matrix out of range", i);
            chain = (flags & 8) != 0;
            if (chain) {                     // a run of 2..64 plain records
                int32_t count;
                memcpy(&count, &s->xforms[(size_t)xf * RT_REC_FLOATS + 2], 4);
                if (flags != 8 || m < 0 || count < 2 || count > 64 || (uint64_t)m + (uint64_t)count > s->n_xforms)
                    return fail(ctx, RT_E_INVALID, "slot %u: xform chain outside the xform array", i);
                for (int32_t k = m; k < m + count; ++k) {
                    int32_t f2, m2;
                    memcpy(&f2, &s->xforms[(size_t)k * RT_REC_FLOATS], 4);
                    memcpy(&m2, &s->xforms[(size_t)k * RT_REC_FLOATS + 1], 4);
                    if ((f2 & ~7) || ((f2 & 4) && !matrix_ok(m2))) return fail(ctx, RT_E_INVALID, "slot %u: xform chain entry %d is not a plain record", i, k);
                }
            }
        }
        const bool in_always = i >= s->always_begin && i - s->always_begin < s->always_count;
        // quadrics may accept hits below t_min (cone.rs:61-65): the traversal kernels do not carry their code
        if (type >= RT_PRIM_CYLINDER && !in_always) return fail(ctx, RT_E_INVALID, "slot %u: quadrics must be in the always-list", i);
        general_in_bvh |= general && i < s->always_begin;      // (slots behind the always-list are media boundaries: no tree refers to them)
        if (general) need |= EXT_XFORM;
        if (chain) need |= EXT_CHAIN;
        if (type >= RT_PRIM_CYLINDER) need |= EXT_QUADRICS;
        if (type == RT_PRIM_TRIANGLE) need |= EXT_TRI;
        if (mat < 0 || (uint32_t)mat >= s->n_materials) return fail(ctx, RT_E_INVALID, "slot %u: material out of range", i);
        tris |= type == RT_PRIM_TRIANGLE;
    }
    {   // `lights`: plain records, or a root record (253) + runs of list members (254 records point at theirs, always behind themselves)
        std::vector<uint8_t> depth(s->n_lights, 0);     // nesting level of the list a record belongs to; 0 = belongs to none
        auto word = [&](uint32_t i, int k) { int32_t v; memcpy(&v, &s->lights[(size_t)i * RT_REC_FLOATS + k], 4); return v; };
        const bool rooted = word(0, 0) == 253;
        auto claim = [&](uint32_t owner, int level) {   // the run of a 253 / 254 record
            const int32_t first = word(owner, 1), count = word(owner, 2);
            if (first <= (int32_t)owner || count < 1 || (uint64_t)first + (uint64_t)count > s->n_lights || level > 4) return false;
            for (int32_t k = first; k < first + count; ++k) {
                if (depth[k]) return false;             // a record belongs to one list
                depth[k] = (uint8_t)level;
            }
            return true;
        };
        if (rooted && !claim(0, 1)) return fail(ctx, RT_E_INVALID, "lights: bad root record");
        for (uint32_t i = rooted ? 1u : 0u; i < s->n_lights; ++i) {
            const int32_t type = word(i, 0);
            const bool known = type == RT_PRIM_SPHERE || type == RT_PRIM_MOVING_SPHERE || (type >= RT_PRIM_XY_RECT && type <= RT_PRIM_YZ_RECT) ||
                               type == RT_PRIM_CYLINDER || type == 255 || (type == 254 && rooted);
            if (!known) return fail(ctx, RT_E_UNSUPPORTED_PRIMITIVE, "light %u: type %d", i, type);
            if (rooted && !depth[i]) return fail(ctx, RT_E_INVALID, "light %u belongs to no list", i);
            if (type == 254 && !claim(i, depth[i] + 1)) return fail(ctx, RT_E_INVALID, "light %u: bad list record (members behind it, at most 4 lists deep)", i);
            if (type == RT_PRIM_CYLINDER) need |= EXT_QUADRICS;    // its pdf_value runs the cylinder's hit test
        }
        if (rooted) need |= EXT_CHAIN;                  // the kernel set that carries the rarely needed loops
    }
    if (tris && !s->tri_attrs) return fail(ctx, RT_E_INVALID, "triangles without tri_attrs");
    for (uint32_t i = 0; i < s->n_moving_slots; ++i)
        if (s->moving_slots[i] >= s->n_slots) return fail(ctx, RT_E_INVALID, "moving slot out of range");
    for (uint32_t i = 0; i < s->n_rect_slots; ++i)
        if (s->rect_slots[i] >= s->n_slots) return fail(ctx, RT_E_INVALID, "rect slot out of range");
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        int32_t type, tex;
        memcpy(&type, &s->materials[i * RT_REC_FLOATS], 4);
        memcpy(&tex, &s->materials[i * RT_REC_FLOATS + 1], 4);
        if (type < 0 || type > RTG_MAT_ISOTROPIC) return fail(ctx, RT_E_UNSUPPORTED_PRIMITIVE, "material %u: type %d", i, type);
        bool textured = type == RTG_MAT_LAMBERTIAN || type == RTG_MAT_DIFFUSE_LIGHT || type == RTG_MAT_ISOTROPIC;
        if (textured && (tex < 0 || (uint32_t)tex >= s->n_textures)) return fail(ctx, RT_E_INVALID, "material %u: texture out of range", i);
    }
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        int32_t w[8];
        memcpy(w, &s->textures[i * RT_REC_FLOATS], 32);
        if (w[0] < 0 || w[0] > RTG_TEX_IMAGE) return fail(ctx, RT_E_UNSUPPORTED_PRIMITIVE, "texture %u: type %d", i, w[0]);
        if (w[0] == RTG_TEX_CHECKER && (w[4] < 0 || w[5] < 0 || (uint32_t)w[4] >= s->n_textures || (uint32_t)w[5] >= s->n_textures))
            return fail(ctx, RT_E_INVALID, "texture %u: checker child out of range", i);
        if (w[0] == RTG_TEX_NOISE) need |= EXT_NOISE;
        if (w[0] == RTG_TEX_IMAGE) need |= EXT_IMAGE;
        if (w[0] == RTG_TEX_NOISE && (w[4] < 0 || (uint32_t)w[4] >= s->n_perlin)) return fail(ctx, RT_E_INVALID, "texture %u: perlin table out of range", i);
        if (w[0] == RTG_TEX_IMAGE && w[4] >= 0 && (w[5] <= 0 || w[7] <= 0 || ((uint64_t)w[4] + (uint64_t)w[5] * (uint64_t)w[7]) * 4 > s->n_texel_bytes))
            return fail(ctx, RT_E_INVALID, "texture %u: image outside the texel atlas", i);
    }
    for (uint32_t i = 0; i < s->n_media; ++i) {
        int32_t mat;
        memcpy(&mat, &s->media[i * RT_REC_FLOATS + 5], 4);
        if (mat < 0 || (uint32_t)mat >= s->n_materials) return fail(ctx, RT_E_INVALID, "medium %u: material out of range", i);
        int32_t kind, first, count;
        memcpy(&kind, &s->media[i * RT_REC_FLOATS + 7], 4);
        if (kind == 0) continue;
        if (kind != 1) return fail(ctx, RT_E_UNSUPPORTED_PRIMITIVE, "medium %u: boundary kind %d", i, kind);
        memcpy(&first, &s->media[i * RT_REC_FLOATS], 4);
        memcpy(&count, &s->media[i * RT_REC_FLOATS + 1], 4);
        // boundary slots: behind the always-list, spheres / rectangles under any record (checked with the other slots above), solid
        // phase function; [2] = 1 + the record of the medium's own wrappers
        int32_t outer;
        memcpy(&outer, &s->media[i * RT_REC_FLOATS + 2], 4);
        if (outer < 0 || (uint32_t)outer > s->n_xforms) return fail(ctx, RT_E_INVALID, "medium %u: xform out of range", i);
        if (outer > 0) {
            int32_t flags;
            memcpy(&flags, &s->xforms[(size_t)(outer - 1) * RT_REC_FLOATS], 4);
            if (flags & 4) need |= EXT_XFORM;
            if (flags & 8) need |= EXT_CHAIN | EXT_XFORM;
        }
        if (first < 0 || count < 1 || count > 64 || (uint64_t)first < (uint64_t)s->always_begin + s->always_count || (uint64_t)first + (uint64_t)count > s->n_slots)
            return fail(ctx, RT_E_INVALID, "medium %u: boundary slots [%d, %d) outside the slot array's tail", i, first, first + count);
        for (int32_t k = first; k < first + count; ++k) {
            int32_t w4;
            memcpy(&w4, &s->slots[(size_t)k * RT_SLOT_FLOATS + 4], 4);
            const int type = w4 & 15;
            if (!(type == RT_PRIM_SPHERE || (type >= RT_PRIM_XY_RECT && type <= RT_PRIM_YZ_RECT)))
                return fail(ctx, RT_E_UNSUPPORTED_PRIMITIVE, "medium %u: boundary slot %d is not a sphere or rectangle", i, k);
        }
        // the phase function's texture: an Isotropic material (the one kind ConstantMedium::new makes, constant_medium.rs:22-38; its
        // texture index was range-checked above) whose texture reads no u, v — the kernels keep no u, v of a boundary slot's hit.  The
        // walk is texture_needs_uv()'s (rt_kernels.h): an image, or a checker with an image anywhere below it.
        int32_t mtype, tex;
        memcpy(&mtype, &s->materials[(size_t)mat * RT_REC_FLOATS], 4);
        memcpy(&tex, &s->materials[(size_t)mat * RT_REC_FLOATS + 1], 4);
        if (mtype != RTG_MAT_ISOTROPIC) return fail(ctx, RT_E_INVALID, "medium %u: material %d is not Isotropic", i, mat);
        bool reads_uv = false;
        std::vector<int32_t> todo(1, tex);
        for (size_t seen = 0; !todo.empty() && !reads_uv; ++seen) {
            if (seen > 4096) { reads_uv = true; break; }                  // a cycle of checkers: be conservative
            const int32_t t = todo.back();
            todo.pop_back();
            int32_t w[8];
            memcpy(w, &s->textures[(size_t)t * RT_REC_FLOATS], 32);       // (checker children were range-checked above)
            if (w[0] == RTG_TEX_IMAGE) reads_uv = true;
            else if (w[0] == RTG_TEX_CHECKER) { todo.push_back(w[4]); todo.push_back(w[5]); }
        }
        if (reads_uv) return fail(ctx, RT_E_UNSUPPORTED_PRIMITIVE, "medium %u: an image texture on a non-sphere boundary is not supported (u, v of the boundary hit)", i);
    }

    int st = set_device(ctx);
    if (st) return st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((st = settle_all(ctx)) != RT_OK) return st;     // renders in flight read the tables released below
    free_all(ctx->scene_allocs);
    ctx->have_scene = false;
    DScene &d = ctx->sc;
    auto &own = ctx->scene_allocs;
#define UP(field, src, bytes) if ((st = upload(ctx, own, src, bytes, &d.field)) != RT_OK) return st
    if (!device_build) UP(nodes, s->bvh_nodes, (size_t)s->n_bvh_nodes * RT_NODE_FLOATS * 4);
    d.n_top = 0;
    {   // the top of the tree in breadth-first order (DScene::top_order / top_child), for the walk that keeps it in LDS
        std::vector<uint32_t> order, child;
        if (!device_build && s->bvh_root_ref != RT_BVH_EMPTY && s->bvh_root_ref >= 0) {
            order.push_back((uint32_t)s->bvh_root_ref);
            for (size_t k = 0; k < order.size(); ++k) {       // (the queue is the order itself)
                int32_t refs[4];
                memcpy(refs, &s->bvh_nodes[(size_t)order[k] * RT_NODE_FLOATS + 12], 16);
                for (int32_t r : refs) {
                    uint32_t at = 0xffffffffu;
                    if (r >= 0 && r != RT_BVH_EMPTY && order.size() < kMaxTopNodes) { at = (uint32_t)order.size(); order.push_back((uint32_t)r); }
                    child.push_back(at);
                }
            }
            d.n_top = (uint32_t)order.size();
        }
        UP(top_order, order.data(), order.size() * 4);
        UP(top_child, child.data(), child.size() * 4);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the vectors go out of scope
    }
    {   // device copy of the slots: spheres and rectangles under a transform get that transform's record in
        // words 8..15 (rt_kernels.h SlotRec), so a hit test is one 64-byte fetch
        std::vector<float> dev_slots(s->slots, s->slots + (size_t)s->n_slots * RT_SLOT_FLOATS);
        for (uint32_t i = 0; i < s->n_slots; ++i) {
            int32_t w4;
            memcpy(&w4, &dev_slots[(size_t)i * RT_SLOT_FLOATS + 4], 4);
            int type = w4 & 15, xf = (w4 >> 8) - 1;
            if (xf >= 0 && type_has_inline_xform(type))
                memcpy(&dev_slots[(size_t)i * RT_SLOT_FLOATS + 8], &s->xforms[(size_t)xf * RT_REC_FLOATS], RT_REC_FLOATS * 4);
        }
        UP(slots, dev_slots.data(), dev_slots.size() * 4);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // dev_slots goes out of scope
    }
    UP(attrs, s->tri_attrs, tris ? (size_t)s->n_slots * RT_ATTR_FLOATS * 4 : 0);
    UP(xforms, s->xforms, (size_t)s->n_xforms * RT_REC_FLOATS * 4);
    UP(matrices, s->matrices, (size_t)s->n_matrices * RT_MATRIX_FLOATS * 4);
    UP(media, s->media, (size_t)s->n_media * RT_REC_FLOATS * 4);
    UP(lights, s->lights, (size_t)s->n_lights * RT_REC_FLOATS * 4);
    {   // device copy of the materials: word 7 = kind of the material's texture + 1 (0 = untextured), and a solid
        // colour is copied into words 2..4, so that most bounces need no texture fetch (rt_kernels.h material_texture)
        std::vector<float> dev_mats(s->materials, s->materials + (size_t)s->n_materials * RT_REC_FLOATS);
        for (uint32_t i = 0; i < s->n_materials; ++i) {
            float *m = &dev_mats[(size_t)i * RT_REC_FLOATS];
            int32_t type, tex, kind_plus_1 = 0;
            memcpy(&type, m, 4);
            memcpy(&tex, m + 1, 4);
            if (type == RTG_MAT_LAMBERTIAN || type == RTG_MAT_DIFFUSE_LIGHT || type == RTG_MAT_ISOTROPIC) {
                const float *t = &s->textures[(size_t)tex * RT_REC_FLOATS];
                int32_t kind;
                memcpy(&kind, t, 4);
                kind_plus_1 = kind + 1;
                if (kind == RTG_TEX_SOLID) memcpy(m + 2, t + 1, 12);
            }
            memcpy(m + 7, &kind_plus_1, 4);
        }
        UP(materials, dev_mats.data(), dev_mats.size() * 4);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // dev_mats goes out of scope
    }
    {   // device copy of the textures: a checker whose two branches are solid colours (the ground of C1, the spheres of
        // TwoSpheres) becomes one record of kind RT_TEX_CHECKER_SOLIDS holding both colours, so that its lookup is one
        // fetch instead of a checker record followed by a dependent fetch of the branch taken (rt_kernels.h texture_value)
        std::vector<float> dev_tex(s->textures, s->textures + (size_t)s->n_textures * RT_REC_FLOATS);
        for (uint32_t i = 0; i < s->n_textures; ++i) {
            int32_t w[8];
            memcpy(w, &s->textures[(size_t)i * RT_REC_FLOATS], 32);
            if (w[0] != RTG_TEX_CHECKER) continue;
            const float *odd = &s->textures[(size_t)w[4] * RT_REC_FLOATS], *even = &s->textures[(size_t)w[5] * RT_REC_FLOATS];
            int32_t ko, ke;
            memcpy(&ko, odd, 4); memcpy(&ke, even, 4);
            if (ko != RTG_TEX_SOLID || ke != RTG_TEX_SOLID) continue;
            float *t = &dev_tex[(size_t)i * RT_REC_FLOATS];
            const int32_t kind = RT_TEX_CHECKER_SOLIDS;
            const float scale = s->textures[(size_t)i * RT_REC_FLOATS + 6];
            memcpy(t, &kind, 4);
            memcpy(t + 1, odd + 1, 12);
            memcpy(t + 4, even + 1, 12);
            t[7] = scale;
        }
        UP(textures, dev_tex.data(), dev_tex.size() * 4);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // dev_tex goes out of scope
    }
    UP(perlin, s->perlin, (size_t)s->n_perlin * RT_PERLIN_WORDS * 4);
    UP(texels, s->texels, (size_t)s->n_texel_bytes);
    UP(moving, s->moving_slots, (size_t)s->n_moving_slots * 4);
    UP(rects, s->rect_slots, (size_t)s->n_rect_slots * 4);
#undef UP
    d.root_ref = s->bvh_root_ref;
    ctx->stack_depth = s->bvh_stack_depth;
    ctx->n_bvh_nodes = s->n_bvh_nodes;
    ctx->built_on_device = 0;
    ctx->bvh_has_general_xform = general_in_bvh;
    ctx->features = need;
    ctx->bvh_build_ms = 0.0f;
    if (device_build) {
        // Morton order decides where a slot goes: build over the boxes, then gather slots and triangle attributes into
        // that order and translate the two slot-index lists.  The always-list (the tail) stays where it is.
        const uint32_t nb = s->always_begin;
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t i = 0; i < nb; ++i)
            for (int a = 0; a < 3; ++a) {
                const float c = 0.5f * (s->slot_boxes[6 * (size_t)i + a] + s->slot_boxes[6 * (size_t)i + 3 + a]);
                if (c < lo[a]) lo[a] = c;
                if (c > hi[a]) hi[a] = c;
            }
        const float *d_boxes = nullptr;
        std::vector<void *> tmp;
        auto cleanup = [&](int rc) { (void)hipStreamSynchronize(ctx->stream); free_all(tmp); return rc; };
        auto retire = [&](const void *p) {   // scene_allocs -> tmp
            for (size_t k = 0; k < own.size(); ++k)
                if (own[k] == p) { tmp.push_back(own[k]); own.erase(own.begin() + (long)k); return; }
        };
        if ((st = upload(ctx, tmp, s->slot_boxes, (size_t)nb * 24, &d_boxes)) != RT_OK) return cleanup(st);
        rtb::Result res;
        std::string err;
        const uint32_t leaf_max = s->bvh_leaf_max ? s->bvh_leaf_max : (nb >= 100000u ? 2u : 1u);   // as the host builder chooses
        st = rtb::build_bvh(ctx->stream, d_boxes, nb, leaf_max, s->build_bvh_on_device, lo, hi, &res, &err);
        if (res.nodes) own.push_back(res.nodes);
        if (res.perm) tmp.push_back(res.perm);
        if (st != RT_OK) return cleanup(fail(ctx, st, "device BVH build: %s", err.c_str()));
        if (res.stack_depth > kMaxStackDepth) return cleanup(fail(ctx, RT_E_UNSUPPORTED_PRIMITIVE, "device BVH build: tree too deep (stack %u)", res.stack_depth));
        if (nb) {
            const float4 *sorted_slots = nullptr, *sorted_attrs = nullptr;
            const uint32_t *inv = nullptr;
            if ((st = upload(ctx, own, nullptr, (size_t)s->n_slots * RT_SLOT_FLOATS * 4, &sorted_slots)) != RT_OK) return cleanup(st);
            rtb::gather_records(ctx->stream, d.slots, res.perm, nb, RT_SLOT_FLOATS, const_cast<float4 *>(sorted_slots));
#define HIP_TRY_C(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return cleanup(fail(ctx, RT_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_))); } while (0)
            if (s->always_count)
                HIP_TRY_C(hipMemcpyAsync(const_cast<float4 *>(sorted_slots) + 4 * (size_t)nb, d.slots + 4 * (size_t)nb, (size_t)s->always_count * RT_SLOT_FLOATS * 4,
                                         hipMemcpyDeviceToDevice, ctx->stream));
            retire(d.slots);          // the unsorted copy goes when the gathers are done (cleanup)
            d.slots = sorted_slots;
            if (tris) {
                if ((st = upload(ctx, own, nullptr, (size_t)s->n_slots * RT_ATTR_FLOATS * 4, &sorted_attrs)) != RT_OK) return cleanup(st);
                rtb::gather_records(ctx->stream, d.attrs, res.perm, nb, RT_ATTR_FLOATS, const_cast<float4 *>(sorted_attrs));
                if (s->always_count)
                    HIP_TRY_C(hipMemcpyAsync(const_cast<float4 *>(sorted_attrs) + 4 * (size_t)nb, d.attrs + 4 * (size_t)nb, (size_t)s->always_count * RT_ATTR_FLOATS * 4,
                                             hipMemcpyDeviceToDevice, ctx->stream));
                retire(d.attrs);
                d.attrs = sorted_attrs;
            }
            if ((st = upload(ctx, tmp, nullptr, (size_t)nb * 4, &inv)) != RT_OK) return cleanup(st);
            rtb::invert_permutation(ctx->stream, res.perm, nb, const_cast<uint32_t *>(inv));
            rtb::remap_slot_list(ctx->stream, const_cast<uint32_t *>(d.moving), s->n_moving_slots, inv, nb);
            rtb::remap_slot_list(ctx->stream, const_cast<uint32_t *>(d.rects), s->n_rect_slots, inv, nb);
            HIP_TRY_C(hipGetLastError());
#undef HIP_TRY_C
        }
        d.nodes = reinterpret_cast<const float4 *>(res.nodes);
        d.root_ref = res.root_ref;
        ctx->stack_depth = res.stack_depth;
        ctx->n_bvh_nodes = res.n_nodes;
        ctx->built_on_device = 1;
        ctx->bvh_build_ms = res.build_ms;
        if ((st = cleanup(RT_OK)) != RT_OK) return st;
    }
    d.n_slots = s->n_slots;
    d.always_begin = s->always_begin; d.always_count = s->always_count;
    d.n_media = s->n_media; d.n_lights = s->n_lights; d.n_moving = s->n_moving_slots; d.n_rects = s->n_rect_slots;
    d.far_limit = s->far_limit > 0.0f ? s->far_limit : r_min * 1024.0f;     // (rt_api.h; no spheres: infinity)
    {   // ... measured from the middle of the spheres of ordinary size (a ground sphere of 5 000 smallest radii is not one)
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < s->n_slots; ++i) {
            int32_t w4;
            memcpy(&w4, &s->slots[i * RT_SLOT_FLOATS + 4], 4);
            const int type = w4 & 15;
            const float *q = &s->slots[i * RT_SLOT_FLOATS];
            if ((type != RT_PRIM_SPHERE && type != RT_PRIM_MOVING_SPHERE) || !(fabsf(q[3]) <= 64.0f * r_min)) continue;
            for (int k = 0; k < 3; ++k) { if (q[k] < lo[k]) lo[k] = q[k]; if (q[k] > hi[k]) hi[k] = q[k]; }
        }
        for (int k = 0; k < 3; ++k) d.far_centre[k] = (lo[k] <= hi[k] && std::isfinite(lo[k]) && std::isfinite(hi[k])) ? 0.5f * (lo[k] + hi[k]) : 0.0f;
        double reach = 0.0;      // ... and how far the centre of one of them lies from there at most (a MovingSphere: either end of its way)
        for (uint32_t i = 0; i < s->n_slots; ++i) {
            int32_t w4;
            memcpy(&w4, &s->slots[i * RT_SLOT_FLOATS + 4], 4);
            const int type = w4 & 15;
            if ((type != RT_PRIM_SPHERE && type != RT_PRIM_MOVING_SPHERE) || !(fabsf(s->slots[i * RT_SLOT_FLOATS + 3]) <= 64.0f * r_min)) continue;
            for (int e = 0; e < (type == RT_PRIM_MOVING_SPHERE ? 2 : 1); ++e) {
                const float *q = &s->slots[i * RT_SLOT_FLOATS + (e ? 8 : 0)];
                double d2 = 0.0;
                for (int k = 0; k < 3; ++k) d2 += ((double)q[k] - d.far_centre[k]) * ((double)q[k] - d.far_centre[k]);
                if (std::isfinite(d2) && d2 > reach) reach = d2;
            }
        }
        d.far_reach = (float)(std::sqrt(reach) * 1.0001);
        d.far_rmin = std::isfinite(r_min) ? r_min : 0.0f;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_scene = true;
    return RT_OK;
}

int rt_scene_query(rt_ctx *ctx, rt_scene_info *info) {
    if (!ctx) return RT_E_INVALID;
    if (!info || info->struct_size != sizeof(rt_scene_info)) return fail(ctx, RT_E_INVALID, "rt_scene_query: bad struct size");
    if (!ctx->have_scene) return fail(ctx, RT_E_INVALID, "rt_scene_query: upload a scene first");
    info->n_slots = ctx->sc.n_slots; info->n_bvh_nodes = ctx->n_bvh_nodes; info->bvh_stack_depth = ctx->stack_depth;
    info->bvh_root_ref = ctx->sc.root_ref; info->built_on_device = ctx->built_on_device; info->bvh_build_ms = ctx->bvh_build_ms;
    return RT_OK;
}

int rt_scene_read_bvh(rt_ctx *ctx, float *nodes, float *slots) {
    if (!ctx) return RT_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RT_E_INVALID, "rt_scene_read_bvh: upload a scene first");
    int st = set_device(ctx);
    if (st) return st;
    if (nodes && ctx->n_bvh_nodes) HIP_TRY(ctx, hipMemcpyAsync(nodes, ctx->sc.nodes, (size_t)ctx->n_bvh_nodes * RT_NODE_FLOATS * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (slots && ctx->sc.n_slots) HIP_TRY(ctx, hipMemcpyAsync(slots, ctx->sc.slots, (size_t)ctx->sc.n_slots * RT_SLOT_FLOATS * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_sampler_upload(rt_ctx *ctx, const rt_sampler_desc *s) {
    if (!ctx) return RT_E_INVALID;
    if (!s || !s->samples || !s->disk_samples || !s->shuffled_indices || s->sets == 0 || s->samples_in_set == 0)
        return fail(ctx, RT_E_INVALID, "rt_sampler_upload: bad descriptor");
    size_t n = (size_t)s->sets * s->samples_in_set;
    for (size_t i = 0; i < n; ++i)
        if (s->shuffled_indices[i] >= s->samples_in_set) return fail(ctx, RT_E_INVALID, "shuffled index %zu out of range", i);
    int st = set_device(ctx);
    if (st) return st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((st = settle_all(ctx)) != RT_OK) return st;
    free_all(ctx->sampler_allocs);
    ctx->have_sampler = false;
    auto &own = ctx->sampler_allocs;
    if ((st = upload(ctx, own, s->samples, n * 8, &ctx->sc.samples)) != RT_OK) return st;
    if ((st = upload(ctx, own, s->disk_samples, n * 8, &ctx->sc.disk)) != RT_OK) return st;
    if ((st = upload(ctx, own, s->shuffled_indices, n * 4, &ctx->sc.shuffled)) != RT_OK) return st;
    ctx->sc.sets = s->sets;
    ctx->sc.spp = s->samples_in_set;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_sampler = true;
    return RT_OK;
}

int rt_render(rt_ctx *ctx, const rt_render_params *p, float *out_rgba_sum, rt_stats *stats) {
    if (!ctx) return RT_E_INVALID;
    if (!p || p->struct_size != sizeof(rt_render_params) || !out_rgba_sum) return fail(ctx, RT_E_INVALID, "rt_render: bad arguments");
    if (!ctx->have_scene || !ctx->have_sampler) return fail(ctx, RT_E_INVALID, "rt_render: upload a scene and a sampler first");
    if (p->image_width < 2 || p->image_height < 2) return fail(ctx, RT_E_INVALID, "image must be at least 2x2 (u divides by W-1, camera.rs:197)");
    if ((int64_t)p->image_width * p->image_height > (1ll << 30)) return fail(ctx, RT_E_INVALID, "image too large");
    if (p->samples_per_pixel != (int32_t)ctx->sc.spp) return fail(ctx, RT_E_INVALID, "samples_per_pixel %d does not match the sampler table (%u)", p->samples_per_pixel, ctx->sc.spp);
    if (p->spp_begin < 0 || p->spp_count < 0 || p->spp_begin + p->spp_count > p->samples_per_pixel) return fail(ctx, RT_E_INVALID, "sample range outside [0, spp)");
    if (p->collect_stats && !stats) return fail(ctx, RT_E_INVALID, "collect_stats without a stats struct");
    if (p->max_ray_depth > RT_MAX_RAY_DEPTH) return fail(ctx, RT_E_INVALID, "max_ray_depth %d: at most %d (a path's remaining depth travels in 16 bits)", p->max_ray_depth, RT_MAX_RAY_DEPTH);
    if (p->projection < RT_PROJ_PERSPECTIVE || p->projection > RT_PROJ_SPHERICAL_PANORAMIC) return fail(ctx, RT_E_INVALID, "unknown projection %d", p->projection);
    int x0 = p->tile_x0 < 0 ? 0 : p->tile_x0, y0 = p->tile_y0 < 0 ? 0 : p->tile_y0;
    int x1 = p->tile_x1 > p->image_width ? p->image_width : p->tile_x1, y1 = p->tile_y1 > p->image_height ? p->image_height : p->tile_y1;
    if (p->cancel && *p->cancel) return fail(ctx, RT_E_CANCELLED, "cancelled");
    if (x1 <= x0 || y1 <= y0 || p->spp_count == 0) return RT_OK;
    int st = set_device(ctx);
    if (st) return st;

    RenderArgs a{};
    a.width = p->image_width; a.height = p->image_height;
    a.spp = p->samples_per_pixel; a.spp_begin = p->spp_begin; a.spp_count = p->spp_count;
    a.max_depth = p->max_ray_depth;
    memcpy(a.bg, p->background, sizeof a.bg);
    a.seed = p->seed;
    a.x0 = x0; a.y0 = y0; a.x1 = x1; a.y1 = y1;
    a.cam = p->camera;
    a.projection = p->projection; a.psi_max = p->psi_max; a.lambda_max = p->lambda_max;

    const bool host_out = p->out_location == RT_MEM_HOST;
    hipStream_t stream = host_out ? ctx->stream : (p->stream ? (hipStream_t)p->stream : nullptr);
    size_t n_px = (size_t)p->image_width * p->image_height;
    float4 *d_out = reinterpret_cast<float4 *>(out_rgba_sum);
    uint32_t *d_rays = p->collect_stats ? p->pixel_rays : nullptr;
    void *tmp_out = nullptr, *tmp_rays = nullptr;
    if (host_out) {
        HIP_TRY(ctx, hipMalloc(&tmp_out, n_px * 16));
        hipError_t e = hipMemcpyAsync(tmp_out, out_rgba_sum, n_px * 16, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess && d_rays) {
            e = hipMalloc(&tmp_rays, n_px * 4);
            if (e == hipSuccess) e = hipMemcpyAsync(tmp_rays, p->pixel_rays, n_px * 4, hipMemcpyHostToDevice, stream);
        }
        if (e != hipSuccess) {
            (void)hipFree(tmp_out);
            if (tmp_rays) (void)hipFree(tmp_rays);
            return fail(ctx, RT_E_HIP, "staging the image failed: %s", hipGetErrorString(e));
        }
        d_out = static_cast<float4 *>(tmp_out);
        if (d_rays) d_rays = static_cast<uint32_t *>(tmp_rays);
    }
    {
        int rc = p->collect_stats ? render_wavefront<true>(ctx, a, p, d_out, d_rays, stream) : render_wavefront<false>(ctx, a, p, d_out, d_rays, stream);
        if (rc != RT_OK) {
            (void)hipDeviceSynchronize();     // the pipelines' streams may hold work of the call that failed: nothing of it outlives the call
            if (tmp_out) (void)hipFree(tmp_out);
            if (tmp_rays) (void)hipFree(tmp_rays);
            return rc;
        }
    }
    hipError_t e = hipGetLastError();
    unsigned long long h_stats[RT_STATS_TOTAL_WORDS] = {};
    if (e == hipSuccess && p->collect_stats) {
        e = hipMemcpyAsync(h_stats, ctx->d_stats, sizeof h_stats, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    if (e == hipSuccess && host_out) {
        e = hipMemcpyAsync(out_rgba_sum, tmp_out, n_px * 16, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess && tmp_rays) e = hipMemcpyAsync(p->pixel_rays, tmp_rays, n_px * 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
    }
    if (tmp_out) (void)hipFree(tmp_out);
    if (tmp_rays) (void)hipFree(tmp_rays);
    if (e != hipSuccess) return fail(ctx, RT_E_HIP, "rt_render: %s", hipGetErrorString(e));
    if (p->collect_stats) {
        auto unpack = [](const unsigned long long *h, rt_stats *st) {
            memset(st, 0, sizeof *st);
            st->paths = h[0]; st->rays = h[1]; st->node_visits = h[2];
            constexpr int P = RT_N_PRIM_TYPES;   // order: flush_counters (rt_wavefront.h)
            for (int i = 0; i < P; ++i) st->prim_tests[i] = h[3 + i];
            st->medium_tests = h[3 + P]; st->light_pdf_evals = h[4 + P];
            for (int i = 0; i < 5; ++i) st->shades[i] = h[5 + P + i];
            for (int i = 0; i < 4; ++i) st->tex_evals[i] = h[10 + P + i];
            st->accepted_hits = h[14 + P];
            for (int i = 0; i < 5; ++i) { st->phase_lanes[i] = h[15 + P + i]; st->phase_slots[i] = h[20 + P + i]; }
            st->box_tests = h[25 + P];
        };
        unpack(h_stats, stats);
        for (int k = 0; k < RT_K_COUNT; ++k) unpack(h_stats + RT_KSTAT_BASE + 64 * k, &ctx->kernel_stats[k]);
        ctx->have_kernel_stats = true;
    }
    return RT_OK;
}

int rt_last_kernel_stats(rt_ctx *ctx, int32_t kernel, rt_stats *out) {
    if (!ctx) return RT_E_INVALID;
    if (!out || kernel < 0 || kernel >= RT_K_COUNT) return fail(ctx, RT_E_INVALID, "rt_last_kernel_stats: kernel %d (0 generate, 1 traverse, 2 shade, 3 tail)", kernel);
    if (!ctx->have_kernel_stats) return fail(ctx, RT_E_INVALID, "rt_last_kernel_stats: no render with collect_stats yet");
    *out = ctx->kernel_stats[kernel];
    return RT_OK;
}

int rt_resolve(rt_ctx *ctx, const float *rgba_sum, int32_t n_pixels, int32_t spp, float *out_rgba, int32_t location, void *stream_) {
    if (!ctx) return RT_E_INVALID;
    if (!rgba_sum || !out_rgba || n_pixels < 0 || spp <= 0) return fail(ctx, RT_E_INVALID, "rt_resolve: bad arguments");
    if (n_pixels == 0) return RT_OK;
    int st = set_device(ctx);
    if (st) return st;
    float inv_spp = 1.0f / (float)spp;  // (spp as f32).recip(), main.rs:1363
    dim3 grid((unsigned)((n_pixels + 255) / 256)), block(256);
    if (location == RT_MEM_DEVICE) {
        hipStream_t stream = stream_ ? (hipStream_t)stream_ : nullptr;
        hipLaunchKernelGGL(resolve_kernel, grid, block, 0, stream, (const float4 *)rgba_sum, n_pixels, inv_spp, (float4 *)out_rgba);
        HIP_TRY(ctx, hipGetLastError());
        return RT_OK;
    }
    void *d_in = nullptr, *d_o = nullptr;
    size_t bytes = (size_t)n_pixels * 16;
    HIP_TRY(ctx, hipMalloc(&d_in, bytes));
    hipError_t e = hipMalloc(&d_o, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, rgba_sum, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(resolve_kernel, grid, block, 0, ctx->stream, (const float4 *)d_in, n_pixels, inv_spp, (float4 *)d_o);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_rgba, d_o, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    if (d_o) (void)hipFree(d_o);
    if (e != hipSuccess) return fail(ctx, RT_E_HIP, "rt_resolve: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_resolve_rgb8(rt_ctx *ctx, const float *rgba_sum, int32_t n_pixels, int32_t spp, uint8_t *out_rgb, int32_t location, void *stream_) {
    if (!ctx) return RT_E_INVALID;
    if (!rgba_sum || !out_rgb || n_pixels < 0 || spp <= 0) return fail(ctx, RT_E_INVALID, "rt_resolve_rgb8: bad arguments");
    if (n_pixels == 0) return RT_OK;
    int st = set_device(ctx);
    if (st) return st;
    const float inv_spp = 1.0f / (float)spp;
    dim3 grid((unsigned)((n_pixels + 255) / 256)), block(256);
    if (location == RT_MEM_DEVICE) {
        hipLaunchKernelGGL(resolve_rgb8_kernel, grid, block, 0, stream_ ? (hipStream_t)stream_ : nullptr, (const float4 *)rgba_sum, n_pixels, inv_spp, out_rgb);
        HIP_TRY(ctx, hipGetLastError());
        return RT_OK;
    }
    void *d_in = nullptr, *d_o = nullptr;
    const size_t in_bytes = (size_t)n_pixels * 16, out_bytes = (size_t)n_pixels * 3;
    HIP_TRY(ctx, hipMalloc(&d_in, in_bytes));
    hipError_t e = hipMalloc(&d_o, out_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, rgba_sum, in_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(resolve_rgb8_kernel, grid, block, 0, ctx->stream, (const float4 *)d_in, n_pixels, inv_spp, (uint8_t *)d_o);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_rgb, d_o, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    if (d_o) (void)hipFree(d_o);
    if (e != hipSuccess) return fail(ctx, RT_E_HIP, "rt_resolve_rgb8: %s", hipGetErrorString(e));
    return RT_OK;
}

// Layout of the device buffer a first-hit query works in: rays 28N | prim 4N | t 4N | normal 12N | uv 8N | ff N | stack overflow
struct FirstHitLayout {
    size_t N, off_prim, off_t, off_n, off_uv, off_ff, off_ovf, ovf_bytes, total, n_threads;
    uint32_t lds_depth;
};
static FirstHitLayout first_hit_layout(const rt_ctx *ctx, int32_t n) {
    FirstHitLayout L{};
    L.N = (size_t)n;
    // traversal stack: up to 48 entries per lane in LDS (48 KB per block), a deeper tree's rest in HBM
    L.lds_depth = ctx->stack_depth < 48u ? ctx->stack_depth : 48u;
    L.n_threads = (L.N + 255) / 256 * 256;
    L.ovf_bytes = (size_t)(ctx->stack_depth - L.lds_depth) * L.n_threads * sizeof(StackEntry);
    L.off_prim = 28 * L.N; L.off_t = 32 * L.N; L.off_n = 36 * L.N; L.off_uv = 48 * L.N; L.off_ff = 56 * L.N;
    L.off_ovf = (57 * L.N + 255) / 256 * 256;
    L.total = L.off_ovf + L.ovf_bytes;
    return L;
}
// Stage the rays into d, run first_hit_kernel, copy the results back, wait.  d holds L.total bytes; nothing is allocated here.
static hipError_t first_hit_run(rt_ctx *ctx, char *d, const FirstHitLayout &L, const float *rays_soa, int32_t n, float t_min, float t_max,
                                uint32_t *prim_id, float *t, float *normal3, uint8_t *front_face, float *uv2) {
    const size_t N = L.N;
    hipStream_t stream = ctx->stream;
    hipError_t e = hipMemcpyAsync(d, rays_soa, 28 * N, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        dim3 grid((unsigned)((N + 255) / 256)), block(256);
        size_t lds = (size_t)L.lds_depth * 256 * sizeof(StackEntry);
        hipLaunchKernelGGL(first_hit_kernel, grid, block, lds, stream, ctx->sc, (const float *)d, n, t_min, t_max,
                           (uint32_t *)(d + L.off_prim), (float *)(d + L.off_t), (float *)(d + L.off_n), (uint8_t *)(d + L.off_ff),
                           (float *)(d + L.off_uv), L.lds_depth, L.ovf_bytes ? (StackEntry *)(d + L.off_ovf) : nullptr, (uint32_t)L.n_threads);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(prim_id, d + L.off_prim, 4 * N, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t, d + L.off_t, 4 * N, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(normal3, d + L.off_n, 12 * N, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(front_face, d + L.off_ff, N, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && uv2) e = hipMemcpyAsync(uv2, d + L.off_uv, 8 * N, hipMemcpyDeviceToHost, stream);
    hipError_t e2 = hipStreamSynchronize(stream);   // also after a failed copy: nothing of this call may still be in flight
    return e != hipSuccess ? e : e2;
}

int rt_trace_first_hit(rt_ctx *ctx, const float *rays_soa, int32_t n, float t_min, float t_max, uint32_t *prim_id, float *t,
                       float *normal3, uint8_t *front_face, float *uv2) {
    if (!ctx) return RT_E_INVALID;
    if (!rays_soa || n < 0 || !prim_id || !t || !normal3 || !front_face) return fail(ctx, RT_E_INVALID, "rt_trace_first_hit: bad arguments");
    if (!ctx->have_scene) return fail(ctx, RT_E_INVALID, "rt_trace_first_hit: upload a scene first");
    if (n == 0) return RT_OK;
    int st = set_device(ctx);
    if (st) return st;
    const FirstHitLayout L = first_hit_layout(ctx, n);
    char *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d, L.total));
    hipError_t e = first_hit_run(ctx, d, L, rays_soa, n, t_min, t_max, prim_id, t, normal3, front_face, uv2);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, RT_E_HIP, "rt_trace_first_hit: %s", hipGetErrorString(e));
    return RT_OK;
}

// rt_trace_first_hit without an allocation: the same kernel on the same layout, in this device's copy of g_noalloc_scratch.
// Nothing in the scratch has to start at zero (the kernel writes every result of rays 0..n-1 and pushes before it pops), so
// the previous call's contents are simply overwritten; calls that overlap on one GPU would race on it.
int rt_trace_first_hit_noalloc(rt_ctx *ctx, const float *rays_soa, int32_t n, float t_min, float t_max, uint32_t *prim_id, float *t,
                               float *normal3, uint8_t *front_face, float *uv2) {
    if (!ctx) return RT_E_INVALID;
    if (!rays_soa || n < 0 || !prim_id || !t || !normal3 || !front_face) return fail(ctx, RT_E_INVALID, "rt_trace_first_hit_noalloc: bad arguments");
    if (!ctx->have_scene) return fail(ctx, RT_E_INVALID, "rt_trace_first_hit_noalloc: upload a scene first");
    if (!ctx->noalloc_scratch) return fail(ctx, RT_E_HIP, "rt_trace_first_hit_noalloc: no scratch address for device %d", ctx->device);
    if (n == 0) return RT_OK;
    const FirstHitLayout L = first_hit_layout(ctx, n);
    if (L.total > (size_t)RT_NOALLOC_SCRATCH_BYTES) {
        const size_t per_ray = 57 + (size_t)(ctx->stack_depth - L.lds_depth) * sizeof(StackEntry);
        return fail(ctx, RT_E_INVALID, "rt_trace_first_hit_noalloc: %d rays need %zu bytes of scratch, the static array holds %u "
                    "(about %zu rays with this scene's BVH); split the batch or use rt_trace_first_hit",
                    n, L.total, (unsigned)RT_NOALLOC_SCRATCH_BYTES, ((size_t)RT_NOALLOC_SCRATCH_BYTES - 512) / per_ray / 256 * 256);
    }
    int st = set_device(ctx);
    if (st) return st;
    hipError_t e;
    {   // every context of this process on this GPU shares the one array: one call at a time (the call blocks to its end anyway)
        std::lock_guard<std::mutex> hold(g_noalloc_mutex[ctx->device < kMaxDevices ? ctx->device : 0]);
        e = first_hit_run(ctx, ctx->noalloc_scratch, L, rays_soa, n, t_min, t_max, prim_id, t, normal3, front_face, uv2);
    }
    if (e != hipSuccess) return fail(ctx, RT_E_HIP, "rt_trace_first_hit_noalloc: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_synchronize(rt_ctx *ctx, void *stream) {
    if (!ctx) return RT_E_INVALID;
    int st = set_device(ctx);
    if (st) return st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream ? (hipStream_t)stream : nullptr));
    for (WfSpace &S : ctx->wf)      // frames that have ended by now: look at how they ended
        if (S.in_flight && hipEventQuery(S.done) == hipSuccess && (st = settle(ctx, S)) != RT_OK) return st;
    return RT_OK;
}

int rt_device_info(rt_ctx *ctx, char *name, int32_t name_len, int32_t *cu_count, int32_t *clock_mhz) {
    if (!ctx) return RT_E_INVALID;
    if (name && name_len > 0) {
        // some boxes report an empty marketing name: fall back to the ISA name (gfx950:sramecc+:xnack-)
        strncpy(name, ctx->prop.name[0] ? ctx->prop.name : ctx->prop.gcnArchName, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = ctx->prop.clockRate / 1000;
    return RT_OK;
}

}  // extern "C"
