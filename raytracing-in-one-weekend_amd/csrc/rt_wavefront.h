// rt_wavefront.h — the streaming ("wavefront") form of the render hot path.
//
// The reference renders a pixel sample as one recursive call chain (camera.rs:256-309) on one
// worker thread.  A 64-lane wave that does the same per lane spends most of its issue slots on
// lanes that wait: paths end at different bounces, rays visit different numbers of BVH nodes,
// and shading and traversal are different code.  Here the three kinds of work are separate
// kernels over a pool of in-flight paths that lives in HBM:
//
//   generate  : pixel samples ("items") -> new paths: RNG stream, camera ray (camera.rs:82-96,196)
//   traverse  : BVH walk of every path's current ray; persistent waves that refill idle lanes
//               from the pool as soon as enough of them have finished their ray
//   shade     : ray_color's body for the hit found (camera.rs:264-308); survivors are compacted
//               into the other pool buffer, finished paths park their radiance in the sample buffer
//   tail      : takes the place of traverse + shade once every item is out and few paths are left: several bounces
//               of every remaining path in one launch, a lane per path, survivors compacted like shade's
//   fold      : pixel_color += ray_color, in sample order (camera.rs:195-250), from the sample buffer
//
// Every path consumes its own RNG stream in the reference's call order, whatever lane, wave or
// iteration its pieces run in, so the image is the same bits as the per-lane megakernel's (and the
// fold is the reference's order exactly, independent of pool size and launch geometry).
//
// Pool entry = 72 B in five SoA arrays: origin+time | direction+flags | throughput+item | rng state |
// hit (t, id, slot, floor_id).  Two pool buffers (ping-pong).  A path in flight carries no radiance:
// every reference material either scatters with emitted() = black (material.rs:24-26) or emits and
// ends the path (diffuse_light.rs:36-41), so radiance is only ever added by the bounce that ends it.
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H

#include "rt_kernels.h"

namespace rtk {

struct PathQ {
    float4 *ray_o;   // o.x o.y o.z time
    float4 *ray_d;   // d.x d.y d.z flags (depth | WF_ALIVE | WF_WALK)
    float4 *thr;     // throughput.xyz, item index
    float2 *rng;     // PCG state lo, hi (the stream constant follows from pixel and sample)
    float4 *hit;     // best.t, best.id, best.slot, floor_id
};

// remaining depth in 16 bits (rt_render refuses max_ray_depth > RT_MAX_RAY_DEPTH = 65535), then the two state bits
enum : uint32_t { WF_DEPTH_MASK = 0xffffu, WF_ALIVE = 0x10000u, WF_WALK = 0x20000u };

// The pool is cut into RT_WF_SHARDS shards of `cap` entries, each with its own counters in its own
// 256-byte line: an atomic on one address costs several ns whoever issues it, so the per-wave
// appends and fetches of a launch (tens of thousands) must not all meet on one word.  A shard's live
// entries are its first count_in entries; shade compacts shard s of the current buffer into shard s
// of the next one.
#define RT_WF_SHARDS 64u
#define RT_WF_RING 16u
struct WfShard {
    uint32_t count_in;   // live entries of this shard in the current pool buffer
    uint32_t n_surv;     // entries shade has appended to this shard of the next buffer
    uint32_t cursor;     // traverse's fetch cursor in this shard
    uint32_t item_head;  // (shard 0 only) first item not handed out when this iteration's generate runs
    uint32_t tail_mode;  // != 0: every item is out and at most WfArgs::tail_max paths live, this iteration may be run by wf_tail_kernel
                         // (n = the n-th iteration in a row for which that holds)
    uint32_t pad[59];
};
static_assert(RT_WF_SHARDS == 64u, "generate sums the shard counters with one wave");
struct WfCounters {      // one per iteration, in a ring; written by the previous iteration's generate
    WfShard sh[RT_WF_SHARDS];
};

struct WfArgs {
    RenderArgs a;
    uint32_t pool;        // entries per pool buffer = RT_WF_SHARDS * cap
    uint32_t cap;         // entries per shard (multiple of 256)
    uint32_t pipe, n_pipes;   // this pipeline renders the tiles t with t % n_pipes == pipe (interleaved: the pipelines then carry
                          // the same mix of cheap and expensive pixels and drain together; top / bottom halves did not)
    uint32_t n_items;     // items of this pipeline in this batch (its tiles * spp_batch * 64)
    uint32_t spp_batch;   // samples per pixel in this batch
    uint32_t sample0;     // absolute index of the batch's first sample
    uint32_t tiles_x, n_tiles;   // 8x8 pixel tiles over the window
    uint32_t spp_m, spp_k, tx_m, tx_k;   // n / spp_batch and n / tiles_x as (n * m) >> k, exact for n < 2^26
    uint32_t tiles_pipe, tile_stride;    // this pipeline's l-th tile in hand-out order is its tile (l * tile_stride) % tiles_pipe (stride 1: in order)
    uint32_t tp_m, tp_k;                 // n / tiles_pipe as (n * m) >> k; tp_k == 0: hand out tile by tile (all samples of a tile in a row)
    uint32_t sort;                       // shade deals a block's entries to its threads in material order
    uint32_t tail_max;                   // live paths at or below which (with every item out) an iteration is in tail mode; 0 = never
};

// floor(n / d) for n < 2^26 with (m, k) from the host's magic_div(): k = 26 + ceil(log2 d), m = floor(2^k / d) + 1.
RT_DEV uint32_t div_magic(uint32_t n, uint32_t m, uint32_t k) { return (uint32_t)(((uint64_t)n * (uint64_t)m) >> k); }

// item -> (pixel, sample): 64 consecutive items are one 8x8 tile at one sample; a tile's samples are
// consecutive groups.  The sample buffer uses the same index, so the fold reads it coalesced.
RT_DEV bool item_pixel(const WfArgs &w, uint32_t g, int *x, int *y, uint32_t *sample) {
    const uint32_t group = g >> 6, ln = g & 63u;
    const uint32_t t = div_magic(group, w.spp_m, w.spp_k), k = group - t * w.spp_batch;
    const uint32_t ty = div_magic(t, w.tx_m, w.tx_k), tx = t - ty * w.tiles_x;
    *x = w.a.x0 + (int)(tx * 8u + (ln & 7u));
    *y = w.a.y0 + (int)(ty * 8u + (ln >> 3));
    *sample = w.sample0 + k;
    return *x < w.a.x1 && *y < w.a.y1;
}

struct WfPath {
    Path ps;
    rts::Rng rng;
    uint32_t pixel, sample, item;
};

RT_DEV void store_entry(const PathQ &q, uint32_t i, const WfPath &p, const Trav &tv, bool walk) {
    const uint32_t flags = ((uint32_t)p.ps.depth & WF_DEPTH_MASK) | WF_ALIVE | (walk ? WF_WALK : 0u);
    q.ray_o[i] = make_float4(p.ps.ray.o.x, p.ps.ray.o.y, p.ps.ray.o.z, p.ps.ray.time);
    q.ray_d[i] = make_float4(p.ps.ray.d.x, p.ps.ray.d.y, p.ps.ray.d.z, __uint_as_float(flags));
    q.thr[i] = make_float4(p.ps.throughput.x, p.ps.throughput.y, p.ps.throughput.z, __uint_as_float(p.item));
    q.rng[i] = make_float2(__uint_as_float((uint32_t)p.rng.state), __uint_as_float((uint32_t)(p.rng.state >> 32)));
    q.hit[i] = make_float4(tv.best.t, __int_as_float(tv.best.id), __int_as_float(tv.best.slot), __int_as_float(tv.floor_id));
}

// The words of entry i that settle_trace() may change after store_entry() (same lane, same addresses: in program order).
RT_DEV void restore_trace(const PathQ &q, uint32_t i, int depth, const Trav &tv, bool walk) {
    const uint32_t flags = ((uint32_t)depth & WF_DEPTH_MASK) | WF_ALIVE | (walk ? WF_WALK : 0u);
    reinterpret_cast<float *>(&q.ray_d[i])[3] = __uint_as_float(flags);
    q.hit[i] = make_float4(tv.best.t, __int_as_float(tv.best.id), __int_as_float(tv.best.slot), __int_as_float(tv.floor_id));
}

// Append the lanes with `keep` to q at positions taken from *counter (one atomic per wave).
RT_DEV uint32_t wave_append(uint32_t *counter, bool keep) {
    const unsigned long long m = __ballot(keep);
    if (m == 0ull) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// wave_append() with a shard of its own per lane: the lanes with `keep` are appended to the shards `target` names (one atomic per
// distinct shard among them).  A shard that is full passes what does not fit on to the next one (and takes the overshoot off its
// counter again, so a counter ends at the number of entries placed: once a counter has passed `cap` every position below `cap` is
// taken, so nobody is handed one of those twice).  Ends because the pool holds every live path.  Returns the pool index.
RT_DEV uint32_t wave_append_to(WfCounters *c, uint32_t cap, bool keep, uint32_t target) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t pos = 0;
    unsigned long long todo = __ballot(keep);
    while (todo != 0ull) {      // wave-uniform
        const int leader = __builtin_ctzll(todo);
        const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)target, leader);
        const bool mine = keep && ((todo >> lane) & 1ull) != 0ull && target == s;
        const unsigned long long m = __ballot(mine);
        const uint32_t cnt = (uint32_t)__popcll(m);
        uint32_t base = 0;
        if ((int)lane == leader) {
            base = atomicAdd(&c->sh[s].n_surv, cnt);
            if (base + cnt > cap) atomicSub(&c->sh[s].n_surv, base >= cap ? cnt : base + cnt - cap);
        }
        base = (uint32_t)__shfl((int)base, leader, 64);
        const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const bool placed = mine && at < cap;
        if (placed) pos = s * cap + at;
        if (mine && !placed) target = (s + 1u) % RT_WF_SHARDS;
        todo &= ~__ballot(placed);
    }
    return pos;
}

// `which` = the kernel the counters come from (RT_K_*): the counting variant keeps them per kernel as well (words
// RT_KSTAT_BASE + 64 * which ..., rt_last_kernel_stats), so that the work of a frame can be priced kernel by kernel.
enum { RT_K_GENERATE = 0, RT_K_TRAVERSE = 1, RT_K_SHADE = 2, RT_K_TAIL = 3, RT_K_COUNT = 4 };
#define RT_KSTAT_BASE 128
#define RT_STATS_TOTAL_WORDS (RT_KSTAT_BASE + 64 * RT_K_COUNT)
template <bool STATS>
RT_DEV void flush_counters(const Counters &cnt, uint32_t paths, unsigned long long *stats, int which) {
    if (kTimers && !STATS) {
        for (int i = 0; i < RT_N_REGIONS; ++i) {
            uint32_t w = wave_sum(cnt.region[i]);
            if ((threadIdx.x & 63u) == 0 && w) atomicAdd(&stats[RT_STAT_WORDS + 1 + i], (unsigned long long)w);
        }
    }
    if (!STATS) return;
    uint32_t vals[RT_STAT_WORDS];   // the order rt_render copies into rt_stats
    constexpr int P = RT_N_PRIM_TYPES;
    vals[0] = paths;
    vals[1] = cnt.rays; vals[2] = cnt.node_visits;
    for (int i = 0; i < P; ++i) vals[3 + i] = cnt.prim_tests[i];
    vals[3 + P] = cnt.medium_tests; vals[4 + P] = cnt.light_pdf_evals;
    for (int i = 0; i < 5; ++i) vals[5 + P + i] = cnt.shades[i];
    for (int i = 0; i < 4; ++i) vals[10 + P + i] = cnt.tex_evals[i];
    vals[14 + P] = cnt.accepted;
    for (int i = 0; i < 5; ++i) { vals[15 + P + i] = cnt.phase_lanes[i]; vals[20 + P + i] = cnt.phase_slots[i]; }
    vals[25 + P] = cnt.box_tests;
    for (int i = 0; i < RT_STAT_WORDS; ++i) {
        uint32_t w = wave_sum(vals[i]);
        if ((threadIdx.x & 63u) == 0 && w) { atomicAdd(&stats[i], (unsigned long long)w); atomicAdd(&stats[RT_KSTAT_BASE + 64 * which + i], (unsigned long long)w); }
    }
}

/* ------------------------------------------------------------------ generate */

// Fills every shard of the next pool buffer behind shade's survivors with new paths and publishes the
// next iteration's counters.  Thread j looks at pool entry j: items are numbered through the free
// entries in pool order, so nothing needs an atomic.
#ifndef RT_WF_SHADE_WAVES
#define RT_WF_SHADE_WAVES 1     // (1: shade_min_waves' choice)
#endif
#ifndef RT_WF_GEN_WAVES
#define RT_WF_GEN_WAVES 1       // min waves per SIMD the generate kernel is compiled for (1 = the compiler's choice: 112 VGPRs, 4 waves)
#endif
// (the register budgets the kernels had before the rare folds of settle_trace() were given to the whole wave: what those loops
// need beyond it is spilled around them, where it costs nothing)
constexpr int gen_min_waves(bool stats, int ext) { return stats ? 1 : RT_WF_GEN_WAVES > 1 ? RT_WF_GEN_WAVES : (ext & (EXT_QUADRICS | EXT_CHAIN)) ? 4 : (ext & EXT_TRI) ? 6 : 7; }
constexpr int shade_min_waves(bool stats, int ext) { return stats ? 1 : RT_WF_SHADE_WAVES > 1 ? RT_WF_SHADE_WAVES : (ext & (EXT_MEDIA | EXT_NOISE | EXT_QUADRICS | EXT_CHAIN)) ? 5 : 7; }
template <bool STATS, int EXT>
__global__ void __launch_bounds__(256, gen_min_waves(STATS, EXT)) wf_generate_kernel(DScene sc, WfArgs w, PathQ nxt, WfCounters *ctr, uint32_t it,
                                                          float4 *__restrict__ samples, uint32_t *status, unsigned long long *stats,
                                                          uint32_t *pixel_rays) {
    const WfCounters *c = ctr + (it % RT_WF_RING);
    // grid = (chunks of 256 entries per shard, 64 shards).  The host launches cap / 256 chunks while items are being handed
    // out and, once the pool only shrinks, as many as the largest shard last reported needs (status[3]): an iteration on a
    // nearly empty pool then dispatches 64 workgroups instead of 16 384 that look at a counter and leave.
    const uint32_t shard = blockIdx.y, off0 = blockIdx.x * blockDim.x;   // block-uniform
    const uint32_t j = shard * w.cap + off0 + threadIdx.x;
    const uint32_t n_surv = c->sh[shard].n_surv;
    if (off0 != 0 && off0 + blockDim.x <= n_surv) return;   // block-uniform: only survivors here (most blocks in steady state)
    __shared__ uint32_t s_sum[3];
    if (threadIdx.x < 64u) {                          // one wave sums the shard counters (64 lines, one load each)
        const uint32_t sv = c->sh[threadIdx.x].n_surv;
        const uint32_t before = wave_sum(threadIdx.x < shard ? w.cap - sv : 0u);
        const uint32_t all = wave_sum(w.cap - sv), surv = wave_sum(sv);
        if (threadIdx.x == 0) { s_sum[0] = before; s_sum[1] = all; s_sum[2] = surv; }
        if (off0 == 0 && shard == 0) {     // largest shard of the next iteration for the host's grid sizing, once all items are out
            uint32_t mx = sv;
            for (int o = 32; o > 0; o >>= 1) { const uint32_t other = (uint32_t)__shfl_down((int)mx, o, 64); mx = other > mx ? other : mx; }
            if (threadIdx.x == 0) status[3] = c->sh[0].item_head == w.n_items ? mx : 0xffffffffu;   // (nothing new in this iteration: next count_in = n_surv)
        }
    }
    __syncthreads();
    const uint32_t item_head = c->sh[0].item_head;
    const uint32_t room_before = s_sum[0];            // free entries in the shards before this one
    const uint32_t left = w.n_items - item_head;
    const uint32_t left_here = left > room_before ? left - room_before : 0u;
    const uint32_t room = w.cap - n_surv;
    const uint32_t n_new = room < left_here ? room : left_here;
    if (off0 == 0 && threadIdx.x == 0) {              // first block of the shard: next iteration's counters
        WfShard *n = &ctr[(it + 1u) % RT_WF_RING].sh[shard];
        n->count_in = n_surv + n_new; n->n_surv = 0; n->cursor = 0;
        const uint32_t room_all = s_sum[1], surv_all = s_sum[2];
        const uint32_t issued = room_all < left ? room_all : left;
        // (every shard's first block writes the same answer into its own line, so a kernel reads it with the counters it reads anyway)
        // (the value counts the tail iterations: 1 for the first one)
        n->tail_mode = (item_head + issued == w.n_items && surv_all + issued != 0u && surv_all + issued <= w.tail_max) ? c->sh[shard].tail_mode + 1u : 0u;
        if (shard == 0) {
            n->item_head = item_head + issued;
            status[0] = surv_all + issued; status[1] = item_head + issued; status[2] = it + 1u;
        }
    }
    if (off0 >= n_surv + n_new || off0 + blockDim.x <= n_surv) return;   // block-uniform: nothing to make in this block
    const uint32_t off = off0 + threadIdx.x;
    const bool mine = off >= n_surv && off < n_surv + n_new;
    Counters cnt{};
    // li = the item's place in the pipeline's hand-out order; g = its index over the whole batch (tile-major), as the sample
    // buffer and the fold use it
    const uint32_t li = item_head + room_before + (off - n_surv), lg = li >> 6;
    // The pipeline's li-th item: sample by sample (group lg = sample lk of the lt-th tile in hand-out order, all tiles at one
    // sample before the next sample), so neighbouring groups — which end up in the same shard — are different tiles.
    uint32_t lt, lk;
    if (w.tp_k) { lk = div_magic(lg, w.tp_m, w.tp_k); lt = lg - lk * w.tiles_pipe; }
    else { lt = div_magic(lg, w.spp_m, w.spp_k); lk = lg - lt * w.spp_batch; }
    // Tiles are handed out along a stride that is coprime to their number (a golden-section step through the frame), not in
    // raster order: a shard's stretch of the item sequence then holds tiles from all over the frame, so every shard — and every
    // traverse block, which works on one shard — carries the same mix of cheap (sky) and expensive (glass, crowded) pixels.
    const uint32_t pt = (lt * w.tile_stride) % w.tiles_pipe;      // (the product stays below 2^32, rt_device.hip)
    const uint32_t g = (((pt * w.n_pipes + w.pipe) * w.spp_batch + lk) << 6) | (li & 63u), e = j;
    WfPath p;
    int x, y;
    const bool inside = mine && item_pixel(w, g, &x, &y, &p.sample);
    const bool start = inside && w.a.max_depth > 0;
    p.item = g;
    Trav tv;
    tv.best.t = 0.0f; tv.best.id = -1; tv.best.slot = 0; tv.floor_id = -1; tv.ref = RT_BVH_EMPTY; tv.sp = 0;
    if (mine && !start) {     // no path: a dead entry (compacted away by the next shade)
        if (inside) samples[g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // depth <= 0 contributes 0 (camera.rs:260)
        nxt.ray_d[e] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
    }
    phase_count<STATS>(cnt, PH_START, start);
    bool have_ray = false;
    if (start) {
        p.pixel = (uint32_t)y * (uint32_t)w.a.width + (uint32_t)x;
        rts::path_rng(p.rng, w.a.seed, p.pixel, p.sample);
        RG_BEGIN(RG_CAMERA);
        have_ray = camera_ray(sc, w.a, x, y, p.pixel, p.sample, p.rng, &p.ps.ray);
        RG_END(RG_CAMERA);
        if (!have_ray) {       // FishEye outside its disk: the sample adds nothing (camera.rs:231-233)
            samples[g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            nxt.ray_d[e] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
        }
    }
    if (have_ray) {
        p.ps.throughput = v3(1.0f, 1.0f, 1.0f);
        p.ps.radiance = v3(0.0f, 0.0f, 0.0f);
        p.ps.depth = w.a.max_depth;
        if (STATS) { cnt.rays += 1; if (pixel_rays) atomicAdd(&pixel_rays[p.pixel], 1u); }
    }
    LateFolds late;
    late.literal = false; late.axis = false;
    bool walk = false;
    if (have_ray) {
        walk = begin_trace<STATS, EXT>(sc, p.ps.ray, 0.001f, INFINITY, p.rng, tv, cnt, late);
        store_entry(nxt, e, p, tv, walk);
    }
    if (settle_trace<STATS, EXT>(sc, have_ray, p.ps.ray, 0.001f, INFINITY, tv, walk, late, cnt)) restore_trace(nxt, e, w.a.max_depth, tv, walk);   // (all lanes)
    if (STATS || kTimers) flush_counters<STATS>(cnt, inside ? 1u : 0u, stats, RT_K_GENERATE);
}

/* ------------------------------------------------------------------ traverse */

#ifndef RT_WF_SORT
#define RT_WF_SORT 1           // shade: deal a block's entries to its threads in material order
#endif
#ifndef RT_WF_FETCH_MIN
#define RT_WF_FETCH_MIN 12     // refill when at least this many lanes of the wave are idle
#endif
#ifndef RT_LEAF_FRAC
#define RT_LEAF_FRAC 4u        // traverse: a leaf step also runs once a quarter of the wave's live lanes hold a leaf
#endif
#ifndef RT_WF_NODE_STEPS
#define RT_WF_NODE_STEPS 3     // node steps per turn of the traverse loop
#endif
#ifndef RT_WF_CHUNK
#define RT_WF_CHUNK 64u        // entries a wave takes from its shard's cursor at a time
#endif

// Persistent waves.  A lane holds one ray; when RT_WF_FETCH_MIN lanes have finished theirs the wave
// takes the next entries of the pool for them, so the node loop runs with most lanes busy.
// Inside, node steps and leaf steps are decoupled: a lane that reaches a leaf puts it aside (`pend`) and goes on
// with the next entry of its stack; the wave runs the primitive tests once RT_LEAF_MIN lanes hold a leaf (or nobody
// can take a node step).  A lane that reaches a second leaf before that waits.  Testing a leaf later than a
// depth-first walk would only means that best.t shrinks later (a few more node visits); the closest hit and its
// tie-break do not depend on the order of the tests.
#ifndef RT_WF_TRAV_WAVES
#define RT_WF_TRAV_WAVES 6      // min waves per SIMD the traverse kernel is compiled for (80 VGPRs, no spills)
#endif
template <bool STATS, int EXT, bool LT = false>
__global__ void __launch_bounds__(256, RT_WF_TRAV_WAVES) wf_traverse_kernel(DScene sc, PathQ q, WfCounters *ctr, uint32_t it, uint32_t cap, uint32_t lds_depth,
                                                          StackEntry *ovf, unsigned long long *stats, uint32_t with_tail, uint32_t n_top) {
    extern __shared__ StackEntry lds_stack[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const Stack stack{(LdsStackPtr)lds_stack + tid, 256u, lds_depth, ovf ? ovf + (size_t)blockIdx.x * 256u + tid : nullptr, gridDim.x * 256u};
    const uint32_t shard = blockIdx.x % RT_WF_SHARDS;
    WfShard *c = &ctr[it % RT_WF_RING].sh[shard];
    if (with_tail && c->tail_mode) return;       // wf_tail_kernel runs this iteration
    const uint32_t n = c->count_in, shard_base = shard * cap;
    TopLds top{(LdsF4Ptr)0, 0u};
    if (LT) {       // the top n_top nodes of the tree (breadth-first) behind the stacks, child refs into the image flagged (rt_kernels.h, TopLds)
        if (n == 0) return;      // block-uniform (this workgroup's shard is empty: nothing to stage the image for)
        RT_LDS f32x4 *img = (RT_LDS f32x4 *)(lds_stack + lds_depth * 256u);
        for (uint32_t k = tid; k < 4u * n_top; k += 256u) {
            const uint32_t node = k >> 2, quarter = k & 3u;
            float4 v = sc.nodes[4 * (size_t)sc.top_order[node] + quarter];
            if (quarter == 3u) {
                const uint32_t *ch = sc.top_child + 4 * node;
                if (ch[0] < n_top) v.x = __int_as_float((int32_t)(RT_TOP_FLAG | ch[0]));
                if (ch[1] < n_top) v.y = __int_as_float((int32_t)(RT_TOP_FLAG | ch[1]));
                if (ch[2] < n_top) v.z = __int_as_float((int32_t)(RT_TOP_FLAG | ch[2]));
                if (ch[3] < n_top) v.w = __int_as_float((int32_t)(RT_TOP_FLAG | ch[3]));
            }
            img[quarter * n_top + node] = f32x4{v.x, v.y, v.z, v.w};
        }
        top.img = img; top.count = n_top;
        __syncthreads();
    }
    const int32_t root_ref = LT ? (int32_t)RT_TOP_FLAG : sc.root_ref;     // (the image's node 0 is the root)
    Counters cnt{};
    uint32_t w_next = 0, w_end = 0;
    bool exhausted = n == 0;
    int32_t idx = -1;
    int32_t pend = RT_BVH_EMPTY;      // a leaf put aside (< 0), or RT_BVH_EMPTY
    Ray r;
    r.o = v3(0.0f, 0.0f, 0.0f); r.d = r.o; r.time = 0.0f;
    Trav tv;
    tv.ref = RT_BVH_EMPTY; tv.sp = 0; tv.best.t = 0.0f; tv.best.id = -1; tv.best.slot = 0; tv.floor_id = -1;
    tv.inv = r.o; tv.oi = r.o; tv.fat = 0.0f;
    const float T_MIN = 0.001f;
    RG_BEGIN(RG_TRAV_ALL);
    uint32_t dbg_turns = 0;
    for (;;) {
        if (kTimers) ++dbg_turns;
        const unsigned long long m_idle = __ballot(idx < 0);
        const uint32_t n_idle = (uint32_t)__popcll(m_idle);
        if (n_idle == 64u && exhausted) break;
        if (!exhausted && n_idle >= RT_WF_FETCH_MIN) {
            RG_BEGIN(RG_FETCH);
            if (w_next >= w_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&c->cursor, RT_WF_CHUNK);
                base = (uint32_t)__shfl((int)base, 0, 64);
                if (base >= n) { exhausted = true; continue; }
                w_next = base;
                w_end = base + RT_WF_CHUNK < n ? base + RT_WF_CHUNK : n;
            }
            const uint32_t avail = w_end - w_next;
            if (idx < 0) {
                const uint32_t rank = (uint32_t)__popcll(m_idle & ((1ull << lane) - 1ull));
                if (rank < avail) {
                    const uint32_t i = shard_base + w_next + rank;
                    const float4 d = q.ray_d[i];
                    const uint32_t flags = __float_as_uint(d.w);
                    if ((flags & (WF_ALIVE | WF_WALK)) == (WF_ALIVE | WF_WALK)) {
                        const float4 o = q.ray_o[i], h = q.hit[i];
                        idx = (int32_t)i;
                        r.o = v3(o.x, o.y, o.z); r.d = v3(d.x, d.y, d.z); r.time = o.w;
                        tv.best.t = h.x; tv.best.id = f2i(h.y); tv.best.slot = f2i(h.z); tv.floor_id = f2i(h.w);
                        tv.inv = box_inv(r.d);
                        tv.oi = r.o * tv.inv;
                        tv.fat = ray_fat(sc, r);
                        tv.sp = 0;
                        tv.ref = root_ref;
                    }
                }
            }
            w_next += n_idle < avail ? n_idle : avail;
            RG_END(RG_FETCH);
            if (__ballot(idx >= 0) == 0ull) continue;   // nothing to walk among what was fetched
        }
        // node steps (RT_WF_NODE_STEPS per turn of the loop: the ballots and the refill test above are paid once for them)
#pragma unroll
        for (int k = 0; k < RT_WF_NODE_STEPS; ++k) {
            if (idx >= 0 && tv.ref < 0 && pend == RT_BVH_EMPTY) { pend = tv.ref; tv.ref = pop_ref(stack, tv.sp); }   // leaf aside, walk on
            const bool on_node = idx >= 0 && tv.ref >= 0 && tv.ref != RT_BVH_EMPTY;
            phase_count<STATS>(cnt, PH_NODE, on_node);
            RG_BEGIN(RG_NODE);
            if (on_node) trav_node_step<STATS, LT>(sc, T_MIN, tv, stack, cnt, top);
            RG_END(RG_NODE);
        }
        if (idx >= 0 && tv.ref < 0 && pend == RT_BVH_EMPTY) { pend = tv.ref; tv.ref = pop_ref(stack, tv.sp); }
        const bool has_leaf = idx >= 0 && pend != RT_BVH_EMPTY;
        const unsigned long long m_leaf = __ballot(has_leaf);
        const unsigned long long m_node = __ballot(idx >= 0 && tv.ref >= 0 && tv.ref != RT_BVH_EMPTY);
        // ... or a RT_LEAF_FRAC-th of the lanes that still hold a ray: a wave that is draining (or thin from the start) would
        // otherwise test a leaf only when every lane is stuck on its second one, i.e. walk on with a best.t that could be shorter
        const uint32_t n_leaf = (uint32_t)__popcll(m_leaf), n_live = (uint32_t)__popcll(__ballot(idx >= 0));
        if (m_leaf != 0ull && (n_leaf >= RT_LEAF_MIN || n_leaf * RT_LEAF_FRAC >= n_live || m_node == 0ull)) {
            phase_count<STATS>(cnt, PH_LEAF, has_leaf);
            RG_BEGIN(RG_LEAF);
            if (has_leaf) { trav_leaf_test<STATS, EXT>(sc, r, T_MIN, tv, pend, cnt); pend = RT_BVH_EMPTY; }
            RG_END(RG_LEAF);
        }
        if (idx >= 0 && tv.ref == RT_BVH_EMPTY && pend == RT_BVH_EMPTY) {
            q.hit[idx] = make_float4(tv.best.t, __int_as_float(tv.best.id), __int_as_float(tv.best.slot), __int_as_float(tv.floor_id));
            idx = -1;
        }
    }
    RG_END(RG_TRAV_ALL);
    if (kTimers && !STATS && lane == 0) atomicMax(&stats[RT_STAT_WORDS], (unsigned long long)dbg_turns);   // development builds: longest wave, in loop turns
    if (STATS || kTimers) flush_counters<STATS>(cnt, 0u, stats, RT_K_TRAVERSE);
}

/* ------------------------------------------------------------------ shade */

// One thread per pool entry: finish the hit, run ray_color's body, start the next ray (everything of
// world.hit() that comes before the BVH walk), append the survivor to the next buffer.

// The bounce of pool entry i (flags = word 3 of its direction quad `d`, 0 for a lane without an entry).  Called by all lanes of
// the wave (shade_hit shares the wave's noise hits).
template <bool STATS, int EXT>
RT_DEV void shade_entry(const DScene &sc, const WfArgs &w, const PathQ &cur, const PathQ &nxt, WfCounters *ctr_it, uint32_t shard, uint32_t i,
                        float4 d, uint32_t flags, float4 *__restrict__ samples, uint32_t *pixel_rays, Counters &cnt) {
    const V3 background = v3(w.a.bg[0], w.a.bg[1], w.a.bg[2]);
    bool survive = false, walk = false;
    WfPath p;
    Trav tv;
    tv.ref = RT_BVH_EMPTY; tv.sp = 0; tv.best.t = 0.0f; tv.best.id = -1; tv.best.slot = 0; tv.floor_id = -1;
    const bool alive = (flags & WF_ALIVE) != 0u;
    phase_count<STATS>(cnt, PH_SHADE, alive);
    bool cont = false;
    float4 th = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float2 rg = make_float2(0.0f, 0.0f);
    if (alive) {
        RG_BEGIN(RG_LOAD);
        const float4 o = cur.ray_o[i], h = cur.hit[i];
        th = cur.thr[i]; rg = cur.rng[i];
        p.ps.ray.o = v3(o.x, o.y, o.z); p.ps.ray.d = v3(d.x, d.y, d.z); p.ps.ray.time = o.w;
        tv.best.t = h.x; tv.best.id = f2i(h.y); tv.best.slot = f2i(h.z); tv.floor_id = f2i(h.w);
        RG_END(RG_LOAD);
    }
    RG_BEGIN(RG_FINISH);
    finish_trace<STATS, EXT>(sc, alive, p.ps.ray, 0.001f, INFINITY, tv, cnt);     // (all lanes; rare, and before the path is unpacked)
    RG_END(RG_FINISH);
    if (alive) {
        RG_BEGIN(RG_LOAD);
        p.ps.throughput = v3(th.x, th.y, th.z); p.item = __float_as_uint(th.w);
        p.ps.radiance = v3(0.0f, 0.0f, 0.0f);
        p.ps.depth = (int)(flags & WF_DEPTH_MASK);
        int px, py;
        item_pixel(w, p.item, &px, &py, &p.sample);
        p.pixel = (uint32_t)py * (uint32_t)w.a.width + (uint32_t)px;
        p.rng.state = ((uint64_t)__float_as_uint(rg.y) << 32) | (uint64_t)__float_as_uint(rg.x);
        p.rng.inc = (((((uint64_t)p.pixel << 24) ^ (uint64_t)p.sample)) << 1u) | 1u;   // rts::path_rng's stream
        RG_END(RG_LOAD);
    }
    RG_BEGIN(RG_SHADE_ALL);
#ifdef RT_DEBUG_PATH
    if (alive && p.pixel == RT_DEBUG_PIXEL && p.sample == RT_DEBUG_SAMPLE)
        printf("PATH arrive depth_left %d | ray o %.9g %.9g %.9g d %.9g %.9g %.9g | hit t %.9g (bits %08x) id %d slot %d\n", p.ps.depth, p.ps.ray.o.x, p.ps.ray.o.y, p.ps.ray.o.z,
               p.ps.ray.d.x, p.ps.ray.d.y, p.ps.ray.d.z, tv.best.t, __float_as_uint(tv.best.t), tv.best.id, tv.best.slot);
#endif
    cont = shade_hit<STATS, EXT>(sc, alive, p.ps, tv.best, background, p.rng, cnt);     // (all lanes: the wave shares its noise hits)
    RG_END(RG_SHADE_ALL);
    if (alive) {
        if (cont && p.ps.depth > 0) survive = true;
        else {
            if (cont) end_on_black(p.ps);      // the next call would return black at depth 0 (camera.rs:260)
            samples[p.item] = make_float4(p.ps.radiance.x, p.ps.radiance.y, p.ps.radiance.z, 0.0f);
        }
    }
    phase_count<STATS>(cnt, PH_BEGIN, survive);
    LateFolds late;
    late.literal = false; late.axis = false;
    if (survive) {
        if (STATS) { cnt.rays += 1; if (pixel_rays) atomicAdd(&pixel_rays[p.pixel], 1u); }
        RG_BEGIN(RG_BEGIN_ALL);
        walk = begin_trace<STATS, EXT>(sc, p.ps.ray, 0.001f, INFINITY, p.rng, tv, cnt, late);
        RG_END(RG_BEGIN_ALL);
    }
#ifdef RT_DEBUG_PATH
    if (alive && p.pixel == RT_DEBUG_PIXEL && p.sample == RT_DEBUG_SAMPLE) {
        printf("PATH shade depth_left %d survive %d walk %d | new ray o %.9g %.9g %.9g d %.9g %.9g %.9g time %.9g | best t %.9g id %d slot %d | med ok %d %d t %.9g %.9g | o bits %08x %08x %08x d bits %08x %08x %08x time bits %08x\n",
               p.ps.depth, (int)survive, (int)walk, p.ps.ray.o.x, p.ps.ray.o.y, p.ps.ray.o.z, p.ps.ray.d.x, p.ps.ray.d.y, p.ps.ray.d.z, p.ps.ray.time,
               tv.best.t, tv.best.id, tv.best.slot, (int)late.med_ok[0], (int)late.med_ok[1], late.med_t[0], late.med_t[1],
               __float_as_uint(p.ps.ray.o.x), __float_as_uint(p.ps.ray.o.y), __float_as_uint(p.ps.ray.o.z), __float_as_uint(p.ps.ray.d.x), __float_as_uint(p.ps.ray.d.y), __float_as_uint(p.ps.ray.d.z), __float_as_uint(p.ps.ray.time));
    }
#endif
    RG_BEGIN(RG_STORE);
    const uint32_t pos = shard * w.cap + wave_append(&ctr_it->sh[shard].n_surv, survive);
    if (survive) store_entry(nxt, pos, p, tv, walk);
    RG_END(RG_STORE);
    // the rare serial folds last (all lanes), when only the ray is still needed
    if (settle_trace<STATS, EXT>(sc, survive, p.ps.ray, 0.001f, INFINITY, tv, walk, late, cnt)) restore_trace(nxt, pos, p.ps.depth, tv, walk);
}

template <bool STATS, int EXT>
__global__ void __launch_bounds__(256, shade_min_waves(STATS, EXT)) wf_shade_kernel(DScene sc_in, WfArgs w, PathQ cur, PathQ nxt, WfCounters *ctr, uint32_t it,
                                                       float4 *__restrict__ samples, unsigned long long *stats, uint32_t *pixel_rays, uint32_t with_tail) {
    DScene sc = sc_in;
    const uint32_t shard = blockIdx.y, off0 = blockIdx.x * blockDim.x;   // block-uniform; grid as in wf_generate_kernel
    uint32_t i = shard * w.cap + off0 + threadIdx.x;
    WfCounters *ctr_it = ctr + (it % RT_WF_RING);
    WfShard *c = &ctr_it->sh[shard];
    const uint32_t n = c->count_in;
    if (off0 >= n || (with_tail && c->tail_mode)) return;   // block-uniform (tail mode: wf_tail_kernel runs this iteration)
    Counters cnt{};
    uint32_t flags = 0;
    float4 d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#if RT_WF_SORT
    // Block-local sort by what the bounce will do (material type / miss / nothing): the 256 entries of
    // the block are dealt to the threads in key order, so most waves run one branch of ray_color's
    // material switch instead of all of them.  Which thread takes which entry changes nothing else.
    // (not for a scene whose hits are nearly all one kind — WfArgs::sort, chosen by the host: the Cornell box is 95 % solid
    // lambertian, and the deal costs it more than the few glass and lamp hits per wave do: +2.8 % without)
    if (w.sort) {
        __shared__ uint32_t s_count[8][4];
        __shared__ uint16_t s_perm[256];
        __shared__ uint32_t s_perlin[(EXT & EXT_NOISE) ? RT_PERLIN_WORDS : 1];
        const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
        uint32_t key = 7u;
        if (off0 + threadIdx.x < n) {
            const uint32_t f = __float_as_uint(cur.ray_d[i].w);
            if (f & WF_ALIVE) {
                const float4 h = cur.hit[i];
                const int32_t id = f2i(h.y), slot = f2i(h.z);
                if (id < 0) key = 6u;
                else {
                    const int mat = slot < 0 ? f2i(sc.media[2 * (uint32_t)(-slot - 1) + 1].y) : f2i(sc.slots[4 * (size_t)slot + 1].y);
                    // keys: 0 lambertian, solid colour  1 lambertian, checker / image  2 lambertian, Perlin noise (7 octaves:
                    // by far the longest branch)  3 metal  4 dielectric  5 emitter, isotropic  6 miss  7 no path
                    const int mtype = f2i(sc.materials[2 * mat].x), tkind = f2i(sc.materials[2 * mat + 1].w) - 1;   // rt_kernels.h material_texture()
                    key = mtype == RTG_MAT_METAL ? 3u : mtype == RTG_MAT_DIELECTRIC ? 4u : 5u;
                    if (mtype == RTG_MAT_LAMBERTIAN) key = tkind == RTG_TEX_SOLID ? 0u : tkind == RTG_TEX_NOISE ? 2u : 1u;
                }
            }
        }
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const unsigned long long m = __ballot(key == k);
            if (key == k) rank = (uint32_t)__popcll(m & ((1ull << ln) - 1ull));
            if (ln == 0) s_count[k][wv] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        uint32_t before = 0;     // entries with a smaller key, or the same key in an earlier wave
        for (uint32_t k = 0; k < 8u; ++k)
            for (uint32_t v = 0; v < 4u; ++v)
                if (k < key || (k == key && v < wv)) before += s_count[k][v];
        s_perm[before + rank] = (uint16_t)threadIdx.x;
        // A noise texture is 7 octaves x 8 corners of dependent table look-ups (perlin.rs:84-123): blocks that
        // hold such hits (key 2) copy the 6 KB table into LDS first (block-uniform decision).
        if ((EXT & EXT_NOISE) && s_count[2][0] + s_count[2][1] + s_count[2][2] + s_count[2][3] != 0u) {
            for (uint32_t k = threadIdx.x; k < RT_PERLIN_WORDS; k += blockDim.x) s_perlin[k] = sc.perlin[k];
            sc.perlin_lds = s_perlin;
        }
        __syncthreads();
        i = shard * w.cap + off0 + s_perm[threadIdx.x];
        if (off0 + s_perm[threadIdx.x] < n) { d = cur.ray_d[i]; flags = __float_as_uint(d.w); }
    } else if (off0 + threadIdx.x < n) {
        d = cur.ray_d[i]; flags = __float_as_uint(d.w);
    }
#else
    if (off0 + threadIdx.x < n) { d = cur.ray_d[i]; flags = __float_as_uint(d.w); }
#endif
    shade_entry<STATS, EXT>(sc, w, cur, nxt, ctr_it, shard, i, d, flags, samples, pixel_rays, cnt);
    if (STATS || kTimers) flush_counters<STATS>(cnt, 0u, stats, RT_K_SHADE);
}

/* ------------------------------------------------------------------ tail */

// The last paths of a batch.  Once every item has been handed out the pool only shrinks, and a few long paths keep it
// alive for dozens of iterations that each cost three launches' worth of fixed time (the traverse launch lasts as long as the
// longest single walk among its rays).  In "tail mode" (WfShard::tail_mode, decided by the previous generate: all items out, at
// most tail_max paths live) an iteration is run by this kernel instead of traverse + shade: a lane takes one path through up to
// max_bounces bounces — the walk, the shade and begin_trace of each, the same functions in the same order, hence the same bits —
// without any launch boundary, so a bounce costs a wave the longest walk among ITS 64 rays; what is still alive then is appended to
// the next pool buffer exactly as shade does it (compacted: the next tail iteration deals full sets again, the lanes of the
// paths that ended in between are not carried along).  Launched in every iteration the host marks (with_tail of traverse and
// shade); a no-op (one counter read per wave) outside tail mode.  Its waves are independent of each other.
template <bool STATS, int EXT>
__global__ void __launch_bounds__(256) wf_tail_kernel(DScene sc, WfArgs w, PathQ q, PathQ nxt, WfCounters *ctr, uint32_t it, uint32_t first_bounces,
                                                     uint32_t later_bounces, uint32_t set_lanes_max, uint32_t thin_below, uint32_t lds_depth, StackEntry *ovf, float4 *__restrict__ samples,
                                                     unsigned long long *stats, uint32_t *pixel_rays, uint32_t regroup, uint32_t hint) {
    extern __shared__ StackEntry lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wpb = blockDim.x >> 6;
    const uint32_t gw = blockIdx.x * wpb + (threadIdx.x >> 6), n_gw = gridDim.x * wpb;   // this wave among all waves of the launch
    WfCounters *c = ctr + (it % RT_WF_RING);            // this iteration's counters: count_in fixed, n_surv appended to below
    const uint32_t nth = c->sh[0].tail_mode;
    if (!nth) return;                                   // the same answer in every block
    // the first tail iteration is short: most of what is alive when the tail begins ends within a few bounces, and the survivors
    // are worth re-dealing into full waves; from then on little ends (what lives that long lives to the depth limit)
    const uint32_t max_bounces = nth == 1u ? first_bounces : later_bounces;
    const uint32_t my_count = c->sh[lane].count_in;     // lane s looks at shard s
    // Paths per wave: 64 while that makes more than thin_below waves (the launch is then bound by instruction issue, and a full
    // wave is the cheapest way to issue); fewer when the paths are few (the launch is then bound by its slowest wave, and a bounce
    // costs a wave the longest walk among its rays: the fewer rays, the shorter)
    uint32_t set_lanes = set_lanes_max;
    {
        uint32_t live = wave_sum(my_count);
        live = (uint32_t)__shfl((int)live, 0, 64);
        while (set_lanes > 8u && (live + set_lanes / 2u - 1u) / (set_lanes / 2u) <= thin_below) set_lanes >>= 1;
    }
    // The live paths sit unevenly in the shards (a shard holds neighbouring tiles: sky dies at once, a glass sphere lives to the
    // depth limit), so the sets of 64 paths are numbered through all shards and dealt to the waves of the launch round-robin.
    uint32_t sets_incl = (my_count + set_lanes - 1u) / set_lanes;          // inclusive prefix sum of the shards' set counts
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)sets_incl, o, 64);
        if (lane >= o) sets_incl += up;
    }
    const uint32_t n_sets = (uint32_t)__shfl((int)sets_incl, 63, 64);
    if (blockIdx.x * wpb >= n_sets) return;             // block-uniform: no set for any wave of this workgroup
    const Stack stack{(LdsStackPtr)lds_stack + threadIdx.x, blockDim.x, lds_depth, ovf ? ovf + (size_t)blockIdx.x * blockDim.x + threadIdx.x : nullptr, gridDim.x * blockDim.x};
    const V3 background = v3(w.a.bg[0], w.a.bg[1], w.a.bg[2]);
    const float T_MIN = 0.001f;
    Counters cnt{};
    const unsigned long long tail_t0 = (kTimers && !STATS) ? __builtin_readcyclecounter() : 0ull;
    for (uint32_t set = gw; set < n_sets; set += n_gw) {
        const uint32_t shard = (uint32_t)__popcll(__ballot(sets_incl <= set));          // first shard whose prefix exceeds `set`
        const uint32_t first_set = shard ? (uint32_t)__shfl((int)sets_incl, (int)shard - 1, 64) : 0u;
        const uint32_t n = (uint32_t)__shfl((int)my_count, (int)shard, 64), off = (set - first_set) * set_lanes;
        const uint32_t i = shard * w.cap + off + lane;
        bool alive = false, walk = false;
        uint32_t where = 0xffffffffu;      // the slot the path's last hit was on, once it has had one in this launch
        WfPath p;
        p.ps.ray.o = v3(0.0f, 0.0f, 0.0f); p.ps.ray.d = p.ps.ray.o; p.ps.ray.time = 0.0f;
        p.ps.throughput = p.ps.ray.o; p.ps.radiance = p.ps.ray.o; p.ps.depth = 0;
        p.pixel = 0; p.sample = 0; p.item = 0; p.rng.state = 0; p.rng.inc = 1;
        Trav tv;
        tv.ref = RT_BVH_EMPTY; tv.sp = 0; tv.best.t = 0.0f; tv.best.id = -1; tv.best.slot = 0; tv.floor_id = -1;
        tv.inv = p.ps.ray.o; tv.oi = p.ps.ray.o; tv.fat = 0.0f;
        if (lane < set_lanes && off + lane < n) {
            const float4 d = q.ray_d[i];
            const uint32_t flags = __float_as_uint(d.w);
            if (flags & WF_ALIVE) {
                const float4 o = q.ray_o[i], th = q.thr[i], h = q.hit[i];
                const float2 rg = q.rng[i];
                alive = true;
                walk = (flags & WF_WALK) != 0u;
                p.ps.ray.o = v3(o.x, o.y, o.z); p.ps.ray.d = v3(d.x, d.y, d.z); p.ps.ray.time = o.w;
                p.ps.throughput = v3(th.x, th.y, th.z); p.item = __float_as_uint(th.w);
                p.ps.depth = (int)(flags & WF_DEPTH_MASK);
                int px, py;
                item_pixel(w, p.item, &px, &py, &p.sample);
                p.pixel = (uint32_t)py * (uint32_t)w.a.width + (uint32_t)px;
                p.rng.state = ((uint64_t)__float_as_uint(rg.y) << 32) | (uint64_t)__float_as_uint(rg.x);
                p.rng.inc = (((((uint64_t)p.pixel << 24) ^ (uint64_t)p.sample)) << 1u) | 1u;   // rts::path_rng's stream
                tv.best.t = h.x; tv.best.id = f2i(h.y); tv.best.slot = f2i(h.z); tv.floor_id = f2i(h.w);
                if (walk) {     // as the traverse kernel starts a walk: the box-test arithmetic of begin_trace()
                    const Ray &r = p.ps.ray;
                    tv.inv = box_inv(r.d);
                    tv.oi = r.o * tv.inv;
                    tv.fat = ray_fat(sc, r);
                    tv.sp = 0;
                    tv.ref = sc.root_ref;
                }
            }
        }
        for (uint32_t bounce = 0; bounce < max_bounces && __ballot(alive) != 0ull; ++bounce) {
            RG_BEGIN(RG_TRAV_ALL);
            {      // the walk, as wf_traverse_kernel does it without the refill: leaves put aside, RT_WF_NODE_STEPS node steps per turn
                const bool walking = alive && walk;
                // What lives this long is caught inside something and hits it again and again from within (rt_device.hip, tail
                // kernel): the slot of the last hit is tested before the walk, so the walk starts with best.t at the far side of
                // that object instead of infinity and prunes everything beyond it from the first node on.  One more candidate for
                // the same fold (the walk meets the slot again and leaves it as it is): the same closest hit, the same tie-break.
                if (hint && walking && where < sc.always_begin) hint_sphere(sc, where, p.ps.ray, T_MIN, tv);
                int32_t pend = RT_BVH_EMPTY;
                for (;;) {
#pragma unroll
                    for (int k = 0; k < RT_WF_NODE_STEPS; ++k) {
                        if (walking && tv.ref < 0 && pend == RT_BVH_EMPTY) { pend = tv.ref; tv.ref = pop_ref(stack, tv.sp); }
                        const bool on_node = walking && tv.ref >= 0 && tv.ref != RT_BVH_EMPTY;
                        phase_count<STATS>(cnt, PH_NODE, on_node);
                        RG_BEGIN(RG_NODE);
                        if (on_node) trav_node_step<STATS>(sc, T_MIN, tv, stack, cnt);
                        RG_END(RG_NODE);
                    }
                    if (walking && tv.ref < 0 && pend == RT_BVH_EMPTY) { pend = tv.ref; tv.ref = pop_ref(stack, tv.sp); }
                    const bool has_leaf = walking && pend != RT_BVH_EMPTY;
                    const unsigned long long m_leaf = __ballot(has_leaf);
                    const unsigned long long m_node = __ballot(walking && tv.ref >= 0 && tv.ref != RT_BVH_EMPTY);
                    if (m_leaf == 0ull && m_node == 0ull) break;
                    const uint32_t n_leaf = (uint32_t)__popcll(m_leaf);
                    const uint32_t n_live = (uint32_t)__popcll(__ballot(walking && (tv.ref != RT_BVH_EMPTY || pend != RT_BVH_EMPTY)));
                    if (m_leaf != 0ull && (n_leaf >= RT_LEAF_MIN || n_leaf * RT_LEAF_FRAC >= n_live || m_node == 0ull)) {
                        phase_count<STATS>(cnt, PH_LEAF, has_leaf);
                        RG_BEGIN(RG_LEAF);
                        if (has_leaf) { trav_leaf_test<STATS, EXT & (EXT_XFORM | EXT_TRI | EXT_CHAIN)>(sc, p.ps.ray, T_MIN, tv, pend, cnt); pend = RT_BVH_EMPTY; }
                        RG_END(RG_LEAF);
                    }
                }
            }
            RG_END(RG_TRAV_ALL);
            phase_count<STATS>(cnt, PH_SHADE, alive);
            bool survive = false;
            RG_BEGIN(RG_SHADE_ALL);
            finish_trace<STATS, EXT>(sc, alive, p.ps.ray, T_MIN, INFINITY, tv, cnt);
            if (alive && tv.best.id >= 0) where = (uint32_t)tv.best.slot;
            const bool cont = shade_hit<STATS, EXT>(sc, alive, p.ps, tv.best, background, p.rng, cnt);
            if (alive) {
                survive = cont && p.ps.depth > 0;
                if (!survive) {
                    if (cont) end_on_black(p.ps);
                    samples[p.item] = make_float4(p.ps.radiance.x, p.ps.radiance.y, p.ps.radiance.z, 0.0f);
                    alive = false;
                }
            }
            RG_END(RG_SHADE_ALL);
            phase_count<STATS>(cnt, PH_BEGIN, survive);
            RG_BEGIN(RG_BEGIN_ALL);
            LateFolds late;
            late.literal = false; late.axis = false;
            if (survive) {
                if (STATS) { cnt.rays += 1; if (pixel_rays) atomicAdd(&pixel_rays[p.pixel], 1u); }
                walk = begin_trace<STATS, EXT>(sc, p.ps.ray, T_MIN, INFINITY, p.rng, tv, cnt, late);
            }
            settle_trace<STATS, EXT>(sc, survive, p.ps.ray, T_MIN, INFINITY, tv, walk, late, cnt);   // (all lanes)
            RG_END(RG_BEGIN_ALL);
        }
        // What is still alive goes on in the next iteration.  Regrouped: appended to the shard its last hit's slot names, so that the
        // next iteration's sets — consecutive entries of a shard — hold paths that are in the same place.  What lives this long
        // is caught somewhere (the Book-1 frame: nine of ten are reflected to and fro inside the big metal sphere, the rest
        // inside a handful of others), so such a set walks the same nodes and runs one branch of the material switch, bounce
        // after bounce.  Where an entry sits changes no bit: its sample is parked by item index.  Otherwise: to the set's own
        // shard, as shade does.
        const uint32_t pos = regroup ? wave_append_to(c, w.cap, alive, (where == 0xffffffffu ? shard : where) % RT_WF_SHARDS) : shard * w.cap + wave_append(&c->sh[shard].n_surv, alive);
        if (alive) store_entry(nxt, pos, p, tv, walk);
    }
    if (kTimers && !STATS) {   // development builds: the kernel's own timers apart from the pipeline's (stats + 64)
        const unsigned long long dt = __builtin_readcyclecounter() - tail_t0;
        if (lane == 0) atomicMax(&stats[64 + RT_STAT_WORDS], dt);
        flush_counters<STATS>(cnt, 0u, stats + 64, RT_K_TAIL);
        return;
    }
    if (STATS) flush_counters<STATS>(cnt, 0u, stats, RT_K_TAIL);
}

/* ------------------------------------------------------------------ fold */

// pixel_color += ray_color(sample k), k ascending (camera.rs:195-250): one thread per pixel of the window.
__global__ void __launch_bounds__(256) wf_fold_kernel(WfArgs w, const float4 *__restrict__ samples, float4 *__restrict__ out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gid >> 6, ln = gid & 63u;
    if (t >= w.n_tiles) return;
    const uint32_t ty = t / w.tiles_x, tx = t - ty * w.tiles_x;
    const int x = w.a.x0 + (int)(tx * 8u + (ln & 7u)), y = w.a.y0 + (int)(ty * 8u + (ln >> 3));
    if (x >= w.a.x1 || y >= w.a.y1) return;
    const size_t opix = (size_t)y * (size_t)w.a.width + (size_t)x;
    float4 o = out[opix];
    const float4 *s = samples + ((size_t)t * w.spp_batch) * 64u + ln;
    for (uint32_t k = 0; k < w.spp_batch; ++k) {
        const float4 v = s[(size_t)k * 64u];
        o.x += v.x; o.y += v.y; o.z += v.z;
    }
    o.w += (float)w.spp_batch;
    out[opix] = o;
}

}  // namespace rtk
#endif
