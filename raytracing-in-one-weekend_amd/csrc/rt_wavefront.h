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
//   fold      : pixel_color += ray_color, in sample order (camera.rs:195-250), from the sample buffer
//
// Every path consumes its own RNG stream in the reference's call order, whatever lane, wave or
// iteration its pieces run in, so the image is the same bits as the per-lane megakernel's (and the
// fold is the reference's order exactly, independent of pool size and launch geometry).
//
// Pool entry = 6 float4 in SoA arrays (96 B): origin+time | direction+flags | throughput+pixel |
// radiance+sample | rng state+item | hit (t, id, slot, floor_id).  Two pool buffers (ping-pong).
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H

#include "rt_kernels.h"

namespace rtk {

struct PathQ {
    float4 *ray_o;   // o.x o.y o.z time
    float4 *ray_d;   // d.x d.y d.z flags (depth | WF_ALIVE | WF_WALK)
    float4 *thr;     // throughput.xyz, pixel index
    float4 *rad;     // radiance.xyz, absolute sample index
    float4 *rng;     // PCG state lo, hi, item index, -
    float4 *hit;     // best.t, best.id, best.slot, floor_id
};

enum : uint32_t { WF_DEPTH_MASK = 0xffu, WF_ALIVE = 0x100u, WF_WALK = 0x200u };

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
    uint32_t pad[60];
};
struct WfCounters {      // one per iteration, in a ring; written by the previous iteration's generate
    WfShard sh[RT_WF_SHARDS];
};

struct WfArgs {
    RenderArgs a;
    uint32_t pool;        // entries per pool buffer = RT_WF_SHARDS * cap
    uint32_t cap;         // entries per shard (multiple of 256)
    uint32_t n_items;     // items of this batch = n_tiles * spp_batch * 64
    uint32_t spp_batch;   // samples per pixel in this batch
    uint32_t sample0;     // absolute index of the batch's first sample
    uint32_t tiles_x, n_tiles;   // 8x8 pixel tiles over the window
};

// item -> (pixel, sample): 64 consecutive items are one 8x8 tile at one sample; a tile's samples are
// consecutive groups.  The sample buffer uses the same index, so the fold reads it coalesced.
RT_DEV bool item_pixel(const WfArgs &w, uint32_t g, int *x, int *y, uint32_t *sample) {
    const uint32_t group = g >> 6, ln = g & 63u;
    const uint32_t t = group / w.spp_batch, k = group - t * w.spp_batch;
    const uint32_t ty = t / w.tiles_x, tx = t - ty * w.tiles_x;
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
    q.thr[i] = make_float4(p.ps.throughput.x, p.ps.throughput.y, p.ps.throughput.z, __uint_as_float(p.pixel));
    q.rad[i] = make_float4(p.ps.radiance.x, p.ps.radiance.y, p.ps.radiance.z, __uint_as_float(p.sample));
    q.rng[i] = make_float4(__uint_as_float((uint32_t)p.rng.state), __uint_as_float((uint32_t)(p.rng.state >> 32)),
                           __uint_as_float(p.item), 0.0f);
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

template <bool STATS>
RT_DEV void flush_counters(const Counters &cnt, uint32_t paths, unsigned long long *stats) {
    if (!STATS) return;
    uint32_t vals[31];
    vals[0] = paths;
    vals[1] = cnt.rays; vals[2] = cnt.node_visits;
    for (int i = 0; i < 6; ++i) vals[3 + i] = cnt.prim_tests[i];
    vals[9] = cnt.medium_tests; vals[10] = cnt.light_pdf_evals;
    for (int i = 0; i < 5; ++i) vals[11 + i] = cnt.shades[i];
    for (int i = 0; i < 4; ++i) vals[16 + i] = cnt.tex_evals[i];
    vals[20] = cnt.accepted;
    for (int i = 0; i < 5; ++i) { vals[21 + i] = cnt.phase_lanes[i]; vals[26 + i] = cnt.phase_slots[i]; }
    for (int i = 0; i < 31; ++i) {
        uint32_t w = wave_sum(vals[i]);
        if ((threadIdx.x & 63u) == 0 && w) atomicAdd(&stats[i], (unsigned long long)w);
    }
}

/* ------------------------------------------------------------------ generate */

// Fills every shard of the next pool buffer behind shade's survivors with new paths and publishes the
// next iteration's counters.  Thread j looks at pool entry j: items are numbered through the free
// entries in pool order, so nothing needs an atomic.
template <bool STATS>
__global__ void __launch_bounds__(256) wf_generate_kernel(DScene sc, WfArgs w, PathQ nxt, WfCounters *ctr, uint32_t it,
                                                          float4 *__restrict__ samples, uint32_t *status, unsigned long long *stats,
                                                          uint32_t *pixel_rays) {
    const WfCounters *c = ctr + (it % RT_WF_RING);
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t shard = (blockIdx.x * blockDim.x) / w.cap, off0 = (blockIdx.x * blockDim.x) - shard * w.cap;   // block-uniform
    const uint32_t item_head = c->sh[0].item_head;
    uint32_t room_before = 0;                         // free entries in the shards before this one
    for (uint32_t t = 0; t < shard; ++t) room_before += w.cap - c->sh[t].n_surv;
    const uint32_t n_surv = c->sh[shard].n_surv;
    const uint32_t left = w.n_items - item_head;
    const uint32_t left_here = left > room_before ? left - room_before : 0u;
    const uint32_t room = w.cap - n_surv;
    const uint32_t n_new = room < left_here ? room : left_here;
    if (off0 == 0 && threadIdx.x == 0) {              // first block of the shard: next iteration's counters
        WfShard *n = &ctr[(it + 1u) % RT_WF_RING].sh[shard];
        n->count_in = n_surv + n_new; n->n_surv = 0; n->cursor = 0;
        if (shard == 0) {
            uint32_t room_all = 0, surv_all = 0;
            for (uint32_t t = 0; t < RT_WF_SHARDS; ++t) { room_all += w.cap - c->sh[t].n_surv; surv_all += c->sh[t].n_surv; }
            const uint32_t issued = room_all < left ? room_all : left;
            n->item_head = item_head + issued;
            status[0] = surv_all + issued; status[1] = item_head + issued; status[2] = it + 1u;
        }
    }
    if (off0 >= n_surv + n_new || off0 + blockDim.x <= n_surv) return;   // block-uniform: nothing to make in this block
    const uint32_t off = off0 + threadIdx.x;
    const bool mine = off >= n_surv && off < n_surv + n_new;
    Counters cnt{};
    const uint32_t g = item_head + room_before + (off - n_surv), e = j;
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
    if (start) {
        p.pixel = (uint32_t)y * (uint32_t)w.a.width + (uint32_t)x;
        rts::path_rng(p.rng, w.a.seed, p.pixel, p.sample);
        p.ps.ray = camera_ray(sc, w.a, x, y, p.pixel, p.sample, p.rng);
        p.ps.throughput = v3(1.0f, 1.0f, 1.0f);
        p.ps.radiance = v3(0.0f, 0.0f, 0.0f);
        p.ps.depth = w.a.max_depth;
        if (STATS) { cnt.rays += 1; if (pixel_rays) atomicAdd(&pixel_rays[p.pixel], 1u); }
        const bool walk = begin_trace<STATS>(sc, p.ps.ray, 0.001f, INFINITY, p.rng, tv, cnt);
        store_entry(nxt, e, p, tv, walk);
    }
    if (STATS) flush_counters<STATS>(cnt, inside ? 1u : 0u, stats);
}

/* ------------------------------------------------------------------ traverse */

#ifndef RT_WF_FETCH_MIN
#define RT_WF_FETCH_MIN 20     // refill when at least this many lanes of the wave are idle
#endif
#ifndef RT_WF_CHUNK
#define RT_WF_CHUNK 64u        // entries a wave takes from its shard's cursor at a time
#endif

// Persistent waves.  A lane holds one ray; when RT_WF_FETCH_MIN lanes have finished theirs the wave
// takes the next entries of the pool for them, so the node loop runs with most lanes busy.  Inside,
// "while-while": node steps for everyone standing on an inner node, leaf steps once enough lanes
// stand on a leaf (RT_LEAF_MIN) or nobody is left on an inner node.
template <bool STATS>
__global__ void __launch_bounds__(256) wf_traverse_kernel(DScene sc, PathQ q, WfCounters *ctr, uint32_t it, uint32_t cap, unsigned long long *stats) {
    extern __shared__ uint2 lds_stack[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint2 *stack = lds_stack + tid;
    const uint32_t stride = blockDim.x;
    const uint32_t shard = blockIdx.x % RT_WF_SHARDS;
    WfShard *c = &ctr[it % RT_WF_RING].sh[shard];
    const uint32_t n = c->count_in, shard_base = shard * cap;
    Counters cnt{};
    uint32_t w_next = 0, w_end = 0;
    bool exhausted = n == 0;
    int32_t idx = -1;
    Ray r;
    r.o = v3(0.0f, 0.0f, 0.0f); r.d = r.o; r.time = 0.0f;
    Trav tv;
    tv.ref = RT_BVH_EMPTY; tv.sp = 0; tv.best.t = 0.0f; tv.best.id = -1; tv.best.slot = 0; tv.floor_id = -1;
    tv.inv = r.o; tv.oi = r.o;
    const float T_MIN = 0.001f;
    for (;;) {
        const unsigned long long m_idle = __ballot(idx < 0);
        const uint32_t n_idle = (uint32_t)__popcll(m_idle);
        if (n_idle == 64u && exhausted) break;
        if (!exhausted && n_idle >= RT_WF_FETCH_MIN) {
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
                        // the same box-test arithmetic as begin_trace()
                        float dx = fabsf(r.d.x) < 1.0e-30f ? 1.0e-30f : r.d.x;
                        float dy = fabsf(r.d.y) < 1.0e-30f ? 1.0e-30f : r.d.y;
                        float dz = fabsf(r.d.z) < 1.0e-30f ? 1.0e-30f : r.d.z;
                        tv.inv = v3(1.0f / dx, 1.0f / dy, 1.0f / dz);
                        tv.oi = r.o * tv.inv;
                        tv.sp = 0;
                        tv.ref = sc.root_ref;
                    }
                }
            }
            w_next += n_idle < avail ? n_idle : avail;
            if (__ballot(idx >= 0) == 0ull) continue;   // nothing to walk among what was fetched
        }
        // node steps
        phase_count<STATS>(cnt, PH_NODE, idx >= 0 && tv.ref >= 0 && tv.ref != RT_BVH_EMPTY);
        if (idx >= 0 && tv.ref >= 0 && tv.ref != RT_BVH_EMPTY) trav_node_step<STATS>(sc, T_MIN, tv, stack, stride, cnt);
        const bool at_leaf = idx >= 0 && tv.ref < 0;
        const unsigned long long m_leaf = __ballot(at_leaf);
        const unsigned long long m_node = __ballot(idx >= 0 && tv.ref >= 0 && tv.ref != RT_BVH_EMPTY);
        if (m_leaf != 0ull && (__popcll(m_leaf) >= RT_LEAF_MIN || m_node == 0ull)) {
            phase_count<STATS>(cnt, PH_LEAF, at_leaf);
            if (at_leaf) trav_leaf_step<STATS>(sc, r, T_MIN, tv, stack, stride, cnt);
        }
        if (idx >= 0 && tv.ref == RT_BVH_EMPTY) {
            q.hit[idx] = make_float4(tv.best.t, __int_as_float(tv.best.id), __int_as_float(tv.best.slot), __int_as_float(tv.floor_id));
            idx = -1;
        }
    }
    if (STATS) flush_counters<STATS>(cnt, 0u, stats);
}

/* ------------------------------------------------------------------ shade */

// One thread per pool entry: finish the hit, run ray_color's body, start the next ray (everything of
// world.hit() that comes before the BVH walk), append the survivor to the next buffer.
template <bool STATS>
__global__ void __launch_bounds__(256) wf_shade_kernel(DScene sc, WfArgs w, PathQ cur, PathQ nxt, WfCounters *ctr, uint32_t it,
                                                       float4 *__restrict__ samples, unsigned long long *stats, uint32_t *pixel_rays) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t shard = (blockIdx.x * blockDim.x) / w.cap, off0 = (blockIdx.x * blockDim.x) - shard * w.cap;   // block-uniform
    WfShard *c = &ctr[it % RT_WF_RING].sh[shard];
    const uint32_t n = c->count_in;
    if (off0 >= n) return;   // block-uniform
    Counters cnt{};
    const V3 background = v3(w.a.bg[0], w.a.bg[1], w.a.bg[2]);
    bool survive = false, walk = false;
    WfPath p;
    Trav tv;
    tv.ref = RT_BVH_EMPTY; tv.sp = 0; tv.best.t = 0.0f; tv.best.id = -1; tv.best.slot = 0; tv.floor_id = -1;
    uint32_t flags = 0;
    float4 d = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (off0 + threadIdx.x < n) { d = cur.ray_d[i]; flags = __float_as_uint(d.w); }
    const bool alive = (flags & WF_ALIVE) != 0u;
    phase_count<STATS>(cnt, PH_SHADE, alive);
    if (alive) {
        const float4 o = cur.ray_o[i], th = cur.thr[i], ra = cur.rad[i], rg = cur.rng[i], h = cur.hit[i];
        p.ps.ray.o = v3(o.x, o.y, o.z); p.ps.ray.d = v3(d.x, d.y, d.z); p.ps.ray.time = o.w;
        p.ps.throughput = v3(th.x, th.y, th.z); p.pixel = __float_as_uint(th.w);
        p.ps.radiance = v3(ra.x, ra.y, ra.z); p.sample = __float_as_uint(ra.w);
        p.ps.depth = (int)(flags & WF_DEPTH_MASK);
        p.item = __float_as_uint(rg.z);
        p.rng.state = ((uint64_t)__float_as_uint(rg.y) << 32) | (uint64_t)__float_as_uint(rg.x);
        p.rng.inc = (((((uint64_t)p.pixel << 24) ^ (uint64_t)p.sample)) << 1u) | 1u;   // rts::path_rng's stream
        tv.best.t = h.x; tv.best.id = f2i(h.y); tv.best.slot = f2i(h.z); tv.floor_id = f2i(h.w);
        finish_trace<STATS>(sc, p.ps.ray, 0.001f, INFINITY, tv, cnt);
        const bool cont = shade_hit<STATS>(sc, p.ps, tv.best, background, p.rng, cnt);
        if (cont && p.ps.depth > 0) survive = true;
        else samples[p.item] = make_float4(p.ps.radiance.x, p.ps.radiance.y, p.ps.radiance.z, 0.0f);
    }
    phase_count<STATS>(cnt, PH_BEGIN, survive);
    if (survive) {
        if (STATS) { cnt.rays += 1; if (pixel_rays) atomicAdd(&pixel_rays[p.pixel], 1u); }
        walk = begin_trace<STATS>(sc, p.ps.ray, 0.001f, INFINITY, p.rng, tv, cnt);
    }
    const uint32_t pos = shard * w.cap + wave_append(&c->n_surv, survive);
    if (survive) store_entry(nxt, pos, p, tv, walk);
    if (STATS) flush_counters<STATS>(cnt, 0u, stats);
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
