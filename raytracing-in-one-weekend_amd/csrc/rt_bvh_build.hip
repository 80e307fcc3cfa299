// rt_bvh_build.hip — the traversal BVH built on the GPU (load time of multi-million-triangle scenes).
//
// The reference has nothing like it: its BvhNode::new (raytracer/src/bvh.rs:15-55) sorts on the CPU, and its
// TriangleMesh has no tree at all (triangle_mesh.rs:250-267).  The host flattener's binned-SAH build
// (rt_host_flatten.cpp) takes seconds for the 2.26 M triangles of BASELINE config 4; this one takes milliseconds:
//
//   1. 63-bit Morton code of every leaf box centre                      one thread per primitive
//   2. radix sort of (code, primitive)                                   hipCUB (a plain library sort)
//   3. binary radix tree over the sorted codes (Karras 2012)             one thread per inner node, no atomics
//   4. depth of every inner node (walk to the root)                      one thread per inner node
//   5. boxes bottom-up, one launch per tree level, deepest first         the launch boundary is the only fence
//      (the usual one-kernel scheme — the second child to arrive does the parent — needs two agent-scope
//      fences per step, each an L2 write-back / invalidate across the 8 XCDs: 14.5 ms for 2.26 M leaves
//      against 0.5 ms for ~50 launches that each skim a 9 MB depth array)
//   6. collapse to the 4-wide, 8-bit-quantised nodes of include/rt_api.h: an inner node at even depth adopts
//      its grandchildren                                                 one thread per inner node + a prefix sum
//
// Step 3 has a second form (algo = RT_BVH_DEVICE_PLOC, the default of the device build): parallel locally-ordered
// clustering (Meister & Bittner 2018) — bottom-up, every cluster looks at its 2 x 32 neighbours in Morton order for the one
// it makes the smallest box with, mutual choices merge, the survivors are compacted, repeat until one is left (about 30
// rounds of three small kernels and two prefix sums).  It gives boxes as it goes (no step 5) and a tree close to the host's
// binned SAH, where Morton splits are a third looser.  Its leaves are then put in depth-first order, so that the two
// leaves of a pair are neighbours and memory order follows the tree.
//
// Which tree the kernels walk changes no result: the closest hit and its tie-break do not depend on the order of
// the search (rt_kernels.h node4_visit), which is what the parity test of this path checks — images bit-identical to
// the host-built tree's.  All accesses are HBM streams over arrays of n elements; the build is bandwidth- and
// latency-bound integer work, no LDS tiling needed.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "rt_api.h"
#include "rt_bvh_build.h"

namespace rtb {
namespace {

#define RTB_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { *err = std::string(#call) + ": " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP; } \
    } while (0)

__device__ __forceinline__ unsigned long long spread3(unsigned long long x) {   // 21 bits -> every third bit
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

struct Bounds { float lo[3], inv[3]; };   // centroid -> [0, 1): (c - lo) * inv

__global__ void __launch_bounds__(256) morton_kernel(const float *__restrict__ boxes, uint32_t n, Bounds b, unsigned long long *keys, uint32_t *vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *bx = boxes + 6 * (size_t)i;
    unsigned long long code = 0;
    for (int a = 0; a < 3; ++a) {
        float c = 0.5f * (bx[a] + bx[3 + a]);
        float u = (c - b.lo[a]) * b.inv[a];
        u = !(u > 0.0f) ? 0.0f : (u > 0.99999994f ? 0.99999994f : u);   // a NaN centre goes to cell 0
        const unsigned long long q = (unsigned long long)(u * 2097152.0f);
        code |= spread3(q) << (2 - a);
    }
    keys[i] = code;
    vals[i] = i;
}

// Length of the common prefix of the sorted keys i and j, ties broken by the position (Karras 2012, sec. 4).
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)i ^ (unsigned)j);
    return __clzll((long long)(a ^ b));
}

// Child encoding in the binary tree: >= 0 inner node, < 0 leaf ~c (a position in sorted order).
__global__ void __launch_bounds__(256) radix_tree_kernel(const unsigned long long *__restrict__ keys, int n, int2 *child, int *parent_inner,
                                                         int *parent_leaf) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int delta_min = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > delta_min) lmax *= 2;   // ends: delta is -1 outside [0, n)
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > delta_min) l += t;
    const int j = i + l * d;
    const int delta_node = delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > delta_node) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    int2 c;
    c.x = first == gamma ? ~gamma : gamma;
    c.y = last == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    child[i] = c;
    if (c.x >= 0) parent_inner[c.x] = i; else parent_leaf[~c.x] = i;
    if (c.y >= 0) parent_inner[c.y] = i; else parent_leaf[~c.y] = i;
    if (i == 0) parent_inner[0] = -1;
}

struct Box6 { float v[6]; };
__device__ __forceinline__ Box6 load_box(const float *p) { Box6 b; for (int k = 0; k < 6; ++k) b.v[k] = p[k]; return b; }
__device__ __forceinline__ Box6 child_box(int c, const float *boxes, const uint32_t *perm, const float *ibox) {
    return c >= 0 ? load_box(ibox + 6 * (size_t)c) : load_box(boxes + 6 * (size_t)perm[~c]);
}

// Boxes of the inner nodes at one depth: their inner children are one level deeper and were written by the previous launch.
__global__ void __launch_bounds__(256) fit_level_kernel(const float *__restrict__ boxes, const uint32_t *__restrict__ perm, int n_inner,
                                                        const int2 *__restrict__ child, const uint32_t *__restrict__ depth, uint32_t level, float *ibox) {
    const int node = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (node >= n_inner || depth[node] != level) return;
    const int2 c = child[node];
    const Box6 a = child_box(c.x, boxes, perm, ibox), b = child_box(c.y, boxes, perm, ibox);
    float *o = ibox + 6 * (size_t)node;
    for (int k = 0; k < 3; ++k) { o[k] = fminf(a.v[k], b.v[k]); o[3 + k] = fmaxf(a.v[3 + k], b.v[3 + k]); }
}

// An inner node whose two children are leaves becomes ONE leaf of two slots when leaf_max >= 2.
__device__ __forceinline__ bool is_pair(int2 c, uint32_t leaf_max) { return leaf_max >= 2u && c.x < 0 && c.y < 0; }

__global__ void __launch_bounds__(256) depth_kernel(int n_inner, const int2 *child, const int *parent_inner, uint32_t leaf_max, uint32_t *depth_out,
                                                    uint32_t *is_node4, uint32_t *max_depth) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    uint32_t depth = 0;
    if (i < n_inner) {   // (no early return: every lane of the wave takes part in the reduction below)
        for (int p = parent_inner[i]; p >= 0 && depth < 4096u; p = parent_inner[p]) ++depth;   // (bounded: the host refuses depth >= 4096)
        depth_out[i] = depth;
        is_node4[i] = ((depth & 1u) == 0u && !is_pair(child[i], leaf_max)) ? 1u : 0u;
    }
    // one atomic per wave: the wave's maximum
    uint32_t m = depth;
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63u) == 0u) atomicMax(max_depth, m);
}

struct Entry { Box6 box; int32_t ref; };

// ref of a child of a 4-wide node: inner node at even depth -> its 4-wide index; a leaf or a pair -> leaf ref
// (newpos: depth-first position of a leaf; the left leaf of a pair comes first and the right one directly after it)
__device__ __forceinline__ int32_t entry_ref(int c, const int2 *child, const uint32_t *index4, const uint32_t *newpos, uint32_t leaf_max) {
    if (c < 0) return ~(int32_t)((newpos[~c] << 3) | 0u);
    const int2 cc = child[c];
    if (is_pair(cc, leaf_max)) return ~(int32_t)((newpos[~cc.x] << 3) | 1u);
    return (int32_t)index4[c];
}

__global__ void __launch_bounds__(256) emit_kernel(int n_inner, const int2 *child, const float *boxes, const uint32_t *perm, const float *ibox,
                                                   const uint32_t *is_node4, const uint32_t *index4, const uint32_t *newpos, uint32_t leaf_max,
                                                   float *nodes) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_inner || !is_node4[i]) return;
    Entry e[4];
    int n = 0;
    const int2 c = child[i];
    const int cs[2] = {c.x, c.y};
    for (int k = 0; k < 2; ++k) {
        const int ck = cs[k];
        if (ck >= 0 && !is_pair(child[ck], leaf_max)) {   // an inner node at odd depth: adopt its children
            const int2 g = child[ck];
            e[n].box = child_box(g.x, boxes, perm, ibox); e[n].ref = entry_ref(g.x, child, index4, newpos, leaf_max); ++n;
            e[n].box = child_box(g.y, boxes, perm, ibox); e[n].ref = entry_ref(g.y, child, index4, newpos, leaf_max); ++n;
        } else {
            e[n].box = child_box(ck, boxes, perm, ibox); e[n].ref = entry_ref(ck, child, index4, newpos, leaf_max); ++n;
        }
    }
    // quantise exactly as the host builder does (rt_host_flatten.cpp): planes on the 8-bit grid of the union box,
    // rounded outwards and checked in double
    uint32_t q[6] = {0, 0, 0, 0, 0, 0};
    float origin[3], scale[3];
    for (int a = 0; a < 3; ++a) {
        float lo = FLT_MAX, hi = -FLT_MAX;
        for (int k = 0; k < n; ++k) { lo = fminf(lo, e[k].box.v[a]); hi = fmaxf(hi, e[k].box.v[3 + a]); }
        origin[a] = lo;
        scale[a] = (hi - lo) / 255.0f;
        for (int guard = 0; guard < 64 && (double)origin[a] + (double)scale[a] * 255.0 < (double)hi; ++guard) scale[a] = nextafterf(scale[a], FLT_MAX);
        for (int k = 0; k < n; ++k) {
            int ql = 0, qh = 0;
            if (scale[a] > 0.0f) {
                ql = (int)floor(((double)e[k].box.v[a] - (double)origin[a]) / (double)scale[a]);
                qh = (int)ceil(((double)e[k].box.v[3 + a] - (double)origin[a]) / (double)scale[a]);
                ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
                qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                while (ql > 0 && (double)origin[a] + (double)scale[a] * ql > (double)e[k].box.v[a]) --ql;
                while (qh < 255 && (double)origin[a] + (double)scale[a] * qh < (double)e[k].box.v[3 + a]) ++qh;
            }
            q[2 * a] |= (uint32_t)ql << (8 * k);
            q[2 * a + 1] |= (uint32_t)qh << (8 * k);
        }
    }
    float *d = nodes + (size_t)index4[i] * RT_NODE_FLOATS;
    for (int a = 0; a < 3; ++a) { d[a] = origin[a]; d[3 + a] = scale[a]; }
    for (int k = 0; k < 6; ++k) d[6 + k] = __uint_as_float(q[k]);
    for (int k = 0; k < 4; ++k) d[12 + k] = __int_as_float(k < n ? e[k].ref : RT_BVH_EMPTY);
}

/* ---- PLOC: clusters live in arrays of (box, id); id < 0 = leaf ~position, id >= 0 = inner node ---- */

__device__ __forceinline__ float union_half_area(const float *a, const float *b) {
    const float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]), dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ void __launch_bounds__(256) ploc_init_kernel(const float *__restrict__ boxes, const uint32_t *__restrict__ perm, int n, float *cbox, int *cid) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const float *b = boxes + 6 * (size_t)perm[i];
    for (int k = 0; k < 6; ++k) cbox[6 * (size_t)i + k] = b[k];
    cid[i] = ~i;
}

#define RTB_PLOC_RADIUS 32
// nearest neighbour of every cluster among the RADIUS before and after it; `pair_up`: neighbours by position (a round in
// which no two clusters chose each other, possible only with exact area ties, is repeated this way so the loop ends)
__global__ void __launch_bounds__(256) ploc_nn_kernel(const float *__restrict__ cbox, int c, int pair_up, int radius, int *nn) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= c) return;
    if (pair_up) { nn[i] = (i ^ 1) < c ? (i ^ 1) : i; return; }
    float mine[6];
    for (int k = 0; k < 6; ++k) mine[k] = cbox[6 * (size_t)i + k];
    const int lo = i - radius < 0 ? 0 : i - radius, hi = i + radius >= c ? c - 1 : i + radius;
    float best = FLT_MAX;
    int best_j = i;
    for (int j = lo; j <= hi; ++j) {
        if (j == i) continue;
        const float a = union_half_area(mine, cbox + 6 * (size_t)j);
        if (a < best || best_j == i) { best = a; best_j = j; }   // ties -> the lower index (also takes a NaN / inf area once)
    }
    nn[i] = best_j;
}

__global__ void __launch_bounds__(256) ploc_flag_kernel(const int *__restrict__ nn, int c, uint32_t *valid, uint32_t *merger) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= c) return;
    const int j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    valid[i] = (!mutual || i < j) ? 1u : 0u;     // the lower one of a pair carries the merged cluster
    merger[i] = (mutual && i < j) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) ploc_merge_kernel(const float *__restrict__ cbox, const int *__restrict__ cid, const int *__restrict__ nn, int c,
                                                         const uint32_t *__restrict__ valid, const uint32_t *__restrict__ pos,
                                                         const uint32_t *__restrict__ merger, const uint32_t *__restrict__ mpos, int node_base,
                                                         float *cbox_out, int *cid_out, int2 *child, float *ibox, int *parent_inner, int *parent_leaf,
                                                         uint32_t *leaf_count) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= c || !valid[i]) return;
    const uint32_t o = pos[i];
    if (!merger[i]) {
        for (int k = 0; k < 6; ++k) cbox_out[6 * (size_t)o + k] = cbox[6 * (size_t)i + k];
        cid_out[o] = cid[i];
        return;
    }
    const int j = nn[i], node = node_base + (int)mpos[i];
    const int a = cid[i], b = cid[j];
    float u[6];
    for (int k = 0; k < 3; ++k) {
        u[k] = fminf(cbox[6 * (size_t)i + k], cbox[6 * (size_t)j + k]);
        u[3 + k] = fmaxf(cbox[6 * (size_t)i + 3 + k], cbox[6 * (size_t)j + 3 + k]);
    }
    for (int k = 0; k < 6; ++k) { cbox_out[6 * (size_t)o + k] = u[k]; ibox[6 * (size_t)node + k] = u[k]; }
    cid_out[o] = node;
    child[node] = make_int2(a, b);
    if (a >= 0) parent_inner[a] = node; else parent_leaf[~a] = node;
    if (b >= 0) parent_inner[b] = node; else parent_leaf[~b] = node;
    leaf_count[node] = (a >= 0 ? leaf_count[a] : 1u) + (b >= 0 ? leaf_count[b] : 1u);
}

// Depth-first position of every leaf: the leaves of all left siblings on its way to the root come before it.
__global__ void __launch_bounds__(256) leaf_order_kernel(int n, const int2 *__restrict__ child, const int *__restrict__ parent_inner,
                                                         const int *__restrict__ parent_leaf, const uint32_t *__restrict__ leaf_count,
                                                         const uint32_t *__restrict__ sorted_prim, uint32_t *newpos, uint32_t *perm_out) {
    const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p >= n) return;
    uint32_t offset = 0;
    int cur = ~p;
    int guard = 0;   // a tree deeper than this is refused by the host afterwards; the bound keeps a broken one from hanging the wave
    for (int node = parent_leaf[p]; node >= 0 && guard < 8192; node = parent_inner[node], ++guard) {
        const int2 c = child[node];
        if (c.y == cur) offset += c.x >= 0 ? leaf_count[c.x] : 1u;
        cur = node;
    }
    if (offset >= (uint32_t)n) offset = (uint32_t)p;   // only with a broken tree; stay inside the arrays
    newpos[p] = offset;
    perm_out[offset] = sorted_prim[p];
}

__global__ void __launch_bounds__(256) identity_order_kernel(int n, const uint32_t *__restrict__ sorted_prim, uint32_t *newpos, uint32_t *perm_out) {
    const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (p >= n) return;
    newpos[p] = (uint32_t)p;
    perm_out[p] = sorted_prim[p];
}

}  // namespace

int build_bvh(hipStream_t stream, const float *d_boxes, uint32_t n, uint32_t leaf_max, uint32_t algo, const float lo[3], const float hi[3],
              Result *out, std::string *err) {
    *out = Result{};
    if (n == 0) { out->root_ref = RT_BVH_EMPTY; out->stack_depth = 1; return RT_OK; }
    if (n >= (1u << 28)) { *err = "too many primitives for the device BVH build"; return RT_E_INVALID; }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    char *scratch = nullptr;
    // from here on every error leaves through done(): events and scratch are released whatever was created so far
    auto done = [&](int rc) { if (scratch) (void)hipFree(scratch); if (ev0) (void)hipEventDestroy(ev0); if (ev1) (void)hipEventDestroy(ev1); return rc; };
#define RTB_TRY2(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { *err = std::string(#call) + ": " + hipGetErrorString(e_); return done(e_ == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP); } } while (0)
    RTB_TRY2(hipEventCreate(&ev0));
    RTB_TRY2(hipEventCreate(&ev1));
    RTB_TRY2(hipEventRecord(ev0, stream));
    const int ni = (int)n - 1;
    const unsigned gn = (n + 255u) / 256u, gi = ((unsigned)(ni > 0 ? ni : 1) + 255u) / 256u;
    // one scratch allocation: keys x2, vals x2, child, parents, ibox, depth, is_node4, index4, max_depth
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_k0 = take(8ull * n), o_k1 = take(8ull * n), o_v0 = take(4ull * n), o_v1 = take(4ull * n), o_child = take(8ull * n),
                 o_pi = take(4ull * n), o_pl = take(4ull * n), o_ibox = take(24ull * n), o_arr = take(4ull * n), o_is4 = take(4ull * n),
                 o_ix4 = take(4ull * (n + 1)), o_md = take(256);
    size_t sort_bytes = 0, scan_bytes = 0;
    RTB_TRY2(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint32_t *)nullptr,
                                               (uint32_t *)nullptr, (int)n, 0, 63, stream));
    RTB_TRY2(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n + 1, stream));
    const size_t o_tmp = take(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
    const bool ploc = algo != RT_BVH_DEVICE_LBVH;
    // PLOC: two cluster arrays (box + id), the neighbour choice, two flag arrays and their prefix sums
    const size_t o_cb0 = take(ploc ? 24ull * n : 0), o_cb1 = take(ploc ? 24ull * n : 0), o_ci0 = take(ploc ? 4ull * n : 0), o_ci1 = take(ploc ? 4ull * n : 0),
                 o_nn = take(ploc ? 4ull * n : 0), o_va = take(ploc ? 4ull * (n + 1) : 0), o_vp = take(ploc ? 4ull * (n + 1) : 0),
                 o_me = take(ploc ? 4ull * (n + 1) : 0), o_mp = take(ploc ? 4ull * (n + 1) : 0);
    RTB_TRY2(hipMalloc((void **)&scratch, off));
    auto *k0 = (unsigned long long *)(scratch + o_k0), *k1 = (unsigned long long *)(scratch + o_k1);
    auto *v0 = (uint32_t *)(scratch + o_v0), *v1 = (uint32_t *)(scratch + o_v1);
    auto *child = (int2 *)(scratch + o_child);
    auto *parent_inner = (int *)(scratch + o_pi), *parent_leaf = (int *)(scratch + o_pl);
    auto *ibox = (float *)(scratch + o_ibox);
    auto *depth = (uint32_t *)(scratch + o_arr), *is4 = (uint32_t *)(scratch + o_is4), *ix4 = (uint32_t *)(scratch + o_ix4);
    auto *max_depth = (uint32_t *)(scratch + o_md);

    Bounds b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = lo[a];
        const float ext = hi[a] - lo[a];
        b.inv[a] = ext > 0.0f ? 1.0f / ext : 0.0f;
    }
    hipLaunchKernelGGL(morton_kernel, dim3(gn), dim3(256), 0, stream, d_boxes, n, b, k0, v0);
    RTB_TRY2(hipcub::DeviceRadixSort::SortPairs(scratch + o_tmp, sort_bytes, k0, k1, v0, v1, (int)n, 0, 63, stream));
    uint32_t *perm = nullptr;
    RTB_TRY2(hipMalloc((void **)&perm, 4ull * n));
    out->perm = perm;
    const uint32_t *sorted_prim = v1;                 // primitive at Morton position p
    uint32_t *newpos = (uint32_t *)k0;                // (the unsorted keys and values are not needed any more)
    uint32_t *leaf_count = v0;
    if (n == 1 || (n == 2 && leaf_max >= 2u)) {   // the whole scene is one leaf
        RTB_TRY2(hipMemcpyAsync(perm, v1, 4ull * n, hipMemcpyDeviceToDevice, stream));
        out->root_ref = ~(int32_t)((0u << 3) | (n - 1u));
        out->stack_depth = 1;
    } else {
        RTB_TRY2(hipMemsetAsync(max_depth, 0, 4, stream));
        int root = 0;
        if (!ploc) {
            hipLaunchKernelGGL(radix_tree_kernel, dim3(gi), dim3(256), 0, stream, k1, (int)n, child, parent_inner, parent_leaf);
            hipLaunchKernelGGL(identity_order_kernel, dim3(gn), dim3(256), 0, stream, (int)n, sorted_prim, newpos, perm);
        } else {
            float *cb[2] = {(float *)(scratch + o_cb0), (float *)(scratch + o_cb1)};
            int *ci[2] = {(int *)(scratch + o_ci0), (int *)(scratch + o_ci1)};
            int *nn = (int *)(scratch + o_nn);
            auto *valid = (uint32_t *)(scratch + o_va), *vpos = (uint32_t *)(scratch + o_vp), *merger = (uint32_t *)(scratch + o_me), *mpos = (uint32_t *)(scratch + o_mp);
            hipLaunchKernelGGL(ploc_init_kernel, dim3(gn), dim3(256), 0, stream, d_boxes, sorted_prim, (int)n, cb[0], ci[0]);
            int c = (int)n, node_base = 0, cur = 0, pair_up = 0, radius = RTB_PLOC_RADIUS;
            if (const char *e = getenv("RT_KNOBS") ? getenv("RT_BVH_PLOC_RADIUS") : nullptr) { const int v = atoi(e); if (v >= 1 && v <= 256) radius = v; }
            for (int round = 0; c > 1; ++round) {
                if (round > 4096) { *err = "device BVH build: clustering does not converge"; return done(RT_E_HIP); }
                const unsigned gc = ((unsigned)c + 255u) / 256u;
                hipLaunchKernelGGL(ploc_nn_kernel, dim3(gc), dim3(256), 0, stream, cb[cur], c, pair_up, radius, nn);
                hipLaunchKernelGGL(ploc_flag_kernel, dim3(gc), dim3(256), 0, stream, nn, c, valid, merger);
                RTB_TRY2(hipMemsetAsync(valid + c, 0, 4, stream));    // the sums run over c + 1 flags: the last output is the total
                RTB_TRY2(hipMemsetAsync(merger + c, 0, 4, stream));
                RTB_TRY2(hipcub::DeviceScan::ExclusiveSum(scratch + o_tmp, scan_bytes, valid, vpos, c + 1, stream));
                RTB_TRY2(hipcub::DeviceScan::ExclusiveSum(scratch + o_tmp, scan_bytes, merger, mpos, c + 1, stream));
                uint32_t tot[2] = {0, 0};
                RTB_TRY2(hipMemcpyAsync(&tot[0], vpos + c, 4, hipMemcpyDeviceToHost, stream));
                RTB_TRY2(hipMemcpyAsync(&tot[1], mpos + c, 4, hipMemcpyDeviceToHost, stream));
                RTB_TRY2(hipStreamSynchronize(stream));
                if (tot[1] == 0u) {                                    // nobody chose each other (exact ties): pair neighbours up once
                    if (pair_up) { *err = "device BVH build: clustering made no progress"; return done(RT_E_HIP); }
                    pair_up = 1;
                    continue;
                }
                if ((int)tot[0] != c - (int)tot[1] || node_base + (int)tot[1] > ni) { *err = "device BVH build: inconsistent clustering"; return done(RT_E_HIP); }
                hipLaunchKernelGGL(ploc_merge_kernel, dim3(gc), dim3(256), 0, stream, cb[cur], ci[cur], nn, c, valid, vpos, merger, mpos, node_base, cb[cur ^ 1],
                                   ci[cur ^ 1], child, ibox, parent_inner, parent_leaf, leaf_count);
                cur ^= 1; c = (int)tot[0]; node_base += (int)tot[1]; pair_up = 0;
                out->rounds += 1;
            }
            if (node_base != ni) { *err = "device BVH build: clustering lost nodes"; return done(RT_E_HIP); }
            root = ni - 1;                               // the last merge
            RTB_TRY2(hipMemsetAsync(parent_inner + root, 0xff, 4, stream));   // -1
            hipLaunchKernelGGL(leaf_order_kernel, dim3(gn), dim3(256), 0, stream, (int)n, child, parent_inner, parent_leaf, leaf_count, sorted_prim, newpos, perm);
        }
        hipLaunchKernelGGL(depth_kernel, dim3(gi), dim3(256), 0, stream, ni, child, parent_inner, leaf_max, depth, is4, max_depth);
        RTB_TRY2(hipMemsetAsync(is4 + ni, 0, 4, stream));   // the scan runs over ni + 1 flags: its last output is the node count
        RTB_TRY2(hipcub::DeviceScan::ExclusiveSum(scratch + o_tmp, scan_bytes, is4, ix4, ni + 1, stream));
        uint32_t h[3] = {0, 0, 0};
        RTB_TRY2(hipMemcpyAsync(&h[2], ix4 + root, 4, hipMemcpyDeviceToHost, stream));
        RTB_TRY2(hipMemcpyAsync(&h[0], ix4 + ni, 4, hipMemcpyDeviceToHost, stream));
        RTB_TRY2(hipMemcpyAsync(&h[1], max_depth, 4, hipMemcpyDeviceToHost, stream));
        RTB_TRY2(hipStreamSynchronize(stream));
        const uint32_t n4 = h[0];
        if (n4 == 0 || n4 > (uint32_t)ni) { *err = "device BVH build: inconsistent node count"; return done(RT_E_HIP); }
        float *nodes = nullptr;
        RTB_TRY2(hipMalloc((void **)&nodes, (size_t)n4 * RT_NODE_FLOATS * 4));
        out->nodes = nodes;
        out->n_nodes = n4;
        if (h[1] >= 4096u) { *err = "device BVH build: implausible tree depth"; return done(RT_E_HIP); }
        for (uint32_t level = h[1] + 1u; !ploc && level-- > 0u;)   // (the clustering made its boxes as it went)
            hipLaunchKernelGGL(fit_level_kernel, dim3(gi), dim3(256), 0, stream, d_boxes, sorted_prim, ni, child, depth, level, ibox);
        hipLaunchKernelGGL(emit_kernel, dim3(gi), dim3(256), 0, stream, ni, child, d_boxes, sorted_prim, ibox, is4, ix4, newpos, leaf_max, nodes);
        out->root_ref = (int32_t)h[2];           // the root is at depth 0, hence a 4-wide node
        out->stack_depth = 3u * (h[1] / 2u + 1u) + 2u;   // a visit pushes at most three children
    }
    RTB_TRY2(hipGetLastError());
    RTB_TRY2(hipEventRecord(ev1, stream));
    RTB_TRY2(hipStreamSynchronize(stream));
    (void)hipEventElapsedTime(&out->build_ms, ev0, ev1);
    return done(RT_OK);
}

namespace {
__global__ void __launch_bounds__(256) gather_kernel(const float4 *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n, uint32_t quads, float4 *dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per float4
    if (i >= (size_t)n * quads) return;
    const uint32_t e = (uint32_t)(i / quads), k = (uint32_t)(i - (size_t)e * quads);
    dst[i] = src[(size_t)perm[e] * quads + k];
}
__global__ void __launch_bounds__(256) invert_kernel(const uint32_t *__restrict__ perm, uint32_t n, uint32_t *inv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[perm[i]] = i;
}
__global__ void __launch_bounds__(256) remap_kernel(uint32_t *list, uint32_t count, const uint32_t *__restrict__ inv, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && list[i] < n) list[i] = inv[list[i]];   // entries >= n are always-list slots: they do not move
}
}  // namespace

void gather_records(hipStream_t stream, const void *src, const uint32_t *perm, uint32_t n, uint32_t floats_per_record, void *dst) {
    const uint32_t quads = floats_per_record / 4u;
    const size_t total = (size_t)n * quads;
    if (total == 0) return;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const float4 *)src, perm, n, quads, (float4 *)dst);
}

void invert_permutation(hipStream_t stream, const uint32_t *perm, uint32_t n, uint32_t *inv) {
    if (n) hipLaunchKernelGGL(invert_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, perm, n, inv);
}

void remap_slot_list(hipStream_t stream, uint32_t *list, uint32_t count, const uint32_t *inv, uint32_t n) {
    if (count) hipLaunchKernelGGL(remap_kernel, dim3((count + 255u) / 256u), dim3(256), 0, stream, list, count, inv, n);
}

}  // namespace rtb
