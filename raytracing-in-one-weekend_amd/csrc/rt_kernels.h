// rt_kernels.h — device functions of the path tracer (gfx950): primitive tests, the BVH walk, textures, pdfs,
// one bounce of ray_color, camera rays, and the small stand-alone kernels (parity hook, resolve).
//
// The render itself is the streaming pipeline of rt_wavefront.h (generate / traverse / shade / fold over a pool of
// in-flight paths); this file holds what those kernels call.  `ray_color` (camera.rs:253-309) is evaluated
// front-to-back: the recursion  L = e0 + w0 * (e1 + w1 * (...))  becomes a running throughput, and because every
// reference material either scatters with emitted() = black or emits and ends the path, radiance is only ever added
// by the bounce that ends a path.  Everything that decides WHERE a path goes (hit tests, scatter directions, pdfs,
// random draws) follows the reference operation by operation — single IEEE f32 operations, no contraction — so a
// path is bit-identical to the CPU oracle's; only the radiance products are associated differently.
//
// The closest-hit search is the part that is NOT the reference's algorithm: a 4-wide BVH over ALL leaf primitives
// (64-byte nodes holding the four child boxes quantised to 8 bits on the node's own grid, 64-byte primitive slots),
// walked with a per-lane stack of node refs in LDS (overflow in HBM).  It reproduces HittableList::hit semantics
// (hittable_list.rs:39-50): closest t, and on an exact tie the object later in the list — the larger primitive id —
// wins.  The only tests decided without the reference's exact arithmetic are the walk's box tests, which are conservative
// (a box is widened by its quantisation and a rounding margin): whatever is accepted as a hit went through the exact test.
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_api.h"
#include "rt_portable_math.h"
#include "rt_scene_graph.h"
#include "rt_seed_schedule.h"

#pragma clang fp contract(off)

namespace rtk {

#define RT_DEV __device__ __forceinline__
#define RT_MAX_MEDIA 4
#define RT_F32_MAX 3.40282347e+38f
// traversal: run a leaf step once RT_LEAF_MIN lanes stand on a leaf (or nobody is on an inner node)
#ifndef RT_LEAF_MIN
#define RT_LEAF_MIN 12
#endif

struct V3 {
    float x, y, z;
};
RT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
RT_DEV V3 operator*(V3 a, float k) { return {a.x * k, a.y * k, a.z * k}; }
RT_DEV V3 operator*(float k, V3 a) { return {a.x * k, a.y * k, a.z * k}; }
RT_DEV V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV V3 operator/(V3 a, float k) { return {a.x / k, a.y / k, a.z / k}; }
RT_DEV float length_squared(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
RT_DEV float length(V3 v) { return sqrtf(length_squared(v)); }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RT_DEV V3 normalize(V3 v) {  // math/src/vec3.rs:690-700
    float lensq = length_squared(v);
    if (lensq == 0.0f) return v3(0.0f, 0.0f, 0.0f);
    return v * (1.0f / sqrtf(lensq));
}
RT_DEV V3 reflect_unit_vector(V3 v, V3 n) {  // vec3.rs:740-753
    float two = 2.0f;
    return v - two * dot(v, n) * n;
}
RT_DEV V3 refract(V3 uv, V3 n, float eta) {  // vec3.rs:756-771
    float cos_theta = dot(-uv, n);
    V3 r_par = eta * (uv + cos_theta * n);
    V3 r_perp = -sqrtf(1.0f - length_squared(r_par)) * n;
    return r_par + r_perp;
}
RT_DEV float f32_min(float a, float b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }
RT_DEV float f32_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
RT_DEV int f2i(float f) { return __float_as_int(f); }

struct Ray {
    V3 o, d;
    float time;
    RT_DEV V3 at(float t) const { return o + t * d; }
};

struct DScene {
    const float4 *nodes;
    const float4 *slots;
    const float4 *attrs;
    const float4 *xforms;     // 2 x float4 per record
    const float4 *matrices;   // 9 x float4 per general Transform: world2object, obj2world, normal2world (3 rows each)
    const float4 *media;
    const float4 *lights;
    const float4 *materials;
    const float4 *textures;
    const uint32_t *perlin;
    const uint32_t *perlin_lds;   // Perlin table 0 staged in LDS by the shade kernel (null elsewhere)
    uint32_t noise_coop;          // a wave with at most this many noise-texture hits evaluates them with all its lanes (coop_turbulence); 0 = never
    const uint32_t *texels;
    const uint32_t *moving;
    const uint32_t *rects;
    const float2 *samples;
    const float2 *disk;
    const uint32_t *shuffled;
    int32_t root_ref;
    uint32_t n_slots, always_begin, always_count, n_media, n_lights, n_moving, n_rects, sets, spp;
    // The top of the tree in breadth-first order, for the walk that keeps it in LDS (wf_traverse_kernel, TopLds): node index of
    // the k-th node from the root, and for each of its four children that child's own place in that order (or ~0u: a leaf, no
    // child, or further down than the table goes).  n_top entries (0: no table — a tree built on the device).
    const uint32_t *top_order;
    const uint32_t *top_child;
    uint32_t n_top;
    // rays that start further than far_limit (on some axis) from far_centre walk the tree with every box grown (ray_fat):
    // far_reach = how far the centre of a sphere of ordinary size lies from far_centre at most, far_rmin = the smallest sphere radius
    float far_limit, far_centre[3], far_reach, far_rmin;
};

#define RT_N_REGIONS 24
#define RT_STAT_WORDS (26 + RT_N_PRIM_TYPES)   // counters of one render (flush_counters); the region timers follow
struct Counters {  // only live in the STATS instantiation
    uint32_t rays, node_visits, prim_tests[RT_N_PRIM_TYPES], medium_tests, light_pdf_evals, shades[5], tex_evals[4], accepted;
    uint32_t phase_lanes[5], phase_slots[5];
    uint32_t box_tests;              // child boxes tested by node4_visit (the non-empty children of every visited node)
    uint32_t region[RT_N_REGIONS];   // cycles a wave spent in a code region (counted on its first active lane)
};
// Region timers (development builds with -DRT_REGION_TIMERS=1 only; they time the product kernels, not the
// counting variant, whose counters distort it): RG_BEGIN/RG_END bracket a piece of code, also inside
// divergent control flow — the scalar cycle counter is read once per wave execution of the region.
#ifndef RT_REGION_TIMERS
#define RT_REGION_TIMERS 0
#endif
constexpr bool kTimers = RT_REGION_TIMERS != 0;
enum { RG_LOAD = 0, RG_FINISH, RG_RECORD, RG_TEXTURE, RG_EMIT, RG_METAL, RG_DIEL, RG_ISO, RG_LAMBERT, RG_L_LIGHTS, RG_L_COSINE,
       RG_L_PDF, RG_B_MEDIA, RG_B_RECTS, RG_B_ALWAYS, RG_B_REST, RG_STORE, RG_NODE, RG_LEAF, RG_FETCH, RG_CAMERA, RG_SHADE_ALL, RG_BEGIN_ALL,
       RG_TRAV_ALL };
#define RG_BEGIN(k) unsigned long long rg_t_##k = (kTimers && !STATS) ? __builtin_readcyclecounter() : 0ull
#define RG_END(k)                                                                                              \
    do {                                                                                                       \
        if (kTimers && !STATS) {                                                                               \
            const unsigned long long rg_m = __ballot(1);                                                       \
            if ((rg_m & (0ull - rg_m)) == (1ull << (threadIdx.x & 63u)))                                       \
                cnt.region[k] += (uint32_t)(__builtin_readcyclecounter() - rg_t_##k);                          \
        }                                                                                                      \
    } while (0)
enum { PH_NODE = 0, PH_LEAF = 1, PH_SHADE = 2, PH_START = 3, PH_BEGIN = 4 };

struct Hit {      // what the search tracks
    float t;
    int32_t id;   // primitive id (tie-break key); -1 = nothing
    int32_t slot; // slot index, or -(medium index + 1)
};

struct Rec {      // HitRecord (hittable.rs:6-14)
    V3 p, normal;
    float t, u, v;
    bool front_face;
    int32_t material;
};

/* ------------------------------------------------------------------ primitive tests */

// Mat4 x Vec4 (math/src/mat4.rs:434-441), rows 0..2: a*x + b*y + c*z + d*w, summed left to right (w is 0 or 1)
RT_DEV float mat_row(float4 m, V3 v, float w) { return m.x * v.x + m.y * v.y + m.z * v.z + m.w * w; }
RT_DEV V3 mat_apply(const float4 *m, V3 v, float w) { return v3(mat_row(m[0], v, w), mat_row(m[1], v, w), mat_row(m[2], v, w)); }

// Ray into the frame of a wrapped primitive: Translate (transform.rs:34) then RotateY (:117-126), or a general
// Transform (transform.rs:190 + math/src/ray.rs:42-54).
//
// EXT = what a kernel instance has code for (a bit mask).  The shade / generate / tail kernels inline every primitive,
// material and texture the backend covers: 90 KB of code and 90-130 VGPRs for a Cornell box that needs a sixth of it, with an
// instruction cache of 64 KB per two CUs.  So they are instantiated for a few feature sets and rt_scene_upload picks the
// smallest one that covers the scene (rt_device.hip: kernel_set).  Code for a feature that is compiled out is never
// reached by a scene without it, so which instance runs changes no bit.  Quadrics are never reached through the BVH
// (rt_api.h); the traversal kernel therefore only distinguishes "general Transform inside the tree" or not.
enum { EXT_NONE = 0, EXT_XFORM = 1 /* general Transform */, EXT_QUADRICS = 2, EXT_NOISE = 4 /* Perlin noise texture */,
       EXT_MEDIA = 8 /* ConstantMedium + Isotropic */, EXT_TRI = 16 /* triangle meshes */, EXT_IMAGE = 32 /* image textures (u, v) */,
       EXT_CHAIN = 64 /* chains of xform records: wrappers nested in any order */, EXT_ALL = 127 };
template <int EXT = EXT_ALL>
RT_DEV Ray to_object_one(const DScene &sc, float4 a, float4 b, const Ray &r) {
    int flags = f2i(a.x);
    float sin_t = a.y, cos_t = a.z;
    Ray q = r;
    if ((EXT & EXT_XFORM) && (flags & 4)) {
        const float4 *m = sc.matrices + 9 * (size_t)f2i(a.y);
        q.o = mat_apply(m, r.o, 1.0f);
        q.d = mat_apply(m, r.d, 0.0f);
        return q;
    }
    if (flags & 2) q.o = r.o - v3(a.w, b.x, b.y);
    if (flags & 1) {
        V3 o = q.o, d = q.d;
        q.o.x = cos_t * o.x - sin_t * o.z;
        q.o.z = sin_t * o.x + cos_t * o.z;
        q.d.x = cos_t * d.x - sin_t * d.z;
        q.d.z = sin_t * d.x + cos_t * d.z;
    }
    return q;
}
// A chain record (flags 8: first, count) stands for `count` consecutive records of sc.xforms, outermost wrapper first: any
// nesting of Translate / RotateY / Transform (each wrapper's hit() transforms the ray it is given and hands it to its child).
// Only kernels with EXT_CHAIN carry the loops (measured: -3 % on GeometricPrimitives and Blocks if the full set carried
// them, profiles/r2/ab_chain_records.txt); rt_scene_upload asks for that set when a slot has such a record.
template <int EXT = EXT_ALL>
RT_DEV Ray to_object_rec(const DScene &sc, float4 a, float4 b, const Ray &r) {
    if ((EXT & EXT_CHAIN) && (f2i(a.x) & 8)) {
        const int first = f2i(a.y), count = f2i(a.z);
        Ray q = r;
        for (int k = 0; k < count; ++k) q = to_object_one<EXT>(sc, sc.xforms[2 * (first + k)], sc.xforms[2 * (first + k) + 1], q);
        return q;
    }
    return to_object_one<EXT>(sc, a, b, r);
}
template <int EXT = EXT_ALL>
RT_DEV Ray to_object(const DScene &sc, int xf, const Ray &r) { return to_object_rec<EXT>(sc, sc.xforms[2 * xf], sc.xforms[2 * xf + 1], r); }

// The way out of one record's wrappers for a hit record (p, face-corrected normal, front_face): Transform
// (transform.rs:195-201), RotateY (:128-143), Translate (:35-44).  `d_in` = direction of the ray these wrappers were given.
template <int EXT = EXT_ALL>
RT_DEV void record_out_one(const DScene &sc, float4 xa, float4 xb, V3 d_in, V3 &p, V3 &nn, bool &front) {
    int flags = f2i(xa.x);
    if ((EXT & EXT_XFORM) && (flags & 4)) {   // p and normal to the outer space, front_face / t / u / v kept
        const float4 *m = sc.matrices + 9 * (size_t)f2i(xa.y);
        p = mat_apply(m + 3, p, 1.0f);
        nn = normalize(mat_apply(m + 6, nn, 0.0f));
    }
    if (flags & 1) {
        float sin_t = xa.y, cos_t = xa.z;
        V3 p0 = p, n0 = nn;
        p.x = cos_t * p0.x + sin_t * p0.z;
        p.z = -sin_t * p0.x + cos_t * p0.z;
        nn.x = cos_t * n0.x + sin_t * n0.z;
        nn.z = -sin_t * n0.x + cos_t * n0.z;
    }
    if (flags & 2) {
        p = p + v3(xa.w, xb.x, xb.y);
        front = dot(d_in, nn) < 0.0f;  // transform.rs:36: HitRecord::new again (direction is unchanged by Translate)
        nn = front ? nn : -nn;
    }
}

// ... of a whole record: a chain's wrappers innermost first, each one seeing the ray its own wrappers were given.
template <int EXT = EXT_ALL>
RT_DEV void record_out(const DScene &sc, float4 xa, float4 xb, const Ray &r_world, V3 &p, V3 &nn, bool &front) {
    if ((EXT & EXT_CHAIN) && (f2i(xa.x) & 8)) {
        const int first = f2i(xa.y), count = f2i(xa.z);
        for (int k = count - 1; k >= 0; --k) {
            Ray q = r_world;
            for (int j = 0; j < k; ++j) q = to_object_one<EXT>(sc, sc.xforms[2 * (first + j)], sc.xforms[2 * (first + j) + 1], q);
            record_out_one<EXT>(sc, sc.xforms[2 * (first + k)], sc.xforms[2 * (first + k) + 1], q.d, p, nn, front);
        }
    } else
        record_out_one<EXT>(sc, xa, xb, r_world.d, p, nn, front);
}

// A slot as the kernels read it: all 64 bytes at once, so that the loads of a test do not wait on
// each other.  In the DEVICE copy of the slot array, spheres and rectangles under a transform carry
// that transform's record in words 8..15 (unused by those types; written by rt_scene_upload), so
// their test needs no second, dependent load; moving spheres and triangles go through sc.xforms.
struct SlotRec {
    float4 q0, meta, q2, q3;
};
RT_DEV SlotRec load_slot(const DScene &sc, uint32_t slot) {
    const float4 *p = sc.slots + 4 * (size_t)slot;
    SlotRec s;
    s.q0 = p[0]; s.meta = p[1]; s.q2 = p[2]; s.q3 = p[3];
    return s;
}
// load_slot() for a slot index that is the same in every lane of the wave (the always-list scan): through a pointer in the
// constant address space the four quads are ONE scalar load (s_load_dwordx16 into SGPRs, through the scalar cache) instead of four
// vector loads of one address by 64 lanes — no vector-memory instruction, no trip through the texture data path, and the slot's
// 16 words cost no vector registers while the next one is being fetched.  The slot array is read-only for every kernel.
// Measured (profiles/r3/ab_always_scalar_loads.txt): Blocks +3.6 %, GeometricPrimitives +2.7 %, the Cornell box with smoke +1.6 %,
// but the plain Cornell box -1.8 % (its small kernel set has the registers to fetch a slot ahead as a vector load, and a scalar
// load is waited for in order): so the small set keeps the vector loads.
#ifndef RT_ALWAYS_SCALAR
#define RT_ALWAYS_SCALAR 1
#endif
typedef float f32x4c __attribute__((ext_vector_type(4)));
template <int EXT>
RT_DEV SlotRec load_slot_uniform(const DScene &sc, uint32_t slot) {
    if ((EXT & EXT_ALL) == EXT_NONE) return load_slot(sc, slot);
#if RT_ALWAYS_SCALAR
    typedef const __attribute__((address_space(4))) f32x4c *ConstQuad;
    const ConstQuad p = (ConstQuad)(uintptr_t)(sc.slots + 4 * (size_t)slot);
    const f32x4c a = p[0], b = p[1], c = p[2], d = p[3];
    SlotRec s;
    s.q0 = make_float4(a.x, a.y, a.z, a.w); s.meta = make_float4(b.x, b.y, b.z, b.w);
    s.q2 = make_float4(c.x, c.y, c.z, c.w); s.q3 = make_float4(d.x, d.y, d.z, d.w);
    return s;
#else
    return load_slot(sc, slot);
#endif
}
__host__ __device__ inline bool type_has_inline_xform(int type) { return type != RT_PRIM_MOVING_SPHERE && type != RT_PRIM_TRIANGLE && type != RT_PRIM_HYPERBOLOID; }

RT_DEV bool sphere_roots(V3 center, float radius, const Ray &r, float t_min, float t_max, float *t_out) {  // sphere.rs:45-66
    V3 oc = r.o - center;
    float a = length_squared(r.d);
    float half_b = dot(oc, r.d);
    float c = length_squared(oc) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return false;
    float sqrtd = sqrtf(disc);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    *t_out = root;
    return true;
}

RT_DEV V3 moving_center(float4 q0, float time0, float4 q2, float time) {  // sphere.rs:144-147
    V3 c0 = v3(q0.x, q0.y, q0.z), c1 = v3(q2.x, q2.y, q2.z);
    return c0 + ((time - time0) / (q2.w - time0)) * (c1 - c0);
}

RT_DEV float comp(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// rectangles.rs:22-34 / 92-104 / 162-174.  type: 2 XY (A=x,B=y,AX=z) 3 XZ (x,z,y) 4 YZ (y,z,x)
template <int AX, int A, int B>
RT_DEV bool rect_test_axes(float4 q0, float k, const Ray &r, float t_min, float t_max, float *t_out, float *a_out, float *b_out) {
    float t = (k - comp(r.o, AX)) / comp(r.d, AX);
    if (t < t_min || t > t_max) return false;
    float a = comp(r.o, A) + t * comp(r.d, A);
    float b = comp(r.o, B) + t * comp(r.d, B);
    if (a < q0.x || a > q0.y || b < q0.z || b > q0.w) return false;
    *t_out = t; *a_out = a; *b_out = b;
    return true;
}
// (one instantiation per kind: the kind is wave-uniform in the always-list, so the axes cost a scalar branch instead of selects)
RT_DEV bool rect_test(int type, float4 q0, float k, const Ray &r, float t_min, float t_max, float *t_out, float *a_out, float *b_out) {
    if (type == RT_PRIM_XY_RECT) return rect_test_axes<2, 0, 1>(q0, k, r, t_min, t_max, t_out, a_out, b_out);
    if (type == RT_PRIM_XZ_RECT) return rect_test_axes<1, 0, 2>(q0, k, r, t_min, t_max, t_out, a_out, b_out);
    return rect_test_axes<0, 1, 2>(q0, k, r, t_min, t_max, t_out, a_out, b_out);
}

struct TriHit {
    float t, b0, b1, b2;
    V3 n;
};

// triangle_mesh.rs:185-234 (the back-face cull needs the interpolated normal, so attrs are read here)
RT_DEV bool tri_test(const DScene &sc, uint32_t slot, float4 q0, float p3z, float4 q2, const Ray &r, float t_min, float t_max, TriHit *h) {
    V3 p1 = v3(q0.x, q0.y, q0.z), p2 = v3(q0.w, q2.x, q2.y), p3 = v3(q2.z, q2.w, p3z);
    V3 e1 = p2 - p1, e2 = p3 - p1;
    V3 s1 = cross(r.d, e2);
    float div = dot(s1, e1);
    if (div == 0.0f) return false;
    float inv_div = 1.0f / div;
    V3 d = r.o - p1;
    float b1 = dot(d, s1) * inv_div;
    if (b1 < 0.0f || b1 > 1.0f) return false;
    V3 s2 = cross(d, e1);
    float b2 = dot(r.d, s2) * inv_div;
    if (b2 < 0.0f || (b1 + b2) > 1.0f) return false;
    float t = dot(e2, s2) * inv_div;
    if (t < t_min || t > t_max) return false;
    float b0 = 1.0f - b1 - b2;
    float4 a0 = sc.attrs[4 * (size_t)slot], a1 = sc.attrs[4 * (size_t)slot + 1], a2 = sc.attrs[4 * (size_t)slot + 2];
    V3 n1 = v3(a0.x, a0.y, a0.z), n2 = v3(a0.w, a1.x, a1.y), n3 = v3(a1.z, a1.w, a2.x);
    V3 n = normalize(b0 * n1 + b1 * n2 + b2 * n3);
    if (dot(r.d, n) > 0.0f) return false;
    h->t = t; h->b0 = b0; h->b1 = b1; h->b2 = b2; h->n = n;
    return true;
}

/* ---- quadrics (cylinder.rs, cone.rs, paraboloid.rs, hyperboloid.rs; "Physically based rendering" 3.4-3.5) ---- */

// math/src/polynomial.rs:11-39.  roots start as the caller's [0, 0]; the linear case writes one of them.
RT_DEV uint32_t poly_quadratic(float a, float b, float c, float *r0, float *r1) {
    *r0 = 0.0f; *r1 = 0.0f;
    if (a == 0.0f) {                       // poly_linear(b, c)
        if (b == 0.0f) return 0u;
        *r0 = -c / b;
        return 1u;
    }
    float delta = b * b - 4.0f * a * c;
    if (delta == 0.0f) {                   // sic: (-b / 2) * a
        *r0 = -b / 2.0f * a;
        *r1 = -b / 2.0f * a;
        return 2u;
    }
    if (!(delta < 0.0f)) {   // is_sign_positive; a NaN delta counts as positive whatever its sign bit
        float sgn = b != b ? b : copysignf(1.0f, b);   // f32::signum
        float q = -(b + sgn * sqrtf(delta)) / 2.0f;
        float x0 = q / a, x1 = c / q;
        if (x0 > x1) { float t = x0; x0 = x1; x1 = t; }
        *r0 = x0; *r1 = x1;
        return 2u;
    }
    return 0u;
}

struct QuadHit {
    float t, phi, v;
    V3 p;
};

// The hit tests of the four quadrics share their shape: roots, pick one, clip against z / phi, try the other.
// Cone, paraboloid and hyperboloid ignore t_min and accept any t > 0 (cone.rs:61-65, paraboloid.rs:78-88).
RT_DEV bool quadric_test(int type, const SlotRec &s, const Ray &r, float t_min, float t_max, QuadHit *h) {
    const float two_pi = 3.14159265358979323846f * 2.0f;
    const V3 o = r.o, d = r.d;
    const float4 q0 = s.q0;
    float a, b, c, zlo, zhi, phi_max = q0.w;
    V3 p1 = v3(0.0f, 0.0f, 0.0f), p2 = p1;
    if (type == RT_PRIM_CYLINDER) {            // cylinder.rs:55-57
        a = d.x * d.x + d.y * d.y;
        b = 2.0f * (d.x * o.x + d.y * o.y);
        c = o.x * o.x + o.y * o.y - q0.x * q0.x;
        zlo = q0.y; zhi = q0.z;
    } else if (type == RT_PRIM_CONE) {         // cone.rs:43-52
        float k = q0.x / q0.y;
        k = k * k;
        a = d.x * d.x + d.y * d.y - k * d.z * d.z;
        b = 2.0f * (d.x * o.x + d.y * o.y - k * d.z * (o.z - q0.y));
        c = o.x * o.x + o.y * o.y - k * (o.z - q0.y) * (o.z - q0.y);
        zlo = 0.0f; zhi = q0.y;
    } else if (type == RT_PRIM_PARABOLOID) {   // paraboloid.rs:59-63
        float k = q0.z / (q0.x * q0.x);
        a = k * (d.x * d.x + d.y * d.y);
        b = 2.0f * k * (d.x * o.x + d.y * o.y) - d.z;
        c = k * (o.x * o.x + o.y * o.y) - o.z;
        zlo = q0.y; zhi = q0.z;
    } else {                                   // hyperboloid.rs:84-86
        float ah = s.q3.x, ch = s.q3.y;
        a = ah * d.x * d.x + ah * d.y * d.y - ch * d.z * d.z;
        b = 2.0f * (ah * d.x * o.x + ah * d.y * o.y - ch * d.z * o.z);
        c = ah * o.x * o.x + ah * o.y * o.y - ch * o.z * o.z - 1.0f;
        zlo = s.meta.w; zhi = s.q2.w;
        p1 = v3(q0.x, q0.y, q0.z); p2 = v3(s.q2.x, s.q2.y, s.q2.z);
    }
    float t0, t1;
    if (poly_quadratic(a, b, c, &t0, &t1) == 0u) return false;
    float thit;
    if (type == RT_PRIM_CYLINDER) {            // cylinder.rs:68-76
        if (t0 > t_max || t1 < t_min) return false;
        thit = t0 < t_min ? t1 : t0;
        if (thit > t_max) return false;
    } else {                                   // cone.rs:61-68 = paraboloid.rs:78-88 = hyperboloid.rs:97-107
        if (t0 > t_max || t1 <= 0.0f) return false;
        thit = t0 <= 0.0f ? t1 : t0;
        if (thit > t_max) return false;
    }
    V3 p;
    float phi, v = 0.0f;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        p = r.at(thit);
        if (type == RT_PRIM_HYPERBOLOID) {     // hyperboloid.rs:111-117, 133-139
            v = (p.z - p1.z) / (p2.z - p1.z);
            V3 pr = (1.0f - v) * p1 + v * p2;
            phi = pm::atan2f_(pr.x * p.y - p.x * pr.y, p.x * pr.x + p.y * pr.y);
            if (phi < 0.0f) phi += pass == 0 ? 2.0f : two_pi;   // sic: the first attempt adds 2, not 2 pi (:116)
        } else {
            phi = pm::atan2f_(p.y, p.x);
            if (phi < 0.0f) phi = phi + two_pi;
        }
        if (!(p.z < zlo || p.z > zhi || phi > phi_max)) break;   // inside the clipping parameters
        if (pass == 1 || thit == t1) return false;
        thit = t1;
        if (t1 > t_max) return false;
    }
    h->t = thit; h->phi = phi; h->v = v; h->p = p;
    return true;
}

// u, v and the outward normal of an accepted quadric hit (cylinder.rs:112-121, cone.rs:104-114,
// paraboloid.rs:123-137, hyperboloid.rs:148-162)
RT_DEV V3 quadric_record(int type, const SlotRec &s, const QuadHit &h, float *u, float *v) {
    const float4 q0 = s.q0;
    const float phi_max = q0.w;
    const V3 p = h.p;
    *u = h.phi / phi_max;
    V3 dpdu = v3(-phi_max * p.y, phi_max * p.x, 0.0f), dpdv;
    if (type == RT_PRIM_CYLINDER) {
        *v = (p.z - q0.y) / (q0.z - q0.y);
        dpdv = v3(0.0f, 0.0f, q0.z - q0.y);
        return normalize(cross(dpdv, dpdu));
    }
    if (type == RT_PRIM_CONE) {
        *v = p.z / q0.y;
        dpdu = v3(-phi_max * p.y, phi_max * p.y, 0.0f);   // sic: p.y twice (cone.rs:109)
        dpdv = v3(-p.x / (1.0f - *v), -p.y / (1.0f - *v), q0.y);
    } else if (type == RT_PRIM_PARABOLOID) {
        *v = (p.z - q0.y) / (q0.z - q0.y);
        dpdv = (q0.z - q0.y) * v3(p.x / (2.0f * p.z), p.y / (2.0f * p.z), 1.0f);
    } else {
        *v = h.v;
        const V3 p1 = v3(q0.x, q0.y, q0.z), p2 = v3(s.q2.x, s.q2.y, s.q2.z);
        float sin_phi, cos_phi;
        pm::sincosf_(h.phi, &sin_phi, &cos_phi);
        dpdv = v3((p2.x - p1.x) * cos_phi - (p2.y - p1.y) * sin_phi, (p2.x - p1.x) * sin_phi + (p2.y - p1.y) * cos_phi, p2.z - p1.z);
    }
    return normalize(cross(dpdu, dpdv));
}

// Hit test of one slot: true + t when `obj.hit(r, t_min, t_max)` of that leaf would return Some.
// EXT: see to_object_rec.
template <bool STATS, int EXT = EXT_ALL>
RT_DEV bool slot_hit_rec(const DScene &sc, uint32_t slot, const SlotRec &s, const Ray &r_world, float t_min, float t_max, float *t_out,
                         int32_t *id_out, Counters &cnt, bool have_object, const Ray r_object /* the ray already in the slot's frame, if have_object */) {
    const float4 q0 = s.q0, meta = s.meta;
    int w4 = f2i(meta.x);
    int type = w4 & 15;
    int xf = (w4 >> 8) - 1;
    *id_out = f2i(meta.z);
    if (STATS) cnt.prim_tests[type] += 1;
    Ray r = r_world;
    if (have_object) r = r_object;
    else if (xf >= 0) r = type_has_inline_xform(type) ? to_object_rec<EXT>(sc, s.q2, s.q3, r_world) : to_object<EXT>(sc, xf, r_world);
    if ((EXT & EXT_QUADRICS) && type >= RT_PRIM_CYLINDER) {
        QuadHit qh;
        if (!quadric_test(type, s, r, t_min, t_max, &qh)) return false;
        *t_out = qh.t;
        return true;
    }
    float a, b;
    switch (type) {
    case RT_PRIM_SPHERE: case RT_PRIM_MOVING_SPHERE: {   // one sphere_roots for both kinds: only the centre differs
        V3 c = v3(q0.x, q0.y, q0.z);
        if (type == RT_PRIM_MOVING_SPHERE) c = moving_center(q0, meta.w, s.q2, r.time);
        return sphere_roots(c, q0.w, r, t_min, t_max, t_out);
    }
    case RT_PRIM_XY_RECT: case RT_PRIM_XZ_RECT: case RT_PRIM_YZ_RECT:
        return rect_test(type, q0, meta.w, r, t_min, t_max, t_out, &a, &b);
    default: {
        if (!(EXT & EXT_TRI)) return false;
        TriHit th;
        if (!tri_test(sc, slot, q0, meta.w, s.q2, r, t_min, t_max, &th)) return false;
        *t_out = th.t;
        return true;
    }
    }
}
template <bool STATS, int EXT = EXT_ALL>
RT_DEV bool slot_hit(const DScene &sc, uint32_t slot, const Ray &r_world, float t_min, float t_max, float *t_out, int32_t *id_out,
                     Counters &cnt) {
    return slot_hit_rec<STATS, EXT>(sc, slot, load_slot(sc, slot), r_world, t_min, t_max, t_out, id_out, cnt, false, r_world);
}

RT_DEV void sphere_uv(V3 p, float *u, float *v) {  // sphere.rs:25-41
    float theta = pm::acosf_(-p.y);
    float phi = pm::atan2f_(-p.z, p.x) + 3.14159265358979323846f;
    *u = phi / (float)(2.0 * 3.14159265358979323846);
    *v = theta / 3.14159265358979323846f;
}

// Full HitRecord of a slot for a known t: the same arithmetic as the leaf's hit() followed by
// the wrappers on the way out (RotateY transform.rs:128-143, Translate :35-44, Transform :192-203, FlipFace).
// t_min / t_max: the bounds the accepted test ran with (a quadric repeats its test to know which root it took).
template <int EXT = EXT_ALL>
RT_DEV Rec slot_record(const DScene &sc, uint32_t slot, const Ray &r_world, float t, bool want_uv, float t_min, float t_max) {
    const SlotRec s = load_slot(sc, slot);
    float4 q0 = s.q0, meta = s.meta;
    int w4 = f2i(meta.x);
    int type = w4 & 15;
    int xf = (w4 >> 8) - 1;
    float4 xa = make_float4(0.0f, 0.0f, 0.0f, 0.0f), xb = xa;
    if (xf >= 0) {
        if (type_has_inline_xform(type)) { xa = s.q2; xb = s.q3; }
        else { xa = sc.xforms[2 * xf]; xb = sc.xforms[2 * xf + 1]; }
    }
    Ray r = xf >= 0 ? to_object_rec<EXT>(sc, xa, xb, r_world) : r_world;
    Rec rec;
    rec.t = t;
    rec.material = f2i(meta.y);
    rec.u = 0.0f; rec.v = 0.0f;
    V3 p = r.at(t), n;
    if ((EXT & EXT_QUADRICS) && type >= RT_PRIM_CYLINDER) {
        QuadHit qh;
        quadric_test(type, s, r, t_min, t_max, &qh);   // same arithmetic as the accepted test
        p = qh.p;
        n = quadric_record(type, s, qh, &rec.u, &rec.v);
    } else
    switch (type) {
    case RT_PRIM_SPHERE: case RT_PRIM_MOVING_SPHERE: {
        V3 c = v3(q0.x, q0.y, q0.z);
        if (type == RT_PRIM_MOVING_SPHERE) c = moving_center(q0, meta.w, s.q2, r.time);
        n = (p - c) / q0.w;
        if ((EXT & EXT_IMAGE) && want_uv) sphere_uv(n, &rec.u, &rec.v);
        break;
    }
    case RT_PRIM_XY_RECT: case RT_PRIM_XZ_RECT: case RT_PRIM_YZ_RECT: {
        int AX = type == RT_PRIM_XY_RECT ? 2 : (type == RT_PRIM_XZ_RECT ? 1 : 0);
        int A = type == RT_PRIM_YZ_RECT ? 1 : 0;
        int B = type == RT_PRIM_XY_RECT ? 1 : 2;
        float a = comp(r.o, A) + t * comp(r.d, A);
        float b = comp(r.o, B) + t * comp(r.d, B);
        rec.u = (a - q0.x) / (q0.y - q0.x);
        rec.v = (b - q0.z) / (q0.w - q0.z);
        n = v3(AX == 0 ? 1.0f : 0.0f, AX == 1 ? 1.0f : 0.0f, AX == 2 ? 1.0f : 0.0f);
        break;
    }
    default: {
        if (!(EXT & EXT_TRI)) { n = v3(0.0f, 0.0f, 0.0f); break; }
        TriHit th;
        tri_test(sc, slot, q0, meta.w, s.q2, r, -RT_F32_MAX, RT_F32_MAX, &th);  // same arithmetic as the accepted test
        n = th.n;
        float4 a2 = sc.attrs[4 * (size_t)slot + 2], a3 = sc.attrs[4 * (size_t)slot + 3];
        rec.u = th.b0 * a2.y + th.b1 * a2.w + th.b2 * a3.y;
        rec.v = th.b0 * a2.z + th.b1 * a3.x + th.b2 * a3.z;
        break;
    }
    }
    bool front = dot(r.d, n) < 0.0f;  // hittable.rs:26
    V3 nn = front ? n : -n;
    if (xf >= 0) record_out<EXT>(sc, xa, xb, r_world, p, nn, front);
    if (w4 & 16) front = !front;  // flip_face.rs:24-27
    rec.p = p; rec.normal = nn; rec.front_face = front;
    return rec;
}

/* ------------------------------------------------------------------ closest hit */

// HittableList::hit fold (hittable_list.rs:43-49) for candidates that may arrive in any order:
// smaller t wins, equal t -> later object (larger primitive id).  A NaN t never wins here; rays
// that can produce NaN go through closest_hit_exhaustive().
RT_DEV void consider(Hit &best, float t, int32_t id, int32_t slot) {
    if (t < best.t || (t == best.t && id > best.id)) { best.t = t; best.id = id; best.slot = slot; }
}

// Rays from far away.  sphere.rs:46-66 in f32 loses the discriminant to cancellation when the ray starts thousands of radii from
// the sphere: half_b^2 - a*c is the difference of two numbers of size |oc|^2 |d|^2, and the roundings on the way to it (oc, two
// three-term dot products and a norm without contraction, the squares and the product: u = 2^-24 each) add up to less than
// 15 u |oc|^2 |d|^2 + 2 u r^2 |d|^2 < eps |oc|^2 |d|^2 with eps = 2^-20.  The true value is |d|^2 (r^2 - dist^2), dist = the
// distance of the centre from the ray's line, so the test can accept a sphere the ray passes at up to sqrt(r^2 + eps |oc|^2)
// from its centre, and the root it then reports lies within sqrt(eps) |oc| of the point of closest approach along the ray —
// nowhere near the sphere.  (The length of d does not matter: it scales everything alike.)  The reference's flat `world` takes
// such a hit (hittable_list.rs:39-50); a walk that culls by the sphere's box never gets to test it.  So a ray that starts
// further than far_limit (2^10 x the smallest sphere radius, rt_api.h) from the middle of the scene's spheres walks the tree
// with every box grown by
//      sqrt(r_min^2 + eps D^2) - r_min  +  sqrt(eps) D,      D = |o - middle| + far_reach >= |oc| for every sphere of ordinary size,
// the most the ray can miss a sphere by and the most the reported root can lie beyond that: whatever its exact test can
// accept, with whatever t, it reaches.  (Spheres of more than 64 smallest radii — the r = 1000 ground of RandomWorld — are left
// to the padding of their boxes, 1e-4 of their coordinates: the band in which they can be mis-accepted is eps |oc|^2 / 2r wide,
// 0.009 for that ground from 3 000 units away.)  Path segments inside the ground sphere of RandomWorld are such rays (one path
// in 10 000 gets in there and bounces to the depth limit); an Orthographic view plane that dips into it makes them by the
// thousand.  0 = an ordinary ray.  -DRT_FAR_WALK=0 compiles the growth out (A/B: profiles/r3/ab_far_rays.txt).
#ifndef RT_FAR_WALK
#define RT_FAR_WALK 1
#endif
RT_DEV float ray_fat(const DScene &sc, const Ray &r) {
    if (!RT_FAR_WALK) return 0.0f;
    const float dx = r.o.x - sc.far_centre[0], dy = r.o.y - sc.far_centre[1], dz = r.o.z - sc.far_centre[2];
    if (!(fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) > sc.far_limit)) return 0.0f;
    const float D = __builtin_sqrtf(dx * dx + dy * dy + dz * dz) * 1.000001f + sc.far_reach;
    const float s = 9.765625e-4f * D;        // sqrt(eps) D
    return (__builtin_sqrtf(__builtin_fmaf(s, s, sc.far_rmin * sc.far_rmin)) - sc.far_rmin) + s;
}

// 1 / d for the box tests only.  A component below 1e-30 in magnitude is replaced by 1e-30, so the reciprocal stays finite
// and lo*inv - o*inv cannot turn into inf - inf; the reciprocal is the hardware estimate (v_rcp_f32, 1 ulp) instead of three
// IEEE divisions of ~10 instructions each: a relative error of 1.2e-7 in a box's ray parameters is three orders of magnitude
// below the padding of the boxes (1e-4 x max(1, |coords|)) for any ray that starts within ~800 scene extents of the box.
// Nothing the reference computes goes through it: the primitive tests use the ray itself.
RT_DEV V3 box_inv(V3 d) {
    const float dx = fabsf(d.x) < 1.0e-30f ? 1.0e-30f : d.x;
    const float dy = fabsf(d.y) < 1.0e-30f ? 1.0e-30f : d.y;
    const float dz = fabsf(d.z) < 1.0e-30f ? 1.0e-30f : d.z;
    return v3(__builtin_amdgcn_rcpf(dx), __builtin_amdgcn_rcpf(dy), __builtin_amdgcn_rcpf(dz));
}

// Slab test in the traversal's own arithmetic (boxes are padded by the flattener, so this need
// not follow aabb3.rs): t = lo*inv - o*inv as one fma; `inv` is finite (see traverse()).
RT_DEV bool slab(V3 lo, V3 hi, V3 inv, V3 oi, float t_min, float t_max, float *t_near) {
    float tx0 = __builtin_fmaf(lo.x, inv.x, -oi.x), tx1 = __builtin_fmaf(hi.x, inv.x, -oi.x);
    float ty0 = __builtin_fmaf(lo.y, inv.y, -oi.y), ty1 = __builtin_fmaf(hi.y, inv.y, -oi.y);
    float tz0 = __builtin_fmaf(lo.z, inv.z, -oi.z), tz1 = __builtin_fmaf(hi.z, inv.z, -oi.z);
    float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), t_min));
    float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), t_max));
    *t_near = tn;
    return tn <= tf * 1.0000004f;
}

// Next stack entry whose box can still hold a closer-or-equal hit, or RT_BVH_EMPTY.
// Traversal stack of one lane: 4-byte node refs, the first lds_depth entries in LDS (column `lane` of a
// [depth][block] array, so a wave's accesses hit 64 different banks), deeper ones in an HBM overflow area.
// Entries carry no entry distance: LDS bytes per wave decide how many waves a CU holds, and the walk is
// latency-bound; a stale entry costs one node visit (both child boxes then fail against best.t).
// (The LDS pointer carries its address space in the type: a pointer that may be either LDS or global becomes a flat_load,
// which goes down both memory paths; typed, a pop from the LDS part is a ds_read_b32.)
#define RT_LDS __attribute__((address_space(3)))
typedef uint32_t StackEntry;
typedef RT_LDS StackEntry *LdsStackPtr;
struct Stack {
    LdsStackPtr lds;  uint32_t stride;
    uint32_t lds_depth;
    StackEntry *ovf;  uint32_t ovf_stride;   // unused (null) when lds_depth covers the BVH depth
};
RT_DEV void stack_push(const Stack &st, uint32_t &sp, int32_t ref) {
    if (sp < st.lds_depth) st.lds[sp * st.stride] = (uint32_t)ref;
    else st.ovf[(size_t)(sp - st.lds_depth) * st.ovf_stride] = (uint32_t)ref;
    ++sp;
}
RT_DEV int32_t pop_ref(const Stack &st, uint32_t &sp) {
    if (sp == 0) return RT_BVH_EMPTY;
    --sp;
    return (int32_t)(sp < st.lds_depth ? st.lds[sp * st.stride] : st.ovf[(size_t)(sp - st.lds_depth) * st.ovf_stride]);
}

// One visit of a 4-wide node (rt_api.h layout, 64 bytes): all four child boxes are tested against [t_min, best_t], the
// nearest hit child is returned, the other hit children are pushed farthest first (so the nearer one is popped
// first); with no hit the next stack entry is returned.  The order only decides how soon best_t shrinks: the
// closest hit and its tie-break do not depend on it.
// The walk is bound by VALU issue (PMC: 70-80 % of the issue slots on C1 / C3), so the visit is written for instruction
// count: the plane a ray enters a box through on an axis follows from the sign of its direction there (sx = scale * inv < 0:
// the parameter fma(q, sx, ox) falls as q grows, so it is the upper plane; the rounding of an fma is monotone, so this is
// the min / max of the two parameters exactly) — six selects per node instead of six min / max per child; (key, ref) pairs
// go through the sorting network together, so no child ref is looked up by index afterwards.
// The first `count` nodes of the tree in breadth-first order as a traversal workgroup holds them in LDS: the four 16-byte quarters
// of the records as four planes (quarter q of the k-th node at [q * count + k], so a wave's reads of one quarter of 64 different
// nodes spread over the banks).  A ref with RT_TOP_FLAG names a node of this image by its place k; the image's own child refs
// are rewritten accordingly when it is staged.  Every other ref is what it is everywhere else (a node of sc.nodes, a leaf, nothing).
// What it is for: the walk's node fetches return through the vector memory's data path (64 B per clock and CU), which PMC
// shows busy 60-75 % of a traversal launch — more than its vector ALUs; the nodes near the root are the ones every ray visits,
// and from LDS they cost that path nothing.
#define RT_TOP_FLAG 0x40000000
typedef float f32x4 __attribute__((ext_vector_type(4)));      // (a builtin vector: HIP's float4 class would be copied through a generic reference)
typedef const RT_LDS f32x4 *LdsF4Ptr;
RT_DEV float4 lds_f4(LdsF4Ptr p) { const f32x4 v = *p; return make_float4(v.x, v.y, v.z, v.w); }
struct TopLds {
    LdsF4Ptr img;
    uint32_t count;
};

template <bool STATS, bool LT = false, bool FAT = false>
RT_DEV int32_t node4_visit(const DScene &sc, int32_t ref, V3 inv, V3 oi, float t_min, float best_t, const Stack &stack, uint32_t &sp, Counters &cnt,
                           const TopLds &top = TopLds{(LdsF4Ptr)0, 0u}, float fat = 0.0f) {
    float4 n0, n1, n2, rf;
    if (LT && (ref & RT_TOP_FLAG)) {
        const LdsF4Ptr n = top.img + (uint32_t)(ref & ~RT_TOP_FLAG);
        n0 = lds_f4(n); n1 = lds_f4(n + top.count); n2 = lds_f4(n + 2u * top.count); rf = lds_f4(n + 3u * top.count);
    } else {
        const float4 *n = sc.nodes + 4 * (size_t)ref;
        n0 = n[0]; n1 = n[1]; n2 = n[2]; rf = n[3];
    }
    // plane = origin + scale * q, so its ray parameter (plane - o) * inv is q * (scale * inv) + (origin * inv - o * inv)
    const float sx = n0.w * inv.x, sy = n1.x * inv.y, sz = n1.y * inv.z;
    const float ox = __builtin_fmaf(n0.x, inv.x, -oi.x), oy = __builtin_fmaf(n0.y, inv.y, -oi.y), oz = __builtin_fmaf(n0.z, inv.z, -oi.z);
    // FAT (ray_fat): every box grown by `fat` on all sides — its near planes' parameters come earlier by fat * |inv|, its far planes' later
    const float ex = FAT ? fat * fabsf(inv.x) : 0.0f, ey = FAT ? fat * fabsf(inv.y) : 0.0f, ez = FAT ? fat * fabsf(inv.z) : 0.0f;
    const float oxn = FAT ? ox - ex : ox, oxf = FAT ? ox + ex : ox, oyn = FAT ? oy - ey : oy, oyf = FAT ? oy + ey : oy, ozn = FAT ? oz - ez : oz, ozf = FAT ? oz + ez : oz;
    const uint32_t qlx = __float_as_uint(n1.z), qhx = __float_as_uint(n1.w), qly = __float_as_uint(n2.x), qhy = __float_as_uint(n2.y),
                   qlz = __float_as_uint(n2.z), qhz = __float_as_uint(n2.w);
    const bool bx = inv.x < 0.0f, by = inv.y < 0.0f, bz = inv.z < 0.0f;
    const uint32_t qnx = bx ? qhx : qlx, qfx = bx ? qlx : qhx, qny = by ? qhy : qly, qfy = by ? qly : qhy, qnz = bz ? qhz : qlz, qfz = bz ? qlz : qhz;
    int32_t rr[4] = {f2i(rf.x), f2i(rf.y), f2i(rf.z), f2i(rf.w)};
    // sort key: entry distance (distances are >= t_min > 0 in the renderer, so their bit patterns order like unsigned
    // integers); a miss sorts last
    uint32_t key[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int sh = 8 * c;
        const float tnx = __builtin_fmaf((float)((qnx >> sh) & 255u), sx, oxn), tfx = __builtin_fmaf((float)((qfx >> sh) & 255u), sx, oxf);
        const float tny = __builtin_fmaf((float)((qny >> sh) & 255u), sy, oyn), tfy = __builtin_fmaf((float)((qfy >> sh) & 255u), sy, oyf);
        const float tnz = __builtin_fmaf((float)((qnz >> sh) & 255u), sz, ozn), tfz = __builtin_fmaf((float)((qfz >> sh) & 255u), sz, ozf);
        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, t_min));
        const float tf = fminf(fminf(tfx, tfy), fminf(tfz, best_t));
        const bool hit = tn <= tf * 1.0000004f && rr[c] != RT_BVH_EMPTY;
        if (STATS && rr[c] != RT_BVH_EMPTY) cnt.box_tests += 1;
        key[c] = hit ? __float_as_uint(tn) : 0xffffffffu;
    }
#define RT_CSWAP(i, j) { const bool sw_ = key[j] < key[i]; const uint32_t ki_ = key[i], kj_ = key[j]; const int32_t ri_ = rr[i], rj_ = rr[j]; \
                         key[i] = sw_ ? kj_ : ki_; key[j] = sw_ ? ki_ : kj_; rr[i] = sw_ ? rj_ : ri_; rr[j] = sw_ ? ri_ : rj_; }
    RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
#undef RT_CSWAP
    if (key[0] == 0xffffffffu) return pop_ref(stack, sp);
#pragma unroll
    for (int j = 3; j >= 1; --j)
        if (key[j] != 0xffffffffu) stack_push(stack, sp, rr[j]);
    return rr[0];
}

// "while-while" traversal: every lane walks inner nodes until it stands on a leaf (or is done);
// only then does the wave run the primitive tests, so node steps and leaf steps are not
// serialised against each other inside one iteration.
template <bool STATS, int EXT = EXT_XFORM | EXT_TRI>
RT_DEV void traverse(const DScene &sc, const Ray &r, float t_min, int32_t floor_id, Hit &best, const Stack &stack, Counters &cnt) {
    int32_t ref = sc.root_ref;
    if (ref == RT_BVH_EMPTY) return;
    V3 inv = box_inv(r.d);
    V3 oi = r.o * inv;
    const float fat = ray_fat(sc, r);
    uint32_t sp = 0;
    for (;;) {
        while (ref >= 0 && ref != RT_BVH_EMPTY) {
            if (STATS) cnt.node_visits += 1;
            ref = fat > 0.0f ? node4_visit<STATS, false, true>(sc, ref, inv, oi, t_min, best.t, stack, sp, cnt, TopLds{(LdsF4Ptr)0, 0u}, fat)
                             : node4_visit<STATS>(sc, ref, inv, oi, t_min, best.t, stack, sp, cnt);
        }
        if (ref == RT_BVH_EMPTY) return;
        uint32_t code = (uint32_t)~ref;
        uint32_t first = code >> 3, count = (code & 7u) + 1u;
        for (uint32_t i = 0; i < count; ++i) {
            float t;
            int32_t id;
            if (slot_hit<STATS, EXT>(sc, first + i, r, t_min, best.t, &t, &id, cnt) && id > floor_id) consider(best, t, id, (int32_t)(first + i));
        }
        ref = pop_ref(stack, sp);
    }
}

// ConstantMedium::hit (constant_medium.rs:46-94) for a sphere boundary.  Draws random_real()
// exactly when the reference does.
// boundary.hit() of a medium whose boundary is not a plain sphere: the boundary's leaves are slots [first, first + count)
// (a Block's six sides, a rectangle, a sphere; under Translate / RotateY) and the hit is HittableList::hit's fold over them,
// literally: the later candidate replaces the kept one unless kept.t < t (hittable_list.rs:43-49; block.rs:111-113).
template <bool STATS, int EXT>
RT_DEV bool boundary_hit(const DScene &sc, uint32_t first, uint32_t count, const Ray &r, float t_min, float t_max, float *t_out, Counters &cnt) {
    bool have = false;
    float kept = 0.0f;
    for (uint32_t i = 0; i < count; ++i) {
        float t; int32_t id;
        if (!slot_hit<STATS, EXT & (EXT_XFORM | EXT_CHAIN)>(sc, first + i, r, t_min, t_max, &t, &id, cnt)) continue;
        if (!have || !(kept < t)) kept = t;
        have = true;
    }
    *t_out = kept;
    return have;
}

template <bool STATS, int EXT>
RT_DEV bool medium_hit(const DScene &sc, uint32_t m, const Ray &r_world, float t_min, float t_max, rts::Rng &rng, float *t_out, Counters &cnt) {
    float4 a = sc.media[2 * m], b = sc.media[2 * m + 1];
    if (STATS) cnt.medium_tests += 1;
    V3 c = v3(a.x, a.y, a.z);
    float t1, t2;
    Ray r = r_world;
    if (f2i(b.w) == 1) {     // general boundary (rt_api.h medium record)
        const uint32_t first = (uint32_t)f2i(a.x), count = (uint32_t)f2i(a.y);
        // (the boundary slots carry the medium's own wrappers too, so they take the ray as it is)
        if (!boundary_hit<STATS, EXT>(sc, first, count, r_world, -RT_F32_MAX, RT_F32_MAX, &t1, cnt)) return false;
        if (!boundary_hit<STATS, EXT>(sc, first, count, r_world, t1 + 0.0001f, RT_F32_MAX, &t2, cnt)) return false;
        const int outer = f2i(a.z) - 1;          // the medium's own wrappers: the ray's length is measured in their frame
        if (outer >= 0) r = to_object<EXT>(sc, outer, r_world);
    } else {
    if (!sphere_roots(c, a.w, r, -RT_F32_MAX, RT_F32_MAX, &t1)) return false;
    if (!sphere_roots(c, a.w, r, t1 + 0.0001f, RT_F32_MAX, &t2)) return false;
    }
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return false;
    if (t1 < 0.0f) t1 = 0.0f;
    float ray_length = length(r.d);
    float distance_inside = (t2 - t1) * ray_length;
    float hit_distance = b.x * pm::logf_(rts::random_real(rng));
    if (hit_distance > distance_inside) return false;
    *t_out = t1 + hit_distance / ray_length;
    return true;
}

RT_DEV bool finite_f(float x) { return fabsf(x) < INFINITY; }

/* ------------------------------------------------------------------ the always-list */

// The always-list is the reference's flat `world` scan for the slots the BVH does not hold (every leaf of a scene of up to
// 24 leaves: the Cornell box, Blocks; quadrics anywhere): every slot is tested against every ray with the SAME (t_min, t_max)
// (hittable_list.rs:39-50).  The slot index is wave-uniform: slot i + 1 is fetched while slot i is tested (through the scalar
// cache outside the small kernel set, load_slot_uniform), and consecutive slots under one transform (the six sides of a block)
// share the transformed ray.  (A two-phase scan — a conservative reciprocal-estimate pre-test per rectangle, then the exact test
// per lane on its own candidates — changed no bit and was slower on MI355X: profiles/r2/experiments/always_filter.patch.)
template <bool STATS, int EXT = EXT_ALL>
RT_DEV void always_scan(const DScene &sc, const Ray &r, float t_min, float t_max, int32_t floor_id, Hit &best, Counters &cnt) {
    if (!sc.always_count) return;
    SlotRec nx = load_slot_uniform<EXT>(sc, sc.always_begin);
    int frame_xf = -1;
    Ray r_frame = r;
    for (uint32_t i = 0; i < sc.always_count; ++i) {
        const SlotRec cu = nx;
        const uint32_t s = sc.always_begin + i;
        if (i + 1 < sc.always_count) nx = load_slot_uniform<EXT>(sc, s + 1);
        const int w4 = f2i(cu.meta.x), xf = (w4 >> 8) - 1;   // wave-uniform
        if (xf != frame_xf) {
            frame_xf = xf;
            r_frame = xf < 0 ? r : (type_has_inline_xform(w4 & 15) ? to_object_rec<EXT>(sc, cu.q2, cu.q3, r) : to_object<EXT>(sc, xf, r));
        }
        float t; int32_t id;
        if (slot_hit_rec<STATS, EXT>(sc, s, cu, r, t_min, t_max, &t, &id, cnt, true, r_frame) && id > floor_id) consider(best, t, id, (int32_t)s);
    }
}

// Rays the BVH cannot serve with reference semantics: anything non-finite, or a null direction.
// NaN hit distances are ACCEPTED by the reference's comparisons (sphere.rs:61-65,
// rectangles.rs:25-32) and the list fold then keeps the last such object; those rays take the
// literal fold over all slots.
RT_DEV bool ray_is_finite(const Ray &r) {
    return finite_f(r.o.x) && finite_f(r.o.y) && finite_f(r.o.z) && finite_f(r.d.x) && finite_f(r.d.y) && finite_f(r.d.z) &&
           finite_f(r.time) && !(r.d.x == 0.0f && r.d.y == 0.0f && r.d.z == 0.0f);
}

// The reference's fold, literally: candidates in list order, `best` replaced unless best.t < t.
// Because slots are not stored in list order this runs as two passes over all slots: the fold's
// result is the minimum (ties -> later) over the candidates that come after the last candidate
// with an unordered (NaN) t, or that NaN candidate itself if it is the last one.
template <bool STATS, int EXT = EXT_ALL>
RT_DEV void closest_hit_exhaustive(const DScene &sc, const Ray &r, float t_min, float t_max, const float *med_t, const bool *med_ok,
                                   Hit &best, Counters &cnt) {
    int32_t last_nan_id = -1, last_nan_slot = 0;
    float last_nan_t = 0.0f;
    int32_t last_id = -1;
    for (uint32_t s = 0; s < sc.n_slots; ++s) {
        float t; int32_t id;
        if (!slot_hit<STATS, EXT>(sc, s, r, t_min, t_max, &t, &id, cnt)) continue;
        if (id > last_id) last_id = id;
        if (t != t && id > last_nan_id) { last_nan_id = id; last_nan_slot = (int32_t)s; last_nan_t = t; }
    }
    for (uint32_t m = 0; (EXT & EXT_MEDIA) && m < sc.n_media; ++m) {
        if (!med_ok[m]) continue;
        int32_t id = f2i(sc.media[2 * m + 1].z);
        if (id > last_id) last_id = id;
        if (med_t[m] != med_t[m] && id > last_nan_id) { last_nan_id = id; last_nan_slot = -(int32_t)(m + 1); last_nan_t = med_t[m]; }
    }
    best.t = INFINITY; best.id = -1; best.slot = 0;
    if (last_id < 0) return;
    if (last_nan_id == last_id) { best.t = last_nan_t; best.id = last_nan_id; best.slot = last_nan_slot; return; }
    bool any = false;
    for (uint32_t s = 0; s < sc.n_slots; ++s) {
        float t; int32_t id;
        if (!slot_hit<false, EXT>(sc, s, r, t_min, t_max, &t, &id, cnt)) continue;
        if (id <= last_nan_id) continue;
        if (!any || t < best.t || (t == best.t && id > best.id)) { best.t = t; best.id = id; best.slot = (int32_t)s; any = true; }
    }
    for (uint32_t m = 0; (EXT & EXT_MEDIA) && m < sc.n_media; ++m) {
        if (!med_ok[m]) continue;
        int32_t id = f2i(sc.media[2 * m + 1].z);
        if (id <= last_nan_id) continue;
        float t = med_t[m];
        if (!any || t < best.t || (t == best.t && id > best.id)) { best.t = t; best.id = id; best.slot = -(int32_t)(m + 1); any = true; }
    }
}

// world.hit(r, t_min, t_max): surfaces through the BVH, media in list order (they draw random
// numbers), everything folded with list semantics.  `with_media` = false for the parity hook.
template <bool STATS, int EXT = EXT_ALL>
RT_DEV bool closest_hit(const DScene &sc, const Ray &r, float t_min, float t_max, bool with_media, rts::Rng &rng, Hit &best,
                        const Stack &stack, Counters &cnt) {
    float med_t[RT_MAX_MEDIA];
    bool med_ok[RT_MAX_MEDIA];
#pragma unroll
    for (uint32_t m = 0; m < RT_MAX_MEDIA; ++m) {
        med_ok[m] = false; med_t[m] = 0.0f;
        if ((EXT & EXT_MEDIA) && with_media && m < sc.n_media) med_ok[m] = medium_hit<STATS, EXT>(sc, m, r, t_min, t_max, rng, &med_t[m], cnt);
    }
    if (!ray_is_finite(r)) {
        closest_hit_exhaustive<STATS, EXT>(sc, r, t_min, t_max, med_t, med_ok, best, cnt);
        return best.id >= 0;
    }
    // A finite ray gets a NaN distance from exactly one kind of leaf: a rectangle whose plane
    // contains the ray (d[axis] == 0 and o[axis] == k -> 0/0).  The fold keeps only what comes after
    // the last such candidate, so find it first (rect list only) and let everything else pass a
    // "later than that" filter.  Costs nothing unless a direction component is exactly zero.
    int32_t floor_id = -1, nan_slot = 0;
    float nan_t = 0.0f;
    if (r.d.x == 0.0f || r.d.y == 0.0f || r.d.z == 0.0f)
        for (uint32_t i = 0; i < sc.n_rects; ++i) {
            float t; int32_t id;
            uint32_t s = sc.rects[i];
            if (slot_hit<STATS, EXT>(sc, s, r, t_min, t_max, &t, &id, cnt) && t != t && id > floor_id) { floor_id = id; nan_slot = (int32_t)s; nan_t = t; }
        }
    best.t = t_max; best.id = -1; best.slot = 0;
    bool any = false;
    // t == t_max is accepted by every leaf (e.g. `t > t_max` rejects, rectangles.rs:25), so the
    // first candidate must win even at t == t_max: start below any id.
    always_scan<STATS, EXT>(sc, r, t_min, t_max, floor_id, best, cnt);
    traverse<STATS, (EXT & EXT_CHAIN) | EXT_XFORM | EXT_TRI>(sc, r, t_min, floor_id, best, stack, cnt);
    // MovingSphere boxes cover the shutter interval only; isotropic.rs:32 can hand a ray any "time"
    if (sc.n_moving && !(r.time >= 0.0f && r.time <= 1.0f))
        for (uint32_t i = 0; i < sc.n_moving; ++i) {
            float t; int32_t id;
            uint32_t s = sc.moving[i];
            if (slot_hit<STATS, EXT>(sc, s, r, t_min, t_max, &t, &id, cnt) && id > floor_id) consider(best, t, id, (int32_t)s);
        }
#pragma unroll
    for (uint32_t m = 0; m < RT_MAX_MEDIA; ++m)
        if ((EXT & EXT_MEDIA) && med_ok[m] && f2i(sc.media[2 * m + 1].z) > floor_id) consider(best, med_t[m], f2i(sc.media[2 * m + 1].z), -(int32_t)(m + 1));
    if (best.id < 0 && floor_id >= 0) { best.t = nan_t; best.id = floor_id; best.slot = nan_slot; }   // the NaN candidate was the last one
    any = best.id >= 0;
    return any;
}

/* ------------------------------------------------------------------ the rare serial folds, by the whole wave */

// A ray that is not finite takes the literal fold over every slot, twice (closest_hit_exhaustive), and a finite ray with a
// direction component of exactly zero looks at every rectangle for a NaN candidate: loops of dependent loads over thousands
// of slots (Book-2: 3 409 slots, 2 400 of them rectangles — 1.5 ms for one lane) that a launch of the pipeline's kernels cannot
// end before.  Such rays are rare — Book-2 at 100 spp: one path in 64 million turns NaN, and forty rays come out of
// random_in_unit_sphere with a zero — but a NaN path stays NaN until its depth runs out, and each of its bounces held a shade
// launch for those 1.5 ms: 45 of the frame's 80 ms (profiles/r3/ab_coop_folds.txt).  So the wave runs these loops together, one
// such ray at a time: lane l tests slots l, l + 64, ... of the leader's ray, and the partial results are folded with the serial
// loop's own rules — which do not depend on the order of the tests: the last NaN candidate is the one with the largest id (the
// first slot among equals), the closest hit the smallest t, then the largest id, then the first slot.  Same candidates, same
// comparisons, same winner, same bits.  All lanes of the wave must call these (with `need` false for a lane without such a ray).

RT_DEV Ray wave_ray_of(const Ray &r, bool mine, int leader) {
    Ray o;     // (a lane without a ray may hold anything in r: it hands on zeros)
    o.o = v3(__shfl(mine ? r.o.x : 0.0f, leader, 64), __shfl(mine ? r.o.y : 0.0f, leader, 64), __shfl(mine ? r.o.z : 0.0f, leader, 64));
    o.d = v3(__shfl(mine ? r.d.x : 0.0f, leader, 64), __shfl(mine ? r.d.y : 0.0f, leader, 64), __shfl(mine ? r.d.z : 0.0f, leader, 64));
    o.time = __shfl(mine ? r.time : 0.0f, leader, 64);
    return o;
}

// closest_hit_exhaustive() for the lanes with `need`
template <bool STATS, int EXT = EXT_ALL>
RT_DEV void coop_exhaustive(const DScene &sc, bool need, const Ray &r, float t_min, float t_max, const float *med_t, const bool *med_ok,
                            Hit &best, Counters &cnt) {
    const uint32_t lane = threadIdx.x & 63u;
    for (unsigned long long todo = __ballot(need); todo != 0ull; todo &= todo - 1ull) {      // wave-uniform
        const int leader = __builtin_ctzll(todo);
        const Ray rr = wave_ray_of(r, need, leader);
        // pass 1: the largest id among the candidates, and the NaN candidate with the largest id
        int32_t last_id = -1, nan_id = -1, nan_slot = 0x7fffffff;
        float nan_t = 0.0f;
        for (uint32_t s = lane; s < sc.n_slots; s += 64u) {
            float t; int32_t id;
            if (!slot_hit<STATS, EXT>(sc, s, rr, t_min, t_max, &t, &id, cnt)) continue;
            if (id > last_id) last_id = id;
            if (t != t && id > nan_id) { nan_id = id; nan_slot = (int32_t)s; nan_t = t; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int32_t oid = __shfl_xor(last_id, o, 64), nid = __shfl_xor(nan_id, o, 64), ns = __shfl_xor(nan_slot, o, 64);
            const float nt = __shfl_xor(nan_t, o, 64);
            if (oid > last_id) last_id = oid;
            if (nid > nan_id || (nid == nan_id && ns < nan_slot)) { nan_id = nid; nan_slot = ns; nan_t = nt; }
        }
        float lm_t[RT_MAX_MEDIA];
        bool lm_ok[RT_MAX_MEDIA];
#pragma unroll
        for (uint32_t m = 0; m < RT_MAX_MEDIA; ++m) {      // the leader's media candidates, after the slots as in the serial fold
            lm_ok[m] = (EXT & EXT_MEDIA) && m < sc.n_media && __shfl(need && med_ok[m] ? 1 : 0, leader, 64) != 0;
            lm_t[m] = __shfl(need ? med_t[m] : 0.0f, leader, 64);
            if (!lm_ok[m]) continue;
            const int32_t id = f2i(sc.media[2 * m + 1].z);
            if (id > last_id) last_id = id;
            if (lm_t[m] != lm_t[m] && id > nan_id) { nan_id = id; nan_slot = -(int32_t)(m + 1); nan_t = lm_t[m]; }
        }
        Hit res;
        res.t = INFINITY; res.id = -1; res.slot = 0;
        if (last_id >= 0) {                                // wave-uniform from here on
            if (nan_id == last_id) { res.t = nan_t; res.id = nan_id; res.slot = nan_slot; }
            else {
                // pass 2: the closest of what comes after that NaN candidate (none of those has a NaN distance)
                bool any = false;
                float bt = 0.0f;
                int32_t bid = -1, bslot = 0x7fffffff;
                for (uint32_t s = lane; s < sc.n_slots; s += 64u) {
                    float t; int32_t id;
                    if (!slot_hit<false, EXT>(sc, s, rr, t_min, t_max, &t, &id, cnt)) continue;
                    if (id <= nan_id) continue;
                    if (!any || t < bt || (t == bt && id > bid)) { bt = t; bid = id; bslot = (int32_t)s; any = true; }
                }
                for (int o = 32; o > 0; o >>= 1) {
                    const bool oany = __shfl_xor(any ? 1 : 0, o, 64) != 0;
                    const float ot = __shfl_xor(bt, o, 64);
                    const int32_t oid = __shfl_xor(bid, o, 64), oslot = __shfl_xor(bslot, o, 64);
                    if (oany && (!any || ot < bt || (ot == bt && (oid > bid || (oid == bid && oslot < bslot))))) { bt = ot; bid = oid; bslot = oslot; any = true; }
                }
#pragma unroll
                for (uint32_t m = 0; m < RT_MAX_MEDIA; ++m) {
                    if (!lm_ok[m]) continue;
                    const int32_t id = f2i(sc.media[2 * m + 1].z);
                    if (id <= nan_id) continue;
                    if (!any || lm_t[m] < bt || (lm_t[m] == bt && id > bid)) { bt = lm_t[m]; bid = id; bslot = -(int32_t)(m + 1); any = true; }
                }
                if (any) { res.t = bt; res.id = bid; res.slot = bslot; }
            }
        }
        if ((int)lane == leader) best = res;
    }
}

// The id of the last NaN candidate among the rectangles (closest_hit's floor_id) for the lanes with `need`
template <bool STATS, int EXT = EXT_ALL>
RT_DEV void coop_rect_floor(const DScene &sc, bool need, const Ray &r, float t_min, float t_max, int32_t &floor_id, Counters &cnt) {
    const uint32_t lane = threadIdx.x & 63u;
    for (unsigned long long todo = __ballot(need); todo != 0ull; todo &= todo - 1ull) {
        const int leader = __builtin_ctzll(todo);
        const Ray rr = wave_ray_of(r, need, leader);
        int32_t f = -1;
        for (uint32_t i = lane; i < sc.n_rects; i += 64u) {
            float t; int32_t id;
            if (slot_hit<STATS, EXT>(sc, sc.rects[i], rr, t_min, t_max, &t, &id, cnt) && t != t && id > f) f = id;
        }
        for (int o = 32; o > 0; o >>= 1) { const int32_t of = __shfl_xor(f, o, 64); if (of > f) f = of; }
        if ((int)lane == leader) floor_id = f;
    }
}

/* ------------------------------------------------------------------ resumable traversal (render kernel) */

// The render kernel does not call closest_hit(): it keeps the traversal of every lane in this
// struct and advances it one node or one leaf at a time, so the wave can decide when to traverse
// and when to shade (see render_kernel).  The result is the same as closest_hit()'s: the steps,
// the pruning and the fold are identical, only their interleaving with other lanes differs.
struct Trav {
    int32_t ref;        // current node (>= 0), leaf (< 0) or RT_BVH_EMPTY when finished
    uint32_t sp;
    Hit best;
    int32_t floor_id;   // id of the last NaN candidate (see closest_hit), -1 if none
    V3 inv, oi;
    float fat;          // ray_fat(): how much every box is grown for this ray (0: an ordinary ray)
};

// What begin_trace() leaves for settle_trace(): the rare serial folds (above), which the wave runs together at a place of the
// caller's choosing — the end of the kernel, where little else is alive (run where begin_trace is called, their loops over the
// slots cost every kernel 15 ... 25 VGPRs, and the scenes that never need them 5 ... 10 % of their frame).
struct LateFolds {
    bool literal;                 // not a finite ray: the literal fold over every slot (tv.best is still to be found, there is no walk)
    bool axis;                    // a direction component of exactly zero: a rectangle may hold a NaN candidate (the floor_id)
    float med_t[RT_MAX_MEDIA];    // the media candidates (they have drawn their random numbers)
    bool med_ok[RT_MAX_MEDIA];
};

// world.hit() from the always-list to the start of the BVH walk, for a finite ray whose last NaN candidate is tv.floor_id.
template <bool STATS, int EXT = EXT_ALL>
RT_DEV bool begin_trace_rest(const DScene &sc, const Ray &r, float t_min, float t_max, Trav &tv, const float *med_t, const bool *med_ok, Counters &cnt) {
    tv.best.t = t_max; tv.best.id = -1; tv.best.slot = 0;
    RG_BEGIN(RG_B_ALWAYS);
    always_scan<STATS, EXT>(sc, r, t_min, t_max, tv.floor_id, tv.best, cnt);
    RG_END(RG_B_ALWAYS);
    RG_BEGIN(RG_B_REST);
    if (sc.n_moving && !(r.time >= 0.0f && r.time <= 1.0f))
        for (uint32_t i = 0; i < sc.n_moving; ++i) {
            float t; int32_t id;
            uint32_t s = sc.moving[i];
            if (slot_hit<STATS, EXT>(sc, s, r, t_min, t_max, &t, &id, cnt) && id > tv.floor_id) consider(tv.best, t, id, (int32_t)s);
        }
#pragma unroll
    for (uint32_t m = 0; m < RT_MAX_MEDIA; ++m)
        if ((EXT & EXT_MEDIA) && med_ok[m] && f2i(sc.media[2 * m + 1].z) > tv.floor_id) consider(tv.best, med_t[m], f2i(sc.media[2 * m + 1].z), -(int32_t)(m + 1));
    if (sc.root_ref == RT_BVH_EMPTY) { RG_END(RG_B_REST); return false; }
    tv.inv = box_inv(r.d);
    tv.oi = r.o * tv.inv;
    tv.fat = ray_fat(sc, r);
    tv.ref = sc.root_ref;
    RG_END(RG_B_REST);
    return true;
}

// world.hit() up to the point where the BVH walk starts.  Returns true if there is a walk to do.  What `late` names is still
// owed: every lane of the wave must call settle_trace() afterwards, before anybody looks at tv.
template <bool STATS, int EXT = EXT_ALL>
RT_DEV bool begin_trace(const DScene &sc, const Ray &r, float t_min, float t_max, rts::Rng &rng, Trav &tv, Counters &cnt, LateFolds &late) {
    RG_BEGIN(RG_B_MEDIA);
#pragma unroll
    for (uint32_t m = 0; m < RT_MAX_MEDIA; ++m) {
        late.med_ok[m] = false; late.med_t[m] = 0.0f;
        if ((EXT & EXT_MEDIA) && m < sc.n_media) late.med_ok[m] = medium_hit<STATS, EXT>(sc, m, r, t_min, t_max, rng, &late.med_t[m], cnt);
    }
    RG_END(RG_B_MEDIA);
    tv.floor_id = -1; tv.sp = 0; tv.ref = RT_BVH_EMPTY;
    late.literal = !ray_is_finite(r);
    late.axis = !late.literal && (r.d.x == 0.0f || r.d.y == 0.0f || r.d.z == 0.0f);
    if (late.literal) { tv.best.t = INFINITY; tv.best.id = -1; tv.best.slot = 0; return false; }
    return begin_trace_rest<STATS, EXT>(sc, r, t_min, t_max, tv, late.med_t, late.med_ok, cnt);
}

// The folds begin_trace() left, for the lanes with `active`; called by ALL lanes of the wave.  True for a lane whose tv and
// `walk` are new (a finite ray's only if a rectangle did hold a NaN candidate: everything after the rectangles is then done
// again with that floor_id — without counting its tests a second time).
template <bool STATS, int EXT = EXT_ALL>
RT_DEV bool settle_trace(const DScene &sc, bool active, const Ray &r, float t_min, float t_max, Trav &tv, bool &walk, const LateFolds &late, Counters &cnt) {
    const bool literal = active && late.literal, axis = active && late.axis;
    if (__builtin_expect(__ballot(literal || axis) == 0ull, 1)) return false;
    RG_BEGIN(RG_B_RECTS);
    bool changed = false;
    if (__ballot(literal) != 0ull) {
        coop_exhaustive<STATS, EXT>(sc, literal, r, t_min, t_max, late.med_t, late.med_ok, tv.best, cnt);
        changed = literal;
    }
    if (__ballot(axis) != 0ull) {
        int32_t floor_id = -1;
        coop_rect_floor<STATS, EXT>(sc, axis, r, t_min, t_max, floor_id, cnt);
        if (axis && floor_id >= 0) {
            tv.floor_id = floor_id; tv.sp = 0; tv.ref = RT_BVH_EMPTY;
            walk = begin_trace_rest<false, EXT>(sc, r, t_min, t_max, tv, late.med_t, late.med_ok, cnt);
            changed = true;
        }
    }
    RG_END(RG_B_RECTS);
    return changed;
}

template <bool STATS, bool LT = false>
RT_DEV void trav_node_step(const DScene &sc, float t_min, Trav &tv, const Stack &stack, Counters &cnt, const TopLds &top = TopLds{(LdsF4Ptr)0, 0u}) {
    if (STATS) cnt.node_visits += 1;
    if (RT_FAR_WALK && tv.fat > 0.0f) tv.ref = node4_visit<STATS, LT, true>(sc, tv.ref, tv.inv, tv.oi, t_min, tv.best.t, stack, tv.sp, cnt, top, tv.fat);
    else tv.ref = node4_visit<STATS, LT>(sc, tv.ref, tv.inv, tv.oi, t_min, tv.best.t, stack, tv.sp, cnt, top);
}

template <bool STATS, int EXT>
RT_DEV void trav_leaf_step(const DScene &sc, const Ray &r, float t_min, Trav &tv, const Stack &stack, Counters &cnt) {
    uint32_t code = (uint32_t)~tv.ref;
    uint32_t first = code >> 3, count = (code & 7u) + 1u;
    for (uint32_t i = 0; i < count; ++i) {
        float t;
        int32_t id;
        if (slot_hit<STATS, EXT>(sc, first + i, r, t_min, tv.best.t, &t, &id, cnt) && id > tv.floor_id) consider(tv.best, t, id, (int32_t)(first + i));
    }
    tv.ref = pop_ref(stack, tv.sp);
}

// The primitive tests of one leaf, for a walk that keeps the leaf aside and goes on (wf_traverse_kernel): no pop.
template <bool STATS, int EXT>
RT_DEV void trav_leaf_test(const DScene &sc, const Ray &r, float t_min, Trav &tv, int32_t leaf_ref, Counters &cnt) {
    uint32_t code = (uint32_t)~leaf_ref;
    uint32_t first = code >> 3, count = (code & 7u) + 1u;
    for (uint32_t i = 0; i < count; ++i) {
        float t;
        int32_t id;
        if (slot_hit<STATS, EXT>(sc, first + i, r, t_min, tv.best.t, &t, &id, cnt) && id > tv.floor_id) consider(tv.best, t, id, (int32_t)(first + i));
    }
}

// A candidate ahead of the walk: the slot `slot` (one the walk can reach: a BVH leaf), if it is a plain sphere.  Its test is the
// leaf's own (sphere.rs:45-66 with the walk's bounds), so an accepted hit is what the walk would fold when it meets the leaf — if
// it meets it: the walk only tests a leaf whose box the ray passes through, and in f32 a sphere can accept a ray that passes far
// from it (cancellation), so the hit only counts when its point lies on the sphere to within half the padding of
// the boxes, i.e. inside every box above the leaf.  The fold does not depend on the order candidates arrive in, and the walk
// leaves the slot as it is when it meets it again (same t, same id).
RT_DEV void hint_sphere(const DScene &sc, uint32_t slot, const Ray &r, float t_min, Trav &tv) {
    const SlotRec s = load_slot(sc, slot);
    const int w4 = f2i(s.meta.x);
    if ((w4 & 15) != RT_PRIM_SPHERE || (w4 >> 8) != 0) return;
    const V3 c = v3(s.q0.x, s.q0.y, s.q0.z);
    float t;
    if (!sphere_roots(c, s.q0.w, r, t_min, tv.best.t, &t)) return;
    const float r2 = s.q0.w * s.q0.w;
    if (!(fabsf(length_squared(r.at(t) - c) - r2) <= 1.0e-4f * r2)) return;
    const int32_t id = f2i(s.meta.z);
    if (id > tv.floor_id) consider(tv.best, t, id, (int32_t)slot);
}

// After the walk: if nothing came after the last NaN candidate, that candidate is the hit (the last rectangle, in the order of
// sc.rects, that answers with its id).  Called by ALL lanes of the wave; the wave looks for one lane's candidate at a time.
template <bool STATS, int EXT = EXT_ALL>
RT_DEV void finish_trace(const DScene &sc, bool active, const Ray &r, float t_min, float t_max, Trav &tv, Counters &cnt) {
    const bool need = active && !(tv.best.id >= 0 || tv.floor_id < 0);
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long todo = __ballot(need);
    if (__builtin_expect(todo == 0ull, 1)) return;
    for (; todo != 0ull; todo &= todo - 1ull) {
        const int leader = __builtin_ctzll(todo);
        const Ray rr = wave_ray_of(r, need, leader);
        const int32_t want = __shfl(need ? tv.floor_id : -1, leader, 64);
        int32_t at = -1;
        float at_t = 0.0f;
        for (uint32_t i = lane; i < sc.n_rects; i += 64u) {
            float t; int32_t id;
            if (slot_hit<false, EXT>(sc, sc.rects[i], rr, t_min, t_max, &t, &id, cnt) && id == want) { at = (int32_t)i; at_t = t; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int32_t oat = __shfl_xor(at, o, 64);
            const float ot = __shfl_xor(at_t, o, 64);
            if (oat > at) { at = oat; at_t = ot; }
        }
        if ((int)lane == leader && at >= 0) { tv.best.t = at_t; tv.best.id = want; tv.best.slot = (int32_t)sc.rects[at]; }
    }
}

/* ------------------------------------------------------------------ textures */

RT_DEV float perlin_smooth(float t) { return t * t * (3.0f - 2.0f * t); }  // perlin.rs:60-62

RT_DEV float perlin_noise(const uint32_t *tab, V3 p) {  // perlin.rs:84-108 + 38-58
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = perlin_smooth(p.x - fx), v = perlin_smooth(p.y - fy), w = perlin_smooth(p.z - fz);
    int i = pm::f2i_sat(fx), j = pm::f2i_sat(fy), k = pm::f2i_sat(fz);
    float uu = perlin_smooth(u), vv = perlin_smooth(v), ww = perlin_smooth(w);
    float accum = 0.0f;
    // The 8 corners as one rolled loop (same order as the reference's three nested ones): fully unrolled, the
    // corner gradients and weights of a 7-octave turbulence cost the shade kernel 40 VGPRs, i.e. a wave per SIMD.
#pragma unroll 1
    for (int corner = 0; corner < 8; ++corner) {
        const int di = corner >> 2, dj = (corner >> 1) & 1, dk = corner & 1;
        {
            {
                uint32_t idx = tab[768 + ((uint32_t)(i + di) & 255u)] ^ tab[1024 + ((uint32_t)(j + dj) & 255u)] ^
                               tab[1280 + ((uint32_t)(k + dk) & 255u)];
                V3 c = v3(__uint_as_float(tab[3 * idx]), __uint_as_float(tab[3 * idx + 1]), __uint_as_float(tab[3 * idx + 2]));
                V3 weight = v3(u - (float)di, v - (float)dj, w - (float)dk);
                accum += ((float)di * uu + (1.0f - (float)di) * (1.0f - uu)) * ((float)dj * vv + (1.0f - (float)dj) * (1.0f - vv)) *
                         ((float)dk * ww + (1.0f - (float)dk) * (1.0f - ww)) * dot(c, weight);
            }
        }
    }
    return accum;
}

RT_DEV float perlin_turbulence(const uint32_t *tab, V3 p, int depth) {  // perlin.rs:110-123
    V3 temp_p = p;
    float weight = 1.0f, acc = 0.0f;
#pragma unroll 1
    for (int d = 0; d < depth; ++d) {
        acc = acc + weight * perlin_noise(tab, temp_p);
        weight *= 0.5f;
        temp_p = temp_p * 2.0f;
    }
    return fabsf(acc);
}

// The product that perlin_noise() adds for one corner (perlin.rs:38-58, :84-108): the same expression as in the loop above, so
// the same bits.
RT_DEV float perlin_corner_term(const uint32_t *tab, V3 p, int corner) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = perlin_smooth(p.x - fx), v = perlin_smooth(p.y - fy), w = perlin_smooth(p.z - fz);
    int i = pm::f2i_sat(fx), j = pm::f2i_sat(fy), k = pm::f2i_sat(fz);
    float uu = perlin_smooth(u), vv = perlin_smooth(v), ww = perlin_smooth(w);
    const int di = corner >> 2, dj = (corner >> 1) & 1, dk = corner & 1;
    uint32_t idx = tab[768 + ((uint32_t)(i + di) & 255u)] ^ tab[1024 + ((uint32_t)(j + dj) & 255u)] ^
                   tab[1280 + ((uint32_t)(k + dk) & 255u)];
    V3 c = v3(__uint_as_float(tab[3 * idx]), __uint_as_float(tab[3 * idx + 1]), __uint_as_float(tab[3 * idx + 2]));
    V3 weight = v3(u - (float)di, v - (float)dj, w - (float)dk);
    return ((float)di * uu + (1.0f - (float)di) * (1.0f - uu)) * ((float)dj * vv + (1.0f - (float)dj) * (1.0f - vv)) *
           ((float)dk * ww + (1.0f - (float)dk) * (1.0f - ww)) * dot(c, weight);
}

RT_DEV float lane_read(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// perlin_turbulence(table, p, 7) for the lanes with `need`, computed by the whole wave.  A noise hit is 7 octaves x 8 corners of
// dependent table look-ups, ~2 200 instructions, and a wave usually holds a handful of them (the Perlin sphere of the Book-2 scene
// is 1-3 % of all hits): run per lane, those few lanes cost the wave the whole sequence.  Here the wave takes the hits one at a
// time: lane 8 * o + c computes corner c of octave o (56 lanes, one term each, ~45 instructions), then the terms are added in
// the reference's order — corners 0..7 within an octave (perlin.rs:38-58), octaves 0..6 with their weights (perlin.rs:110-123) —
// so the value has the bits of the per-lane loop.  The point of octave o is p doubled o times, i.e. p * 2^o exactly.
// Must be called by all 64 lanes of the wave.
RT_DEV float coop_turbulence(const DScene &sc, unsigned long long m_need, V3 p, int table) {
    const uint32_t lane = threadIdx.x & 63u;
    const int oct = (int)(lane >> 3), corner = (int)(lane & 7u);
    const float scale = (float)(1u << oct);
    float out = 0.0f;
    while (m_need != 0ull) {      // wave-uniform
        const int src = __builtin_ctzll(m_need);
        m_need &= m_need - 1ull;
        const float px = lane_read(p.x, src), py = lane_read(p.y, src), pz = lane_read(p.z, src);
        const int tb = __builtin_amdgcn_readlane(table, src);
        const uint32_t *tab = (sc.perlin_lds && tb == 0) ? sc.perlin_lds : sc.perlin + (size_t)tb * RT_PERLIN_WORDS;
        const float term = perlin_corner_term(tab, v3(px * scale, py * scale, pz * scale), corner);
        float accum = 0.0f;       // perlin_noise of this lane's octave: the eight terms in corner order
#pragma unroll
        for (int c = 0; c < 8; ++c) accum += __shfl(term, (int)(lane & ~7u) + c, 64);
        float acc = 0.0f, weight = 1.0f;
#pragma unroll
        for (int o = 0; o < 7; ++o) {
            acc = acc + weight * lane_read(accum, 8 * o);
            weight *= 0.5f;
        }
        if ((int)lane == src) out = fabsf(acc);
    }
    return out;
}

// Device-only texture kind (rt_scene_upload writes it into its copy of the texture records; never in an rt_scene_desc)
enum { RT_TEX_CHECKER_SOLIDS = 4 };

template <int EXT = EXT_ALL>
RT_DEV bool texture_needs_uv(const DScene &sc, int tex) {
    if (!(EXT & EXT_IMAGE)) return false;     // u, v only matter to an image texture
    for (int level = 0; level < 8; ++level) {
        int type = f2i(sc.textures[2 * tex].x);
        if (type == RTG_TEX_IMAGE) return true;
        if (type != RTG_TEX_CHECKER) return false;
        float4 b = sc.textures[2 * tex + 1];
        // either branch may be an image; be conservative
        int odd = f2i(b.x), even = f2i(b.y);
        if (f2i(sc.textures[2 * odd].x) == RTG_TEX_IMAGE || f2i(sc.textures[2 * even].x) == RTG_TEX_IMAGE) return true;
        tex = even;
    }
    return true;
}

// (have_turb: `turb` is perlin_turbulence(p, 7) for the texture itself if that is a noise texture — coop_turbulence)
template <bool STATS, int EXT = EXT_ALL>
RT_DEV V3 texture_value(const DScene &sc, int tex, float u, float v, V3 p, Counters &cnt, bool have_turb = false, float turb = 0.0f) {
    for (int level = 0; level < 16; ++level) {
        float4 a = sc.textures[2 * tex], b = sc.textures[2 * tex + 1];
        int type = f2i(a.x);
        if (STATS) cnt.tex_evals[type & 3] += 1;
        if (type == RTG_TEX_SOLID) return v3(a.y, a.z, a.w);
        if (type == RT_TEX_CHECKER_SOLIDS) {  // device copy of a checker over two solid colours: odd rgb | even rgb, scale
            if (STATS) cnt.tex_evals[RTG_TEX_CHECKER] += 1;   // (the line above has counted the solid colour: type & 3 == 0)
            float f = b.w;
            float sines = pm::sinf_(f * p.x) * pm::sinf_(f * p.y) * pm::sinf_(f * p.z);
            return sines < 0.0f ? v3(a.y, a.z, a.w) : v3(b.x, b.y, b.z);
        }
        if (type == RTG_TEX_CHECKER) {  // checker_texture.rs:37-45
            float f = b.z;
            float sines = pm::sinf_(f * p.x) * pm::sinf_(f * p.y) * pm::sinf_(f * p.z);
            tex = sines < 0.0f ? f2i(b.x) : f2i(b.y);
            continue;
        }
        if ((EXT & EXT_NOISE) && type == RTG_TEX_NOISE) {  // noise_texture.rs:28-30
            const uint32_t *tab = (sc.perlin_lds && f2i(b.x) == 0) ? sc.perlin_lds : sc.perlin + (size_t)f2i(b.x) * RT_PERLIN_WORDS;
            const float tu = (have_turb && level == 0) ? turb : perlin_turbulence(tab, p, 7);
            float k = 1.0f + pm::sinf_(b.z * p.z + 10.0f * tu);
            float g = 1.0f * 0.5f;
            return v3(g * k, g * k, g * k);
        }
        if (!(EXT & EXT_IMAGE)) return v3(0.0f, 0.0f, 0.0f);
        // image_texture.rs:50-86
        int off = f2i(b.x);
        if (off < 0) return v3(0.0f, 1.0f, 1.0f);
        int width = f2i(b.y), height = f2i(b.w);
        float uc = f32_clamp(u, 0.0f, 1.0f);
        float vc = 1.0f - f32_clamp(v, 0.0f, 1.0f);
        int i = pm::f2i_sat(uc * (float)width), j = pm::f2i_sat(vc * (float)height);
        if (i >= width) i = width - 1;
        if (j >= height) j = height - 1;
        uint32_t px = sc.texels[(size_t)off + (size_t)j * (size_t)width + (size_t)i];
        float scale = 1.0f / 255.0f;
        return v3(scale * (float)(px & 255u), scale * (float)((px >> 8) & 255u), scale * (float)((px >> 16) & 255u));
    }
    return v3(0.0f, 0.0f, 0.0f);
}

/* ------------------------------------------------------------------ pdfs */

struct Onb {
    V3 u, v, w;
};
RT_DEV Onb onb_from(V3 n) {  // onb.rs:32-47
    Onb o;
    o.w = normalize(n);
    V3 a = fabsf(o.w.x) > 0.9f ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
    o.v = normalize(cross(o.w, a));
    o.u = cross(o.w, o.v);
    return o;
}
RT_DEV V3 onb_local(const Onb &o, V3 a) { return a.x * o.u + a.y * o.v + a.z * o.w; }

RT_DEV V3 random_in_unit_sphere(rts::Rng &rng) {  // types.rs:95-104
    for (;;) {
        float x = rts::random_real_range(rng, -1.0f, 1.0f);
        float y = rts::random_real_range(rng, -1.0f, 1.0f);
        float z = rts::random_real_range(rng, -1.0f, 1.0f);
        V3 p = v3(x, y, z);
        if (length_squared(p) >= 1.0f) continue;
        return p;
    }
}

// pdf_value of one entry of `lights` for the ray r = (origin, v, time 0)
template <bool STATS, int EXT = EXT_ALL>
RT_DEV float light_pdf_one(float4 a, float4 b, const Ray &r, V3 origin, V3 v, Counters &cnt) {
    const int type = f2i(a.x);
    float val = 0.0f;
    if (STATS) cnt.light_pdf_evals += 1;
    if (type == RT_PRIM_SPHERE || type == RT_PRIM_MOVING_SPHERE) {  // sphere.rs:91-105; :204-219: a MovingSphere where it is at time 0
        float t;
        V3 c = v3(a.y, a.z, a.w);
        if (sphere_roots(c, b.x, r, 0.001f, RT_F32_MAX, &t)) {
            float cos_theta_max = sqrtf(1.0f - b.x * b.x / length_squared(c - origin));
            float solid_angle = 2.0f * 3.14159265358979323846f * (1.0f - cos_theta_max);
            val = 1.0f / solid_angle;
        }
    } else if ((EXT & EXT_QUADRICS) && type == RT_PRIM_CYLINDER) {  // cylinder.rs:134-150: zmin zmax phi_max | radius
        SlotRec cyl;
        cyl.q0 = make_float4(b.x, a.y, a.z, a.w);
        cyl.meta = cyl.q2 = cyl.q3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        QuadHit qh;
        if (quadric_test(RT_PRIM_CYLINDER, cyl, r, 0.0001f, __builtin_inff(), &qh)) {
            float qu, qv;
            const V3 n = quadric_record(RT_PRIM_CYLINDER, cyl, qh, &qu, &qv);   // |dot(n, -v)|: the record's face flip drops out
            const float area = (a.z - a.y) * b.x * a.w;
            const float pdf = (1.0f / area) / (fabsf(dot(n, -v)) / length_squared(origin - qh.p));
            val = isinf(pdf) ? 0.0f : pdf;
        }
    } else if (type != 255) {  // rectangles.rs:54-67
        float t, pa, pb;
        float4 q0 = make_float4(a.y, a.z, a.w, b.x);
        if (rect_test(type, q0, b.y, r, 0.001f, RT_F32_MAX, &t, &pa, &pb)) {
            int AX = type == RT_PRIM_XY_RECT ? 2 : (type == RT_PRIM_XZ_RECT ? 1 : 0);
            V3 n = v3(AX == 0 ? 1.0f : 0.0f, AX == 1 ? 1.0f : 0.0f, AX == 2 ? 1.0f : 0.0f);
            bool front = dot(v, n) < 0.0f;
            V3 nn = front ? n : -n;
            float area = (q0.y - q0.x) * (q0.w - q0.z);
            float distance_squared = t * t * length_squared(v);
            float cosine = fabsf(dot(v, nn) / length(v));
            val = distance_squared / (cosine * area);
        }
    }
    return val;
}

// HittableList::pdf_value of `lights` (hittable_list.rs:59-69): sum + weight * pdf over the members, in order.  Lists inside
// the list (records 254 under a root record 253, rt_api.h) are the same fold one level down, started at 0 and added to the
// outer sum as one member: an explicit stack of at most 4 folds, in the kernel set with EXT_CHAIN only.
template <bool STATS, int EXT = EXT_ALL>
RT_DEV float lights_pdf_value(const DScene &sc, V3 origin, V3 v, Counters &cnt) {
    if (sc.n_lights == 0) return 0.0f;
    Ray r{origin, v, 0.0f};
    if ((EXT & EXT_CHAIN) && f2i(sc.lights[0].x) == 253) {
        uint32_t at[4], end[4];
        float weight[4], sum[4];
        int sp = 0;
        at[0] = (uint32_t)f2i(sc.lights[0].y); end[0] = at[0] + (uint32_t)f2i(sc.lights[0].z);
        weight[0] = 1.0f / (float)f2i(sc.lights[0].z); sum[0] = 0.0f;
        for (;;) {
            if (at[sp] == end[sp]) {
                if (sp == 0) return sum[0];
                const float inner = sum[sp];
                --sp;
                sum[sp] = sum[sp] + weight[sp] * inner;
                continue;
            }
            const uint32_t i = at[sp]++;
            const float4 a = sc.lights[2 * i], b = sc.lights[2 * i + 1];
            if (f2i(a.x) == 254 && sp < 3) {
                ++sp;
                at[sp] = (uint32_t)f2i(a.y); end[sp] = at[sp] + (uint32_t)f2i(a.z);
                weight[sp] = 1.0f / (float)f2i(a.z); sum[sp] = 0.0f;
                continue;
            }
            sum[sp] = sum[sp] + weight[sp] * light_pdf_one<STATS, EXT>(a, b, r, origin, v, cnt);
        }
    }
    // The plain list.  The loop body is light_pdf_one() spelled out: as a call to it (force-inlined, or a lambda) the compiler
    // schedules the shade kernel differently and C3 loses 2 % (profiles/r2/ab_lights_loop_form.txt).
    float weight = 1.0f / (float)sc.n_lights;
    float sum = 0.0f;
    for (uint32_t i = 0; i < sc.n_lights; ++i) {
        float4 a = sc.lights[2 * i], b = sc.lights[2 * i + 1];
        int type = f2i(a.x);
        float val = 0.0f;
        if (STATS) cnt.light_pdf_evals += 1;
        if (type == RT_PRIM_SPHERE || type == RT_PRIM_MOVING_SPHERE) {
            float t;
            V3 c = v3(a.y, a.z, a.w);
            if (sphere_roots(c, b.x, r, 0.001f, RT_F32_MAX, &t)) {
                float cos_theta_max = sqrtf(1.0f - b.x * b.x / length_squared(c - origin));
                float solid_angle = 2.0f * 3.14159265358979323846f * (1.0f - cos_theta_max);
                val = 1.0f / solid_angle;
            }
        } else if ((EXT & EXT_QUADRICS) && type == RT_PRIM_CYLINDER) {
            val = light_pdf_one<false, EXT>(a, b, r, origin, v, cnt);
        } else if (type != 255) {
            float t, pa, pb;
            float4 q0 = make_float4(a.y, a.z, a.w, b.x);
            if (rect_test(type, q0, b.y, r, 0.001f, RT_F32_MAX, &t, &pa, &pb)) {
                int AX = type == RT_PRIM_XY_RECT ? 2 : (type == RT_PRIM_XZ_RECT ? 1 : 0);
                V3 n = v3(AX == 0 ? 1.0f : 0.0f, AX == 1 ? 1.0f : 0.0f, AX == 2 ? 1.0f : 0.0f);
                bool front = dot(v, n) < 0.0f;
                V3 nn = front ? n : -n;
                float area = (q0.y - q0.x) * (q0.w - q0.z);
                float distance_squared = t * t * length_squared(v);
                float cosine = fabsf(dot(v, nn) / length(v));
                val = distance_squared / (cosine * area);
            }
        }
        sum = sum + weight * val;
    }
    return sum;
}

RT_DEV float powi5(float a) {
    float a2 = a * a, a4 = a2 * a2;
    return a * a4;
}
RT_DEV float schlick(float cosine, float ior) {  // types.rs:165-169
    float r0 = (1.0f - ior) / (1.0f + ior);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powi5(1.0f - cosine);
}

/* ------------------------------------------------------------------ one bounce of ray_color */

// Texture of a material.  Word 7 of the DEVICE copy of a material record (rt_scene_upload) holds the kind of
// its texture + 1, and for a solid colour words 2..4 hold that colour: most bounces then need no texture fetch
// (one link less in the slot -> material -> texture chain of dependent loads).
template <bool STATS, int EXT = EXT_ALL>
RT_DEV V3 material_texture(const DScene &sc, float4 ma, float4 mb, float u, float v, V3 p, Counters &cnt, bool have_turb = false, float turb = 0.0f) {
    if (f2i(mb.w) == RTG_TEX_SOLID + 1) {
        if (STATS) cnt.tex_evals[RTG_TEX_SOLID] += 1;
        return v3(ma.z, ma.w, mb.x);
    }
    return texture_value<STATS, EXT>(sc, f2i(ma.y), u, v, p, cnt, have_turb, turb);
}

struct Path {
    Ray ray;
    V3 throughput;   // product of the weights so far
    V3 radiance;     // sum of throughput * emitted
    int depth;       // remaining depth (camera.rs:258)
};

// Everything ray_color does after world.hit() returned `best` (camera.rs:264-308).  Returns true
// while the path continues (ps.ray is then the scattered ray and ps.depth is one less).
// A path that ends on black (depth used up, camera.rs:260; an absorbed ray or the back of an emitter: `emitted` is black).  The
// recursion multiplies that black by every attenuation factor on its way up, so a factor that is NaN or infinite (a light's
// pdf_value from inside the light, sphere.rs:95-99) makes the sample NaN, not black: the forward form has to say so as well.
RT_DEV void end_on_black(Path &ps) { ps.radiance = ps.radiance + ps.throughput * v3(0.0f, 0.0f, 0.0f); }

template <bool STATS, int EXT>
RT_DEV bool shade_material(const DScene &sc, Path &ps, const Rec &rec, float4 ma, float4 mb, rts::Rng &rng, Counters &cnt, bool have_turb, float turb);

// The hit record of `best` and its material record.  Returns false for a miss (the background is added, camera.rs:305).
template <bool STATS, int EXT = EXT_ALL>
RT_DEV bool shade_record(const DScene &sc, Path &ps, const Hit &best, V3 background, Rec &rec, float4 &ma, float4 &mb, Counters &cnt) {
    if (best.id < 0) {
        ps.radiance = ps.radiance + ps.throughput * background;
        return false;
    }
    if (STATS) cnt.accepted += 1;
    if ((EXT & EXT_MEDIA) && best.slot < 0) {  // ConstantMedium record (constant_medium.rs:85-93)
        uint32_t m = (uint32_t)(-best.slot - 1);
        float4 mb = sc.media[2 * m + 1];
        rec.t = best.t;
        rec.p = ps.ray.at(best.t);
        rec.normal = v3(1.0f, 0.0f, 0.0f);
        rec.front_face = true;
        const int outer = f2i(mb.w) == 1 ? f2i(sc.media[2 * m].z) - 1 : -1;
        if (outer >= 0) {      // a medium under wrappers of its own: the point is made in their frame and taken out through them
            const float4 xa = sc.xforms[2 * outer], xb = sc.xforms[2 * outer + 1];
            rec.p = to_object_rec<EXT>(sc, xa, xb, ps.ray).at(best.t);
            record_out<EXT>(sc, xa, xb, ps.ray, rec.p, rec.normal, rec.front_face);
        }
        rec.material = f2i(mb.y);
        rec.u = 0.0f; rec.v = 0.0f;
        int tex = f2i(sc.materials[2 * rec.material].y);
        if (texture_needs_uv<EXT>(sc, tex)) {  // u,v of the boundary's first hit
            float4 ma = sc.media[2 * m];
            float t1;
            V3 c = v3(ma.x, ma.y, ma.z);
            // (a general boundary with a texture that reads u, v is refused at upload)
            if (f2i(mb.w) != 1 && sphere_roots(c, ma.w, ps.ray, -RT_F32_MAX, RT_F32_MAX, &t1)) sphere_uv((ps.ray.at(t1) - c) / ma.w, &rec.u, &rec.v);
        }
    } else {
        int mat = f2i(sc.slots[4 * (size_t)best.slot + 1].y);
        float4 ma = sc.materials[2 * mat];
        int tkind = f2i(sc.materials[2 * mat + 1].w);   // 0 untextured, kind + 1 otherwise
        bool want_uv = tkind > RTG_TEX_SOLID + 1 && texture_needs_uv<EXT>(sc, f2i(ma.y));
        RG_BEGIN(RG_RECORD);
        rec = slot_record<EXT>(sc, (uint32_t)best.slot, ps.ray, best.t, want_uv, 0.001f, INFINITY);
        RG_END(RG_RECORD);
    }
    ma = sc.materials[2 * rec.material]; mb = sc.materials[2 * rec.material + 1];
    return true;
}

// ray_color's body for the lanes with `alive` (camera.rs:264-308); true while the path continues (ps.ray is then the scattered ray
// and ps.depth one less).  Called by ALL lanes of the wave: between the hit record and the material switch the wave evaluates the
// noise textures among its hits together (coop_turbulence).
template <bool STATS, int EXT = EXT_ALL>
RT_DEV bool shade_hit(const DScene &sc, bool alive, Path &ps, const Hit &best, V3 background, rts::Rng &rng, Counters &cnt) {
    Rec rec;
    rec.p = v3(0.0f, 0.0f, 0.0f); rec.normal = rec.p; rec.t = 0.0f; rec.u = 0.0f; rec.v = 0.0f; rec.front_face = false; rec.material = 0;
    float4 ma = make_float4(0.0f, 0.0f, 0.0f, 0.0f), mb = ma;
    bool hit = false;
    if (alive) hit = shade_record<STATS, EXT>(sc, ps, best, background, rec, ma, mb, cnt);
    bool have_turb = false;
    float turb = 0.0f;
    if ((EXT & EXT_NOISE) && sc.noise_coop) {
        const bool need = hit && f2i(mb.w) == RTG_TEX_NOISE + 1;
        const unsigned long long m = __ballot(need);
        if (m != 0ull && (uint32_t)__popcll(m) <= sc.noise_coop) {
            RG_BEGIN(RG_TEXTURE);
            int table = 0;
            if (need) table = f2i(sc.textures[2 * f2i(ma.y) + 1].x);
            turb = coop_turbulence(sc, m, rec.p, table);
            have_turb = need;
            RG_END(RG_TEXTURE);
        }
    }
    if (!hit) return false;
    return shade_material<STATS, EXT>(sc, ps, rec, ma, mb, rng, cnt, have_turb, turb);
}

template <bool STATS, int EXT>
RT_DEV bool shade_material(const DScene &sc, Path &ps, const Rec &rec, float4 ma, float4 mb, rts::Rng &rng, Counters &cnt, bool have_turb, float turb) {
    int mtype = f2i(ma.x);
    if (STATS) cnt.shades[mtype < 5 ? mtype : 4] += 1;
    const Ray &r = ps.ray;
    switch (mtype) {
    case RTG_MAT_DIFFUSE_LIGHT: {  // diffuse_light.rs:43-56; scatter = None -> return emitted
        RG_BEGIN(RG_EMIT);
        if (rec.front_face) {
            V3 e = material_texture<STATS, EXT>(sc, ma, mb, rec.u, rec.v, rec.p, cnt, have_turb, turb);
            ps.radiance = ps.radiance + ps.throughput * e;
        } else {
            end_on_black(ps);
        }
        RG_END(RG_EMIT);
        return false;
    }
    case RTG_MAT_METAL: {  // metal.rs:21-36
        RG_BEGIN(RG_METAL);
        V3 reflected = reflect_unit_vector(r.d, normalize(rec.normal));
        V3 scattered = reflected + mb.y * random_in_unit_sphere(rng);
        RG_END(RG_METAL);
        if (!(dot(scattered, rec.normal) > 0.0f)) { end_on_black(ps); return false; }  // None -> emitted = 0
        ps.throughput = ps.throughput * v3(ma.z, ma.w, mb.x);
        ps.ray = Ray{rec.p, scattered, r.time};
        break;
    }
    case RTG_MAT_DIELECTRIC: {  // dielectric.rs:17-67
        RG_BEGIN(RG_DIEL);
        float ior = mb.z;
        float eta = rec.front_face ? 1.0f / ior : ior;
        V3 uv = normalize(r.d);
        float cos_theta = f32_min(dot(-uv, rec.normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        V3 dir;
        if (eta * sin_theta > 1.0f) {
            dir = reflect_unit_vector(uv, normalize(rec.normal));
        } else {
            float reflect_probability = schlick(cos_theta, eta);
            if (rts::random_real(rng) < reflect_probability) dir = reflect_unit_vector(uv, rec.normal);
            else dir = refract(uv, rec.normal, eta);
        }
        ps.throughput = ps.throughput * v3(1.0f, 1.0f, 1.0f);
        ps.ray = Ray{rec.p, dir, r.time};
        RG_END(RG_DIEL);
        break;
    }
    case RTG_MAT_ISOTROPIC: {  // isotropic.rs:26-35
        if (!(EXT & EXT_MEDIA)) return false;     // (only a ConstantMedium carries this material)
        RG_BEGIN(RG_ISO);
        V3 dir = random_in_unit_sphere(rng);
        V3 att = material_texture<STATS, EXT>(sc, ma, mb, rec.u, rec.v, rec.p, cnt, have_turb, turb);
        ps.throughput = ps.throughput * att;
        ps.ray = Ray{rec.p, dir, rec.t};  // sic: time = hit t
        RG_END(RG_ISO);
        break;
    }
    default: {  // Lambertian: lambertian.rs:33-51 + camera.rs:271-301
        RG_BEGIN(RG_LAMBERT);
        RG_BEGIN(RG_TEXTURE);
        V3 albedo = material_texture<STATS, EXT>(sc, ma, mb, rec.u, rec.v, rec.p, cnt, have_turb, turb);
        RG_END(RG_TEXTURE);
        Onb uvw = onb_from(rec.normal);
        // MixturePdf::generate (pdf.rs:72-78): the lights' direction or the cosine one.  A sphere light (sphere.rs:107-112 +
        // types.rs:153-163) and the cosine lobe (types.rs:141-151) are the same computation — two draws, one sincos, two square
        // roots, a change of basis — with different arguments, so the lanes of both kinds run it together instead of one kind
        // after the other; every lane still makes its own draws in its own order and evaluates its own expressions.
        V3 dir = v3(1.0f, 0.0f, 0.0f);          // (a lights entry with the trait default, hittable.rs:51-53)
        int lobe = 0;                           // 0 cosine lobe, 1 sphere light, 2 anything else in `lights`
        Onb basis = uvw;
        float radius = 0.0f, distance_squared = 1.0f;
        RG_BEGIN(RG_L_LIGHTS);
        if (rts::random_real(rng) < 0.5f) {     // lights.random(rec.p): hittable_list.rs:71-75 then the object's random()
            lobe = 2;
            int idx;
            if ((EXT & EXT_CHAIN) && f2i(sc.lights[0].x) == 253) {   // lists in the list: every level draws its member
                idx = 0;
                do idx = f2i(sc.lights[2 * idx].y) + rts::random_int(rng, 0, f2i(sc.lights[2 * idx].z) - 1);
                while (f2i(sc.lights[2 * idx].x) == 254);
            } else
                idx = rts::random_int(rng, 0, (int)sc.n_lights - 1);
            const float4 la = sc.lights[2 * idx], lb = sc.lights[2 * idx + 1];
            const int type = f2i(la.x);
            // :221-226 for a MovingSphere, whose target is center(time1) - center(time0) (sic); cylinder.rs:152-161 for a
            // Cylinder: a sphere of its radius around the centre of its box
            if (type == RT_PRIM_SPHERE || type == RT_PRIM_MOVING_SPHERE || ((EXT & EXT_QUADRICS) && type == RT_PRIM_CYLINDER)) {
                const V3 direction = (type == RT_PRIM_SPHERE ? v3(la.y, la.z, la.w) : v3(lb.y, lb.z, lb.w)) - rec.p;
                distance_squared = length_squared(direction);
                basis = onb_from(direction);
                radius = lb.x;
                lobe = 1;
            } else if (type != 255) {           // rectangles.rs:69-77 / 139-147 / 209-217: a0,a1,b0,b1,k = la.y la.z la.w lb.x lb.y
                const float ra = rts::random_real_range(rng, la.y, la.z);
                const float rb = rts::random_real_range(rng, la.w, lb.x);
                const V3 pt = type == RT_PRIM_XY_RECT ? v3(ra, rb, lb.y) : (type == RT_PRIM_XZ_RECT ? v3(ra, lb.y, rb) : v3(lb.y, ra, rb));
                dir = pt - rec.p;
            }
        }
        RG_END(RG_L_LIGHTS);
        RG_BEGIN(RG_L_COSINE);
        if (lobe != 2) {
            const float r1 = rts::random_real(rng), r2 = rts::random_real(rng);
            const float root_a = sqrtf(lobe == 0 ? 1.0f - r2 : 1.0f - radius * radius / distance_squared);
            const float z = lobe == 0 ? root_a : 1.0f + r2 * (root_a - 1.0f);
            const float phi = (float)(2.0 * 3.14159265358979323846) * r1;
            float sn, cs;
            pm::sincosf_(phi, &sn, &cs);
            const float root_b = sqrtf(lobe == 0 ? r2 : 1.0f - z * z);
            dir = onb_local(basis, v3(cs * root_b, sn * root_b, z));
        }
        RG_END(RG_L_COSINE);
        // MixturePdf::value (pdf.rs:68-70)
        RG_BEGIN(RG_L_PDF);
        float light_val = lights_pdf_value<STATS, EXT>(sc, rec.p, dir, cnt);
        RG_END(RG_L_PDF);
        float cosine = dot(normalize(dir), uvw.w);
        float cos_val = cosine <= 0.0f ? 0.0f : cosine * (float)0.318309886183790671537767526745028724;
        float pdf_val = 0.5f * light_val + 0.5f * cos_val;
        if (fabsf(pdf_val) < 1.0E-5f) pdf_val = (__float_as_uint(pdf_val) >> 31) ? -1.0E-4f : 1.0E-4f;
        float sc_cos = dot(rec.normal, normalize(dir));
        float spdf = sc_cos < 0.0f ? 0.0f : sc_cos / 3.14159265358979323846f;
        ps.throughput = ps.throughput * ((albedo * spdf) / pdf_val);
        ps.ray = Ray{rec.p, dir, r.time};
        RG_END(RG_LAMBERT);
        break;
    }
    }
    ps.depth -= 1;
    return true;
}

/* ------------------------------------------------------------------ kernels */

struct RenderArgs {
    int32_t width, height, spp, spp_begin, spp_count, max_depth;
    float bg[3];
    uint32_t seed;
    int32_t x0, y0, x1, y1;
    int32_t projection;
    float psi_max, lambda_max;
    rt_camera cam;
};

// camera.rs:196-198 + 82-96
// One camera ray (camera.rs:196-249).  Perspective and Orthographic take a lens sample, so a pixel sample uses
// sampler slots 2s and 2s+1 (the reference's square and disk look-ups share one counter, sampling.rs:137-163);
// FishEye and SphericalPanoramic take none and use slot s.  FishEye returns false outside the unit disk: the
// sample then adds nothing and draws no ray time (camera.rs:221-233).
RT_DEV bool camera_ray(const DScene &sc, const RenderArgs &a, int x, int y, uint32_t pixel, uint32_t s, rts::Rng &rng, Ray *out) {
    const bool lens = a.projection == RT_PROJ_PERSPECTIVE || a.projection == RT_PROJ_ORTHOGRAPHIC;
    uint32_t i0 = rts::sampler_index(a.seed, pixel, lens ? 2u * s : s, sc.sets, sc.spp, sc.shuffled);
    float2 off = sc.samples[i0];
    float u = ((float)x + off.x) / (float)(a.width - 1);
    float v = 1.0f - ((float)y + off.y) / (float)(a.height - 1);
    V3 cu = v3(a.cam.u[0], a.cam.u[1], a.cam.u[2]), cv = v3(a.cam.v[0], a.cam.v[1], a.cam.v[2]), cw = v3(a.cam.w[0], a.cam.w[1], a.cam.w[2]);
    V3 origin = v3(a.cam.origin[0], a.cam.origin[1], a.cam.origin[2]);
    Ray r;
    if (lens) {
        uint32_t i1 = rts::sampler_index(a.seed, pixel, 2u * s + 1u, sc.sets, sc.spp, sc.shuffled);
        float2 dk = sc.disk[i1];
        float rdx = a.cam.lens_radius * dk.x, rdy = a.cam.lens_radius * dk.y;
        V3 llc = v3(a.cam.lower_left_corner[0], a.cam.lower_left_corner[1], a.cam.lower_left_corner[2]);
        V3 hor = v3(a.cam.horizontal[0], a.cam.horizontal[1], a.cam.horizontal[2]);
        V3 ver = v3(a.cam.vertical[0], a.cam.vertical[1], a.cam.vertical[2]);
        V3 offset = cu * rdx + cv * rdy;
        if (a.projection == RT_PROJ_PERSPECTIVE) {          // camera.rs:82-96
            r.o = origin + offset;
            r.d = llc + u * hor + v * ver - origin - offset;
        } else {                                            // camera.rs:98-112
            r.o = llc + u * hor + v * ver + offset;
            r.d = -cw;
        }
    } else {
        float pnx = 2.0f * u - 1.0f, pny = 2.0f * v - 1.0f;
        r.o = origin;
        if (a.projection == RT_PROJ_FISHEYE) {              // camera.rs:114-148
            float rsqr = pnx * pnx + pny * pny;
            if (!(rsqr <= 1.0f)) return false;
            float rr = sqrtf(rsqr);
            float psi = rr * a.psi_max;
            float sin_psi, cos_psi;
            pm::sincosf_(psi, &sin_psi, &cos_psi);
            float cos_alpha = pnx / rr, sin_alpha = pny / rr;
            r.d = (sin_psi * cos_alpha) * cu + (sin_psi * sin_alpha) * cv - cos_psi * cw;
        } else {                                            // camera.rs:150-184
            float lambda = pnx * a.lambda_max, psi = pny * a.psi_max;
            float phi = 3.14159265358979323846f - lambda;
            float theta = 0.5f * 3.14159265358979323846f - psi;
            float sin_theta, cos_theta, sin_phi, cos_phi;
            pm::sincosf_(theta, &sin_theta, &cos_theta);
            pm::sincosf_(phi, &sin_phi, &cos_phi);
            r.d = (sin_theta * sin_phi) * cu + cos_theta * cv + (sin_theta * cos_phi) * cw;
        }
    }
    r.time = rts::random_real(rng);
    *out = r;
    return true;
}

// Lane-occupancy telemetry; call where the wave is converged.
template <bool STATS>
RT_DEV void phase_count(Counters &cnt, int ph, bool active) {
    if (STATS) {
        const unsigned long long m = __ballot(active);
        if (active) {
            cnt.phase_lanes[ph] += 1;
            if ((m & (0ull - m)) == (1ull << (threadIdx.x & 63u))) cnt.phase_slots[ph] += 64;
        }
    }
}

RT_DEV uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Parity hook: world.hit without media for caller-supplied rays (rt_trace_first_hit).
__global__ void __launch_bounds__(256) first_hit_kernel(DScene sc, const float *__restrict__ rays, int n, float t_min, float t_max,
                                                        uint32_t *prim_id, float *t_out, float *normal3, uint8_t *front_face, float *uv2,
                                                        uint32_t lds_depth, StackEntry *ovf, uint32_t ovf_stride) {
    extern __shared__ StackEntry lds_stack[];
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    Ray r;
    r.o = v3(rays[i], rays[(size_t)n + i], rays[2 * (size_t)n + i]);
    r.d = v3(rays[3 * (size_t)n + i], rays[4 * (size_t)n + i], rays[5 * (size_t)n + i]);
    r.time = rays[6 * (size_t)n + i];
    Counters cnt{};
    rts::Rng rng{0, 1};
    Hit best;
    const Stack stack{(LdsStackPtr)lds_stack + threadIdx.x, blockDim.x, lds_depth, ovf ? ovf + i : nullptr, ovf_stride};   // ovf == null: the whole stack fits LDS
    bool h = closest_hit<false>(sc, r, t_min, t_max, false, rng, best, stack, cnt);
    Rec rec{};
    if (h) rec = slot_record(sc, (uint32_t)best.slot, r, best.t, true, t_min, t_max);
    prim_id[i] = h ? (uint32_t)best.id : RT_MISS;
    t_out[i] = h ? rec.t : 0.0f;
    normal3[3 * i] = h ? rec.normal.x : 0.0f; normal3[3 * i + 1] = h ? rec.normal.y : 0.0f; normal3[3 * i + 2] = h ? rec.normal.z : 0.0f;
    front_face[i] = h && rec.front_face ? 1 : 0;
    if (uv2) { uv2[2 * i] = h ? rec.u : 0.0f; uv2[2 * i + 1] = h ? rec.v : 0.0f; }
}

// main.rs:1363-1388
__global__ void __launch_bounds__(256) resolve_kernel(const float4 *__restrict__ sum, int n, float inv_spp, float4 *__restrict__ out) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    float4 s = sum[i], o;
    o.x = !pm::is_normal_(s.x) ? 0.0f : f32_clamp(sqrtf(s.x * inv_spp), 0.0f, 1.0f);
    o.y = !pm::is_normal_(s.y) ? 0.0f : f32_clamp(sqrtf(s.y * inv_spp), 0.0f, 1.0f);
    o.z = !pm::is_normal_(s.z) ? 0.0f : f32_clamp(sqrtf(s.z * inv_spp), 0.0f, 1.0f);
    o.w = s.w;
    out[i] = o;
}

// main.rs:1363-1388 followed by image 0.24.6 `to_rgb8` (f32 -> u8: round(clamp(c, 0, 1) * 255), halves away from zero)
RT_DEV uint32_t to_u8(float sum, float inv_spp) {
    const float c = !pm::is_normal_(sum) ? 0.0f : f32_clamp(sqrtf(sum * inv_spp), 0.0f, 1.0f);
    const float x = c * 255.0f, tr = truncf(x);   // f32::round, halves away from zero (x >= 0); not floor(x + 0.5): 0.49999997 + 0.5
    return (uint32_t)(x - tr >= 0.5f ? tr + 1.0f : tr);   // rounds up to 1.  x - tr is exact
}
__global__ void __launch_bounds__(256) resolve_rgb8_kernel(const float4 *__restrict__ sum, int n, float inv_spp, uint8_t *__restrict__ out) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const float4 s = sum[i];
    out[3 * (size_t)i] = (uint8_t)to_u8(s.x, inv_spp);
    out[3 * (size_t)i + 1] = (uint8_t)to_u8(s.y, inv_spp);
    out[3 * (size_t)i + 2] = (uint8_t)to_u8(s.z, inv_spp);
}

}  // namespace rtk
#endif
