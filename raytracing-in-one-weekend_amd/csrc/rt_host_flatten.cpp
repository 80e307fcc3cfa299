// rt_host_flatten.cpp — lower the object tree to flat arrays + a traversal BVH.
//
// This is the step the reference does not have: its scene is a tree of `Arc<dyn Hittable>`
// (raytracer/src/hittable.rs:44-54) walked by virtual calls.  Here every leaf primitive
// becomes one 64-byte slot; wrappers (flip_face.rs, transform.rs:9-153) become per-slot
// flags; `BvhNode`s (bvh.rs) and `HittableList`s dissolve, because the kernels use their own
// BVH over ALL leaves and reproduce the list semantics (closest hit, later object wins ties,
// hittable_list.rs:39-50) by primitive id.  Layouts: include/rt_api.h.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <functional>
#include <numeric>

#include "rt_host_internal.h"

using namespace rth;

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; ++a) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; }
    }
    void add(const float p[3]) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    void add(const Box &b) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Slot {
    float w[RT_SLOT_FLOATS];
};
struct Attr {
    float w[RT_ATTR_FLOATS];
};

inline float i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

// One xform record's worth of wrappers: Translate(RotateY(.)), either of the two alone, or one general Transform.
struct WrapOp {
    bool has_rot = false, has_trans = false;
    float sin_t = 0, cos_t = 1;
    float off[3] = {0, 0, 0};
    int matrix = -1;         // general Transform (transform.rs:155-213): index into rth_flat::matrices
};
// The wrappers between the world and a leaf, outermost first.  Any nesting of Translate / RotateY / Transform is a
// sequence of WrapOps: a RotateY directly under a Translate shares that Translate's op (the Cornell-box pattern, one
// record as before); everything else opens a new op, and more than one op becomes a chain record (rt_api.h).
struct Wrap {
    bool flip = false;       // parity of FlipFace wrappers with no Translate above them
    bool any_trans = false;
    std::vector<WrapOp> ops;
};

}  // namespace

struct rth_flat {
    std::vector<Slot> slots;
    std::vector<Attr> attrs;
    std::vector<float> nodes;  // RT_NODE_FLOATS each
    std::vector<float> xforms, matrices, media, lights, materials, textures;
    std::vector<uint32_t> perlin;
    std::vector<uint8_t> texels;
    std::vector<uint32_t> moving, rects;
    std::vector<float> slot_boxes;   // device-BVH mode: lo xyz, hi xyz per slot
    bool device_bvh = false;
    int32_t root_ref = RT_BVH_EMPTY;
    uint32_t stack_depth = 1, always_begin = 0, always_count = 0, n_prims = 0;
    bool has_tris = false;
    bool saw_bvh = false;            // `world` holds a BvhNode somewhere (rt_scene_desc::far_limit)
    float r_min = INFINITY;          // smallest sphere radius
};

namespace {

struct Flattener {
    const rtg_graph *g;
    rth_flat *out;
    std::vector<Slot> prims;   // in primitive-id order (before BVH reordering)
    std::vector<Attr> pattrs;
    std::vector<Box> boxes;
    std::vector<uint8_t> always;   // per primitive: tested for every ray, never through the BVH (rt_api.h rt_prim_type)
    uint32_t next_id = 0;
    std::string err;
    // the leaves of a ConstantMedium boundary that is not a plain sphere: slots behind the always-list, reached only through
    // their medium's record; they take no primitive id (rt_scene_graph.h) and appear in no tree or list
    std::vector<Slot> boundary;
    bool boundary_mode = false;
    static constexpr size_t kMaxOps = 64;  // xform records per chain (rt_api.h; the walk itself stops at 64 levels of nesting)

    bool fail(const std::string &m) {
        if (err.empty()) err = m;
        return false;
    }

    static void op_record(const WrapOp &w, float rec[RT_REC_FLOATS]) {
        for (int i = 0; i < RT_REC_FLOATS; ++i) rec[i] = 0.0f;
        if (w.matrix >= 0) {
            rec[0] = i2f(4); rec[1] = i2f(w.matrix);
        } else {
            rec[0] = i2f((w.has_rot ? 1 : 0) | (w.has_trans ? 2 : 0));
            rec[1] = w.sin_t; rec[2] = w.cos_t;
            rec[3] = w.off[0]; rec[4] = w.off[1]; rec[5] = w.off[2];
        }
    }
    // index of `n` consecutive records equal to `recs` in out->xforms, appended if there is no such run yet
    int xform_run(const float *recs, size_t n) {
        const size_t len = n * RT_REC_FLOATS;
        for (size_t i = 0; i + len <= out->xforms.size(); i += RT_REC_FLOATS)
            if (memcmp(&out->xforms[i], recs, len * sizeof(float)) == 0) return (int)(i / RT_REC_FLOATS);
        out->xforms.insert(out->xforms.end(), recs, recs + len);
        return (int)((out->xforms.size() - len) / RT_REC_FLOATS);
    }
    int xform_index(const Wrap &w) {
        if (w.ops.empty()) return -1;
        std::vector<float> recs(w.ops.size() * RT_REC_FLOATS);
        for (size_t k = 0; k < w.ops.size(); ++k) op_record(w.ops[k], &recs[k * RT_REC_FLOATS]);
        const int first = xform_run(recs.data(), w.ops.size());
        if (w.ops.size() == 1) return first;
        // chain record: the ops of the run are applied to the ray outermost first, to the hit record in reverse
        float rec[RT_REC_FLOATS] = {i2f(8), i2f(first), i2f((int32_t)w.ops.size())};
        return xform_run(rec, 1);
    }

    // Object-space box -> world box through the wrappers, innermost first; inside one op RotateY then Translate (the
    // wrappers' own bounding_box code is not followed: any conservative box serves the traversal).  The box of the eight
    // transformed corners is taken after every op, so it contains what the ops make of any point of the object's box.
    Box to_world(const Box &b, const Wrap &wr) {
        Box cur = b;
        for (size_t k = wr.ops.size(); k-- > 0;) {
            const WrapOp &w = wr.ops[k];
            Box r;
            r.reset();
            for (int i = 0; i < 8; ++i) {
                float p[3] = {(i & 1) ? cur.hi[0] : cur.lo[0], (i & 2) ? cur.hi[1] : cur.lo[1], (i & 4) ? cur.hi[2] : cur.lo[2]};
                if (w.matrix >= 0) {
                    const float *m = &out->matrices[(size_t)w.matrix * RT_MATRIX_FLOATS + 12];   // obj2world
                    float q[3];
                    for (int a = 0; a < 3; ++a) q[a] = m[4 * a] * p[0] + m[4 * a + 1] * p[1] + m[4 * a + 2] * p[2] + m[4 * a + 3];
                    for (int a = 0; a < 3; ++a) p[a] = q[a];
                }
                if (w.has_rot) {
                    float x = w.cos_t * p[0] + w.sin_t * p[2], z = -w.sin_t * p[0] + w.cos_t * p[2];
                    p[0] = x; p[2] = z;
                }
                if (w.has_trans) for (int a = 0; a < 3; ++a) p[a] += w.off[a];
                r.add(p);
            }
            cur = r;
        }
        return cur;
    }

    void emit(int type, const Wrap &w, int material, const float q0[4], float scalar, const float q2[4], const Box &obj_box,
              const Attr *attr) {
        Slot s{};
        for (int i = 0; i < 4; ++i) { s.w[i] = q0[i]; s.w[8 + i] = q2 ? q2[i] : 0.0f; }
        int xf = xform_index(w);
        s.w[4] = i2f(type | (w.flip ? 16 : 0) | ((xf + 1) << 8));
        s.w[5] = i2f(material);
        s.w[6] = i2f(boundary_mode ? -1 : (int32_t)next_id++);
        s.w[7] = scalar;
        if (boundary_mode) { boundary.push_back(s); return; }
        Box wb = to_world(obj_box, w);
        // Conservative padding: the slab test must never cull a primitive whose own hit test
        // accepts the ray (DESIGN.md "BVH conservativeness").
        float m = 1.0f;
        for (int a = 0; a < 3; ++a) m = std::max(m, std::max(std::fabs(wb.lo[a]), std::fabs(wb.hi[a])));
        float pad = 1.0e-4f * m;
        for (int a = 0; a < 3; ++a) { wb.lo[a] -= pad; wb.hi[a] += pad; }
        prims.push_back(s);
        always.push_back(type >= RT_PRIM_CYLINDER ? 1 : 0);   // quadrics may accept t < t_min (cone.rs:61-65): no box test for them
        Attr a{};
        if (attr) a = *attr;
        pattrs.push_back(a);
        boxes.push_back(wb);
    }

    void emit_rect(int gtype, const float *f, const Wrap &w, int material) {
        // rectangles.rs: a = first in-plane axis, b = second, k = plane offset
        float q0[4] = {f[0], f[1], f[2], f[3]};
        float k = f[4];
        Box b;
        int type;
        if (gtype == RTG_XY_RECT) { type = RT_PRIM_XY_RECT; b = Box{{f[0], f[2], k}, {f[1], f[3], k}}; }
        else if (gtype == RTG_XZ_RECT) { type = RT_PRIM_XZ_RECT; b = Box{{f[0], k, f[2]}, {f[1], k, f[3]}}; }
        else { type = RT_PRIM_YZ_RECT; b = Box{{k, f[0], f[2]}, {k, f[1], f[3]}}; }
        if (!boundary_mode) out->rects.push_back((uint32_t)prims.size());
        emit(type, w, material, q0, k, nullptr, b, nullptr);
    }

    bool walk(int32_t idx, Wrap w, int depth) {
        if (idx < 0 || (uint32_t)idx >= g->n_nodes) return fail("node index out of range");
        if (depth > 64) return fail("scene graph nesting too deep");
        const rtg_node &n = g->nodes[idx];
        const int32_t *ch = g->children + n.child_begin;
        switch (n.type) {
        case RTG_SPHERE: {
            float q0[4] = {n.f[0], n.f[1], n.f[2], n.f[3]};
            float r = std::fabs(n.f[3]);
            if (r > 0.0f && r < out->r_min) out->r_min = r;
            Box b{{n.f[0] - r, n.f[1] - r, n.f[2] - r}, {n.f[0] + r, n.f[1] + r, n.f[2] + r}};
            emit(RT_PRIM_SPHERE, w, n.material, q0, 0.0f, nullptr, b, nullptr);
            return true;
        }
        case RTG_MOVING_SPHERE: {
            float r = std::fabs(n.f[8]);
            if (r > 0.0f && r < out->r_min) out->r_min = r;
            float q0[4] = {n.f[0], n.f[1], n.f[2], n.f[8]};
            float q2[4] = {n.f[3], n.f[4], n.f[5], n.f[7]};
            Box b;
            b.reset();
            for (int e = 0; e < 2; ++e) {  // box over the shutter interval; rays with a time outside it
                const float *c = n.f + 3 * e;  // are handled by the kernels' moving-sphere list
                float lo[3] = {c[0] - r, c[1] - r, c[2] - r}, hi[3] = {c[0] + r, c[1] + r, c[2] + r};
                b.add(lo); b.add(hi);
            }
            out->moving.push_back((uint32_t)prims.size());  // primitive order; remapped after the BVH build
            emit(RT_PRIM_MOVING_SPHERE, w, n.material, q0, n.f[6], q2, b, nullptr);
            return true;
        }
        case RTG_XY_RECT: case RTG_XZ_RECT: case RTG_YZ_RECT:
            emit_rect(n.type, n.f, w, n.material);
            return true;
        case RTG_BLOCK: {  // block.rs:27-79: side order XY@z1, XY@z0, XZ@y1, XZ@y0, YZ@x1, YZ@x0
            const float *p0 = n.f, *p1 = n.f + 3;
            float xy1[5] = {p0[0], p1[0], p0[1], p1[1], p1[2]}, xy0[5] = {p0[0], p1[0], p0[1], p1[1], p0[2]};
            float xz1[5] = {p0[0], p1[0], p0[2], p1[2], p1[1]}, xz0[5] = {p0[0], p1[0], p0[2], p1[2], p0[1]};
            float yz1[5] = {p0[1], p1[1], p0[2], p1[2], p1[0]}, yz0[5] = {p0[1], p1[1], p0[2], p1[2], p0[0]};
            emit_rect(RTG_XY_RECT, xy1, w, n.material); emit_rect(RTG_XY_RECT, xy0, w, n.material);
            emit_rect(RTG_XZ_RECT, xz1, w, n.material); emit_rect(RTG_XZ_RECT, xz0, w, n.material);
            emit_rect(RTG_YZ_RECT, yz1, w, n.material); emit_rect(RTG_YZ_RECT, yz0, w, n.material);
            return true;
        }
        case RTG_CYLINDER: case RTG_PARABOLOID: {   // cylinder.rs:41, paraboloid.rs:41-46: box = [-r, r]^2 x [zmin, zmax]
            float q0[4] = {n.f[0], n.f[1], n.f[2], n.f[3]};
            Box b{{-n.f[0], -n.f[0], n.f[1]}, {n.f[0], n.f[0], n.f[2]}};
            emit(n.type == RTG_CYLINDER ? RT_PRIM_CYLINDER : RT_PRIM_PARABOLOID, w, n.material, q0, 0.0f, nullptr, b, nullptr);
            return true;
        }
        case RTG_CONE: {                            // cone.rs:123-128
            float q0[4] = {n.f[0], n.f[1], 0.0f, n.f[2]};
            Box b{{-n.f[0], -n.f[0], 0.0f}, {n.f[0], n.f[0], n.f[1]}};
            emit(RT_PRIM_CONE, w, n.material, q0, 0.0f, nullptr, b, nullptr);
            return true;
        }
        case RTG_HYPERBOLOID: {                     // hyperboloid.rs:26-76
            float d[11];
            if (!hyperboloid_derive(n.f, n.f + 3, d)) return fail("Hyperboloid without finite coefficients");
            float q0[4] = {d[0], d[1], d[2], n.f[6]}, q2[4] = {d[3], d[4], d[5], d[7]};
            Box b{{-d[8], -d[8], d[6]}, {d[8], d[8], d[7]}};
            emit(RT_PRIM_HYPERBOLOID, w, n.material, q0, d[6], q2, b, nullptr);
            prims.back().w[12] = d[9]; prims.back().w[13] = d[10];   // ah, ch
            return true;
        }
        case RTG_TRANSFORM: {
            if (n.child_count != 1) return fail("Transform needs one child");
            if (w.ops.size() >= kMaxOps) return fail("more than 64 nested Translate / RotateY / Transform wrappers");
            M4 o2w = m4_identity(), w2o;
            memcpy(o2w.a, n.f, 12 * sizeof(float));
            if (!m4_invert(o2w, &w2o)) return fail("Matrix is not invertible");          // transform.rs:166
            M4 n2w = m4_transpose(m4_adjoint(w2o));                                       // transform.rs:171
            WrapOp op;
            op.matrix = (int)(out->matrices.size() / RT_MATRIX_FLOATS);
            w.ops.push_back(op);
            out->matrices.insert(out->matrices.end(), w2o.a, w2o.a + 12);
            out->matrices.insert(out->matrices.end(), o2w.a, o2w.a + 12);
            out->matrices.insert(out->matrices.end(), n2w.a, n2w.a + 12);
            return walk(ch[0], w, depth + 1);
        }
        case RTG_FLIP_FACE:
            // A FlipFace under a Translate is overwritten by Translate's HitRecord::new
            // (transform.rs:36-44), so only flips with no Translate above them count.  RotateY and Transform keep the
            // inner front_face (transform.rs:136-143, 198-202), so a flip on either side of them counts.
            if (n.child_count != 1) return fail("FlipFace needs one child");
            if (!w.any_trans) w.flip = !w.flip;
            return walk(ch[0], w, depth + 1);
        case RTG_TRANSLATE:
            if (n.child_count != 1) return fail("Translate needs one child");
        {
            if (w.ops.size() >= kMaxOps) return fail("more than 64 nested Translate / RotateY / Transform wrappers");
            WrapOp op;
            op.has_trans = true;
            op.off[0] = n.f[0]; op.off[1] = n.f[1]; op.off[2] = n.f[2];
            w.ops.push_back(op);
            w.any_trans = true;
            return walk(ch[0], w, depth + 1);
        }
        case RTG_ROTATE_Y: {
            if (n.child_count != 1) return fail("RotateY needs one child");
            // directly under a Translate: that Translate's op takes the rotation too (one record for Translate(RotateY(.)))
            const bool joins = !w.ops.empty() && w.ops.back().has_trans && !w.ops.back().has_rot && w.ops.back().matrix < 0;
            if (!joins) {
                if (w.ops.size() >= kMaxOps) return fail("more than 64 nested Translate / RotateY / Transform wrappers");
                w.ops.push_back(WrapOp{});
            }
            WrapOp &op = w.ops.back();
            if (n.i[0] == 1) { op.sin_t = n.f[1]; op.cos_t = n.f[2]; }   // rth_add_rotate_y_sincos: the host's own values
            else {
                float rads = (n.f[0] * 3.14159265358979323846f) / 180.0f;  // types.rs:44-46
                pm::sincosf_(rads, &op.sin_t, &op.cos_t);
            }
            op.has_rot = true;
            return walk(ch[0], w, depth + 1);
        }
        case RTG_BVH: case RTG_LIST:
            if (n.type == RTG_BVH) out->saw_bvh = true;
            for (int i = 0; i < n.child_count; ++i)
                if (!walk(ch[i], w, depth + 1)) return false;
            return true;
        case RTG_CONSTANT_MEDIUM: {
            if (n.child_count != 1) return fail("ConstantMedium needs one boundary");
            // (a FlipFace around a medium turns its record's front_face, which Isotropic — the one material a medium has — never reads)
            const rtg_node &b = g->nodes[ch[0]];
            if (b.type == RTG_SPHERE && w.ops.empty()) {
                float rec[RT_REC_FLOATS] = {b.f[0], b.f[1], b.f[2], b.f[3], -1.0f / n.f[0] /* constant_medium.rs:24 */,
                                            i2f(n.material), i2f((int32_t)next_id++), 0.0f};
                out->media.insert(out->media.end(), rec, rec + RT_REC_FLOATS);
                return true;
            }
            // any other boundary (constant_medium.rs:52-58 takes a Hittable; scene_cornell_box_smoke, main.rs:642-680, uses
            // Translate(RotateY(Block))), and any medium under wrappers of its own: the boundary is ONE Block / rectangle / sphere
            // under any nesting of Translate / RotateY / Transform / FlipFace.  Its leaves become boundary slots that the medium's
            // record points at — each under the medium's wrappers followed by its own, so boundary.hit() is the list fold over
            // them for the ray as rt_render has it (block.rs:111-113, hittable_list.rs:39-50; t is the same in every frame) —
            // and the record names the medium's own wrappers, in whose frame ConstantMedium::hit measures the ray's length and
            // makes its hit point (constant_medium.rs:76-93).
            {
                int32_t leaf = ch[0];
                for (int d = 0; d < 64; ++d) {
                    const rtg_node &k = g->nodes[leaf];
                    if (k.type == RTG_TRANSLATE || k.type == RTG_ROTATE_Y || k.type == RTG_FLIP_FACE || k.type == RTG_TRANSFORM) {
                        if (k.child_count != 1) return fail("wrapper needs one child");
                        leaf = g->children[k.child_begin];
                    } else break;
                }
                const int lt = g->nodes[leaf].type;
                if (!(lt == RTG_BLOCK || lt == RTG_SPHERE || lt == RTG_XY_RECT || lt == RTG_XZ_RECT || lt == RTG_YZ_RECT))
                    return fail("ConstantMedium boundary must be a Sphere, Block or rectangle, bare or under Translate / RotateY / Transform / FlipFace");
            }
            const uint32_t id = next_id++;
            const size_t first = boundary.size();
            Wrap outer = w;
            outer.flip = false;
            const int outer_xf = xform_index(outer);
            boundary_mode = true;
            const bool ok = walk(ch[0], outer, depth + 1);
            boundary_mode = false;
            if (!ok) return false;
            float rec[RT_REC_FLOATS] = {i2f((int32_t)first) /* + the number of other slots, added when the slot array is laid out */,
                                        i2f((int32_t)(boundary.size() - first)), i2f(outer_xf + 1), 0.0f, -1.0f / n.f[0], i2f(n.material), i2f((int32_t)id), i2f(1)};
            out->media.insert(out->media.end(), rec, rec + RT_REC_FLOATS);
            return true;
        }
        case RTG_TRIANGLE_MESH: {
            if (n.i[0] < 0 || (uint32_t)n.i[0] >= g->n_meshes) return fail("mesh index out of range");
            const rtg_mesh &m = g->meshes[n.i[0]];
            out->has_tris = true;
            for (uint32_t nd = 0; nd < m.n_nodes; ++nd)
                for (uint32_t k = 0; k + 2 < m.nodes[nd].index_count; k += 3) {
                    const uint32_t *ix = m.indices + m.nodes[nd].index_begin + k;
                    const rtg_vertex &a = m.vertices[ix[0]], &b = m.vertices[ix[1]], &c = m.vertices[ix[2]];
                    float q0[4] = {a.pos[0], a.pos[1], a.pos[2], b.pos[0]};
                    float q2[4] = {b.pos[1], b.pos[2], c.pos[0], c.pos[1]};
                    Box bx;
                    bx.reset();
                    bx.add(a.pos); bx.add(b.pos); bx.add(c.pos);
                    Attr at{};
                    memcpy(at.w + 0, a.normal, 12); memcpy(at.w + 3, b.normal, 12); memcpy(at.w + 6, c.normal, 12);
                    memcpy(at.w + 9, a.uv, 8); memcpy(at.w + 11, b.uv, 8); memcpy(at.w + 13, c.uv, 8);
                    emit(RT_PRIM_TRIANGLE, w, n.material, q0, c.pos[2], q2, bx, &at);
                }
            return true;
        }
        default:
            return fail("unknown node type");
        }
    }

    // lights list: pdf_value / random are forwarded WITHOUT transforming by Translate / RotateY
    // (transform.rs:48-54, 146-152); FlipFace, Block, Bvh, ... keep the trait defaults
    // (hittable.rs:47-53: pdf 0, random = unit_x) -> light type 255.
    // One record of `lights` for node idx; a nested list's members are queued and laid out behind the records made so far.
    bool light_record(int32_t idx, int depth, float rec[RT_REC_FLOATS], int32_t *list_node) {
        if (idx < 0 || (uint32_t)idx >= g->n_nodes || depth > 64) return fail("bad light node");
        const rtg_node &n = g->nodes[idx];
        for (int i = 0; i < RT_REC_FLOATS; ++i) rec[i] = 0.0f;
        *list_node = -1;
        switch (n.type) {
        case RTG_SPHERE:
            rec[0] = i2f(RT_PRIM_SPHERE); rec[1] = n.f[0]; rec[2] = n.f[1]; rec[3] = n.f[2]; rec[4] = n.f[3];
            break;
        case RTG_XY_RECT: case RTG_XZ_RECT: case RTG_YZ_RECT:
            rec[0] = i2f(n.type == RTG_XY_RECT ? RT_PRIM_XY_RECT : n.type == RTG_XZ_RECT ? RT_PRIM_XZ_RECT : RT_PRIM_YZ_RECT);
            for (int i = 0; i < 5; ++i) rec[1 + i] = n.f[i];
            break;
        case RTG_TRANSLATE: case RTG_ROTATE_Y: case RTG_TRANSFORM:   // transform.rs:48-54, 146-152, 206-212
            if (n.child_count != 1) return fail("wrapper needs one child");
            return light_record(g->children[n.child_begin], depth + 1, rec, list_node);
        case RTG_MOVING_SPHERE: {   // objects/sphere.rs:204-226: pdf_value tests the ray at time 0, i.e. against center(0); random aims at
            // center(time1) - center(time0) (sic: the difference of the two centres, not a centre).  Both points are computed here
            // with the reference's expression (sphere.rs:144-147), so the kernels only ever see a sphere at one place and one target
            const float *c0 = n.f, *c1 = n.f + 3;
            const float time0 = n.f[6], time1 = n.f[7];
            auto center = [&](float time, float out[3]) {
                const float s = (time - time0) / (time1 - time0);
                for (int a = 0; a < 3; ++a) out[a] = c0[a] + s * (c1[a] - c0[a]);
            };
            float at0[3], a1[3], a0[3];
            center(0.0f, at0); center(time1, a1); center(time0, a0);
            rec[0] = i2f(RT_PRIM_MOVING_SPHERE);
            rec[1] = at0[0]; rec[2] = at0[1]; rec[3] = at0[2]; rec[4] = n.f[8];
            for (int a = 0; a < 3; ++a) rec[5 + a] = a1[a] - a0[a];
            break;
        }
        case RTG_CYLINDER: {   // cylinder.rs:134-161: pdf_value from its own hit() and area(); random() aims at aabb.center()
            // like a sphere of the cylinder's radius (aabb3.rs:52-54: (min + max) * 0.5)
            const float radius = n.f[0], zmin = n.f[1], zmax = n.f[2];
            rec[0] = i2f(RT_PRIM_CYLINDER);
            rec[1] = zmin; rec[2] = zmax; rec[3] = n.f[3]; rec[4] = radius;
            rec[5] = (-radius + radius) * 0.5f; rec[6] = (-radius + radius) * 0.5f; rec[7] = (zmin + zmax) * 0.5f;
            break;
        }
        case RTG_LIST:         // hittable_list.rs:59-75: the mean of its members' pdfs; random() draws one of them
            if (n.child_count < 1) return fail("an empty list in `lights` (the reference indexes objects[random_int(0,-1)], hittable_list.rs:71-75)");
            rec[0] = i2f(254);
            *list_node = idx;
            break;
        default:               // FlipFace, Block, Bvh, ...: the trait defaults (hittable.rs:47-53: pdf 0, random = unit_x)
            rec[0] = i2f(255);
            break;
        }
        return true;
    }

    // lights list: pdf_value / random are forwarded WITHOUT transforming by Translate / RotateY / Transform.  Without nested
    // lists: one record per member, as before.  With them: a root record (253), then run after run, each list's members together.
    bool lights(int32_t root) {
        struct Pending { int32_t node; size_t rec; int depth; };   // a list whose members still have to be laid out; its record
        std::vector<Pending> todo;
        std::vector<float> recs;
        auto lay_out = [&](int32_t list, int depth) -> bool {
            const rtg_node &n = g->nodes[list];
            if (depth > 4) return fail("`lights`: lists nested more than 4 deep");
            for (int i = 0; i < n.child_count; ++i) {
                float rec[RT_REC_FLOATS];
                int32_t sub;
                if (!light_record(g->children[n.child_begin + i], 0, rec, &sub)) return false;
                if (sub >= 0) todo.push_back({sub, recs.size() / RT_REC_FLOATS, depth + 1});
                recs.insert(recs.end(), rec, rec + RT_REC_FLOATS);
            }
            return true;
        };
        if (!lay_out(root, 1)) return false;
        if (!todo.empty()) {                                        // nested: start again behind a root record
            const int32_t n_top = g->nodes[root].child_count;
            todo.clear(); recs.clear();
            float rec[RT_REC_FLOATS] = {i2f(253), i2f(1), i2f(n_top)};
            recs.insert(recs.end(), rec, rec + RT_REC_FLOATS);
            if (!lay_out(root, 1)) return false;
            for (size_t k = 0; k < todo.size(); ++k) {              // grows while it is walked
                const Pending p = todo[k];
                const size_t first = recs.size() / RT_REC_FLOATS;
                if (!lay_out(p.node, p.depth)) return false;
                recs[p.rec * RT_REC_FLOATS + 1] = i2f((int32_t)first);
                recs[p.rec * RT_REC_FLOATS + 2] = i2f(g->nodes[p.node].child_count);
            }
        }
        out->lights = recs;
        return true;
    }
};

/* ------------------------------------------------------------------ BVH (binned SAH) */

struct BuildNode {
    Box box[2];
    int32_t ref[2];
};

struct BvhBuilder {
    const std::vector<Box> &boxes;
    std::vector<uint32_t> order;        // primitive indices, leaves are contiguous runs
    std::vector<float> cx, cy, cz;      // centroids
    std::vector<BuildNode> nodes;
    std::vector<std::pair<uint32_t, uint32_t>> leaves;  // (first, count) in `order`
    uint32_t max_depth = 0;
    static constexpr int kBins = 16, kDepthCap = 44;
    // Primitives per leaf.  A leaf visit is the divergent, expensive step of the GPU walk (different
    // primitive kinds and counts per lane), a node visit is cheap and uniform: measured on MI355X,
    // one primitive per leaf is fastest for the sphere/rect scenes, two for the 2.26M-triangle mesh.
    int kLeafMax = 1;

    explicit BvhBuilder(const std::vector<Box> &b) : boxes(b) {
        kLeafMax = b.size() >= 100000 ? 2 : 1;
        if (const char *e = getenv("RT_KNOBS") ? getenv("RTH_BVH_LEAF_MAX") : nullptr) { int v = atoi(e); if (v >= 1 && v <= 4) kLeafMax = v; }
    }

    Box range_box(uint32_t b, uint32_t e) const {
        Box r;
        r.reset();
        for (uint32_t i = b; i < e; ++i) r.add(boxes[order[i]]);
        return r;
    }
    float centroid(uint32_t prim, int axis) const { return axis == 0 ? cx[prim] : axis == 1 ? cy[prim] : cz[prim]; }

    // returns a ref: >=0 inner node, <0 leaf (~ref = leaf index here; remapped to slots later)
    int32_t build(uint32_t b, uint32_t e, uint32_t depth) {
        max_depth = std::max(max_depth, depth);
        uint32_t n = e - b;
        if (n <= (uint32_t)kLeafMax) {
            leaves.emplace_back(b, n);
            return ~(int32_t)(leaves.size() - 1);
        }
        // centroid bounds
        float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t i = b; i < e; ++i)
            for (int a = 0; a < 3; ++a) {
                float c = centroid(order[i], a);
                clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c);
            }
        uint32_t mid = 0;
        bool split_found = false;
        // remaining depth budget: fall back to an exact median split when SAH could overrun it
        uint32_t need = 0;
        for (uint32_t m = (n + kLeafMax - 1) / kLeafMax; m > 1; m = (m + 1) / 2) ++need;
        bool force_median = depth + need + 2 >= (uint32_t)kDepthCap;
        int best_axis = -1;
        if (!force_median) {
            float best_cost = FLT_MAX;
            int best_bin = -1;
            for (int a = 0; a < 3; ++a) {
                float ext = chi[a] - clo[a];
                if (!(ext > 0.0f)) continue;
                Box bin_box[kBins];
                uint32_t bin_n[kBins] = {};
                for (auto &bb : bin_box) bb.reset();
                float scale = (float)kBins / ext;
                for (uint32_t i = b; i < e; ++i) {
                    int k = std::min(kBins - 1, (int)((centroid(order[i], a) - clo[a]) * scale));
                    bin_box[k].add(boxes[order[i]]);
                    ++bin_n[k];
                }
                float right_area[kBins];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int k = kBins - 1; k > 0; --k) {
                    if (bin_n[k]) acc.add(bin_box[k]);
                    right_area[k] = bin_n[k] || cnt ? acc.half_area() : 0.0f;
                    cnt += bin_n[k];
                }
                acc.reset();
                uint32_t left_n = 0;
                for (int k = 0; k < kBins - 1; ++k) {
                    if (bin_n[k]) acc.add(bin_box[k]);
                    left_n += bin_n[k];
                    uint32_t right_n = n - left_n;
                    if (left_n == 0 || right_n == 0) continue;
                    float cost = acc.half_area() * (float)left_n + right_area[k + 1] * (float)right_n;
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
                }
            }
            if (best_axis >= 0) {
                float ext = chi[best_axis] - clo[best_axis];
                float scale = (float)kBins / ext;
                auto it = std::partition(order.begin() + b, order.begin() + e, [&](uint32_t p) {
                    int k = std::min(kBins - 1, (int)((centroid(p, best_axis) - clo[best_axis]) * scale));
                    return k <= best_bin;
                });
                mid = (uint32_t)(it - order.begin());
                split_found = mid > b && mid < e;
            }
        }
        if (!split_found) {  // median along the widest centroid axis
            int a = 0;
            for (int k = 1; k < 3; ++k)
                if (chi[k] - clo[k] > chi[a] - clo[a]) a = k;
            mid = b + n / 2;
            std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e,
                             [&](uint32_t p, uint32_t q) { return centroid(p, a) < centroid(q, a); });
        }
        int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Box lb = range_box(b, mid), rb = range_box(mid, e);
        int32_t l = build(b, mid, depth + 1);
        int32_t r = build(mid, e, depth + 1);
        nodes[me].box[0] = lb; nodes[me].box[1] = rb;
        nodes[me].ref[0] = l; nodes[me].ref[1] = r;
        return me;
    }
};

}  // namespace

extern "C" {

int rth_flatten(const rtg_graph *g, rth_flat **out_flat) { return rth_flatten_opts(g, 0u, out_flat); }

int rth_flatten_opts(const rtg_graph *g, uint32_t flags, rth_flat **out_flat) {
    if (!g || !out_flat) { set_error("null argument"); return RT_E_INVALID; }
    if (g->world_root < 0 || g->lights_root < 0) { set_error("scene roots are not set"); return RT_E_INVALID; }
    auto flat = std::make_unique<rth_flat>();
    Flattener fl{g, flat.get(), {}, {}, {}, {}, 0, {}, {}, false};
    if (!fl.walk(g->world_root, Wrap{}, 0)) {
        set_error("flatten: %s", fl.err.c_str());
        return RT_E_UNSUPPORTED_PRIMITIVE;
    }
    if (!fl.lights(g->lights_root)) {
        set_error("flatten: %s", fl.err.c_str());
        return RT_E_UNSUPPORTED_PRIMITIVE;
    }
    flat->n_prims = fl.next_id;

    // tables
    for (uint32_t i = 0; i < g->n_materials; ++i) {
        const rtg_material &m = g->materials[i];
        float rec[RT_REC_FLOATS] = {i2f(m.type), i2f(m.tex), m.color[0], m.color[1], m.color[2], m.fuzz, m.ior, 0.0f};
        flat->materials.insert(flat->materials.end(), rec, rec + RT_REC_FLOATS);
    }
    std::vector<uint64_t> image_offset(g->n_images);
    for (uint32_t i = 0; i < g->n_images; ++i) {
        image_offset[i] = flat->texels.size();
        size_t bytes = (size_t)g->images[i].width * g->images[i].height * 4;
        flat->texels.insert(flat->texels.end(), g->images[i].rgba, g->images[i].rgba + bytes);
    }
    if (flat->texels.size() > 0x7fffffffull * 4ull) { set_error("flatten: texel atlas too large"); return RT_E_INVALID; }
    for (uint32_t i = 0; i < g->n_textures; ++i) {
        const rtg_texture &t = g->textures[i];
        float rec[RT_REC_FLOATS] = {i2f(t.type), t.color[0], t.color[1], t.color[2], 0, 0, 0, 0};
        switch (t.type) {
        case RTG_TEX_CHECKER: rec[4] = i2f(t.odd); rec[5] = i2f(t.even); rec[6] = t.scale; break;
        case RTG_TEX_NOISE:   rec[4] = i2f(t.perlin); rec[6] = t.scale; break;
        case RTG_TEX_IMAGE:
            if (t.image >= 0) {
                rec[4] = i2f((int32_t)(image_offset[t.image] / 4));  // texel (RGBA8 word) offset
                rec[5] = i2f(g->images[t.image].width);
                rec[7] = i2f(g->images[t.image].height);
            } else {
                rec[4] = i2f(-1);  // blank() -> cyan
            }
            break;
        default: break;
        }
        flat->textures.insert(flat->textures.end(), rec, rec + RT_REC_FLOATS);
    }
    for (uint32_t i = 0; i < g->n_perlins; ++i) {
        const rtg_perlin &p = g->perlins[i];
        size_t base = flat->perlin.size();
        flat->perlin.resize(base + RT_PERLIN_WORDS);
        memcpy(&flat->perlin[base], p.ranvec, 256 * 3 * 4);
        memcpy(&flat->perlin[base + 768], p.perm_x, 1024);
        memcpy(&flat->perlin[base + 1024], p.perm_y, 1024);
        memcpy(&flat->perlin[base + 1280], p.perm_z, 1024);
    }

    // BVH + slot order
    const size_t n_all = fl.prims.size();
    std::vector<uint32_t> new_index(n_all);
    std::vector<uint32_t> in_bvh, in_always;   // primitive indices
    for (size_t i = 0; i < n_all; ++i) (fl.always[i] ? in_always : in_bvh).push_back((uint32_t)i);
    const size_t n = in_bvh.size();
    size_t flat_max = 24;   // measured on MI355X: Cornell box (13 leaves) and Blocks (20) are faster as a flat scan (Blocks: 10.5 vs 9.1 Grays/s)
    if (const char *e = getenv("RT_KNOBS") ? getenv("RTH_FLAT_MAX") : nullptr) { int v = atoi(e); if (v >= 0 && v <= 4096) flat_max = (size_t)v; }
    if (n <= flat_max) {  // tiny scenes (Cornell box: 13 leaves): a flat scan beats any tree
        flat->slots = fl.prims;
        flat->attrs = fl.pattrs;
        std::iota(new_index.begin(), new_index.end(), 0u);
        flat->always_begin = 0;
        flat->always_count = (uint32_t)n_all;
        flat->root_ref = RT_BVH_EMPTY;
        flat->stack_depth = 1;
    } else if (flags & RTH_FLATTEN_DEVICE_BVH) {   // no tree here: slots in primitive order, one box each, the always-list behind them
        flat->slots.resize(n_all);
        flat->attrs.resize(n_all);
        flat->slot_boxes.resize(6 * n);
        for (size_t i = 0; i < n_all; ++i) {
            const uint32_t prim = i < n ? in_bvh[i] : in_always[i - n];
            flat->slots[i] = fl.prims[prim];
            flat->attrs[i] = fl.pattrs[prim];
            new_index[prim] = (uint32_t)i;
            if (i < n) { memcpy(&flat->slot_boxes[6 * i], fl.boxes[prim].lo, 12); memcpy(&flat->slot_boxes[6 * i + 3], fl.boxes[prim].hi, 12); }
        }
        flat->device_bvh = true;
        flat->always_begin = (uint32_t)n;
        flat->always_count = (uint32_t)in_always.size();
        flat->root_ref = RT_BVH_EMPTY;
        flat->stack_depth = 1;
    } else {
        std::vector<Box> bvh_boxes(n);
        for (size_t i = 0; i < n; ++i) bvh_boxes[i] = fl.boxes[in_bvh[i]];
        BvhBuilder bb(bvh_boxes);
        bb.order.resize(n);
        std::iota(bb.order.begin(), bb.order.end(), 0u);
        bb.cx.resize(n); bb.cy.resize(n); bb.cz.resize(n);
        for (size_t i = 0; i < n; ++i) {
            bb.cx[i] = 0.5f * (bvh_boxes[i].lo[0] + bvh_boxes[i].hi[0]);
            bb.cy[i] = 0.5f * (bvh_boxes[i].lo[1] + bvh_boxes[i].hi[1]);
            bb.cz[i] = 0.5f * (bvh_boxes[i].lo[2] + bvh_boxes[i].hi[2]);
        }
        bb.nodes.reserve(n);
        int32_t root = bb.build(0, (uint32_t)n, 0);
        flat->slots.resize(n_all);
        flat->attrs.resize(n_all);
        for (size_t i = 0; i < n_all; ++i) {   // BVH leaves in leaf order, then the always-list in primitive order
            const uint32_t prim = i < n ? in_bvh[bb.order[i]] : in_always[i - n];
            flat->slots[i] = fl.prims[prim];
            flat->attrs[i] = fl.pattrs[prim];
            new_index[prim] = (uint32_t)i;
        }
        auto leaf_ref = [&](int32_t r) -> int32_t {
            if (r >= 0) return r;
            auto lf = bb.leaves[~r];
            return ~(int32_t)((lf.first << 3) | (lf.second - 1));
        };
        // Collapse the binary tree into a 4-wide one: a node adopts the children of its largest inner
        // child until it has four.  The GPU walk is bound by the latency of dependent node fetches, and a
        // 4-wide node halves the number of fetches on a ray's way down.
        struct Child { Box box; int32_t ref; };
        uint32_t max_depth4 = 0;
        std::function<int32_t(int32_t, uint32_t)> collapse = [&](int32_t inner, uint32_t depth) -> int32_t {
            max_depth4 = std::max(max_depth4, depth);
            std::vector<Child> ch = {{bb.nodes[inner].box[0], bb.nodes[inner].ref[0]}, {bb.nodes[inner].box[1], bb.nodes[inner].ref[1]}};
            while (ch.size() < 4) {
                int best = -1;
                float best_area = -1.0f;
                for (size_t i = 0; i < ch.size(); ++i)
                    if (ch[i].ref >= 0 && ch[i].box.half_area() > best_area) { best = (int)i; best_area = ch[i].box.half_area(); }
                if (best < 0) break;
                const BuildNode &bn = bb.nodes[ch[best].ref];
                ch[best] = {bn.box[0], bn.ref[0]};
                ch.push_back({bn.box[1], bn.ref[1]});
            }
            const size_t me = flat->nodes.size() / RT_NODE_FLOATS;
            flat->nodes.resize((me + 1) * RT_NODE_FLOATS, 0.0f);
            // Child boxes on the 8-bit grid of the node's box (a visit then fetches 64 instead of 128 bytes: the
            // walk's scattered node fetches keep the texture path busier than its arithmetic keeps the SIMDs).
            // Rounded outwards and checked in exact arithmetic, so culling stays conservative.
            Box parent;
            parent.reset();
            for (const Child &k : ch) parent.add(k.box);
            float origin[3], scale[3];
            uint32_t q[6] = {0, 0, 0, 0, 0, 0};   // lo.x hi.x lo.y hi.y lo.z hi.z, child c in byte c
            for (int a = 0; a < 3; ++a) {
                origin[a] = parent.lo[a];
                scale[a] = (parent.hi[a] - parent.lo[a]) / 255.0f;
                while ((double)origin[a] + (double)scale[a] * 255.0 < (double)parent.hi[a]) scale[a] = std::nextafter(scale[a], FLT_MAX);
                for (size_t c = 0; c < ch.size(); ++c) {
                    int ql = 0, qh = 255;
                    if (scale[a] > 0.0f) {
                        ql = (int)std::floor(((double)ch[c].box.lo[a] - (double)origin[a]) / (double)scale[a]);
                        qh = (int)std::ceil(((double)ch[c].box.hi[a] - (double)origin[a]) / (double)scale[a]);
                        ql = std::min(255, std::max(0, ql));
                        qh = std::min(255, std::max(0, qh));
                        while (ql > 0 && (double)origin[a] + (double)scale[a] * ql > (double)ch[c].box.lo[a]) --ql;
                        while (qh < 255 && (double)origin[a] + (double)scale[a] * qh < (double)ch[c].box.hi[a]) ++qh;
                    } else {
                        ql = qh = 0;   // flat node box along this axis: every child plane is the origin
                    }
                    q[2 * a] |= (uint32_t)ql << (8 * c);
                    q[2 * a + 1] |= (uint32_t)qh << (8 * c);
                }
            }
            {
                float *d = &flat->nodes[me * RT_NODE_FLOATS];
                for (int a = 0; a < 3; ++a) { d[a] = origin[a]; d[3 + a] = scale[a]; }
                for (int k = 0; k < 6; ++k) memcpy(d + 6 + k, &q[k], 4);
            }
            for (size_t c = 0; c < 4; ++c) {
                int32_t ref = RT_BVH_EMPTY;
                if (c < ch.size()) ref = ch[c].ref >= 0 ? collapse(ch[c].ref, depth + 1) : leaf_ref(ch[c].ref);
                flat->nodes[me * RT_NODE_FLOATS + 12 + c] = i2f(ref);   // (indexed afresh: the recursion grows the vector)
            }
            return (int32_t)me;
        };
        flat->nodes.clear();
        flat->root_ref = root >= 0 ? collapse(root, 1) : leaf_ref(root);
        flat->stack_depth = 3 * max_depth4 + 2;   // a visit pushes at most three children
        flat->always_begin = (uint32_t)n;
        flat->always_count = (uint32_t)in_always.size();
    }
    for (auto &m : flat->moving) m = new_index[m];
    for (auto &m : flat->rects) m = new_index[m];
    if (!fl.boundary.empty()) {     // boundary slots of media go behind everything else; their records learn where
        if (flags & RTH_FLATTEN_DEVICE_BVH) { set_error("flatten: a device-built BVH and a ConstantMedium over a non-sphere boundary are not supported together"); return RT_E_UNSUPPORTED_PRIMITIVE; }
        const size_t base = flat->slots.size();
        flat->slots.insert(flat->slots.end(), fl.boundary.begin(), fl.boundary.end());
        flat->attrs.resize(flat->slots.size());
        for (size_t m = 0; m * RT_REC_FLOATS < flat->media.size(); ++m) {
            float *rec = &flat->media[m * RT_REC_FLOATS];
            int32_t kind, first;
            memcpy(&kind, rec + 7, 4);
            if (kind != 1) continue;
            memcpy(&first, rec, 4);
            first += (int32_t)base;
            memcpy(rec, &first, 4);
        }
    }
    if (!flat->has_tris) flat->attrs.clear();
    *out_flat = flat.release();
    return RT_OK;
}

void rth_flat_free(rth_flat *f) { delete f; }

void rth_flat_desc(const rth_flat *f, rt_scene_desc *d) {
    memset(d, 0, sizeof *d);
    d->struct_size = sizeof *d;
    d->bvh_nodes = f->nodes.data();            d->n_bvh_nodes = (uint32_t)(f->nodes.size() / RT_NODE_FLOATS);
    d->bvh_root_ref = f->root_ref;             d->bvh_stack_depth = f->stack_depth;
    d->slots = f->slots.empty() ? nullptr : f->slots[0].w;   d->n_slots = (uint32_t)f->slots.size();
    d->always_begin = f->always_begin;         d->always_count = f->always_count;
    d->tri_attrs = f->attrs.empty() ? nullptr : f->attrs[0].w;
    d->xforms = f->xforms.data();              d->n_xforms = (uint32_t)(f->xforms.size() / RT_REC_FLOATS);
    d->matrices = f->matrices.data();          d->n_matrices = (uint32_t)(f->matrices.size() / RT_MATRIX_FLOATS);
    d->slot_boxes = f->slot_boxes.data();      d->build_bvh_on_device = f->device_bvh ? 1u : 0u;
    d->media = f->media.data();                d->n_media = (uint32_t)(f->media.size() / RT_REC_FLOATS);
    d->lights = f->lights.data();              d->n_lights = (uint32_t)(f->lights.size() / RT_REC_FLOATS);
    d->materials = f->materials.data();        d->n_materials = (uint32_t)(f->materials.size() / RT_REC_FLOATS);
    d->textures = f->textures.data();          d->n_textures = (uint32_t)(f->textures.size() / RT_REC_FLOATS);
    d->perlin = f->perlin.data();              d->n_perlin = (uint32_t)(f->perlin.size() / RT_PERLIN_WORDS);
    d->texels = f->texels.data();              d->n_texel_bytes = f->texels.size();
    d->moving_slots = f->moving.data();        d->n_moving_slots = (uint32_t)f->moving.size();
    d->rect_slots = f->rects.data();           d->n_rect_slots = (uint32_t)f->rects.size();
    d->n_prims = f->n_prims;
    d->far_limit = f->saw_bvh ? INFINITY : f->r_min * 1024.0f;
}

}  // extern "C"
