// rt_host_scene.cpp — scene graph storage, the reference's constructors and scene builders.
//
// Restates (not copies) raytracer/src/main.rs:106-199 (RandomWorld), :362-522 (CornellBox),
// :685-828 (Chapter2Final), :969-1029 (MeshTest), perlin.rs:65-82, triangle_mesh.rs:43-72 and
// the math helpers they use (math/src/mat4.rs, quat.rs).  Every random_* call of the
// reference builders is drawn, in the reference's order, from a seeded host stream
// (include/rt_seed_schedule.h) because the reference itself is unseeded.
#include "rt_host_internal.h"

namespace rth {

static thread_local std::string g_error;

void set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

M4 m4_identity() {
    M4 m{};
    m.at(0, 0) = m.at(1, 1) = m.at(2, 2) = m.at(3, 3) = 1.0f;
    return m;
}

M4 m4_mul(const M4 &l, const M4 &r) {
    M4 res{};
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col)
            for (int k = 0; k < 4; ++k) res.at(row, col) += l.at(row, k) * r.at(k, col);
    return res;
}

M4 m4_transpose(const M4 &m) {
    M4 t{};
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) t.at(r, c) = m.at(c, r);
    return t;
}

#define A(r, c) m.at(r, c)
float m4_det(const M4 &m) {
    // Laplace expansion exactly as written in mat4.rs:542-560, including the two index
    // typos in l2 and l4 (harmless for affine matrices).
    float k1 = A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0);
    float l1 = A(2, 2) * A(3, 3) - A(2, 3) * A(3, 2);
    float k2 = A(0, 0) * A(1, 2) - A(0, 2) * A(1, 0);
    float l2 = A(2, 1) * A(3, 3) - A(1, 2) * A(3, 1);
    float k3 = A(0, 0) * A(1, 3) - A(0, 3) * A(1, 0);
    float l3 = A(2, 1) * A(3, 2) - A(2, 2) * A(3, 1);
    float k4 = A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1);
    float l4 = A(2, 0) * A(3, 3) - A(3, 2) * A(3, 0);
    float k5 = A(0, 1) * A(1, 3) - A(0, 3) * A(1, 1);
    float l5 = A(2, 0) * A(3, 2) - A(2, 2) * A(3, 0);
    float k6 = A(0, 2) * A(1, 3) - A(0, 3) * A(1, 2);
    float l6 = A(2, 0) * A(3, 1) - A(2, 1) * A(3, 0);
    return k1 * l1 - k2 * l2 + k3 * l3 + k4 * l4 - k5 * l5 + k6 * l6;
}

M4 m4_adjoint(const M4 &m) {
    float m1 = A(2, 2) * A(3, 3) - A(2, 3) * A(3, 2);
    float m2 = A(2, 1) * A(3, 3) - A(2, 3) * A(3, 1);
    float m3 = A(2, 1) * A(3, 2) - A(2, 2) * A(3, 1);
    float m4 = A(0, 2) * A(1, 3) - A(0, 3) * A(1, 2);
    float m5 = A(0, 1) * A(1, 3) - A(0, 3) * A(1, 1);
    float m6 = A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1);
    float m7 = A(2, 0) * A(3, 3) - A(2, 3) * A(3, 0);
    float m8 = A(2, 0) * A(3, 2) - A(2, 2) * A(3, 0);
    float m9 = A(1, 2) * A(3, 3) - A(1, 3) * A(3, 2);
    float m10 = A(1, 0) * A(3, 3) - A(1, 3) * A(3, 0);
    float m11 = A(1, 0) * A(3, 2) - A(1, 2) * A(3, 0);
    float m12 = A(0, 0) * A(1, 3) - A(0, 3) * A(1, 0);
    float m13 = A(0, 0) * A(1, 2) - A(0, 2) * A(1, 0);
    float m14 = A(2, 0) * A(3, 1) - A(2, 1) * A(3, 0);
    float m15 = A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0);
    float m16 = A(2, 0) * A(3, 1) - A(2, 1) * A(3, 0);
    M4 r{};
    r.at(0, 0) = A(1, 1) * m1 - A(1, 2) * m2 + A(1, 3) * m3;
    r.at(0, 1) = -A(0, 1) * m1 + A(0, 2) * m2 - A(0, 3) * m3;
    r.at(0, 2) = A(3, 1) * m4 - A(3, 2) * m5 + A(3, 3) * m6;
    r.at(0, 3) = -A(2, 1) * m4 + A(2, 2) * m5 - A(2, 3) * m6;
    r.at(1, 0) = -A(1, 0) * m1 + A(1, 2) * m7 - A(1, 3) * m8;
    r.at(1, 1) = A(0, 0) * m1 - A(0, 2) * m7 + A(0, 3) * m8;
    r.at(1, 2) = -A(0, 0) * m9 + A(0, 2) * m10 - A(0, 3) * m11;
    r.at(1, 3) = A(2, 0) * m4 - A(2, 2) * m12 + A(2, 3) * m13;
    r.at(2, 0) = A(1, 0) * m2 - A(1, 1) * m7 + A(1, 3) * m16;
    r.at(2, 1) = -A(0, 0) * m2 + A(0, 1) * m7 - A(0, 3) * m16;
    r.at(2, 2) = A(3, 0) * m5 - A(3, 1) * m12 + A(3, 3) * m15;
    r.at(2, 3) = -A(2, 0) * m5 + A(2, 1) * m12 - A(2, 3) * m15;
    r.at(3, 0) = -A(1, 0) * m3 + A(1, 1) * m8 - A(1, 2) * m14;
    r.at(3, 1) = A(0, 0) * m3 - A(0, 1) * m8 + A(0, 2) * m14;
    r.at(3, 2) = -A(3, 0) * m6 + A(3, 1) * m13 - A(3, 2) * m15;
    r.at(3, 3) = A(2, 0) * m6 - A(2, 1) * m13 + A(2, 2) * m15;
    return r;
}
#undef A

bool m4_invert(const M4 &m, M4 *out) {
    float d = m4_det(m);
    if (d == 0.0f) return false;  // the reference asserts (mat4.rs:610)
    float inv = 1.0f / d;
    M4 adj = m4_adjoint(m);
    for (int i = 0; i < 16; ++i) out->a[i] = adj.a[i] * inv;
    return true;
}

V3 m4_transform_point(const M4 &m, V3 p) {
    return {m.at(0, 0) * p.x + m.at(0, 1) * p.y + m.at(0, 2) * p.z + m.at(0, 3),
            m.at(1, 0) * p.x + m.at(1, 1) * p.y + m.at(1, 2) * p.z + m.at(1, 3),
            m.at(2, 0) * p.x + m.at(2, 1) * p.y + m.at(2, 2) * p.z + m.at(2, 3)};
}

V3 m4_transform_vector(const M4 &m, V3 p) {
    return {m.at(0, 0) * p.x + m.at(0, 1) * p.y + m.at(0, 2) * p.z,
            m.at(1, 0) * p.x + m.at(1, 1) * p.y + m.at(1, 2) * p.z,
            m.at(2, 0) * p.x + m.at(2, 1) * p.y + m.at(2, 2) * p.z};
}

M4 m4_translate(V3 t) {
    M4 m = m4_identity();
    m.at(0, 3) = t.x;
    m.at(1, 3) = t.y;
    m.at(2, 3) = t.z;
    return m;
}

M4 m4_scale(V3 s) {
    M4 m = m4_identity();
    m.at(0, 0) = s.x;
    m.at(1, 1) = s.y;
    m.at(2, 2) = s.z;
    return m;
}

M4 quat_to_rotation_matrix(float w, float x, float y, float z) {
    float s = (1.0f + 1.0f) / (w * w + x * x + y * y + z * z);
    M4 m = m4_identity();
    m.at(0, 0) = 1.0f - s * (y * y + z * z);
    m.at(0, 1) = s * (x * y - w * z);
    m.at(0, 2) = s * (x * z + w * y);
    m.at(1, 0) = s * (x * y + w * z);
    m.at(1, 1) = 1.0f - s * (x * x + z * z);
    m.at(1, 2) = s * (y * z - w * x);
    m.at(2, 0) = s * (x * z - w * y);
    m.at(2, 1) = s * (y * z + w * x);
    m.at(2, 2) = 1.0f - s * (x * x + y * y);
    return m;
}

void quat_axis_angle(float angle_deg, V3 axis, float q[4]) {
    float lensq = length_squared(axis);
    if (lensq == 0.0f) {
        q[0] = 1.0f; q[1] = q[2] = q[3] = 0.0f;
        return;
    }
    // f32::to_radians = x * (PI / 180)
    float angle = (angle_deg * (3.14159265358979323846f / 180.0f)) / (1.0f + 1.0f);
    float s, c;
    pm::sincosf_(angle, &s, &c);
    float k = s / std::sqrt(lensq);
    q[0] = c; q[1] = axis.x * k; q[2] = axis.y * k; q[3] = axis.z * k;
}

bool hyperboloid_derive(const float a1[3], const float a2[3], float out[11]) {
    V3 p1 = v3(a1[0], a1[1], a1[2]), p2 = v3(a2[0], a2[1], a2[2]);
    float radius1 = std::sqrt(p1.x * p1.x + p1.y * p1.y), radius2 = std::sqrt(p2.x * p2.x + p2.y * p2.y);
    float rmax = f32_max(radius1, radius2);
    float zmin = f32_min(p1.z, p2.z), zmax = f32_max(p1.z, p2.z);
    if (p2.z == 0.0f) { V3 t = p1; p1 = p2; p2 = t; }
    V3 pp = p1;
    float ah = 0.0f, ch = 0.0f;
    bool ok = false;
    for (int it = 0; it < 64 && !ok; ++it) {   // the reference loops until ah is finite (:42-54)
        pp = pp + 2.0f * (p2 - p1);
        float xy1 = pp.x * pp.x + pp.y * pp.y, xy2 = p2.x * p2.x + p2.y * p2.y;
        ah = (1.0f / xy1 - (pp.z * pp.z) / (xy1 * p2.z * p2.z)) / (1.0f - (xy2 * pp.z * pp.z) / (xy1 * p2.z * p2.z));
        ch = (ah * xy2 - 1.0f) / (p2.z * p2.z);
        ok = std::isfinite(ah);
    }
    if (!ok) return false;
    const float r[11] = {p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, zmin, zmax, rmax, ah, ch};
    memcpy(out, r, sizeof r);
    return true;
}

}  // namespace rth

using namespace rth;

extern "C" {

const char *rth_last_error(void) { return g_error.c_str(); }

rth_scene *rth_scene_new(void) { return new rth_scene(); }
void rth_scene_free(rth_scene *s) { delete s; }

const rtg_graph *rth_scene_graph(rth_scene *s) {
    if (!s) return nullptr;
    for (size_t i = 0; i < s->images.size(); ++i) s->images[i].rgba = s->image_pixels[i].data();
    s->mesh_views.resize(s->meshes.size());
    for (size_t i = 0; i < s->meshes.size(); ++i) {
        const Mesh &m = *s->meshes[i];
        s->mesh_views[i] = rtg_mesh{m.verts.data(), (uint32_t)m.verts.size(), m.indices.data(), (uint32_t)m.indices.size(),
                                    m.nodes.data(), (uint32_t)m.nodes.size()};
    }
    rtg_graph &g = s->view;
    g.nodes = s->nodes.data();           g.n_nodes = (uint32_t)s->nodes.size();
    g.children = s->children.data();     g.n_children = (uint32_t)s->children.size();
    g.materials = s->materials.data();   g.n_materials = (uint32_t)s->materials.size();
    g.textures = s->textures.data();     g.n_textures = (uint32_t)s->textures.size();
    g.perlins = s->perlins.data();       g.n_perlins = (uint32_t)s->perlins.size();
    g.images = s->images.data();         g.n_images = (uint32_t)s->images.size();
    g.meshes = s->mesh_views.data();     g.n_meshes = (uint32_t)s->mesh_views.size();
    g.world_root = s->world_root;
    g.lights_root = s->lights_root;
    return &g;
}

/* ------------------------------------------------------------------ textures */

static int push_texture(rth_scene *s, const rtg_texture &t) {
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}

int rth_add_solid_texture(rth_scene *s, float r, float g, float b) {
    rtg_texture t{};
    t.type = RTG_TEX_SOLID;
    t.color[0] = r; t.color[1] = g; t.color[2] = b;
    t.odd = t.even = t.perlin = t.image = -1;
    return push_texture(s, t);
}

int rth_add_checker_texture(rth_scene *s, int odd_tex, int even_tex, float repeat) {
    int n = (int)s->textures.size();
    if (odd_tex < 0 || odd_tex >= n || even_tex < 0 || even_tex >= n) {
        set_error("checker texture: child texture out of range");
        return RT_E_INVALID;
    }
    rtg_texture t{};
    t.type = RTG_TEX_CHECKER;
    t.odd = odd_tex; t.even = even_tex; t.scale = repeat;
    t.perlin = t.image = -1;
    return push_texture(s, t);
}

// perlin.rs:65-82.  rng.gen::<f32>() is a 24-bit uniform in [0,1); SliceRandom::shuffle is a
// Fisher-Yates pass from the top (rand 0.7.3 seq/mod.rs): for i in (1..n).rev() swap(i, gen_range(0, i+1)).
static void shuffle_i32(rts::Rng &rng, int32_t *v, int n) {
    for (int i = n - 1; i >= 1; --i) {
        int j = rts::random_int(rng, 0, i);
        int32_t t = v[i]; v[i] = v[j]; v[j] = t;
    }
}

int rth_add_noise_texture(rth_scene *s, float scale, uint32_t seed) {
    rts::Rng rng;
    rts::host_rng(rng, seed, 0x9e71u + (uint32_t)s->perlins.size());
    rtg_perlin p{};
    for (int i = 0; i < 256; ++i) {
        float x = -1.0f + 2.0f * rts::random_real(rng);
        float y = -1.0f + 2.0f * rts::random_real(rng);
        float z = -1.0f + 2.0f * rts::random_real(rng);
        V3 n = normalize(v3(x, y, z));
        p.ranvec[i][0] = n.x; p.ranvec[i][1] = n.y; p.ranvec[i][2] = n.z;
    }
    for (int i = 0; i < 256; ++i) p.perm_x[i] = p.perm_y[i] = p.perm_z[i] = i;
    shuffle_i32(rng, p.perm_x, 256);
    shuffle_i32(rng, p.perm_y, 256);
    shuffle_i32(rng, p.perm_z, 256);
    s->perlins.push_back(p);
    rtg_texture t{};
    t.type = RTG_TEX_NOISE;
    t.scale = scale;
    t.perlin = (int)s->perlins.size() - 1;
    t.odd = t.even = t.image = -1;
    return push_texture(s, t);
}

int rth_add_noise_texture_table(rth_scene *s, float scale, const rtg_perlin *table) {
    if (!table) { set_error("noise texture: null table"); return RT_E_INVALID; }
    for (int i = 0; i < 256; ++i)
        if ((uint32_t)table->perm_x[i] > 255u || (uint32_t)table->perm_y[i] > 255u || (uint32_t)table->perm_z[i] > 255u) {
            set_error("noise texture: permutation entry outside 0..255");
            return RT_E_INVALID;
        }
    s->perlins.push_back(*table);
    rtg_texture t{};
    t.type = RTG_TEX_NOISE;
    t.scale = scale;
    t.perlin = (int)s->perlins.size() - 1;
    t.odd = t.even = t.image = -1;
    return push_texture(s, t);
}

int rth_add_image_texture(rth_scene *s, int width, int height, const uint8_t *rgba) {
    rtg_texture t{};
    t.type = RTG_TEX_IMAGE;
    t.odd = t.even = t.perlin = -1;
    if (width <= 0 || height <= 0 || !rgba) {
        t.image = -1;  // ImageTexture::blank(): value() returns cyan
        return push_texture(s, t);
    }
    s->image_pixels.emplace_back(rgba, rgba + (size_t)width * height * 4);
    s->images.push_back(rtg_image{width, height, nullptr});
    t.image = (int)s->images.size() - 1;
    return push_texture(s, t);
}

/* ------------------------------------------------------------------ materials */

static int push_material(rth_scene *s, int type, int tex, float r, float g, float b, float fuzz, float ior) {
    if (tex >= (int)s->textures.size()) {
        set_error("material: texture %d out of range", tex);
        return RT_E_INVALID;
    }
    rtg_material m{};
    m.type = type; m.tex = tex;
    m.color[0] = r; m.color[1] = g; m.color[2] = b;
    m.fuzz = fuzz; m.ior = ior;
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}

int rth_add_lambertian(rth_scene *s, int albedo_tex) { return push_material(s, RTG_MAT_LAMBERTIAN, albedo_tex, 0, 0, 0, 0, 0); }
int rth_add_metal(rth_scene *s, float r, float g, float b, float fuzz) { return push_material(s, RTG_MAT_METAL, -1, r, g, b, fuzz, 0); }
int rth_add_dielectric(rth_scene *s, float ior) { return push_material(s, RTG_MAT_DIELECTRIC, -1, 0, 0, 0, 0, ior); }
int rth_add_diffuse_light(rth_scene *s, int emit_tex) { return push_material(s, RTG_MAT_DIFFUSE_LIGHT, emit_tex, 0, 0, 0, 0, 0); }
int rth_add_isotropic(rth_scene *s, int albedo_tex) { return push_material(s, RTG_MAT_ISOTROPIC, albedo_tex, 0, 0, 0, 0, 0); }

/* ------------------------------------------------------------------ hittables */

static int push_node(rth_scene *s, rtg_node n, const int32_t *children, int n_children) {
    for (int i = 0; i < n_children; ++i)
        if (children[i] < 0 || children[i] >= (int)s->nodes.size()) {
            set_error("node: child %d out of range", children[i]);
            return RT_E_INVALID;
        }
    if (n.material >= (int)s->materials.size()) {
        set_error("node: material %d out of range", n.material);
        return RT_E_INVALID;
    }
    n.child_begin = (int32_t)s->children.size();
    n.child_count = n_children;
    s->children.insert(s->children.end(), children, children + n_children);
    s->nodes.push_back(n);
    return (int)s->nodes.size() - 1;
}

int rth_add_sphere(rth_scene *s, const float c[3], float r, int mtl) {
    rtg_node n{};
    n.type = RTG_SPHERE; n.material = mtl;
    n.f[0] = c[0]; n.f[1] = c[1]; n.f[2] = c[2]; n.f[3] = r;
    return push_node(s, n, nullptr, 0);
}

int rth_add_moving_sphere(rth_scene *s, const float c0[3], const float c1[3], float t0, float t1, float r, int mtl) {
    rtg_node n{};
    n.type = RTG_MOVING_SPHERE; n.material = mtl;
    for (int i = 0; i < 3; ++i) { n.f[i] = c0[i]; n.f[3 + i] = c1[i]; }
    n.f[6] = t0; n.f[7] = t1; n.f[8] = r;
    return push_node(s, n, nullptr, 0);
}

int rth_add_rect(rth_scene *s, int type, float a0, float a1, float b0, float b1, float k, int mtl) {
    if (type != RTG_XY_RECT && type != RTG_XZ_RECT && type != RTG_YZ_RECT) {
        set_error("rect: bad type %d", type);
        return RT_E_INVALID;
    }
    rtg_node n{};
    n.type = type; n.material = mtl;
    n.f[0] = a0; n.f[1] = a1; n.f[2] = b0; n.f[3] = b1; n.f[4] = k;
    return push_node(s, n, nullptr, 0);
}

int rth_add_block(rth_scene *s, const float p0[3], const float p1[3], int mtl) {
    rtg_node n{};
    n.type = RTG_BLOCK; n.material = mtl;
    for (int i = 0; i < 3; ++i) { n.f[i] = p0[i]; n.f[3 + i] = p1[i]; }
    return push_node(s, n, nullptr, 0);
}

static int wrapper(rth_scene *s, int type, int child, const float *f, int nf) {
    rtg_node n{};
    n.type = type; n.material = -1;
    for (int i = 0; i < nf; ++i) n.f[i] = f[i];
    int32_t c = child;
    return push_node(s, n, &c, 1);
}

int rth_add_flip_face(rth_scene *s, int child) { return wrapper(s, RTG_FLIP_FACE, child, nullptr, 0); }
int rth_add_translate(rth_scene *s, int child, const float offset[3]) { return wrapper(s, RTG_TRANSLATE, child, offset, 3); }
int rth_add_rotate_y(rth_scene *s, int child, float degrees) { return wrapper(s, RTG_ROTATE_Y, child, &degrees, 1); }
// A RotateY as the reference holds it (transform.rs:57-62): sin and cos as the host computed them, not the angle
int rth_add_rotate_y_sincos(rth_scene *s, int child, float sin_theta, float cos_theta) {
    const float f[3] = {0.0f, sin_theta, cos_theta};
    int n = wrapper(s, RTG_ROTATE_Y, child, f, 3);
    if (n >= 0) s->nodes[n].i[0] = 1;
    return n;
}

int rth_add_transform(rth_scene *s, int child, const float obj2world[16]) {
    if (!obj2world) { set_error("transform: null matrix"); return RT_E_INVALID; }
    if (obj2world[12] != 0.0f || obj2world[13] != 0.0f || obj2world[14] != 0.0f || obj2world[15] != 1.0f) {
        set_error("transform: only affine matrices (last row 0,0,0,1) are supported");
        return RT_E_UNSUPPORTED_PRIMITIVE;
    }
    M4 m, inv;
    memcpy(m.a, obj2world, sizeof m.a);
    if (!m4_invert(m, &inv)) { set_error("transform: Matrix is not invertible"); return RT_E_INVALID; }
    return wrapper(s, RTG_TRANSFORM, child, obj2world, 12);
}

static int quadric(rth_scene *s, int type, int mtl, const float *f, int nf) {
    if (mtl < 0) { set_error("quadric: material %d out of range", mtl); return RT_E_INVALID; }
    rtg_node n{};
    n.type = type; n.material = mtl;
    for (int i = 0; i < nf; ++i) n.f[i] = f[i];
    return push_node(s, n, nullptr, 0);
}

int rth_add_cylinder(rth_scene *s, float radius, float z0, float z1, float phi_max, int mtl) {
    const float f[4] = {radius, f32_min(z0, z1), f32_max(z0, z1), phi_max};
    return quadric(s, RTG_CYLINDER, mtl, f, 4);
}

int rth_add_cone(rth_scene *s, float radius, float height, float phi_max, int mtl) {
    const float f[3] = {radius, height, phi_max};
    return quadric(s, RTG_CONE, mtl, f, 3);
}

int rth_add_paraboloid(rth_scene *s, float radius, float zmin, float zmax, float phi_max, int mtl) {
    const float f[4] = {radius, f32_min(zmin, zmax), f32_max(zmin, zmax), phi_max};
    return quadric(s, RTG_PARABOLOID, mtl, f, 4);
}

int rth_add_hyperboloid(rth_scene *s, const float p1[3], const float p2[3], float phi_max, int mtl) {
    float d[11];
    if (!p1 || !p2 || !hyperboloid_derive(p1, p2, d)) { set_error("hyperboloid: the implicit form has no finite coefficients"); return RT_E_INVALID; }
    const float f[7] = {p1[0], p1[1], p1[2], p2[0], p2[1], p2[2], phi_max};
    return quadric(s, RTG_HYPERBOLOID, mtl, f, 7);
}

int rth_add_bvh(rth_scene *s, const int32_t *children, int n) {
    if (n <= 0) { set_error("bvh: empty child list"); return RT_E_INVALID; }
    rtg_node node{};
    node.type = RTG_BVH; node.material = -1;
    return push_node(s, node, children, n);
}

int rth_add_list(rth_scene *s, const int32_t *children, int n) {
    rtg_node node{};
    node.type = RTG_LIST; node.material = -1;
    return push_node(s, node, children, n);
}

int rth_add_constant_medium(rth_scene *s, int boundary, int albedo_tex, float density) {
    int mtl = rth_add_isotropic(s, albedo_tex);
    if (mtl < 0) return mtl;
    rtg_node n{};
    n.type = RTG_CONSTANT_MEDIUM; n.material = mtl;
    n.f[0] = density;
    int32_t c = boundary;
    return push_node(s, n, &c, 1);
}

int rth_add_triangle_mesh(rth_scene *s, const rtg_vertex *verts, uint32_t n_verts, const uint32_t *indices,
                          uint32_t n_indices, const rtg_mesh_node *nodes, uint32_t n_nodes,
                          const float obj2world[16], int mtl) {
    if (!verts || !indices || n_indices % 3 != 0) { set_error("mesh: bad arrays"); return RT_E_INVALID; }
    for (uint32_t i = 0; i < n_indices; ++i)
        if (indices[i] >= n_verts) { set_error("mesh: index %u out of range", indices[i]); return RT_E_INVALID; }
    M4 o2w = m4_identity();
    if (obj2world) memcpy(o2w.a, obj2world, sizeof o2w.a);
    M4 w2o;
    if (!m4_invert(o2w, &w2o)) { set_error("mesh: obj2world is not invertible"); return RT_E_INVALID; }
    M4 n2w = m4_transpose(w2o);  // triangle_mesh.rs:49
    auto mesh = std::make_unique<Mesh>();
    mesh->verts.resize(n_verts);
    for (uint32_t i = 0; i < n_verts; ++i) {
        const rtg_vertex &v = verts[i];
        if (!obj2world) { mesh->verts[i] = v; continue; }   // already baked: TriangleMesh::vertices as the reference keeps them
        V3 p = m4_transform_point(o2w, v3(v.pos[0], v.pos[1], v.pos[2]));
        V3 nn = normalize(m4_transform_vector(n2w, v3(v.normal[0], v.normal[1], v.normal[2])));
        rtg_vertex o = v;
        o.pos[0] = p.x; o.pos[1] = p.y; o.pos[2] = p.z;
        o.normal[0] = nn.x; o.normal[1] = nn.y; o.normal[2] = nn.z;
        mesh->verts[i] = o;
    }
    mesh->indices.assign(indices, indices + n_indices);
    for (uint32_t i = 0; i < n_nodes; ++i) {  // triangle_mesh.rs:77: empty nodes are dropped
        if (nodes[i].index_count == 0) continue;
        if (nodes[i].index_begin % 3 || nodes[i].index_count % 3 || nodes[i].index_begin + nodes[i].index_count > n_indices) {
            set_error("mesh: node %u index range invalid", i);
            return RT_E_INVALID;
        }
        mesh->nodes.push_back(nodes[i]);
    }
    s->meshes.push_back(std::move(mesh));
    rtg_node n{};
    n.type = RTG_TRIANGLE_MESH; n.material = mtl;
    n.i[0] = (int32_t)s->meshes.size() - 1;
    return push_node(s, n, nullptr, 0);
}

int rth_set_roots(rth_scene *s, int world_list, int lights_list) {
    int n = (int)s->nodes.size();
    if (world_list < 0 || world_list >= n || lights_list < 0 || lights_list >= n ||
        s->nodes[world_list].type != RTG_LIST || s->nodes[lights_list].type != RTG_LIST) {
        set_error("roots must be RTG_LIST nodes");
        return RT_E_INVALID;
    }
    s->world_root = world_list;
    s->lights_root = lights_list;
    return RT_OK;
}

/* ------------------------------------------------------------------ scene builders */

namespace {

struct Builder {
    rth_scene *s;
    rts::Rng rng;
    float rr() { return rts::random_real(rng); }
    float rrr(float lo, float hi) { return rts::random_real_range(rng, lo, hi); }
    int lambert(float r, float g, float b) { return rth_add_lambertian(s, rth_add_solid_texture(s, r, g, b)); }
    int light(float r, float g, float b) { return rth_add_diffuse_light(s, rth_add_solid_texture(s, r, g, b)); }
};

// main.rs:106-199
int build_random_world(Builder &b) {
    rth_scene *s = b.s;
    std::vector<int32_t> world;
    int odd = rth_add_solid_texture(s, 0.2f, 0.3f, 0.1f);
    int even = rth_add_solid_texture(s, 0.9f, 0.9f, 0.9f);
    int ground = rth_add_lambertian(s, rth_add_checker_texture(s, odd, even, 4.0f));
    const float gc[3] = {0.0f, -1000.0f, 0.0f};
    world.push_back(rth_add_sphere(s, gc, 1000.0f, ground));

    const V3 ref = v3(std::sqrt(4.0f), std::sqrt(0.2f), std::sqrt(0.0f));  // main.rs:128 sqrt(Point(4,0.2,0))
    for (int a = -11; a < 11; ++a) {
        for (int bb = -11; bb < 11; ++bb) {
            float cx = ((float)a + 0.9f) + b.rr();
            float cz = (float)bb + 0.9f * b.rr();
            V3 center = v3(cx, 0.2f, cz);
            if (std::sqrt(length_squared(center - ref)) > 0.9f) {
                float choose = b.rr();
                const float c0[3] = {center.x, center.y, center.z};
                if (choose < 0.8f) {
                    float c1r = b.rrr(0.0f, 1.0f), c1g = b.rrr(0.0f, 1.0f), c1b = b.rrr(0.0f, 1.0f);
                    float c2r = b.rrr(0.0f, 1.0f), c2g = b.rrr(0.0f, 1.0f), c2b = b.rrr(0.0f, 1.0f);
                    int m = b.lambert(c1r * c2r, c1g * c2g, c1b * c2b);
                    float dy = b.rrr(0.0f, 0.5f);
                    const float c1[3] = {center.x + 0.0f, center.y + dy, center.z + 0.0f};
                    world.push_back(rth_add_moving_sphere(s, c0, c1, 0.0f, 1.0f, 0.2f, m));
                } else if (choose < 0.95f) {
                    float r = b.rrr(0.5f, 1.0f), g = b.rrr(0.5f, 1.0f), bl = b.rrr(0.5f, 1.0f);
                    float fuzz = b.rrr(0.0f, 0.5f);
                    world.push_back(rth_add_sphere(s, c0, 0.2f, rth_add_metal(s, r, g, bl, fuzz)));
                } else {
                    world.push_back(rth_add_sphere(s, c0, 0.2f, rth_add_dielectric(s, 1.5f)));
                }
            }
        }
    }
    const float s1[3] = {0.0f, 1.0f, 0.0f}, s2[3] = {-4.0f, 1.0f, 0.0f}, s3[3] = {4.0f, 1.0f, 0.0f};
    world.push_back(rth_add_sphere(s, s1, 1.0f, rth_add_dielectric(s, 1.5f)));
    world.push_back(rth_add_sphere(s, s2, 1.0f, b.lambert(0.4f, 0.2f, 0.1f)));
    world.push_back(rth_add_sphere(s, s3, 1.0f, rth_add_metal(s, 0.7f, 0.6f, 0.5f, 0.0f)));

    int32_t l = rth_add_rect(s, RTG_XZ_RECT, 213.0f, 343.0f, 227.0f, 332.0f, 554.0f, b.light(0, 0, 0));
    int wl = rth_add_list(s, world.data(), (int)world.size());
    int ll = rth_add_list(s, &l, 1);
    return rth_set_roots(s, wl, ll);
}

// The dummy `lights` list of the small scenes (main.rs:219-228 and its copies): a black XZRect far away
int finish_with_dummy_light(Builder &b, std::vector<int32_t> &world) {
    rth_scene *s = b.s;
    for (int32_t w : world) if (w < 0) return w;
    int32_t l = rth_add_rect(s, RTG_XZ_RECT, 213.0f, 343.0f, 227.0f, 332.0f, 554.0f, b.light(0, 0, 0));
    int wl = rth_add_list(s, world.data(), (int)world.size());
    int ll = rth_add_list(s, &l, 1);
    return rth_set_roots(s, wl, ll);
}

// main.rs:201-231
int build_two_spheres(Builder &b) {
    rth_scene *s = b.s;
    int checker = rth_add_lambertian(s, rth_add_checker_texture(s, rth_add_solid_texture(s, 0.2f, 0.3f, 0.1f), rth_add_solid_texture(s, 0.9f, 0.9f, 0.9f), 4.0f));
    const float lo[3] = {0, -10, 0}, hi[3] = {0, 10, 0};
    std::vector<int32_t> world = {rth_add_sphere(s, lo, 10, checker), rth_add_sphere(s, hi, 10, checker)};
    return finish_with_dummy_light(b, world);
}

// main.rs:233-286
int build_two_perlin_spheres(Builder &b, const rth_scene_assets *assets, uint32_t seed) {
    rth_scene *s = b.s;
    if (!assets) { set_error("PerlinSpheres needs assets->uvgrid01 and uvgrid03"); return RT_E_INVALID; }
    int noise_mtl = rth_add_lambertian(s, rth_add_noise_texture(s, 3.0f, seed));
    const rth_image &g1 = assets->uvgrid01, &g3 = assets->uvgrid03;
    const float c0[3] = {0, -1000, 0}, c1[3] = {4, 4, 0}, c2[3] = {-6, 6, 2};
    std::vector<int32_t> world;
    world.push_back(rth_add_sphere(s, c0, 1000, rth_add_lambertian(s, rth_add_image_texture(s, g1.width, g1.height, g1.rgba))));
    world.push_back(rth_add_sphere(s, c1, 3, rth_add_lambertian(s, rth_add_image_texture(s, g3.width, g3.height, g3.rgba))));
    world.push_back(rth_add_sphere(s, c2, 6, noise_mtl));
    world.push_back(rth_add_flip_face(s, rth_add_rect(s, RTG_XZ_RECT, -1000, 1000, -1000, 1000, 1000, b.light(1, 1, 1))));
    return finish_with_dummy_light(b, world);
}

// main.rs:288-313 (built by the reference but not reachable from its scene switch, main.rs:1311)
int build_textured_spheres(Builder &b, const rth_scene_assets *assets) {
    rth_scene *s = b.s;
    const rth_image *earth = assets ? &assets->earthmap : nullptr;
    int etex = rth_add_image_texture(s, earth ? earth->width : 0, earth ? earth->height : 0, earth ? earth->rgba : nullptr);
    const float c[3] = {0, 0, 0};
    std::vector<int32_t> world = {rth_add_sphere(s, c, 2, rth_add_lambertian(s, etex))};
    return finish_with_dummy_light(b, world);
}

// main.rs:315-360
int build_simple_light(Builder &b, uint32_t seed) {
    rth_scene *s = b.s;
    int noise_mtl = rth_add_lambertian(s, rth_add_noise_texture(s, 3.0f, seed));
    const float c0[3] = {0, -1000, 0}, c1[3] = {0, 2, 0}, c2[3] = {0, 8, 0};
    std::vector<int32_t> world;
    world.push_back(rth_add_sphere(s, c0, 1000, noise_mtl));
    world.push_back(rth_add_sphere(s, c1, 2, noise_mtl));
    world.push_back(rth_add_rect(s, RTG_XY_RECT, 3, 5, 1, 3, -2, b.light(4, 4, 4)));
    world.push_back(rth_add_sphere(s, c2, 2, b.light(4, 2, 0)));
    return finish_with_dummy_light(b, world);
}

// main.rs:362-522
int build_cornell_box(Builder &b) {
    rth_scene *s = b.s;
    int red = b.lambert(0.65f, 0.05f, 0.05f);
    int white = b.lambert(0.73f, 0.73f, 0.73f);
    int green = b.lambert(0.12f, 0.45f, 0.15f);
    std::vector<int32_t> world;
    world.push_back(rth_add_rect(s, RTG_YZ_RECT, 0, 555, 0, 555, 555, green));
    world.push_back(rth_add_rect(s, RTG_YZ_RECT, 0, 555, 0, 555, 0, red));
    world.push_back(rth_add_rect(s, RTG_XZ_RECT, 0, 555, 0, 555, 0, white));
    world.push_back(rth_add_rect(s, RTG_XZ_RECT, 0, 555, 0, 555, 555, white));
    world.push_back(rth_add_rect(s, RTG_XY_RECT, 0, 555, 0, 555, 555, white));
    world.push_back(rth_add_flip_face(s, rth_add_rect(s, RTG_XZ_RECT, 213, 343, 227, 332, 554, b.light(15, 15, 15))));
    const float p0[3] = {0, 0, 0}, p1[3] = {165, 330, 165}, off[3] = {265, 0, 295};
    world.push_back(rth_add_translate(s, rth_add_rotate_y(s, rth_add_block(s, p0, p1, white), 15.0f), off));
    const float gc[3] = {190, 90, 190};
    world.push_back(rth_add_sphere(s, gc, 90, rth_add_dielectric(s, 1.5f)));

    int lm = b.light(0, 0, 0);
    int32_t lights[2] = {rth_add_rect(s, RTG_XZ_RECT, 213, 343, 227, 332, 554, lm), rth_add_sphere(s, gc, 90, lm)};
    int wl = rth_add_list(s, world.data(), (int)world.size());
    int ll = rth_add_list(s, lights, 2);
    return rth_set_roots(s, wl, ll);
}

// main.rs:524-683, scene_cornell_box_smoke: defined in the reference but called by no `Scene` variant (it also returns no
// `lights`).  Kept for the one thing only it has: ConstantMedium over a boundary that is not a sphere (constant_medium.rs:52-58).
// `lights` = its light rectangle, as scene_cornell_box does; the light is not wrapped in FlipFace there (main.rs:626-633).
int build_cornell_box_smoke(Builder &b) {
    rth_scene *s = b.s;
    int red = b.lambert(0.65f, 0.05f, 0.05f);
    int white = b.lambert(0.73f, 0.73f, 0.73f);
    int green = b.lambert(0.12f, 0.45f, 0.15f);
    std::vector<int32_t> world;
    world.push_back(rth_add_rect(s, RTG_YZ_RECT, 0, 555, 0, 555, 555, green));
    world.push_back(rth_add_rect(s, RTG_YZ_RECT, 0, 555, 0, 555, 0, red));
    world.push_back(rth_add_rect(s, RTG_XZ_RECT, 0, 555, 0, 555, 0, white));
    world.push_back(rth_add_rect(s, RTG_XZ_RECT, 0, 555, 0, 555, 555, white));
    world.push_back(rth_add_rect(s, RTG_XY_RECT, 0, 555, 0, 555, 555, white));
    world.push_back(rth_add_rect(s, RTG_XZ_RECT, 113, 443, 127, 432, 554, b.light(7, 7, 7)));
    const float p0[3] = {0, 0, 0}, p1[3] = {165, 330, 165}, off1[3] = {265, 0, 295};
    int box1 = rth_add_translate(s, rth_add_rotate_y(s, rth_add_block(s, p0, p1, white), 15.0f), off1);
    world.push_back(rth_add_constant_medium(s, box1, rth_add_solid_texture(s, 0, 0, 0), 0.01f));     // from_colored_object, constant_medium.rs:28-38
    const float q1[3] = {165, 165, 165}, off2[3] = {130, 0, 65};
    int box2 = rth_add_translate(s, rth_add_rotate_y(s, rth_add_block(s, p0, q1, white), -18.0f), off2);
    world.push_back(rth_add_constant_medium(s, box2, rth_add_solid_texture(s, 1, 1, 1), 0.01f));
    int32_t l = rth_add_rect(s, RTG_XZ_RECT, 113, 443, 127, 432, 554, b.light(0, 0, 0));
    int wl = rth_add_list(s, world.data(), (int)world.size());
    int ll = rth_add_list(s, &l, 1);
    return rth_set_roots(s, wl, ll);
}

// main.rs:685-828
int build_final_chapter2(Builder &b, const rth_scene_assets *assets, uint32_t seed) {
    rth_scene *s = b.s;
    std::vector<int32_t> world;
    int white = b.lambert(0.73f, 0.73f, 0.73f);
    int ground = b.lambert(0.48f, 0.83f, 0.53f);
    std::vector<int32_t> boxes;
    for (int i = 0; i < 20; ++i)
        for (int j = 0; j < 20; ++j) {
            float w = 100.0f;
            float x0 = -1000.0f + (float)i * w, z0 = -1000.0f + (float)j * w, y0 = 0.0f;
            float x1 = x0 + w, y1 = b.rrr(1.0f, 101.0f), z1 = z0 + w;
            const float p0[3] = {x0, y0, z0}, p1[3] = {x1, y1, z1};
            boxes.push_back(rth_add_block(s, p0, p1, ground));
        }
    world.push_back(rth_add_bvh(s, boxes.data(), (int)boxes.size()));
    world.push_back(rth_add_flip_face(s, rth_add_rect(s, RTG_XZ_RECT, 123, 423, 147, 412, 554, b.light(17, 17, 17))));
    const float c0[3] = {400, 400, 200}, c1[3] = {400.0f + 30.0f, 400.0f + 0.0f, 200.0f + 0.0f};
    world.push_back(rth_add_moving_sphere(s, c0, c1, 0, 1, 50, b.lambert(0.7f, 0.3f, 0.1f)));
    const float g1[3] = {260, 150, 45}, m1[3] = {0, 150, 145}, bd[3] = {360, 150, 145}, fog[3] = {0, 0, 0};
    world.push_back(rth_add_sphere(s, g1, 50, rth_add_dielectric(s, 1.5f)));
    world.push_back(rth_add_sphere(s, m1, 50, rth_add_metal(s, 0.8f, 0.8f, 0.9f, 10.0f)));
    int boundary = rth_add_sphere(s, bd, 70, rth_add_dielectric(s, 1.5f));
    world.push_back(boundary);
    world.push_back(rth_add_constant_medium(s, boundary, rth_add_solid_texture(s, 0.2f, 0.4f, 0.9f), 0.2f));
    int fog_boundary = rth_add_sphere(s, fog, 5000, rth_add_dielectric(s, 1.5f));
    world.push_back(rth_add_constant_medium(s, fog_boundary, rth_add_solid_texture(s, 1, 1, 1), 0.0001f));
    const rth_image *earth = assets ? &assets->earthmap : nullptr;
    int etex = rth_add_image_texture(s, earth ? earth->width : 0, earth ? earth->height : 0, earth ? earth->rgba : nullptr);
    const float ec[3] = {400, 200, 400}, pc[3] = {220, 280, 300};
    world.push_back(rth_add_sphere(s, ec, 100, rth_add_lambertian(s, etex)));
    world.push_back(rth_add_sphere(s, pc, 80, rth_add_lambertian(s, rth_add_noise_texture(s, 0.1f, seed))));
    std::vector<int32_t> balls;
    for (int i = 0; i < 1000; ++i) {
        float x = 165.0f * b.rr(), y = 165.0f * b.rr(), z = 165.0f * b.rr();
        const float c[3] = {x, y, z};
        balls.push_back(rth_add_sphere(s, c, 10, white));
    }
    const float off[3] = {-100, 270, 395};
    world.push_back(rth_add_translate(s, rth_add_rotate_y(s, rth_add_bvh(s, balls.data(), (int)balls.size()), 15.0f), off));

    int32_t l = rth_add_rect(s, RTG_XZ_RECT, 213, 343, 227, 332, 554, b.light(0, 0, 0));
    int wl = rth_add_list(s, world.data(), (int)world.size());
    int ll = rth_add_list(s, &l, 1);
    return rth_set_roots(s, wl, ll);
}

// main.rs:969-1029.  mesh_grid = K > 1 places K x K copies (BASELINE config 4's
// "multi-million-triangle" variant; not in the reference): copy (i,j) is the teapot turned by a
// seeded angle about Y and moved to a 40-unit grid, baked through the same obj2world path.
int build_mesh_test(Builder &b, const rth_scene_assets *assets) {
    rth_scene *s = b.s;
    if (!assets || !assets->teapot) { set_error("MeshTest needs assets->teapot"); return RT_E_INVALID; }
    std::vector<int32_t> world;
    const rth_image &f = assets->uvgrid01, &t9 = assets->uvgrid09;
    int floor_mtl = rth_add_lambertian(s, rth_add_image_texture(s, f.width, f.height, f.rgba));
    world.push_back(rth_add_rect(s, RTG_XZ_RECT, -1000, 1000, -1000, 1000, 0, floor_mtl));
    world.push_back(rth_add_flip_face(s, rth_add_rect(s, RTG_XZ_RECT, -1000, 1000, -1000, 1000, 1000, b.light(1, 1, 1))));
    int teapot_mtl = rth_add_lambertian(s, rth_add_image_texture(s, t9.width, t9.height, t9.rgba));

    float q[4];
    quat_axis_angle(180.0f, v3(1, 0, 0), q);
    M4 r = quat_to_rotation_matrix(q[0], q[1], q[2], q[3]);
    const Mesh &m = assets->teapot->mesh;
    int K = assets->mesh_grid > 1 ? assets->mesh_grid : 1;
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            M4 xf;
            if (K == 1) {
                xf = m4_mul(m4_translate(v3(0, 0, 0)), r);  // main.rs:1006-1008: t * r
            } else {
                float qy[4];
                quat_axis_angle(360.0f * b.rr(), v3(0, 1, 0), qy);
                M4 ry = quat_to_rotation_matrix(qy[0], qy[1], qy[2], qy[3]);
                V3 t = v3(((float)i - 0.5f * (float)(K - 1)) * 40.0f, 0.0f, ((float)j - 0.5f * (float)(K - 1)) * 40.0f);
                xf = m4_mul(m4_translate(t), m4_mul(ry, r));
            }
            int n = rth_add_triangle_mesh(s, m.verts.data(), (uint32_t)m.verts.size(), m.indices.data(),
                                          (uint32_t)m.indices.size(), m.nodes.data(), (uint32_t)m.nodes.size(), xf.a, teapot_mtl);
            if (n < 0) return n;
            world.push_back(n);
        }
    int32_t l = rth_add_rect(s, RTG_XZ_RECT, -1000, 1000, -1000, 1000, 1000, b.light(0, 0, 0));
    int wl = rth_add_list(s, world.data(), (int)world.size());
    int ll = rth_add_list(s, &l, 1);
    return rth_set_roots(s, wl, ll);
}

// types.rs:188-231 ("Graphics Gems 2, pg. 378")
M4 random_rotation_matrix(Builder &bd) {
    float x1 = bd.rr(), x2 = bd.rr(), x3 = bd.rr();
    const float pi = 3.14159265358979323846f, two_pi = pi * 2.0f;
    float z = x1, phi = two_pi * x2, r = std::sqrt(1.0f - z * z), w = pi * x3;
    float sin_phi, cos_phi, sin_w, cos_w;
    pm::sincosf_(phi, &sin_phi, &cos_phi);
    pm::sincosf_(w, &sin_w, &cos_w);
    float a = cos_w, b = sin_w * cos_phi * r, c = sin_w * sin_phi * r, d = sin_w * z;
    M4 m = m4_identity();
    m.at(0, 0) = 1.0f - 2.0f * (c * c + d * d); m.at(0, 1) = 2.0f * (b * c + a * d);         m.at(0, 2) = 2.0f * (b * d - a * c);
    m.at(1, 0) = 2.0f * (b * c - a * d);        m.at(1, 1) = 1.0f - 2.0f * (b * b + d * d);  m.at(1, 2) = 2.0f * (c * d + a * b);
    m.at(2, 0) = 2.0f * (b * d + a * c);        m.at(2, 1) = 2.0f * (c * d - a * b);         m.at(2, 2) = 1.0f - 2.0f * (b * b + c * c);
    return m;
}

M4 rotation(float angle_deg, V3 axis) {   // quat::to_rotation_matrix(Quat::axis_angle(angle, axis))
    float q[4];
    quat_axis_angle(angle_deg, axis, q);
    return quat_to_rotation_matrix(q[0], q[1], q[2], q[3]);
}

int image_tex(rth_scene *s, const rth_image &im) { return rth_add_image_texture(s, im.width, im.height, im.rgba); }

// The floor, the sky light and the `lights` list shared by scene_geometric_primitives, scene_mesh and scene_blocks
void floor_and_sky(Builder &b, int floor_tex, std::vector<int32_t> &world) {
    rth_scene *s = b.s;
    world.push_back(rth_add_rect(s, RTG_XZ_RECT, -1000, 1000, -1000, 1000, 0, rth_add_lambertian(s, floor_tex)));
    world.push_back(rth_add_flip_face(s, rth_add_rect(s, RTG_XZ_RECT, -1000, 1000, -1000, 1000, 1000, b.light(1, 1, 1))));
}
int finish_with_sky_light(Builder &b, std::vector<int32_t> &world) {
    rth_scene *s = b.s;
    for (int32_t w : world) if (w < 0) return w;
    int32_t l = rth_add_rect(s, RTG_XZ_RECT, -1000, 1000, -1000, 1000, 1000, b.light(0, 0, 0));
    int wl = rth_add_list(s, world.data(), (int)world.size());
    int ll = rth_add_list(s, &l, 1);
    return rth_set_roots(s, wl, ll);
}

// main.rs:830-967
int build_geometric_primitives(Builder &b, const rth_scene_assets *assets) {
    rth_scene *s = b.s;
    if (!assets) { set_error("GeometricPrimitives needs the uv-grid textures"); return RT_E_INVALID; }
    const int grid01 = image_tex(s, assets->uvgrid01), grid03 = image_tex(s, assets->uvgrid03), grid08 = image_tex(s, assets->uvgrid08),
              grid10 = image_tex(s, assets->uvgrid10);
    std::vector<int32_t> world;
    floor_and_sky(b, grid01, world);
    const V3 ux = v3(1, 0, 0), uy = v3(0, 1, 0), uz = v3(0, 0, 1);
    const float lo[3] = {-0.5f, -0.5f, -0.5f}, hi[3] = {0.5f, 0.5f, 0.5f};   // Block::unit_cube, block.rs:18-20
    const float pi = 3.14159265358979323846f, two_pi = pi * 2.0f, to_rad = pi / 180.0f;

    // the reference shares one Arc<Block> between the two Transforms; here each gets its own node so that
    // every leaf has one primitive id (include/rt_scene_graph.h)
    int block_mtl = rth_add_lambertian(s, grid03);
    {
        M4 r = rotation(180.0f, uy), t = m4_translate(v3(65, 65, 0)), sc = m4_scale(v3(35, 35, 35));
        world.push_back(rth_add_transform(s, rth_add_block(s, lo, hi, block_mtl), m4_mul(m4_mul(t, r), sc).a));
    }
    {
        M4 r = random_rotation_matrix(b), sc = m4_scale(v3(35, 35, 35)), t = m4_translate(v3(-65, 65, 0));
        world.push_back(rth_add_transform(s, rth_add_block(s, lo, hi, block_mtl), m4_mul(m4_mul(t, r), sc).a));
    }
    {   // Cylinder::unit(None): radius 1, z in [-0.5, 0.5], full turn
        int cyl = rth_add_cylinder(s, 1.0f, -0.5f, 0.5f, two_pi, rth_add_lambertian(s, grid01));
        M4 r = rotation(45.0f, uy), t = m4_translate(v3(40, 30, 30)), sc = m4_scale(v3(15, 15, 65));
        world.push_back(rth_add_transform(s, cyl, m4_mul(m4_mul(t, r), sc).a));
    }
    {   // partial cylinder
        int cyl = rth_add_cylinder(s, 1.0f, -0.5f, 0.5f, pi, rth_add_lambertian(s, grid08));
        M4 r = rotation(-45.0f, uy), t = m4_translate(v3(-40, 30, 30)), sc = m4_scale(v3(15, 15, 65));
        world.push_back(rth_add_transform(s, cyl, m4_mul(m4_mul(t, r), sc).a));
    }
    int cone_mtl = rth_add_lambertian(s, grid03);
    const M4 s25 = m4_scale(v3(25, 25, 25));
    {
        M4 r = rotation(90.0f, ux), t = m4_translate(v3(-25, 25, 50));
        world.push_back(rth_add_transform(s, rth_add_cone(s, 1.0f, 1.0f, two_pi, cone_mtl), m4_mul(m4_mul(t, r), s25).a));
    }
    {
        M4 r = m4_mul(rotation(-90.0f, ux), rotation(-90.0f, uz)), t = m4_translate(v3(25, 0, 50));
        world.push_back(rth_add_transform(s, rth_add_cone(s, 1.0f, 1.0f, 330.0f * to_rad, cone_mtl), m4_mul(m4_mul(t, r), s25).a));
    }
    int hyp_mtl = rth_add_lambertian(s, grid10);
    {
        M4 r = rotation(90.0f, ux), t = m4_translate(v3(-0.0f, 25, 100));
        world.push_back(rth_add_transform(s, rth_add_paraboloid(s, 0.5f, -0.5f, 1.0f, two_pi, hyp_mtl), m4_mul(m4_mul(t, r), s25).a));
    }
    {
        M4 r = m4_mul(rotation(-120.0f, uy), rotation(90.0f, ux)), t = m4_translate(v3(-30, 25, 100));
        world.push_back(rth_add_transform(s, rth_add_paraboloid(s, 0.5f, -0.5f, 1.0f, 300.0f * to_rad, hyp_mtl), m4_mul(m4_mul(t, r), s25).a));
    }
    return finish_with_sky_light(b, world);
}

// main.rs:1031-1133.  Colours: math/src/color_palette.rs material_design, through From<u32> (math/src/colors.rs:34-41)
int build_blocks(Builder &b, const rth_scene_assets *assets) {
    rth_scene *s = b.s;
    if (!assets) { set_error("Blocks needs assets->uvgrid01"); return RT_E_INVALID; }
    std::vector<int32_t> world;
    floor_and_sky(b, image_tex(s, assets->uvgrid01), world);
    auto checkers = [&](uint32_t odd, uint32_t even) {
        auto tex = [&](uint32_t c) {
            float rgba[4];
            rth_color_from_u32(c, rgba);
            return rth_add_solid_texture(s, rgba[0], rgba[1], rgba[2]);
        };
        int o = tex(odd), e = tex(even);
        return rth_add_lambertian(s, rth_add_checker_texture(s, o, e, 4.0f));
    };
    const uint32_t YELLOW = 0xffffeb3bu, BLACK = 0xff000000u, GREEN = 0xff4caf50u, RED = 0xfff44336u;
    const float lo[3] = {-0.5f, -0.5f, -0.5f}, hi[3] = {0.5f, 0.5f, 0.5f};
    const V3 uy = v3(0, 1, 0);
    const M4 sc = m4_scale(v3(5, 10, 5));
    {
        M4 r = rotation(20.0f, uy), t = m4_translate(v3(-7, 5, 10));
        world.push_back(rth_add_transform(s, rth_add_block(s, lo, hi, checkers(YELLOW, BLACK)), m4_mul(m4_mul(t, r), sc).a));
    }
    {
        M4 t = m4_translate(v3(0, 5, 0));
        world.push_back(rth_add_transform(s, rth_add_block(s, lo, hi, checkers(GREEN, BLACK)), m4_mul(t, sc).a));
    }
    {
        M4 r = rotation(-20.0f, uy), t = m4_translate(v3(7, 5, 10));
        world.push_back(rth_add_transform(s, rth_add_block(s, lo, hi, checkers(RED, BLACK)), m4_mul(m4_mul(t, r), sc).a));
    }
    return finish_with_sky_light(b, world);
}

}  // namespace

int rth_build_scene(rth_scene *s, const char *name, uint32_t seed, const rth_scene_assets *assets) {
    if (!s || !name) { set_error("null argument"); return RT_E_INVALID; }
    if (!s->nodes.empty()) { set_error("scene is not empty"); return RT_E_INVALID; }
    Builder b{s, {}};
    rts::host_rng(b.rng, seed, 0x5ce9e000u);
    std::string n(name);
    if (n == "RandomWorld") return build_random_world(b);
    if (n == "CornellBox") return build_cornell_box(b);
    if (n == "CornellBoxSmoke") return build_cornell_box_smoke(b);
    if (n == "Chapter2Final") return build_final_chapter2(b, assets, seed);
    if (n == "MeshTest") return build_mesh_test(b, assets);
    if (n == "TwoSpheres") return build_two_spheres(b);
    if (n == "PerlinSpheres") return build_two_perlin_spheres(b, assets, seed);
    if (n == "TexturedSpheres") return build_textured_spheres(b, assets);
    if (n == "SimpleLight") return build_simple_light(b, seed);
    if (n == "GeometricPrimitives") return build_geometric_primitives(b, assets);
    if (n == "Blocks") return build_blocks(b, assets);
    set_error("scene '%s' is not covered by this backend", name);
    return RT_E_UNSUPPORTED_PRIMITIVE;
}

}  // extern "C"
